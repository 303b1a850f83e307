#!/usr/bin/env python
# coding: utf-8
"""Field slice of a trained network — the query half of reference generate_df.py:50-148 (`generate_df`):

    python generate_df.py path/to/mesh.obj path/to/model.pth path/to/output/ [-d 0] [-w0 30] [-w 512] [-t 1e-3]
                          [--gt_mode siren] [-a 1]                       (the reference's CLI, generate_df.py:248-266)

`field_slice(model, options)` samples the reference's plane (:68-79: `width`^2 points, first coordinate fixed at 0, the other
two running from +1 to -1), evaluates value, gradient and Hessian with the HIP query kernels (`src.evaluate.evaluate`,
one call), and derives what the reference draws: the predicted value, |grad f| and the normal map — normalised gradient,
replaced by the sign-aligned top Hessian eigenvector where the RAW gradient is shorter than 0.04 (:84-100; note the
raw norm, unlike `extract_fields`), third component made non-negative, mapped to RGB (:105-107).

This is the parity artefact of SURVEY.md §8(f) ("learned field samples and grad f on a fixed grid"): the three arrays
are checked against the reference's own outputs in tests/golden/g9_slice.npz.  The ground-truth panels of the figure
(:109-127) need open3d's distance queries; they are drawn only when a mesh is given AND the caller passes the
ground-truth distances (`gt_distances`), otherwise the figure holds the two predicted panels.
"""
import argparse

import numpy as np
import torch

from src.evaluate import evaluate
from src.model import SIREN
from src.util import normalize


def slice_samples(width, bordes=(1, -1), ejeplano=(2, 1, 0), offsetplano=0.0):
    """(width^2, 3) sample plane exactly as reference generate_df.py:68-79 builds it."""
    ranges = np.linspace(bordes[0], bordes[1], width)
    i_1, i_2 = np.meshgrid(ranges, ranges)
    planes = np.array([np.expand_dims(i_1, 2), np.expand_dims(i_2, 2), np.expand_dims(np.ones_like(i_1) * offsetplano, 2)])
    return np.concatenate(np.concatenate(planes[list(ejeplano)], axis=2), axis=0)


def field_slice(model, options):
    """dict(samples, pred_distances (S,1), pred_grad_norm (S,1), normals (S,3), grad_map (W,W,3) uint8)."""
    width = options['width']
    samples = slice_samples(width)
    n = width * width
    gradients = np.zeros((n, 3)); hessians = np.zeros((n, 3, 3))
    dev = torch.device(options.get('device', 'cuda:0'))
    pred = evaluate(model, samples, device=dev, gradients=gradients, hessians=hessians)
    gnorm = np.linalg.norm(gradients, axis=1).reshape((n, 1))
    g = normalize(gradients)
    # top eigenvector of the Hessian's lower triangle, as torch.linalg.eigh reads it (reference :86-87)
    lam, V = np.linalg.eigh(hessians, UPLO='L')
    pn = V[..., 2]
    pn = np.where(np.sum(g * pn, axis=-1)[..., None] < 0, -1.0, 1.0) * pn
    normals = np.where(np.concatenate([gnorm, gnorm, gnorm], axis=-1) < 0.04, pn, g)
    normals = normals * np.hstack([np.ones((n, 2)), np.sign(normals[:, 2]).reshape((n, 1))])
    grad_map = (((normals + np.ones_like(normals)) / 2).reshape(width, width, 3) * 255).astype(np.uint8)
    return {"samples": samples, "pred_distances": pred, "pred_grad_norm": gnorm, "normals": normals, "grad_map": grad_map}


def generate_df(model_path, mesh_path, output_path, options, gt_distances=None):
    """Reference signature (generate_df.py:50) + optional ground-truth distances for the two upper panels."""
    model = SIREN(n_in_features=3, n_out_features=1, hidden_layer_config=options['hidden_layer_nodes'],
                  w0=options['weight0'], ww=None, activation=options.get('activation', 'sine'))
    model.load_state_dict(torch.load(model_path, weights_only=True))
    model.to(torch.device(options.get('device', 'cuda:0')))
    out = field_slice(model, options)
    np.savez_compressed(output_path + 'field_slice.npz', **out)
    from PIL import Image
    Image.fromarray(out["grad_map"]).save(output_path + 'pred_grad.png', 'PNG')
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:                                   # pragma: no cover
        return out
    w = options['width']
    panels = [("Predicted value slice", np.clip(out["pred_distances"], None, 1.5)),
              (r"$\|\nabla f\|$", np.clip(out["pred_grad_norm"], None, 1.5))]
    if gt_distances is not None:
        d = np.asarray(gt_distances).reshape(-1, 1)
        if options['gt_mode'] == 'tanh':
            t = np.tanh(options['alpha'] * d)
            panels = [("Ground truth slice", np.clip(d * t, None, 1.5)),
                      ("Ground truth gradient norm", np.clip(t + options['alpha'] * d * (1 - t ** 2), None, 1.5))] + panels
    fig, axes = plt.subplots(nrows=len(panels) // 2, ncols=2, figsize=(10, 4.5 * (len(panels) // 2)), dpi=200, squeeze=False)
    for ax, (title, img) in zip(axes.flat, panels):
        pos = ax.imshow(img.reshape(w, w), cmap='bwr_r', interpolation='none', vmin=-1.5, vmax=1.5)
        ax.contour(np.ma.masked_outside(img, -options.get('surf_thresh', 0.01), options.get('surf_thresh', 0.01)).reshape(w, w),
                   colors='black', linewidths=0.5)
        ax.set_title(title); ax.set_xticks([]); ax.set_yticks([])
    fig.colorbar(pos, ax=axes.ravel().tolist(), shrink=0.8)
    fig.savefig(output_path + 'distance_fields.png')
    plt.close(fig)
    return out


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description='Field slice of a trained model')
    parser.add_argument('mesh_path', metavar='path/to/mesh.obj', type=str, help='path to input preprocessed mesh (ground-truth panels)')
    parser.add_argument('model_path', metavar='path/to/pth', type=str, help='path to input model')
    parser.add_argument('output_path', metavar='path/to/output/', type=str, help='path to output folder')
    parser.add_argument('-d', '--device', type=int, default=0, help='torch device')
    parser.add_argument('-w0', '--weight0', type=float, default=30, help='w0 parameter of SIREN')
    parser.add_argument('-w', '--width', type=int, default=512, help='width of generated image')
    parser.add_argument('-t', '--surf_thresh', type=float, default=1e-3, help='on surface threshold')
    parser.add_argument('--gt_mode', type=str, default='siren', help='ground truth function')
    parser.add_argument('-a', '--alpha', type=float, default=1, help='alpha for ground truth')
    args = parser.parse_args()
    d = vars(args)
    d['hidden_layer_nodes'] = [256] * 8
    d['activation'] = 'sine'
    d['device'] = f"cuda:{args.device}"
    generate_df(args.model_path, args.mesh_path, args.output_path, d)
