# coding: utf-8
"""Differential quantities the sphere tracer derives at ray hits — the query half of reference
src/render_st.py:42-65 (BASELINE config 4).  The marching loop itself (`propagate_rays`, `grad_descent`, :136-172)
runs on the device too; ray set-up and shading (numpy glue) stay with the caller."""
import numpy as np
import torch
import weakref

from . import hip_ops
from .diff_operators import _source, gradient
from ._lib import DudfError


def compute_grad(inputs, outputs):
    """reference src/render_st.py:64-65"""
    return gradient(outputs, inputs)


def compute_normals_and_cd(inputs, outputs):
    """(pred_normals (1,N,3), principal directions (1,N,3,2) on the CPU) — reference src/render_st.py:57-62:
    eigh of the Hessian, normal = eigenvector of the largest eigenvalue, the other two as curvature directions."""
    model, coords = _source(outputs, inputs)
    x2 = coords.detach().reshape(-1, 3)
    _, _, _, _, V = hip_ops.query_frame(model.hip_cfg, model.flat_parameters(), x2)
    lead = coords.shape[:-1]
    normals = V[:, :, 2].reshape(lead + (3,))
    from .diff_operators import tag_field
    normals = tag_field(normals, "eig_normal", weakref.ref(model), coords)   # compute_curvature(inputs, normals) finds its way back
    return normals, V[:, :, :2].reshape(lead + (3, 2)).detach().cpu()


def compute_curvature(inputs, normals, curvature='mean', device=None):
    """reference src/render_st.py:42-55: shape operator = jacobian(normals, inputs) — third derivatives of f, obtained
    here from third-order Taylor jets along combinations of the Hessian's eigenvectors (csrc/dudf_sweep.hip
    SWEEP_FWD_J) instead of autograd through eigh.  'mean' -> trace/2, 'gaussian' -> -det [[J, n],[n^T, 0]];
    (1,N,1) CPU tensors like the reference; anything else -> None.  The sign convention of `normals` is the one
    `compute_normals_and_cd` returned (the same eigh), so the caller's re-orientation (:104-108) applies unchanged."""
    if curvature not in ('mean', 'gaussian'):
        return None
    model, coords = _source(normals, inputs)
    x2 = coords.detach().reshape(-1, 3)
    _, _, mean, gauss, _ = hip_ops.query_curvature(model.hip_cfg, model.flat_parameters(), x2,
                                                   want_shape=(curvature == 'gaussian'))
    out = mean if curvature == 'mean' else gauss
    return out.detach().cpu()[None, ..., None]


def _device_of(model, device):
    return torch.device(device) if device is not None else model.flat_parameters().device


def propagate_rays(model, rays, t0, mask_rays, network_config, rendering_config, device=None):
    """reference src/render_st.py:136-161, same signature and in-place contract: `t0` (M,3) float64 and `mask_rays` (M,)
    bool numpy arrays are updated, the bool hit mask is returned.  One upload, `max_iterations` marching iterations on
    the GPU (a host round trip every 8 of them for the `np.sum(mask_rays) > 0` test, not three copies per iteration),
    one download."""
    dev = _device_of(model, device)
    d_rays = torch.from_numpy(np.ascontiguousarray(rays, dtype=np.float64)).to(dev)
    d_t0 = torch.from_numpy(np.ascontiguousarray(t0, dtype=np.float64)).to(dev)
    d_mask = torch.from_numpy(np.ascontiguousarray(mask_rays).astype(np.uint8)).to(dev)
    with torch.cuda.device(dev):
        hits, _ = hip_ops.trace_rays(model.hip_cfg, model.flat_parameters(), d_rays, d_t0, d_mask,
                                     network_config['gt_mode'], network_config['alpha'],
                                     rendering_config['surface_threshold'], rendering_config['max_iterations'])
    t0[...] = d_t0.cpu().numpy()
    mask_rays[...] = d_mask.cpu().numpy().astype(bool)
    hits = hits.cpu().numpy().astype(bool)
    if np.sum(hits) == 0:
        raise ValueError(f"Ray tracing did not converge in {rendering_config['max_iterations']} iterations to any point at "
                         f"distance {rendering_config['surface_threshold']} or lower from surface.")
    return hits


def grad_descent(model, t0, mask_rays, network_config, rendering_config, device=None):
    """reference src/render_st.py:163-172: `gd_steps` projection steps of the hit points, in place on `t0`."""
    if rendering_config['gd_steps'] <= 0:
        return
    dev = _device_of(model, device)
    d_t0 = torch.from_numpy(np.ascontiguousarray(t0, dtype=np.float64)).to(dev)
    d_hits = torch.from_numpy(np.ascontiguousarray(mask_rays).astype(np.uint8)).to(dev)
    with torch.cuda.device(dev):
        hip_ops.descend_rays(model.hip_cfg, model.flat_parameters(), d_t0, d_hits, network_config['gt_mode'],
                             network_config['alpha'], rendering_config['gd_steps'])
    t0[...] = d_t0.cpu().numpy()
