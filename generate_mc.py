#!/usr/bin/env python
# coding: utf-8
"""Mesh extraction from a trained network — reference generate_mc.py:9-67 `generate_mc`, with what this build's hot
path covers: `extract_fields` (value + gradient on the N^3 grid) feeding the extractors.

    python generate_mc.py <config.json>            keys as the reference's configs/mc_cfg.json (+ optional "luts_path")

algorithms 'meshudf' (the reference's default; SURVEY.md §8(f) row 4: the Lewiner-table marching cubes of
src/marching_cubes as host C++), 'cap' (row 2, on the device) and 'both' (what train.py asks for with gt_mode 'tanh').
MeshUDF needs the Lewiner look-up tables, which are an input, not part of this package: `luts=` / config key "luts_path"
(an .npz or the reference's `_marching_cubes_lewiner_luts.py`), $DUDF_MESHUDF_LUTS, or that module on sys.path (inside a
reference checkout: `sys.path.append('src/marching_cubes')`) — `diffudf_amd.marching_cubes.load_luts`.
'siren' (skimage's marching cubes on an SDF) is outside the build: it raises."""
import json
import sys

import torch

from src.model import SIREN
from src.render_mc import extract_fields, extract_mesh_CAP, extract_mesh_MESHUDF
from diffudf_amd.marching_cubes import MeshUDFError


def generate_mc(model, gt_mode, device, N, output_path, alpha=None, algorithm='meshudf', from_file=None, luts=None):
    """`luts`: the Lewiner tables for the MeshUDF half (dict, path or None = look them up; see the module docstring)."""
    if from_file is not None:
        model = SIREN(n_in_features=3, n_out_features=1, hidden_layer_config=from_file["hidden_layer_nodes"],
                      w0=from_file["w0"], ww=from_file.get("ww"), activation=from_file.get('activation', 'sine'))
        model.load_state_dict(torch.load(from_file["model_path"], weights_only=True))
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    model.to(dev)
    if algorithm in ('cap', 'both', 'meshudf'):
        u, g = extract_fields(model, torch.Tensor([[]]).to(dev), N, gt_mode, dev, alpha)
        dot = output_path.rfind('.')
        if algorithm == 'meshudf':
            _, _, mesh = extract_mesh_MESHUDF(u, g, dev, smooth_borders=True, luts=luts)
            mesh.export(output_path)
            print(f'Saved to {output_path}')
            return mesh
        mesh = extract_mesh_CAP(u, g, N)                       # device tensors straight through: no host round trip
        if algorithm == 'cap':
            mesh.export(output_path)
            print(f'Saved to {output_path}')
            return mesh
        path_mu, path_cap = output_path[:dot] + '_MU' + output_path[dot:], output_path[:dot] + '_CAP' + output_path[dot:]
        mesh.export(path_cap)
        try:
            _, _, mesh_mu = extract_mesh_MESHUDF(u, g, dev, smooth_borders=True, luts=luts)
        except MeshUDFError as e:                              # no look-up tables: the CAP half is still written, and says so
            print(f'Saved to {path_cap} (MeshUDF half skipped: {e})')
            return None, mesh
        mesh_mu.export(path_mu)
        print(f'Saved to {path_mu}, {path_cap}')
        return mesh_mu, mesh
    raise ValueError(f"algorithm '{algorithm}' is not part of this build ('cap', 'meshudf', 'both')")


if __name__ == "__main__":
    cfg = json.load(open(sys.argv[1]))
    generate_mc(None, cfg["gt_mode"], cfg.get("device", 0), cfg["nsamples"], cfg["output_path"], cfg.get("alpha"),
                cfg["algorithm"], from_file={"w0": cfg["w0"], "model_path": cfg["model_path"],
                                             "hidden_layer_nodes": cfg["hidden_layer_nodes"],
                                             "activation": cfg.get("activation", "sine"), "ww": cfg.get("ww")},
                luts=cfg.get("luts_path"))
