# coding: utf-8
"""Where do the HIP sampler and its numpy restatement part?  (tools; GPU box)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffudf_amd import synth
from diffudf_amd.dataset import PointCloud
from oracle import sampler_oracle as SO

beetle = os.path.join(ROOT, "tests", "golden", "beetle")
ds = PointCloud(beetle, 3000, [0.333, 0.666], 1, device="cuda:0", onlyPCloud=True, seed=11, surfacePoints=5000)
pos, pn = ds.pc_pos.cpu().numpy(), ds.pc_nrm.cpu().numpy()
x, nrm, sdf = ds.sample(2)
xo, no, so = SO.sample_batch(None, pos, pn, 999, 999, 999, seed=11, step=2)
off_h = sdf[1998:].cpu().numpy(); off_o = so[1998:, 0]
d = off_h != off_o
print("near |offset| (cloud-only mode): differ", int(d.sum()), "of", d.size, "max rel", np.max(np.abs(off_h - off_o) / off_o))
base = 1000 * 2
u1 = synth.uniform01(11, base + 405, 0, 999); u2 = synth.uniform01(11, base + 406, 0, 999)
r = np.sqrt(-2.0 * np.log1p(-u1)); c = np.cos(2.0 * np.pi * u2)
i = np.flatnonzero(d)[:5]
print("examples u1,u2,r,c,off64,off32_oracle,off32_hip:")
for k in i:
    print(u1[k], u2[k], r[k], c[k], 0.01 * r[k] * c[k], off_o[k], off_h[k])
dx = (x.cpu().numpy() != xo).any(axis=1)
print("x rows differing by stratum: on", int(dx[:999].sum()), "far", int(dx[999:1998].sum()), "near", int(dx[1998:].sum()))
