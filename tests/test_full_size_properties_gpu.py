# coding: utf-8
"""GPU: size-independent properties at BASELINE.json's full sizes (the oracle-direct comparison of the same step at the
same sizes is tests/test_full_size_oracle_gpu.py).

  * per-point independence: values / gradients of a 100 000-point query equal the oracle's on a random subset;
  * shard additivity (the multi-GPU invariant): loss terms and d(theta) of the full batch equal the sums over uneven
    shards computed with the same n_global — 8x256 at 100 k points and 8x512 at 125 k (config 2's per-GPU share);
  * linearity of the backward in the upstream cotangent;
  * repeatability (float atomics reorder sums: agreement to rounding, not bit-for-bit)."""
import numpy as np
import pytest
import torch

from diffudf_amd import synth
from oracle import dudf_oracle as O

pytestmark = pytest.mark.gpu
W = [1e4, 1e4, 0.0, 1e3]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _setup(hidden, n, seed):
    from diffudf_amd import hip_ops as hip
    P = synth.siren_params(hidden, seed=seed)
    th = torch.from_numpy(synth.flatten_params(P)).cuda()
    x, nrm, sdf = [torch.from_numpy(a).cuda() for a in synth.training_batch(n, seed=seed + 1, step=0)]
    return hip, hip.make_cfg(hidden), P, th, x, nrm, sdf.reshape(-1)


def _loss_grad(hip, cfg, th, x, nrm, sdf, n_global, cot=None):
    ws = hip.workspace_for(cfg, x.shape[0], th.device)
    terms = hip.loss_forward(cfg, 0, th, x, nrm, sdf, n_global, W, 100.0, ws).clone()
    c = torch.ones(4, device="cuda") if cot is None else cot
    g = hip.loss_backward(cfg, 0, th, x, nrm, sdf, n_global, W, 100.0, c, None, ws).clone()
    return terms.double().cpu().numpy(), g.double().cpu().numpy()


def test_query_is_per_point_at_full_size():
    hip, cfg, P, th, x, nrm, sdf = _setup([256] * 8, 100000, 123)
    f, g = hip.query(cfg, th, x)
    idx = np.random.default_rng(0).choice(100000, 256, replace=False)
    # + every column of three one-group passes (the last pass of a workgroup that holds 25 groups: all eight waves share it)
    ng = (100000 + 127) // 128 * 8
    lone = [b for b in range(256) if ((b + 1) * ng // 256 - b * ng // 256) % 8 == 1][:3]
    extra = np.concatenate([np.arange(16) + 16 * ((b + 1) * ng // 256 - 1) for b in lone])
    idx = np.unique(np.concatenate([idx, extra[extra < 100000]]))
    P64 = [(a.astype(np.float64), b.astype(np.float64)) for a, b in P]
    yo, cache = O.forward(P64, x[idx].double().cpu().numpy())
    go = O.input_gradient(P64, cache)[0]
    assert rel(f[idx].cpu().numpy(), yo) < 5e-6
    assert rel(g[idx].cpu().numpy(), go) < 2e-5
    # the same points queried alone give the same numbers (no dependence on the batch they travel in)
    f0, _ = hip.query(cfg, th, x, want_grad=False)                   # the value-only variant of the forward sweep
    assert rel(f0[idx].cpu().numpy(), yo) < 5e-6
    f2, g2 = hip.query(cfg, th, x[idx].contiguous())
    assert rel(f2.cpu().numpy(), f[idx].cpu().numpy()) < 1e-6 and rel(g2.cpu().numpy(), g[idx].cpu().numpy()) < 1e-6


@pytest.mark.parametrize("hidden,n", [([256] * 8, 100000), ([512] * 8, 125000)])
def test_shard_additivity_linearity_repeatability(hidden, n):
    hip, cfg, P, th, x, nrm, sdf = _setup(hidden, n, 7)
    t_full, g_full = _loss_grad(hip, cfg, th, x, nrm, sdf, n)
    cuts = [0, n // 3 + 17, n // 2 + 5, n]                       # three uneven shards
    t_sum, g_sum = np.zeros(4), np.zeros_like(g_full)
    for a, b in zip(cuts[:-1], cuts[1:]):
        t, g = _loss_grad(hip, cfg, th, x[a:b].contiguous(), nrm[a:b].contiguous(), sdf[a:b].contiguous(), n)
        t_sum += t; g_sum += g
    assert rel(t_sum, t_full) < 2e-6
    assert rel(g_sum, g_full) < 2e-5
    # linear in the upstream cotangent of the four terms
    cot = torch.tensor([2.0, -0.5, 3.0, 0.25], device="cuda")
    _, g_c = _loss_grad(hip, cfg, th, x, nrm, sdf, n, cot)
    parts = []
    for k in range(4):
        e = torch.zeros(4, device="cuda"); e[k] = 1.0
        parts.append(_loss_grad(hip, cfg, th, x, nrm, sdf, n, e)[1])
    g_lin = sum(float(cot[k]) * parts[k] for k in range(4))
    assert rel(g_c, g_lin) < 2e-5
    # twice the same call: same numbers up to the reordering of float atomics
    t2, g2 = _loss_grad(hip, cfg, th, x, nrm, sdf, n)
    assert rel(t2, t_full) < 1e-6 and rel(g2, g_full) < 1e-5
    assert np.isfinite(g_full).all() and np.abs(g_full).max() > 0
