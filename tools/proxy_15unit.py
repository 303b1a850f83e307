# coding: utf-8
"""What would the 15-unit dataflow of DESIGN r03 §9.1 cost?  A measured proxy from kernels that exist.

15-unit plan: the reverse sweep keeps only df/dx (reads C, stores nothing = today's QUERY variant of the reverse sweep), the
adjoint forward sweep stores Q_l instead of e_l and no longer reads r_l (same store count, one load less), and the adjoint
reverse sweep carries the a_l recurrence as a second column per point (twice the columns through the same matrices, reads
c, s, Q, stores q, zbar).  Proxy for that last kernel: today's adjoint reverse sweep on TWICE the columns — the same matmul,
tail and store work per column, one load per column less than the real thing would need.

    python tools/proxy_15unit.py        ->  per-kernel ms at 100 000 points and the two step sums
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffudf_amd import _lib, hip_ops as hip, synth          # noqa: E402


def profiled(fn, reps=8):
    lib = _lib.load()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.dudf_profile_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    lib.dudf_profile_enable(0)
    buf = ctypes.create_string_buffer(4096)
    lib.dudf_profile_dump(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        out[name] = float(tot) / reps
    return out


def main():
    hidden, n = [256] * 8, 100000
    cfg = hip.make_cfg(hidden)
    th = torch.from_numpy(synth.flatten_params(synth.siren_params(hidden, seed=123))).cuda()
    W = [1e4, 1e4, 0.0, 1e3]
    res = {}
    for tag, m in (("N", n), ("2N", 2 * n)):
        x, nrm, sdf = [torch.from_numpy(a).cuda() for a in synth.training_batch(m, seed=124)]
        sdf = sdf.reshape(-1)
        ws = hip.workspace_for(cfg, m, "cuda")
        ones = torch.ones(4, device="cuda")

        def step():
            hip.loss_forward(cfg, hip.LOSS_S1, th, x, nrm, sdf, m, W, 100.0, ws)
            hip.loss_backward(cfg, hip.LOSS_S1, th, x, nrm, sdf, m, W, 100.0, ones, None, ws)
        res["train_" + tag] = profiled(step)
        if tag == "N":
            res["query"] = profiled(lambda: hip.query(cfg, th, x, want_grad=True))
    t, q, t2 = res["train_N"], res["query"], res["train_2N"]
    today = sum(t.get(k, 0.0) for k in ("sweep_fwd", "sweep_rev", "sweep_adj_fwd", "sweep_adj_rev", "wgrad_hidden", "wgrad_small"))
    proxy = t["sweep_fwd"] + q["sweep_rev"] + t["sweep_adj_fwd"] + t2["sweep_adj_rev"] + t["wgrad_hidden"] + t["wgrad_small"]
    print("17-unit step (today), ms per kernel at %d points:" % n, {k: round(v, 3) for k, v in t.items()})
    print("reverse sweep that keeps only df/dx (query variant): %.3f ms (training variant %.3f)" % (q["sweep_rev"], t["sweep_rev"]))
    print("adjoint reverse sweep on twice the columns: %.3f ms (once: %.3f)" % (t2["sweep_adj_rev"], t["sweep_adj_rev"]))
    print("MFMA-kernel sum today %.3f ms; 15-unit proxy %.3f ms (adjoint forward sweep unchanged: it would drop one load)" % (today, proxy))


if __name__ == "__main__":
    main()
