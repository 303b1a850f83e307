#!/usr/bin/env python
# coding: utf-8
"""Why does 1 synthetic trajectory in ~25 leave the fp64 curve at step 15 by 8e-4 (tools/traj_stats.py)?

Runs the g12 synthetic fixture R times keeping theta_t of every step on the device, picks the EARLY LEAVERS (first step > 1e-4 well before
the majority's) and a majority run, finds the first step t* at which the leaver's theta separates from the majority run's by more than the
spread between two majority runs, and then looks at the step that produced it (t* - 1): on that step's batch, which points sit on the other
side of one of loss_s1's kinks  sign(y) | sign(tdf - y) | sign(|grad f| - tau)  under the leaver's theta than under the majority run's —
evaluated with the library's own value / gradient query — and how far from the kink they are.  A kink is a discontinuity of the REFERENCE's
gradient (torch.abs / torch.sign in src/loss_functions.py:131-155): a trajectory that crosses one within the rounding noise has two
continuations, in every implementation.      python tools/traj_kink.py [R] [stash]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from diffudf_amd import hip_ops, synth
from diffudf_amd.engine import TrainEngine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
stash = int(sys.argv[2]) if len(sys.argv) > 2 else 7
hip_ops.set_option("stash", stash)
G = np.load(os.path.join(REPO, "tests", "golden", "g12_traj50.npz"))
hidden = list(G["hidden"])
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
n, s1 = int(G["synth_n_points"]), int(G["synth_s1_steps"])
T = min(s1, 26)
syn = [synth.training_batch(n, seed=int(G["batch_seed"]), step=t) for t in range(T)]
syn = [(d(x), d(nr), d(sd.reshape(-1))) for x, nr, sd in syn]
ref = G["synth_s1_f64_hist"][:T]
theta0 = synth.flatten_params(synth.siren_params(hidden, seed=int(G["param_seed"])))
W, ALPHA = [1e4, 1e4, 0.0, 1e3], 100.0
runs = []
for rep in range(R):
    eng = TrainEngine(hidden, d(theta0))
    th, h = [], []
    for x, nr, sd in syn:
        th.append(eng.theta.clone())
        h.append(eng.step(0, x, nr, sd, W, ALPHA, 1e-4).cpu().numpy().copy())
    e = np.abs(np.array(h, dtype=np.float64) - ref).max(axis=1) / np.abs(ref).max(axis=1)
    first = int(np.flatnonzero(e > 1e-4)[0]) if (e > 1e-4).any() else T
    runs.append((first, e, th))
firsts = np.array([r[0] for r in runs])
print(f"stash {stash}: first step > 1e-4 per run: {firsts.tolist()}")
major = int(np.median(firsts))
leavers = [i for i, f in enumerate(firsts) if f <= major - 5]
normal = [i for i, f in enumerate(firsts) if f >= major - 1]
if not leavers:
    print(f"no early leaver in {R} runs (majority leaves at step {major})")
    sys.exit(0)
cfg = hip_ops.make_cfg(hidden)


def kinks(theta, x, sdf):
    f, g = hip_ops.query(cfg, theta, x)
    u = sdf.double(); y = f.double(); gn = g.double().norm(dim=1)
    tan = torch.tanh(ALPHA * u)
    tdf = u * tan
    tau = (tan + u * ALPHA * (1 - tan * tan)).abs()
    on = u == 0
    m_y = torch.where(on, y, tdf - y)                # sign(y) on the surface, sign(tdf - y) off it
    m_g = gn - tau
    return m_y, m_g, on


M0, M1 = normal[0], normal[1]
for L in leavers[:3]:
    thL, thM, thN = runs[L][2], runs[M0][2], runs[M1][2]
    sep = [float((thL[t] - thM[t]).abs().max()) for t in range(T)]
    spread = [float((thN[t] - thM[t]).abs().max()) for t in range(T)]
    tstar = next((t for t in range(1, T) if sep[t] > 20 * max(spread[t], 1e-12)), None)
    print(f"\nleaver run {L} (leaves at step {firsts[L]}; majority run {M0} at {firsts[M0]}):")
    print("  max |theta_L - theta_M| per step: " + " ".join(f"{v:.1e}" for v in sep[:firsts[L] + 2]))
    print("  max |theta_N - theta_M| per step: " + " ".join(f"{v:.1e}" for v in spread[:firsts[L] + 2]) + "   (two majority runs)")
    print("  loss error vs fp64 per step, leaver:   " + " ".join(f"{v:.1e}" for v in runs[L][1][:firsts[L] + 2]))
    print("  loss error vs fp64 per step, majority: " + " ".join(f"{v:.1e}" for v in runs[M0][1][:firsts[L] + 2]))
    if tstar is None:
        print("  theta never separates beyond 20x the majority spread")
        continue
    t = tstar - 1
    x, nr, sd = syn[t]
    (yL, gL, on), (yM, gM, _), (yN, gN, _) = kinks(thL[t], x, sd), kinks(thM[t], x, sd), kinks(thN[t], x, sd)
    for name, a, b, c in (("sign(y) | sign(tdf - y)", yL, yM, yN), ("sign(|grad f| - tau)", gL, gM, gN)):
        flip = ((a > 0) != (b > 0)).nonzero().flatten()
        flipN = ((c > 0) != (b > 0)).nonzero().flatten()
        print(f"  step {t} (the step whose update first separates theta, t* = {tstar}): {name}: {flip.numel()} of {n} points on the other side under the leaver's theta "
              f"({flipN.numel()} between the two majority runs)")
        for i in flip[:6].tolist():
            print(f"      point {i} ({'on' if bool(on[i]) else 'off'} surface): margin leaver {float(a[i]):+.3e}, majority {float(b[i]):+.3e}, other majority run {float(c[i]):+.3e}")
    # how much of the separation that explains: the gradient of the step under both thetas
    ws = hip_ops.workspace_for(cfg, n, "cuda:0")
    ones = torch.ones(4, device="cuda:0")
    gs = []
    for th in (thL[t], thM[t]):
        hip_ops.loss_forward(cfg, 0, th, x, nr, sd, n, W, ALPHA, ws)
        gs.append(hip_ops.loss_backward(cfg, 0, th, x, nr, sd, n, W, ALPHA, ones, None, ws).double().clone())
    print(f"  d(theta) of step {t}: |g_L - g_M| max {float((gs[0] - gs[1]).abs().max()):.3e} against |g| max {float(gs[1].abs().max()):.3e} "
          f"(theta differed by {sep[t]:.1e} going in)")
