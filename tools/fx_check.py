#!/usr/bin/env python
# coding: utf-8
"""Do the fixed-point stash's preconditions hold on the device?  (VERDICT r05 #1 / ADVICE r05: `fx24_pack` / `c24_pack` store the low 24 bits of
fma(v, 2^-E, 3): a value a grid step outside [-1, 1] 2^E would read back as ~ +5 2^E, silently.)  Runs training steps of every kind on a DEBUG build
that counts such granules inside the packers and prints the counters:

    bash tools/build_dbg.sh fxcheck sweep_bf16 "-DDUDF_FX_CHECK=1"
    DUDF_LIB=dbg/libdudf_fxcheck.so python tools/fx_check.py [trajectory repeats]

Workloads: the g12 fixtures (beetle x50 on the oracle sampler's batches; synthetic s1 x40 -> s2 x10) R times, the 100 000-point headline step,
the full loss_s1 (Hessian term on) at 29 970 points x20, 8x512 at 20 000 points, and a batch with huge and tiny coordinates/cotangent scales."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from diffudf_amd import hip_ops, synth, mesh, _lib
from diffudf_amd.engine import TrainEngine
from oracle import sampler_oracle as SO

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _lib.load()
assert hasattr(lib, "dudf_dbg_fx_violations"), "not a -DDUDF_FX_CHECK=1 build: set DUDF_LIB (see the docstring)"
lib.dudf_dbg_fx_violations.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int]


def counters(reset=True):
    torch.cuda.synchronize()
    out = (ctypes.c_uint * 2)()
    assert lib.dudf_dbg_fx_violations(out, 1 if reset else 0) == 0
    return int(out[0]), int(out[1])


d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
torch.zeros(1, device="cuda:0")
counters()
total = [0, 0]


def report(what, launches):
    c = counters()
    total[0] += c[0]; total[1] += c[1]
    print(f"{what}: {launches} steps, granules outside the grid: S/Q/A/Z {c[0]}, C {c[1]}", flush=True)


G = np.load(os.path.join(REPO, "tests", "golden", "g12_traj50.npz"))
hidden = list(G["hidden"])
W = [1e4, 1e4, 0.0, 1e3]
tri, pos, nrm = mesh.prepare(os.path.join(REPO, "tests", "golden", "beetle"), int(G["surface_points"]), seed=int(G["batch_seed"]))
bs = int(G["beetle_batch_size"]); n_on, n_off = int(bs * 0.333), int(bs * 0.666)
beetle = [SO.sample_batch(tri, pos, nrm, n_on, n_off // 2, n_off - n_off // 2, seed=int(G["batch_seed"]), step=t) for t in range(int(G["beetle_steps"]))]
beetle = [(d(x), d(n_), d(s[:, 0])) for x, n_, s in beetle]
n, s1, s2 = int(G["synth_n_points"]), int(G["synth_s1_steps"]), int(G["synth_s2_steps"])
syn = [synth.training_batch(n, seed=int(G["batch_seed"]), step=t) for t in range(s1 + s2)]
syn = [(d(x), d(nr), d(sd.reshape(-1))) for x, nr, sd in syn]
theta0 = synth.flatten_params(synth.siren_params(hidden, seed=int(G["param_seed"])))
for rep in range(R):
    eng = TrainEngine(hidden, d(theta0))
    for x, n_, s in beetle:
        eng.step(0, x, n_, s, W, 100.0, 1e-4)
    eng = TrainEngine(hidden, d(theta0))
    for x, nr, sd in syn[:s1]:
        eng.step(0, x, nr, sd, W, 100.0, 1e-4)
    for x, nr, sd in syn[s1:]:
        eng.step(1, x, nr, sd, [1e5, 1e5], 100.0, 1e-5)
report(f"g12 trajectories x{R} (stash mode {hip_ops.stash_mode(hip_ops.make_cfg(hidden), n)})", R * (len(beetle) + s1 + s2))

for hid, npts, steps, full in (([256] * 8, 100000, 10, False), ([256] * 8, 29970, 20, True), ([512] * 8, 20000, 6, False), ([256] * 3, 4097, 20, True)):
    eng = TrainEngine(hid, d(synth.flatten_params(synth.siren_params(hid, seed=123))))
    for t in range(steps):
        x, nr, sd = [d(a) for a in synth.training_batch(npts, seed=9, step=t)]
        sd = sd.reshape(-1)
        eng.step(0, x, nr, sd, [1e4, 1e4, 1e4 if full else 0.0, 1e3], 100.0, 1e-4, n_hess=int((sd == 0).sum()) if full else 0)
    report(f"{hid[0]}x{len(hid)} at {npts} points, loss_s1 {'with the Hessian term' if full else 'Eikonal'} (stash mode {hip_ops.stash_mode(hip_ops.make_cfg(hid), npts, int(npts // 3) if full else 0)})", steps)

# extreme scales: coordinates up to 1e3 (first-layer arguments ~1e4 rad), loss weights 1e12 and 1e-12
hid = [256] * 8
for scale, w in ((1e3, [1e12, 1e12, 0.0, 1e12]), (1e-3, [1e-12, 1e-12, 0.0, 1e-12]), (1.0, [1e4, 1e4, 0.0, 1e3])):
    eng = TrainEngine(hid, d(synth.flatten_params(synth.siren_params(hid, seed=5))))
    for t in range(6):
        x, nr, sd = [d(a) for a in synth.training_batch(20000, seed=11, step=t)]
        eng.step(0, x * scale, nr, sd.reshape(-1), w, 100.0, 1e-4)
    report(f"8x256, coordinates x{scale:g}, weights {w[0]:g}", 6)
print(f"TOTAL granules outside the grid: S/Q/A/Z {total[0]}, C {total[1]}")
sys.exit(1 if (total[0] or total[1]) else 0)
