# coding: utf-8
"""GPU: BASELINE config 0/1 — the reference's example mesh (data/beetle).  GPU batch sampler against its numpy
restatement, and N training steps (sampler -> loss_s1 -> torch.optim.Adam) against the curve the REFERENCE produced
on the same batches (tests/golden/g5_beetle.npz, made by tests/golden/make_golden.py)."""
import os
import numpy as np
import pytest
import torch

from diffudf_amd import synth
from oracle import sampler_oracle as SO

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BEETLE = os.path.join(HERE, "golden", "beetle")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_gpu_sampler_matches_oracle(golden_dir):
    from diffudf_amd.dataset import PointCloud
    G = np.load(os.path.join(golden_dir, "g5_beetle.npz"))
    bs = int(G["batch_size"])
    ds = PointCloud(BEETLE, bs, [0.333, 0.666], 1, device="cuda:0", seed=int(G["batch_seed"]))
    assert ds.n_global == 2997 and ds.samplesOnSurface == 999
    x, nrm, sdf = ds.sample(0)
    # the batch the reference trajectory was generated on
    # (coordinates within one fp32 ulp at |x| < 1; distances to 2 ulp of 1.0 — the oracle measures them in fp64)
    assert np.abs(x.cpu().numpy() - G["batch0_x"]).max() <= 1.2e-7
    assert np.abs(sdf.cpu().numpy() - G["batch0_sdf"][:, 0]).max() <= 2.4e-7
    assert (sdf[:999] == 0).all() and (sdf[999:] > 0).all() and (nrm[999:] == 0).all()
    # sharded: rank 1 of 3 produces exactly its slices, at a later step
    ds3 = PointCloud(BEETLE, bs, [0.333, 0.666], 1, device="cuda:0", seed=7, rank=1, world=3)
    tri, pos, pn = ds3.tri.cpu().numpy(), ds3.pc_pos.cpu().numpy(), ds3.pc_nrm.cpu().numpy()
    xs, ns, ss = ds3.sample(5)
    xo, no, so = SO.sample_batch(tri, pos, pn, 999, 999, 999, seed=7, step=5, rank=1, world=3)
    assert np.abs(xs.cpu().numpy() - xo).max() < 1e-6 and np.array_equal(ns.cpu().numpy(), no)
    assert np.abs(ss.cpu().numpy() - so[:, 0]).max() < 2e-6
    # iterator contract of the reference dataset: (1,N,3), (1,N,3), (1,N,1)
    a, b, c = next(iter(ds))
    assert a.shape == (1, 2997, 3) and b.shape == (1, 2997, 3) and c.shape == (1, 2997, 1)


def test_gpu_sampler_point_cloud_only():
    """`PointCloud(onlyPCloud=True)` (reference src/dataset.py:80-131): nearest-cloud-point distance for the far
    stratum (the reference's `shortestDistance`, :72-78, recomputed below in torch exactly as written there), |offset|
    for the near stratum."""
    from diffudf_amd.dataset import PointCloud
    ds = PointCloud(BEETLE, 3000, [0.333, 0.666], 1, device="cuda:0", onlyPCloud=True, seed=11, surfacePoints=5000)
    assert ds.tri is None
    pos, pn = ds.pc_pos.cpu().numpy(), ds.pc_nrm.cpu().numpy()
    x, nrm, sdf = ds.sample(2)
    xo, no, so = SO.sample_batch(None, pos, pn, 999, 999, 999, seed=11, step=2)
    assert np.abs(x.cpu().numpy() - xo).max() < 1e-6 and np.array_equal(nrm.cpu().numpy(), no)
    assert np.abs(sdf.cpu().numpy() - so[:, 0]).max() < 2e-6
    P, X = x[999:1998].double().cpu(), ds.pc_pos.double().cpu()
    sq = (X * X).sum(1).repeat(P.shape[0], 1) - 2 * (P @ X.T)                  # reference :73-78, in fp64
    ref = torch.sqrt(sq.min(dim=1)[0] + (P * P).sum(1))
    assert np.abs(sdf[999:1998].cpu().numpy() - ref.numpy()).max() < 2e-6
    # near stratum: the stored distance is the displacement along the unit normal
    k = np.linalg.norm(x[1998:].cpu().numpy()[:, None, :] - pos[None], axis=2).min(1)
    assert (k <= sdf[1998:].cpu().numpy() + 1e-6).all()
    # sharding keeps the union independent of the world size
    parts = [PointCloud(BEETLE, 3000, [0.333, 0.666], 1, device="cuda:0", onlyPCloud=True, seed=11, surfacePoints=5000,
                        rank=r, world=2).sample(2) for r in range(2)]
    got = torch.cat([torch.cat([parts[0][2][:499], parts[1][2][:500]]),
                     torch.cat([parts[0][2][499:998], parts[1][2][500:1000]]),
                     torch.cat([parts[0][2][998:], parts[1][2][1000:]])])
    assert torch.equal(got, sdf)


def _fresh_model(G):
    from src.model import SIREN
    hidden = list(G["hidden"])
    model = SIREN(3, 1, hidden, w0=30)
    sd = {}
    for i, (wt, b) in enumerate(synth.siren_params(hidden, seed=int(G["param_seed"]))):
        sd[f"net.{i}.0.weight"] = torch.from_numpy(wt); sd[f"net.{i}.0.bias"] = torch.from_numpy(b)
    model.load_state_dict(sd)
    return model.to("cuda:0")


def _fixture_batches(G):
    """The EXACT batches the reference trajectory of g5_beetle.npz was run on: regenerated on the host with the
    sampler oracle, as tests/golden/make_golden.py does (batch 0 is stored in the fixture and compared bit for bit)."""
    from diffudf_amd import mesh
    tri, pos, nrm = mesh.prepare(BEETLE, int(G["surface_points"]), seed=int(G["batch_seed"]))
    bs = int(G["batch_size"])
    n_on, n_off = int(bs * 0.333), int(bs * 0.666)
    out = [SO.sample_batch(tri, pos, nrm, n_on, n_off // 2, n_off - n_off // 2, seed=int(G["batch_seed"]), step=t)
           for t in range(int(G["steps"]))]
    assert np.array_equal(out[0][0], G["batch0_x"]) and np.array_equal(out[0][2], G["batch0_sdf"])
    return out


@pytest.mark.parametrize("source", ["fixture_batches", "hip_sampler"])
@pytest.mark.parametrize("name,w", [("s1eik", [1e4, 1e4, 0.0, 1e3]), ("s1full", [1e4, 1e4, 1e4, 1e3])])
def test_beetle_training_follows_reference(golden_dir, name, w, source):
    """North star: loss within 1e-4 (relative) of the reference after N steps on the beetle mesh.  The reference's own
    loop (loss dict -> sum -> backward -> torch.optim.Adam, train.py:195-222) is run (a) on the fixture's exact
    batches and (b) on the batches of the HIP sampler (equal to them within one fp32 ulp of the coordinates).
    Eikonal loss: every one of the 12 steps within 1e-4 of the reference's fp64 curve (measured on MI355X: <= 7e-7;
    the reference's own fp32 run is within 2e-7).  With the eigenvector term on, the reference's fp32 run leaves its own
    fp64 run by 1.7e-3 at step 2 and 0.36 later (chaotic: 1/(lambda_2 - lambda_j) factors), so the bar there is that
    drift, see DESIGN.md §4."""
    from src.dataset import PointCloud
    from src.loss_functions import loss_s1
    G = np.load(os.path.join(golden_dir, "g5_beetle.npz"))
    hidden = list(G["hidden"])
    model = _fresh_model(G)
    steps = int(G["steps"])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
    if source == "fixture_batches":
        batches = [(d(x)[None], d(n)[None], d(s)[None]) for x, n, s in _fixture_batches(G)]
    else:
        ds = PointCloud(BEETLE, int(G["batch_size"]), [0.333, 0.666], 1, device="cuda:0", seed=int(G["batch_seed"]))
        batches = [next(iter(ds)) for _ in range(steps)]
    opt = torch.optim.Adam(lr=1e-4, params=model.parameters())
    hist = []
    for x, nrm, sdf in batches:
        opt.zero_grad()
        loss = loss_s1(model, x, {"normals": nrm, "sdf": sdf}, w, 100)
        total = torch.zeros((1, 1), device="cuda:0")
        for l in loss.values():
            total += l
        total.backward()
        opt.step()
        hist.append([l.item() for l in loss.values()])
    hist = np.array(hist)
    ref = G[f"{name}_f64_hist"]
    drift = np.abs(G[f"{name}_f32_hist"] - ref).max(axis=1) / np.abs(ref).max(axis=1)     # the reference's own fp32 run
    per_step = np.abs(hist - ref).max(axis=1) / np.abs(ref).max(axis=1)
    et = rel(model.flat_parameters().cpu().numpy()[G["sample"]], G[f"{name}_f64_theta_sample"])
    print(f"beetle {name} [{source}]: per-step curve err {np.array2string(per_step, precision=1)}; theta err {et:.2e}; "
          f"reference fp32 drift {np.array2string(drift, precision=1)}")
    if name == "s1eik":
        assert per_step.max() < 1e-4                      # the north-star bar, all 12 steps
        assert per_step.max() < 5e-6 and et < 1e-5        # ... and what the build actually holds, with margin
    else:
        assert per_step[:2].max() < 5e-4                  # before the chaos sets in (reference fp32: 3e-6 at step 1)
        assert per_step.max() < 2.0 * drift.max()
    # same-theta parity at the END of the run: HIP loss at the trained parameters vs the fp64 oracle at the same ones
    from oracle import dudf_oracle as O
    x, nrm, sdf = batches[-1]
    theta = model.flat_parameters().cpu().numpy().astype(np.float64)
    P = synth.unflatten_params(theta, hidden)
    xo, no, so = [t.reshape(-1, t.shape[-1]).cpu().numpy().astype(np.float64) for t in (x, nrm, sdf)]
    t_ref, g_ref, _ = O.loss_and_grad("s1", P, xo, no, so.reshape(-1, 1), w, 100.0)
    model.zero_grad()
    loss = loss_s1(model, x, {"normals": nrm, "sdf": sdf}, w, 100)
    sum(loss.values()).backward()
    got = np.array([l.item() for l in loss.values()])
    gh = np.concatenate([p.grad.reshape(-1).cpu().numpy() for p in model.parameters()])
    gr = np.concatenate([np.concatenate([a.reshape(-1), b.reshape(-1)]) for a, b in g_ref])
    tol_t, tol_g = 1e-5, (5e-4 if name == "s1full" else 1e-4)
    if name == "s1full":
        # The trained parameters differ from run to run (float atomics + a chaotic loss), and now and then they put an
        # on-surface point next to a degenerate Hessian (lambda_2 ~ lambda_j): its 1/(lambda_2 - lambda_j) factor then
        # dominates the gradient and no fp32 evaluation of it — the reference's own included — is accurate.  The bar
        # therefore scales with what the SAME oracle loses when it is evaluated in fp32 at these parameters.
        P32 = [(a.astype(np.float32), b.astype(np.float32)) for a, b in P]
        t32, g32, _ = O.loss_and_grad("s1", P32, xo.astype(np.float32), no.astype(np.float32), so.reshape(-1, 1).astype(np.float32), w, 100.0)
        g32 = np.concatenate([np.concatenate([a.reshape(-1), b.reshape(-1)]) for a, b in g32]).astype(np.float64)
        noise_g = rel(g32, gr)
        noise_t = rel([float(v) for v in t32.values()], [float(v) for v in t_ref.values()])
        print(f"  same-theta parity: terms {rel(got, [float(v) for v in t_ref.values()]):.1e} (fp32 oracle {noise_t:.1e}); "
              f"dtheta {rel(gh, gr):.1e} (fp32 oracle {noise_g:.1e})")
        tol_t, tol_g = max(tol_t, 6.0 * noise_t), max(tol_g, 6.0 * noise_g)      # (measured: HIP <= 2.3 x the fp32 oracle's own error)
    assert rel(got, [float(v) for v in t_ref.values()]) < tol_t
    assert rel(gh, gr) < tol_g


def test_beetle_training_engine_path(golden_dir):
    """The same 12 Eikonal steps through the flat-buffer TrainEngine (dudf_loss_forward/backward + dudf_adam_step,
    what bench.py times): within 1e-4 of the reference's curve on the fixture's batches."""
    from diffudf_amd.engine import TrainEngine
    G = np.load(os.path.join(golden_dir, "g5_beetle.npz"))
    hidden = list(G["hidden"])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
    theta = d(synth.flatten_params(synth.siren_params(hidden, seed=int(G["param_seed"]))))
    eng = TrainEngine(hidden, theta)
    hist = []
    for x, n_, s in _fixture_batches(G):
        hist.append(eng.step(0, d(x), d(n_), d(s[:, 0]), [1e4, 1e4, 0.0, 1e3], 100.0, 1e-4).cpu().numpy().copy())
    ref = G["s1eik_f64_hist"]
    per_step = np.abs(np.array(hist) - ref).max(axis=1) / np.abs(ref).max(axis=1)
    et = rel(theta.cpu().numpy()[G["sample"]], G["s1eik_f64_theta_sample"])
    print(f"beetle s1eik [TrainEngine]: per-step {np.array2string(per_step, precision=1)}; theta err {et:.2e}")
    assert per_step.max() < 1e-4 and et < 1e-4
