# coding: utf-8
"""Pin the oracle (oracle/dudf_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os
import numpy as np
import pytest

from diffudf_amd import synth
from oracle import dudf_oracle as O

W_S1EIK = [1e4, 1e4, 0.0, 1e3]
W_S1FULL = [1e4, 1e4, 1e4, 1e3]
W_S2 = [1e5, 1e5]
W_SIREN = [3e3, 1e2, 1e2, 5e1]


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def flat(grads):
    return np.concatenate([np.concatenate([w.reshape(-1), b.reshape(-1)]) for w, b in grads])


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(os.path.join(golden_dir, "g1_tiny.npz"))


@pytest.fixture(scope="module")
def g2(golden_dir):
    return np.load(os.path.join(golden_dir, "g2_8x256.npz"))


def test_g1_fields_fp64(g1):
    hid = list(g1["hidden"])
    P = synth.siren_params(hid, seed=int(g1["param_seed"]), dtype=np.float64)
    x, nrm, sdf = synth.training_batch(int(g1["n_points"]), seed=int(g1["batch_seed"]), dtype=np.float64)
    y, g, H = O.query(P, x, want_grad=True, want_hess=True)
    assert rel(y, g1["f64_y"]) < 1e-13
    assert rel(g, g1["f64_g"]) < 1e-13
    assert rel(H, g1["f64_H"]) < 1e-12


@pytest.mark.parametrize("case,mode,w", [("s1eik", "s1", W_S1EIK), ("s1full", "s1", W_S1FULL), ("s2", "s2", W_S2),
                                         ("siren", "siren", W_SIREN)])
def test_g1_loss_and_param_grads_fp64(g1, case, mode, w):
    hid = list(g1["hidden"])
    P = synth.siren_params(hid, seed=int(g1["param_seed"]), dtype=np.float64)
    x, nrm, sdf = synth.training_batch(int(g1["n_points"]), seed=int(g1["batch_seed"]), dtype=np.float64)
    terms, grads, _ = O.loss_and_grad(mode, P, x, nrm, sdf, w, 100.0)
    got = np.array([float(v) for v in terms.values()])
    assert rel(got, g1[f"f64_{case}_terms"]) < 1e-12
    assert rel(flat(grads), g1[f"f64_{case}_dtheta"]) < 1e-11


def test_g1_full_s1_terms_fp64(g1):
    """Hessian-on loss_s1: all four TERMS (the eigh-based one included)."""
    hid = list(g1["hidden"])
    P = synth.siren_params(hid, seed=int(g1["param_seed"]), dtype=np.float64)
    x, nrm, sdf = synth.training_batch(int(g1["n_points"]), seed=int(g1["batch_seed"]), dtype=np.float64)
    y, g, H = O.query(P, x, want_grad=True, want_hess=True)
    terms, _ = O.loss_s1_terms(y, g, H, nrm, sdf, W_S1FULL, 100.0)
    got = np.array([float(v) for v in terms.values()])
    assert rel(got, g1["f64_s1full_terms"]) < 1e-10


def test_g1_fp32_oracle_within_reference_noise(g1):
    """The oracle run in fp32 must sit inside the reference's own fp32-vs-fp64 noise band."""
    hid = list(g1["hidden"])
    P = synth.siren_params(hid, seed=int(g1["param_seed"]), dtype=np.float64)
    P32 = [(w.astype(np.float32), b.astype(np.float32)) for w, b in P]
    x, nrm, sdf = [a.astype(np.float32) for a in
                   synth.training_batch(int(g1["n_points"]), seed=int(g1["batch_seed"]), dtype=np.float64)]
    y, g, H = O.query(P32, x, want_grad=True, want_hess=True)
    assert y.dtype == np.float32
    assert rel(y, g1["f32_y"]) < 2e-5
    assert rel(g, g1["f32_g"]) < 5e-5
    assert rel(H, g1["f32_H"]) < 1e-4


def test_g2_8x256_fields_and_grads(g2):
    hid = list(g2["hidden"])
    P32 = synth.siren_params(hid, seed=int(g2["param_seed"]), dtype=np.float32)
    P = [(w.astype(np.float64), b.astype(np.float64)) for w, b in P32]
    x, nrm, sdf = [a.astype(np.float64) for a in
                   synth.training_batch(int(g2["n_points"]), seed=int(g2["batch_seed"]), dtype=np.float32)]
    y, g, H = O.query(P, x, want_grad=True, want_hess=True)
    assert rel(y, g2["f64_y"]) < 1e-12
    assert rel(g, g2["f64_g"]) < 1e-12
    assert rel(H, g2["f64_H"]) < 1e-11
    sample = g2["sample"]
    for case, mode, w in (("s1eik", "s1", W_S1EIK), ("s1full", "s1", W_S1FULL), ("s2", "s2", W_S2),
                          ("siren", "siren", W_SIREN)):
        terms, grads, _ = O.loss_and_grad(mode, P, x, nrm, sdf, w, 100.0)
        got = np.array([float(v) for v in terms.values()])
        assert rel(got, g2[f"f64_{case}_terms"]) < 1e-10, case
        fg = flat(grads)
        assert rel(fg[sample], g2[f"f64_{case}_dtheta_sample"]) < 1e-10, case
        nrm_ref = g2[f"f64_{case}_dtheta_norm"]
        assert abs(np.linalg.norm(fg) - nrm_ref[0]) / nrm_ref[0] < 1e-10, case
    # the reference's own fp32 noise against its fp64 run: documents the tolerance the HIP tests use
    assert rel(g2["f32_y"], g2["f64_y"]) < 2e-5
    assert rel(g2["f32_g"], g2["f64_g"]) < 5e-5


def test_g3_trajectory_small_net(golden_dir):
    """20 Adam steps of loss_s1 (Eikonal-only) and loss_s2 on a 4x64 net follow the reference curve."""
    G = np.load(os.path.join(golden_dir, "g3_traj.npz"))
    hid = list(G["hidden"])
    P32 = synth.siren_params(hid, seed=int(G["param_seed"]), dtype=np.float32)
    for name, mode, w, lr in (("s1eik", "s1", W_S1EIK, 1e-4), ("s2", "s2", W_S2, 1e-6), ("s1full", "s1", W_S1FULL, 1e-4)):
        theta = synth.flatten_params(P32, dtype=np.float64)
        m = np.zeros_like(theta); v = np.zeros_like(theta)
        hist = []
        for t in range(int(G["steps"])):
            x, nrm, sdf = [a.astype(np.float64) for a in
                           synth.training_batch(int(G["n_points"]), seed=int(G["batch_seed"]), step=t, dtype=np.float64)]
            P = synth.unflatten_params(theta, hid)
            terms, grads, _ = O.loss_and_grad(mode, P, x, nrm, sdf, w, 100.0)
            hist.append([float(vv) for vv in terms.values()])
            O.adam_step(theta, flat(grads), m, v, t + 1, lr)
        hist = np.array(hist)
        # kinks (abs / sign) make long trajectories chaotic at the 1e-10 level; 1e-6 relative is ample
        assert rel(hist, G[f"{name}_f64_hist"]) < 1e-6, name
        assert rel(theta, G[f"{name}_f64_theta"]) < 1e-6, name


def test_g4_query_and_inverse(golden_dir):
    G = np.load(os.path.join(golden_dir, "g4_query.npz"))
    n = int(G["grid_n"])
    P32 = synth.siren_params([256] * 8, seed=123, dtype=np.float32)
    ax = np.linspace(-1.0, 1.0, n, dtype=np.float32)
    grid = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    P = [(w.astype(np.float64), b.astype(np.float64)) for w, b in P32]
    y, g, H = O.query(P, grid.astype(np.float64), want_grad=True, want_hess=True)
    # fixture is the reference's fp32 evaluate(): compare at the fp32 noise level
    assert rel(y, G["values"][:, 0]) < 2e-5
    assert rel(g, G["gradients"]) < 5e-5
    assert rel(H, G["hessians"]) < 1e-4
    assert np.allclose(O.inv_tanh(np.abs(G["values"]), 100), G["inv_tanh"], rtol=0, atol=0)


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_curvature_oracle_against_reference(golden_dir, tag):
    """Row A14: shape operator / mean / gaussian curvature of the top-eigenvector field (reference
    src/render_st.py:42-62, `jacobian` src/diff_operators.py:214-227) from the oracle's third-derivative tensor.  The
    reference stores its jacobian in fp32 even when run in fp64 (`torch.zeros(...)` default dtype), hence 2e-7."""
    G = np.load(os.path.join(golden_dir, "g6_curvature.npz"))
    hid = list(G[f"{tag}_hidden"])
    P = synth.siren_params(hid, seed=int(G[f"{tag}_param_seed"]), dtype=np.float64)
    x = G[f"{tag}_x"].astype(np.float64)
    n, pcd, mean, gauss, J = O.curvatures(P, x)
    sgn = np.sign((n * G[f"{tag}_f64_n"]).sum(1))
    assert np.abs(np.abs((n * G[f"{tag}_f64_n"]).sum(1)) - 1).max() < 1e-9
    assert rel(J * sgn[:, None, None], G[f"{tag}_f64_shape_op"]) < 2e-7
    assert rel(mean * sgn, G[f"{tag}_f64_mean"]) < 2e-7
    assert rel(gauss, G[f"{tag}_f64_gauss"]) < 2e-7
    # symmetric trilinear form: T is invariant under index permutations
    T = O.third_derivatives(P, x[:4])
    assert np.abs(T - np.transpose(T, (0, 2, 1, 3))).max() == 0 and np.abs(T - np.transpose(T, (0, 3, 2, 1))).max() == 0


def _ray_agreement(hits_a, mask_a, t0_a, hits_b, mask_b, t0_b, thr):
    """Fraction of rays with the same fate, and the largest position difference among those."""
    same = (hits_a == hits_b) & (mask_a == mask_b)
    d = np.abs(t0_a - t0_b).max(axis=1)
    return same.mean(), d[same].max(), d[~same].max() if (~same).any() else 0.0


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_ray_marching_oracle_against_reference(golden_dir, tag):
    """§8(f) row 3: the sphere-tracing loop (reference src/render_st.py:136-172) restated in oracle/rays_oracle.py against
    the same loop driven around the reference model (tests/golden/g7_rays.npz)."""
    from oracle import rays_oracle as R
    G = np.load(os.path.join(golden_dir, "g7_rays.npz"))
    P = synth.siren_params(list(G[f"{tag}_hidden"]), seed=int(G[f"{tag}_param_seed"]), dtype=np.float64)
    rays, t0 = G[f"{tag}_rays"].copy(), G[f"{tag}_t0"].copy()
    mask = np.ones(len(t0), dtype=bool)
    thr = float(G[f"{tag}_surface_threshold"])
    hits, it = R.propagate_rays(P, rays, t0, mask, "tanh", float(G[f"{tag}_alpha"]), thr, int(G[f"{tag}_max_iterations"]))
    frac, dsame, ddiff = _ray_agreement(hits, mask, t0, G[f"{tag}_hits"], G[f"{tag}_mask"], G[f"{tag}_t0_traced"], thr)
    assert it == int(G[f"{tag}_iterations"]) and frac >= 0.99, (it, frac)
    assert dsame < 2e-3 and ddiff < 0.2          # rays with the same fate sit on the same point; stragglers are one step off
    hits_ref = G[f"{tag}_hits"]
    t1 = G[f"{tag}_t0_traced"].copy()
    R.grad_descent(P, t1, hits_ref, "tanh", float(G[f"{tag}_alpha"]), int(G[f"{tag}_gd_steps"]))
    assert np.abs(t1 - G[f"{tag}_t0_descended"]).max() < 1e-4


# ---- g11: ww != w0 and a latent vector in front of the coordinates (reference src/model.py:89-106, src/evaluate.py:19-22) -------
def _g11(golden_dir):
    return np.load(os.path.join(golden_dir, "g11_ww_latent.npz"))


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_oracle_ww_against_the_reference(golden_dir, tag):
    """The oracle with the frequency pair (w0, ww) = (30, 15) against the reference's own SIREN(3, 1, hidden, w0=30, ww=15)."""
    G = _g11(golden_dir)
    hid = list(G[f"ww_{tag}_hidden"]); n = int(G[f"ww_{tag}_n"]); seed = int(G[f"ww_{tag}_param_seed"])
    P = synth.siren_params(hid, seed=seed, w0=15.0, dtype=np.float64)
    x, nrm, sdf = [a.astype(np.float32).astype(np.float64) for a in synth.training_batch(n, seed=seed + 1, dtype=np.float64)]
    W = (30.0, 15.0)
    y, g, H = O.query(P, x, w0=W, want_grad=True, want_hess=True)
    assert rel(y, G[f"ww_{tag}_f64_y"]) < 1e-11 and rel(g, G[f"ww_{tag}_f64_g"]) < 1e-11 and rel(H, G[f"ww_{tag}_f64_H"]) < 1e-10
    for name, w in (("s1eik", [1e4, 1e4, 0.0, 1e3]), ("s1full", [1e4, 1e4, 1e4, 1e3])):
        terms, grads, _ = O.loss_and_grad("s1", P, x, nrm, sdf, w, 100.0, w0=W)
        assert rel(np.array([float(v) for v in terms.values()]), G[f"ww_{tag}_f64_{name}_terms"]) < 1e-11
        flat = np.concatenate([np.concatenate([a.reshape(-1), b.reshape(-1)]) for a, b in grads])
        if tag == "tiny":
            assert rel(flat, G[f"ww_{tag}_f64_{name}_dtheta"]) < 1e-9
        else:
            assert rel(flat[::97], G[f"ww_{tag}_f64_{name}_dtheta_sample"]) < 1e-9
            assert abs(np.linalg.norm(flat) / G[f"ww_{tag}_f64_{name}_dtheta_norm"][0] - 1) < 1e-10
    # the C ABI runs ONE frequency on a first layer scaled by rho = w0 / ww: the same function, restated here
    P2 = [(P[0][0] * 2.0, P[0][1] * 2.0)] + list(P[1:])
    y2, g2, _ = O.query(P2, x, w0=15.0)
    assert rel(y2, y) < 1e-12 and rel(g2, g) < 1e-12


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_latent_vector_folds_into_the_first_bias(golden_dir, tag):
    """evaluate(model, samples, latent_vec) of the reference = the 3-input network with b_1 + W_1[:, :k] latent (what
    diffudf_amd.model.SIREN.folded builds), checked with the oracle against the reference's values and gradients[..., k:]."""
    G = _g11(golden_dir)
    hid = list(G[f"lat_{tag}_hidden"]); k = int(G[f"lat_{tag}_k"]); seed = int(G[f"lat_{tag}_param_seed"])
    P = synth.siren_params(hid, seed=seed, n_in=3 + k, dtype=np.float64)
    lat = G[f"lat_{tag}_latent"][0]; x = G[f"lat_{tag}_x"].astype(np.float64)
    P3 = [(P[0][0][:, k:], P[0][1] + P[0][0][:, :k] @ lat)] + list(P[1:])
    y, g, _ = O.query(P3, x)
    assert rel(y, G[f"lat_{tag}_f64_y"]) < 1e-11 and rel(g, G[f"lat_{tag}_f64_g"]) < 1e-11
