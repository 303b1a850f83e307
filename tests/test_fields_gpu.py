# coding: utf-8
"""GPU: query-side rows at BASELINE.json's sizes and the boundary gaps VERDICT r01 listed.
  * generate_df's field slice against the reference's own outputs (g9_slice.npz), and at 512^2 (config 4's size);
  * value + gradient + Hessian + eigen-frame + curvature on 512^2 points: per-point independence against the oracle on a
    subset, symmetry, chunk invariance;
  * one slab of the 256^3 grid of config 5 (`dudf_grid_fields`): index-derived coordinates, chunk invariance, subset vs
    oracle;
  * `extract_fields`' Hessian-eigenvector fallback (reference src/render_mc.py:77-93), forced by a vanishing gradient;
  * `divergence` / `laplace` and the (y, x) lookup for tensors derived from the forward's output."""
import os

import numpy as np
import pytest
import torch

from diffudf_amd import synth
from oracle import dudf_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make_model(hidden, seed, out_scale=None):
    from src.model import SIREN
    m = SIREN(3, 1, hidden, w0=30)
    P32 = synth.siren_params(hidden, seed=seed)
    if out_scale is not None:
        P32[-1] = ((P32[-1][0] * np.float32(out_scale)).astype(np.float32), P32[-1][1])
    sd = {}
    for i, (w, b) in enumerate(P32):
        sd[f"net.{i}.0.weight"] = torch.from_numpy(w); sd[f"net.{i}.0.bias"] = torch.from_numpy(b)
    m.load_state_dict(sd)
    return m.to("cuda:0"), [(w.astype(np.float64), b.astype(np.float64)) for w, b in P32]


def oracle_fgh(P, x):
    y, cache = O.forward(P, x)
    g, rev = O.input_gradient(P, cache)
    H, _ = O.hessian(P, x, cache, rev)
    return y, g, H


def test_generate_df_slice_matches_reference(golden_dir):
    import generate_df as GD
    G = np.load(os.path.join(golden_dir, "g9_slice.npz"))
    model, _ = make_model(list(G["hidden"]), int(G["param_seed"]))
    w = int(G["width"])
    out = GD.field_slice(model, {"width": w, "device": "cuda:0"})
    assert np.array_equal(out["samples"], G["samples"])
    assert rel(out["pred_distances"], G["pred_distances"]) < 5e-6
    assert rel(out["pred_grad_norm"], G["pred_grad_norm"]) < 2e-5
    # normals: normalised gradients (or eigenvectors where |grad f| < 0.04); skip the handful of points that sit ON a
    # decision boundary of the reference's own fp32 run (|grad f| within 1e-4 of 0.04, third component within 1e-5 of 0)
    gn = G["pred_grad_norm"][:, 0]
    safe = (np.abs(gn - 0.04) > 1e-4) & (np.abs(G["normals"][:, 2]) > 1e-5)
    eig = gn < 0.04
    assert safe.sum() > 0.98 * len(gn)
    assert np.abs(out["normals"][safe & ~eig] - G["normals"][safe & ~eig]).max() < 5e-5
    if (safe & eig).any():                     # eigenvector branch: direction up to the eigen-gap conditioning
        c = np.abs((out["normals"][safe & eig] * G["normals"][safe & eig]).sum(1))
        assert c.min() > 1 - 1e-3
    diff = np.abs(out["grad_map"].astype(int) - G["grad_map_u8"].astype(int)).reshape(-1, 3).max(1)
    assert (diff[safe] <= 1).all()             # the 8-bit normal map: at most one level off


def test_config4_size_hessian_frame_curvature_512x512():
    """BASELINE config 4: 512^2 = 262 144 points through value + gradient + Hessian + eigh (+ curvature)."""
    import generate_df as GD
    from diffudf_amd import hip_ops as hip
    model, P = make_model([256] * 8, 123)
    w = 512
    out = GD.field_slice(model, {"width": w, "device": "cuda:0"})
    x = out["samples"].astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    f, g, H, lam, V = hip.query_frame(model.hip_cfg, model.flat_parameters(), xt)
    f, g, H, lam, V = [t.cpu().numpy().astype(np.float64) for t in (f, g, H, lam, V)]
    assert rel(f, out["pred_distances"][:, 0]) < 1e-6                       # evaluate() and query_frame agree
    idx = np.random.default_rng(1).choice(w * w, 192, replace=False)
    yo, go, Ho = oracle_fgh(P, x[idx].astype(np.float64))
    assert rel(f[idx], yo) < 5e-6 and rel(g[idx], go) < 2e-5 and rel(H[idx], Ho) < 2e-5
    assert np.abs(H - np.transpose(H, (0, 2, 1))).max() < 2e-5 * np.abs(H).max()    # symmetric without being symmetrised
    # eigen-decomposition of the LOWER triangle: H_l V = V diag(lam), V orthonormal, lam ascending
    Hl = np.tril(H) + np.transpose(np.tril(H, -1), (0, 2, 1))
    assert np.abs(np.einsum("nij,njk->nik", Hl, V) - V * lam[:, None, :]).max() < 2e-5 * np.abs(H).max()
    assert np.abs(np.einsum("nji,njk->nik", V, V) - np.eye(3)).max() < 1e-5
    assert (np.diff(lam, axis=1) >= 0).all()
    # the same points in another order / chunking give the same numbers
    perm = torch.from_numpy(np.random.default_rng(2).permutation(w * w)[:70001]).cuda()
    f2, g2, H2 = hip.query_hessian(model.hip_cfg, model.flat_parameters(), xt[perm].contiguous())
    assert rel(f2.cpu().numpy(), f[perm.cpu().numpy()]) < 1e-6
    assert rel(H2.cpu().numpy(), H[perm.cpu().numpy()]) < 2e-6
    # curvature of the eigen-normal field on the full 512^2 set; subset against the oracle's third derivatives
    lam3, V3, mean, gauss, J = hip.query_curvature(model.hip_cfg, model.flat_parameters(), xt, want_shape=True)
    J = J.cpu().numpy().astype(np.float64); mean = mean.cpu().numpy().astype(np.float64)
    assert np.isfinite(mean).all() and rel(0.5 * np.trace(J, axis1=1, axis2=2), mean) < 1e-5
    no, _, mo, gaus_o, Jo = O.curvatures(P, x[idx[:64]].astype(np.float64))
    lam_o = np.linalg.eigvalsh(np.tril(Ho[:64]) + np.transpose(np.tril(Ho[:64], -1), (0, 2, 1)))
    gap = np.minimum(lam_o[:, 2] - lam_o[:, 1], lam_o[:, 1] - lam_o[:, 0]) / (lam_o[:, 2] - lam_o[:, 0])
    ok = gap > 0.05                                                         # away from eigenvalue crossings
    sgn = np.sign((V3.cpu().numpy()[idx[:64], :, 2] * no).sum(1))
    assert ok.sum() > 30
    assert rel((J[idx[:64]] * sgn[:, None, None])[ok], Jo[ok]) < 2e-4


def test_config5_grid_slab_256():
    """BASELINE config 5's grid at the reference's default resolution 256 (configs/mc_cfg.json): one slab of 2^21 points
    and the last, ragged chunk — coordinates derived from the linear index inside the kernel."""
    from diffudf_amd import hip_ops as hip
    model, P = make_model([256] * 8, 123)
    cfg, theta = model.hip_cfg, model.flat_parameters()
    N = 256
    total = N ** 3
    df = torch.full((total,), -1.0, device="cuda"); vec = torch.zeros(total, 3, device="cuda")
    start, count = 5 * N * N + 77, 1 << 21
    hip.grid_fields(cfg, theta, N, start, count, "tanh", 100.0, df, vec)
    last = total - 1000
    hip.grid_fields(cfg, theta, N, last, 1000, "tanh", 100.0, df, vec)
    assert float(df[:start].max()) == -1.0 and float(df[start + count:last].max()) == -1.0      # nothing outside the ranges
    idx = np.concatenate([start + np.random.default_rng(3).choice(count, 200, replace=False), [start, start + count - 1, total - 1]])
    voxel = 2.0 / (N - 1)
    ijk = np.stack([(idx // (N * N)) % N, (idx // N) % N, idx % N], 1)
    xs = (ijk.astype(np.float32) * np.float32(voxel) + np.float32(-1.0)).astype(np.float32)      # reference :42-49, fp32
    yo, go, _ = oracle_fgh(P, xs.astype(np.float64))
    a = np.abs(yo)
    df_ref = np.where(a < 1.0 / 100.0, np.sqrt(a / 100.0), a)                                   # inverse('tanh', |f|, 100)
    assert np.allclose(df[idx].cpu().numpy(), df_ref, rtol=3e-5, atol=1e-7)
    v_ref = -go / np.maximum(np.linalg.norm(go, axis=1, keepdims=True), 1e-12)
    assert np.abs(vec[idx].cpu().numpy() - v_ref).max() < 1e-4
    # chunk invariance: the same points through the point query
    f2, g2 = hip.query(cfg, theta, torch.from_numpy(xs).cuda())
    a2 = f2.abs()
    df2 = torch.where(a2 < 0.01, torch.sqrt(a2 / 100.0), a2)
    assert np.allclose(df[idx].cpu().numpy(), df2.cpu().numpy(), rtol=2e-5, atol=1e-7)


def test_extract_fields_takes_the_eigenvector_fallback():
    """reference src/render_mc.py:77-93: where the normalised gradient is shorter than 0.04 (a vanishing gradient:
    F.normalize's eps = 1e-12 keeps it short) the direction field is the top Hessian eigenvector, sign-aligned with
    -grad f.  Forced for EVERY grid point by scaling the output layer by 1e-20: |grad f| ~ 1e-19, while the Hessian's
    eigenvectors do not care about the scale."""
    from src.render_mc import extract_fields
    model, P = make_model([64] * 3, 5, out_scale=1e-20)
    n = 10
    df, vecs = extract_fields(model, None, n, "tanh", torch.device("cuda:0"), 100, chunk=300)
    ax = np.linspace(-1.0, 1.0, n, dtype=np.float32)
    grid = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    yo, go, Ho = oracle_fgh(P, grid)
    assert np.linalg.norm(go, axis=1).max() < 1e-14                       # every point is on the fallback branch
    lam, V = np.linalg.eigh(np.tril(Ho) + np.transpose(np.tril(Ho, -1), (0, 2, 1)))
    nh = V[:, :, 2]
    gneg = -go / np.maximum(np.linalg.norm(go, axis=1, keepdims=True), 1e-12)
    nh = np.where((gneg * nh).sum(1, keepdims=True) < 0, -1.0, 1.0) * nh
    v = vecs.reshape(-1, 3).cpu().numpy().astype(np.float64)
    assert np.abs(np.linalg.norm(v, axis=1) - 1).max() < 1e-4             # unit eigenvectors, not the tiny gradients
    gap = (lam[:, 2] - lam[:, 1]) / np.abs(lam).max(axis=1)
    ok = gap > 0.02
    assert ok.sum() > 0.8 * len(v)
    assert ((v[ok] * nh[ok]).sum(1) > 1 - 1e-3).all()                     # same direction AND same sign
    a = np.abs(yo)
    assert np.allclose(df.reshape(-1).cpu().numpy(), np.where(a < 0.01, np.sqrt(a / 100.0), a), rtol=3e-5, atol=1e-7)


def test_divergence_laplace_and_derived_outputs(golden_dir):
    """reference src/diff_operators.py:196-212 on the call shapes the reference itself uses: y squeezed / reshaped
    before `gradient` / `hessian` (src/loss_functions.py:141), `divergence(gradient(y, x), x)`, and the divergence of the
    eigen-normal field (= trace of the shape operator, g6 fixture)."""
    from src import diff_operators as dif
    from src.render_st import compute_normals_and_cd
    G8 = np.load(os.path.join(golden_dir, "g8_operators.npz")); G6 = np.load(os.path.join(golden_dir, "g6_curvature.npz"))
    for tag in ("tiny", "full"):
        model, P = make_model(list(G6[f"{tag}_hidden"]), int(G6[f"{tag}_param_seed"]))
        x = torch.from_numpy(G6[f"{tag}_x"]).to("cuda:0")
        mo = model(x[None])
        xin, y = mo["model_in"], mo["model_out"]
        lap = dif.laplace(y, xin)
        assert lap.shape == (1, len(x), 1) and rel(lap[0].cpu().numpy(), G8[f"{tag}_laplace"]) < 2e-5
        g = dif.gradient(y, xin)
        div = dif.divergence(g, xin)
        assert rel(div[0].cpu().numpy(), G8[f"{tag}_div_grad"]) < 2e-5
        # tensors DERIVED from the forward's output resolve through model_in
        h1 = dif.hessian(y.squeeze(-1), xin)
        h2 = dif.hessian(y, xin)
        assert torch.equal(h1, h2)
        g1 = dif.gradient(y.reshape(1, -1, 1).clone(), xin)
        assert torch.equal(g1, g)
        for bad in (y * 1.0, 2 * y, y.abs(), torch.tanh(y.squeeze(-1))):   # FUNCTIONS of the output have no HIP path: loud
            with pytest.raises(Exception):
                dif.gradient(bad, xin)
            with pytest.raises(Exception):
                dif.hessian(bad, xin)
        normals, _ = compute_normals_and_cd(xin, y)
        dn = dif.divergence(normals, xin)[0, :, 0].cpu().numpy().astype(np.float64)
        sgn = np.sign((normals[0].cpu().numpy() * G6[f"{tag}_f64_n"]).sum(1))
        assert rel(dn * sgn, 2.0 * G6[f"{tag}_f64_mean"]) < 1e-4           # trace of the shape operator = 2 x mean curvature
        with pytest.raises(Exception):
            dif.gradient(g[..., 0], xin)                                   # slices of a gradient: documented as unsupported


def test_curvature_and_frame_at_width_512():
    """BASELINE config 3's width: Hessian quads, eigen-frame and third-order jets at H = 512 (f32-input MFMA kernel),
    against the oracle's Hessian and third derivatives."""
    from diffudf_amd import hip_ops as hip
    model, P = make_model([512] * 3, 9)
    x = (synth.uniform01(5, 91, 0, 3 * 96).reshape(96, 3) * 1.8 - 0.9).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    f, g, H, lam, V = hip.query_frame(model.hip_cfg, model.flat_parameters(), xt)
    yo, go, Ho = oracle_fgh(P, x.astype(np.float64))
    assert rel(f.cpu().numpy(), yo) < 5e-6 and rel(g.cpu().numpy(), go) < 2e-5 and rel(H.cpu().numpy(), Ho) < 2e-5
    no, _, mo, gaus_o, Jo = O.curvatures(P, x.astype(np.float64))
    lam3, V3, mean, gauss, J = hip.query_curvature(model.hip_cfg, model.flat_parameters(), xt, want_shape=True)
    lam_o = np.linalg.eigvalsh(np.tril(Ho) + np.transpose(np.tril(Ho, -1), (0, 2, 1)))
    gap = np.minimum(lam_o[:, 2] - lam_o[:, 1], lam_o[:, 1] - lam_o[:, 0]) / (lam_o[:, 2] - lam_o[:, 0])
    ok = gap > 0.05
    sgn = np.sign((V3.cpu().numpy()[:, :, 2] * no).sum(1))
    assert ok.sum() > 40
    assert rel((J.cpu().numpy() * sgn[:, None, None])[ok], Jo[ok]) < 2e-4
    assert rel((mean.cpu().numpy() * sgn)[ok], mo[ok]) < 2e-4


def test_gradient_is_part_of_the_graph():
    """reference src/diff_operators.py:208-212 builds df/dx with create_graph=True: a loss written in plain PyTorch on
    `model(x)['model_out']` and `gradient(y, x)` — here the reference's own loss_siren terms, src/loss_functions.py:82-104,
    restated on the two tensors — must train the network: parameter gradients against the fused loss_siren's."""
    from src import diff_operators as dif
    from src.loss_functions import loss_siren
    model, P = make_model([256] * 8, 123)
    x, nrm, sdf = [torch.from_numpy(a).cuda() for a in synth.training_batch(900, seed=4, step=0)]
    w = [3e3, 1e2, 1e2, 5e1]
    mo = model(x[None])
    y, xin = mo["model_out"], mo["model_in"]
    g = dif.gradient(y, xin)
    assert g.requires_grad and g.shape == (1, 900, 3)
    on = (sdf[None] == 0)
    t0 = torch.where(on, y.abs(), torch.zeros_like(y)).mean() * w[0]
    t1 = torch.where(~on, torch.exp(-1e2 * y.abs()), torch.zeros_like(y)).mean() * w[1]
    cosv = torch.nn.functional.cosine_similarity(g, nrm[None], dim=-1)[..., None]
    t2 = torch.where(on, 1 - cosv, torch.zeros_like(y)).mean() * w[2]
    t3 = ((g.norm(dim=-1) - 1) ** 2).mean() * w[3]
    (t0 + t1 + t2 + t3).backward()
    got = np.concatenate([p.grad.reshape(-1).cpu().numpy() for p in model.parameters()])
    model.zero_grad()
    terms = loss_siren(model, x[None], {"normals": nrm[None], "sdf": sdf[None]}, w)
    sum(terms.values()).backward()
    ref = np.concatenate([p.grad.reshape(-1).cpu().numpy() for p in model.parameters()])
    tv = np.array([t.item() for t in terms.values()])
    assert rel(np.array([t0.item(), t1.item(), t2.item(), t3.item()]), tv) < 1e-5
    assert rel(got, ref) < 2e-5
    with torch.no_grad():                                # a plain query outside the graph
        mo2 = model(x[None])
        assert not dif.gradient(mo2["model_out"], mo2["model_in"]).requires_grad
