# coding: utf-8
"""GPU: the two constructor arguments of the reference's SIREN that no shipped config uses (VERDICT r03 missing #3, #4).

  * `ww != w0` (reference src/model.py:89-106: first SineLayer w0, the others ww) through the C ABI's `dudf_net_cfg.ww`:
    value, df/dx, Hessian, `loss_s1` terms and d(theta) against the reference's own fp64 outputs (tests/golden/g11_ww_latent.npz,
    written by tests/golden/make_golden.py from the imported reference) and against the oracle run with the frequency pair;
  * a latent vector in front of the coordinates (reference src/evaluate.py:19-22) through `evaluate(model, samples, latent_vec)`:
    values and gradients[..., k:] against the reference's `evaluate`."""
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


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "g11_ww_latent.npz"))


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_ww_against_the_reference(G, tag):
    from diffudf_amd import hip_ops as hip
    hid = list(G[f"ww_{tag}_hidden"]); n = int(G[f"ww_{tag}_n"]); seed = int(G[f"ww_{tag}_param_seed"])
    P = synth.siren_params(hid, seed=seed, w0=15.0, dtype=np.float64)
    P32 = [(w.astype(np.float32), b.astype(np.float32)) for w, b in P]
    x, nrm, sdf = synth.training_batch(n, seed=seed + 1)
    cfg = hip.make_cfg(hid, 30.0, ww=15.0)
    assert cfg.ww == 15.0 and hip.make_cfg(hid, 30.0, ww=30.0).ww == 0.0
    th = torch.from_numpy(synth.flatten_params(P32)).cuda()
    xd, nd, sd = [torch.from_numpy(a).cuda() for a in (x, nrm, sdf.reshape(-1))]
    f, g, h = hip.query_hessian(cfg, th, xd)
    ef, eg, eh = rel(f.cpu().numpy(), G[f"ww_{tag}_f64_y"]), rel(g.cpu().numpy(), G[f"ww_{tag}_f64_g"]), rel(h.cpu().numpy(), G[f"ww_{tag}_f64_H"])
    f2, g2 = hip.query(cfg, th, xd)                       # the plain (non-quad) sweeps
    assert rel(f2.cpu().numpy(), G[f"ww_{tag}_f64_y"]) < 5e-6 and rel(g2.cpu().numpy(), G[f"ww_{tag}_f64_g"]) < 2e-5
    msg = [f"ww {hid[0]}x{len(hid)}: f {ef:.1e} g {eg:.1e} H {eh:.1e}"]
    assert ef < 5e-6 and eg < 2e-5 and eh < 2e-5
    n_on = n // 3
    for name, w, tol in (("s1eik", [1e4, 1e4, 0.0, 1e3], 1e-4), ("s1full", [1e4, 1e4, 1e4, 1e3], 5e-4)):
        nh = n_on if w[2] else 0
        ws = hip.workspace_for(cfg, n, "cuda", n_hess=nh) if nh else hip.workspace_for(cfg, n, "cuda")
        kw = {"n_hess": nh} if nh else {}
        terms = hip.loss_forward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, w, 100.0, ws, **kw).cpu().numpy()
        dth = hip.loss_backward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, w, 100.0, torch.ones(4, device="cuda"), None, ws, **kw).cpu().numpy()
        et = rel(terms, G[f"ww_{tag}_f64_{name}_terms"])
        if tag == "tiny":                                  # [32]*3 runs at the built width 32: theta has the fixture's layout
            assert dth.size == G[f"ww_{tag}_f64_{name}_dtheta"].size
            ed = rel(dth, G[f"ww_{tag}_f64_{name}_dtheta"])
        else:
            ed = rel(dth[::97], G[f"ww_{tag}_f64_{name}_dtheta_sample"])
            assert abs(np.linalg.norm(dth.astype(np.float64)) / G[f"ww_{tag}_f64_{name}_dtheta_norm"][0] - 1) < tol
        # ... and the full d(theta) against the oracle with the frequency pair
        _, gr, _ = O.loss_and_grad("s1", P, x.astype(np.float64), nrm.astype(np.float64), sdf.astype(np.float64), w, 100.0, w0=(30.0, 15.0))
        flat = np.concatenate([np.concatenate([a.reshape(-1), b.reshape(-1)]) for a, b in gr])
        eo = rel(dth, flat) if dth.size == flat.size else None
        msg.append(f"{name}: terms {et:.1e} dtheta vs reference {ed if ed is None else format(ed, '.1e')} vs oracle {eo if eo is None else format(eo, '.1e')}")
        assert et < 1e-5
        assert ed is None or ed < tol
        assert eo is None or eo < tol
    print("; ".join(msg))


def test_ww_through_the_module_api(G):
    """SIREN(..., ww=15): forward / gradient / loss_s1(...).backward() through the reference-shaped Python API (tiny network,
    zero-padded to the next built width: the parameter gradients come back in the caller's shapes)."""
    from diffudf_amd.model import SIREN
    from diffudf_amd.diff_operators import gradient
    from diffudf_amd.loss_functions import loss_s1
    tag = "tiny"
    hid = list(G[f"ww_{tag}_hidden"]); n = int(G[f"ww_{tag}_n"]); seed = int(G[f"ww_{tag}_param_seed"])
    P = synth.siren_params(hid, seed=seed, w0=15.0, dtype=np.float64)
    m = SIREN(3, 1, hid, w0=30, ww=15)
    m.load_state_dict({f"net.{i}.0.{k}": torch.from_numpy(np.asarray(a, dtype=np.float32)) for i, (w, b) in enumerate(P)
                       for k, a in (("weight", w), ("bias", b))})
    m = m.cuda()
    x, nrm, sdf = [torch.from_numpy(a).cuda()[None] for a in synth.training_batch(n, seed=seed + 1)]
    out = m(x)
    y, xin = out["model_out"], out["model_in"]
    assert rel(y.detach().cpu().numpy()[0, :, 0], G[f"ww_{tag}_f64_y"]) < 5e-6
    assert rel(gradient(y, xin).detach().cpu().numpy()[0], G[f"ww_{tag}_f64_g"]) < 2e-5
    m.zero_grad()
    terms = loss_s1(m, x, {"normals": nrm, "sdf": sdf}, [1e4, 1e4, 0.0, 1e3], 100)
    total = torch.zeros((1, 1), device="cuda")
    for v in terms.values():
        total = total + v
    total.backward()
    got = np.concatenate([p.grad.detach().reshape(-1).cpu().numpy() for p in m.parameters()])
    assert rel([float(v) for v in terms.values()], G[f"ww_{tag}_f64_s1eik_terms"]) < 1e-5
    assert rel(got, G[f"ww_{tag}_f64_s1eik_dtheta"]) < 1e-4


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_latent_vector_through_evaluate(G, tag):
    from diffudf_amd.model import SIREN
    from diffudf_amd.evaluate import evaluate
    from diffudf_amd._lib import DudfError
    hid = list(G[f"lat_{tag}_hidden"]); k = int(G[f"lat_{tag}_k"]); seed = int(G[f"lat_{tag}_param_seed"])
    P = synth.siren_params(hid, seed=seed, n_in=3 + k, dtype=np.float64)
    m = SIREN(3 + k, 1, hid, w0=30)
    m.load_state_dict({f"net.{i}.0.{kk}": torch.from_numpy(np.asarray(a, dtype=np.float32)) for i, (w, b) in enumerate(P)
                       for kk, a in (("weight", w), ("bias", b))})
    m = m.cuda()
    xs = G[f"lat_{tag}_x"]
    lat = torch.from_numpy(G[f"lat_{tag}_latent"].astype(np.float32))
    grads = np.zeros((xs.shape[0], 3))
    vals = evaluate(m, xs, latent_vec=lat, gradients=grads)
    ev, eg = rel(vals[:, 0], G[f"lat_{tag}_f64_y"]), rel(grads, G[f"lat_{tag}_f64_g"])
    print(f"latent {hid[0]}x{len(hid)} k={k}: values {ev:.1e} gradients {eg:.1e}")
    assert ev < 5e-6 and eg < 2e-5
    with pytest.raises(DudfError):                       # the reference fails here too (make_golden.py::make_g11)
        evaluate(m, xs, latent_vec=lat, hessians=np.zeros((xs.shape[0], 3, 3)))
    with pytest.raises(DudfError):                       # training a latent-conditioned network is not a reference recipe
        m(torch.from_numpy(np.concatenate([np.repeat(lat.numpy(), xs.shape[0], 0), xs], 1)).cuda())


def test_ww_at_width_512_against_the_oracle():
    """`ww != w0` through the 512-wide kernel (its own tail burst and relay; R, E, C at 24 bits): value, df/dx, the `loss_s1` terms
    and d(theta) with and without the Hessian term against the oracle run with the frequency pair (pinned by the fixture cases
    above to the reference's own outputs)."""
    from diffudf_amd import hip_ops as hip
    hid, n, seed = [512] * 3, 700, 21
    P = synth.siren_params(hid, seed=seed, w0=15.0, dtype=np.float64)
    P32 = [(w.astype(np.float32), b.astype(np.float32)) for w, b in P]
    x, nrm, sdf = synth.training_batch(n, seed=seed + 1)
    cfg = hip.make_cfg(hid, 30.0, ww=15.0)
    th = torch.from_numpy(synth.flatten_params(P32)).cuda()
    xd, nd, sd = [torch.from_numpy(a).cuda() for a in (x, nrm, sdf.reshape(-1))]
    f, g = hip.query(cfg, th, xd)
    yo, go, _ = O.query(P, x.astype(np.float64), want_grad=True, want_hess=False, w0=(30.0, 15.0))
    assert rel(f.cpu().numpy(), yo) < 5e-6 and rel(g.cpu().numpy(), go) < 2e-5
    n_on = int((sdf.reshape(-1) == 0).sum())
    for name, w, tol in (("s1eik", [1e4, 1e4, 0.0, 1e3], 1e-4), ("s1full", [1e4, 1e4, 1e4, 1e3], 5e-4)):
        nh = n_on if w[2] else 0
        ws = hip.workspace_for(cfg, n, "cuda", n_hess=nh) if nh else hip.workspace_for(cfg, n, "cuda")
        kw = {"n_hess": nh} if nh else {}
        terms = hip.loss_forward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, w, 100.0, ws, **kw).cpu().numpy()
        dth = hip.loss_backward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, w, 100.0, torch.ones(4, device="cuda"), None, ws, **kw).cpu().numpy()
        t_ref, gr, _ = O.loss_and_grad("s1", P, x.astype(np.float64), nrm.astype(np.float64), sdf.astype(np.float64), w, 100.0, w0=(30.0, 15.0))
        flat = np.concatenate([np.concatenate([a.reshape(-1), b.reshape(-1)]) for a, b in gr])
        et, ed = rel(terms, np.array([float(v) for v in t_ref.values()])), rel(dth, flat)
        print(f"ww 512x3 {name}: stash mode {hip.stash_mode(cfg)} terms {et:.1e} dtheta {ed:.1e}")
        assert et < 1e-5 and ed < tol
