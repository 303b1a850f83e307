# coding: utf-8
"""GPU: oracle-direct parity at BASELINE.json's full sizes — the step `bench.py` times, against the fp64 oracle.

    SIREN 8x256 at 100 000 points (the headline metric) and 8x512 at 125 000 points (config 3's per-GPU share):
      * the four `loss_s1` terms and the FULL d(theta) against `oracle.dudf_oracle.loss_and_grad` in fp64 (torch CPU
        backend, the one `bench.py::cpu_baseline` runs), evaluated in point chunks: every term of `loss_s1` is a mean
        over points, so the batch result is the n_chunk/N-weighted sum of the chunk results;
      * every stashed per-point intermediate (s, c, q of the forward / reverse sweeps; A, e, zbar of the adjoint sweeps)
        of one random column out of EVERY 128-column pass of every workgroup, all layers.
    Tolerances are the ones of tests/test_hip_parity.py (reference: src/loss_functions.py:123-155, src/model.py:94-108).
"""
import numpy as np
import pytest
import torch

from diffudf_amd import synth
from oracle import dudf_oracle as O

pytestmark = pytest.mark.gpu
W = [1e4, 1e4, 0.0, 1e3]
TOL_TERM, TOL_DTHETA = 1e-5, 1e-4
CHUNK = 12500


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def disambiguate(P64, x, nrm, sdf, alpha=100.0):
    """`loss_s1` is a sum of ABSOLUTE values: |y| on the surface, |tdf - y| off it, | |grad f| - tau | everywhere
    (reference src/loss_functions.py:9-22, :136-137).  A point that sits within fp32 noise of one of those kinks gets the
    opposite sign of a whole 1/N cotangent in fp32 and in fp64 — at 100 000 points a handful always do, and ONE flip moves
    db_out by 6e-5 of its size (measured; the reference's own fp32 run flips the same way against its fp64 run).  Such
    points say nothing about the kernels, so they are replaced by copies of their stratum's first unambiguous point: same
    batch size, same on / far / near thirds."""
    n = x.shape[0]
    Pt = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in P64]
    ys, gs = [], []
    with torch.no_grad():
        for lo in range(0, n, CHUNK):
            y, g, _ = O.query(Pt, torch.from_numpy(x[lo:lo + CHUNK].astype(np.float64)), xp=torch)
            ys.append(y.numpy()); gs.append(g.numpy())
    y, g = np.concatenate(ys), np.concatenate(gs)
    u = sdf[:, 0].astype(np.float64)
    tan = np.tanh(alpha * u)
    tau = np.abs(tan + u * alpha * (1.0 - tan * tan))
    gn = np.linalg.norm(g, axis=1)
    margin_y = np.where(u == 0, np.abs(y), np.abs(u * tan - y))
    bad = (margin_y < 2e-5 * np.abs(y).max()) | (np.abs(gn - tau) < 2e-4 * gn.max())
    x, nrm, sdf = x.copy(), nrm.copy(), sdf.copy()
    third = n // 3
    for a, b in ((0, third), (third, 2 * third), (2 * third, n)):
        good = a + int(np.flatnonzero(~bad[a:b])[0])
        idx = a + np.flatnonzero(bad[a:b])
        x[idx], nrm[idx], sdf[idx] = x[good], nrm[good], sdf[good]
    return x, nrm, sdf, int(bad.sum())


def oracle_full(P64, x, nrm, sdf, n):
    """loss terms (4,) and flat d(theta) of the whole batch, fp64, chunked over points."""
    Pt = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in P64]
    terms = np.zeros(4)
    grad = None
    with torch.no_grad():
        for lo in range(0, n, CHUNK):
            hi = min(n, lo + CHUNK)
            xs, ns, ss = [torch.from_numpy(a[lo:hi].astype(np.float64)) for a in (x, nrm, sdf)]
            t, g, _ = O.loss_and_grad("s1", Pt, xs, ns, ss, W, 100.0, xp=torch)
            f = (hi - lo) / n
            terms += f * np.array([float(v) for v in t.values()])
            flat = torch.cat([torch.cat([w.reshape(-1), b.reshape(-1)]) for w, b in g]).numpy() * f
            grad = flat if grad is None else grad + flat
    return terms, grad


@pytest.mark.parametrize("hidden,n", [([256] * 8, 100000), ([512] * 8, 125000)])
def test_step_against_oracle_at_full_size(hidden, n):
    from diffudf_amd import hip_ops as hip
    P32 = synth.siren_params(hidden, seed=123)
    P64 = [(w.astype(np.float64), b.astype(np.float64)) for w, b in P32]
    x, nrm, sdf = synth.training_batch(n, seed=124)
    x, nrm, sdf, n_amb = disambiguate(P64, x, nrm, sdf)
    assert n_amb < n // 100
    th = torch.from_numpy(synth.flatten_params(P32)).cuda()
    xd, nd, sd = [torch.from_numpy(a).cuda() for a in (x, nrm, sdf.reshape(-1))]
    cfg = hip.make_cfg(hidden)
    ws = hip.workspace_for(cfg, n, "cuda")
    terms = hip.loss_forward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, W, 100.0, ws).double().cpu().numpy()
    dth = hip.loss_backward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, W, 100.0, torch.ones(4, device="cuda"), None, ws)
    dth = dth.double().cpu().numpy()

    # one random column of every 128-column pass (a pass = the 8 waves x 16 columns a workgroup walks together)
    rng = np.random.default_rng(5)
    starts = np.arange(0, n, 128)
    idx = np.minimum(starts + rng.integers(0, 128, starts.size), n - 1)
    # ... plus every column of a few ONE-group passes (a workgroup's share of 16-column groups is walked in passes of 8; the
    # last pass of 112 workgroups holds a single group at 100 000 points, which all eight waves then share: "oct mode")
    ng = (n + 127) // 128 * 8
    lone = [b for b in range(256) if ((b + 1) * ng // 256 - b * ng // 256) % 8 == 1][:3]
    extra = np.concatenate([np.arange(16) + 16 * ((b + 1) * ng // 256 - 1) for b in lone]) if lone else np.zeros(0, int)
    idx = np.unique(np.concatenate([idx, extra[extra < n]])).astype(np.int64)
    L = len(hidden)
    got = {}
    for name in ("s", "c", "q", "A", "e", "zbar"):
        for l in range(L):
            got[name, l] = hip.read_stash(cfg, name, l, n, ws)[torch.from_numpy(idx).cuda()].double().cpu().numpy()
    sub = [a[idx].astype(np.float64) for a in (x, nrm, sdf)]
    _, _, dbg = O.loss_and_grad("s1", P64, *sub, W, 100.0)
    f = idx.size / n                                     # the oracle's cotangents carry 1 / (its own batch size)
    worst = {}
    for l in range(L):
        for name, ref, scale, tol in (("s", dbg["cache"]["s"][l], 1.0, 5e-5), ("c", dbg["cache"]["c"][l], 1.0, 5e-5),
                                      ("q", dbg["rev"]["q"][l], 1.0, 5e-5), ("A", dbg["trace"]["A"][l], f, 2e-4),
                                      ("e", dbg["trace"]["e"][l], f, 2e-4), ("zbar", dbg["trace"]["zbar"][l], f, 2e-4)):
            e = rel(got[name, l], ref * scale)
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < tol, f"{hidden[0]}x{L} n={n}: stash {name}[{l}] rel err {e:.2e} on {idx.size} columns"

    t_ref, g_ref = oracle_full(P64, x, nrm, sdf, n)
    et, ed = rel(terms, t_ref), rel(dth, g_ref)
    # per parameter tensor too: a wrong thin layer or bias must not hide under the max-norm of the hidden matrices
    H = hidden[0]
    offs, o = [], 0
    for w_, b_ in P32:
        offs.append((o, o + w_.size)); o += w_.size
        offs.append((o, o + b_.size)); o += b_.size
    per = max(rel(dth[a:b], g_ref[a:b]) for a, b in offs)
    print(f"full size {H}x{L} n={n} ({n_amb} points at a kink of the loss replaced): terms {et:.2e} dtheta {ed:.2e} (worst single tensor {per:.2e}); stash "
          + " ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    assert et < TOL_TERM
    assert ed < TOL_DTHETA
    assert per < TOL_DTHETA


# ---------------------------------------------------------------------------------------------------------------------
# The reference's SHIPPED loss at its own batch: configs/train_cfg.json — 30000 x [0.333, 0.666] = 29 970 points of which the
# 9 990 on-surface ones carry the Hessian / eigenvector term, weights [1e4, 1e4, 1e4, 1e3] (reference
# src/loss_functions.py:139-145, train.py:204-210).  Default build: one pair launch per sweep (Hessian quads + plain columns in
# one grid), quads on fp16x3.
W_FULL = [1e4, 1e4, 1e4, 1e3]


def disambiguate_full(P64, x, nrm, sdf, n_on):
    """`disambiguate` plus the on-surface points the eigenvector term makes ill-conditioned in fp32: the top eigenvector's
    cotangent carries 1 / (lambda_2 - lambda_j) (torch.linalg.eigh's backward, reference src/loss_functions.py:142), so a
    point next to a degenerate Hessian amplifies the Hessian's own fp32 rounding without bound, and 1 - |cos| has a kink at
    cos = 0.  Returns the cleaned batch, the number of replaced points and the relative eigen-gap of every on-surface point."""
    x, nrm, sdf, n_kink = disambiguate(P64, x, nrm, sdf)
    Pt = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in P64]
    lam, cs = [], []
    with torch.no_grad():
        for lo in range(0, n_on, 2500):
            hi = min(n_on, lo + 2500)
            _, _, H = O.query(Pt, torch.from_numpy(x[lo:hi].astype(np.float64)), want_hess=True, xp=torch)
            l_, V = torch.linalg.eigh(H)
            m = torch.from_numpy(nrm[lo:hi].astype(np.float64))
            cs.append((torch.nn.functional.cosine_similarity(m, V[..., 2], dim=-1)).numpy())
            lam.append(l_.numpy())
    lam, cs = np.concatenate(lam), np.concatenate(cs)
    gap = (lam[:, 2] - lam[:, 1]) / np.abs(lam).max(axis=1)
    bad = (gap < 0.05) | (np.abs(cs) < 1e-3)
    good = int(np.flatnonzero(~bad & (gap > 0.3))[0])
    idx = np.flatnonzero(bad)
    x[idx], nrm[idx], sdf[idx] = x[good], nrm[good], sdf[good]
    gap[idx] = gap[good]
    return x, nrm, sdf, n_kink + idx.size, gap


def oracle_full_hessian(P64, x, nrm, sdf, n, n_on, chunk=2500):
    """`oracle_full` with the Hessian term on.  A chunk of off-surface points only is evaluated with the Hessian weight set to
    zero: their eigenvector cotangent is identically zero (the reference's torch.where, src/loss_functions.py:48-52), so the
    result is the same and the oracle skips three quarters of its work there."""
    Pt = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in P64]
    terms, grad = np.zeros(4), None
    cuts = list(range(0, n_on, chunk)) + list(range(n_on, n, 4 * chunk)) + [n]
    with torch.no_grad():
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            hi = min(hi, n_on) if lo < n_on else hi
            xs, ns, ss = [torch.from_numpy(a[lo:hi].astype(np.float64)) for a in (x, nrm, sdf)]
            t, g, _ = O.loss_and_grad("s1", Pt, xs, ns, ss, W_FULL if lo < n_on else W, 100.0, xp=torch)
            f = (hi - lo) / n
            terms += f * np.array([float(v) for v in t.values()])
            flat = torch.cat([torch.cat([w.reshape(-1), b.reshape(-1)]) for w, b in g]).numpy() * f
            grad = flat if grad is None else grad + flat
    return terms, grad


def test_full_loss_s1_at_the_reference_batch():
    from diffudf_amd import hip_ops as hip
    hidden, n = [256] * 8, 29970
    n_on = n // 3
    P32 = synth.siren_params(hidden, seed=123)
    P64 = [(w.astype(np.float64), b.astype(np.float64)) for w, b in P32]
    x, nrm, sdf = synth.training_batch(n, seed=125)
    assert (sdf[:n_on, 0] == 0).all() and (sdf[n_on:, 0] != 0).all()
    x, nrm, sdf, n_amb, gap = disambiguate_full(P64, x, nrm, sdf, n_on)
    assert n_amb < n // 20
    th = torch.from_numpy(synth.flatten_params(P32)).cuda()
    xd, nd, sd = [torch.from_numpy(a).cuda() for a in (x, nrm, sdf.reshape(-1))]
    cfg = hip.make_cfg(hidden)
    ws = hip.workspace_for(cfg, n, "cuda", n_hess=n_on)
    terms = hip.loss_forward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, W_FULL, 100.0, ws, n_hess=n_on).double().cpu().numpy()
    dth = hip.loss_backward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, W_FULL, 100.0, torch.ones(4, device="cuda"), None, ws,
                            n_hess=n_on).double().cpu().numpy()

    # stash columns: one well-conditioned point of every 128-column pass of the QUAD part (32 points per pass) and one column
    # of every pass of the PLAIN part of the pair grid
    rng = np.random.default_rng(7)
    qi = []
    for lo in range(0, n_on, 32):
        cand = lo + np.flatnonzero(gap[lo:lo + 32] > 0.3)
        if cand.size:
            qi.append(int(rng.choice(cand)))
    qi = np.array(qi, dtype=np.int64)
    starts = np.arange(n_on, n, 128)
    pi = np.minimum(starts + rng.integers(0, 128, starts.size), n - 1).astype(np.int64)
    idx = np.concatenate([qi, pi])
    nq = qi.size
    L = len(hidden)
    sel = torch.from_numpy(idx).cuda()
    got = {}
    for name in ("s", "c", "q", "r", "A", "e", "zbar", "zs"):
        for l in range(L):
            for ch in range(4):
                if ch and name == "c":
                    continue
                t = hip.read_stash(cfg, name, l, n, ws, channel=ch)[sel].double().cpu().numpy()
                got[name, l, ch] = t
    sub = [a[idx].astype(np.float64) for a in (x, nrm, sdf)]
    _, _, dbg = O.loss_and_grad("s1", P64, *sub, W_FULL, 100.0)
    f = idx.size / n
    s_, c_, a_, q_ = dbg["cache"]["s"], dbg["cache"]["c"], dbg["rev"]["a"], dbg["rev"]["q"]
    zd, ad, tr = dbg["tang"]["zd"], dbg["tang"]["ad"], dbg["trace"]
    w0 = 30.0
    worst = {}

    def chk(tag, a, b, tol):
        e = rel(a, b)
        worst[tag] = max(worst.get(tag, 0.0), e)
        assert e < tol, f"{tag}: rel err {e:.2e} (tol {tol:.0e}) on {a.shape[0]} columns"

    Q, Pn = slice(0, nq), slice(nq, None)                # quad points | plain points among the selected ones
    for l in range(L):
        # ---- Hessian quads (channel 0 = value path, 1 + k = d/dx_k; dudf_sweep_common.h::epilogue)
        chk("quad c", got["c", l, 0][Q], c_[l][Q], 5e-5)
        chk("quad s", got["s", l, 0][Q], s_[l][Q], 5e-5)
        chk("quad zs0", got["zs", l, 0][Q], s_[l][Q], 5e-5)
        chk("quad q", got["q", l, 0][Q], q_[l][Q], 5e-5)
        chk("quad a", got["r", l, 0][Q], a_[l][Q], 5e-5)
        chk("quad A", got["A", l, 0][Q], tr["A"][l][Q] * f, 2e-4)
        chk("quad E", got["e", l, 0][Q], tr["E"][l][Q] * f, 2e-4)
        chk("quad zbar", got["zbar", l, 0][Q], tr["zbar"][l][Q] * f, 2e-4)
        for k in range(3):
            cd = -w0 * s_[l] * zd[k][l]
            qd = w0 * (cd * a_[l] + c_[l] * ad[k][l])
            chk("quad hdot", got["s", l, 1 + k][Q], (w0 * c_[l] * zd[k][l])[Q], 2e-4)
            chk("quad zdot", got["zs", l, 1 + k][Q], zd[k][l][Q], 2e-4)
            chk("quad qdot", got["q", l, 1 + k][Q], qd[Q], 2e-4)
            chk("quad adot", got["r", l, 1 + k][Q], ad[k][l][Q], 2e-4)
            chk("quad Adot", got["A", l, 1 + k][Q], tr["Ad"][k][l][Q] * f, 2e-4)
            chk("quad Edot", got["e", l, 1 + k][Q], tr["Ed"][k][l][Q] * f, 2e-4)
            chk("quad zdbar", got["zbar", l, 1 + k][Q], tr["zdbar"][k][l][Q] * f, 2e-4)
        # ---- plain columns (off-surface points: their Hessian cotangent is zero, so the quad formulas reduce to the plain ones)
        chk("plain s", got["s", l, 0][Pn], s_[l][Pn], 5e-5)
        chk("plain c", got["c", l, 0][Pn], c_[l][Pn], 5e-5)
        chk("plain q", got["q", l, 0][Pn], q_[l][Pn], 5e-5)
        chk("plain A", got["A", l, 0][Pn], tr["A"][l][Pn] * f, 2e-4)
        chk("plain e", got["e", l, 0][Pn], -tr["E"][l][Pn] * f, 2e-4)
        chk("plain zbar", got["zbar", l, 0][Pn], tr["zbar"][l][Pn] * f, 2e-4)

    t_ref, g_ref = oracle_full_hessian(P64, x, nrm, sdf, n, n_on)
    et, ed = rel(terms, t_ref), rel(dth, g_ref)
    offs, o = [], 0
    for w_, b_ in P32:
        offs.append((o, o + w_.size)); o += w_.size
        offs.append((o, o + b_.size)); o += b_.size
    per = max(rel(dth[a:b], g_ref[a:b]) for a, b in offs)
    print(f"full loss_s1 8x256 n={n} ({n_on} on quads; {n_amb} points at a kink / next to a degenerate Hessian replaced): terms "
          f"{et:.2e} (each: {np.array2string(np.abs(terms - t_ref) / np.abs(t_ref), precision=1)}) dtheta {ed:.2e} (worst single "
          f"tensor {per:.2e}); stash " + " ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    assert et < TOL_TERM
    assert ed < 5e-4 and per < 5e-4
