# coding: utf-8
"""ORACLE — test infrastructure only.  Never imported by the product path.

CPU restatement of DiffUDF's training hot path (reference = LIA-DiTella/DiffUDF):
SIREN forward, the input derivatives the reference obtains by repeated
`torch.autograd.grad`, the hyperbolic-scaled UDF losses, and `loss.backward()`
to the parameters — written out as explicit per-layer recurrences (SURVEY.md
Appendix A) instead of an autograd tape, so that every intermediate the HIP
kernels produce can be checked one by one.

Parity pin: this file is checked against golden vectors produced by importing
the reference itself (`tests/golden/make_golden.py`, fixtures `tests/golden/*.npz`)
in `tests/test_oracle_golden.py`.  The reference has no tests / golden vectors of
its own, so those generated fixtures are the pin.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.

Array backend: every function takes `xp` = `numpy` (default, any float dtype) or
`torch` (CPU tensors; used for the multi-threaded cpu_baseline timing).  Only
operators and functions that exist under the same name in both are used.

Conventions: params = [(W_1,b_1), ..., (W_L,b_L), (W_out,b_out)], W (out,in)
row-major as `nn.Linear` stores it; x (N,3); everything is per point, no leading
batch-of-1 dimension (the reference carries one: (1,N,3)).
"""
import numpy as np


def _wl(w0, l):
    """Frequency of hidden layer l (0-based): `w0` is one number, or the pair (w0, ww) of reference src/model.py:89-106 —
    the first SineLayer runs w0, every other one ww."""
    if isinstance(w0, (tuple, list)):
        return w0[0] if l == 0 else w0[1]
    return w0


# --------------------------------------------------------------------------
# forward value  —  reference src/model.py:29-30 (SineLayer), :131-135 (SIREN.forward)
# --------------------------------------------------------------------------
def forward(params, x, w0=30.0, xp=np):
    """y (N,), cache = {'s': [s_1..s_L], 'c': [c_1..c_L]} with s_l = sin(w0 z_l), c_l = cos(w0 z_l)."""
    h = x
    S, C = [], []
    for l, (W, b) in enumerate(params[:-1]):
        z = xp.matmul(h, W.T) + b            # nn.Linear
        t = _wl(w0, l) * z                   # SineLayer: torch.sin(self.w0 * x)
        s = xp.sin(t)
        S.append(s)
        C.append(xp.cos(t))
        h = s
    Wo, bo = params[-1]
    y = (xp.matmul(h, Wo.T) + bo)[:, 0]
    return y, {"s": S, "c": C}


# --------------------------------------------------------------------------
# dy/dx by one reverse sweep  —  reference src/diff_operators.py:208-212 `gradient`
# --------------------------------------------------------------------------
def input_gradient(params, cache, w0=30.0, xp=np):
    """g (N,3) and the per-layer reverse quantities a_l (adjoint of h_l), q_l = w0 c_l a_l."""
    L = len(params) - 1
    N = cache["c"][0].shape[0]
    Wo = params[-1][0]
    a = Wo[0][None, :] + 0.0 * cache["c"][-1]          # a_L = W_out^T broadcast to (N,H)
    A, Q = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        A[l] = a
        q = _wl(w0, l) * cache["c"][l] * a
        Q[l] = q
        a = xp.matmul(q, params[l][0])                  # a_{l-1} = W_l^T q_l
    return a, {"a": A, "q": Q}


# --------------------------------------------------------------------------
# Hessian  —  reference src/diff_operators.py:187-193 `hessian` (rows h_i = grad(g[:,i], x))
# forward-over-reverse with three tangents
# --------------------------------------------------------------------------
def hessian(params, x, cache, rev, w0=30.0, xp=np):
    """H (N,3,3) with H[n,i,k] = d(df/dx_i)/dx_k, plus the tangent caches (zd[k][l], ad[k][l])."""
    L = len(params) - 1
    cols, zd_all, ad_all = [], [], []
    for k in range(3):
        hd = 0.0 * x                                     # forward tangents: hdot_0 = e_k
        hd[:, k] = 1.0
        zd = []
        for l in range(L):
            zk = xp.matmul(hd, params[l][0].T)          # zdot_l = W_l hdot_{l-1}
            zd.append(zk)
            hd = _wl(w0, l) * cache["c"][l] * zk       # hdot_l = w0 c_l zdot_l
        ad = 0.0 * cache["c"][-1]                        # reverse tangents: adot_L = 0
        adk = [None] * L
        for l in range(L - 1, -1, -1):
            adk[l] = ad
            cd = -_wl(w0, l) * cache["s"][l] * zd[l]   # d c_l / d x_k
            qd = _wl(w0, l) * (cd * rev["a"][l] + cache["c"][l] * ad)
            ad = xp.matmul(qd, params[l][0])            # adot_{l-1} = W_l^T qdot_l
        cols.append(ad)                                 # (N,3) indexed by i: column k of H
        zd_all.append(zd)
        ad_all.append(adk)
    H = xp.stack(cols, -1)
    return H, {"zd": zd_all, "ad": ad_all}


# --------------------------------------------------------------------------
# small helpers shared by the losses
# --------------------------------------------------------------------------
def _norm(v, xp):
    return xp.sqrt((v * v).sum(-1))


def _cos_sim(a, b, xp, eps=1e-8):
    """F.cosine_similarity(a, b, dim=-1): each norm clamped to eps (SURVEY.md §8(a) A7 [probe])."""
    na = xp.clip(_norm(a, xp), eps, None)
    nb = xp.clip(_norm(b, xp), eps, None)
    return (a * b).sum(-1) / (na * nb)


def top_eigvec(H, xp=np):
    """Eigenvector of the largest eigenvalue, lower triangle only, ascending order —
    reference src/loss_functions.py:141-143 (`torch.linalg.eigh`, `eigenvectors[..., 2]`)."""
    if xp is np:
        lam, V = np.linalg.eigh(H)          # UPLO='L'
    else:
        lam, V = xp.linalg.eigh(H)
    return lam, V


# --------------------------------------------------------------------------
# loss_s1  —  reference src/loss_functions.py:123-155 (+ :9-22, :45-53)
# --------------------------------------------------------------------------
def loss_s1_terms(y, g, H, normals, sdf, weights, alpha, xp=np):
    """Dict of the four weighted scalars + the per-point cotangents (ybar, gbar[, nbar]) of their SUM.

    `H` may be None when weights[2] == 0 (the reference then skips the Hessian entirely).
    """
    N = y.shape[0]
    u = sdf[:, 0]
    on = (u == 0)
    tan = xp.tanh(alpha * u)
    tdf = u * tan
    zero = 0.0 * y
    t_on = xp.where(on, xp.abs(y), zero)                              # sdf_constraint_on_surf  :9-14
    t_off = xp.where(on, zero, xp.abs(tdf - y))                       # sdf_constraint_off_surf :17-22
    out = {
        "sdf_on_surf": t_on.mean() * weights[0],
        "sdf_off_surf": t_off.mean() * weights[1],
    }
    ybar = (weights[0] / N) * xp.where(on, xp.sign(y), zero) - (weights[1] / N) * xp.where(on, zero, xp.sign(tdf - y))
    cot = {"ybar": ybar}
    if weights[2] != 0:
        lam, V = top_eigvec(H, xp)
        n = V[..., 2]
        cs = _cos_sim(normals, n, xp)
        t_h = xp.where(on, 1.0 - xp.abs(cs), zero)                    # principal_curvature_alignment :45-53
        out["hessian_constraint"] = t_h.mean() * weights[2]
        cot["lam"], cot["V"], cot["cos"] = lam, V, cs
        # cotangent of the normal n = V[...,2], then through eigh:
        #   nbar = -(w2/N) sign(cos) [u=0] ( m/(|m||n|) - cos n/|n|^2 )
        #   Hbar = sum_{j=0,1} (v_j . nbar)/(lam_2 - lam_j) * 1/2 (v_j n^T + n v_j^T)      (symmetric, all 9 entries)
        mn = xp.clip(_norm(normals, xp), 1e-8, None)
        nn = xp.clip(_norm(n, xp), 1e-8, None)
        onf = xp.where(on, 1.0 + zero, zero)
        nbar = (-(weights[2] / N) * xp.sign(cs) * onf)[:, None] * (normals / (mn * nn)[:, None] - (cs / (nn * nn))[:, None] * n)
        Hbar = 0.0 * H
        for jj in range(2):
            vj = V[..., jj]
            coef = (vj * nbar).sum(-1) / (lam[..., 2] - lam[..., jj])
            Hbar = Hbar + (0.5 * coef)[:, None, None] * (vj[:, :, None] * n[:, None, :] + n[:, :, None] * vj[:, None, :])
        cot["nbar"], cot["Hbar"] = nbar, Hbar
    else:
        out["hessian_constraint"] = 0.0 * out["sdf_on_surf"]          # torch.Tensor([0])  :147
    if weights[3] != 0:
        gn = _norm(g, xp)
        target = xp.abs(tan + u * alpha * (1.0 - tan * tan))         # :136
        d = gn - target
        out["grad_constraint"] = xp.abs(d).mean() * weights[3]
        safe = xp.where(gn > 0, gn, 1.0 + zero)
        cot["gbar"] = ((weights[3] / N) * xp.sign(d) * xp.where(gn > 0, 1.0 / safe, zero))[:, None] * g
    else:
        out["grad_constraint"] = 0.0 * out["sdf_on_surf"]
        cot["gbar"] = None
    # keep reference key order (losses.csv column order): on, off, hessian, grad
    out = {k: out[k] for k in ("sdf_on_surf", "sdf_off_surf", "hessian_constraint", "grad_constraint")}
    return out, cot


# --------------------------------------------------------------------------
# loss_siren  —  reference src/loss_functions.py:82-104 (+ :24-32)
# --------------------------------------------------------------------------
def loss_siren_terms(y, g, normals, sdf, weights, xp=np):
    N = y.shape[0]
    s = sdf[:, 0]
    on = (s == 0)
    zero = 0.0 * y
    ay = xp.abs(y)
    ex = xp.exp(-1e2 * ay)
    gn = _norm(g, xp)
    mn = _norm(normals, xp)
    cs = _cos_sim(g, normals, xp)
    out = {
        "sdf_on_surf": xp.where(on, ay, zero).mean() * weights[0],
        "sdf_off_surf": xp.where(on, zero, ex).mean() * weights[1],
        "normal_constraint": xp.where(on, 1.0 - cs, zero).mean() * weights[2],
        "grad_constraint": ((gn - 1.0) ** 2).mean() * weights[3],
    }
    ybar = (weights[0] / N) * xp.where(on, xp.sign(y), zero) - (weights[1] / N) * xp.where(on, zero, 1e2 * xp.sign(y) * ex)
    gnc = xp.clip(gn, 1e-8, None)
    mnc = xp.clip(mn, 1e-8, None)
    # d cos / d g with the clamped-norm form; inside the clamp (|g| < eps) the norm factor is constant
    dcos = normals / (gnc * mnc)[:, None] - xp.where(gn > 1e-8, cs / (gnc * gnc), zero)[:, None] * g
    safe = xp.where(gn > 0, gn, 1.0 + zero)
    gbar = -(weights[2] / N) * xp.where(on, 1.0 + zero, zero)[:, None] * dcos \
        + ((weights[3] / N) * 2.0 * (gn - 1.0) * xp.where(gn > 0, 1.0 / safe, zero))[:, None] * g
    return out, {"ybar": ybar, "gbar": gbar}


# --------------------------------------------------------------------------
# loss_s2  —  reference src/loss_functions.py:106-121
# --------------------------------------------------------------------------
def loss_s2_terms(y, sdf, weights, xp=np, stats=None):
    """`stats` = (n, sum, sumsq) of the on-surface predictions over the GLOBAL batch; when None it
    is formed from this batch (single-rank case)."""
    u = sdf[:, 0]
    on = (u == 0)
    zero = 0.0 * y
    p = xp.where(on, y, zero)
    if stats is None:
        n = on.sum() * 1.0
        sm = p.sum()
        sq = (p * p).sum()
    else:
        n, sm, sq = stats
    mu = sm / n
    var = (sq - n * mu * mu) / (n - 1.0)                  # torch.std: unbiased
    if stats is None:                                      # two-pass form when the data is local (better conditioned)
        var = (xp.where(on, (y - mu) ** 2, zero)).sum() / (n - 1.0)
    sd = xp.sqrt(var)
    out = {"sdf_on_surf": xp.abs(mu) * weights[0], "std_on_surf": sd * weights[1]}
    ybar = xp.where(on, weights[0] * xp.sign(mu) / n + weights[1] * (y - mu) / ((n - 1.0) * sd), zero)
    return out, {"ybar": ybar, "gbar": None}


# --------------------------------------------------------------------------
# loss.backward() to the parameters  —  reference train.py:221 through
# src/loss_functions.py / src/diff_operators.py graphs; SURVEY.md Appendix A.5, Eikonal subset
# --------------------------------------------------------------------------
def param_grad(params, x, cache, rev, ybar, gbar, w0=30.0, xp=np):
    """d(sum of loss terms)/d(params) for cotangents ybar (N,) on y and gbar (N,3) on df/dx (or None).

    Returns (grads, trace): grads = [(dW_l, db_l)...] in `params` order; trace holds the per-layer
    intermediates (A_l, e_l, zbar_l) the HIP sweeps 3 and 4 stash, for stage-by-stage checks.
    """
    L = len(params) - 1
    dW = [0.0 * W for W, _ in params]
    db = [0.0 * b for _, b in params]
    s, c = cache["s"], cache["c"]
    cbar = [None] * L
    trace = {"A": [None] * L, "e": [None] * L, "zbar": [None] * L}
    if gbar is not None:
        # (i) adjoint of the reverse sweep, runs forward in l
        Aprev = gbar
        for l in range(L):
            Ql = xp.matmul(Aprev, params[l][0].T)               # Q_l = W_l A_{l-1}
            dW[l] = dW[l] + xp.matmul(rev["q"][l].T, Aprev)      # q_l A_{l-1}^T summed over points
            cbar[l] = _wl(w0, l) * rev["a"][l] * Ql
            trace["e"][l] = _wl(w0, l) * s[l] * cbar[l]          # = w0^2 s_l a_l Q_l
            Aprev = _wl(w0, l) * c[l] * Ql
            trace["A"][l] = Aprev
        dW[L] = dW[L] + Aprev.sum(0)[None, :]
    # (ii) adjoint of the forward sweep, runs backward in l
    Wo = params[-1][0]
    hbar = ybar[:, None] * Wo[0][None, :]
    dW[L] = dW[L] + xp.matmul(ybar[None, :], s[L - 1])
    db[L] = db[L] + ybar.sum()[None]
    for l in range(L - 1, -1, -1):
        zbar = _wl(w0, l) * c[l] * hbar
        if cbar[l] is not None:
            zbar = zbar - _wl(w0, l) * s[l] * cbar[l]
        trace["zbar"][l] = zbar
        hprev = x if l == 0 else s[l - 1]
        dW[l] = dW[l] + xp.matmul(zbar.T, hprev)
        db[l] = db[l] + zbar.sum(0)
        if l > 0:
            hbar = xp.matmul(zbar, params[l][0])
    return list(zip(dW, db)), trace


def param_grad_hessian(params, x, cache, rev, tang, ybar, gbar, Hbar, w0=30.0, xp=np, want_trace=False):
    """Full SURVEY.md Appendix A.5: d(loss)/d(params) when the loss also depends on the Hessian
    (cotangent Hbar (N,3,3) on H[n,i,k] = d(df/dx_i)/dx_k), i.e. `backward()` through
    reference src/diff_operators.py:187-193 + torch.linalg.eigh (src/loss_functions.py:141-145).
    `tang` is the second return value of `hessian()` (zd[k][l], ad[k][l]).
    want_trace: also return the per-layer adjoints the parity tests compare a kernel's stash with — value channel
    A[l], E[l] (= w0 c sbar_rev - w0 s cbar_rev; e_l of the plain path is its negative where the tangents vanish), zbar[l];
    tangent channels Ad[k][l], Ed[k][l] (= zdotbar_rev), zdbar[k][l]."""
    L = len(params) - 1
    s, c = cache["s"], cache["c"]
    a = rev["a"]
    dW = [0.0 * W for W, _ in params]
    db = [0.0 * b for _, b in params]
    zd, ad = tang["zd"], tang["ad"]
    # forward tangents hdot_{l-1}^k (inputs of layer l) and reverse tangents' qdot_l^k
    hd = [[None] * (L + 1) for _ in range(3)]
    qd = [[None] * L for _ in range(3)]
    cd = [[None] * L for _ in range(3)]
    for k in range(3):
        e = 0.0 * x
        e[:, k] = 1.0
        hd[k][0] = e
        for l in range(L):
            hd[k][l + 1] = _wl(w0, l) * c[l] * zd[k][l]
            cd[k][l] = -_wl(w0, l) * s[l] * zd[k][l]
            qd[k][l] = _wl(w0, l) * (cd[k][l] * a[l] + c[l] * ad[k][l])
    # (i) adjoint of the reverse sweeps, forward in l
    Ap = gbar if gbar is not None else 0.0 * x
    Adp = [Hbar[:, :, k] for k in range(3)]
    cbar_rev, sbar_rev, zdbar_rev = [None] * L, [None] * L, [[None] * L for _ in range(3)]
    tr = {"A": [None] * L, "Ad": [[None] * L for _ in range(3)], "E": [None] * L, "Ed": zdbar_rev, "zbar": [None] * L,
          "zdbar": [[None] * L for _ in range(3)]}
    for l in range(L):
        W = params[l][0]
        Q = xp.matmul(Ap, W.T)
        Qd = [xp.matmul(Adp[k], W.T) for k in range(3)]
        dW[l] = dW[l] + xp.matmul(rev["q"][l].T, Ap)
        for k in range(3):
            dW[l] = dW[l] + xp.matmul(qd[k][l].T, Adp[k])
        chat = [_wl(w0, l) * a[l] * Qd[k] for k in range(3)]
        Anew = _wl(w0, l) * c[l] * Q
        cb = _wl(w0, l) * a[l] * Q
        sb = 0.0 * Q
        for k in range(3):
            Anew = Anew + _wl(w0, l) * cd[k][l] * Qd[k]
            cb = cb + _wl(w0, l) * ad[k][l] * Qd[k]
            sb = sb - _wl(w0, l) * zd[k][l] * chat[k]
            zdbar_rev[k][l] = -_wl(w0, l) * s[l] * chat[k]
        cbar_rev[l], sbar_rev[l] = cb, sb
        Adp = [_wl(w0, l) * c[l] * Qd[k] for k in range(3)]
        Ap = Anew
        tr["A"][l] = Ap
        tr["E"][l] = _wl(w0, l) * c[l] * sb - _wl(w0, l) * s[l] * cb
        for k in range(3):
            tr["Ad"][k][l] = Adp[k]
    dW[L] = dW[L] + Ap.sum(0)[None, :]
    # (ii) adjoint of the forward sweeps, backward in l
    Wo = params[-1][0]
    hbar = ybar[:, None] * Wo[0][None, :]
    hdbar = [0.0 * hbar for _ in range(3)]
    dW[L] = dW[L] + xp.matmul(ybar[None, :], s[L - 1])
    db[L] = db[L] + ybar.sum()[None]
    for l in range(L - 1, -1, -1):
        W = params[l][0]
        zdbar = [zdbar_rev[k][l] + _wl(w0, l) * c[l] * hdbar[k] for k in range(3)]
        cb = cbar_rev[l]
        for k in range(3):
            cb = cb + _wl(w0, l) * zd[k][l] * hdbar[k]
        sb = sbar_rev[l] + hbar
        zbar = _wl(w0, l) * c[l] * sb - _wl(w0, l) * s[l] * cb
        tr["zbar"][l] = zbar
        for k in range(3):
            tr["zdbar"][k][l] = zdbar[k]
        hprev = x if l == 0 else s[l - 1]
        dW[l] = dW[l] + xp.matmul(zbar.T, hprev)
        for k in range(3):
            dW[l] = dW[l] + xp.matmul(zdbar[k].T, hd[k][l])
        db[l] = db[l] + zbar.sum(0)
        if l > 0:
            hbar = xp.matmul(zbar, W)
            hdbar = [xp.matmul(zdbar[k], W) for k in range(3)]
    return (list(zip(dW, db)), tr) if want_trace else list(zip(dW, db))


# --------------------------------------------------------------------------
# one-call conveniences used by the tests / cpu_baseline
# --------------------------------------------------------------------------
def query(params, x, w0=30.0, want_grad=True, want_hess=False, xp=np):
    y, cache = forward(params, x, w0, xp)
    g = H = None
    if want_grad or want_hess:
        g, rev = input_gradient(params, cache, w0, xp)
    if want_hess:
        H, _ = hessian(params, x, cache, rev, w0, xp)
    return y, g, H


def loss_and_grad(mode, params, x, normals, sdf, weights, alpha=100.0, w0=30.0, xp=np, s2_stats=None):
    """mode in {'s1','s2','siren'}.  Returns (terms dict, grads list, debug dict)."""
    y, cache = forward(params, x, w0, xp)
    g, rev = (None, None)
    if mode in ("s1", "siren"):
        g, rev = input_gradient(params, cache, w0, xp)
    H = tang = None
    if mode == "s1":
        if weights[2] != 0:
            H, tang = hessian(params, x, cache, rev, w0, xp)
        terms, cot = loss_s1_terms(y, g, H, normals, sdf, weights, alpha, xp)
    elif mode == "siren":
        terms, cot = loss_siren_terms(y, g, normals, sdf, weights, xp)
    elif mode == "s2":
        terms, cot = loss_s2_terms(y, sdf, weights, xp, s2_stats)
    else:
        raise ValueError(mode)
    if mode == "s1" and weights[2] != 0:
        grads, trace = param_grad_hessian(params, x, cache, rev, tang, cot["ybar"], cot.get("gbar"), cot["Hbar"], w0, xp,
                                          want_trace=True)
    else:
        grads, trace = param_grad(params, x, cache, rev, cot["ybar"], cot.get("gbar"), w0, xp)
    dbg = {"y": y, "g": g, "H": H, "tang": tang, "cache": cache, "rev": rev, "cot": cot, "trace": trace}
    return terms, grads, dbg


def adam_step(theta, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (reference train.py:334-337), one step, in place on numpy arrays.
    `step` is the 1-based step count AFTER this update."""
    m *= beta1; m += (1.0 - beta1) * grad
    v *= beta2; v += (1.0 - beta2) * grad * grad
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    theta -= (lr / bc1) * (m / denom)
    return theta


# --------------------------------------------------------------------------
# inverse hyperbolic map + MC field extraction pieces — reference src/inverses.py:18-19,
# src/render_mc.py:72-93
# --------------------------------------------------------------------------
def inv_tanh(pred_df, alpha):
    return np.where(pred_df < 1.0 / alpha, np.sqrt(pred_df / alpha), pred_df)


# --------------------------------------------------------------------------
# third derivatives and curvature of the eigenvector field  —  reference src/render_st.py:42-62
# (`compute_normals_and_cd`: n = top eigenvector of hessian(y,x); `compute_curvature`: jacobian(n, x)
#  [src/diff_operators.py:214-227], mean = trace/2, gaussian = -det [[J, n],[n^T, 0]])
# --------------------------------------------------------------------------
_MONO3 = [(i, j, k) for i in range(4) for j in range(4) for k in range(4) if i + j + k <= 3]


def _poly_mul(a, b):
    """Product of two trivariate polynomials truncated at total degree 3; coefficient arrays (..., 4, 4, 4)."""
    out = np.zeros_like(a)
    for (i, j, k) in _MONO3:
        for (p, q, r) in _MONO3:
            if i + p + j + q + k + r <= 3:
                out[..., i + p, j + q, k + r] += a[..., i, j, k] * b[..., p, q, r]
    return out


def third_derivatives(params, x, w0=30.0):
    """T (N,3,3,3) = d^3 f / dx_a dx_b dx_c by propagating the degree-3 Taylor polynomial of every activation in the
    three input variables (numpy only).  Composition with sin:  sin(a+u) = s (1 - u^2/2) + c (u - u^3/6) + O(u^4)."""
    x = np.asarray(x)
    N = x.shape[0]
    h = np.zeros((N, 3, 4, 4, 4), dtype=x.dtype)
    h[:, :, 0, 0, 0] = x
    h[:, 0, 1, 0, 0] = 1.0; h[:, 1, 0, 1, 0] = 1.0; h[:, 2, 0, 0, 1] = 1.0
    for l, (W, b) in enumerate(params[:-1]):
        z = np.einsum("of,nfijk->noijk", W, h)
        z[:, :, 0, 0, 0] += b
        a0 = _wl(w0, l) * z[:, :, 0, 0, 0]
        s, c = np.sin(a0)[..., None, None, None], np.cos(a0)[..., None, None, None]
        u = _wl(w0, l) * z
        u[:, :, 0, 0, 0] = 0.0
        u2 = _poly_mul(u, u)
        u3 = _poly_mul(u2, u)
        h = c * (u - u3 / 6.0) - s * (u2 / 2.0)
        h[:, :, 0, 0, 0] = s[..., 0, 0, 0]
    Wo, bo = params[-1]
    y = np.einsum("f,nfijk->nijk", Wo[0], h)
    T = np.zeros((N, 3, 3, 3), dtype=x.dtype)
    fact = [1.0, 1.0, 2.0, 6.0]
    for a in range(3):
        for b_ in range(3):
            for c_ in range(3):
                e = [0, 0, 0]
                e[a] += 1; e[b_] += 1; e[c_] += 1
                T[:, a, b_, c_] = y[:, e[0], e[1], e[2]] * fact[e[0]] * fact[e[1]] * fact[e[2]]
    return T


def shape_operator(H, T):
    """J (N,3,3) = d n_i / d x_k for n = eigenvector of the largest eigenvalue of H (lower triangle, as eigh reads it):
    dn/dx_k = sum_{j<2} v_j (v_j^T (dH/dx_k) n) / (lam_2 - lam_j).  Also returns (lam, V)."""
    Hl = np.tril(H) + np.transpose(np.tril(H, -1), (0, 2, 1))
    lam, V = np.linalg.eigh(Hl)
    n = V[:, :, 2]
    J = np.zeros_like(H)
    for j in range(2):
        vj = V[:, :, j]
        coef = np.einsum("na,nabk,nb->nk", vj, T, n) / (lam[:, 2] - lam[:, j])[:, None]
        J += vj[:, :, None] * coef[:, None, :]
    return J, lam, V


def curvatures(params, x, w0=30.0):
    """(normal n (N,3), principal directions (N,3,2), mean (N,), gaussian (N,), J (N,3,3)) — reference
    src/render_st.py:42-62.  The sign of n (and with it of J and of the mean curvature) is eigh's, i.e. arbitrary."""
    y, cache = forward(params, x, w0)
    g, rev = input_gradient(params, cache, w0)
    H, _ = hessian(params, x, cache, rev, w0)
    T = third_derivatives(params, x, w0)
    J, lam, V = shape_operator(H, T)
    n = V[:, :, 2]
    mean = 0.5 * np.trace(J, axis1=1, axis2=2)
    ext = np.zeros((len(x), 4, 4), dtype=J.dtype)
    ext[:, :3, :3] = J; ext[:, :3, 3] = n; ext[:, 3, :3] = n
    gauss = -np.linalg.det(ext)
    return n, V[:, :, :2], mean, gauss, J
