# coding: utf-8
"""Input derivatives of the SIREN field (reference src/diff_operators.py:187-227), from the HIP sweeps
instead of repeated `torch.autograd.grad`.

`gradient(y, x)` keeps the reference call shape: `y`/`x` are the "model_out"/"model_in" entries of
`SIREN.forward(...)`.  The result is the analytic reverse sweep  a_{l-1} = W_l^T (w0 cos(w0 z_l) * a_l)
evaluated by the kernel (SWEEP_REV); with autograd enabled it is a node of the graph (its backward = the adjoint
sweeps + weight-gradient GEMM), so a loss written on it in plain PyTorch trains the network as with the reference's
`create_graph=True`.  The fused losses in `loss_functions.py` (one forward, one backward for all terms) remain the fast
path; `fields()` below returns value and gradient from one forward.  `hessian` / `laplace` / `divergence` are queries
(no graph): the Hessian enters training only through `loss_s1`'s fused term.
"""
import torch

from . import hip_ops
from ._lib import DudfError


# Operations that only re-shape / re-type / copy a tensor: the result is still the same field, point for point.
_SHAPE_ONLY = {"squeeze", "unsqueeze", "reshape", "view", "view_as", "reshape_as", "flatten", "unflatten", "contiguous",
               "detach", "clone", "float", "to", "type", "__getitem__", "select"}


class FieldTensor(torch.Tensor):
    """What `SIREN.forward` returns as 'model_out' and what `gradient` / `compute_normals_and_cd` return: an ordinary
    tensor that remembers WHICH field of the network it is (`_dudf_kind`: 'value', 'grad', 'eig_normal') and hands
    that on to anything computed from it — as 'view:<field>' through shape-only operations (squeeze, reshape, view,
    detach, clone ...: still the same numbers per point), as 'fn:<field>' through everything else (2 * y, tanh(y),
    y.abs(), a sum with another tensor ...).  The reference's operators take (y, x) pairs of an autograd graph; here
    the pair has to name a field the kernels can evaluate, and a FUNCTION of one must fail loudly instead of being
    mistaken for it (round-2 advice: `gradient(2 * y, x)` silently returned grad f)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        out = super().__torch_function__(func, types, args, kwargs or {})
        kind = None
        for a in args:
            k = getattr(a, "_dudf_kind", None) if isinstance(a, torch.Tensor) else None
            if k is not None:
                base = k.split(":", 1)[-1]
                shape_only = getattr(func, "__name__", "") in _SHAPE_ONLY and not k.startswith("fn:")
                kind = ("view:" if shape_only else "fn:") + base
                break
        if kind is not None:
            for o in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(o, FieldTensor) and getattr(o, "_dudf_kind", None) is None:
                    o._dudf_kind = kind
        return out


def tag_field(t, kind, model_ref=None, coords=None):
    t = t.as_subclass(FieldTensor)
    t._dudf_kind = kind
    if model_ref is not None:
        t._dudf_src = (model_ref, coords)
    return t


def _source(y, x):
    """(model, coords) behind a reference-style (y, x) pair.  `y` carries the tag when it is exactly what `forward`
    returned; any tensor DERIVED from it (the reference's own `pred_sdf.squeeze(-1)`, src/loss_functions.py:141, a
    reshape, a slice of all points) has lost Python attributes, so the lookup falls back to `x`: `forward` registers
    the 'model_in' tensor it hands out.  What stays unsupported: an `x` that is not the 'model_in' object itself."""
    src = getattr(y, "_dudf_src", None)
    if src is None and getattr(x, "_dudf_model", None) is not None:
        src = (x._dudf_model, x)
    if src is None:
        raise DudfError("gradient/hessian: neither `y` nor `x` comes from diffudf_amd.model.SIREN.forward "
                        "(generic autograd graphs have no HIP path here)")
    model, coords = src[0](), src[1]
    if model is None:
        raise DudfError("the SIREN that produced `y` no longer exists")
    if x is not coords:
        raise DudfError("gradient(y, x): `x` must be the 'model_in' tensor returned together with `y`")
    return model, coords


def _kind(y, coords):
    """What field of the network `y` is.  Tagged results keep their tag; a tensor obtained from the model output through
    shape-only operations that still has one value per point (squeeze / reshape / view, e.g. reference
    src/loss_functions.py:141) IS the model output; a function of a field (2 * y, tanh(y), a slice of a gradient) has
    no HIP path and raises."""
    k = getattr(y, "_dudf_kind", None)
    n = coords.numel() // 3
    if k in ("value", "grad", "eig_normal"):
        return k
    if k == "view:value" and y.numel() == n:
        return "value"
    if k is None:
        raise DudfError("this tensor does not come from diffudf_amd.model.SIREN.forward (generic autograd graphs have no "
                        "HIP path here)")
    raise DudfError(f"no HIP path for a tensor derived from the '{k.split(':', 1)[-1]}' field (slices of a gradient: use "
                    "hessian(y, x); arbitrary functions of the output: use fields())")


class _InputGradient(torch.autograd.Function):
    """df/dx as a node of the autograd graph: the reference builds it with `create_graph=True` so that a loss written on
    it trains the network (src/diff_operators.py:208-212, used by src/loss_functions.py:31-32, :89-101).  Forward: the fused
    value + reverse sweeps with the training stash kept; backward: the two adjoint sweeps + weight-gradient GEMM with the
    incoming cotangent of df/dx (`dudf_fields_backward`).  The stash lives in the network's workspace, so — like the
    reference's graph — call backward() before the next forward of a different batch on the same network."""

    @staticmethod
    def forward(ctx, model, coords, *params):
        x2 = coords.detach().reshape(-1, 3).contiguous().float()
        ws = hip_ops.workspace_for(model.hip_cfg, x2.shape[0], x2.device)
        _, g = hip_ops.fields_forward(model.hip_cfg, model.flat_parameters(), x2, ws)
        ctx.model, ctx.x2, ctx.shape = model, x2, coords.shape
        return g.reshape(coords.shape)

    @staticmethod
    def backward(ctx, gbar):
        model, x2 = ctx.model, ctx.x2
        theta = model.flat_parameters()
        ws = hip_ops.workspace_for(model.hip_cfg, x2.shape[0], x2.device)
        hip_ops.fields_forward(model.hip_cfg, theta, x2, ws)            # the stash may belong to another node by now
        zeros = torch.zeros(x2.shape[0], dtype=torch.float32, device=x2.device)
        dtheta = hip_ops.fields_backward(model.hip_cfg, theta, x2, zeros, gbar.reshape(-1, 3).contiguous().float(), ws)
        return (None, None) + tuple(model.split_flat(dtheta))


def gradient(y, x, grad_outputs=None):
    """dy/dx, shaped like x — reference src/diff_operators.py:208-212.  With autograd enabled the result is part of the
    graph (differentiable with respect to the network's parameters, like the reference's `create_graph=True`); inside
    `torch.no_grad()` it is a plain query."""
    model, coords = _source(y, x)
    if _kind(y, coords) != "value":
        raise DudfError("gradient(y, x): `y` must be the model output; for second derivatives use hessian / laplace")
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        g = _InputGradient.apply(model, coords, *model.parameters())
    else:
        x2 = coords.detach().reshape(-1, 3)
        _, g = hip_ops.query(model.hip_cfg, model.flat_parameters(), x2, want_grad=True)
        g = g.reshape(coords.shape)
    if grad_outputs is not None:
        g = g * grad_outputs.reshape(coords.shape[:-1] + (1,))
        return tag_field(g, "fn:grad")
    return tag_field(g, "grad")                        # divergence(gradient(y, x), x) finds its way (laplace)


def hessian(y, x):
    """(1,N,3,3) like reference src/diff_operators.py:187-193: row i = grad(df/dx_i, x).  Forward-over-reverse in
    the kernels (three tangent columns per point), not repeated autograd.  Plain tensor, like `gradient`."""
    model, coords = _source(y, x)
    if _kind(y, coords) != "value":
        raise DudfError("hessian(y, x): `y` must be the model output")
    x2 = coords.detach().reshape(-1, 3)
    _, _, h = hip_ops.query_hessian(model.hip_cfg, model.flat_parameters(), x2)
    return h.reshape(coords.shape[:-1] + (3, 3)) if coords.dim() == 3 else h.unsqueeze(0)


def divergence(y, x):
    """sum_i d y_i / d x_i, (1,N,1) — reference src/diff_operators.py:201-205, for the two vector fields this path
    produces: the gradient of the model output (trace of the Hessian) and the eigen-normal field of
    `render_st.compute_normals_and_cd` (trace of the shape operator the curvature kernel returns)."""
    model, coords = _source(y, x)
    kind = _kind(y, coords)
    x2 = coords.detach().reshape(-1, 3)
    if kind == "grad":
        _, _, h = hip_ops.query_hessian(model.hip_cfg, model.flat_parameters(), x2)
    elif kind == "eig_normal":
        _, _, _, _, h = hip_ops.query_curvature(model.hip_cfg, model.flat_parameters(), x2, want_shape=True)
    else:
        raise DudfError("divergence: only gradient(y, x) and compute_normals_and_cd's normals have a HIP path")
    return (h[:, 0, 0] + h[:, 1, 1] + h[:, 2, 2]).reshape(coords.shape[:-1] + (1,))


def laplace(y, x):
    """div(grad f) = trace of the Hessian — reference src/diff_operators.py:196-198."""
    h = hessian(y, x)
    return (h[..., 0, 0] + h[..., 1, 1] + h[..., 2, 2]).unsqueeze(-1)


def jacobian(y, x):
    """(jac (1,N,C,3), status) like reference src/diff_operators.py:214-227, for the two fields this path produces:
    the model output (C = 1: the gradient) and the eigen-normal field returned by `render_st.compute_normals_and_cd`
    (C = 3: the shape operator d n_i / d x_k, third derivatives of f, from `dudf_query_curvature`).  status = -1 if
    the result holds a NaN, as in the reference."""
    import torch
    model, coords = _source(y, x)
    kind = _kind(y, coords)
    x2 = coords.detach().reshape(-1, 3)
    if kind == "value":
        _, g = hip_ops.query(model.hip_cfg, model.flat_parameters(), x2, want_grad=True)
        jac = g.reshape(1, -1, 1, 3)
    elif kind == "eig_normal":
        _, _, _, _, shape = hip_ops.query_curvature(model.hip_cfg, model.flat_parameters(), x2, want_shape=True)
        jac = shape.reshape(1, -1, 3, 3)
    else:
        raise DudfError("jacobian: only the model output and compute_normals_and_cd's normals have a HIP path")
    return jac, (-1 if bool(torch.isnan(jac).any()) else 0)


class _Fields(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        x2 = x.detach().reshape(-1, 3).contiguous().float()
        theta = model.flat_parameters()
        ws = hip_ops.workspace_for(model.hip_cfg, x2.shape[0], x2.device)
        f, g = hip_ops.fields_forward(model.hip_cfg, theta, x2, ws)
        ctx.model, ctx.ws, ctx.x2 = model, ws, x2
        ctx.stamp = _stamp(ws)
        return f, g

    @staticmethod
    def backward(ctx, fbar, gbar):
        model, ws, x2 = ctx.model, ctx.ws, ctx.x2
        if _stamp(ws, peek=True) != ctx.stamp:
            raise DudfError("fields(): the workspace was reused by another forward before backward()")
        theta = model.flat_parameters()
        fb = None if fbar is None else fbar.contiguous().float()
        gb = None if gbar is None else gbar.contiguous().float()
        if fb is None:
            fb = torch.zeros(x2.shape[0], device=x2.device)
        dtheta = hip_ops.fields_backward(model.hip_cfg, theta, x2, fb, gb, ws)
        return (None, None) + tuple(model.split_flat(dtheta))


def _stamp(ws, peek=False):
    if not peek:
        ws.generation = getattr(ws, "generation", 0) + 1
    return getattr(ws, "generation", 0)


def fields(model, x):
    """(f (n,), df/dx (n,3)) for points x (..., 3), differentiable with respect to the model parameters:
    lets a loss written in plain PyTorch on top of the two fields train through the HIP adjoint sweeps."""
    return _Fields.apply(model, x, *model.parameters())
