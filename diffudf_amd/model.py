# coding: utf-8
"""`SIREN` with the reference's constructor, state_dict keys and forward contract
(reference src/model.py:48-135), computing through the HIP kernels.

What is kept from the reference interface
  * `SIREN(n_in_features, n_out_features, hidden_layer_config=[], w0=30, ww=None,
           delay_init=False, activation='sine')`
  * `.net` is an `nn.Sequential` of `nn.Sequential(nn.Linear[, SineLayer])`, so `state_dict()` keys are
    `net.{i}.0.weight|bias` and every reference checkpoint loads (reference generate_mc.py:21, train.py:327)
  * initialisation distributions (src/model.py:7-19, :111-113)
  * `forward(x) -> {"model_in": x.clone().detach().requires_grad_(True), "model_out": y}` (dict order matters:
    reference src/evaluate.py:26 unpacks `.values()`)

What is different inside
  * all 18 parameters are views into ONE flat fp32 buffer in state_dict order — the `theta` the C ABI takes;
    `torch.optim.Adam(model.parameters())` updates it in place exactly as it updates the reference's tensors
  * ANY `hidden_layer_config` runs (reference src/model.py:94-108): the buffer has the layout of SIREN(3, 1, [Hp]*L) with
    Hp the smallest built width >= the widest layer, zero outside the caller's shapes (exact for a sine MLP, see
    hip_ops.padded_width); the parameters are strided views of it with the CALLER's shapes, so state_dict, optimizers and
    autograd never see a padded entry
  * `forward` runs the fused HIP value sweep; there is no PyTorch/CPU fallback: CPU inputs raise.
  * `ww != w0` (first SineLayer w0, the others ww): one extra float of the C ABI's network descriptor (round 4)
  * `n_in_features = 3 + k` (a latent vector in front of the coordinates, reference src/evaluate.py:19-22): such a network is
    QUERIED — `evaluate(model, samples, latent_vec, ...)` — as the 3-input network it is for that vector (`folded`); training
    it is not part of the reference's recipes (every `SIREN(...)` the reference constructs has 3 inputs) and raises.
"""
import math
import weakref

import torch
from torch import nn

from . import hip_ops
from ._lib import DudfError


class SineLayer(nn.Module):
    """sin(w0 * x) — reference src/model.py:22-33.  Kept for module structure / repr; the HIP sweep
    applies it in registers."""

    def __init__(self, w0=30):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        raise DudfError("SineLayer is evaluated inside the fused HIP sweep; call the SIREN module")

    def __repr__(self):
        return f"SineLayer(w0={self.w0})"


class _SirenValue(torch.autograd.Function):
    """y = SIREN(x) by the HIP forward sweep.  backward: cotangent on y -> parameters (and to x via df/dx)."""

    @staticmethod
    def forward(ctx, model, coords, *params):
        x2 = coords.reshape(-1, 3)
        theta = model.flat_parameters()
        f, _ = hip_ops.query(model.hip_cfg, theta, x2, want_grad=False)
        ctx.model = model
        ctx.save_for_backward(coords)
        return f.reshape(coords.shape[:-1] + (1,))

    @staticmethod
    def backward(ctx, grad_y):
        model = ctx.model
        (coords,) = ctx.saved_tensors
        x2 = coords.reshape(-1, 3).contiguous()
        theta = model.flat_parameters()
        ws = hip_ops.workspace_for(model.hip_cfg, x2.shape[0], x2.device)
        _, g = hip_ops.fields_forward(model.hip_cfg, theta, x2, ws)
        ybar = grad_y.reshape(-1).contiguous().float()
        dtheta = hip_ops.fields_backward(model.hip_cfg, theta, x2, ybar, None, ws)
        gx = (g * ybar[:, None]).reshape(coords.shape) if ctx.needs_input_grad[1] else None
        return (None, gx) + tuple(model.split_flat(dtheta))


class SIREN(nn.Module):
    def __init__(self, n_in_features, n_out_features, hidden_layer_config=[], w0=30, ww=None, delay_init=False,
                 activation='sine'):
        super().__init__()
        self.w0 = w0
        self.ww = w0 if ww is None else ww
        self.activation = activation
        self.n_in_features, self.n_out_features = n_in_features, n_out_features
        self.hidden_layer_config = list(hidden_layer_config)
        dims = [n_in_features] + self.hidden_layer_config + [n_out_features]
        blocks = []
        for i in range(len(dims) - 1):
            mods = [nn.Linear(dims[i], dims[i + 1])]
            if i < len(dims) - 2:
                mods.append(SineLayer(self.w0 if i == 0 else self.ww))
            blocks.append(nn.Sequential(*mods))
        self.net = nn.Sequential(*blocks)
        if not delay_init and activation == 'sine':
            with torch.no_grad():
                for i, blk in enumerate(self.net):
                    lin = blk[0]
                    fan_in = lin.weight.size(-1)
                    bound = (1.0 / fan_in) if i == 0 else (math.sqrt(6.0 / fan_in) / self.ww)
                    lin.weight.uniform_(-bound, bound)          # biases keep nn.Linear's default
        self._flat = None
        self._flatten()

    # ---- flat parameter storage -------------------------------------------------------------------------
    def _linears(self):
        return [blk[0] for blk in self.net]

    def _padded_dims(self):
        """[(rows, cols)] of every layer's weight in the C-ABI layout: SIREN(n_in, n_out, [Hp]*L) (fixed at construction)."""
        dims = self.__dict__.get("_dims_cache")
        if dims is None:
            dims = self.__dict__["_dims_cache"] = self._padded_dims_uncached()
        return dims

    def _padded_dims_uncached(self):
        L = len(self.hidden_layer_config)
        if L == 0:
            return [(self.n_out_features, self.n_in_features)]
        try:
            hp = hip_ops.padded_width(self.hidden_layer_config)
        except DudfError:                                 # no HIP path for this network (hip_cfg raises when it is used)
            hp = None
        dims = []
        for i, l in enumerate(self._linears()):
            o, k = l.weight.shape
            dims.append((o if (hp is None or i == L) else hp, k if (hp is None or i == 0) else hp))
        return dims

    def _views_of(self, flat):
        """Per-parameter views (the caller's shapes; strided where the layer is narrower than the padded width) of a
        buffer in the C-ABI layout, in `parameters()` order."""
        out, off = [], 0
        for l, (po, pk) in zip(self._linears(), self._padded_dims()):
            o, k = l.weight.shape
            out.append(flat[off:off + po * pk].view(po, pk)[:o, :k])
            off += po * pk
            out.append(flat[off:off + po][:o])
            off += po
        return out

    def _flat_numel(self):
        return sum(po * pk + po for po, pk in self._padded_dims())

    def _flatten(self):
        """(Re)build the flat buffer and point every parameter at its (strided) view of it."""
        lins = self._linears()
        with torch.no_grad():
            ref = lins[0].weight
            flat = torch.zeros(self._flat_numel(), dtype=ref.dtype, device=ref.device)
            params = [p for l in lins for p in (l.weight, l.bias)]
            views = self._views_of(flat)
            for p, v in zip(params, views):
                v.copy_(p.detach())
                p.data = v
        self._flat = flat
        # what "every parameter still aliases the flat buffer" means, as plain numbers: flat_parameters() runs several times per
        # training step and must not rebuild 18 views to find out
        self.__dict__["_flat_sig"] = [(v.data_ptr(), v.stride()) for v in views]

    def _is_flat(self):
        if self._flat is None:
            return False
        sig = self.__dict__.get("_flat_sig")
        if sig is None:
            return False
        params = [p for l in self._linears() for p in (l.weight, l.bias)]
        dt, dev = self._flat.dtype, self._flat.device
        for p, (ptr, stride) in zip(params, sig):
            if p.data_ptr() != ptr or p.stride() != stride or p.dtype != dt or p.device != dev:
                return False
        return True

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        if not self._is_flat():
            self._flatten()
        return out

    def flat_parameters(self):
        """theta for the C ABI: flat fp32 CUDA tensor aliasing every parameter (state_dict order)."""
        if not self._is_flat():
            self._flatten()
        return self._flat

    def split_flat(self, flat):
        """Views of a theta-shaped tensor, one per parameter, in `parameters()` order."""
        return self._views_of(flat)

    @property
    def hip_cfg(self):
        if self.activation != 'sine':
            raise DudfError("only activation='sine' has a HIP path (the reference's 'relu' variant is unused by its configs)")
        return hip_ops.make_cfg(self.hidden_layer_config, self.w0, self.n_in_features, self.n_out_features, ww=self.ww)

    def folded(self, latent_vec):
        """(cfg, theta) of the 3-input network this one becomes for ONE latent vector in front of the coordinates (reference
        src/evaluate.py:19-22 concatenates [latent | xyz]):  W_1 [latent | x] + b_1 = W_1[:, k:] x + (b_1 + W_1[:, :k] latent).
        theta is a fresh flat buffer in the C-ABI layout of SIREN(3, 1, hidden); everything behind the first layer is copied."""
        k = self.n_in_features - 3
        lat = torch.as_tensor(latent_vec).reshape(-1)
        if k < 1 or lat.numel() != k:
            raise DudfError(f"latent vector of {lat.numel()} entries for a network with {self.n_in_features} inputs (3 + {max(k, 0)})")
        flat = self.flat_parameters()
        po, pk = self._padded_dims()[0]
        w1 = flat[:po * pk].view(po, pk)
        b1 = flat[po * pk:po * pk + po]
        lat = lat.to(device=flat.device, dtype=flat.dtype)
        theta = torch.cat([w1[:, k:].reshape(-1), b1 + w1[:, :k] @ lat, flat[po * pk + po:]])
        return hip_ops.make_cfg(self.hidden_layer_config, self.w0, 3, self.n_out_features, ww=self.ww), theta

    # ---- forward --------------------------------------------------------------------------------------------
    def forward(self, x):
        """Same contract as reference src/model.py:116-135."""
        if x.shape[-1] != 3 or self.n_in_features != 3:
            raise DudfError(f"SIREN.forward: the HIP path takes 3-D points (got {x.shape[-1]} columns for n_in_features = "
                            f"{self.n_in_features}); a latent-conditioned network is queried with evaluate(model, samples, latent_vec)")
        coords_org = x.clone().detach().requires_grad_(True)
        if coords_org.device.type != "cuda":
            raise DudfError("SIREN.forward: inputs must be on the GPU; this build has no CPU/PyTorch fallback path")
        theta = self.flat_parameters()
        if theta.dtype != torch.float32:
            raise DudfError("HIP path is fp32; call model.float()")
        y = _SirenValue.apply(self, coords_org, *self.parameters())
        # lets diff_operators.gradient(y, x) find the network that produced y
        from .diff_operators import tag_field
        y = tag_field(y, "value", weakref.ref(self), coords_org)
        # ... and anything DERIVED from y (squeeze, reshape: Python attributes do not survive) through model_in
        coords_org._dudf_model = weakref.ref(self)          # (no self-reference: no cycle)
        return {"model_in": coords_org, "model_out": y}
