# coding: utf-8
"""`torch.optim.Adam` for a `diffudf_amd.model.SIREN` whose parameters and gradients live in flat buffers.

The reference constructs `torch.optim.Adam(lr=..., params=model.parameters())` (reference train.py:334-337, defaults
betas (0.9, 0.999), eps 1e-8, no weight decay) and only ever calls `step()` and edits `param_groups[*]['lr']`.  This class
IS that optimizer — same constructor, same `param_groups`, and `step()` falls back to torch's implementation whenever its
fast path does not apply — but when every parameter is a view of the model's flat theta and every `.grad` a view of the
flat gradient buffer (what `train.py::_zero_flat_grad` sets up), one `dudf_adam_step` launch updates everything: the kernel
replays torch's CUDA Adam operation by operation (csrc/dudf_misc.hip; tests/test_api_gpu.py holds a 10-step trajectory to
2e-7), instead of torch's six foreach launches over 18 tensors and their host-side bookkeeping (0.3 ms of Python a step,
which is what the stage-2 epochs of the reference recipe are bound by).  The fast path's moments are flat tensors of its
own; `state_dict()` / `load_state_dict()` convert them to and from torch.optim.Adam's per-parameter layout (`step`,
`exp_avg`, `exp_avg_sq`), so a checkpoint written by either implementation resumes in the other.
"""
import torch

from . import hip_ops


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, model=None, **kw):
        super().__init__(params, lr=lr, **kw)
        self._model = model
        self._m = self._v = None
        self._t = 0
        self._fell_back = False
        self._sched = self._row = self._sched_lrs = None
        self._sched_t0 = 0

    # ---- graph-replayable mode ------------------------------------------------------------------------------------------------
    def use_schedule(self, lrs):
        """From the next `step()` on, step k (k = 0, 1, ...) uses learning rate lrs[k]; the step count and the learning rate
        then live in DEVICE memory (`dudf_adam_step_scheduled`: a table of the two step-dependent scalars + a row cursor that
        `step()` advances on the stream), so a `step()` captured in a HIP graph replays correctly — the trainer calls
        `replayed()` once per replay to keep the host-side count (checkpoints) in step.  `param_groups[0]['lr']` must agree with
        the schedule whenever `step()` runs eagerly (checked), so a loop that edits it keeps working.  Bit-identical to the
        unscheduled path."""
        if self._model is None:
            raise RuntimeError("use_schedule needs the flat fast path (model=...)")
        dev = self._model.flat_parameters().device
        g = self.param_groups[0]
        self._sched_lrs = [float(v) for v in lrs]
        self._sched_t0 = self._t
        tab = hip_ops.adam_schedule(self._sched_lrs, self._t + 1, g["betas"][0], g["betas"][1])
        self._sched = torch.from_numpy(tab).to(dev)
        self._row = torch.zeros(1, dtype=torch.int64, device=dev)

    def replayed(self, n=1):
        """A captured graph containing `step()` was replayed n times.  The replay took its learning rate from the device table: the
        host-side view has to agree with it (a scheduler or a user editing `param_groups` mid-run would otherwise be ignored silently)."""
        for _ in range(n):
            k = self._t - self._sched_t0
            if self._sched_lrs is None or k >= len(self._sched_lrs):
                raise RuntimeError(f"diffudf_amd.optim.Adam: replay of step {k} past a schedule of {0 if self._sched_lrs is None else len(self._sched_lrs)}")
            if self.param_groups[0]["lr"] != self._sched_lrs[k]:
                raise RuntimeError(f"diffudf_amd.optim.Adam: param_groups lr {self.param_groups[0]['lr']} != scheduled {self._sched_lrs[k]} at replayed step {k} "
                                   "(under use_schedule the learning rates come from the table)")
            self._t += 1

    def _flat_views(self):
        """(theta, dtheta) if the fast path applies right now, else None."""
        m = self._model
        if m is None or self._fell_back or len(self.param_groups) != 1:
            return None
        g = self.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            return None
        flat_g, sig = getattr(m, "_dudf_flat_grad", None), getattr(m, "_dudf_flat_grad_sig", None)
        if flat_g is None or sig is None:
            return None
        theta = m.flat_parameters()
        params = g["params"]
        if theta.device.type != "cuda" or len(params) != len(sig) or flat_g.numel() < theta.numel():
            return None
        mine = list(m.parameters())
        if len(mine) != len(params) or any(a is not b for a, b in zip(mine, params)):
            return None
        for p, (ptr, stride) in zip(params, sig):
            if p.grad is None or p.grad.data_ptr() != ptr or p.grad.stride() != stride:
                return None
        return theta, flat_g[:theta.numel()]

    @torch.no_grad()
    def step(self, closure=None):
        fv = self._flat_views() if closure is None else None
        if fv is None:
            if self._t > 0 and not self._fell_back:
                raise RuntimeError("diffudf_amd.optim.Adam: the flat parameter / gradient layout changed after the first "
                                   "step; its moments cannot follow")
            self._fell_back = True
            return super().step(closure)
        theta, dtheta = fv
        if self._m is None:
            self._m, self._v = torch.zeros_like(theta), torch.zeros_like(theta)
        g = self.param_groups[0]
        self._t += 1
        if self._sched is not None:
            k = self._t - 1 - self._sched_t0
            if k >= len(self._sched_lrs):
                raise RuntimeError(f"diffudf_amd.optim.Adam: step {k} of a schedule of {len(self._sched_lrs)}")
            if g["lr"] != self._sched_lrs[k]:
                raise RuntimeError(f"diffudf_amd.optim.Adam: param_groups lr {g['lr']} != scheduled {self._sched_lrs[k]} at step {k}")
            hip_ops.adam_step_scheduled(theta, dtheta, self._m, self._v, self._sched, self._row, g["betas"][0], g["betas"][1], g["eps"])
            self._row += 1
            return None
        hip_ops.adam_step(theta, dtheta, self._m, self._v, self._t, g["lr"], g["betas"][0], g["betas"][1], g["eps"])
        return None

    # ---- checkpoints: torch.optim.Adam's per-parameter state <-> the flat moments ------------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        if self._m is not None and self._t > 0 and not self._fell_back and self._model is not None:
            ms, vs = self._model.split_flat(self._m), self._model.split_flat(self._v)
            sd["state"] = {i: {"step": torch.tensor(float(self._t)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
                           for i, (m, v) in enumerate(zip(ms, vs))}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = state_dict.get("state", {})
        if not st:
            return
        if self._model is None:
            return                                      # plain torch.optim.Adam behaviour
        if self._t > 0 and self._fell_back:
            return                                      # already on torch's own step(): its state was just loaded
        theta = self._model.flat_parameters()
        m, v = torch.zeros_like(theta), torch.zeros_like(theta)
        steps = set()
        params = list(self._model.parameters())
        if len(st) != len(params):
            raise ValueError(f"optimizer state for {len(st)} parameters, the model has {len(params)}")
        for (idx, s), mv, vv in zip(sorted(st.items()), self._model.split_flat(m), self._model.split_flat(v)):
            mv.copy_(s["exp_avg"].to(theta.device)); vv.copy_(s["exp_avg_sq"].to(theta.device))
            steps.add(int(float(s["step"])))
        if len(steps) != 1:
            raise ValueError("diffudf_amd.optim.Adam: per-parameter step counts differ; the flat update has one")
        self._m, self._v, self._t = m, v, steps.pop()
        self.state.clear()                              # the flat moments are the state from here on (the fast path owns it)
