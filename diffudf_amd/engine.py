# coding: utf-8
"""One training step of the reference loop (reference train.py:195-222) on flat device buffers:

    zero_grad -> loss dict -> sum -> backward -> [all-reduce] -> Adam

driven through the C ABI.  This is what `bench.py` times and what `train.py` runs; the
autograd-facing wrappers in `loss_functions.py` give the same numbers term by term.

Multi-GPU (one process per GPU, torch.distributed 'nccl' == RCCL over xGMI): the point batch is
sharded, every rank computes its share  sum_local(.)/n_global  of each term and of d(loss)/d(theta),
and ONE all-reduce(sum) of the flat [dtheta | 4 terms] buffer makes them global.  loss_s2 needs one
extra 3-double all-reduce of (count, sum, sum of squares) between its forward and backward
(SURVEY.md §8(e)).  Parameters stay replicated: every rank applies the identical Adam update.
"""
import inspect

import torch

from . import hip_ops as _hip_ops

LOSS_S1, LOSS_S2, LOSS_SIREN = 0, 1, 2


class TrainEngine:
    def __init__(self, hidden, theta, w0=30.0, process_group=None, betas=(0.9, 0.999), eps=1e-8, ops=None,
                 collectives=None, ww=None, wgrad_max_workgroups=None):
        """`ops` defaults to the HIP kernels.  It is a parameter only so that the CPU/gloo tests can drive the
        distributed bookkeeping below with a stand-in compute backend; nothing in the product passes it.
        `collectives` (N > 1 ranks): "staggered" = five all-reduces per step (three hidden-layer groups, each behind the
        weight-gradient GEMM of the next group, + the two thin layers), "fused" = ONE all-reduce of the flat
        [dtheta | terms] buffer after the whole backward; default "staggered" (bench.py --collectives).  Both give the same
        numbers; which is faster on xGMI is a latency question the first hardware run has to answer —
        `phase_times()` is there to read it off."""
        self.ops = _hip_ops if ops is None else ops
        self.cfg = self.ops.make_cfg(hidden, w0) if ww is None else self.ops.make_cfg(hidden, w0, ww=ww)
        n_theta = self.ops.theta_count(self.cfg)
        if theta.numel() != n_theta or theta.dtype != torch.float32:
            raise ValueError(f"theta must be a flat fp32 tensor of {n_theta} elements")
        if ops is None and theta.device.type != "cuda":
            raise ValueError("theta must live on the GPU: the HIP path has no CPU fallback")
        self.theta = theta
        self.device = theta.device
        # [dtheta | terms(4)] in one buffer so that one collective moves both
        self.flat = torch.zeros(n_theta + 4, dtype=torch.float32, device=self.device)
        self.dtheta = self.flat[:n_theta]
        self.terms = self.flat[n_theta:]
        self.exp_avg = torch.zeros(n_theta, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(n_theta, dtype=torch.float32, device=self.device)
        self.betas, self.eps = betas, eps
        self.t = 0
        self.ones = torch.ones(4, dtype=torch.float32, device=self.device)
        self.pg = process_group
        self.collectives = collectives or "staggered"
        if self.collectives not in ("staggered", "fused"):
            raise ValueError("collectives must be 'staggered' or 'fused'")
        self._out_kw = "out" in inspect.signature(self.ops.loss_forward).parameters
        self.profile = False                              # phase_times(): torch events around the phases of a step
        self._ev = {}
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        self._dist = self.world > 1
        # staggered collectives: leave 16 CUs to the RCCL kernels that overlap the weight-gradient GEMMs; otherwise the whole
        # chip.  The cap is a process-wide option of the library, so THIS engine sets it right in front of its own
        # weight-gradient launches (loss_and_grad) — another engine of the process may want another value.
        self.wgrad_max_workgroups = wgrad_max_workgroups if wgrad_max_workgroups is not None else (
            240 if (self._dist and self.collectives == "staggered") else 256)
        if not 8 <= int(self.wgrad_max_workgroups) <= 256:
            raise ValueError(f"wgrad_max_workgroups must be 8..256 (one workgroup per CU at most); got {self.wgrad_max_workgroups}")
        self._set_cap = getattr(self.ops, "set_wgrad_max_workgroups", None) if ops is None else None
        if self._dist:
            self._prime_collectives()

    def _prime_collectives(self):
        """Everything a step's collectives create lazily — the backend's communicator and streams, their hardware queues, pinned staging
        buffers — exists before the first step: one round of the step's own all-reduces over the (still zero) flat buffer, then a
        device-wide wait.  A queue created in the middle of step 0 makes the driver rebuild the GPU's run list under the running
        weight-gradient GEMM; here it happens while nothing of ours is in flight."""
        if self.collectives == "staggered" and hasattr(self.ops, "weight_gradient"):     # (the path loss_and_grad takes: `overlapped`)
            L = self.cfg.n_hidden_layers
            sl = self.ops.layer_slices(self.cfg)
            cuts = [(sl[b][0], sl[e - 1][1]) for b, e in self._layer_groups(L)] + [(sl[0][0], sl[0][1]), (sl[L][0], self.flat.numel())]
            works = [torch.distributed.all_reduce(self.flat[lo:hi], group=self.pg, async_op=True) for lo, hi in cuts]
            for w in works:
                w.wait()
        else:
            torch.distributed.all_reduce(self.flat, group=self.pg)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _allreduce(self, t):
        if self._dist:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    # ---- per-phase timing of a step (bench.py --gpus N: `phases_ms`) ---------------------------------------------------
    def _mark(self, name):
        """Bracket a phase on the compute stream: call before and after; phase_times() averages the intervals."""
        if self.profile and self.device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev.setdefault(name, []).append(e)

    def phase_times(self):
        """{phase: mean ms} over the steps run with `self.profile = True` (intervals between the paired marks; a collective's
        entry is how long the compute stream WAITED for it, not how long it took)."""
        torch.cuda.synchronize(self.device)
        out = {}
        for name, evs in self._ev.items():
            pairs = [(evs[i], evs[i + 1]) for i in range(0, len(evs) - 1, 2)]
            if pairs:
                out[name] = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
        self._ev = {}
        return out

    def loss_and_grad(self, mode, x, normals, sdf, weights, alpha=100.0, n_global=None, n_hess=0, _adam_lr=None):
        """Fills self.terms (global loss terms) and self.dtheta (global gradient); returns self.terms.
        n_hess: for loss_s1 with a Hessian weight, the number of leading on-surface points (C-ABI contract).
        (_adam_lr: `step()` passes its learning rate so that, with more than one rank, Adam runs per layer group as soon
        as that group's all-reduce has completed.)"""
        ops = self.ops
        n = x.shape[0]
        n_global = n * self.world if n_global is None else n_global
        ws = ops.workspace_for(self.cfg, n, self.device, n_hess) if n_hess else ops.workspace_for(self.cfg, n, self.device)
        kw = {"n_hess": n_hess} if n_hess else {}
        overlapped = self._dist and hasattr(ops, "weight_gradient") and self.collectives == "staggered"
        self._mark("forward+loss")
        stats = None
        if mode == LOSS_S2:
            stats = ops.s2_forward_stats(self.cfg, self.theta, x, sdf, ws)
            self._allreduce(stats)                       # (count, sum, sum sq) of the on-surface predictions
            self.terms.zero_()
            self.terms[:2] = ops.s2_terms(stats, weights)
        else:
            if self._out_kw:                             # straight into the tail of the flat [dtheta | terms] buffer
                ops.loss_forward(self.cfg, mode, self.theta, x, normals, sdf, n_global, weights, alpha, ws, out=self.terms, **kw)
            else:                                        # a backend without `out` (tests/test_distributed_gloo.py's stand-in)
                self.terms.copy_(ops.loss_forward(self.cfg, mode, self.theta, x, normals, sdf, n_global, weights, alpha, ws, **kw))
        self._mark("forward+loss")
        if self._set_cap is not None:
            self._set_cap(self.wgrad_max_workgroups)     # process-wide option: set by every engine before ITS launches
        if not overlapped:
            self._mark("backward")
            ops.loss_backward(self.cfg, mode, self.theta, x, normals, sdf, n_global, weights, alpha, self.ones,
                              stats, ws, dtheta=self.dtheta, **kw)
            self._mark("backward")
            # one collective: gradient + the four loss scalars (loss_s2's terms are global already)
            self._mark("allreduce_wait[fused]")
            self._allreduce(self.dtheta if mode == LOSS_S2 else self.flat)
            self._mark("allreduce_wait[fused]")
            if _adam_lr is not None:
                self._mark("adam")
                self._adam_slice(0, self.theta.numel(), _adam_lr)
                self._mark("adam")
            return self.terms
        # ---- N > 1: the all-reduce of a layer group overlaps the weight-gradient GEMM of the next one (RCCL runs on its
        # own stream behind the kernels already queued), and Adam for a group starts when its collective has completed
        self._mark("backward_sweeps")
        ops.loss_backward_sweeps(self.cfg, mode, self.theta, normals, sdf, n_global, weights, alpha, self.ones, stats, ws,
                                 n_local=n, **kw)
        self._mark("backward_sweeps")
        have_g = mode != LOSS_S2
        sl = ops.layer_slices(self.cfg)
        L = self.cfg.n_hidden_layers
        pending = []
        for b, e in self._layer_groups(L):
            self._mark(f"wgrad[{b}:{e}]")
            ops.weight_gradient(self.cfg, n, have_g, b, e, self.dtheta, ws, **kw)
            self._mark(f"wgrad[{b}:{e}]")
            lo, hi = sl[b][0], sl[e - 1][1]
            pending.append((torch.distributed.all_reduce(self.flat[lo:hi], group=self.pg, async_op=True), lo, hi))
        self._mark("wgrad[thin]")
        ops.weight_gradient(self.cfg, n, have_g, -1, 0, self.dtheta, ws, **kw)          # first and output layer together
        self._mark("wgrad[thin]")
        tail_hi = self.flat.numel() if mode != LOSS_S2 else self.theta.numel()           # + the 4 loss terms unless global already
        pending.append((torch.distributed.all_reduce(self.flat[sl[0][0]:sl[0][1]], group=self.pg, async_op=True), sl[0][0], sl[0][1]))
        pending.append((torch.distributed.all_reduce(self.flat[sl[L][0]:tail_hi], group=self.pg, async_op=True), sl[L][0], sl[L][1]))
        for i, (work, lo, hi) in enumerate(pending):
            self._mark(f"allreduce_wait[{i}]")
            work.wait()                                  # the compute stream waits for this collective only
            self._mark(f"allreduce_wait[{i}]")
            if _adam_lr is not None:
                self._mark(f"adam[{i}]")
                self._adam_slice(lo, hi, _adam_lr)
                self._mark(f"adam[{i}]")
        return self.terms

    @staticmethod
    def _layer_groups(L, target=3):
        """Hidden matrices 1 .. L-1 in `target` contiguous groups, last layers first (their gradients are ready first
        in the reference's backward; here all are ready, the order only staggers the collectives)."""
        idx = list(range(1, L))
        if not idx:
            return []
        k = min(target, len(idx))
        cuts = [round(i * len(idx) / k) for i in range(k + 1)]
        groups = [(idx[cuts[i]], idx[cuts[i + 1] - 1] + 1) for i in range(k) if cuts[i + 1] > cuts[i]]
        return groups[::-1]

    def _adam_slice(self, lo, hi, lr):
        self.ops.adam_step(self.theta[lo:hi], self.dtheta[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.t, lr,
                           self.betas[0], self.betas[1], self.eps)

    def adam(self, lr):
        self.t += 1
        self._adam_slice(0, self.theta.numel(), lr)

    def step(self, mode, x, normals, sdf, weights, alpha=100.0, lr=1e-4, n_global=None, n_hess=0):
        self.t += 1                                      # Adam's bias correction wants the number of THIS step ...
        try:
            return self.loss_and_grad(mode, x, normals, sdf, weights, alpha, n_global, n_hess, _adam_lr=lr)
        except Exception:
            self.t -= 1                                  # ... and a step that failed has not happened
            raise
