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
import torch

from . import hip_ops as _hip_ops

LOSS_S1, LOSS_S2, LOSS_SIREN = 0, 1, 2


class TrainEngine:
    def __init__(self, hidden, theta, w0=30.0, process_group=None, betas=(0.9, 0.999), eps=1e-8, ops=None):
        """`ops` defaults to the HIP kernels.  It is a parameter only so that the CPU/gloo tests can drive the
        distributed bookkeeping below with a stand-in compute backend; nothing in the product passes it."""
        self.ops = _hip_ops if ops is None else ops
        self.cfg = self.ops.make_cfg(hidden, w0)
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
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)

    def _allreduce(self, t):
        if self.world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def loss_and_grad(self, mode, x, normals, sdf, weights, alpha=100.0, n_global=None, n_hess=0):
        """Fills self.terms (global loss terms) and self.dtheta (global gradient); returns self.terms.
        n_hess: for loss_s1 with a Hessian weight, the number of leading on-surface points (C-ABI contract)."""
        ops = self.ops
        n = x.shape[0]
        n_global = n * self.world if n_global is None else n_global
        ws = ops.workspace_for(self.cfg, n, self.device, n_hess) if n_hess else ops.workspace_for(self.cfg, n, self.device)
        kw = {"n_hess": n_hess} if n_hess else {}
        if mode == LOSS_S2:
            stats = ops.s2_forward_stats(self.cfg, self.theta, x, sdf, ws)
            self._allreduce(stats)                       # (count, sum, sum sq) of the on-surface predictions
            self.terms.zero_()
            self.terms[:2] = ops.s2_terms(stats, weights)
            ops.loss_backward(self.cfg, mode, self.theta, x, normals, sdf, n_global, weights, alpha, self.ones,
                              stats, ws, dtheta=self.dtheta)
            self._allreduce(self.dtheta)
        else:
            terms = ops.loss_forward(self.cfg, mode, self.theta, x, normals, sdf, n_global, weights, alpha, ws, **kw)
            ops.loss_backward(self.cfg, mode, self.theta, x, normals, sdf, n_global, weights, alpha, self.ones,
                              None, ws, dtheta=self.dtheta, **kw)
            self.terms.copy_(terms)
            self._allreduce(self.flat)                   # one collective: gradient + the four loss scalars
        return self.terms

    def adam(self, lr):
        self.t += 1
        self.ops.adam_step(self.theta, self.dtheta, self.exp_avg, self.exp_avg_sq, self.t, lr, self.betas[0],
                           self.betas[1], self.eps)

    def step(self, mode, x, normals, sdf, weights, alpha=100.0, lr=1e-4, n_global=None, n_hess=0):
        terms = self.loss_and_grad(mode, x, normals, sdf, weights, alpha, n_global, n_hess)
        self.adam(lr)
        return terms
