# coding: utf-8
"""N>1 path on CPU: two gloo ranks drive diffudf_amd.engine.TrainEngine (the real sharding / all-reduce /
Adam bookkeeping) with an oracle-backed stand-in for the HIP kernels, and must reproduce the single-rank
step on the same global batch."""
import os
import socket
import numpy as np
import torch
import torch.multiprocessing as mp

from diffudf_amd import synth
from diffudf_amd.engine import TrainEngine, LOSS_S1, LOSS_S2
from oracle import dudf_oracle as O

HIDDEN = [32, 32]
N_GLOBAL = 96
W_S1 = [1e4, 1e4, 0.0, 1e3]
W_S2 = [1e5, 1e5]


class OracleOps:
    """Same call surface as diffudf_amd.hip_ops, computed by the oracle in fp64 on CPU tensors."""

    class Cfg:
        def __init__(self, hidden, w0):
            self.hidden_list, self.w0 = list(hidden), w0

    def make_cfg(self, hidden, w0=30.0):
        return OracleOps.Cfg(hidden, w0)

    def theta_count(self, cfg):
        return sum(o * i + o for o, i in synth.siren_layer_shapes(cfg.hidden_list))

    def workspace_for(self, cfg, n, device):
        return {}

    def _params(self, cfg, theta):
        return synth.unflatten_params(theta.double().numpy(), cfg.hidden_list)

    @staticmethod
    def _flat(grads):
        return np.concatenate([np.concatenate([w.reshape(-1), b.reshape(-1)]) for w, b in grads])

    def loss_forward(self, cfg, mode, theta, x, normals, sdf, n_global, weights, alpha, ws):
        P = self._params(cfg, theta)
        terms, grads, _ = O.loss_and_grad("s1", P, x.double().numpy(), normals.double().numpy(),
                                          sdf.double().numpy().reshape(-1, 1), weights, alpha)
        share = x.shape[0] / n_global
        ws["grads"] = self._flat(grads) * share
        return torch.tensor([float(v) * share for v in terms.values()], dtype=torch.float32)

    def loss_backward(self, cfg, mode, theta, x, normals, sdf, n_global, weights, alpha, cot, stats, ws, dtheta=None,
                      accumulate=False):
        if mode == LOSS_S2:
            P = self._params(cfg, theta)
            xs = x.double().numpy()
            y, cache = O.forward(P, xs)
            st = tuple(float(v) for v in stats)
            _, c = O.loss_s2_terms(y, sdf.double().numpy().reshape(-1, 1), weights, stats=st)
            grads, _ = O.param_grad(P, xs, cache, None, c["ybar"], None)
            g = self._flat(grads)
        else:
            g = ws["grads"]
        dtheta.copy_(torch.from_numpy(g).float())
        return dtheta

    def s2_forward_stats(self, cfg, theta, x, sdf, ws):
        y, _ = O.forward(self._params(cfg, theta), x.double().numpy())
        on = sdf.numpy().reshape(-1) == 0
        p = y[on]
        return torch.tensor([on.sum(), p.sum(), (p * p).sum()], dtype=torch.float64)

    def s2_terms(self, stats, weights):
        n, sm, sq = [float(v) for v in stats]
        mu = sm / n
        sd = np.sqrt((sq - n * mu * mu) / (n - 1))
        return torch.tensor([abs(mu) * weights[0], sd * weights[1]], dtype=torch.float32)

    def adam_step(self, theta, dtheta, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        th, mm, vv = theta.double().numpy(), m.double().numpy(), v.double().numpy()
        O.adam_step(th, dtheta.double().numpy() * grad_scale, mm, vv, step, lr, b1, b2, eps)
        theta.copy_(torch.from_numpy(th).float()); m.copy_(torch.from_numpy(mm).float()); v.copy_(torch.from_numpy(vv).float())


def _batch(idx):
    x, nrm, sdf = synth.training_batch(N_GLOBAL, seed=5)
    return torch.from_numpy(x[idx]), torch.from_numpy(nrm[idx]), torch.from_numpy(sdf[idx].reshape(-1))


def _run(engine, mode, weights, idx):
    x, nrm, sdf = _batch(idx)
    terms = engine.step(mode, x, nrm, sdf, weights, 100.0, lr=1e-4, n_global=N_GLOBAL).clone()
    return terms, engine.dtheta.clone(), engine.theta.clone()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    res = {}
    for name, mode, w in (("s1", LOSS_S1, W_S1), ("s2", LOSS_S2, W_S2)):
        theta = torch.from_numpy(synth.flatten_params(synth.siren_params(HIDDEN, seed=5)))
        eng = TrainEngine(HIDDEN, theta, ops=OracleOps())
        assert eng.world == world
        res[name] = [t.numpy() for t in _run(eng, mode, w, synth.stratified_shard(N_GLOBAL, rank, world))]
    out[rank] = res
    torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_step_equals_single_rank_step():
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    # single-rank reference on the whole batch (in stratified order: same point set)
    ref = {}
    for name, mode, w in (("s1", LOSS_S1, W_S1), ("s2", LOSS_S2, W_S2)):
        theta = torch.from_numpy(synth.flatten_params(synth.siren_params(HIDDEN, seed=5)))
        eng = TrainEngine(HIDDEN, theta, ops=OracleOps())
        ref[name] = [t.numpy() for t in _run(eng, mode, w, np.arange(N_GLOBAL))]
    for name in ("s1", "s2"):
        for r in range(world):
            terms, dth, th = out[r][name]
            assert np.allclose(terms, ref[name][0], rtol=2e-6, atol=0), (name, r, terms, ref[name][0])
            assert np.abs(dth - ref[name][1]).max() <= 2e-6 * np.abs(ref[name][1]).max(), (name, r)
            assert np.abs(th - ref[name][2]).max() <= 1e-6, (name, r)
        # replicas stay identical
        assert np.array_equal(out[0][name][2], out[1][name][2])


def test_stratified_shards_partition_the_batch():
    for n, world in ((29970, 8), (100, 3), (96, 2)):
        parts = [synth.stratified_shard(n, r, world) for r in range(world)]
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(n))
        _, _, sdf = synth.training_batch(n, seed=1)
        on = [(sdf[p, 0] == 0).sum() for p in parts]
        assert max(on) - min(on) <= 1
