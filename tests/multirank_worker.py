# coding: utf-8
"""Child process of tests/test_multirank_gpu.py (not a test file): ONE rank of a world that shares cuda:0 over gloo
(RCCL refuses two ranks on one device; this launcher owns that choice — train.py and bench.py know nothing of it).  Runs the real HIP TrainEngine on this rank's stratified
shard of one global batch and writes what rank 0 ends up with.

    python tests/multirank_worker.py engine <case> <out.npz>        RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the env
    python tests/multirank_worker.py train  <cfg.json> <unused>     train.py's own loop (distributed when WORLD_SIZE > 1)
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HIDDEN = [256] * 8
N_GLOBAL = int(os.environ.get("DUDF_TEST_NGLOBAL", "6000"))       # (test plumbing: the worker's own variables)
STEPS = 3
CASES = {"s1eik": (0, [1e4, 1e4, 0.0, 1e3], 1e-4), "s1full": (0, [1e4, 1e4, 1e4, 1e3], 1e-4), "s2": (1, [1e5, 1e5], 1e-6),
         # the reference's schedule, shortened: Eikonal steps, then stage 2 at a lower rate with the same Adam state (train.py:179-191)
         "sched": None}
SCHED = [(0, [1e4, 1e4, 0.0, 1e3], 1e-4), (0, [1e4, 1e4, 0.0, 1e3], 1e-4), (1, [1e5, 1e5], 1e-5), (1, [1e5, 1e5], 1e-5)]


def run_engine(case, out):
    from diffudf_amd import synth
    from diffudf_amd.engine import TrainEngine
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    rccl1 = os.environ.get("DUDF_TEST_BACKEND") == "nccl1"       # (test plumbing) ONE rank over the real RCCL backend, N > 1 code path forced
    if rccl1:
        assert world == 1
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    elif world > 1:
        torch.distributed.init_process_group("gloo")
    plan = SCHED if case == "sched" else [CASES[case]] * STEPS
    for item in filter(None, os.environ.get("DUDF_TEST_OPTS", "").split(",")):      # (test plumbing) library options: name=value,...
        from diffudf_amd import hip_ops
        k, v = item.split("=", 1)
        hip_ops.set_option(k, int(v))
    dev = torch.device("cuda", 0)
    theta = torch.from_numpy(synth.flatten_params(synth.siren_params(HIDDEN, seed=123))).to(dev)
    eng = TrainEngine(HIDDEN, theta, collectives=os.environ.get("DUDF_TEST_COLLECTIVES") or None)   # (test plumbing: the worker's own variable)
    if rccl1:
        # a world of ONE rank takes the engine's N > 1 code path: this test flips the engine's own switch and sets what its constructor
        # would have set for a larger world (the product has no such option)
        eng._dist = True
        if eng.collectives == "staggered":
            eng.wgrad_max_workgroups = 240
        eng._prime_collectives()
    assert eng.world == world and eng._dist == (world > 1 or rccl1)
    hist, first_grad = [], None
    # (test plumbing) what went INTO and came OUT of every collective, per rank and step.  Always: two-number digests (sum, sum of
    # magnitudes, in double, computed on the device right in front of the collective — no sync), written to <out>.rank<r>.json so that a
    # failing test can name (rank, step, quantity).  With DUDF_TEST_TRACE=<dir> also full copies (tools/multirank_loop.py).
    trace_dir = os.environ.get("DUDF_TEST_TRACE")
    trace, digest = {}, {}

    def dg(tn):
        tn = tn.double()
        return torch.stack([tn.sum(), tn.abs().sum()])
    real_all_reduce = torch.distributed.all_reduce
    base = eng.flat.data_ptr()
    cur = {"t": -1, "k": 0}

    def traced_all_reduce(tensor, *args, **kw):
        if tensor.dtype == torch.float32 and base <= tensor.data_ptr() < base + eng.flat.numel() * 4:
            lo = (tensor.data_ptr() - base) // 4
            key = f"s{cur['t']}_pre{cur['k']}_{lo}_{lo + tensor.numel()}"
            if lo + tensor.numel() == eng.flat.numel():            # the tail slice carries this rank's share of the four loss terms
                digest[f"s{cur['t']}_terms_local"] = tensor[-4:].double().clone()
        else:
            key = f"s{cur['t']}_pre{cur['k']}_stats"
        digest[key] = dg(tensor)
        if trace_dir:
            trace[key] = tensor.clone()
        cur["k"] += 1
        return real_all_reduce(tensor, *args, **kw)
    if world > 1 or rccl1:
        torch.distributed.all_reduce = traced_all_reduce
    for t, (mode, w, lr) in enumerate(plan):
        cur["t"], cur["k"] = t, 0
        digest[f"s{t}_theta"] = dg(theta)
        if trace_dir:
            trace[f"s{t}_theta"] = theta.clone()
        idx = synth.stratified_shard(N_GLOBAL, rank, world)
        parts = np.split(idx, np.flatnonzero(np.diff(idx) != 1) + 1)
        b = [synth.training_batch(N_GLOBAL, seed=5, step=t, lo=int(p[0]), hi=int(p[-1]) + 1) for p in parts]
        x, nrm, sdf = [torch.from_numpy(np.concatenate([q[k] for q in b])).to(dev) for k in range(3)]
        sdf = sdf.reshape(-1)
        n_hess = int((sdf == 0).sum()) if case == "s1full" else 0
        if t == 0:                       # gradient first, then Adam on the whole buffer
            terms = eng.loss_and_grad(mode, x, nrm, sdf, w, 100.0, n_global=N_GLOBAL, n_hess=n_hess)
            first_grad = eng.dtheta.cpu().numpy().copy()
            eng.adam(lr)
        else:                            # what bench.py and train loops call: with N > 1 ranks, per-layer-group all-reduce
            terms = eng.step(mode, x, nrm, sdf, w, 100.0, lr=lr, n_global=N_GLOBAL, n_hess=n_hess)   # overlapped with the GEMM, Adam per group
        hist.append(terms.cpu().numpy().copy())
        digest[f"s{t}_post"] = dg(eng.flat)
        if trace_dir:
            trace[f"s{t}_post"] = eng.flat.clone()
        if t == 0:
            assert x.shape[0] == len(idx) and abs(len(idx) * world - N_GLOBAL) <= 3 * world      # stratified: equal shares of each third
    torch.cuda.synchronize()
    digest["theta_end"] = dg(theta)
    with open(f"{out}.rank{rank}.json", "w") as f:
        json.dump({k: [float(x) for x in v.cpu().tolist()] for k, v in digest.items()}, f)
    if trace_dir:
        trace["theta_end"] = theta.clone()
        np.savez(os.path.join(trace_dir, f"trace_w{world}_r{rank}.npz"), **{k: v.cpu().numpy() for k, v in trace.items()})
    if rank == 0:
        np.savez(out, hist=np.array(hist), dtheta0=first_grad, theta=theta.cpu().numpy())
    if world > 1 or rccl1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def run_train(cfg_path):
    import train
    cfg = json.load(open(cfg_path))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:               # this launcher's choice for a one-GPU box: every rank on cuda:0, gloo transport;
        torch.cuda.set_device(0)                                # train.py adopts a process group that is already initialised
        torch.distributed.init_process_group("gloo")
    train.setup_train(cfg, 0)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[1] == "engine":
        run_engine(sys.argv[2], sys.argv[3])
    else:
        run_train(sys.argv[2])
