#!/usr/bin/env python
# coding: utf-8
"""bench.py — train points/sec of the DiffUDF hot path on MI355X (BASELINE.json metric).

One "step" = one optimizer step of the reference loop (reference train.py:195-222) on one batch of synthetic points
already resident in HBM:
    SIREN 8x256 forward + df/dx + hyperbolic-scaled Eikonal/UDF loss (loss_s1, Hessian weight 0)
    + backward to theta + (all-reduce) + Adam.
Workload at N GPUs: 100 000 points PER GPU (weak scaling), uniform in [-1,1]^3, thirds [on-surface | far | near] like
the reference sampler.  Arithmetic is fp32 throughout; the hidden-layer matmuls run on the bf16 matrix cores with both
fp32 operands split exactly into three bf16 pieces (six products, fp32 accumulate: "bf16x6", fp32-equivalent).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--hidden H] [--no-cpu-baseline] [--no-config3]

Prints ONE JSON line on rank 0.  Timing protocol: W untimed warm-up steps, then EXACTLY K steps between two
barrier + synchronize pairs with the library's HIP-event profiler OFF; the per-kernel durations behind `roofline` come
from a SEPARATE, untimed pass of a few steps with the profiler on (events on the launch stream).
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

W_EIKONAL = [1e4, 1e4, 0.0, 1e3]       # loss_s1 weights with the Hessian term off = the headline metric
ALPHA = 100.0
PEAK_F32_MFMA_TFLOPS = 157.3           # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_TB_S = 8.0                     # HBM3E, same table
PEAK_BF16_MFMA_TFLOPS = 2500.0         # same table: BF16 MFMA, dense
PROFILE_STEPS = 8                       # untimed per-kernel pass


def f0(hidden, layers):
    """matmul flops of one forward channel per point: 2*(3H + (L-1)H^2 + H)."""
    return 2 * (3 * hidden + (layers - 1) * hidden * hidden + hidden)


def shard_batch(n_per_gpu, world, rank, seed):
    from diffudf_amd import synth
    n_global = n_per_gpu * world
    idx = synth.stratified_shard(n_global, rank, world)
    cuts = np.flatnonzero(np.diff(idx) != 1) + 1            # three contiguous windows of the global batch (one per stratum)
    xs, ns, ss = [], [], []
    for part in np.split(idx, cuts):
        x, nrm, sdf = synth.training_batch(n_global, seed=seed, lo=int(part[0]), hi=int(part[-1]) + 1)
        xs.append(x); ns.append(nrm); ss.append(sdf)
    return np.concatenate(xs), np.concatenate(ns), np.concatenate(ss), n_global


def cpu_baseline(hidden, layers, seed, sizes=(100000, 29970), budget_s=9.0):
    """The oracle (analytic restatement of the reference path, torch CPU backend, fp32) timed on this host at the batch
    sizes BASELINE.md §3 names: the bench workload (100 000 points) and the reference's own batch (29 970).  The thread
    count is calibrated once on a 20 000-point slice (torch's intra-op pool degrades badly when oversubscribed)."""
    from diffudf_amd import synth
    from oracle import dudf_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    P = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in synth.siren_params([hidden] * layers, seed=seed)]
    theta0 = torch.cat([torch.cat([w.reshape(-1), b.reshape(-1)]) for w, b in P]).numpy().copy()

    def make(n):
        x, nrm, sdf = [torch.from_numpy(a) for a in synth.training_batch(n, seed=seed)]
        theta = theta0.copy(); m = np.zeros_like(theta); v = np.zeros_like(theta)

        def one(t):
            terms, grads, _ = O.loss_and_grad("s1", P, x, nrm, sdf, W_EIKONAL, ALPHA, xp=torch)
            g = torch.cat([torch.cat([w.reshape(-1), b.reshape(-1)]) for w, b in grads]).numpy()
            O.adam_step(theta, g, m, v, t, 1e-4)
            return float(sum(terms.values()))
        return one

    out = {}
    with torch.no_grad():
        cal = make(20000)
        best = None
        for thr in sorted({c for c in (8, 16, 32, 64, avail) if c <= avail}):
            torch.set_num_threads(thr)
            cal(1)
            t0 = time.perf_counter(); cal(1); dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, thr)
        cores = best[1]
        torch.set_num_threads(cores)
        for n in sizes:
            one = make(n)
            one(1)                                     # warm-up
            t0 = time.perf_counter(); steps = 0
            while True:
                one(steps + 2); steps += 1
                el = time.perf_counter() - t0
                if (steps >= 2 and el > budget_s) or steps >= 20:
                    break
            out[n] = {"value": n * steps / el, "steps": steps, "seconds": round(el, 2)}
    head = out[sizes[0]]
    return {"value": head["value"], "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"{head['steps']} full steps (fwd + df/dx + loss + bwd + Adam) of the bench workload itself, "
                      f"{sizes[0]} points (oracle/dudf_oracle.py on torch {torch.__version__} CPU, fp32, {cores} threads = "
                      f"fastest of 8/16/32/64/{avail} on this host, {avail} logical cores visible)",
            "n29970": {"value": out[29970]["value"], "unit": "points/s", "steps": out[29970]["steps"],
                       "note": "the reference's own batch size (configs/train_cfg.json: 30000 x [0.333, 0.666])"}
            if 29970 in out else None}


class Runner:
    def __init__(self, args, world, rank, dev):
        from diffudf_amd import _lib
        self.args, self.world, self.rank, self.dev = args, world, rank, dev
        self.lib = _lib.load()

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run(self, hidden, layers, points, steps, warmup, loss, profile_steps=PROFILE_STEPS, seed=123):
        """(elapsed seconds for `steps` steps [max over ranks], per-kernel {name: avg ms} from an untimed pass,
        final loss, n_global, n_hess)."""
        from diffudf_amd import hip_ops, synth
        from diffudf_amd.engine import TrainEngine
        hid = [hidden] * layers
        theta = torch.from_numpy(synth.flatten_params(synth.siren_params(hid, seed=seed))).to(self.dev)
        x, nrm, sdf, n_global = shard_batch(points, self.world, self.rank, seed)
        x, nrm, sdf = [torch.from_numpy(a).to(self.dev) for a in (x, nrm, sdf.reshape(-1))]
        eng = TrainEngine(hid, theta)
        weights = W_EIKONAL if loss == "eikonal" else [1e4, 1e4, 1e4, 1e3]
        n_hess = 0
        if loss == "full":                              # shards come out as [on | far | near]: on-surface first
            n_hess = int((sdf == 0).sum())
            assert bool((sdf[:n_hess] == 0).all())
        step = lambda: eng.step(hip_ops.LOSS_S1, x, nrm, sdf, weights, ALPHA, lr=1e-4, n_global=n_global, n_hess=n_hess)  # noqa: E731
        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            terms = step()
        self.barrier()
        el = time.perf_counter() - t0
        final_loss = float(terms.sum())
        if self.world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=self.dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t)
        kern = {}
        if profile_steps:
            self.lib.dudf_profile_enable(1)
            for _ in range(profile_steps):
                step()
            torch.cuda.synchronize()
            self.lib.dudf_profile_enable(0)
            buf = ctypes.create_string_buffer(4096)
            self.lib.dudf_profile_dump(buf, len(buf))
            for line in buf.value.decode().splitlines():
                name, cnt, tot = line.split()
                kern[name] = float(tot) / int(cnt)
        del eng
        return el, kern, final_loss, n_global, n_hess


def roofline_block(args, kern, hidden, layers, points, n_hess, ms_step):
    F0 = f0(hidden, layers)
    hid = 2 * (layers - 1) * hidden * hidden                    # hidden x hidden matmul flops per point
    alg = {"sweep_fwd": F0, "sweep_rev": F0, "sweep_adj_fwd": F0, "sweep_adj_rev": F0,
           "wgrad_hidden": 2 * hid, "wgrad_small": 2 * (F0 - hid)}
    # which matrix-core instruction a kernel runs on: f32-input MFMA (1 MFMA flop per algorithmic flop, 157.3 TF) or the
    # 3-way bf16 split at fp32 accuracy (6 bf16 MFMA flops per algorithmic flop, 2.5 PF dense)
    bf16x6 = {"wgrad_hidden"} if os.environ.get("DUDF_WGRAD", "bf16")[0] != "f" else set()
    if hidden in (128, 256, 512) and layers >= 2 and os.environ.get("DUDF_SWEEP", "bf16")[0] != "f":
        from diffudf_amd import hip_ops
        if hip_ops.sweeps_on_bf16(hidden, layers):
            bf16x6 |= {"sweep_fwd", "sweep_rev", "sweep_adj_fwd", "sweep_adj_rev"}
    n_cols = points + 3 * n_hess                                 # columns the MFMA kernels process: 4 per Hessian-path point
    per = {}
    for k, fl in alg.items():
        if k in kern:
            tf = fl * n_cols / (kern[k] * 1e-3) / 1e12
            mult, peak = (6, PEAK_BF16_MFMA_TFLOPS) if k in bf16x6 else (1, PEAK_F32_MFMA_TFLOPS)
            per[k] = {"avg_ms": round(kern[k], 4), "algorithmic_tflops": round(tf, 2), "mfma": "bf16x6" if k in bf16x6 else "f32",
                      "executed_tflops": round(tf * mult, 2), "peak": peak, "frac": round(tf * mult / peak, 4)}
    if not per:
        return None
    dom = max(per, key=lambda k: per[k]["avg_ms"])
    mfma_ms = sum(kern[k] for k in alg if k in kern)
    # whole step against the same ceiling: algorithmic flops x the bf16 products executed per flop / wall time of a step
    mult_step = 6 if "sweep_fwd" in bf16x6 else 1
    peak_step = PEAK_BF16_MFMA_TFLOPS if mult_step == 6 else PEAK_F32_MFMA_TFLOPS
    step_tf = 6 * F0 * n_cols / (ms_step * 1e-3) / 1e12
    traffic, step_hbm, source = None, None, None
    prof_json = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if os.path.exists(prof_json) and points == 100000 and args.loss == "eikonal" and hidden == 256 and layers == 8:
        try:                                        # PMC bytes were collected on exactly this workload, see `source`
            tr = json.load(open(prof_json))
            meta = tr.get("_meta", {})
            traffic = tr.get(dom, {}).get("hbm_bytes_per_launch")
            tot = 0.0
            for k, d in per.items():
                b = tr.get(k, {}).get("hbm_bytes_per_launch")
                if b:
                    d["hbm_bytes_per_launch"] = b
                    d["hbm_tb_s"] = round(b / (d["avg_ms"] * 1e-3) / 1e12, 2)
                    tot += b
            if tot:
                step_hbm = {"bytes_per_step": tot, "bytes_per_point": round(tot / points),
                            "tb_s_over_mfma_kernels": round(tot / (mfma_ms * 1e-3) / 1e12, 2),
                            "frac_of_8tb_s": round(tot / (mfma_ms * 1e-3) / 1e12 / PEAK_HBM_TB_S, 3)}
            source = {"file": "profiles/hbm_traffic.json", "collected_by": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                      "passes over this bench (tools/pmc_passes.sh, profiles/summarize_pmc.py; FETCH_SIZE x2 per "
                      "MI355X_MICROARCH.md)", "summary": meta.get("summary"), "build": meta.get("build"),
                      "note": "counters cannot be read inside the timed run; these bytes belong to the build named here"}
        except Exception:
            traffic = None
    d = per[dom]
    return {"bound": "mfma", "kernel": dom, "mfma": d["mfma"], "achieved": d["executed_tflops"], "peak": d["peak"],
            "unit": "TFLOP/s", "frac": d["frac"], "traffic": traffic, "traffic_source": source,
            "algorithmic_flops_per_launch": alg[dom] * n_cols, "algorithmic_tflops": d["algorithmic_tflops"],
            "note": ("achieved = executed bf16 MFMA flops (6 per algorithmic flop: exact 3-way bf16 split of both fp32 "
                     "operands, fp32 accumulate) / HIP-event duration of the launch; algorithmic_tflops is the "
                     "fp32-equivalent rate") if d["mfma"] == "bf16x6" else
                    "achieved = algorithmic flops / HIP-event duration on the f32-input MFMA",
            "step_frac": round(step_tf * mult_step / peak_step, 4),
            "step_algorithmic_tflops": round(step_tf, 2),
            "step_note": "step_frac = 6 F0 flops per point x points x executed products per flop / ms_per_step / peak of "
                         "that pipe: the whole step (all kernels, gaps included) against the ceiling of its matmuls",
            "all_mfma_kernels": per, "step_hbm": step_hbm,
            "other_kernels_ms": {k: round(v, 4) for k, v in kern.items() if k not in alg},
            "kernel_times_from": f"untimed pass of {PROFILE_STEPS} steps with HIP events on the launch stream (dudf_profile_*)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=100000, help="points per GPU")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--loss", choices=["eikonal", "full"], default="eikonal",
                    help="eikonal = loss_s1 weights [1e4,1e4,0,1e3] (headline metric); full = Hessian term on "
                         "(reference configs/train_cfg.json weights [1e4,1e4,1e4,1e3]), reported as a secondary number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the secondary 8x512 / 125 000 points-per-GPU block")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N "
                             "--master-addr 127.0.0.1 bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # DUDF_TEST_SHARE_GPU=1: functional check of the N>1 path on a ONE-GPU box (all ranks on cuda:0, gloo instead of
    # RCCL, which refuses two ranks on one device).  Never set for a measurement.
    share = os.environ.get("DUDF_TEST_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)

    R = Runner(args, world, rank, dev)
    el, kern, final_loss, n_global, n_hess = R.run(args.hidden, args.layers, args.points, args.steps, args.warmup, args.loss)
    config3 = None
    headline = args.loss == "eikonal" and args.hidden == 256 and args.layers == 8 and args.points == 100000
    if headline and not args.no_config3:
        # BASELINE.json configs[2]: SIREN 8x512, 1 M synthetic points sharded over 8 GPUs = 125 000 per GPU (weak scaling
        # at other N).  Reported as a block of its own, never as `value`.
        s3 = max(10, args.steps // 4)
        el3, kern3, loss3, ng3, _ = R.run(512, 8, 125000, s3, 3, "eikonal")
        ms3 = el3 / s3 * 1e3
        config3 = {"workload": f"SIREN 8x512 (w0=30), Eikonal loss_s1, 125000 synthetic points per GPU (global batch {ng3}; "
                               "BASELINE.json configs[2] at 8 GPUs), same step definition",
                   "value": ng3 * s3 / el3, "unit": "points/s", "ms_per_step": ms3, "steps": s3, "n_gpus": world,
                   "final_loss": loss3, "roofline": roofline_block(args, kern3, 512, 8, 125000, 0, ms3) if rank == 0 else None}

    if rank == 0:
        ms_step = el / args.steps * 1e3
        value = n_global * args.steps / el
        weights = W_EIKONAL if args.loss == "eikonal" else [1e4, 1e4, 1e4, 1e3]
        out = {
            # the headline label is BASELINE.json's metric and only applies to its configuration
            "metric": "train points/sec (SIREN fwd+∇x+Eikonal loss+bwd), 256×8 net, 100k pts" if headline
            else f"train points/sec (SIREN fwd+∇x+{'Eikonal loss' if args.loss == 'eikonal' else 'Hessian+full loss_s1'}+bwd), "
                 f"{args.hidden}×{args.layers} net, {args.points} pts [secondary configuration]",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 arithmetic throughout. The hidden-layer matmuls of the sweeps and the weight-gradient GEMM "
                          "run on bf16 MFMA with BOTH fp32 operands split exactly into three bf16 pieces (6 products, fp32 "
                          "accumulate: fp32-equivalent, held to the same parity tolerances as the f32-input MFMA kernels, "
                          "which DUDF_SWEEP=f32 / DUDF_WGRAD=f32 select); first/last layer, tails, loss and Adam are plain fp32",
            "config": {"workload": f"SIREN {args.layers}x{args.hidden} (w0=30), loss_s1 weights {weights} "
                                   f"({'Eikonal-only' if args.loss == 'eikonal' else 'Hessian term on'}), alpha=100, {args.points} "
                                   f"synthetic points per GPU (global batch {n_global}), step = fwd + df/dx + loss + bwd + "
                                   f"{'RCCL all-reduce + ' if world > 1 else ''}Adam",
                       "points_per_gpu": args.points, "global_batch": n_global, "hidden": args.hidden,
                       "layers": args.layers, "parallelism": f"point-batch sharding x{world}, replicated theta"},
            "roofline": roofline_block(args, kern, args.hidden, args.layers, args.points, n_hess, ms_step),
            "final_loss": final_loss,
            "config3": config3,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.hidden, args.layers, 123)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
