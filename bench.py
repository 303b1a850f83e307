#!/usr/bin/env python
# coding: utf-8
"""bench.py — train points/sec of the DiffUDF hot path on MI355X (BASELINE.json metric).

One "step" = one optimizer step of the reference loop (reference train.py:195-222) on one batch of synthetic points
already resident in HBM:
    SIREN 8x256 forward + df/dx + hyperbolic-scaled Eikonal/UDF loss (loss_s1, Hessian weight 0)
    + backward to theta + (all-reduce) + Adam.
Workload at N GPUs: 100 000 points PER GPU (weak scaling), uniform in [-1,1]^3, thirds [on-surface | far | near] like
the reference sampler.  Arithmetic is fp32 throughout; the hidden-layer matmuls run on the 16-bit matrix cores with both
fp32 operands split into two fp16 pieces ("fp16x3": three products, fp32 accumulate; --opt split=0: three bf16 pieces,
six products) — fp32-equivalent, held to the fp32 parity tolerances.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--hidden H] [--no-cpu-baseline] [--no-config3]

Prints ONE JSON line on rank 0.  Timing protocol: W untimed warm-up steps, then EXACTLY K steps between two
barrier + synchronize pairs with the library's HIP-event profiler OFF; `value` = global batch x K / that wall time (MAX
over ranks) and `ms_per_step` = wall time / K — the definition rounds 1 and 2 reported.  One event per step on the launch
stream also gives the per-step durations: `ms_per_step_median` (what round 3 reported as `value`; the two differ by < 1 %).
The per-kernel durations behind `roofline` come from a SEPARATE, untimed pass of a few steps with the profiler on (events
on the launch stream, minus what an empty event pair reads; `kernel_times_sum_ms` is printed beside the step it belongs
to), together with the shader clock each kernel ran at (dudf_profile_clocks) and the products per multiply of the kernel
that was actually dispatched (dudf_profile_products).

`--gpus N` with N > 1 and no torchrun environment: this file starts `python -m torch.distributed.run --nproc-per-node N`
on itself as a CHILD process before anything touches the GPU, forwards its output and exits with its return code.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
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
    model, phys = "unknown", None
    try:                                               # report the host (model string, physical cores)
        pairs, pid, cid = set(), None, None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None and cid is not None:
                pairs.add((pid, cid)); pid = cid = None
        phys = len(pairs) or None
    except OSError:
        pass
    return {"value": head["value"], "unit": "points/s", "cores": cores, "kind": "port",
            "threads": cores, "physical_cores": phys, "logical_cores": os.cpu_count(), "logical_cores_visible": avail, "cpu_model": model,
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
        eng = TrainEngine(hid, theta, collectives=self.args.collectives)
        weights = W_EIKONAL if loss == "eikonal" else [1e4, 1e4, 1e4, 1e3]
        n_hess = 0
        if loss == "full":                              # shards come out as [on | far | near]: on-surface first
            n_hess = int((sdf == 0).sum())
            assert bool((sdf[:n_hess] == 0).all())
        step = lambda: eng.step(hip_ops.LOSS_S1, x, nrm, sdf, weights, ALPHA, lr=1e-4, n_global=n_global, n_hess=n_hess)  # noqa: E731
        for _ in range(warmup):
            step()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        self.barrier()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            terms = step()
            marks[i + 1].record()                      # one event per step on the launch stream: no host synchronisation
        self.barrier()
        el = time.perf_counter() - t0
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        med = per_step[steps // 2] if steps % 2 else 0.5 * (per_step[steps // 2 - 1] + per_step[steps // 2])
        final_loss = float(terms.sum())
        if self.world > 1:                              # the slowest rank sets the pace: MAX of both figures
            t = torch.tensor([el, med], dtype=torch.float64, device=self.dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el, med = float(t[0]), float(t[1])
        info = {"median_ms": med, "min_ms": per_step[0], "max_ms": per_step[-1], "kern": {}, "clocks_mhz": {}, "products": {}, "phases_ms": None,
                "event_cost_per_launch_ms": None,
                "event_pair_overhead_ms": None, "profiled_step_ms": None, "collectives": getattr(eng, "collectives", None)}
        if profile_steps:
            # what an empty event pair reads on this stream: subtracted from every per-kernel duration (round 2: the raw
            # figures summed to more than the step they are part of)
            pairs = []
            for _ in range(64):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); b.record(); pairs.append((a, b))
            torch.cuda.synchronize()
            ov = sorted(a.elapsed_time(b) for a, b in pairs)[32]
            self.lib.dudf_profile_enable(1)
            eng.profile = self.world > 1
            pm = [torch.cuda.Event(enable_timing=True) for _ in range(profile_steps + 1)]
            pm[0].record()
            for i in range(profile_steps):
                step()
                pm[i + 1].record()
            torch.cuda.synchronize()
            self.lib.dudf_profile_enable(0)
            eng.profile = False
            buf = ctypes.create_string_buffer(4096)
            self.lib.dudf_profile_dump(buf, len(buf))
            raw, per_step, launches = {}, {}, 0
            for line in buf.value.decode().splitlines():
                name, cnt, tot = line.split()
                # ms per STEP of every launch under this name (with Hessian-path points a sweep is two launches: the quad
                # columns and the plain columns)
                raw[name] = float(tot) / profile_steps
                per_step[name] = int(cnt) / profile_steps
                launches += int(cnt)
            # An empty event pair reads `ov`: that much of every bracketed duration is the bracket itself and is taken off.
            # Nothing else is: the events also serialise the stream a little (profiled steps run slower than timed ones),
            # which is reported side by side (kernel_times_sum_ms vs ms_per_step, profiled_step_ms) instead of being
            # subtracted — round 3 did, and it zeroed the four helper kernels and shaved 5 % off every sweep.
            prof_step = sorted(pm[i].elapsed_time(pm[i + 1]) for i in range(profile_steps))[profile_steps // 2]
            per_launch = ov
            for name, v in raw.items():
                info["kern"][name] = max(v - per_launch * per_step[name], 0.0)
            self.lib.dudf_profile_products(buf, len(buf))
            for line in buf.value.decode().splitlines():
                name, prod = line.split()
                info["products"][name] = int(prod)
            info["event_cost_per_launch_ms"] = per_launch
            self.lib.dudf_profile_clocks(buf, len(buf))
            for line in buf.value.decode().splitlines():
                name, mhz = line.split()
                info["clocks_mhz"][name] = float(mhz)
            info["event_pair_overhead_ms"] = ov
            info["profiled_step_ms"] = prof_step
            if self.world > 1:
                info["phases_ms"] = {k: round(v, 4) for k, v in eng.phase_times().items()}
        del eng
        return el, info, final_loss, n_global, n_hess


# stash traffic of each kernel in BYTES per (layer, feature, column) — DESIGN.md §3.2 — for the stash formats (dudf_stash_mode):
# 0 = every array fp32 (17 units); 6 = R, E as 24-bit floats and C as 24-bit fixed point = 3 bytes (15 units; the default);
# 7 = S, Q, A, Z at 24 bits as well (12.75 units).
#   forward: writes S, C | reverse: reads C, S, writes Q, R | adjoint forward: reads C, R, writes A, E |
#   adjoint reverse: reads C, E, writes Z | weight gradients: read Q, A, Z, S.                  (bytes read, bytes written, layers)
STASH_BYTES = {0: {"sweep_fwd": (0, 8, "L"), "sweep_rev": (8, 8, "L"), "sweep_adj_fwd": (8, 8, "L"), "sweep_adj_rev": (8, 4, "L"),
                   "wgrad_hidden": (16, 0, "L-1"), "wgrad_small": (16, 0, "1")},
               6: {"sweep_fwd": (0, 7, "L"), "sweep_rev": (7, 7, "L"), "sweep_adj_fwd": (6, 7, "L"), "sweep_adj_rev": (6, 4, "L"),
                   "wgrad_hidden": (16, 0, "L-1"), "wgrad_small": (16, 0, "1")},
               7: {"sweep_fwd": (0, 6, "L"), "sweep_rev": (6, 6, "L"), "sweep_adj_fwd": (6, 6, "L"), "sweep_adj_rev": (6, 3, "L"),
                   "wgrad_hidden": (12, 0, "L-1"), "wgrad_small": (12, 0, "1")}}
SPLIT_BIT = {"sweep_fwd": 0, "sweep_rev": 1, "sweep_adj_fwd": 2, "sweep_adj_rev": 3, "wgrad_hidden": 4}


def same_build(meta, args):
    """Were the PMC files collected on the options this run uses?  (a label check against counters of another
    configuration would be meaningless; --opt switches change what runs)"""
    return not args.opt and not os.environ.get("DUDF_LIB")


def roofline_block(args, info, hidden, layers, points, n_hess, ms_step):
    """Every MFMA kernel against BOTH ceilings it could be bound by — the matrix pipe (executed MFMA flops = algorithmic
    flops x products per multiply of its operand split) and HBM (the stash bytes its dataflow moves) — and
    the `roofline` object of the dominant (longest) kernel, priced on whichever of the two it sits closer to."""
    from diffudf_amd import _lib, hip_ops
    # Per-kernel durations: HIP events around every launch of an untimed pass.  The brackets serialise the stream — the profiled
    # step runs ~7 % longer than the timed one and the raw durations SUM to more than the timed step — so each
    # kernel's share of the profiled pass is applied to the step that was actually timed: avg_ms = raw x timed_step_ms / sum of raw.
    # (rocprofv3 --kernel-trace averages of the same command, profiles/r05_h_rocprofv3_kernel_stats.csv, agree within 5 %; the raw
    # event figures stay beside them as avg_ms_events.)
    kern_raw = dict(info["kern"])
    raw_sum = sum(kern_raw.values())
    scale = (ms_step / raw_sum) if raw_sum > 0 else 1.0
    kern = {k: v * scale for k, v in kern_raw.items()}
    F0 = f0(hidden, layers)
    hid = 2 * (layers - 1) * hidden * hidden                    # hidden x hidden matmul flops per point
    alg = {"sweep_fwd": F0, "sweep_rev": F0, "sweep_adj_fwd": F0, "sweep_adj_rev": F0,
           "wgrad_hidden": 2 * hid, "wgrad_small": 2 * (F0 - hid)}
    # which matrix-core path a kernel took: the launchers record the products per algorithmic multiply of the kernel they
    # dispatch (dudf_profile_products) — 1 = f32-input MFMA (157.3 TF), 6 = exact three-piece bf16 split, 3 = fp16 hi/lo split
    # (both on the 2.5 PF dense 16-bit pipe).  Round 3 kept a table here and it went stale.
    LABEL = {1: ("f32", PEAK_F32_MFMA_TFLOPS), 3: ("fp16x3", PEAK_BF16_MFMA_TFLOPS), 6: ("bf16x6", PEAK_BF16_MFMA_TFLOPS)}
    stash_mode = max(hip_ops.stash_mode(hip_ops.make_cfg([hidden] * layers), points, n_hess), 0)
    n_cols = points + 3 * n_hess                                 # columns the MFMA kernels process: 4 per Hessian-path point
    Lmap = {"L": layers, "L-1": layers - 1, "1": 1}
    per = {}
    for k, fl in alg.items():
        if k not in kern or kern[k] <= 0:
            continue
        tf = fl * n_cols / (kern[k] * 1e-3) / 1e12
        mult = info["products"].get(k, 1)                        # wgrad_small: fp32 vector ALU, no matrix core
        name, peak = LABEL[mult]
        rd, wr, lk = STASH_BYTES[stash_mode][k]
        sbytes = (rd + wr) * Lmap[lk] * hidden * n_cols
        tbs = sbytes / (kern[k] * 1e-3) / 1e12
        per[k] = {"avg_ms": round(kern[k], 4), "avg_ms_events": round(kern_raw[k], 4), "algorithmic_tflops": round(tf, 2), "mfma": name,
                  "executed_tflops": round(tf * mult, 2), "peak": peak, "frac": round(tf * mult / peak, 4),
                  "stash_bytes_per_launch": sbytes, "stash_tb_s": round(tbs, 2), "hbm_frac": round(tbs / PEAK_HBM_TB_S, 4),
                  "clock_mhz": info["clocks_mhz"].get(k)}
    if not per:
        return None
    dom = max(per, key=lambda k: per[k]["avg_ms"])
    mfma_ms = sum(kern[k] for k in alg if k in kern)
    mult_step = {"fp16x3": 3, "bf16x6": 6, "f32": 1}[per.get("sweep_fwd", per[dom])["mfma"]]
    peak_step = PEAK_F32_MFMA_TFLOPS if mult_step == 1 else PEAK_BF16_MFMA_TFLOPS
    step_tf = 6 * F0 * n_cols / (ms_step * 1e-3) / 1e12
    traffic, step_hbm, source = None, None, None
    # PMC bytes exist for the two workloads this file reports: the headline and config 3
    prof_json = os.path.join(REPO, "profiles", {(256, 100000): {0: "hbm_traffic_fp32.json", 6: "hbm_traffic_mask6.json", 7: "hbm_traffic.json"}[stash_mode],
                                                (512, 125000): {7: "hbm_traffic_8x512_mask7.json"}.get(stash_mode, "hbm_traffic_8x512.json")}.get((hidden, points), "-"))
    if os.path.exists(prof_json) and args.loss == "eikonal" and layers == 8:
        try:                                        # PMC bytes were collected on exactly this workload, see `source`
            tr = json.load(open(prof_json))
            meta = tr.get("_meta", {})
            traffic = tr.get(dom, {}).get("hbm_bytes_per_launch")
            tot = 0.0
            for k, d in per.items():
                # the label against the counters: SQ_INSTS_MFMA x flops per instruction / algorithmic flops = products executed
                mi = tr.get(k, {}).get("mfma_insts_per_launch")
                if mi and d["mfma"] != "f32":
                    flop_per_inst = 2 * 32 * 32 * 16 if k == "wgrad_hidden" else 2 * 16 * 16 * 32
                    measured = mi * flop_per_inst / (alg[k] * n_cols)
                    d["mfma_products_pmc"] = round(measured, 2)
                    if same_build(meta, args) and abs(measured - {"fp16x3": 3, "bf16x6": 6}[d["mfma"]]) > 0.5:
                        raise SystemExit(f"bench.py: {k} is labelled {d['mfma']} but the counters of {prof_json} show "
                                         f"{measured:.2f} products per multiply")
                b = tr.get(k, {}).get("hbm_bytes_per_launch")
                if b:
                    d["hbm_bytes_per_launch"] = b
                    d["hbm_tb_s"] = round(b / (d["avg_ms"] * 1e-3) / 1e12, 2)
                    tot += b
            if tot:
                step_hbm = {"bytes_per_step": tot, "bytes_per_point": round(tot / points),
                            "tb_s_over_mfma_kernels": round(tot / (mfma_ms * 1e-3) / 1e12, 2),
                            "frac_of_8tb_s": round(tot / (mfma_ms * 1e-3) / 1e12 / PEAK_HBM_TB_S, 3),
                            "input_output_bytes_per_step": 28 * points + 4 * (alg["sweep_fwd"] // 2 + 1),
                            "note": "the stash (activations and adjoints kept between the four sweeps and the weight-gradient "
                                    "GEMM) is this dataflow's traffic; the step's inputs and outputs alone are 28 B per point + 4 B "
                                    "per parameter (SURVEY.md §8(d))"}
            source = {"file": "profiles/" + os.path.basename(prof_json), "collected_by": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                      "passes over this bench (tools/pmc_passes.sh, profiles/summarize_pmc.py; FETCH_SIZE x2 per "
                      "MI355X_MICROARCH.md)", "summary": meta.get("summary"), "build": meta.get("build"),
                      "note": "counters cannot be read inside the timed run; these bytes belong to the build named here"}
        except Exception:
            traffic = None
    d = per[dom]
    # the roofline that bounds this path is the matrix pipe (algorithmic intensity ~1e5 flop/B on the step's own
    # inputs and outputs) — `roofline` prices the DOMINANT kernel (longest launch) there: executed MFMA flops = algorithmic flops x
    # products per multiply of the operand split / launch duration, against the dense 16-bit peak.  What the dataflow moves through
    # HBM between its kernels (the stash) is reported beside it, not instead of it: `hbm_stash_frac` of the same kernel, and
    # `wasted_traffic_ratio` = measured HBM bytes per step / the step's algorithmic bytes (28 B per point in + 4 B per parameter out).
    io_bytes = 28 * points + 4 * (F0 // 2 + 1)
    out = {"bound": "mfma", "kernel": dom, "mfma": d["mfma"], "clock_mhz": d["clock_mhz"],
           "achieved": d["executed_tflops"], "peak": d["peak"], "unit": "TFLOP/s", "frac": d["frac"],
           "algorithmic_tflops": d["algorithmic_tflops"], "algorithmic_flops_per_launch": alg[dom] * n_cols,
           "products_per_multiply": info["products"].get(dom, 1), "avg_launch_ms": d["avg_ms"],
           "traffic": traffic, "traffic_source": source,
           "hbm_stash_frac": d["hbm_frac"], "hbm_stash_gb_s": round(d["stash_tb_s"] * 1e3, 1),
           "stash_bytes_per_launch": d["stash_bytes_per_launch"],
           "algorithmic_io_bytes_per_step": io_bytes,
           "wasted_traffic_ratio": round(step_hbm["bytes_per_step"] / io_bytes, 1) if step_hbm else None,
           "stash": {0: "fp32 (17 array-layer units of 4 bytes per value and column)",
                     6: "R, E as 24-bit floats and C as 24-bit fixed point (3 bytes per value, tile-major), S, Q, A, Z fp32: 15 units",
                     7: "all seven arrays at 24 bits, tile-major: R, E floats with a 16-bit significand; C and — relative to a per-column power "
                        "of two — S, Q, A, Z fixed point on a 2^-22 grid: 12.75 units"}[stash_mode],
           "note": "bound = the matrix pipe, as SURVEY.md 8(d) names it: achieved = F0 x columns x products / avg_launch_ms of the "
                   "longest kernel of the step, peak = dense 16-bit MFMA (MI355X_MICROARCH.md); algorithmic_tflops = the same without "
                   "the products of the operand split (fp32-equivalent).  The kernel is nowhere near that ceiling because the "
                   "dataflow streams its stash through HBM: hbm_stash_frac = stash bytes of this kernel / avg_launch_ms / 8 TB/s, "
                   "wasted_traffic_ratio = PMC HBM bytes of the whole step / (28 B per point + 4 B per parameter)"}
    out.update({"step_frac": round(step_tf * mult_step / peak_step, 4), "step_algorithmic_tflops": round(step_tf, 2),
                "step_note": "step_frac = 6 F0 flops per point x points x executed products per flop / ms_per_step / peak of "
                             "that pipe: the whole step (all kernels, gaps included) against the ceiling of its matmuls",
                "all_mfma_kernels": per, "step_hbm": step_hbm,
                "other_kernels_ms": {k: round(v, 4) for k, v in kern.items() if k not in alg},
                "kernel_times_sum_ms": round(raw_sum, 4), "kernel_time_scale": round(scale, 4), "profiled_step_ms": info["profiled_step_ms"],
                "event_pair_overhead_ms": info["event_pair_overhead_ms"], "event_cost_per_launch_ms": info["event_cost_per_launch_ms"],
                "timed_step_ms": round(ms_step, 4),
                "kernel_times_from": f"untimed pass of {PROFILE_STEPS} steps with HIP events on the launch stream (dudf_profile_*), "
                                     "minus what an empty event pair reads (event_pair_overhead_ms) per launch (avg_ms_events; their sum "
                                     "kernel_times_sum_ms belongs beside profiled_step_ms, the same pass), then scaled by kernel_time_scale = "
                                     "timed_step_ms / kernel_times_sum_ms so that the shares add up to the step that was timed (avg_ms); "
                                     "clock_mhz = shader clock over the lifetime of the kernel's first workgroup (s_memtime / "
                                     "s_memrealtime); mfma = the split the dispatched kernel uses (dudf_profile_products)"})
    return out


def self_launch(n):
    """`python bench.py --gpus N` as typed (N > 1, no torchrun environment): one rank per GPU through torch.distributed.run,
    started as a CHILD process — never an exec — and before this process has made any GPU call (importing torch does not
    initialise the device; asserted below and in tests/test_bench_launch.py).  The child's stdout (rank 0's one JSON line)
    and stderr are inherited; the return value is the child's exit code."""
    assert not torch.cuda.is_initialized(), "bench.py touched the GPU before spawning its ranks"
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    if "--share-device" in sys.argv:
        # n processes on ONE GPU: one hardware queue each.  With the default four per process 8 ranks oversubscribe the chip's 24 user
        # queue slots, the hardware scheduler starts time-slicing them, and on this platform a restored queue then skips or replays ONE
        # dispatch on single XCDs (8/120 runs against 0/120 on the same box)
        env.setdefault("GPU_MAX_HW_QUEUES", "1")
    if os.environ.get("DUDF_BENCH_DRY_LAUNCH") == "1":  # tests: show what would be started, start nothing
        print(json.dumps({"launch": cmd, "cuda_initialized": torch.cuda.is_initialized()}))
        return 0
    return subprocess.run(cmd, env=env).returncode


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
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="run-time option of the library (dudf_set_option; include/dudf_hip.h lists them), e.g. --opt stash=7 "
                         "--opt split=0; A/B runs only: the headline is the default build")
    ap.add_argument("--collectives", choices=["staggered", "fused"], default=None, help="N > 1: TrainEngine's all-reduce schedule")
    ap.add_argument("--share-device", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank on cuda:0 over gloo (RCCL refuses two ranks on one device).  A functional "
                         "check of the sharded path, never a measurement: the JSON line says \"share_device\": true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the secondary 8x512 blocks (125 000 points per GPU; 1 M points on one GPU)")
    ap.add_argument("--no-config3-1m", action="store_true", help="skip only the 8x512 / 1 000 000-points-on-one-GPU block (110 GB workspace)")
    args = ap.parse_args()

    if args.share_device:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")  # read by the HIP runtime when this process first touches the GPU (see self_launch)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    share = args.share_device
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)

    from diffudf_amd import hip_ops
    for item in args.opt:
        k, v = item.split("=", 1)
        hip_ops.set_option(k, int(v))
    R = Runner(args, world, rank, dev)
    el, info, final_loss, n_global, n_hess = R.run(args.hidden, args.layers, args.points, args.steps, args.warmup, args.loss)
    config3 = None
    headline = args.loss == "eikonal" and args.hidden == 256 and args.layers == 8 and args.points == 100000
    if headline and not args.no_config3:
        # BASELINE.json configs[2]: SIREN 8x512, 1 M synthetic points sharded over 8 GPUs = 125 000 per GPU (weak scaling
        # at other N).  Reported as a block of its own, never as `value`.
        s3 = max(10, args.steps // 4)
        el3, info3, loss3, ng3, _ = R.run(512, 8, 125000, s3, 3, "eikonal")
        ms3 = el3 / s3 * 1e3
        config3 = {"workload": f"SIREN 8x512 (w0=30), Eikonal loss_s1, 125000 synthetic points per GPU (global batch {ng3}; "
                               "BASELINE.json configs[2] at 8 GPUs), same step definition",
                   "value": ng3 / (ms3 * 1e-3), "unit": "points/s", "ms_per_step": ms3, "ms_per_step_median": info3["median_ms"],
                   "steps": s3, "n_gpus": world, "final_loss": loss3, "phases_ms": info3["phases_ms"],
                   "roofline": roofline_block(args, info3, 512, 8, 125000, 0, ms3) if rank == 0 else None}

    config3_1m = None
    if headline and not args.no_config3 and world == 1 and not args.no_config3_1m:
        # BASELINE.json configs[2] as north_star words it — "1 M points, 8x512, sharded across 8" — on ONE GPU: the whole 1 M-point
        # batch in one 110 GB workspace (288 GB of HBM).  The denominator of the ">= 6x at 8 GPUs vs 1" target on that config
        # (strong scaling: 8 x 125 000 = the same global batch); the `config3` block above is its per-GPU share.
        el4, info4, loss4, ng4, _ = R.run(512, 8, 1000000, 4, 1, "eikonal", profile_steps=2)
        ms4 = el4 / 4 * 1e3
        config3_1m = {"workload": "SIREN 8x512 (w0=30), Eikonal loss_s1, 1 000 000 synthetic points on ONE GPU (BASELINE.json "
                                  "configs[2]'s global batch unsharded), same step definition",
                      "value": ng4 / (ms4 * 1e-3), "unit": "points/s", "ms_per_step": ms4, "ms_per_step_median": info4["median_ms"],
                      "steps": 4, "n_gpus": 1, "scaling_note": "strong-scaling denominator: compare with config3.value at --gpus 8 "
                      "(8 x 125 000 points = this global batch)", "final_loss": loss4,
                      "kernels_ms": {k: round(v, 3) for k, v in info4["kern"].items()}}

    if rank == 0:
        ms_step = el / args.steps * 1e3                 # wall time between the barriers / K, MAX over ranks
        value = n_global / (ms_step * 1e-3)
        weights = W_EIKONAL if args.loss == "eikonal" else [1e4, 1e4, 1e4, 1e3]
        out = {
            # the headline label is BASELINE.json's metric and only applies to its configuration
            "metric": "train points/sec (SIREN fwd+∇x+Eikonal loss+bwd), 256×8 net, 100k pts" if headline
            else f"train points/sec (SIREN fwd+∇x+{'Eikonal loss' if args.loss == 'eikonal' else 'Hessian+full loss_s1'}+bwd), "
                 f"{args.hidden}×{args.layers} net, {args.points} pts [secondary configuration]",
            **({"share_device": True, "share_device_note": "all ranks on cuda:0 over gloo: a functional check, not a measurement"} if args.share_device else {}),
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "ms_per_step_median": info["median_ms"], "ms_per_step_min": info["min_ms"],
            "ms_per_step_max": info["max_ms"],
            "timing_note": "value = global batch x steps / wall time between the two barrier + synchronize pairs (MAX over "
                           "ranks) — the mean, as in rounds 1-2; ms_per_step_median = median of the per-step durations (one "
                           "HIP event per step on the launch stream), which is what round 3 reported as value",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 arithmetic throughout. The hidden-layer matmuls of the sweeps and the weight-gradient GEMM "
                          "run on the 16-bit matrix cores with BOTH fp32 operands split into two fp16 pieces hi + lo (3 products "
                          "hi*hi + hi*lo + lo*hi, fp32 accumulate, power-of-two range scaling: 'fp16x3') or, with "
                          "option split=0, exactly into three bf16 pieces (6 products): fp32-equivalent, held to the same parity "
                          "tolerances as the f32-input MFMA kernels (options sweep_family=0 / wgrad_family=1); first/last layer, tails, "
                          "loss and Adam are plain fp32.  NOTHING of the stash crosses HBM at 32 bits: the seven per-layer arrays the step "
                          "keeps between its sweeps are stored at 24 bits — R = w0^2 s a and E = r Q as fp32 values rounded to a "
                          "16-bit significand (relative error <= 2^-16; read only by the adjoint sweeps); C = cos(w0 z) as fixed point on "
                          "a 2^-22 grid (absolute error 2^-23); S, Q, A, Z (the weight-gradient GEMM's operands) as the same fixed point "
                          "relative to a per-layer, per-column power of two 2^E (absolute error 2^(E-23)).  512-wide networks keep S, Q, "
                          "A, Z at fp32 (their kernel reads them back as the next layer's operand).  roofline.stash names the format of this run (dudf_stash_mode); every fp32 parity tolerance and "
                          "the trajectory bars are held in it (tests/test_traj50_gpu.py)",
            "config": {"workload": f"SIREN {args.layers}x{args.hidden} (w0=30), loss_s1 weights {weights} "
                                   f"({'Eikonal-only' if args.loss == 'eikonal' else 'Hessian term on'}), alpha=100, {args.points} "
                                   f"synthetic points per GPU (global batch {n_global}), step = fwd + df/dx + loss + bwd + "
                                   f"{'RCCL all-reduce + ' if world > 1 else ''}Adam",
                       "points_per_gpu": args.points, "global_batch": n_global, "hidden": args.hidden,
                       "layers": args.layers, "parallelism": f"point-batch sharding x{world}, replicated theta"},
            "roofline": roofline_block(args, info, args.hidden, args.layers, args.points, n_hess, ms_step),
            "phases_ms": info["phases_ms"], "collectives": info["collectives"] if world > 1 else None,
            "final_loss": final_loss,
            "config3": config3,
            "config3_1gpu_1M": config3_1m,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.hidden, args.layers, 123)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
