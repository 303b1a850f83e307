#!/usr/bin/env python
# coding: utf-8
"""Localise an intermittent wrong number in the training step.

P processes share cuda:0 (the contention of tests/test_multirank_gpu.py::test_eight_ranks_on_one_gpu, without the collectives);
each one freezes (theta, its shard of one global batch) and runs the SAME step R times.  Everything a sweep leaves in the workspace is
a per-point function of (theta, batch) — no atomics — so after every repeat the whole workspace is compared BIT FOR BIT with the first
run's (the loss sums' fp64 scratch excepted), d(theta) and the terms to the float-atomics noise.  A mismatch is reported as (rank,
repeat, array, layer, feature tile, column range, how many words, largest difference).

    python tools/stress_repeat.py [--procs 8] [--reps 300] [--n-global 100000] [--presteps 1] [--stash 7] [--deterministic 0]
                                  [--path fused|staggered] [--mode s1|s2] [--hidden 256] [--layers 8] [--opt name=value ...]
The parent only spawns (never touches the GPU)."""
import argparse
import ctypes
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--n-global", type=int, default=100000)
    ap.add_argument("--presteps", type=int, default=1, help="engine steps from the seed-123 weights before theta is frozen")
    ap.add_argument("--stash", type=int, default=-1)
    ap.add_argument("--deterministic", type=int, default=0)
    ap.add_argument("--path", default="staggered")
    ap.add_argument("--mode", default="s1")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--max-report", type=int, default=6)
    ap.add_argument("--side-streams", type=int, default=0, help="N high-priority side streams that pick up behind every weight-gradient group, like the "
                    "async gloo all-reduces of the staggered engine path: wait for the compute stream, D2H + H2D of the group's slice through pinned memory")
    ap.add_argument("--churn", type=int, default=0, help="extra processes that keep starting short-lived GPU processes (import, load the library, three small "
                    "steps on four streams, exit) while the P stress processes run: queue creation / destruction and code-object loads by OTHER processes")
    ap.add_argument("--sync-phases", type=int, default=0, help="1: compare the workspace after every phase, not only after the step")
    return ap.parse_args()


def child(a):
    import numpy as np
    import torch
    from diffudf_amd import hip_ops as hip, synth, _lib
    from diffudf_amd.engine import TrainEngine
    rank, world = a.child, a.procs
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if a.stash >= 0:
        hip.set_option("stash", a.stash)
    if a.deterministic:
        hip.set_option("deterministic", 1)
    for item in a.opt:
        k, v = item.split("=", 1)
        hip.set_option(k, int(v))
    hidden = [a.hidden] * a.layers
    N = a.n_global

    def shard(step):
        idx = synth.stratified_shard(N, rank, world)
        parts = np.split(idx, np.flatnonzero(np.diff(idx) != 1) + 1)
        b = [synth.training_batch(N, seed=5, step=step, lo=int(p[0]), hi=int(p[-1]) + 1) for p in parts]
        x, nrm, sdf = [torch.from_numpy(np.concatenate([q[k] for q in b])).to(dev) for k in range(3)]
        return x, nrm, sdf.reshape(-1)

    theta = torch.from_numpy(synth.flatten_params(synth.siren_params(hidden, seed=123))).to(dev)
    W1 = [1e4, 1e4, 0.0, 1e3]
    if a.presteps:
        # theta_k of the SINGLE-process trajectory on the global batch would need every rank's shard; a frozen theta only has to be
        # "a theta the run visits": take this rank's own shard as if it were the whole batch
        eng = TrainEngine(hidden, theta)
        for t in range(a.presteps):
            x, nrm, sdf = shard(t)
            eng.step(0, x, nrm, sdf, W1, 100.0, lr=1e-4, n_global=N)
        torch.cuda.synchronize()
    x, nrm, sdf = shard(a.presteps)
    n = x.shape[0]
    cfg = hip.make_cfg(hidden)
    ws = hip.workspace_for(cfg, n, dev)
    lay = (ctypes.c_int64 * 10)()
    _lib.check(_lib.load().dudf_debug_stash_layout(ctypes.byref(cfg), n, 0, lay), "layout")
    names = ["S", "C", "Q", "E", "A", "Z", "R", "ZS"]
    offs = sorted((int(lay[i]), names[i]) for i in range(8) if names[i] != "ZS")
    H, L = a.hidden, a.layers
    mask = hip.stash_mode(cfg, n, 0)
    np_cols = int(lay[8]) // 16
    nbytes = ws.nbytes
    tail = 2 * 16 * 4 + 256                       # the fp64 loss sums (atomicAdd order) at the very end
    mode = {"s1": 0, "s2": 1}[a.mode]
    weights = W1 if mode == 0 else [1e5, 1e5]
    ones = torch.ones(4, device=dev)
    flat = torch.zeros(theta.numel() + 4, device=dev)
    dtheta, terms = flat[:theta.numel()], flat[theta.numel():]
    sl = hip.layer_slices(cfg)
    groups = TrainEngine._layer_groups(L)

    def where(off):
        name, base = "head", 0
        for o, nm in offs:
            if off >= o:
                name, base = nm, o
        if name == "head":
            return f"head+{off}"
        if off >= offs[-1][0] and name == offs[-1][1]:
            # past the last stash array: the fixed-point column scales / acc
            last_sz = L * H * np_cols * (3 if (mask & 1) else 4)
            if off - base >= last_sz:
                return f"fx/acc+{off - base - last_sz}"
        rel = off - base
        p24 = {"S": mask & 1, "Q": mask & 1, "A": mask & 1, "Z": mask & 1, "R": mask & 2, "E": mask & 2, "C": mask & 4}[name]
        if p24:
            layer_b = H * np_cols * 3
            li, r = divmod(rel, layer_b)
            tile_b = (np_cols // 16) * 64 * 12
            T, r2 = divmod(r, tile_b)
            g, r3 = divmod(r2, 768)
            return f"{name}[layer {li} tile {T} colgroup {g} (cols {16 * g}..) byte {r3}]"
        layer_b = H * np_cols * 4
        li, r = divmod(rel, layer_b)
        fq, r2 = divmod(r, np_cols * 16)
        return f"{name}[layer {li} fquad {fq} col {r2 // 16} +{r2 % 16}]"

    def step_once(sync_cb=None):
        stats = None
        if mode == 1:
            stats = hip.s2_forward_stats(cfg, theta, x, sdf, ws)
            terms.zero_(); terms[:2] = hip.s2_terms(stats, weights)
        else:
            hip.loss_forward(cfg, mode, theta, x, nrm, sdf, N, weights, 100.0, ws, out=terms)
        if sync_cb:
            sync_cb("forward")
        if a.path == "fused":
            hip.loss_backward(cfg, mode, theta, x, nrm, sdf, N, weights, 100.0, ones, stats, ws, dtheta=dtheta)
        else:
            hip.set_wgrad_max_workgroups(240)
            hip.loss_backward_sweeps(cfg, mode, theta, nrm, sdf, N, weights, 100.0, ones, stats, ws, n_local=n)
            if sync_cb:
                sync_cb("bwd_sweeps")
            for gi, (b, e) in enumerate(groups):
                hip.weight_gradient(cfg, n, mode != 1, b, e, dtheta, ws)
                side_copy(gi, sl[b][0], sl[e - 1][1])
            hip.weight_gradient(cfg, n, mode != 1, -1, 0, dtheta, ws)
            side_copy(len(groups), sl[0][0], sl[0][1]); side_copy(len(groups) + 1, sl[L][0], flat.numel())
            for s_ in side:
                torch.cuda.current_stream().wait_stream(s_)

    side = [torch.cuda.Stream(priority=-1) for _ in range(a.side_streams)]
    pin = [torch.empty(flat.numel()).pin_memory() for _ in side]
    back = [torch.empty_like(flat) for _ in side]

    def side_copy(i, lo, hi):
        if not side:
            return
        s_ = side[i % len(side)]
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            pin[i % len(side)][lo:hi].copy_(flat[lo:hi], non_blocking=True)
            back[i % len(side)][lo:hi].copy_(pin[i % len(side)][lo:hi], non_blocking=True)

    snaps = {}

    def snap_cb(tag):
        torch.cuda.synchronize()
        snaps[tag] = ws.buf.clone()

    step_once(snap_cb if a.sync_phases else None)
    torch.cuda.synchronize()
    ref_ws = ws.buf.clone()
    ref_d = dtheta.double().clone()
    ref_t = terms.double().clone()
    assert torch.isfinite(ref_d).all() and torch.isfinite(ref_t).all()
    lmax = [float(ref_d[lo:hi].abs().max()) for lo, hi in sl]
    print(f"[rank {rank}] n {n} stash mask {mask} terms {ref_t.tolist()} |dtheta|max {float(ref_d.abs().max()):.4e}", flush=True)

    def diff_ws(cur, ref, tag, rep):
        a64, b64 = cur[:(nbytes - tail) // 8 * 8].view(torch.int64), ref[:(nbytes - tail) // 8 * 8].view(torch.int64)
        ne = (a64 != b64)
        cnt = int(ne.sum())
        if cnt == 0:
            return 0
        pos = ne.nonzero().flatten()
        first, last = int(pos[0]) * 8, int(pos[-1]) * 8
        # group the mismatching words by 768-byte / row neighbourhood for a readable summary
        lines = [f"[rank {rank}] rep {rep} {tag}: {cnt} mismatching 8-byte words; first {where(first)}, last {where(last)}"]
        p_cpu = (pos[:: max(1, len(pos) // 12)] * 8).tolist()
        for o in p_cpu[:12]:
            aw = cur[o:o + 8].cpu().numpy(); bw = ref[o:o + 8].cpu().numpy()
            lines.append(f"      @{where(o)}: got {aw.tobytes().hex()} want {bw.tobytes().hex()}")
        print("\n".join(lines), flush=True)
        return cnt

    bad = 0
    reports = 0
    t0 = time.time()
    for rep in range(1, a.reps + 1):
        if a.sync_phases:
            def cb(tag, rep=rep):
                nonlocal bad, reports
                torch.cuda.synchronize()
                if not torch.equal(ws.buf[:nbytes - tail], snaps[tag][:nbytes - tail]):
                    bad += 1
                    if reports < a.max_report:
                        reports += 1
                        diff_ws(ws.buf, snaps[tag], "after " + tag, rep)
            step_once(cb)
        else:
            step_once()
        torch.cuda.synchronize()
        ok_ws = torch.equal(ws.buf[:nbytes - tail], ref_ws[:nbytes - tail])
        d = dtheta.double()
        e_t = float(((terms.double() - ref_t).abs() / ref_t.abs().max()).max())
        e_l = [float((d[lo:hi] - ref_d[lo:hi]).abs().max()) / max(m, 1e-300) for (lo, hi), m in zip(sl, lmax)]
        tol = 0.0 if a.deterministic else 2e-5
        ok_d = max(e_l) <= tol and e_t <= (0.0 if a.deterministic else 2e-6)
        if not (ok_ws and ok_d):
            bad += 1
            if reports < a.max_report:
                reports += 1
                print(f"[rank {rank}] rep {rep}: workspace {'same' if ok_ws else 'DIFFERS'}; terms err {e_t:.2e}; dtheta err per layer "
                      + " ".join(f"{v:.1e}" for v in e_l), flush=True)
                if not ok_ws:
                    diff_ws(ws.buf, ref_ws, "after step", rep)
    print(f"[rank {rank}] {a.reps} repeats in {time.time() - t0:.1f} s: {bad} bad", flush=True)
    return 1 if bad else 0


def main():
    a = parse()
    if a.child >= 0:
        sys.exit(child(a))
    stop = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"stress_stop_{os.getpid()}")
    churn_src = ("import os, sys, subprocess\n"
                 "n = 0\n"
                 "while not os.path.exists(sys.argv[1]):\n"
                 f"    subprocess.run([sys.executable, {os.path.abspath(__file__)!r}, '--child', '0', '--procs', '1', '--reps', '2', '--n-global', '4000', '--side-streams', '4', '--presteps', '1'], "
                 "stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)\n"
                 "    n += 1\n"
                 "print('[churn] generations:', n, flush=True)\n")
    churners = [subprocess.Popen([sys.executable, "-c", churn_src, stop], cwd=REPO) for _ in range(a.churn)]
    procs = []
    for r in range(a.procs):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--child", str(r)], env=env, cwd=REPO))
    rc = 0
    for p in procs:
        rc |= p.wait()
    open(stop, "w").close()
    for c in churners:
        c.wait()
    os.unlink(stop)
    print("STRESS", "FAIL" if rc else "OK", " ".join(sys.argv[1:]), flush=True)
    sys.exit(1 if rc else 0)


if __name__ == "__main__":
    main()
