#!/usr/bin/env python
# coding: utf-8
"""Loop the legs of tests/test_multirank_gpu.py::test_eight_ranks_* with per-rank traces of everything that goes INTO and comes OUT of
every collective (DUDF_TEST_TRACE in tests/multirank_worker.py) and say, for a bad run, which (rank, step, quantity) was first wrong.

    python tools/multirank_loop.py [--runs 40] [--case sched] [--world 8] [--coll staggered] [--n-global 100000] [--keep DIR]

Checks per run (world W, against the 1-rank run of the same case made first):
  T  theta entering every step is bit-identical on all ranks;
  R  what every rank holds after a step's collectives = the sum over ranks of what they put in (1e-6 of the slice's max);
  P  post-collective buffers are bit-identical across ranks;
  L  every rank's LOCAL contribution agrees with the same rank's in the first run of the loop (1e-4: atomics noise x trajectory);
  H  loss curve against the 1-rank run (the test's own 1e-4 bar).
The parent never touches the GPU."""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def layer_of(lo, H=256, L=8):
    bounds, off = [(0, 4 * H, "layer0")], 4 * H
    for i in range(1, L):
        bounds.append((off, off + H * H + H, f"layer{i}")); off += H * H + H
    bounds.append((off, off + H + 1, f"layer{L}")); off += H + 1
    bounds.append((off, off + 4, "terms"))
    return [nm for a, b, nm in bounds if a <= lo < b][0]


SAVE = os.environ.get("DUDF_LOOP_SAVE")            # a directory: keep the offending slices / thetas of bad runs as .npz (1-5 MB each)


def pattern(got, want, lo, hi, H=256):
    """Which rows / columns of which hidden matrices of the slice [lo, hi) are off, and how (got vs want)."""
    out = []
    first_hidden = 4 * H
    if lo < first_hidden or (hi - lo) % (H * H + H):
        return out
    for j in range((hi - lo) // (H * H + H)):
        g = got[j * (H * H + H):][:H * H].reshape(H, H); w = want[j * (H * H + H):][:H * H].reshape(H, H)
        m = np.abs(g - w) > 1e-5 * np.abs(want).max()
        if not m.any():
            continue
        rows, cols = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))

        def runs(ix):
            cuts = np.flatnonzero(np.diff(ix) != 1)
            st = np.concatenate([[ix[0]], ix[cuts + 1]]); en = np.concatenate([ix[cuts], [ix[-1]]])
            return " ".join(f"{a}-{b}" for a, b in zip(st, en))
        sub_g, sub_w = g[np.ix_(rows, cols)], w[np.ix_(rows, cols)]
        ratio = sub_g[np.abs(sub_w) > 1e-3 * np.abs(sub_w).max()] / sub_w[np.abs(sub_w) > 1e-3 * np.abs(sub_w).max()]
        out.append(f"      matrix {j} of the slice (layer {(lo - first_hidden) // (H * H + H) + 1 + j}): {int(m.sum())} elements off; rows {runs(rows)}; cols {runs(cols)}; "
                   f"|got| max {np.abs(sub_g).max():.3e} |want| max {np.abs(sub_w).max():.3e}; got/want quartiles {np.array2string(np.percentile(ratio, [5, 25, 50, 75, 95]), precision=3)}; "
                   f"|got-want| max {np.abs(sub_g - sub_w).max():.3e}; bias row off: {bool((np.abs(got[j * (H * H + H) + H * H:][:H] - want[j * (H * H + H) + H * H:][:H]) > 1e-5 * np.abs(want).max()).any())}")
    return out


def analyse(tdir, world, golden, steps, log):
    tr = [np.load(os.path.join(tdir, f"trace_w{world}_r{r}.npz")) for r in range(world)]
    bad = []
    for t in range(steps):
        th = [x[f"s{t}_theta"] for x in tr]
        for r in range(1, world):
            if not np.array_equal(th[0], th[r]):
                d = np.flatnonzero(th[0] != th[r])
                blk = np.unique(d // 256)
                dv = np.abs(th[0] - th[r])[d]
                bad.append(f"T step {t}: theta of rank {r} != rank 0 at {d.size} elements, first {d[0]} ({layer_of(int(d[0]))}), last {d[-1]} ({layer_of(int(d[-1]))}), max |diff| {np.abs(th[0] - th[r]).max():.3e}; "
                           f"256-element blocks touched {blk.size}, block index mod 8 histogram {np.bincount(blk % 8, minlength=8).tolist()}, mod 16 {np.bincount(blk % 16, minlength=16).tolist()}; "
                           f"|diff| quantiles {np.array2string(np.percentile(dv, [1, 25, 50, 75, 99]), precision=3)}; fraction of a touched block that differs {d.size / (blk.size * 256):.3f}")
                if SAVE and t <= 1:
                    os.makedirs(SAVE, exist_ok=True)
                    np.savez_compressed(os.path.join(SAVE, f"theta_{len(os.listdir(SAVE))}_s{t}_r{r}.npz"), rank0=th[0], rankr=th[r], prev0=tr[0][f"s{max(t - 1, 0)}_theta"], post=tr[r][f"s{max(t - 1, 0)}_post"])
        pres = sorted(k for k in tr[0].files if k.startswith(f"s{t}_pre"))
        post = [x[f"s{t}_post"] for x in tr]
        for r in range(1, world):
            if not np.array_equal(post[0], post[r]):
                d = np.flatnonzero(post[0] != post[r])
                bad.append(f"P step {t}: post-collective buffer of rank {r} != rank 0 at {d.size} elements, first {d[0]} ({layer_of(int(d[0]))}), last {d[-1]} ({layer_of(int(d[-1]))})")
        for k in pres:
            parts = k.split("_")
            loc = [x[k].astype(np.float64) for x in tr]
            if parts[-1] == "stats":
                continue
            lo, hi = int(parts[-2]), int(parts[-1])
            tot = np.sum(loc, axis=0)
            for r in range(world):
                got = post[r][lo:hi]
                e = rel(got, tot)
                if e > 1e-6:
                    d = np.flatnonzero(np.abs(got - tot) > 1e-6 * np.abs(tot).max())
                    bad.append(f"R step {t} slice [{lo},{hi}) ({layer_of(lo)}..): rank {r} holds != sum of inputs, rel {e:.2e}, {d.size} elements, first at +{d[0]}, last at +{d[-1]}")
            for r in range(world):
                gk = (world, r, k)
                if gk in golden:
                    e = rel(loc[r], golden[gk])
                    if e > (1e-4 if t else 2e-5):
                        g = golden[gk]
                        d = np.flatnonzero(np.abs(loc[r] - g) > 1e-5 * np.abs(g).max())
                        bad.append(f"L step {t} slice [{lo},{hi}) ({layer_of(lo)}..): rank {r}'s LOCAL contribution differs from the first run's by {e:.2e} ({d.size} elements; first +{d[0]} last +{d[-1]})")
                        if not any(b.startswith("L") for b in bad[:-1]):          # the FIRST wrong local contribution of the run: its shape
                            bad.extend(pattern(loc[r], g, lo, hi))
                            if SAVE:
                                os.makedirs(SAVE, exist_ok=True)
                                np.savez_compressed(os.path.join(SAVE, f"bad_{len(os.listdir(SAVE))}_s{t}_r{r}_{lo}_{hi}.npz"), got=loc[r].astype(np.float32), want=g.astype(np.float32))
                else:
                    golden[gk] = loc[r]
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=40)
    ap.add_argument("--case", default="sched")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--coll", default="staggered")
    ap.add_argument("--n-global", type=int, default=100000)
    ap.add_argument("--ones", type=int, default=3, help="1-rank runs made first (they must agree with each other)")
    ap.add_argument("--variants", default="", help="';'-separated environment settings cycled run by run on the SAME box, e.g. "
                    "'GPU_MAX_HW_QUEUES=4;GPU_MAX_HW_QUEUES=2;GPU_MAX_HW_QUEUES=4 DUDF_TEST_OPTS=sweep_family=0,wgrad_family=1' (A/B/C interleaved: boxes differ, calls are not comparable)")
    a = ap.parse_args()
    import test_multirank_gpu as T
    os.environ["DUDF_TEST_NGLOBAL"] = str(a.n_global)
    steps = 4 if a.case == "sched" else 3
    work = tempfile.mkdtemp(prefix="mrloop_")
    ones = []
    os.environ.pop("DUDF_TEST_COLLECTIVES", None)
    for i in range(a.ones):
        out = os.path.join(work, f"one{i}.npz")
        d = os.path.join(work, f"one{i}"); os.makedirs(d)
        os.environ["DUDF_TEST_TRACE"] = d
        T._launch(1, ["engine", a.case, out], timeout=600)
        ones.append(np.load(out))
        if i:
            e = np.abs(ones[i]["hist"] - ones[0]["hist"]).max(axis=1) / np.abs(ones[0]["hist"]).max(axis=1)
            print(f"1-rank run {i} vs run 0: curve {np.array2string(e, precision=1)} theta {rel(ones[i]['theta'], ones[0]['theta']):.2e}", flush=True)
    one = ones[0]
    os.environ["DUDF_TEST_COLLECTIVES"] = a.coll
    golden, nbad = {}, 0
    # default: the configuration tests/test_multirank_gpu.py launches (one hardware queue per process); "GPU_MAX_HW_QUEUES=4" = the HIP
    # runtime's own default, with which 8 processes oversubscribe the chip's queue slots (round 6: 8 bad of 120 on the same box)
    variants = [v.strip() for v in a.variants.split(";") if v.strip()] or ["GPU_MAX_HW_QUEUES=1"]
    per = {v: [0, 0] for v in variants}
    base_env = dict(os.environ)
    for run in range(a.runs):
        var = variants[run % len(variants)]
        os.environ.clear(); os.environ.update(base_env)
        for kv in var.split():                                   # whitespace-separated NAME=VALUE settings of one variant
            if "=" in kv:
                k, v = kv.split("=", 1)
                os.environ[k] = v
        d = os.path.join(work, f"run{run}"); os.makedirs(d)
        os.environ["DUDF_TEST_TRACE"] = d
        out = os.path.join(work, f"w{run}.npz")
        T._launch(a.world, ["engine", a.case, out], timeout=600)
        r = np.load(out)
        e_h = np.abs(r["hist"] - one["hist"]).max(axis=1) / np.abs(one["hist"]).max(axis=1)
        bad = analyse(d, a.world, golden, steps, None)
        if e_h.max() > 1e-4:
            bad.append(f"H curve vs 1 rank {np.array2string(e_h, precision=1)}; per term at the first bad step: "
                       f"{np.array2string(np.abs(r['hist'] - one['hist'])[int(np.argmax(e_h > 1e-4))], precision=3)}")
        per[var][0] += 1; per[var][1] += 1 if bad else 0
        print(f"run {run} [{var}]: curve {np.array2string(e_h, precision=1)} dtheta0 {rel(r['dtheta0'], one['dtheta0']):.2e} {'OK' if not bad else 'BAD'}", flush=True)
        for b in bad[:40]:
            print("    " + b, flush=True)
        if bad:
            nbad += 1
        shutil.rmtree(d, ignore_errors=True)
    print(f"{nbad} bad of {a.runs} runs [{a.case} world {a.world} {a.coll}]; per variant: " + "; ".join(f"{v}: {b}/{n}" for v, (n, b) in per.items()), flush=True)
    shutil.rmtree(work, ignore_errors=True)
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
