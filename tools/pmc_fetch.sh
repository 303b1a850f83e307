#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel of one bench configuration: bash tools/pmc_fetch.sh <tag> [bench flags]
# (environment switches such as DUDF_SPLIT are exported by the caller: the program behind `--` must be python3 itself)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pf_${TAG}_$C -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 "$@" > $R/gpurun_out/pf_${TAG}_$C.log 2>&1
done
python3 - "$R" "$TAG" <<'PY'
import csv, glob, sys, collections
R, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{R}/gpurun_out/pf_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            agg[name.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    rd = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) * 2048
    wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) * 1024
    if rd + wr > 5e7:
        print(f"{tag} {k:48s} read {rd/1e9:.3f} GB  write {wr/1e9:.3f} GB")
PY
