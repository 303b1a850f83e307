#!/bin/bash
# A/B of the weight-gradient GEMM variants (DUDF_WGRAD_VAR bit 0: conflict-free producer lanes, bit 1: interleaved split)
# on the headline workload; prints ms/step and the wgrad_hidden launch time per variant.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
for v in 0 1 2 3 3 0; do
  DUDF_WGRAD_VAR=$v python3 "$R/bench.py" --steps 40 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/ab_wgrad_$v.json" 2>/dev/null
  python3 - "$R/gpurun_out/ab_wgrad_$v.json" $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["all_mfma_kernels"]
print("VAR", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "pts/s %.2fM" % (d["value"] / 1e6),
      {n: k[n]["avg_ms"] for n in k}, "final_loss", d["final_loss"])
PY
done
