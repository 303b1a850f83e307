#!/bin/bash
# A/B runs of bench.py under environment switches: bash tools/ab.sh "VAR=val VAR2=val" "VAR=val" ...   ("-" = no switches)
# prints ms/step and every MFMA kernel's launch time per configuration.  Extra bench flags through AB_FLAGS.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
i=0
for cfg in "$@"; do
  i=$((i+1))
  [ "$cfg" = "-" ] && cfg=""
  env $cfg python3 "$R/bench.py" --steps ${AB_STEPS:-40} --warmup 5 --no-cpu-baseline $AB_FLAGS > "$R/gpurun_out/ab_$i.json" 2>"$R/gpurun_out/ab_$i.err"
  python3 - "$R/gpurun_out/ab_$i.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["roofline"]["all_mfma_kernels"]
    print("[%s]" % sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "pts/s %.2fM" % (d["value"] / 1e6),
          {n.replace("sweep_", "").replace("wgrad_", "wg_"): round(k[n]["avg_ms"], 3) for n in k}, "loss %.3f" % d["final_loss"])
except Exception as e:
    print("[%s] FAILED" % sys.argv[2], e)
PY
done
