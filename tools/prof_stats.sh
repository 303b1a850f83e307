#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench command; copies the kernel stats CSV to gpurun_out/<tag>_rocprofv3_kernel_stats.csv
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/${TAG}_stats" -o st --output-format csv -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-config3 > "$R/gpurun_out/${TAG}_stats_bench.json" 2> "$R/gpurun_out/${TAG}_stats.err"
f=$(find "$R/gpurun_out/${TAG}_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$R/gpurun_out/${TAG}_rocprofv3_kernel_stats.csv" && head -12 "$f"
