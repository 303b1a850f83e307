#!/bin/bash
# The record run of a build (one per round, on ONE box):  gpurun -- 'bash tools/gpu_record.sh r06'
# default bench line, rocprofv3 kernel stats of the same command, the reference's recipe (mesh and cloud-only), step sizes, speed-up against torch
# on the same GPU, PMC passes (headline and config 3).  Everything lands in gpurun_out/<tag>_*; summaries are then copied into profiles/ by hand
# (python profiles/summarize_pmc.py gpurun_out/pmc_core_[0-9]* --tag <tag>; ... pmc_core_512_* --tag <tag>_8x512 --traffic hbm_traffic_8x512.json).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T=${1:-rec}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/pmc_core_* $O/pmc_core_512_*
timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "bench rc $?"; python tools/show_bench.py $O/${T}_bench.json | grep -E "value|config3|cpu"
bash tools/prof_stats.sh $T | head -12
python - <<'PY'
import json
c = json.load(open("configs/train_beetle.json")); c["onlyPCloud"] = True; c["experiment_name"] = "cloud"
json.dump(c, open("/tmp/beetle_cloud.json", "w"))
PY
for i in 1 2; do
timeout 300 python train.py configs/train_beetle.json 0 2>&1 | grep -E "training time|Error|error" | sed 's/^/beetle recipe (graph): /' | tee -a $O/${T}_recipe.txt
rm -rf results/beetle
done
timeout 300 python train.py /tmp/beetle_cloud.json 0 2>&1 | grep -E "training time|Error|error" | sed 's/^/beetle recipe, onlyPCloud (graph): /' | tee -a $O/${T}_recipe.txt
rm -rf results/beetle
timeout 300 python tools/step_sizes.py 2>&1 | tee $O/${T}_sizes.txt
timeout 600 python -m pytest tests/test_speedup_vs_unfused_torch.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "ratio|passed|failed" | tee $O/${T}_speedup.txt
bash tools/pmc_passes.sh core 2>&1 | tail -7
bash tools/pmc_passes.sh core "--hidden 512 --points 125000" _512 2>&1 | tail -7
