#!/bin/bash
# One parametrised GPU-call runner (replaces round 5's per-call scripts):  gpurun -- 'bash tools/gpu_r06.sh <tag> "<cmd>" ["<cmd>" ...]'
# Every command's stdout+stderr goes to gpurun_out/r06_<tag>_<i>.txt (first line = the command), its tail to the call's own output.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
tag=$1; shift
i=0
for cmd in "$@"; do
    i=$((i + 1))
    f=$O/r06_${tag}_$i.txt
    echo "\$ $cmd" > $f
    bash -c "$cmd" >> $f 2>&1
    echo "rc $?" >> $f
    echo "=== [$tag $i] $cmd"; tail -${TAILN:-25} $f
done
