# coding: utf-8
"""`python bench.py --gpus N` as the driver types it (VERDICT r03 missing #1): without a torchrun environment the file
starts torch.distributed.run on itself as a CHILD process, before anything has touched the GPU, and hands back the
child's exit code.  CPU part: what would be started, and that the child path really runs (it ends with the "needs an
MI355X" refusal of every rank here).  GPU part (one GPU: the ranks share cuda:0 over gloo): the whole path, one JSON line."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def test_parent_spawns_torchrun_without_touching_the_gpu():
    env = dict(os.environ, DUDF_BENCH_DRY_LAUNCH="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2"], env=env, cwd=REPO,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["cuda_initialized"] is False
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]      # the ranks get the caller's own flags
    src = open(BENCH).read()
    assert "os.exec" not in src and "execv" not in src                          # a child, never a re-exec


def test_child_exit_code_comes_back():
    """No GPU here: every rank refuses ("needs an MI355X"), torchrun fails, and the parent returns that failure instead of
    the round-3 "launch multi-GPU runs with ..." message."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DUDF_BENCH_DRY_LAUNCH"):
        env.pop(k, None)
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env,
                       cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "launch multi-GPU runs with" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("gpus,points", [(2, 20000), (8, 12500)])
def test_bench_gpus_n_as_typed_on_one_gpu(gpus, points):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DUDF_BENCH_DRY_LAUNCH"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-config3", "--points", str(points), "--share-device"], env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 3 and d["scaling"] == "weak" and d["share_device"] is True
    assert d["config"]["global_batch"] == gpus * points and d["value"] > 0
    assert d["collectives"] in ("staggered", "fused")
    ph = d["phases_ms"]
    assert ph and all(v >= 0 for v in ph.values()) and any(k.startswith("wait") or "allreduce" in k or "coll" in k for k in ph), ph
