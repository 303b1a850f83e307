# coding: utf-8
"""GPU: edge shapes through the training path in every stash format (tools/stress_modes.py): 13 networks — one / 17 / skewed-stride
column counts, all-quad batches, 2 ... 34 hidden layers, 256 and 512 wide, `loss_s1` with and without the Hessian term and
`loss_s2` — must give the same terms and d(theta) with the fp32 stash (DUDF_STASH=17), the default (R, E as 24-bit floats, C as
24-bit fixed point) and the opt-in all-24-bit stash.  The formats are chosen when the library first answers dudf_stash_mode,
hence one child process per format.  (HIP against HIP: the oracle comparisons live in test_hip_parity.py and
test_full_size_oracle_gpu.py; this file pins the shapes those do not visit.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(REPO, "tools", "stress_modes.py")


def test_edge_shapes_agree_across_stash_formats(tmp_path):
    outs = {}
    for tag, stash in (("fp32", "17"), ("default", None), ("all24", "17p24")):
        env = dict(os.environ)
        env.pop("DUDF_STASH", None)
        if stash:
            env["DUDF_STASH"] = stash
        outs[tag] = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, TOOL, outs[tag]], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        modes = {ln.split("mode")[1].split()[0] for ln in r.stdout.splitlines() if ln.startswith("case")}
        assert modes <= {{"fp32": "0", "default": "6", "all24": "7"}[tag], "0", "6"}, (tag, modes)   # (deep / 512-wide cases fall back)
    for other in ("default", "all24"):
        r = subprocess.run([sys.executable, TOOL, "--compare", outs["fp32"], outs[other]], cwd=REPO, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        print(other, r.stdout.strip().splitlines()[-1])
