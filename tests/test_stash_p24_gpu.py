# coding: utf-8
"""GPU: the stash formats (dudf_stash_mode, a bit mask).

  mask 6, the DEFAULT of 256- and 512-wide networks: R and E — the two arrays only the adjoint sweeps read — hold fp32 values rounded to 24
          bits (2^-17 relative), and C = cos(w0 z_l) is 24-bit fixed point on a 2^-22 grid (absolute error 2^-23, the size of the
          sin/cos polynomials' own error); all three tile-major: 15 instead of 17 array-layer units, every tolerance unchanged (the
          whole GPU suite runs in this mode), the 12-step beetle trajectory at 3e-7 .. 5e-7 like fp32;
  mask 7, opt-in (DUDF_STASH=17p24): S, Q, A, Z as 24-bit floats as well, the weight-gradient GEMM reading them through transposed
          LDS fragment reads: 12.75 units.  Built, measured, and NOT the default.  What this file pins for it:
  * every single-step tolerance of tests/test_hip_parity.py and tests/test_full_size_oracle_gpu.py holds unchanged in that
    mode (terms 1e-5, d(theta) 1e-4 / 5e-4 with the Hessian term, stash columns 5e-5 / 2e-4) — the kernels are right;
  * the 12-step beetle trajectory does NOT hold the north star's 1e-4: Adam divides every gradient component by its own
    magnitude, so the components that sit at the noise floor flip sign, and a floor 128 times higher (2^-17 against fp32's
    2^-24) moves the loss curve by 1e-4 .. 4e-4 within 12 steps (measured; fp32 stash and mask 6: 3e-7).  That is why mask 7
    stays opt-in — the test asserts the drift stays of that order so that the record in DESIGN.md §6 remains true.
Each case runs in a child process: the stash format is chosen when the library first answers dudf_stash_mode."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_p24(args, timeout=900):
    env = dict(os.environ, DUDF_STASH="17p24")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-s"] + args, cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=timeout)
    return r


def test_stash_modes_are_selected():
    code = ("import ctypes; from diffudf_amd import _lib; lib = _lib.load(); "
            "print(lib.dudf_stash_mode(ctypes.byref(_lib.NetCfg(3, 8, 256, 30.0))), lib.dudf_stash_mode(ctypes.byref(_lib.NetCfg(3, 8, 512, 30.0))))")
    env = dict(os.environ); env.pop("DUDF_STASH", None)
    assert subprocess.check_output([sys.executable, "-c", code], cwd=REPO, env=env, text=True).split() == ["6", "6"]   # default: R, E, C — at 256 and 512
    env["DUDF_STASH"] = "17"
    assert subprocess.check_output([sys.executable, "-c", code], cwd=REPO, env=env, text=True).split() == ["0", "0"]
    env["DUDF_STASH"] = "17p24"                          # (512-wide layers relay S, Q, A, Z through the stash: those stay fp32)
    assert subprocess.check_output([sys.executable, "-c", code], cwd=REPO, env=env, text=True).split() == ["7", "6"]
    env["DUDF_WGRAD"] = "f32"                            # a weight-gradient kernel that reads fp32 rows: its operands stay fp32
    assert subprocess.check_output([sys.executable, "-c", code], cwd=REPO, env=env, text=True).split() == ["6", "6"]
    env["DUDF_SWEEP"] = "f32"                            # ... and sweeps that cannot write the 24-bit arrays: fp32 stash
    assert subprocess.check_output([sys.executable, "-c", code], cwd=REPO, env=env, text=True).split() == ["0", "0"]


def test_p24_single_step_parity_holds_every_tolerance():
    r = run_p24(["tests/test_hip_parity.py", "tests/test_full_size_oracle_gpu.py", "tests/test_full_size_properties_gpu.py",
                 "-k", "not f32_and_bf16x6 and not pair_launch"])
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    print("\n".join(ln for ln in r.stdout.splitlines() if ln.startswith("full ")))


def test_p24_beetle_drift_is_why_it_is_not_the_default():
    r = run_p24(["tests/test_beetle_gpu.py", "-k", "s1eik and fixture_batches"])
    out = r.stdout
    line = [ln for ln in out.splitlines() if "per-step curve err" in ln]
    assert line, out[-2000:]
    import re
    # "... per-step curve err [a b c ...]; theta err ..." — the list may wrap over lines
    txt = out[out.index("per-step curve err"):]
    vals = [float(v) for v in re.findall(r"[0-9.]+e[-+][0-9]+", txt[:txt.index("]")])]
    print("beetle, 24-bit stash: per-step curve error", vals)
    assert len(vals) == 12 and vals[0] < 1e-6           # the first step (no update yet) is exact to fp32
    assert 2e-5 < max(vals) < 3e-3                      # ... and the trajectory leaves the 1e-4 bar: the reason for the default


def test_fp32_stash_still_runs():
    """DUDF_STASH=17 (every array fp32, rounds 1-3) stays a supported mode: the single-step parity tests in it."""
    env = dict(os.environ, DUDF_STASH="17")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_hip_parity.py", "-k",
                        "not f32_and_bf16x6 and not pair_launch"], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
