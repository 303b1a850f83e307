# coding: utf-8
"""CPU (needs hipcc, no GPU): the hand-counted `s_waitcnt vmcnt(N)` behind the inline-asm LDS-DMA of the H=256
sweep kernels must never be larger than the number of vector-memory instructions hipcc actually placed between
the last DMA piece and the wait — otherwise a weight chunk could be read before it has landed."""
import os
import shutil
import pytest

import sys, os as _os
sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from isa_contract import analyse, emit_asm

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_dma_wait_counts(tmp_path):
    asm = str(tmp_path / "sweep.s")
    emit_asm(os.path.join(REPO, "diffudf_amd", "csrc", "dudf_sweep.hip"), asm)
    res = analyse(asm)
    assert len(res) >= 7
    for key, v in res.items():
        assert v["pairs"], key
        for counted, n in v["pairs"]:
            assert counted >= n, f"sweep_kernel<256,{key[0]},{key[1]}>: vmcnt({n}) but only {counted} younger ops"
    # the training variants must not spill
    for key in ((0, 3), (1, 1), (2, 0), (3, 1), (3, 0)):
        assert res[key]["scratch"] == 0, key
