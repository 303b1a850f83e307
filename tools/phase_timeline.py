#!/usr/bin/env python
# coding: utf-8
"""Timeline of one workgroup's k-block steps in the four bf16 sweeps (timing experiment; needs a library built with
-DDUDF_SWEEP_DBG=128, see tools/build_dbg.sh):   DUDF_LIB=dbg/libdudf_st.so python tools/phase_timeline.py
Prints, per sweep and k-block of layer 3, the s_memtime stamps of waves 0 and 4 (SIMD partners) relative to wave 0's
step start, in shader-clock cycles (s_memtime): top | feed done | tail start | tail end | MFMAs issued | DMA wait done."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diffudf_amd import _lib, hip_ops, synth
from diffudf_amd.engine import TrainEngine

lib = _lib.load()
hid = [256] * 8
dev = torch.device("cuda:0")
theta = torch.from_numpy(synth.flatten_params(synth.siren_params(hid, seed=123))).to(dev)
x, nrm, sdf = [torch.from_numpy(a).to(dev) for a in synth.training_batch(100000, seed=123)]
eng = TrainEngine(hid, theta)
for _ in range(5):
    eng.step(hip_ops.LOSS_S1, x, nrm, sdf.reshape(-1), [1e4, 1e4, 0.0, 1e3], 100.0, lr=1e-4, n_global=100000, n_hess=0)
torch.cuda.synchronize()
buf = np.zeros((4, 8, 8, 8), dtype=np.uint64)
names = ["fwd", "rev", "adj_fwd", "adj_rev"]
try:
    fn = lib.dudf_dbg_stamps
    fn.argtypes = [ctypes.c_void_p]
    rc = fn(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    have_sweeps = True
except AttributeError:
    have_sweeps = False
for s in range(4 if have_sweeps else 0):
    t0 = int(buf[s, 0, 0, 0])
    print("== %s (cycles from wave 0's first step top; layer 3)" % names[s])
    for kb in range(8):
        for w in ((0, 4, 1, 5) if os.environ.get("DUDF_TL_ALL") is None else range(8)):
            st = [(int(v) - t0) * 10 if v else -1 for v in buf[s, w, kb]]
            st = [v // 10 if v >= 0 else -1 for v in st]
            print("  kb %d wave %d: top %6d feed %6d | tail %6d..%6d | mfma issued %6d | wait done %6d | mid feed %6d..%6d" % (kb, w, *st))
    step = (int(buf[s, 0, 7, 0]) - t0) * 10 / 7.0
    print("  mean step %.0f cycles" % (step / 10))

# weight-gradient kernel (library built with -DDUDF_WGRAD_DBG=32): last steady-state stage of one workgroup
if hasattr(lib, "dudf_dbg_wstamps") or True:
    try:
        fnw = lib.dudf_dbg_wstamps
        fnw.argtypes = [ctypes.c_void_p]
        wb = np.zeros((8, 12), dtype=np.uint64)
        assert fnw(wb.ctypes.data_as(ctypes.c_void_p)) == 0
        t0 = int(wb[:, 0].min())
        print("== wgrad_hidden, one stage (cycles): 0 start | 1 A frags issued (past the first poll) | 3 5 7 9 MFMA groups issued | 2 reads published | 4 buffer free | 6 loads landed | 8 split written | 10 loads issued, image published")
        for w in range(8):
            print("  wave %d: " % w + " ".join("%d:%6d" % (i, int(wb[w][i]) - t0) for i in (0, 1, 3, 5, 7, 9, 2, 4, 6, 8, 10)))
    except AttributeError:
        pass

# sixteen-wave experiment (DUDF_WGRAD_W16=1, -DDUDF_WGRAD_DBG=32): stage 40 of one workgroup, six stamps per wave
if os.environ.get("DUDF_WGRAD_W16") == "1":
    fnw = lib.dudf_dbg_wstamps
    fnw.argtypes = [ctypes.c_void_p]
    wb = np.zeros(96, dtype=np.uint64)
    assert fnw(wb.ctypes.data_as(ctypes.c_void_p)) == 0
    wb = wb.reshape(16, 6); t0 = int(wb[:, 0].min())
    print("== W16 stage: start | fetch (+ MFMAs if first) | raw landed | split + DMA issued | MFMAs (if second) | barrier passed")
    for w in range(16):
        print("  wave %2d (simd %d, %s): " % (w, w % 4, "M first" if (w >> 2) & 1 else "S first") + " ".join("%6d" % (int(v) - t0) for v in wb[w]))
