#!/usr/bin/env python
# coding: utf-8
"""Per-layer phases of the 512-wide sweep kernel (timing experiment; library built with -DDUDF_SWEEP_DBG=128):
    bash tools/build_dbg.sh stw sweep_bf16 "-DDUDF_SWEEP_DBG=128"; DUDF_LIB=dbg/libdudf_stw.so python tools/phase_timeline_wide.py
k-loop / tail burst / drain + barrier, in shader-clock cycles, for waves 0 and 4 of two workgroups (8x512, 125 000 points)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffudf_amd import _lib, hip_ops, synth
from diffudf_amd.engine import TrainEngine
lib=_lib.load()
hid=[512]*8
theta=torch.from_numpy(synth.flatten_params(synth.siren_params(hid,seed=123))).cuda()
x,nrm,sdf=[torch.from_numpy(a).cuda() for a in synth.training_batch(125000,seed=123)]
eng=TrainEngine(hid,theta)
for _ in range(3):
    eng.step(hip_ops.LOSS_S1,x,nrm,sdf.reshape(-1),[1e4,1e4,0.0,1e3],100.0,lr=1e-4,n_global=125000,n_hess=0)
torch.cuda.synchronize()
buf=np.zeros((4,8,8,8),dtype=np.uint64)
fn=lib.dudf_dbg_stamps; fn.argtypes=[ctypes.c_void_p]; assert fn(buf.ctypes.data_as(ctypes.c_void_p))==0
for s,name in enumerate(["fwd","rev","adj_fwd","adj_rev"]):
    print("==",name,"(last pass) per layer: k-loop cycles | burst cycles | drain+barrier cycles   [wg100 w0 | wg100 w4 | wg101 w0 | wg101 w4]; start offsets vs wg100 w0")
    for j in range(7):
        row=[]
        for k in range(4):
            t=[int(v) for v in buf[s,k,j,:4]]
            row.append("%6d %6d %5d @%7d"%(t[1]-t[0],t[2]-t[1],t[3]-t[2],t[0]-int(buf[s,0,j,0])))
        print("  layer",j," | ".join(row))
