# Runs the three forward-sweep variants (no stores / cos only / h and cos) on the headline workload, for counter passes.
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffudf_amd import hip_ops, synth
hidden = [256] * 8
theta = torch.from_numpy(synth.flatten_params(synth.siren_params(hidden, seed=123))).cuda()
x, nrm, sdf = [torch.from_numpy(a).cuda() for a in synth.training_batch(100000, seed=5, step=0)]
cfg = hip_ops.make_cfg(hidden)
ws = hip_ops.workspace_for(cfg, 100000, theta.device)
for _ in range(6):
    hip_ops.query(cfg, theta, x, want_grad=False)           # sweep_bf16_kernel<256,0,0>
    hip_ops.query(cfg, theta, x, want_grad=True)            # <256,0,2> + <256,1,0>
    hip_ops.loss_forward(cfg, 0, theta, x, nrm, sdf, 100000, [1e4, 1e4, 0.0, 1e3], 100.0, ws)   # <256,0,3> + <256,1,1>
torch.cuda.synchronize()
