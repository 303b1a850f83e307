import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from diffudf_amd import hip_ops as hip, synth
hidden = [512] * 3; n = 300; seed = 9
P = synth.siren_params(hidden, seed=seed, dtype=np.float64)
theta = synth.flatten_params([(w.astype(np.float32), b.astype(np.float32)) for w, b in P])
x, nrm, sdf = synth.training_batch(n, seed=seed + 1)
n_on = int((sdf.reshape(-1) == 0).sum())
cfg = hip.make_cfg(hidden)
print("mode", hip.stash_mode(cfg), "n_on", n_on, flush=True)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
th, xd, nd, sd = dev(theta), dev(x), dev(nrm), dev(sdf.reshape(-1))
for nh in (0, n_on):
    ws = hip.workspace_for(cfg, n, "cuda", n_hess=nh) if nh else hip.workspace_for(cfg, n, "cuda")
    kw = {"n_hess": nh} if nh else {}
    W = [1e4, 1e4, 1e4 if nh else 0.0, 1e3]
    t = hip.loss_forward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, W, 100.0, ws, **kw); torch.cuda.synchronize()
    print("forward ok", nh, t.cpu().numpy(), flush=True)
    d = hip.loss_backward(cfg, hip.LOSS_S1, th, xd, nd, sd, n, W, 100.0, torch.ones(4, device="cuda"), None, ws, **kw); torch.cuda.synchronize()
    print("backward ok", nh, float(d.abs().max()), flush=True)
