# coding: utf-8
"""Diagnosis of the beetle trajectory: which of {sampler inputs, dtheta noise, Adam} moves the
HIP curve away from the reference's.  Runs on the GPU box:  python tools/beetle_debug.py
  A  HIP loss_s1 + torch Adam on the ORACLE's host batches (the exact inputs of tests/golden/g5_beetle.npz)
  B  the same on the HIP sampler's batches
  C  the HIP TrainEngine (dudf_adam_step) on the oracle's batches
and, at step 0, dtheta against the fp64 oracle gradient (relative error, sign disagreements, launch-to-launch noise).
Uses oracle/ as the checker only (this is a test tool, not product code)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from diffudf_amd import mesh, synth, hip_ops  # noqa: E402
from diffudf_amd.engine import TrainEngine  # noqa: E402
from oracle import sampler_oracle as SO  # noqa: E402
from oracle import dudf_oracle as O  # noqa: E402


def main():
    from src.dataset import PointCloud
    from src.loss_functions import loss_s1
    from src.model import SIREN
    G = np.load(os.path.join(ROOT, "tests", "golden", "g5_beetle.npz"))
    hidden = list(G["hidden"]); steps = int(G["steps"]); bs = int(G["batch_size"])
    beetle = os.path.join(ROOT, "tests", "golden", "beetle")
    tri, pos, nrm = mesh.prepare(beetle, 100000, seed=123)
    n_on, n_off = int(bs * 0.333), int(bs * 0.666)
    n_far, n_near = n_off // 2, n_off - n_off // 2
    host = [SO.sample_batch(tri, pos, nrm, n_on, n_far, n_near, seed=123, step=t) for t in range(steps)]
    print("fixture batch0 reproduced on this host:", np.array_equal(host[0][0], G["batch0_x"]),
          np.array_equal(host[0][2], G["batch0_sdf"]))
    ds = PointCloud(beetle, bs, [0.333, 0.666], 1, device="cuda:0", seed=123)
    dev = []
    for t in range(steps):
        x, n_, s = ds.sample(t)
        dev.append((x.clone(), n_.clone(), s.clone()))
        dx = np.abs(x.cpu().numpy() - host[t][0]).max(); dn = np.abs(n_.cpu().numpy() - host[t][1]).max()
        dsd = np.abs(s.cpu().numpy() - host[t][2][:, 0])
        nb = int((s.cpu().numpy() != host[t][2][:, 0]).sum()); nxb = int((x.cpu().numpy() != host[t][0]).any(axis=1).sum())
        if t < 3 or dsd.max() > 2e-6:
            print(f"step {t}: sampler max|dx| {dx:.2e} ({nxb} rows differ) max|dn| {dn:.2e} max|dsdf| {dsd.max():.2e} "
                  f"({nb} of {len(dsd)} values differ), rel max {np.max(dsd / np.maximum(host[t][2][:, 0], 1e-12)):.2e}")

    def fresh():
        model = SIREN(3, 1, hidden, w0=30)
        sd = {}
        for i, (wt, b) in enumerate(synth.siren_params(hidden, seed=int(G["param_seed"]))):
            sd[f"net.{i}.0.weight"] = torch.from_numpy(wt); sd[f"net.{i}.0.bias"] = torch.from_numpy(b)
        model.load_state_dict(sd)
        return model.to("cuda:0")

    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
    for name, w in (("s1eik", [1e4, 1e4, 0.0, 1e3]), ("s1full", [1e4, 1e4, 1e4, 1e3])):
        ref = G[f"{name}_f64_hist"]; ref32 = G[f"{name}_f32_hist"]
        print(f"== {name}: reference fp32-vs-fp64 per step",
              np.array2string(np.abs(ref32 - ref).max(axis=1) / np.abs(ref).max(axis=1), precision=1))
        for tag in ("A host batches + torch Adam", "B HIP sampler + torch Adam"):
            model = fresh()
            opt = torch.optim.Adam(lr=1e-4, params=model.parameters())
            hist = []
            for t in range(steps):
                if tag[0] == "A":
                    x, n_, s = d(host[t][0])[None], d(host[t][1])[None], d(host[t][2])[None]
                else:
                    x, n_, s = dev[t][0][None], dev[t][1][None], dev[t][2][None, :, None]
                opt.zero_grad()
                loss = loss_s1(model, x, {"normals": n_, "sdf": s}, w, 100)
                total = torch.zeros((1, 1), device="cuda:0")
                for l in loss.values():
                    total += l
                total.backward()
                opt.step()
                hist.append([l.item() for l in loss.values()])
            hist = np.array(hist)
            per = np.abs(hist - ref).max(axis=1) / np.abs(ref).max(axis=1)
            th = model.flat_parameters().cpu().numpy()[G["sample"]]
            et = np.abs(th - G[f"{name}_f64_theta_sample"]).max() / np.abs(G[f"{name}_f64_theta_sample"]).max()
            print(f"{tag}: per-step {np.array2string(per, precision=1)}  theta err {et:.2e}")
        if name == "s1eik":
            theta = d(synth.flatten_params(synth.siren_params(hidden, seed=int(G["param_seed"])))).clone()
            eng = TrainEngine(hidden, theta)
            hist = []
            for t in range(steps):
                terms = eng.step(0, d(host[t][0]), d(host[t][1]), d(host[t][2][:, 0]), w, 100.0, 1e-4)
                hist.append(terms.cpu().numpy().copy())
            hist = np.array(hist)
            per = np.abs(hist - ref).max(axis=1) / np.abs(ref).max(axis=1)
            print(f"C host batches + TrainEngine: per-step {np.array2string(per, precision=1)}")

    # ---- step-0 gradient against the fp64 oracle on the fixture batch
    P32 = synth.siren_params(hidden, seed=int(G["param_seed"]))
    P64 = [(a.astype(np.float64), b.astype(np.float64)) for a, b in P32]
    x, n_, s = [a.astype(np.float64) for a in host[0]]
    w = [1e4, 1e4, 0.0, 1e3]
    t_ref, g_ref, _ = O.loss_and_grad("s1", P64, x, n_, s, w, 100.0)
    gr = np.concatenate([np.concatenate([a.reshape(-1), b.reshape(-1)]) for a, b in g_ref])
    cfg = hip_ops.make_cfg(hidden)
    th = d(synth.flatten_params(P32))
    ws = hip_ops.workspace_for(cfg, len(x), "cuda:0")
    outs = []
    for rep in range(3):
        terms = hip_ops.loss_forward(cfg, 0, th, d(host[0][0]), d(host[0][1]), d(host[0][2][:, 0]), len(x), w, 100.0, ws)
        dth = hip_ops.loss_backward(cfg, 0, th, d(host[0][0]), d(host[0][1]), d(host[0][2][:, 0]), len(x), w, 100.0,
                                    torch.ones(4, device="cuda:0"), None, ws)
        outs.append(dth.cpu().numpy().astype(np.float64))
    gh = outs[0]
    print("step0 terms rel err", np.abs(terms.cpu().numpy() - np.array([float(v) for v in t_ref.values()])).max()
          / max(abs(float(v)) for v in t_ref.values()))
    print("step0 dtheta: max-norm rel err %.2e; launch-to-launch %.2e; |g| quantiles" % (
        np.abs(gh - gr).max() / np.abs(gr).max(), np.abs(outs[1] - outs[0]).max() / np.abs(gr).max()),
        np.quantile(np.abs(gr), [0.001, 0.01, 0.1, 0.5, 0.9, 0.999]))
    flips = np.sign(gh) != np.sign(gr)
    print("sign disagreements with fp64:", int(flips.sum()), "of", gr.size, "; largest |g_ref| among them %.3e" %
          (np.abs(gr[flips]).max() if flips.any() else 0.0), "; abs err quantiles",
          np.quantile(np.abs(gh - gr), [0.5, 0.9, 0.99, 1.0]))
    # Adam's first update is lr*g/(|g|+eps): how many parameters move differently by more than 1% of lr
    upd_h = gh / (np.abs(gh) + 1e-8); upd_r = gr / (np.abs(gr) + 1e-8)
    print("first-update differences > 0.01 lr:", int((np.abs(upd_h - upd_r) > 0.01).sum()))
    # the reference's fp32 gradient noise, for scale: torch fp32 autograd of the oracle formulation
    P32t = [(a.astype(np.float32), b.astype(np.float32)) for a, b in P32]
    try:
        _, g32, _ = O.loss_and_grad("s1", P32t, host[0][0], host[0][1], host[0][2], w, 100.0)
        g32 = np.concatenate([np.concatenate([a.reshape(-1), b.reshape(-1)]) for a, b in g32]).astype(np.float64)
        f32 = np.sign(g32) != np.sign(gr)
        print("oracle run in fp32: rel err %.2e, sign disagreements %d, abs err quantiles" % (
            np.abs(g32 - gr).max() / np.abs(gr).max(), int(f32.sum())), np.quantile(np.abs(g32 - gr), [0.5, 0.9, 0.99, 1.0]))
    except Exception as e:  # noqa: BLE001
        print("fp32 oracle run failed:", e)


if __name__ == "__main__":
    main()
