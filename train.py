#!/usr/bin/env python
# coding: utf-8
"""Training entry point with the reference's CLI and config schema:

    python train.py <config.json> <device-index>                       (reference train.py:450-467)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py <config.json> 0

`setup_train(parameter_dict, cuda_device)` and `train_model_tanh` / `train_model_siren` keep the reference
contracts (reference train.py:285-448, :146-283, :23-143): same config keys and defaults, the same LR /
phase schedule (warm-up LR, lr_s1 from `warmup_epochs`, loss_s2 + cosine LR from `s1_epochs`), the same
`results/<exp>/{models,summaries,reconstructions}` layout, `losses.csv` (sep=';'), `params.json`,
`model_best.pth` / `model_current.pth` / `model_final.pth` with reference state_dict keys.

What runs underneath is the MI355X path: the loss dict comes from the fused HIP kernels
(diffudf_amd.loss_functions), the optimizer is diffudf_amd.optim.Adam — torch.optim.Adam whose step() is ONE launch over the
flat parameter / gradient buffers the model's tensors are views of —, and under torch.distributed every rank takes its stratified share of the batch and the gradients
are all-reduced over RCCL before the step.

After training (reference train.py:403-446): the field slice of the best model (`generate_df`, always, as in the
reference) and, when `resolution != 0`, its meshes (`generate_mc` algorithm 'both': CAP-UDF on the device and MeshUDF's
marching cubes in host C++; the latter needs the Lewiner tables — config key `luts_path`, $DUDF_MESHUDF_LUTS or the
reference's table module on sys.path — and is skipped with a message when they are not there) are written to
`reconstructions/`.  gt_mode 'siren' trains, but its post-training artefacts (skimage's SDF marching cubes) are outside
this build: a message says so.  tensorboard is optional.
"""
import argparse
import copy
import json
import os
import os.path as osp
import random
import time
import weakref

import numpy as np
import torch

from diffudf_amd import hip_ops
from diffudf_amd.dataset import PointCloud, SyntheticPointCloud
from diffudf_amd.loss_functions import loss_siren, loss_s1, loss_s2
from diffudf_amd.model import SIREN
from diffudf_amd.optim import Adam
from diffudf_amd.util import create_output_paths, load_experiment_parameters

try:                                                  # optional, as in many deployments
    from torch.utils.tensorboard import SummaryWriter
except Exception:                                     # pragma: no cover
    SummaryWriter = None


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


def _dist():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def _zero_flat_grad(model):
    """`optim.zero_grad()` for the flat layout: every p.grad is a view of ONE buffer [dtheta | 4 loss terms], zeroed
    by one memset; backward() accumulates into the views in place, and that buffer is what gets all-reduced — no
    per-step torch.cat / copy-back."""
    flat = getattr(model, "_dudf_flat_grad", None)
    theta = model.flat_parameters()
    views = getattr(model, "_dudf_flat_grad_views", None)
    if flat is None or views is None or flat.device != theta.device or flat.numel() != theta.numel() + 4:
        flat = model._dudf_flat_grad = torch.zeros(theta.numel() + 4, dtype=torch.float32, device=theta.device)
        views = model._dudf_flat_grad_views = model.split_flat(flat[:theta.numel()])
        model._dudf_flat_grad_sig = [(v.data_ptr(), v.stride()) for v in views]
        # this loop only ever calls train_loss.backward(): the fused loss may write d(theta) straight into the flat buffer
        # (diffudf_amd/loss_functions.py::_FusedLoss.backward; torch.autograd.grad / hooks would need the flag off)
        model.dudf_direct_grad = True
    else:
        flat.zero_()
    for p, v, (ptr, _) in zip(model.parameters(), views, model._dudf_flat_grad_sig):
        if p.grad is None or p.grad.data_ptr() != ptr:
            p.grad = v
    return flat


def _allreduce_step(flat, vals, terms_are_global):
    """Every rank holds its share sum_local/n_global of the gradient and of each loss term: ONE all-reduce(sum) of the
    flat [dtheta | terms] buffer.  loss_s2's terms are already global (its statistics were all-reduced between the
    forward and the backward, every rank computed identical terms), so they stay out of the sum."""
    if not _dist() or torch.distributed.get_world_size() == 1:
        return vals
    n, k = vals.numel(), flat.numel() - 4
    if not terms_are_global:
        flat[k:k + n] = vals
    torch.distributed.all_reduce(flat)
    return vals if terms_are_global else flat[k:k + n].clone()


def _is_main():
    return (not _dist()) or torch.distributed.get_rank() == 0


def _train(dataset, model, device, config, schedule):
    """Shared epoch loop of train_model_tanh / train_model_siren (reference train.py:172-283 / :49-143)."""
    epochs = config["epochs"]
    epochs_til_checkpoint = config.get("epochs_to_checkpoint", 0)
    log_path = config["log_path"]
    optim = config["optimizer"]
    model.to(device)
    summary_path = osp.join(log_path, 'summaries')
    os.makedirs(summary_path, exist_ok=True)
    writer = SummaryWriter(summary_path) if (SummaryWriter is not None and _is_main()) else _NullWriter()
    if _dist():
        model.dudf_n_global = dataset.n_global        # terms/grads are shares of the GLOBAL mean
        model.dudf_allreduce = lambda t: torch.distributed.all_reduce(t)

    losses = dict()
    best_loss = np.inf
    best_weights = None
    recon_time = 0
    keys = list(model.state_dict().keys())

    # The reference reads every loss term back right after the step (five .item() syncs, reference train.py:212-224), so
    # the GPU idles while Python prepares the next step.  Here an epoch's numbers are read back ONE EPOCH LATER: its per-step
    # terms go to a pinned host buffer behind an event, and the parameters it ended with are kept in one of two device
    # snapshots (1.8 MB, a device-to-device copy) — so the bookkeeping below (losses.csv, the printed line, best / current /
    # periodic checkpoints) sees exactly the values and weights of ITS epoch, in the same order, while the next epoch's
    # kernels are already queued.
    snaps = [torch.empty_like(model.flat_parameters()) for _ in range(2)]
    best_flat = torch.empty_like(model.flat_parameters())      # the best epoch's parameters: ONE 1.8 MB copy when the loss improves
    have_best = False                                          # (18 per-tensor clones every improving epoch were 3 % of a stage-1 step)

    def snapshot_state(flat):
        return {k: v.clone() for k, v in zip(keys, model.split_flat(flat))}

    def finish(rec):
        """Epoch bookkeeping of reference train.py:226-283 for the epoch recorded in `rec`."""
        nonlocal best_loss, best_weights, recon_time, have_best
        epoch, names, host, event, snap, lr_now = rec
        event.synchronize()
        running_loss = dict()
        for vals in host.tolist():                     # one row per step of the epoch
            for it, v in zip(names, vals):
                running_loss[it] = running_loss.get(it, 0.0) + v
            writer.add_scalar("train_loss", sum(vals), epoch)
        bad = [it for it, l in running_loss.items() if l != l]
        if bad:                                        # NaN terms: a wrong gt['n_on_surface'] hint, or a diverged run — never train on
            raise RuntimeError(f"epoch {epoch}: loss term(s) {bad} are NaN; the parameters have been updated with NaN gradients since "
                               "(the bookkeeping runs one epoch late) — restart from the last checkpoint")
        for it, l in running_loss.items():
            if it not in losses:
                losses[it] = [0.] * epochs
            losses[it][epoch] = l
            writer.add_scalar(it, l, epoch)
        epoch_loss = sum(running_loss.values()) / dataset.batchesPerEpoch
        if _is_main():
            print(f"Epoch: {epoch} - Loss: {epoch_loss} - Learning Rate: {lr_now:.3e}")
        start_rtime = time.time()
        if _is_main():
            if epoch_loss < best_loss:
                best_loss = epoch_loss
                best_flat.copy_(snap); have_best = True
                if config.get("save_every_epoch", True):
                    torch.save(snapshot_state(best_flat), osp.join(log_path, "models", "model_best.pth"))
            if epoch and epochs_til_checkpoint and (not epoch % epochs_til_checkpoint):
                print(f"Saving model for epoch {epoch}")
                ckpt = osp.join(log_path, "models", f"model_{epoch}.pth")
                torch.save(snapshot_state(snap), ckpt)
                # ... and its mesh, as reference train.py:253-268 (generate_mc, algorithm 'both', at every periodic checkpoint);
                # from the checkpoint FILE: the live parameters are already an epoch further (bookkeeping runs one epoch late)
                if config.get("gt_mode") == 'tanh' and config.get("resolution", 256) and config.get("network"):
                    print("Generating mesh")
                    from generate_mc import generate_mc
                    net = config["network"]
                    os.makedirs(osp.join(log_path, "reconstructions"), exist_ok=True)
                    generate_mc(model=None, gt_mode=config["gt_mode"], device=int(device.index or 0), N=config.get('resolution', 256),
                                output_path=osp.join(log_path, "reconstructions", f'mc_mesh_{epoch}.obj'), alpha=config['alpha'],
                                algorithm='both', from_file={'w0': net["w0"], 'model_path': ckpt,
                                                             'hidden_layer_nodes': net["hidden_layer_nodes"],
                                                             'activation': net.get('activation', 'sine'),
                                                             'ww': net.get('ww')},           # the hidden layers' own frequency, if the config has one (reference train.py:322)
                                luts=config.get("luts_path"))
            elif config.get("save_every_epoch", True):
                torch.save(snapshot_state(snap), osp.join(log_path, "models", "model_current.pth"))
        recon_time += time.time() - start_rtime

    ones_cache = {}

    def one_step(loss_fn, loss_weights, extra):
        """One batch: sample -> zero grad -> loss -> backward -> all-reduce -> Adam (reference train.py:195-224).  Everything it
        launches goes to the current stream and nothing in it reads a value back, so it can be captured in a HIP graph."""
        input_data, normals, sdf = next(iter(dataset))
        flat_grad = _zero_flat_grad(model)
        input_data = input_data.to(device); normals = normals.to(device); sdf = sdf.to(device)
        gt = {'normals': normals, 'sdf': sdf}
        if getattr(dataset, 'n_on_surface', None) is not None:
            gt['n_on_surface'] = dataset.n_on_surface      # [on | far | near]: spares loss_s1 its two syncs for the count
        loss = loss_fn(model, input_data, gt, loss_weights, *extra)
        terms = getattr(loss, "terms", None)
        if terms is not None and terms.requires_grad:
            # this repo's loss dicts are views of one tensor (loss_functions.LossTerms): d(sum of the values) = ones on it.
            # Same gradient as the reference's `train_loss += l ...; train_loss.backward()` below, without its ~25 one-element
            # kernels (zeros, 4 adds, stack; backward: 4 sum_to_size, 4 x (zeros + copy) of the selects, 3 adds) — a fifth of
            # the GPU time of a stage-2 step at the reference's batch size.
            vals = terms.detach()
            ones = ones_cache.get(terms.numel())
            if ones is None:
                ones = ones_cache[terms.numel()] = torch.ones_like(vals)
            terms.backward(ones)
        else:
            train_loss = torch.zeros((1, 1), device=device)
            vals = torch.stack([l.reshape(()) for l in loss.values()]).detach()
            for l in loss.values():
                train_loss += l
            train_loss.backward()
        vals = _allreduce_step(flat_grad, vals, terms_are_global=loss_fn is loss_s2)
        optim.step()
        return vals, list(loss.keys())

    # HIP graphs: the step above is ~40 kernel launches and ~0.3 ms of Python; the reference's batch (29 970
    # points) takes the GPU 0.9-1.9 ms, and stage 2 even less, so the loop is host-bound for a third of the recipe.  With one
    # rank, this repo's optimizer and a sampler that can keep its counter on the device, each (loss, weights) phase runs its
    # first GRAPH_WARMUP steps eagerly, captures the next one, and REPLAYS it from then on: the step count, the learning rate
    # (diffudf_amd.optim.Adam.use_schedule) and the sampler's step (PointCloud.use_device_step) live in device memory and advance
    # inside the graph.  Same kernels, same arguments, same order: the loss curve is the eager loop's (tests/test_graph_step_gpu.py).
    # `"hip_graph": false` in the config (or more than one rank: gloo / RCCL inside a capture is not exercised here) keeps the
    # eager loop.
    GRAPH_WARMUP = 3
    plan = [schedule(e) for e in range(epochs)]
    use_graph = (bool(config.get("hip_graph", True)) and device.type == "cuda" and not (_dist() and torch.distributed.get_world_size() > 1)
                 and hasattr(optim, "use_schedule") and getattr(optim, "_model", None) is not None      # (the flat fast path: use_schedule needs it)
                 and hasattr(dataset, "use_device_step"))
    graphs = {}
    if use_graph:
        lr0 = optim.param_groups[0]['lr']
        optim.use_schedule([(lr0 if p[2] is None else p[2]) for p in plan for _ in range(dataset.batchesPerEpoch)])
        dataset.use_device_step()

    pending = None
    start_ttime = time.time()
    prev_fn = None
    for epoch in range(epochs):
        loss_fn, loss_weights, current_lr, extra = plan[epoch]
        if loss_fn is loss_s2 and prev_fn is loss_s1 and _is_main():
            print('Starting second step...')
        prev_fn = loss_fn
        if current_lr is not None:
            for g in optim.param_groups:
                g['lr'] = current_lr
        step_vals, names = [], None
        for _ in range(dataset.batchesPerEpoch):
            st = None
            if use_graph:
                key = (loss_fn, tuple(float(w) for w in loss_weights), tuple(extra))
                st = graphs.setdefault(key, {"eager": 0, "graph": None})
            if st is None or st["eager"] < GRAPH_WARMUP:
                vals, names = one_step(loss_fn, loss_weights, extra)
                if st is not None:
                    st["eager"] += 1
            elif st["graph"] is None:
                st["graph"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(st["graph"]):
                    st["vals"], st["names"] = one_step(loss_fn, loss_weights, extra)
                # the graph holds raw pointers into the training workspace(s) of hip_ops' cache: remember which, and never replay
                # into one that has been evicted since (a schedule that comes back to an earlier phase after the batch layout changed)
                st["ws"] = [weakref.ref(w) for w in hip_ops._ws_cache.values()]
                st["graph"].replay()                   # a capture records, it does not run: this IS the step just counted by one_step
                vals, names = st["vals"], st["names"]
            elif any(r() is None or r() not in hip_ops._ws_cache.values() for r in st["ws"]):
                st.update({"graph": None, "eager": GRAPH_WARMUP - 1})       # its workspace is gone: one eager step, then capture again
                vals, names = one_step(loss_fn, loss_weights, extra)
                st["eager"] += 1
            else:
                st["graph"].replay()
                optim.replayed(); dataset.replayed()
                vals, names = st["vals"], st["names"]
            step_vals.append(vals.clone() if (st is not None and st["graph"] is not None and dataset.batchesPerEpoch > 1) else vals)
        snap = snaps[epoch & 1]
        snap.copy_(model.flat_parameters())
        host = torch.empty((len(step_vals), len(names)), dtype=torch.float32, pin_memory=True)
        if len(step_vals) == 1:
            host[0].copy_(step_vals[0], non_blocking=True)
        else:
            host.copy_(torch.stack(step_vals), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        if pending is not None:
            finish(pending)                            # the PREVIOUS epoch's numbers: this epoch's kernels are queued behind them
        pending = (epoch, names, host, event, snap, optim.param_groups[0]['lr'])
    if pending is not None:
        finish(pending)

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    total_training_time = time.time() - start_ttime - recon_time
    if have_best:
        best_weights = snapshot_state(best_flat)
    if _is_main() and best_weights is not None:
        torch.save(best_weights, osp.join(log_path, "models", "model_best.pth"))
    return losses, best_weights, total_training_time


def train_model_tanh(dataset, model, device, config):
    """gt_mode 'tanh': loss_s1 then loss_s2 (reference train.py:146-283)."""
    epochs = config["epochs"]

    def schedule(epoch):
        if epoch >= config['s1_epochs']:
            lr = 0.5 * (np.cos(epoch / (epochs - config['s1_epochs']) * np.pi) + 1) * config['lr_s2']
            return loss_s2, config['loss_s2_weights'], lr, (config["alpha"],)
        lr = config['lr_s1'] if epoch >= config['warmup_epochs'] else config['warmup_lr']
        return loss_s1, config['loss_s1_weights'], lr, (config["alpha"],)

    return _train(dataset, model, device, config, schedule)


def train_model_siren(dataset, model, device, config):
    """gt_mode 'siren': SIREN's SDF loss at the optimizer's own LR (reference train.py:23-143)."""
    def schedule(epoch):
        return loss_siren, config["loss_weights"], None, ()

    return _train(dataset, model, device, config, schedule)


def setup_train(parameter_dict, cuda_device):
    if not torch.cuda.is_available():
        raise SystemExit("train.py: no GPU visible; the HIP training path has no CPU fallback")
    rank = world = None
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        if _dist():
            # a launcher initialised torch.distributed (and chose this process's device) already: adopt both
            cuda_device = torch.cuda.current_device()
        else:
            cuda_device = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(cuda_device)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", cuda_device))
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    device = torch.device("cuda", int(cuda_device))
    seed = 123
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)

    full_path = create_output_paths(parameter_dict["checkpoint_path"], parameter_dict["experiment_name"], overwrite=False)
    if _is_main():
        with open(osp.join(full_path, "params.json"), "w") as fout:
            json.dump(parameter_dict, fout, indent=4)

    sp = parameter_dict["sampling_percentiles"]
    if str(parameter_dict["dataset"]).startswith("synthetic"):
        dataset = SyntheticPointCloud(parameter_dict["batch_size"], sp, parameter_dict["batches_per_epoch"], seed=seed,
                                      rank=rank or 0, world=world or 1)
    else:
        dataset = PointCloud(parameter_dict["dataset"], parameter_dict["batch_size"], sp,
                             parameter_dict["batches_per_epoch"], device=device,
                             onlyPCloud=parameter_dict.get('onlyPCloud', False), seed=seed,
                             rank=rank or 0, world=world or 1)

    network_params = parameter_dict["network"]
    model = SIREN(n_in_features=3, n_out_features=1, hidden_layer_config=network_params["hidden_layer_nodes"],
                  w0=network_params["w0"], ww=network_params.get("ww", None),
                  activation=network_params.get('activation', 'sine'))
    if network_params.get('pretrained_dict', 'None') != 'None':
        model.load_state_dict(torch.load(network_params['pretrained_dict'], map_location=device))
    if _is_main():
        print(model)
    model.to(device)
    if _dist():                                        # replicas start identical
        torch.distributed.broadcast(model.flat_parameters(), src=0)

    opt_params = parameter_dict["optimizer"]
    gt_mode = parameter_dict.get("gt_mode", "tanh")
    if opt_params["type"] != "adam":
        raise ValueError('Unknown optimizer')
    if gt_mode == 'tanh':
        optimizer = Adam(lr=opt_params["lr_s1"], params=model.parameters(), model=model)
        config_dict = {
            "epochs": parameter_dict["num_epochs"], "s1_epochs": parameter_dict["s1_epochs"],
            "warmup_epochs": parameter_dict.get("warmup_epochs", 0), "warmup_lr": parameter_dict.get("warmup_lr", 1e-4),
            "batch_size": parameter_dict["batch_size"], "epochs_to_checkpoint": parameter_dict["epochs_to_checkpoint"],
            "gt_mode": gt_mode, "log_path": full_path, "optimizer": optimizer,
            "lr_s1": opt_params["lr_s1"], "lr_s2": opt_params["lr_s2"],
            "loss_s1_weights": parameter_dict["loss_s1_weights"], "loss_s2_weights": parameter_dict["loss_s2_weights"],
            "alpha": parameter_dict["alpha"], "resolution": parameter_dict.get("resolution", 256),
            "save_every_epoch": parameter_dict.get("save_every_epoch", True),
            "network": network_params, "luts_path": parameter_dict.get("luts_path"),
            "hip_graph": parameter_dict.get("hip_graph", True),
        }
        losses, best_weights, training_time = train_model_tanh(dataset, model, device, config_dict)
    elif gt_mode == 'siren':
        optimizer = Adam(lr=opt_params["lr"], params=model.parameters(), model=model)
        config_dict = {
            "epochs": parameter_dict["num_epochs"], "batch_size": parameter_dict["batch_size"],
            "epochs_to_checkpoint": parameter_dict["epochs_to_checkpoint"], "gt_mode": gt_mode, "log_path": full_path,
            "optimizer": optimizer, "loss_weights": parameter_dict["loss_weights"],
            "alpha": parameter_dict.get("alpha", 100), "resolution": parameter_dict.get("resolution", 256),
            "save_every_epoch": parameter_dict.get("save_every_epoch", True),
            "hip_graph": parameter_dict.get("hip_graph", True),
        }
        losses, best_weights, training_time = train_model_siren(dataset, model, device, config_dict)
    else:
        raise ValueError('gt_mode not valid')

    if _is_main():
        import pandas as pd
        pd.DataFrame.from_dict(losses).to_csv(osp.join(full_path, "losses.csv"), sep=";", index=None)
        torch.save(model.state_dict(), osp.join(full_path, "models", "model_final.pth"))
        n_pts = dataset.n_global * dataset.batchesPerEpoch * parameter_dict["num_epochs"]
        print(f"training time {training_time:.2f} s  ({n_pts / training_time:.3e} points/s)")
    meshes = []
    if _is_main():
        # post-training artefacts as in reference train.py:403-446: the field slice of the best model (always) and, when
        # `resolution != 0`, its meshes (algorithm 'both') — rank 0 only, the other ranks go on to tear down
        from generate_df import generate_df
        from generate_mc import generate_mc
        best = osp.join(full_path, "models", "model_best.pth")
        if gt_mode != 'tanh':
            print(f"post-training artefacts skipped: gt_mode '{gt_mode}' needs skimage's SDF marching cubes, which is outside this build")
        elif not osp.exists(best):
            print(f"post-training artefacts skipped: {best} was never written (no epoch improved on the initial loss)")
        else:
            print('Generating distance field slices')
            generate_df(best, None, osp.join(full_path, "reconstructions/"),
                        {'device': f"cuda:{int(device.index or 0)}", 'surf_thresh': 1e-3, 'width': 512, 'weight0': network_params["w0"],
                         'gt_mode': gt_mode, 'alpha': parameter_dict.get('alpha', 1),
                         'hidden_layer_nodes': network_params["hidden_layer_nodes"], 'activation': network_params.get('activation', 'sine')})
            if parameter_dict.get('resolution', 256) != 0:
                print('Generating mesh')
                meshes = generate_mc(model=None, gt_mode=gt_mode, device=int(device.index or 0), N=parameter_dict.get('resolution', 256),
                                     output_path=osp.join(full_path, "reconstructions", 'mc_mesh_best.obj'),
                                     alpha=parameter_dict.get('alpha', 1), algorithm='both',
                                     from_file={'w0': network_params["w0"], 'model_path': best,
                                                'hidden_layer_nodes': network_params["hidden_layer_nodes"],
                                                'activation': network_params.get('activation', 'sine')},
                                     luts=parameter_dict.get('luts_path'))
    return training_time, meshes


if __name__ == "__main__":
    p = argparse.ArgumentParser(usage="python train.py path_to_experiments.json cuda_device")
    p.add_argument("experiment_path", type=str, help="Path to the JSON experiment description file")
    p.add_argument("device", type=int, help="Cuda device")
    args = p.parse_args()
    parameter_dict = load_experiment_parameters(args.experiment_path)
    if not bool(parameter_dict):
        raise ValueError("JSON experiment not found")
    setup_train(parameter_dict, args.device)
    if _dist():
        torch.distributed.destroy_process_group()
