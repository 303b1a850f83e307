# coding: utf-8
"""Field extraction for the marching-cubes consumers — the query half of reference src/render_mc.py:20-99
`extract_fields` (BASELINE config 5).  The mesh extraction itself (CAP-UDF / MeshUDF marching cubes, :103-256)
is outside this build's scope."""
import torch

from . import hip_ops
from ._lib import DudfError


def extract_fields(decoder, latent_vec, N, gt_mode, device, alpha, chunk=1 << 20):
    """Same signature and return contract as the reference: (df_values (N,N,N), vecs (N,N,N,3)) float32 tensors on
    `device`.  df = inverse(gt_mode, |f|, alpha); vecs = -normalize(grad f), or the top Hessian eigenvector aligned with
    it where the normalised gradient is shorter than 0.04 (a vanishing gradient).

    Unlike the reference (3 float64 numpy arrays of N^3 rows and 4096-point chunks through PyTorch autograd, with
    the Hessian evaluated for EVERY grid point), coordinates are derived from the grid index inside the kernel,
    value and gradient come from the fused sweeps in 2^20-point chunks, the inverse map / normalisation are a kernel
    epilogue, and the Hessian path runs only for the points that take the fallback (normally none)."""
    if latent_vec is not None and torch.as_tensor(latent_vec).numel() != 0:
        raise DudfError("extract_fields: latent vectors are not part of the HIP path")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise DudfError("extract_fields: needs the GPU; there is no CPU fallback path")
    cfg = decoder.hip_cfg
    theta = decoder.flat_parameters()
    total = N ** 3
    df = torch.empty(total, dtype=torch.float32, device=dev)
    vec = torch.empty(total, 3, dtype=torch.float32, device=dev)
    flags = []
    start = 0
    ws = hip_ops.query_workspace_for(cfg, min(chunk, total), dev)
    while start < total:
        cnt = min(chunk, total - start)
        w = ws if cnt == ws.n else hip_ops.QueryWorkspace(cfg, cnt, dev)
        flags.append((start, cnt, hip_ops.grid_fields(cfg, theta, N, start, cnt, gt_mode, alpha, df, vec, w)))
        start += cnt
    if int(torch.stack([f for _, _, f in flags]).sum()) > 0:
        # rare path: recompute the flagged points with the Hessian frame (reference :77-93)
        bad = (vec.norm(dim=-1) < 0.04).nonzero().flatten()
        idx = bad.to(torch.int64)
        voxel = 2.0 / (N - 1)
        xyz = torch.stack([(idx // (N * N)) % N, (idx // N) % N, idx % N], 1).float() * voxel - 1.0
        _, _, _, _, V = hip_ops.query_frame(cfg, theta, xyz)
        n_hat = V[:, :, 2]
        sign = torch.where((vec[bad] * n_hat).sum(-1, keepdim=True) < 0, -1.0, 1.0)
        vec[bad] = sign * n_hat
    return df.reshape(N, N, N), vec.reshape(N, N, N, 3)
