# coding: utf-8
"""Field extraction and CAP-UDF meshing for the marching-cubes consumers — reference src/render_mc.py:20-99
`extract_fields` and :201-256 `extract_mesh_CAP` (BASELINE config 5: the batched field+gradient query feeding CAP-UDF
extraction, all on the device).  MeshUDF's marching cubes (:103-199) is not part of this path."""
import numpy as np
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


class TriangleSoup:
    """What `extract_mesh_CAP` returns when `trimesh` is not installed: the two arrays a `trimesh.Trimesh(v, f,
    process=False)` would hold, plus OBJ / PLY export (the reference only ever calls `.export(path)` and reads
    `.vertices` / `.faces`, generate_mc.py:60-75)."""

    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def export(self, path):
        v, f = self.vertices, self.faces
        if str(path).endswith(".ply"):
            with open(path, "w") as fo:
                fo.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
                         "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(v), len(f)))
                np.savetxt(fo, v, fmt="%.17g")
                np.savetxt(fo, np.concatenate([np.full((len(f), 1), 3), f], 1), fmt="%d")
        else:
            with open(path, "w") as fo:
                np.savetxt(fo, v, fmt="v %.17g %.17g %.17g")
                np.savetxt(fo, f + 1, fmt="f %d %d %d")
        return path


def extract_mesh_CAP(ndf, grad, resolution, threshold=0.008, device=None):
    """Same signature as reference src/render_mc.py:201 (`ndf` (N,N,N), `grad` (N,N,N,3): numpy arrays or tensors — the
    outputs of `extract_fields` can be passed straight through without leaving the device).  Active-cell test, sign by
    gradient, per-cell marching cubes and compaction run in `dudf_capudf_count` / `dudf_capudf_emit`; returns a
    `trimesh.Trimesh(process=False)` when trimesh is importable, else a `TriangleSoup` with the same two arrays."""
    dev = torch.device(device) if device is not None else (ndf.device if torch.is_tensor(ndf) and ndf.is_cuda else torch.device("cuda", 0))
    if dev.type != "cuda":
        raise DudfError("extract_mesh_CAP: needs the GPU; there is no CPU fallback path")
    t = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev)   # noqa: E731
    d, g = t(ndf), t(grad)
    if d.shape[0] != resolution:
        raise ValueError("resolution does not match the field")
    v, f = hip_ops.capudf_extract(d, g, threshold)
    v, f = v.cpu().numpy(), f.cpu().numpy()
    try:
        import trimesh
        return trimesh.Trimesh(v, f, process=False)
    except ImportError:
        return TriangleSoup(v, f)
