# coding: utf-8
"""Field extraction and CAP-UDF meshing for the marching-cubes consumers — reference src/render_mc.py:20-99
`extract_fields` and :201-256 `extract_mesh_CAP` (BASELINE config 5: the batched field+gradient query feeding CAP-UDF
extraction, all on the device) — and :101-199 `extract_mesh_MESHUDF`, the MeshUDF marching cubes over the same fields, on
the host (C++ library behind `marching_cubes.udf_mc_lewiner`; serial by nature)."""
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


def extract_mesh_MESHUDF(df_values, normals, device, smooth_borders=False, luts=None, **kwargs):
    """Reference src/render_mc.py:101-199 (`extract_mesh_MESHUDF`): the MeshUDF marching cubes over the (N, N, N) field and
    its (N, N, N, 3) direction field (outputs of `extract_fields`; tensors or numpy arrays), vertices shifted to
    [-1, 1]^3.  The extraction itself — the reference's Cython extension — is `diffudf_amd.marching_cubes.udf_mc_lewiner`
    over the host C++ library (bit-identical vertices and faces, tests/test_meshudf.py); the Lewiner tables come from the
    caller (`luts=`: a dict or a path) or from `marching_cubes.load_luts()` ($DUDF_MESHUDF_LUTS, then the reference's own
    `_marching_cubes_lewiner_luts.py` on `sys.path`); none found: MeshUDFError.  Tensors come back on `device`, as in
    the reference.
    Returns (vertices, faces, mesh).  The reference then cleans the mesh with trimesh (`process`, duplicate / degenerate
    faces, `fill_holes`, optional Laplacian smoothing of the border): done the same way when trimesh is importable;
    without it the raw extraction is returned in a `TriangleSoup` (and `smooth_borders` is ignored)."""
    from .marching_cubes import udf_mc_lewiner
    d = df_values.detach().cpu().numpy() if torch.is_tensor(df_values) else np.asarray(df_values)
    g = normals.detach().cpu().numpy() if torch.is_tensor(normals) else np.asarray(normals)
    d = np.where(d < 0, 0, d).astype(np.float32)
    N = d.shape[0]
    voxel_size = 2.0 / (N - 1)
    verts, faces, _, _ = udf_mc_lewiner(d, g.astype(np.float32), spacing=[voxel_size] * 3, avg_thresh=1.05, max_thresh=1.75, luts=luts)
    verts = verts - 1                                   # voxel origin (-1, -1, -1)
    if len(faces) == 0:
        raise ValueError("Could not find surface in volume")
    try:
        import trimesh
    except ImportError:
        mesh = TriangleSoup(verts, faces)
        return (torch.from_numpy(np.ascontiguousarray(verts)).float().to(device),
                torch.from_numpy(np.ascontiguousarray(faces)).long().to(device), mesh)
    mesh = trimesh.Trimesh(verts, faces).process(validate=False)
    mesh.remove_duplicate_faces(); mesh.remove_degenerate_faces(); mesh.fill_holes()
    mesh2 = trimesh.Trimesh(mesh.vertices, mesh.faces)
    nv, nf, it = 0, 0, 0
    while (nv, nf) != (len(mesh2.vertices), len(mesh2.faces)) and it < 10:
        mesh2 = mesh2.process(validate=False)
        mesh2.remove_duplicate_faces(); mesh2.remove_degenerate_faces()
        nv, nf = len(mesh2.vertices), len(mesh2.faces)
        it += 1
        mesh2 = trimesh.Trimesh(mesh2.vertices, mesh2.faces)
    mesh = trimesh.Trimesh(mesh2.vertices, mesh2.faces)
    if smooth_borders:
        from collections import defaultdict
        from scipy.sparse import coo_matrix
        border = trimesh.grouping.group_rows(mesh.edges_sorted, require_count=1)
        nb = defaultdict(list)
        for u, v in mesh.edges_sorted[border]:
            nb[u].append(v); nb[v].append(u)
        bv = np.array(list(nb.keys()))
        if len(bv) > 0:
            pi, pj = [], []
            for k, ns in enumerate(nb.values()):
                for j in ns:
                    pi.append(k); pj.append(j)
            sp = coo_matrix((np.ones(len(pi)), (pi, pj)), shape=(len(bv), len(mesh.vertices)))
            for _ in range(5):
                avg = sp @ mesh.vertices / sp.sum(axis=1)
                mesh.vertices[bv] = mesh.vertices[bv] + 0.3 * (np.asarray(avg) - mesh.vertices[bv])
    return torch.tensor(mesh.vertices).float().to(device), torch.tensor(mesh.faces).long().to(device), mesh
