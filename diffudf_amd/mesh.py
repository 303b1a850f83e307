# coding: utf-8
"""Host-side mesh preparation for the batch sampler (numpy only, no open3d).

Covers what the reference does once per model with open3d (reference preprocess.py:5-58,
src/preprocess_mesh.py:5-40): read the OBJ, centre it on the vertex mean, scale by 1/(1.1*max|coord|),
and draw an area-uniform surface point cloud that carries TRIANGLE normals.  The per-step work
(src/dataset.py:14-70: pick surface points, uniform domain points, near-surface points, ground-truth
distances) runs on the GPU in csrc/dudf_sample.hip.

open3d is a third-party dependency that is absent here and that no reference test pins:
its RNG stream is not reproduced, its distributions are (area-uniform barycentric sampling; exact
point-to-triangle distance).
"""
import os
import struct

import numpy as np

from . import synth


def load_obj(path):
    """(vertices (V,3) float64, triangles (T,3) int64).  Faces may be `v`, `v/vt`, `v//vn`, `v/vt/vn`; polygons
    are fan-triangulated; negative (relative) indices are honoured."""
    verts, tris = [], []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, dtype=np.float64), np.asarray(tris, dtype=np.int64)


def normalize_vertices(verts):
    """Centre on the vertex mean, then scale so that max|coord| = 1/1.1 (reference src/preprocess_mesh.py:5-15)."""
    v = verts - verts.mean(axis=0, keepdims=True)
    m = np.abs(v).max()
    return v / (m + 0.1 * m)


def triangle_normals_areas(verts, tris):
    a, b, c = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
    n = np.cross(b - a, c - a)
    nn = np.linalg.norm(n, axis=1)
    area = 0.5 * nn
    n = n / np.maximum(nn, 1e-300)[:, None]
    return n, area


def sample_surface(verts, tris, n_points, seed=123):
    """Area-uniform points with the normal of the triangle they fall on (what open3d's
    `sample_points_uniformly(..., use_triangle_normal=True)` delivers, reference src/preprocess_mesh.py:39).
    Counter-based RNG (diffudf_amd.synth) so the cloud is a pure function of (mesh, n_points, seed)."""
    n, area = triangle_normals_areas(verts, tris)
    cdf = np.cumsum(area) / area.sum()
    u = synth.uniform01(seed, 301, 0, n_points)
    r1 = synth.uniform01(seed, 302, 0, n_points)
    r2 = synth.uniform01(seed, 303, 0, n_points)
    t = np.minimum(np.searchsorted(cdf, u, side="right"), len(tris) - 1)
    s = np.sqrt(r1)
    wa, wb, wc = 1.0 - s, s * (1.0 - r2), s * r2
    p = wa[:, None] * verts[tris[t, 0]] + wb[:, None] * verts[tris[t, 1]] + wc[:, None] * verts[tris[t, 2]]
    return p.astype(np.float32), n[t].astype(np.float32)


def triangle_soup(verts, tris):
    """(T,9) float32: the three corners of every triangle, the layout the distance kernel streams through LDS."""
    return np.concatenate([verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]], axis=1).astype(np.float32)


# ---- minimal PLY / OBJ IO for interoperability with the reference's preprocessed files ---------------------
def write_ply_points(path, pos, nrm):
    pos = np.asarray(pos, dtype=np.float32); nrm = np.asarray(nrm, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n"
                 "property float x\nproperty float y\nproperty float z\n"
                 "property float nx\nproperty float ny\nproperty float nz\nend_header\n" % len(pos)).encode())
        f.write(np.concatenate([pos, nrm], axis=1).astype("<f4").tobytes())


def read_ply_points(path):
    """positions (P,3), normals (P,3) float32 from an ascii or binary_little_endian PLY (float/double props)."""
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            header.append(line)
            if line == "end_header":
                break
        fmt = [h.split()[1] for h in header if h.startswith("format")][0]
        n = 0
        props = []
        in_vertex = False
        for h in header:
            p = h.split()
            if p[0] == "element":
                in_vertex = p[1] == "vertex"
                if in_vertex:
                    n = int(p[2])
            elif p[0] == "property" and in_vertex:
                props.append((p[2], p[1]))
        names = [p[0] for p in props]
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n, ndmin=2)
            cols = {nm: data[:, i] for i, nm in enumerate(names)}
        else:
            tmap = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
                    "int": "<i4", "int32": "<i4", "uint": "<u4"}
            dt = np.dtype([(nm, tmap[t]) for nm, t in props])
            raw = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
            cols = {nm: raw[nm] for nm in names}
    pos = np.stack([cols["x"], cols["y"], cols["z"]], 1).astype(np.float32)
    nrm = np.stack([cols["nx"], cols["ny"], cols["nz"]], 1).astype(np.float32)
    return pos, nrm


def write_obj(path, verts, tris):
    with open(path, "w") as f:
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for t in tris:
            f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def prepare(mesh_prefix, surface_points=100000, seed=123, cloud_only=False):
    """What the sampler needs for `<mesh_prefix>`: (triangle soup (T,9), cloud positions (P,3), cloud normals (P,3)).

    Like the reference's PointCloud (src/dataset.py:149-155) it reads `<prefix>_t.obj` and `<prefix>_pc.ply` when
    they exist (files written by the reference's preprocess.py work); otherwise it does that preprocessing
    itself, in memory, from `<prefix>.obj`.  cloud_only (reference :157-159): a `<prefix>_pc.ply` alone is enough and
    no triangles are returned."""
    if cloud_only and os.path.exists(mesh_prefix + "_pc.ply"):
        pos, nrm = read_ply_points(mesh_prefix + "_pc.ply")
        return None, pos, nrm
    t_obj, pc_ply, raw = mesh_prefix + "_t.obj", mesh_prefix + "_pc.ply", mesh_prefix + ".obj"
    if os.path.exists(t_obj):
        verts, tris = load_obj(t_obj)
    elif os.path.exists(raw):
        verts, tris = load_obj(raw)
        verts = normalize_vertices(verts)
    else:
        raise FileNotFoundError(f"neither {t_obj} nor {raw} exists")
    if os.path.exists(pc_ply):
        pos, nrm = read_ply_points(pc_ply)
    else:
        pos, nrm = sample_surface(verts, tris, int(surface_points), seed)
    return (None if cloud_only else triangle_soup(verts, tris)), pos, nrm
