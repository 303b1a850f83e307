# coding: utf-8
"""ORACLE — test infrastructure only.  Never imported by the product path.

numpy restatement of the per-step batch sampler (reference src/dataset.py:14-70 `sampleTrainingData`):
surface picks, uniform domain points, near-surface points displaced along the normal by N(0, 0.01), and the
ground-truth distance to the mesh.  The reference delegates the distance to open3d's RaycastingScene
(open3d 0.17.0, `dudf.yml:305`), which is absent here and which no reference test pins: **parity unpinned**
for that third-party piece; it is restated as the exact point-to-triangle distance (fp64, closest point by
clamped barycentric regions) and checked against a dense brute force over sampled surface points in
tests/test_sampler_oracle.py.  Random numbers: the counter-based generator of diffudf_amd/synth.py, so the HIP
kernel (csrc/dudf_sample.hip) can be compared sample by sample.
"""
import numpy as np

from diffudf_amd import synth


def point_triangle_dist2(p, tri):
    """Squared distance from points p (N,3) to EACH triangle tri (T,9): (N,T), fp64."""
    p = np.asarray(p, dtype=np.float64)[:, None, :]
    a = tri[None, :, 0:3].astype(np.float64); b = tri[None, :, 3:6].astype(np.float64); c = tri[None, :, 6:9].astype(np.float64)
    ab, ac, ap = b - a, c - a, p - a
    d1 = (ab * ap).sum(-1); d2 = (ac * ap).sum(-1)
    bp = p - b
    d3 = (ab * bp).sum(-1); d4 = (ac * bp).sum(-1)
    cp = p - c
    d5 = (ab * cp).sum(-1); d6 = (ac * cp).sum(-1)
    va = d3 * d6 - d5 * d4; vb = d5 * d2 - d1 * d6; vc = d1 * d4 - d3 * d2
    with np.errstate(divide="ignore", invalid="ignore"):
        den = 1.0 / (va + vb + vc)
        v_in, w_in = vb * den, vc * den
        closest = a + ab * v_in[..., None] + ac * w_in[..., None]                    # interior
        t_ab = d1 / (d1 - d3); t_ac = d2 / (d2 - d6); t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    def put(mask, val):
        np.copyto(closest, val, where=mask[..., None])
    put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + (c - b) * t_bc[..., None])
    put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + ac * t_ac[..., None])
    put((d6 >= 0) & (d5 <= d6), c + 0 * ap)
    put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + ab * t_ab[..., None])
    put((d3 >= 0) & (d4 <= d3), b + 0 * ap)
    put((d1 <= 0) & (d2 <= 0), a + 0 * ap)
    d = p - closest
    return (d * d).sum(-1)


def mesh_distance(p, tri, chunk=2048):
    out = np.empty(len(p))
    for i in range(0, len(p), chunk):
        out[i:i + chunk] = np.sqrt(point_triangle_dist2(p[i:i + chunk], tri).min(axis=1))
    return out


def cloud_distance(p, pc_pos, chunk=1024):
    """Distance to the nearest cloud point (reference src/dataset.py:72-78 `shortestDistance`), fp64."""
    out = np.empty(len(p))
    X = pc_pos.astype(np.float64)
    for i in range(0, len(p), chunk):
        d = p[i:i + chunk, None, :].astype(np.float64) - X[None]
        out[i:i + chunk] = np.sqrt((d * d).sum(-1).min(axis=1))
    return out


def sample_batch(tri, pc_pos, pc_nrm, n_on, n_far, n_near, seed, step, rank=0, world=1):
    """This rank's slice [on | far | near]: x (n,3) f32, normals (n,3) f32, sdf (n,1) f32.
    tri=None: the point-cloud-only variant (reference src/dataset.py:80-131)."""
    base = 1000 * step
    P = len(pc_pos)
    sl = lambda m: (m * rank // world, m * (rank + 1) // world)   # noqa: E731
    (o0, o1), (f0, f1), (c0, c1) = sl(n_on), sl(n_far), sl(n_near)
    pick = lambda idx: np.floor(synth.uniform01(seed, base + 400, 0, n_on)[idx] * P).astype(np.int64)  # noqa: E731
    ci = pick(np.arange(o0, o1))
    x_on, n_onv = pc_pos[ci], pc_nrm[ci]
    far = np.stack([synth.uniform01(seed, base + 401 + k, f0, f1 - f0) * 2.0 - 1.0 for k in range(3)], 1).astype(np.float32)
    k = np.floor(synth.uniform01(seed, base + 404, c0, c1 - c0) * n_on).astype(np.int64)
    cn = pick(k)
    u1 = synth.uniform01(seed, base + 405, c0, c1 - c0); u2 = synth.uniform01(seed, base + 406, c0, c1 - c0)
    off = (0.01 * np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)
    near = (pc_pos[cn] + (pc_nrm[cn] * off[:, None]).astype(np.float32)).astype(np.float32)
    x = np.concatenate([x_on, far, near]).astype(np.float32)
    nrm = np.concatenate([n_onv, np.zeros((len(far) + len(near), 3), np.float32)]).astype(np.float32)
    if tri is None:
        sdf = np.concatenate([np.zeros(len(x_on)), cloud_distance(far, pc_pos), np.abs(off)]).astype(np.float32)
    else:
        sdf = np.concatenate([np.zeros(len(x_on)), mesh_distance(far, tri), mesh_distance(near, tri)]).astype(np.float32)
    sdf[len(x_on):] = np.maximum(sdf[len(x_on):], np.float32(1.17549435e-38))   # off-surface samples never carry the on-surface marker sdf == 0 (dudf_sample.hip)
    return x, nrm, sdf.reshape(-1, 1)
