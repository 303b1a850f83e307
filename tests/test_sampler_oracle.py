# coding: utf-8
"""CPU: mesh preparation + the sampler oracle (exact point-to-triangle distance) on the reference's beetle mesh
(tests/golden/beetle.obj is the reference's data file data/beetle/beetle.obj, a fixture, not source)."""
import os
import numpy as np

from diffudf_amd import mesh
from oracle import sampler_oracle as SO

HERE = os.path.dirname(os.path.abspath(__file__))
BEETLE = os.path.join(HERE, "golden", "beetle")


def test_mesh_preparation_matches_reference_recipe(tmp_path):
    verts, tris = mesh.load_obj(BEETLE + ".obj")
    assert verts.shape == (1148, 3) and tris.shape == (2053, 3)          # 1 148 verts / 2 053 tris
    v = mesh.normalize_vertices(verts)
    assert np.allclose(v.mean(0), 0, atol=1e-12) and abs(np.abs(v).max() - 1 / 1.1) < 1e-12
    pos, nrm = mesh.sample_surface(v, tris, 20000, seed=5)
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-5)
    tri = mesh.triangle_soup(v, tris)
    # sampled surface points lie on the mesh; their normal is the normal of a triangle through them
    assert SO.mesh_distance(pos[:500], tri).max() < 1e-6
    # area-uniform: the share of samples per triangle follows its area share
    _, area = mesh.triangle_normals_areas(v, tris)
    big = np.argsort(area)[-50:]
    d2 = SO.point_triangle_dist2(pos[:4000], tri[big])
    share = (d2.min(axis=1) < 1e-12).mean()
    assert abs(share - area[big].sum() / area.sum()) < 0.03
    # PLY / OBJ round trip (interoperability with the reference's preprocessed files)
    mesh.write_ply_points(str(tmp_path / "b_pc.ply"), pos, nrm)
    mesh.write_obj(str(tmp_path / "b_t.obj"), v, tris)
    tri2, pos2, nrm2 = mesh.prepare(str(tmp_path / "b"))
    assert np.array_equal(pos2, pos) and np.array_equal(nrm2, nrm) and np.allclose(tri2, tri, atol=1e-7)


def test_distance_oracle_against_dense_brute_force():
    verts, tris = mesh.load_obj(BEETLE + ".obj")
    v = mesh.normalize_vertices(verts)
    tri = mesh.triangle_soup(v, tris)
    rng = np.random.default_rng(0)
    q = rng.uniform(-1, 1, (40, 3))
    d = SO.mesh_distance(q, tri)
    # independent check: distance to a very dense point sampling of the same surface bounds it from above
    pos, _ = mesh.sample_surface(v, tris, 400000, seed=1)
    dd = np.sqrt(((q[:, None, :] - pos[None].astype(np.float64)) ** 2).sum(-1).min(1))
    assert (d <= dd + 1e-9).all() and np.abs(d - dd).max() < 5e-3


def test_sampler_is_shardable_and_deterministic():
    tri, pos, nrm = mesh.prepare(BEETLE, 5000, seed=3)
    full = SO.sample_batch(tri, pos, nrm, 99, 50, 51, seed=9, step=4)
    parts = [SO.sample_batch(tri, pos, nrm, 99, 50, 51, seed=9, step=4, rank=r, world=3) for r in range(3)]
    x = np.concatenate([p[0] for p in parts])           # the union over ranks is the single-rank batch
    assert sorted(map(tuple, np.round(x, 6))) == sorted(map(tuple, np.round(full[0], 6)))
    assert (full[2][:99] == 0).all() and (full[2][99:] > 0).all() and (full[1][99:] == 0).all()
    near_d = full[2][99 + 50:, 0]
    assert near_d.max() < 0.06 and np.median(near_d) < 0.02           # |N(0, 0.01)| along the normal


def test_point_cloud_only_oracle_matches_reference_formula():
    """Reference src/dataset.py:72-78 computes sqrt(min(|X|^2 - 2 P.X) + |P|^2); the oracle uses min |P - X|."""
    rng = np.random.default_rng(3)
    X = rng.uniform(-0.9, 0.9, (700, 3)).astype(np.float32)
    Nn = rng.normal(size=(700, 3)); Nn = (Nn / np.linalg.norm(Nn, axis=1, keepdims=True)).astype(np.float32)
    x, nrm, sdf = SO.sample_batch(None, X, Nn, 40, 50, 60, seed=5, step=1)
    P = x[40:90].astype(np.float64); Xd = X.astype(np.float64)
    ref = np.sqrt(((Xd * Xd).sum(1)[None] - 2 * P @ Xd.T).min(1) + (P * P).sum(1))
    assert np.abs(sdf[40:90, 0] - ref).max() < 1e-6
    assert (sdf[:40] == 0).all() and (nrm[40:] == 0).all()
    # near: displaced along the unit normal by `off`, stored distance |off| ~ N(0, 0.01)
    assert 0.003 < sdf[90:, 0].mean() < 0.02
    parts = [SO.sample_batch(None, X, Nn, 40, 50, 60, seed=5, step=1, rank=r, world=3) for r in range(3)]
    assert sum(len(p[0]) for p in parts) == 150
