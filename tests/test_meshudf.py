# coding: utf-8
"""MeshUDF marching cubes: the host C++ library against
  * tests/golden/g10_meshudf.npz — outputs of the reference's own Cython extension (rebuilt by oracle/build_ref.py, run
    through the reference's wrapper; tests/golden/make_golden.py g10): vertices, faces, normals and values BIT FOR BIT;
    the look-up tables in the fixture are the extension's `luts` argument as the reference's wrapper passes it;
  * the rebuilt reference extension itself on randomised fields, when /root/reference is there (this container only).
No GPU involved."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from diffudf_amd import marching_cubes as mc  # noqa: E402

G10 = np.load(os.path.join(REPO, "tests", "golden", "g10_meshudf.npz"))
LUTS = {k[4:]: G10[k] for k in G10.files if k.startswith("lut_")}


def test_library_loads_and_exports_the_c_abi():
    import ctypes
    lib = ctypes.CDLL(os.path.join(REPO, "diffudf_amd", "libdudf_meshudf.so"))
    for sym in ("dudf_meshudf_run", "dudf_meshudf_sizes", "dudf_meshudf_copy", "dudf_meshudf_free"):
        assert hasattr(lib, sym), sym
    hdr = open(os.path.join(REPO, "include", "dudf_meshudf.h")).read()
    for sym in ("dudf_meshudf_run", "dudf_meshudf_sizes", "dudf_meshudf_copy", "dudf_meshudf_free"):
        assert sym in hdr
    assert set(mc.LUT_NAMES) == set(LUTS)


@pytest.mark.parametrize("tag", [str(t) for t in G10["cases"]])
def test_bit_identical_to_the_reference_extension(tag):
    udf, g = G10[tag + "_udf"], G10[tag + "_grads"]
    n = udf.shape[0]
    v, f, nn, val = mc.udf_mc_lewiner(udf, g, spacing=[2.0 / (n - 1)] * 3, avg_thresh=1.05, max_thresh=1.75, luts=LUTS)
    assert v.dtype == G10[tag + "_vertices"].dtype and v.shape == G10[tag + "_vertices"].shape
    assert np.array_equal(f, G10[tag + "_faces"])
    assert np.array_equal(v, G10[tag + "_vertices"])
    assert np.array_equal(val, G10[tag + "_values"])
    assert np.array_equal(nn, G10[tag + "_normals"], equal_nan=True)


def test_mesh_properties_and_render_mc_mirror():
    """closed sphere: V - E + F = 2, every edge shared by two triangles; `extract_mesh_MESHUDF` shifts to [-1, 1]^3"""
    import torch
    from src.render_mc import extract_mesh_MESHUDF
    udf, g = G10["sphere_14_0_udf"], G10["sphere_14_0_grads"]
    verts, faces, mesh = extract_mesh_MESHUDF(torch.from_numpy(udf), torch.from_numpy(g), "cpu", luts=LUTS)
    f = np.asarray(mesh.faces)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all()
    assert len(mesh.vertices) - len(counts) + len(f) == 2
    r = np.linalg.norm(np.asarray(mesh.vertices), axis=1)
    assert abs(r.mean() - 0.6) < 0.02 and r.min() > 0.5 and r.max() < 0.7


def test_closed_surfaces_come_out_watertight():
    """size-independent property (no reference needed): random spheres -> closed 2-manifolds, vertices on the sphere"""
    rng = np.random.default_rng(2)
    for _ in range(12):
        n = int(rng.integers(12, 40))
        c = rng.uniform(-0.25, 0.25, 3); rad = rng.uniform(0.3, 0.6)
        ax = np.linspace(-1, 1, n)
        A, B, C = np.meshgrid(ax, ax, ax, indexing="ij")
        r = np.sqrt((A - c[0]) ** 2 + (B - c[1]) ** 2 + (C - c[2]) ** 2)
        sd = r - rad
        g = np.stack([A - c[0], B - c[1], C - c[2]], -1) / np.maximum(r, 1e-9)[..., None]
        v, f, _, _ = mc.udf_mc_lewiner(np.abs(sd).astype(np.float32), (-g * np.sign(sd)[..., None]).astype(np.float32),
                                       spacing=[2.0 / (n - 1)] * 3, luts=LUTS)
        e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        _, counts = np.unique(e, axis=0, return_counts=True)
        assert (counts == 2).all(), (n, rad)
        assert len(v) - len(counts) + len(f) == 2
        d = np.linalg.norm(v - 1 - c, axis=1)             # wrapper output is in [0, 2]^3 (z-y-x order = our array axes)
        assert np.abs(d - rad).max() < 1.5 * (2.0 / (n - 1))


def test_argument_checks_and_empty_field():
    with pytest.raises(ValueError):
        mc.udf_mc_lewiner(np.zeros((4, 4)), np.zeros((4, 4, 3)), luts=LUTS)
    with pytest.raises(NotImplementedError):
        mc.udf_mc_lewiner(np.ones((4, 4, 4), np.float32), np.zeros((4, 4, 4, 3), np.float32), step_size=2, luts=LUTS)
    with pytest.raises(RuntimeError):                       # nothing near the surface: the reference raises the same
        mc.udf_mc_lewiner(np.full((6, 6, 6), 5.0, np.float32), np.zeros((6, 6, 6, 3), np.float32), luts=LUTS)
    bad = dict(LUTS); del bad["CASES"]
    with pytest.raises(KeyError):
        mc.udf_mc_lewiner(np.ones((4, 4, 4), np.float32), np.zeros((4, 4, 4, 3), np.float32), luts=bad)


@pytest.mark.skipif(not os.path.exists("/root/reference/src/marching_cubes/_marching_cubes_lewiner_cy.pyx"),
                    reason="needs the reference sources (build container only)")
def test_randomised_against_the_rebuilt_reference_extension():
    from oracle import build_ref
    ref = build_ref.load()
    if ref is None:
        pytest.skip("Cython / g++ not available")
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    luts = mc.load_reference_luts()                         # the reference's own table module, decoded like its wrapper does
    for k in mc.LUT_NAMES:
        assert np.array_equal(luts[k], LUTS[k]), k          # and the fixture holds exactly those
    rng = np.random.default_rng(11)
    compared = 0
    for trial in range(40):
        n = int(rng.integers(8, 34))
        ax = np.linspace(-1, 1, n)
        A, B, C = np.meshgrid(ax, ax, ax, indexing="ij")
        sd = np.full(A.shape, 10.0); g = np.zeros(A.shape + (3,))
        for _ in range(int(rng.integers(1, 6))):
            c = rng.uniform(-0.6, 0.6, 3); rad = rng.uniform(0.15, 0.5)
            r = np.sqrt((A - c[0]) ** 2 + (B - c[1]) ** 2 + (C - c[2]) ** 2)
            gi = np.stack([A - c[0], B - c[1], C - c[2]], -1) / np.maximum(r, 1e-9)[..., None]
            m = (r - rad) < sd
            sd = np.where(m, r - rad, sd); g = np.where(m[..., None], gi, g)
        if trial % 3 == 1:
            sd = np.where(A + 0.3 * B > 0.1, np.abs(sd) + np.abs(A + 0.3 * B - 0.1), sd)
        udf = np.abs(sd); gg = -g * np.sign(sd)[..., None]
        if trial % 4 == 2:
            gg = gg + 0.4 * rng.standard_normal(gg.shape); gg /= np.maximum(np.linalg.norm(gg, axis=-1, keepdims=True), 1e-9)
        if trial % 5 == 3:
            udf = np.where(udf < 0.02, 0.0, udf); gg = np.where((rng.random(A.shape) < 0.05)[..., None], 0.0, gg)
        if trial % 7 == 4:
            udf = np.round(udf * 64) / 64
        udf, gg = udf.astype(np.float32), gg.astype(np.float32)
        sp = [2.0 / (n - 1)] * 3
        try:
            rv, rf, rn, rval = ref.udf_mc_lewiner(udf, gg, spacing=sp)
        except RuntimeError:
            with pytest.raises(RuntimeError):
                mc.udf_mc_lewiner(udf, gg, spacing=sp, luts=luts)
            continue
        v, f, nn, val = mc.udf_mc_lewiner(udf, gg, spacing=sp, luts=luts)
        assert np.array_equal(f, rf) and np.array_equal(v, rv), (trial, n)
        assert np.array_equal(val, rval) and np.array_equal(nn, rn, equal_nan=True), (trial, n)
        compared += 1
    assert compared >= 30


def test_lut_sources(tmp_path, monkeypatch):
    """The Lewiner tables are an input (never shipped): a dict, an .npz path, $DUDF_MESHUDF_LUTS — and a documented error
    when none of them is there (`generate_mc.py meshudf` must not depend on a reference checkout silently)."""
    udf, g = G10["sphere_14_0_udf"], G10["sphere_14_0_grads"]
    want = mc.udf_mc_lewiner(udf, g, luts=LUTS)
    path = str(tmp_path / "luts.npz")
    mc.save_luts_npz(LUTS, path)
    got = mc.udf_mc_lewiner(udf, g, luts=path)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(want, got))
    monkeypatch.setenv("DUDF_MESHUDF_LUTS", path)
    got = mc.udf_mc_lewiner(udf, g)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(want, got))
    monkeypatch.delenv("DUDF_MESHUDF_LUTS")
    for name in ("_marching_cubes_lewiner_luts", "src.marching_cubes._marching_cubes_lewiner_luts"):
        monkeypatch.setitem(sys.modules, name, None)            # a reference checkout on sys.path must not rescue the default
    with pytest.raises(mc.MeshUDFError, match="look-up tables are an INPUT"):
        mc.udf_mc_lewiner(udf, g)
    with pytest.raises(mc.MeshUDFError, match="does not exist"):
        mc.udf_mc_lewiner(udf, g, luts=str(tmp_path / "nope.npz"))
    np.savez(str(tmp_path / "partial.npz"), CASES=LUTS["CASES"])
    with pytest.raises(mc.MeshUDFError, match="lacks"):
        mc.udf_mc_lewiner(udf, g, luts=str(tmp_path / "partial.npz"))
