# coding: utf-8
"""MeshUDF marching cubes — the wrapper of reference src/marching_cubes/_marching_cubes_lewiner.py:80-141
(`udf_mc_lewiner`) over the host C++ library `libdudf_meshudf.so` (csrc/dudf_meshudf.cpp) instead of the reference's
Cython extension.

The Lewiner look-up tables are NOT part of this package: like the reference's extension, the native entry point takes them
as an argument (`LutProvider` there, a flat int8 buffer + offsets here).  `udf_mc_lewiner(..., luts=...)` accepts
  * a dict {name: int8 array} in the reference's names (`CASES`, `TILING1` ... `SUBCONFIG13`, + the three `EDGESREL*`),
  * a path (str): an `.npz` of those arrays (`save_luts_npz` writes one), a `_marching_cubes_lewiner_luts.py` file in the
    reference's (shape, base64 text) encoding, or a directory holding one — `load_luts`,
  * None: `load_luts()` without a path: the `DUDF_MESHUDF_LUTS` environment variable (same three forms), then a module
    `_marching_cubes_lewiner_luts` importable from `sys.path` (the file a user of the reference already has next to its
    wrapper; scikit-image ships the same tables), decoded exactly as the reference decodes it (:144-148).  Nothing found:
    `MeshUDFError` that says so.  `generate_mc.py` and `train.py` pass a config key `luts_path` through.
"""
import base64
import ctypes
import os

import numpy as np

LUT_NAMES = (
    "EDGESRELX", "EDGESRELY", "EDGESRELZ", "CASESCLASSIC", "CASES",
    "TILING1", "TILING2", "TILING3_1", "TILING3_2", "TILING4_1", "TILING4_2", "TILING5", "TILING6_1_1", "TILING6_1_2",
    "TILING6_2", "TILING7_1", "TILING7_2", "TILING7_3", "TILING7_4_1", "TILING7_4_2", "TILING8", "TILING9", "TILING10_1_1",
    "TILING10_1_1_", "TILING10_1_2", "TILING10_2", "TILING10_2_", "TILING11", "TILING12_1_1", "TILING12_1_1_",
    "TILING12_1_2", "TILING12_2", "TILING12_2_", "TILING13_1", "TILING13_1_", "TILING13_2", "TILING13_2_", "TILING13_3",
    "TILING13_3_", "TILING13_4", "TILING13_5_1", "TILING13_5_2", "TILING14",
    "TEST3", "TEST4", "TEST6", "TEST7", "TEST10", "TEST12", "TEST13", "SUBCONFIG13")      # = enum LutId in dudf_meshudf.cpp

def _edge_corner_tables():
    """Edge index -> the (x, y, z) offsets of the two cube corners it joins, per axis, in the cube numbering the tables use
    (`get_index_in_facelayer` sketch, reference .pyx:690-700): edges 0-3 walk the bottom ring (0,0) -> (1,0) -> (1,1) ->
    (0,1) -> (0,0), edges 4-7 the same ring one level up, edges 8-11 are the verticals at the ring's corners."""
    ring = [(0, 0), (1, 0), (1, 1), (0, 1)]
    ex, ey, ez = np.zeros((12, 2), "int8"), np.zeros((12, 2), "int8"), np.zeros((12, 2), "int8")
    for i in range(4):
        a, b = ring[i], ring[(i + 1) % 4]
        for lvl in (0, 1):
            ex[i + 4 * lvl], ey[i + 4 * lvl], ez[i + 4 * lvl] = (a[0], b[0]), (a[1], b[1]), (lvl, lvl)
        ex[8 + i], ey[8 + i], ez[8 + i] = (a[0], a[0]), (a[1], a[1]), (0, 1)
    return {"EDGESRELX": ex, "EDGESRELY": ey, "EDGESRELZ": ez}


EDGESREL = _edge_corner_tables()

_LIB = None


class MeshUDFError(RuntimeError):
    pass


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdudf_meshudf.so")
        if not os.path.exists(path):
            raise MeshUDFError(f"{path} not built (python -c 'import __graft_entry__ as g; g.build()')")
        lib = ctypes.CDLL(path)
        lib.dudf_meshudf_run.restype = ctypes.c_void_p
        lib.dudf_meshudf_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                         ctypes.c_float]
        lib.dudf_meshudf_sizes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
        lib.dudf_meshudf_sizes.restype = None
        lib.dudf_meshudf_copy.argtypes = [ctypes.c_void_p] * 5
        lib.dudf_meshudf_copy.restype = None
        lib.dudf_meshudf_free.argtypes = [ctypes.c_void_p]
        lib.dudf_meshudf_free.restype = None
        _LIB = lib
    return _LIB


def _decode_module(mcluts):
    """Tables of a `_marching_cubes_lewiner_luts` module, decoded as the reference does: (shape, base64 text) -> int8 array."""
    out = dict(EDGESREL)
    for name in LUT_NAMES[3:]:
        shape, text = getattr(mcluts, name)
        ar = np.frombuffer(base64.decodebytes(text.encode("utf-8")), dtype="int8").copy()
        out[name] = ar.reshape(shape)
    return out


def save_luts_npz(luts, path):
    """Write a table set as the `.npz` `load_luts(path)` reads (one-off conversion of a reference checkout's tables)."""
    np.savez_compressed(path, **{name: np.asarray(luts[name], dtype=np.int8) for name in LUT_NAMES})


def load_luts(path=None):
    """The 48 Lewiner tables as {name: int8 array}.  `path`: an .npz, a `_marching_cubes_lewiner_luts.py`, or a directory that
    holds one; None: $DUDF_MESHUDF_LUTS, then the module on sys.path (see the module docstring).  Raises MeshUDFError."""
    import importlib
    import importlib.util
    path = path or os.environ.get("DUDF_MESHUDF_LUTS")
    if path:
        if os.path.isdir(path):
            path = os.path.join(path, "_marching_cubes_lewiner_luts.py")
        if not os.path.exists(path):
            raise MeshUDFError(f"look-up tables: {path} does not exist")
        if path.endswith(".npz"):
            z = np.load(path)
            missing = [n for n in LUT_NAMES[3:] if n not in z.files]
            if missing:
                raise MeshUDFError(f"look-up tables: {path} lacks {missing[:4]}{' ...' if len(missing) > 4 else ''}")
            out = dict(EDGESREL)
            out.update({n: np.asarray(z[n], dtype=np.int8) for n in LUT_NAMES if n in z.files})
            return out
        spec = importlib.util.spec_from_file_location("_dudf_user_luts", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return _decode_module(mod)
    for name in ("_marching_cubes_lewiner_luts", "src.marching_cubes._marching_cubes_lewiner_luts",
                 "skimage.measure._marching_cubes_lewiner_luts"):
        try:
            mod = importlib.import_module(name)
        except Exception:
            continue
        if all(hasattr(mod, n) for n in LUT_NAMES[3:]):
            return _decode_module(mod)
    raise MeshUDFError("the Lewiner look-up tables are an INPUT of the MeshUDF extraction and are not shipped with this package: "
                       "pass luts={name: int8 array} or luts='<path>' (an .npz written by marching_cubes.save_luts_npz, or the "
                       "reference's src/marching_cubes/_marching_cubes_lewiner_luts.py), set DUDF_MESHUDF_LUTS to such a path, "
                       "or put that module on sys.path")


def load_reference_luts():
    """Kept for callers of round 2: the tables found without an explicit path (`load_luts()`)."""
    return load_luts()


def _pack_luts(luts):
    offs, dims, parts, o = [], [], [], 0
    for name in LUT_NAMES:
        a = np.ascontiguousarray(luts[name] if name in luts else EDGESREL[name], dtype=np.int8)
        if a.ndim > 3:
            raise MeshUDFError(f"look-up table {name}: more than three dimensions")
        shp = list(a.shape) + [1] * (3 - a.ndim)
        offs.append(o); dims.extend(shp); parts.append(a.ravel()); o += a.size
    return np.concatenate(parts), np.asarray(offs, dtype=np.int64), np.asarray(dims, dtype=np.int32)


def marching_cubes_udf(volume, grads, luts, avg_thresh=1.05, max_thresh=1.75):
    """The native entry point (reference `_marching_cubes_lewiner_cy.marching_cubes_udf(im, grads, luts, 1, 0, avg, max)`):
    (vertices (n, 3) float32 in x-y-z grid units, faces (3 m,) int32, normals (n, 3) float32 unit length, values (n,))."""
    volume = np.ascontiguousarray(volume, np.float32)
    grads = np.ascontiguousarray(grads, np.float32)
    if volume.ndim != 3 or grads.shape != volume.shape + (3,):
        raise ValueError("volume must be (nz, ny, nx) and grads (nz, ny, nx, 3)")
    data, offs, dims = _pack_luts(luts)
    lib = _lib()
    nz, ny, nx = volume.shape
    h = lib.dudf_meshudf_run(volume.ctypes.data, grads.ctypes.data, nz, ny, nx, data.ctypes.data, offs.ctypes.data,
                             dims.ctypes.data, len(LUT_NAMES), float(avg_thresh), float(max_thresh))
    if not h:
        raise MeshUDFError("dudf_meshudf_run failed (bad arguments or out of memory)")
    try:
        nv, nf = ctypes.c_longlong(), ctypes.c_longlong()
        lib.dudf_meshudf_sizes(h, ctypes.byref(nv), ctypes.byref(nf))
        v = np.empty((nv.value, 3), np.float32); f = np.empty((nf.value,), np.int32)
        n = np.empty((nv.value, 3), np.float32); vals = np.empty((nv.value,), np.float32)
        lib.dudf_meshudf_copy(h, v.ctypes.data, f.ctypes.data, n.ctypes.data, vals.ctypes.data)
    finally:
        lib.dudf_meshudf_free(h)
    # unit normals, computed like the reference's `Cell.get_normals` (:372-390): length in double, product stored as float
    nd = n.astype(np.float64)
    length = (nd * nd).sum(axis=1)
    scale = np.where(length > 0.0, 1.0 / np.sqrt(np.where(length > 0.0, length, 1.0)), length)
    n = (n * scale[:, None]).astype(np.float32)
    return v, f, n, vals


def udf_mc_lewiner(volume, grads, spacing=(1., 1., 1.), gradient_direction='descent', step_size=1, allow_degenerate=True,
                   use_classic=False, avg_thresh=1.05, max_thresh=1.75, mask=None, luts=None):
    """Reference `udf_mc_lewiner` (src/marching_cubes/_marching_cubes_lewiner.py:80-141): same argument checks, same
    output conventions (vertices in z-y-x order, faces flipped for 'descent', spacing applied).  `step_size` must be 1 and
    `mask` None / `allow_degenerate` True: the reference never calls it otherwise (src/render_mc.py:130-134)."""
    if not isinstance(volume, np.ndarray) or (volume.ndim != 3):
        raise ValueError('Input volume should be a 3D numpy array.')
    if volume.shape[0] < 2 or volume.shape[1] < 2 or volume.shape[2] < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    if len(spacing) != 3:
        raise ValueError("`spacing` must consist of three floats.")
    if int(step_size) != 1 or mask is not None or not allow_degenerate or use_classic:
        raise NotImplementedError("step_size != 1, mask, use_classic and allow_degenerate=False are not used by the reference's "
                                  "MeshUDF path and are not built")
    if luts is None or isinstance(luts, (str, os.PathLike)):
        luts = load_luts(luts)
    vertices, faces, normals, values = marching_cubes_udf(volume, grads, luts, avg_thresh, max_thresh)
    if not len(vertices):
        raise RuntimeError('No surface found at the given iso value.')
    vertices = np.fliplr(vertices)
    normals = np.fliplr(normals)
    faces = faces.reshape(-1, 3)
    if gradient_direction == 'descent':
        faces = np.fliplr(faces)
    elif not gradient_direction == 'ascent':
        raise ValueError("Incorrect input %s in `gradient_direction`, see docstring." % (gradient_direction))
    if not np.array_equal(spacing, (1, 1, 1)):
        vertices = vertices * np.r_[spacing]
    return vertices, faces, normals, values
