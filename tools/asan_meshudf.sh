#!/bin/bash
# AddressSanitizer + UBSan run of the host-only MeshUDF library (CPU; GPU sanitizers are not available on the pool):
# the golden cases bit for bit plus non-cubic and minimal grids.   bash tools/asan_meshudf.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/dbg"
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o "$R/dbg/libdudf_meshudf_asan.so" "$R/diffudf_amd/csrc/dudf_meshudf.cpp"
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
python3 - "$R" <<'PY'
import ctypes, sys
R = sys.argv[1]
sys.path.insert(0, R)
import numpy as np
import diffudf_amd.marching_cubes as m
m._lib()                                                    # bindings set up on the regular build ...
lib = ctypes.CDLL(R + "/dbg/libdudf_meshudf_asan.so")       # ... then the same signatures on the sanitized one
for name in ("dudf_meshudf_run", "dudf_meshudf_sizes", "dudf_meshudf_copy", "dudf_meshudf_free"):
    getattr(lib, name).argtypes = getattr(m._LIB, name).argtypes
    getattr(lib, name).restype = getattr(m._LIB, name).restype
m._LIB = lib
G = np.load(R + "/tests/golden/g10_meshudf.npz")
LUTS = {k[4:]: G[k] for k in G.files if k.startswith("lut_")}
for tag in map(str, G["cases"]):
    udf, g = G[tag + "_udf"], G[tag + "_grads"]
    n = udf.shape[0]
    v, f, nn, val = m.udf_mc_lewiner(udf, g, spacing=[2.0 / (n - 1)] * 3, luts=LUTS)
    assert np.array_equal(v, G[tag + "_vertices"]) and np.array_equal(f, G[tag + "_faces"]), tag
rng = np.random.default_rng(0)
for shp in ((2, 2, 2), (3, 5, 4), (9, 17, 12), (31, 8, 19)):
    u = rng.random(shp).astype(np.float32) * 0.05
    gg = rng.standard_normal(shp + (3,)).astype(np.float32)
    try:
        m.udf_mc_lewiner(u, gg, luts=LUTS)
    except RuntimeError:
        pass
print("MeshUDF library: ASan + UBSan clean on", len(G["cases"]), "golden cases and 4 ragged grids")
PY
