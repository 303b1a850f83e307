#!/usr/bin/env python
# coding: utf-8
"""TEST INFRASTRUCTURE — builds the reference's own MeshUDF marching-cubes extension as the checker of
diffudf_amd/csrc/dudf_meshudf.cpp.

    python oracle/build_ref.py            -> oracle/_ref/_marching_cubes_lewiner_cy.<abi>.so

The sources stay where they lie (/root/reference/src/marching_cubes/_marching_cubes_lewiner_cy.pyx, one file, numpy C API
only): Cython writes the generated C++ into oracle/_ref/, g++ compiles it there.  Nothing is copied into the repository;
oracle/_ref/ is git-ignored.  The prebuilt .so files that ship with the reference target the numpy 1.x ABI and do not load
with this image's numpy 2.2 — hence the rebuild.  Only tests/ (tests/test_meshudf.py, tests/golden/make_golden.py g10) import
the result; the product never does.  Absent reference, Cython or g++: returns False, nothing is built.
"""
import os
import subprocess
import sys
import sysconfig

REF_PYX = "/root/reference/src/marching_cubes/_marching_cubes_lewiner_cy.pyx"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def build(force=False):
    if not os.path.exists(REF_PYX):
        return False
    try:
        import Cython  # noqa: F401
        import numpy as np
    except ImportError:
        return False
    os.makedirs(OUT_DIR, exist_ok=True)
    cpp = os.path.join(OUT_DIR, "_marching_cubes_lewiner_cy.cpp")
    so = os.path.join(OUT_DIR, "_marching_cubes_lewiner_cy" + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(so) and not force and os.path.getmtime(so) >= os.path.getmtime(REF_PYX):
        return True
    subprocess.run([sys.executable, "-m", "cython", "--cplus", "-3", "-o", cpp, REF_PYX], check=True)
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-w", "-I" + sysconfig.get_paths()["include"],
                    "-I" + np.get_include(), "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION", cpp, "-o", so], check=True)
    os.remove(cpp)                                          # 2.7 MB of generated code: not needed once compiled
    return True


def load():
    """(reference wrapper module, its LUT dict) with the rebuilt extension; None when unavailable."""
    if not build():
        return None
    ref_dir = os.path.dirname(REF_PYX)
    sys.path.insert(0, OUT_DIR)                             # the rebuilt extension must win over the reference's prebuilt ones
    if ref_dir not in sys.path:
        sys.path.append(ref_dir)
    import _marching_cubes_lewiner as wrapper
    return wrapper


if __name__ == "__main__":
    print("built" if build(force="--force" in sys.argv) else "reference / Cython not available: nothing built")
