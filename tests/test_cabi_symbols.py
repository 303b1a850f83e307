# coding: utf-8
"""CPU: the C-ABI library loads and exports every symbol include/dudf_hip.h declares."""
import ctypes
import os
import re

from diffudf_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, "include", "dudf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dudf_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    names = declared_symbols()
    assert "dudf_loss_backward" in names and "dudf_query" in names
    assert sorted(_lib.SYMBOLS) == names


def test_library_loads_and_exports_everything():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dudf_version()


def test_host_only_calls():
    """theta count / workspace size are pure host arithmetic: callable without a GPU."""
    lib = _lib.load()
    cfg = _lib.NetCfg(3, 8, 256, 30.0)
    assert lib.dudf_theta_count(ctypes.byref(cfg)) == 461825
    nb = lib.dudf_workspace_bytes(ctypes.byref(cfg), 29970)
    np_ = (29970 + 63) // 64 * 64
    assert nb >= 7 * 8 * 256 * np_ * 4
    nbh = lib.dudf_workspace_bytes_hess(ctypes.byref(cfg), 29970, 9990)       # on-surface third on the Hessian path
    cols = (4 * 9990 + 63) // 64 * 64 + (19980 + 63) // 64 * 64
    assert nbh >= 8 * 8 * 256 * cols * 4
    assert lib.dudf_workspace_bytes_hess(ctypes.byref(cfg), 10, 11) == 0
    bad = _lib.NetCfg(3, 8, 100, 30.0)
    assert lib.dudf_theta_count(ctypes.byref(bad)) == -1
    assert lib.dudf_workspace_bytes(ctypes.byref(bad), 10) == 0
