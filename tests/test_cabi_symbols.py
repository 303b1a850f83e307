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
    # seven stash arrays per layer and column: all fp32 (mask 0), R, E and C at 3 bytes per value (mask 6), or all seven at 3 bytes
    # (mask 7, the default)
    per_value = {0: 7 * 4, 6: 4 * 4 + 3 * 3, 7: 7 * 3}[lib.dudf_stash_mode(ctypes.byref(cfg), 29970, 0)]
    assert nb >= per_value * 8 * 256 * np_
    nbh = lib.dudf_workspace_bytes_hess(ctypes.byref(cfg), 29970, 9990)       # on-surface third on the Hessian path
    cols = (4 * 9990 + 63) // 64 * 64 + (19980 + 63) // 64 * 64
    assert nbh >= (per_value + 4) * 8 * 256 * cols                            # + ZS (fp32)
    assert lib.dudf_stash_mode(ctypes.byref(_lib.NetCfg(3, 8, 512, 30.0)), 1000, 0) in (0, 6)   # 512-wide layers relay S, Q, A, Z through the stash: those stay fp32
    assert lib.dudf_stash_mode(ctypes.byref(_lib.NetCfg(3, 8, 128, 30.0)), 1000, 0) == 0
    assert lib.dudf_stash_mode(ctypes.byref(_lib.NetCfg(3, 8, 100, 30.0)), 1000, 0) == -1
    assert lib.dudf_workspace_bytes_hess(ctypes.byref(cfg), 10, 11) == 0
    # the format is a property of (cfg, n): beyond 2^22 columns the layout falls back to fp32, and dudf_stash_mode says so
    assert lib.dudf_stash_mode(ctypes.byref(cfg), 5_000_000, 0) == 0
    # ABI handshake + options (host-only): unknown names / values are refused, reset restores the defaults
    assert lib.dudf_abi_version() == _lib.ABI_VERSION
    v = ctypes.c_int(-1)
    assert lib.dudf_get_option(b"stash", ctypes.byref(v)) == 0 and v.value == 7
    assert lib.dudf_set_option(b"stash", 0) == 0 and lib.dudf_stash_mode(ctypes.byref(cfg), 29970, 0) == 0
    assert lib.dudf_workspace_bytes(ctypes.byref(cfg), 29970) > nb          # fp32 stash: a larger workspace
    assert lib.dudf_set_option(b"stash", 3) != 0 and lib.dudf_set_option(b"nonsense", 1) != 0
    assert lib.dudf_set_option(b"wgrad_max_workgroups", 7) != 0 and lib.dudf_set_wgrad_max_workgroups(240) == 0
    assert lib.dudf_get_option(b"wgrad_max_workgroups", ctypes.byref(v)) == 0 and v.value == 240
    assert lib.dudf_reset_options() == 0 and lib.dudf_stash_mode(ctypes.byref(cfg), 29970, 0) == 7
    assert lib.dudf_get_option(b"wgrad_max_workgroups", ctypes.byref(v)) == 0 and v.value == 256
    bad = _lib.NetCfg(3, 8, 100, 30.0)
    assert lib.dudf_theta_count(ctypes.byref(bad)) == -1
    assert lib.dudf_workspace_bytes(ctypes.byref(bad), 10) == 0


def test_adam_schedule_table_is_the_host_path_scalars():
    """dudf_adam_schedule (host only): row i = (lr_i / (1 - beta1^t), sqrt(1 - beta2^t)), t = first_step + i, in double, rounded
    once — what dudf_adam_step passes to its kernel; over the reference recipe's whole schedule (train.py:179-191: 1000 warm-up
    epochs, 1000 at lr_s1, 1000 on the cosine)."""
    import math
    import numpy as np
    lib = _lib.load()
    lrs = [1e-4] * 1000 + [1e-5] * 1000 + [0.5 * (math.cos(e / 1000 * math.pi) + 1) * 1e-7 for e in range(2000, 3000)]
    lr = np.array(lrs, dtype=np.float64)
    out = np.empty((lr.size, 2), dtype=np.float32)
    assert lib.dudf_adam_schedule(lr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), lr.size, 1, 0.9, 0.999, ctypes.c_void_p(out.ctypes.data)) == 0
    t = np.arange(1, lr.size + 1)
    want0 = np.array([np.float32(l / (1.0 - 0.9 ** int(k))) for l, k in zip(lrs, t)])
    want1 = np.array([np.float32(math.sqrt(1.0 - 0.999 ** int(k))) for k in t])
    assert np.array_equal(out[:, 0], want0) and np.array_equal(out[:, 1], want1)
    assert lib.dudf_adam_schedule(lr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 3, 0, 0.9, 0.999, ctypes.c_void_p(out.ctypes.data)) != 0   # steps are 1-based


def test_stash_arrays_sit_on_the_cache_line_grid():
    """Every per-column array starts on a 256-byte boundary of the workspace and its row / layer strides are multiples of
    256 bytes: a lane quarter's 256-byte segment of a stash row is then exactly two 128-byte lines.  (A 64-byte offset cost
    9-12 % of HBM traffic in all four sweeps before it was noticed: DESIGN.md §5c, round 3.)"""
    lib = _lib.load()
    out = (ctypes.c_int64 * 10)()
    for hidden, layers in ((256, 8), (512, 8), (128, 3), (64, 2), (32, 1)):
        cfg = _lib.NetCfg(3, layers, hidden, 30.0)
        for n, nh in ((100000, 0), (29970, 9990), (125000, 0), (98304, 0), (131072, 0), (1, 0), (17, 17), (1000003, 333)):
            assert lib.dudf_debug_stash_layout(ctypes.byref(cfg), n, nh, out) == 0
            vals = list(out)
            assert all(v % 256 == 0 for v in vals), (hidden, layers, n, nh, vals)
            assert vals[8] >= 16 * n and vals[9] == vals[8] * hidden // 4
            assert vals[8] % (1 << 15) != 0, "row stride a large power of two: rows share HBM channels"
    assert lib.dudf_debug_stash_layout(ctypes.byref(_lib.NetCfg(3, 8, 100, 30.0)), 10, 0, out) != 0


def test_replayed_steps_follow_the_schedule_on_the_host():
    """`diffudf_amd.optim.Adam.replayed()`: a graph replay takes its learning rate from the device table; the host-side view —
    `param_groups[0]['lr']`, what a scheduler or a checkpoint reads — has to agree with it at every replayed step, and a replay past the
    end of the schedule is an error, not a silent no-op.  Host only: no kernel is launched."""
    import pytest
    from diffudf_amd.model import SIREN
    from diffudf_amd.optim import Adam
    m = SIREN(n_in_features=3, n_out_features=1, hidden_layer_config=[32] * 2, w0=30)
    opt = Adam(m.parameters(), lr=1e-4, model=m)
    opt.use_schedule([1e-4, 1e-4, 5e-5])
    opt.replayed()
    opt.param_groups[0]["lr"] = 2e-4                     # somebody edits the rate behind the table's back
    with pytest.raises(RuntimeError, match="scheduled"):
        opt.replayed()
    opt.param_groups[0]["lr"] = 1e-4
    opt.replayed()
    opt.param_groups[0]["lr"] = 5e-5
    opt.replayed()
    assert opt._t == 3
    with pytest.raises(RuntimeError, match="past a schedule"):
        opt.replayed()
