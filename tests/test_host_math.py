# coding: utf-8
"""CPU: the sin/cos pair of diffudf_amd/csrc/dudf_math.h (host build) against fp64."""
import ctypes
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "diffudf_amd", "libdudf_hostmath.so")


def test_sincos_absolute_error():
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(LIB)
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-60, 60, 2_000_000), rng.uniform(-1e4, 1e4, 1_000_000),
                        rng.uniform(-1, 1, 500_000), np.array([0.0, -0.0, np.pi / 2, np.pi, 47.123])]).astype(np.float32)
    s = np.empty_like(x); c = np.empty_like(x)
    P = ctypes.POINTER(ctypes.c_float)
    lib.dudf_host_sincos(x.ctypes.data_as(P), s.ctypes.data_as(P), c.ctypes.data_as(P), ctypes.c_long(x.size))
    xs = x.astype(np.float64)
    assert np.abs(s - np.sin(xs)).max() < 1.2e-7
    assert np.abs(c - np.cos(xs)).max() < 1.2e-7
