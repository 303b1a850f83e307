# coding: utf-8
"""Deterministic, counter-based synthetic data for tests, fixtures and bench.

Everything here is a pure function of (seed, stream, index), so that
  * rank r of R can reproduce exactly its shard of one global batch,
  * tests on the GPU box regenerate the weights / batches that the golden
    fixtures were made from without the fixtures having to carry them.

The distributions mirror the reference, not its RNG stream:
  * SIREN parameters: first layer weight U(-1/n_in, 1/n_in), every later weight
    U(-sqrt(6/n_in)/w0, +sqrt(6/n_in)/w0)  (reference src/model.py:7-19,111-113),
    biases U(-1/sqrt(n_in), 1/sqrt(n_in)) (torch.nn.Linear default).
  * training batch: [on-surface | far | near] thirds as assembled by the
    reference sampler (reference src/dataset.py:14-70): on-surface points have
    sdf == 0 exactly and unit normals; off-surface points have zero normals.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, stream, start, count):
    """count doubles in [0,1) for counters start..start+count-1 of (seed, stream)."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        bits = _splitmix64(idx ^ key)
        bits = _splitmix64(bits + key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal01(seed, stream, start, count):
    """Standard normals by Box-Muller on two independent uniform streams."""
    u1 = uniform01(seed, 2 * stream + 1000003, start, count)
    u2 = uniform01(seed, 2 * stream + 1000004, start, count)
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)


def siren_layer_shapes(hidden, n_in=3, n_out=1):
    """[(out, in), ...] for every Linear of SIREN(n_in, n_out, hidden)."""
    dims = [n_in] + list(hidden) + [n_out]
    return [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]


def siren_params(hidden, seed=123, w0=30.0, n_in=3, n_out=1, dtype=np.float32):
    """List of (weight (out,in), bias (out,)) drawn like the reference initialises them."""
    params = []
    for li, (o, i) in enumerate(siren_layer_shapes(hidden, n_in, n_out)):
        bound_w = (1.0 / i) if li == 0 else (np.sqrt(6.0 / i) / w0)
        bound_b = 1.0 / np.sqrt(i)
        w = (uniform01(seed, 10 + 2 * li, 0, o * i) * 2.0 - 1.0) * bound_w
        b = (uniform01(seed, 11 + 2 * li, 0, o) * 2.0 - 1.0) * bound_b
        params.append((w.reshape(o, i).astype(dtype), b.astype(dtype)))
    return params


def flatten_params(params, dtype=np.float32):
    """state_dict order: weight then bias, layer by layer (row-major (out,in))."""
    return np.concatenate([np.concatenate([w.reshape(-1), b.reshape(-1)]) for w, b in params]).astype(dtype)


def unflatten_params(theta, hidden, n_in=3, n_out=1):
    out, off = [], 0
    for o, i in siren_layer_shapes(hidden, n_in, n_out):
        w = theta[off:off + o * i].reshape(o, i); off += o * i
        b = theta[off:off + o]; off += o
        out.append((w, b))
    assert off == theta.size
    return out


def training_batch(n_points, seed=123, step=0, lo=0, hi=None, dtype=np.float32):
    """Rows lo..hi-1 of the global synthetic batch of `n_points` points for `step`.

    Returns coords (n,3), normals (n,3), sdf (n,1).  Thirds are [on | far | near]
    by GLOBAL index, so any [lo,hi) window is the exact slice of the global batch.
    """
    hi = n_points if hi is None else hi
    n = hi - lo
    n_on = n_points // 3
    n_far = n_points // 3
    base = 1000 * step
    g = np.arange(lo, hi)
    coords = np.stack([uniform01(seed, base + 100 + c, lo, n) * 2.0 - 1.0 for c in range(3)], axis=1)
    nrm = np.stack([normal01(seed, base + 110 + c, lo, n) for c in range(3)], axis=1)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    far = uniform01(seed, base + 120, lo, n)
    far = np.maximum(far, 1e-6)                       # never exactly 0: 0 means "on surface"
    near = np.abs(normal01(seed, base + 121, lo, n)) * 0.01
    near = np.maximum(near, 1e-7)
    on = g < n_on
    is_far = (g >= n_on) & (g < n_on + n_far)
    sdf = np.where(on, 0.0, np.where(is_far, far, near))
    normals = np.where(on[:, None], nrm, 0.0)
    return coords.astype(dtype), normals.astype(dtype), sdf.reshape(-1, 1).astype(dtype)


def stratified_shard(n_points, rank, world):
    """Index set of rank `rank`: an equal slice of each third, so every rank keeps the
    on/far/near mix.  The union over ranks is a permutation of 0..n-1."""
    n_on = n_points // 3
    n_far = n_points // 3
    bounds = [(0, n_on), (n_on, n_on + n_far), (n_on + n_far, n_points)]
    parts = []
    for a, b in bounds:
        m = b - a
        s = a + (m * rank) // world
        e = a + (m * (rank + 1)) // world
        parts.append(np.arange(s, e))
    return np.concatenate(parts)
