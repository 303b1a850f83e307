# coding: utf-8
"""ORACLE — test infrastructure only.  Never imported by the product path.

numpy restatement of the sphere-tracing loop of the reference (src/render_st.py:136-161 `propagate_rays`, :163-172
`grad_descent`; `inverse` src/inverses.py:3-21; `normalize` src/util.py:35-40) on top of oracle/dudf_oracle.py's forward
and input gradient.  Positions are float64, the network sees float32 copies, the step is float32 — as in the reference.
Pinned by tests/golden/g7_rays.npz (the same loop driven around the REFERENCE model).  A ray whose |step| sits within
rounding of the threshold may retire one iteration earlier or later than in another implementation; the tests compare
the bulk and bound the stragglers.
"""
import numpy as np

from . import dudf_oracle as O


def inverse(gt_mode, pred_df, alpha, min_step=0.01):
    pred_df = np.asarray(pred_df, dtype=np.float32)
    if gt_mode == "tanh":
        return np.where(pred_df < np.float32(1 / alpha), np.sqrt(pred_df / np.float32(alpha)), pred_df)
    if gt_mode == "siren":
        return np.where(pred_df > 0, pred_df, np.float32(min_step))
    out = np.where(pred_df > 0, np.sqrt(np.maximum(pred_df, 0)), np.float32(min_step))
    return (out / np.sqrt(np.float32(alpha))).astype(np.float32)


def _query(params, pts, want_grad, w0=30.0):
    x = pts.astype(np.float32).astype(np.float64)
    y, cache = O.forward(params, x, w0)
    g = O.input_gradient(params, cache, w0)[0] if want_grad else None
    return y.astype(np.float32), (None if g is None else g.astype(np.float32))


def propagate_rays(params, rays, t0, mask, gt_mode, alpha, surface_threshold, max_iterations):
    """In place on t0 (M,3 float64) and mask (M bool); returns (hits, iterations)."""
    hits = np.zeros_like(mask)
    it = 0
    while mask.sum() > 0 and it < max_iterations:
        udfs, _ = _query(params, t0[mask], False)
        steps = inverse(gt_mode, np.abs(udfs), alpha)
        t0[mask] += rays[mask] * steps[:, None]
        close = (udfs < surface_threshold) if gt_mode == "siren" else (np.abs(steps) < surface_threshold)
        inside = np.logical_and(np.all(t0[mask] > -1, axis=1), np.all(t0[mask] < 1, axis=1))
        hits[mask] += np.logical_and(close, inside)
        mask[mask] *= np.logical_and(np.logical_not(close), inside)
        it += 1
    return hits, it


def grad_descent(params, t0, hits, gt_mode, alpha, gd_steps):
    for _ in range(gd_steps):
        udfs, g = _query(params, t0[hits], True)
        steps = inverse(gt_mode, np.abs(udfs), alpha)
        gn = g / np.linalg.norm(g, axis=1, keepdims=True)
        t0[hits] -= gn * steps[:, None]
