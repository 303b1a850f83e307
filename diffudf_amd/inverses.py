# coding: utf-8
"""Distance from the hyperbolic-scaled field value — reference src/inverses.py:3-21 (numpy, host side)."""
import numpy as np


def inv_tanh(pred_df, alpha, min_step=0.01):
    """t(d) = d tanh(alpha d) ~ alpha d^2 near 0: invert that branch below 1/alpha, identity above."""
    pred_df = np.asarray(pred_df)
    return np.where(pred_df < 1.0 / alpha, np.sqrt(pred_df / alpha), pred_df)


def inv_squared(pred_df, alpha, min_step=0.01):
    pred_df = np.asarray(pred_df)
    out = np.full_like(pred_df, min_step)
    pos = pred_df > 0
    out[pos] = np.sqrt(pred_df[pos])
    return out / np.sqrt(alpha)


def inv_siren(pred_df, alpha, min_step=0.01):
    pred_df = np.asarray(pred_df)
    return np.where(pred_df > 0, pred_df, np.full_like(pred_df, min_step))


def inverse(gt_mode, pred_df, alpha, min_step=0.01):
    table = {'siren': inv_siren, 'squared': inv_squared, 'tanh': inv_tanh}
    return table[gt_mode](pred_df, alpha, min_step)
