# coding: utf-8
"""Differential quantities the sphere tracer derives at ray hits — the query half of reference
src/render_st.py:42-65 (BASELINE config 4).  The ray marching / shading loop (:67-281) is outside this build's
scope."""
import weakref

from . import hip_ops
from .diff_operators import _source, gradient
from ._lib import DudfError


def compute_grad(inputs, outputs):
    """reference src/render_st.py:64-65"""
    return gradient(outputs, inputs)


def compute_normals_and_cd(inputs, outputs):
    """(pred_normals (1,N,3), principal directions (1,N,3,2) on the CPU) — reference src/render_st.py:57-62:
    eigh of the Hessian, normal = eigenvector of the largest eigenvalue, the other two as curvature directions."""
    model, coords = _source(outputs, inputs)
    x2 = coords.detach().reshape(-1, 3)
    _, _, _, _, V = hip_ops.query_frame(model.hip_cfg, model.flat_parameters(), x2)
    lead = coords.shape[:-1]
    normals = V[:, :, 2].reshape(lead + (3,))
    normals._dudf_src = (weakref.ref(model), coords)          # compute_curvature(inputs, normals) finds its way back
    normals._dudf_kind = "eig_normal"
    return normals, V[:, :, :2].reshape(lead + (3, 2)).detach().cpu()


def compute_curvature(inputs, normals, curvature='mean', device=None):
    """reference src/render_st.py:42-55: shape operator = jacobian(normals, inputs) — third derivatives of f, obtained
    here from third-order Taylor jets along combinations of the Hessian's eigenvectors (csrc/dudf_sweep.hip
    SWEEP_FWD_J) instead of autograd through eigh.  'mean' -> trace/2, 'gaussian' -> -det [[J, n],[n^T, 0]];
    (1,N,1) CPU tensors like the reference; anything else -> None.  The sign convention of `normals` is the one
    `compute_normals_and_cd` returned (the same eigh), so the caller's re-orientation (:104-108) applies unchanged."""
    if curvature not in ('mean', 'gaussian'):
        return None
    model, coords = _source(normals, inputs)
    x2 = coords.detach().reshape(-1, 3)
    _, _, mean, gauss, _ = hip_ops.query_curvature(model.hip_cfg, model.flat_parameters(), x2,
                                                   want_shape=(curvature == 'gaussian'))
    out = mean if curvature == 'mean' else gauss
    return out.detach().cpu()[None, ..., None]
