# coding: utf-8
"""Differential quantities the sphere tracer derives at ray hits — the query half of reference
src/render_st.py:42-65 (BASELINE config 4).  The ray marching / shading loop (:67-281) is outside this build's
scope."""
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
    return V[:, :, 2].reshape(lead + (3,)), V[:, :, :2].reshape(lead + (3, 2)).detach().cpu()


def compute_curvature(inputs, normals, curvature='mean', device=None):
    """reference src/render_st.py:42-55: the Jacobian of the eigenvector field (third derivatives of f).  The
    directional third-order sweep (last hard part) is not built; no autograd fallback by design."""
    raise DudfError("compute_curvature: needs the third-order directional sweep, which is not built yet")
