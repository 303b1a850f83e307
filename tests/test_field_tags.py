# coding: utf-8
"""CPU: which tensors `gradient` / `hessian` / `laplace` / `jacobian` accept as "the model output" (round-2 advice).

The reference's operators differentiate ANY y of the autograd graph (src/diff_operators.py:187-227).  The HIP path evaluates
named fields of the network, so a FUNCTION of the output must raise instead of silently returning grad f."""
import weakref

import pytest
import torch

from diffudf_amd import diff_operators as D
from diffudf_amd._lib import DudfError


class _Model:                                            # stands in for the SIREN: the tag logic never calls it
    pass


def _pair(n=7):
    m = _Model()
    x = torch.zeros(1, n, 3)
    x._dudf_model = weakref.ref(m)
    y = D.tag_field(torch.zeros(1, n, 1), "value", weakref.ref(m), x)
    return m, y, x


def test_shape_only_views_are_still_the_output():
    m, y, x = _pair()
    for v in (y.squeeze(-1), y.reshape(-1), y.view(-1, 1), y.detach(), y.clone(), y.contiguous(), y[..., 0], y.flatten(),
              y.squeeze(-1).unsqueeze(-1), y.float()):
        model, coords = D._source(v, x)
        assert model is m and D._kind(v, coords) == "value", v._dudf_kind


@pytest.mark.parametrize("fn", [lambda y: 2 * y, lambda y: y * 2, lambda y: y.abs(), lambda y: torch.tanh(y),
                                lambda y: y ** 2, lambda y: y + 1.0, lambda y: (2 * y).squeeze(-1),
                                lambda y: y.squeeze(-1) * 3, lambda y: -y, lambda y: torch.sin(y.reshape(-1))])
def test_functions_of_the_output_raise(fn):
    m, y, x = _pair()
    v = fn(y)
    for op in (D.gradient, D.hessian, D.laplace, D.jacobian):
        with pytest.raises(DudfError):
            op(v, x)


def test_untagged_and_partial_tensors_raise():
    m, y, x = _pair()
    with pytest.raises(DudfError):
        D.hessian(torch.zeros(1, 7, 1), x)               # not from forward(): before round 3 the x fallback accepted it
    with pytest.raises(DudfError):
        D.gradient(y[:, :3], x)                          # a slice of SOME points is not the output
    g = D.tag_field(torch.zeros(1, 7, 3), "grad", weakref.ref(m), x)
    with pytest.raises(DudfError):
        D.gradient(g[..., 0], x)                         # a component of the gradient: hessian(y, x) is the way
    with pytest.raises(DudfError):
        D.hessian(g, x)
