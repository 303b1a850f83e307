# coding: utf-8
"""`src.loss_functions` of the reference, served by diffudf_amd.loss_functions (see src/__init__.py)."""
from diffudf_amd.loss_functions import *  # noqa: F401,F403
from diffudf_amd import loss_functions as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
