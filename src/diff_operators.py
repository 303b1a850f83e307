# coding: utf-8
"""`src.diff_operators` of the reference, served by diffudf_amd.diff_operators (see src/__init__.py)."""
from diffudf_amd.diff_operators import *  # noqa: F401,F403
from diffudf_amd import diff_operators as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
