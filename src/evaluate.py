# coding: utf-8
"""`src.evaluate` of the reference, served by diffudf_amd.evaluate (see src/__init__.py)."""
from diffudf_amd.evaluate import *  # noqa: F401,F403
from diffudf_amd import evaluate as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
