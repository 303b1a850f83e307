# coding: utf-8
"""`src.dataset` of the reference, served by diffudf_amd.dataset (see src/__init__.py)."""
from diffudf_amd.dataset import *  # noqa: F401,F403
from diffudf_amd import dataset as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
