# coding: utf-8
"""`src.model` of the reference, served by diffudf_amd.model (see src/__init__.py)."""
from diffudf_amd.model import *  # noqa: F401,F403
from diffudf_amd import model as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
