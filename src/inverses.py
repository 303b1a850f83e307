# coding: utf-8
"""`src.inverses` of the reference, served by diffudf_amd.inverses (see src/__init__.py)."""
from diffudf_amd.inverses import *  # noqa: F401,F403
from diffudf_amd import inverses as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
