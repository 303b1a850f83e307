# coding: utf-8
"""`src.util` of the reference, served by diffudf_amd.util (see src/__init__.py)."""
from diffudf_amd.util import *  # noqa: F401,F403
from diffudf_amd import util as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
