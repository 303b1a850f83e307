# coding: utf-8
"""`src.render_st` of the reference, served by diffudf_amd.render_st (see src/__init__.py)."""
from diffudf_amd.render_st import *  # noqa: F401,F403
from diffudf_amd import render_st as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
