# coding: utf-8
"""`src.render_mc` of the reference, served by diffudf_amd.render_mc (see src/__init__.py)."""
from diffudf_amd.render_mc import *  # noqa: F401,F403
from diffudf_amd import render_mc as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
