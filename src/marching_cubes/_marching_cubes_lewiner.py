# coding: utf-8
"""Import boundary of the reference's MeshUDF wrapper (`sys.path.append('src/marching_cubes'); from
_marching_cubes_lewiner import udf_mc_lewiner`, reference src/render_mc.py:15-16): the host C++ build."""
from diffudf_amd.marching_cubes import *  # noqa: F401,F403
from diffudf_amd.marching_cubes import udf_mc_lewiner, marching_cubes_udf, load_reference_luts  # noqa: F401
