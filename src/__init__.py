# coding: utf-8
"""Import-path shim: a user of the reference writes `from src.model import SIREN`,
`from src.loss_functions import loss_s1`, ...  These modules re-export the MI355X-native
implementations in `diffudf_amd/` under the reference's module names."""
