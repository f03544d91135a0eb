"""adaptiveviscositysolver_amd -- MI355X-native adaptive variational viscosity hot path.

Product code: the C ABI (include/avs.h -> libavs_hip.so, hand-written HIP for gfx950) plus this
thin host layer.  No CPU fallback: `capi.load()` raises when the HIP library is missing.
"""
from . import capi, scenes  # noqa: F401
from .solver import DevicePrepass, ViscositySolve, pcg_csr  # noqa: F401
