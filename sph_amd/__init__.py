"""MI355X-native SPH particle-pair path: HIP kernels behind a C ABI (include/sph_amd.h)."""
from . import capi  # noqa: F401
from .capi import Sim, SphError  # noqa: F401
