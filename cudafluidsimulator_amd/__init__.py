"""MI355X-native SPH step path behind the reference's Simulator interface.

The compute lives in libsph_hip.so (hand-written gfx950 HIP kernels behind the
C-ABI of include/sph_c_api.h).  This package is the Python mirror of that
boundary -- `Simulator`, `Settings`, `Times` with the reference's names and
semantics (src/simulator.h:19-74, src/times.h:5-35) -- plus the multi-GPU slab
driver.  There is NO CPU fallback: importing the binding without the built
library, or creating a Simulator without a GPU, raises.
"""
from ._lib import (SphError, SphKernelTimes, SphOptions, SphSettings, SphTimes,  # noqa: F401
                   library_path, load_library)
from .simulator import Settings, Simulator, Times, default_settings  # noqa: F401

__all__ = ["Simulator", "Settings", "Times", "default_settings", "SphError",
           "load_library", "library_path"]
