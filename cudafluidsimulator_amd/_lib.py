"""ctypes binding of libsph_hip.so -- declarations mirror include/sph_c_api.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libsph_hip.so"

SPH_MATH_STRICT, SPH_MATH_FAST = 0, 1
SPH_SWEEP_LIST, SPH_SWEEP_DIRECT, SPH_SWEEP_LDS = 0, 1, 2
SWEEPS = {"list": 0, "direct": 1, "lds": 2, "linked": 3}
SPH_FLAG_COUNT_PAIRS, SPH_FLAG_STORE_FORCE, SPH_FLAG_NO_READBACK = 1, 2, 4
SPH_FLAG_EXTERNAL_STATE = 8
SPH_FLAG_MAPPED_POSITIONS = 16

# every symbol include/sph_c_api.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "sph_default_settings", "sph_create", "sph_destroy", "sph_setup", "sph_upload_state",
    "sph_step", "sph_apply_click", "sph_positions_host", "sph_download_state",
    "sph_download_force", "sph_download_grid", "sph_sync", "sph_num_particles",
    "sph_get_kernel_times", "sph_last_error", "sph_phase_grid", "sph_phase_density",
    "sph_phase_force", "sph_phase_readback", "sph_sort_check", "sph_build_info",
    "sph_set_stream", "sph_bind_buffers", "sph_slab_sort", "sph_slab_partition", "sph_slab_copy_segments", "sph_slab_density",
    "sph_slab_force", "sph_initial_positions", "sph_save_state", "sph_load_state",
    "sph_debug_counters", "sph_get_stream", "sph_slab_partition_async", "sph_slab_sort_async",
    "sph_slab_patch_halo", "sph_slab_force_ranges", "sph_num_table_cells", "sph_slab_apply_click", "sph_slab_records",
]


class SphError(RuntimeError):
    pass


class SphSettings(C.Structure):
    _fields_ = [("randomInit", C.c_uint8), ("pad_", C.c_uint8 * 3),
                ("numParticles", C.c_int32), ("h", C.c_float),
                ("v_kernel_coeff", C.c_float), ("d_kernel_coeff", C.c_float),
                ("boxDim", C.c_float), ("numCellsPerDim", C.c_float),
                ("timestep", C.c_float)]


class SphTimes(C.Structure):
    _fields_ = [("buildGrid", C.c_double), ("sphUpdate", C.c_double),
                ("memcpy", C.c_double), ("iters", C.c_int32)]


class SphOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32),
                ("math_mode", C.c_int32), ("sweep", C.c_int32), ("flags", C.c_int32),
                ("capacity", C.c_int32), ("key_order", C.c_int32)]


class SphKernelTimes(C.Structure):
    _fields_ = [("hash", C.c_double), ("sort", C.c_double), ("gather", C.c_double),
                ("density", C.c_double), ("force", C.c_double), ("readback", C.c_double),
                ("pair_tests", C.c_uint64), ("steps", C.c_int64), ("pair_hits", C.c_uint64)]


def library_path():
    # SPH_LIB_PATH: A/B experiments with differently-built libraries (same ABI)
    return os.environ.get("SPH_LIB_PATH") or os.path.join(_HERE, _LIB_NAME)


_lib = None


def load_library():
    """Load libsph_hip.so; fail loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise SphError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C cudafluidsimulator_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    i32p = C.POINTER(C.c_int32)
    hp = C.c_void_p
    L.sph_default_settings.argtypes = [C.POINTER(SphSettings), C.c_int, C.c_int]
    L.sph_initial_positions.argtypes = [C.POINTER(SphSettings), fp]
    L.sph_create.argtypes = [C.POINTER(SphSettings), C.POINTER(SphOptions), C.POINTER(hp)]
    L.sph_destroy.argtypes = [hp]
    L.sph_destroy.restype = None
    L.sph_setup.argtypes = [hp]
    L.sph_upload_state.argtypes = [hp, fp, fp, C.c_int]
    L.sph_step.argtypes = [hp, C.POINTER(SphTimes)]
    L.sph_apply_click.argtypes = [hp, C.c_int, C.c_int]
    L.sph_positions_host.argtypes = [hp]
    L.sph_positions_host.restype = fp
    L.sph_download_state.argtypes = [hp, fp, fp, fp, fp]
    L.sph_download_force.argtypes = [hp, fp]
    L.sph_download_grid.argtypes = [hp, u32p, u32p, i32p]
    L.sph_sync.argtypes = [hp]
    L.sph_debug_counters.argtypes = [hp, C.POINTER(C.c_uint64)]
    L.sph_save_state.argtypes = [hp, C.c_char_p]
    L.sph_load_state.argtypes = [hp, C.c_char_p]
    L.sph_num_particles.argtypes = [hp]
    L.sph_num_table_cells.argtypes = [hp]
    L.sph_get_kernel_times.argtypes = [hp, C.POINTER(SphKernelTimes), C.c_int]
    L.sph_last_error.argtypes = [hp]
    L.sph_last_error.restype = C.c_char_p
    for name in ("sph_phase_grid", "sph_phase_density", "sph_phase_force",
                 "sph_phase_readback"):
        getattr(L, name).argtypes = [hp]
    L.sph_sort_check.argtypes = [C.c_int, u32p, C.c_int, C.c_int, u32p, u32p]
    L.sph_build_info.restype = C.c_char_p
    L.sph_set_stream.argtypes = [hp, C.c_void_p]
    L.sph_bind_buffers.argtypes = [hp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.sph_slab_sort.argtypes = [hp, C.c_int, C.c_int, C.c_int, u32p, C.c_int, i32p]
    L.sph_slab_copy_segments.argtypes = [hp, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         i32p, i32p]
    L.sph_slab_partition.argtypes = [hp, C.c_int, C.c_int, C.c_int, u32p, C.c_int, i32p, C.c_void_p]
    L.sph_slab_density.argtypes = [hp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.sph_slab_force.argtypes = [hp, C.c_int, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L
