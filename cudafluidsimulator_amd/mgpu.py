"""ctypes binding of libsph_mgpu.so (include/sph_mgpu.h): the in-process C++ multi-GPU
driver -- z-slabs, one per MI355X, RCCL send/recv of the one-cell halo.  The data path is
C++ + HIP + RCCL; this module only creates the object and calls step().

`MultiGpuSimulator` mirrors `Simulator` (setup / simulate / simulateAndTime /
getPosition).  Two ways to run it:
  * one process drives all GPUs:        MultiGpuSimulator(settings, world=N)
  * one process per GPU (torchrun):     MultiGpuSimulator(settings, world=N, rank=r,
                                                          unique_id=<128 bytes from rank 0>)
transport="loopback" steps N slabs on ONE device (tests, one-GPU box).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import SphError, SphSettings, SphTimes, load_library

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LOCAL = 8
TRANSPORTS = {"loopback": 0, "rccl": 1, "rccl_self": 2, "mailbox": 3, "streams": 4}

# every symbol include/sph_mgpu.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "sph_mgpu_unique_id", "sph_mgpu_create", "sph_mgpu_destroy", "sph_mgpu_setup",
    "sph_mgpu_upload_state", "sph_mgpu_step", "sph_mgpu_step_phase", "sph_mgpu_positions_host",
    "sph_mgpu_download_state",
    "sph_mgpu_sync", "sph_mgpu_get_stats", "sph_mgpu_last_error", "sph_mgpu_queue_click",
]


class SphMgpuOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("world", C.c_int32), ("rank_begin", C.c_int32),
                ("rank_count", C.c_int32), ("transport", C.c_int32), ("devices", C.c_int32 * MAX_LOCAL),
                ("sweep", C.c_int32), ("math_mode", C.c_int32), ("face_capacity", C.c_int32),
                ("slab_capacity", C.c_int32), ("recut_every", C.c_int32)]


class SphMgpuStats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("host_syncs", C.c_int64), ("overflow_rounds", C.c_int64),
                ("recuts", C.c_int64), ("local_slabs", C.c_int32), ("owned", C.c_int32 * MAX_LOCAL),
                ("kernel_s", C.c_double * MAX_LOCAL), ("grid_s", C.c_double * MAX_LOCAL),
                ("density_s", C.c_double * MAX_LOCAL), ("force_s", C.c_double * MAX_LOCAL)]


def library_path():
    return os.environ.get("SPH_MGPU_LIB_PATH") or os.path.join(_HERE, "libsph_mgpu.so")


_lib_mgpu = None


def load_mgpu_library():
    """Load libsph_mgpu.so (and libsph_hip.so first); fail loudly if it is missing."""
    global _lib_mgpu
    if _lib_mgpu is not None:
        return _lib_mgpu
    load_library()
    path = library_path()
    if not os.path.exists(path):
        raise SphError(f"{path} is missing: build it with `make -C cudafluidsimulator_amd/csrc`. "
                       "There is no CPU fallback.")
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    hp = C.c_void_p
    L.sph_mgpu_unique_id.argtypes = [C.c_void_p]
    L.sph_mgpu_create.argtypes = [C.POINTER(SphSettings), C.POINTER(SphMgpuOptions), C.c_void_p, C.POINTER(hp)]
    L.sph_mgpu_destroy.argtypes = [hp]
    L.sph_mgpu_destroy.restype = None
    L.sph_mgpu_setup.argtypes = [hp]
    L.sph_mgpu_upload_state.argtypes = [hp, fp, fp, C.c_int]
    L.sph_mgpu_step.argtypes = [hp, C.POINTER(SphTimes)]
    L.sph_mgpu_step_phase.argtypes = [hp, C.c_int, C.POINTER(SphTimes)]
    L.sph_mgpu_positions_host.argtypes = [hp]
    L.sph_mgpu_positions_host.restype = fp
    L.sph_mgpu_download_state.argtypes = [hp, fp, fp, fp, C.POINTER(C.c_int)]
    L.sph_mgpu_sync.argtypes = [hp]
    L.sph_mgpu_queue_click.argtypes = [hp, C.c_int, C.c_int]
    L.sph_mgpu_get_stats.argtypes = [hp, C.POINTER(SphMgpuStats), C.c_int]
    L.sph_mgpu_last_error.argtypes = [hp]
    L.sph_mgpu_last_error.restype = C.c_char_p
    _lib_mgpu = L
    return L


def unique_id():
    """128 opaque bytes (ncclUniqueId): made on rank 0, handed to every rank."""
    buf = (C.c_char * 128)()
    if load_mgpu_library().sph_mgpu_unique_id(buf):
        raise SphError("sph_mgpu_unique_id failed")
    return bytes(buf)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class MultiGpuSimulator:
    def __init__(self, settings, world, transport="rccl", rank=None, devices=None, unique_id=None,
                 sweep="list", math="strict", face_capacity=0, slab_capacity=0, recut_every=0):
        self.settings = settings
        self._L = load_mgpu_library()
        self._h = C.c_void_p()
        o = SphMgpuOptions()
        o.struct_size = C.sizeof(SphMgpuOptions)
        o.world = world
        if rank is None:  # this process drives every slab
            o.rank_begin, o.rank_count = 0, world
            devs = list(devices) if devices is not None else (
                [0] * world if transport != "rccl" else list(range(world)))
        else:             # one process per GPU
            o.rank_begin, o.rank_count = rank, 1
            devs = list(devices) if devices is not None else [0]
        for k, d in enumerate(devs[:MAX_LOCAL]):
            o.devices[k] = d
        o.transport = TRANSPORTS[transport]
        o.sweep = _lib.SWEEPS[sweep]
        o.math_mode = _lib.SPH_MATH_FAST if math == "fast" else _lib.SPH_MATH_STRICT
        o.face_capacity, o.slab_capacity, o.recut_every = face_capacity, slab_capacity, recut_every
        uid = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        rc = self._L.sph_mgpu_create(C.byref(settings), C.byref(o), uid, C.byref(self._h))
        if rc:
            self._h = C.c_void_p()
            raise SphError(f"sph_mgpu_create failed ({rc}): {self._L.sph_mgpu_last_error(None).decode()}")
        self.mouseClicked = False
        self.clickCoords = (0, 0)

    def _check(self, rc, what):
        if rc:
            raise SphError(f"{what} failed ({rc}): {self._L.sph_mgpu_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.sph_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n(self):
        return self.settings.numParticles

    def setup(self):
        self._check(self._L.sph_mgpu_setup(self._h), "sph_mgpu_setup")

    def upload_state(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, dtype=np.float32)
        vel = np.ascontiguousarray(vel, dtype=np.float32) if vel is not None else None
        self._check(self._L.sph_mgpu_upload_state(self._h, _fp(pos), _fp(vel), len(pos)), "sph_mgpu_upload_state")

    def simulate(self):
        if self.mouseClicked:  # simulator.cu:482-489: applied after this step's force sweep
            self._check(self._L.sph_mgpu_queue_click(self._h, int(self.clickCoords[0]), int(self.clickCoords[1])),
                        "sph_mgpu_queue_click")
            self.mouseClicked = False
        self._check(self._L.sph_mgpu_step(self._h, None), "sph_mgpu_step")

    def step_phase(self, phase, times=None):
        self._check(self._L.sph_mgpu_step_phase(self._h, int(phase), C.byref(times) if times is not None else None),
                    "sph_mgpu_step_phase")

    def simulateAndTime(self, times):
        self._check(self._L.sph_mgpu_step(self._h, C.byref(times)), "sph_mgpu_step")

    def moveParticles(self, mouse_pos):
        """Rides on the next simulate(): the impulse needs the slabs' grids of a step."""
        self._check(self._L.sph_mgpu_queue_click(self._h, int(mouse_pos[0]), int(mouse_pos[1])),
                    "sph_mgpu_queue_click")

    def getPosition(self):
        p = self._L.sph_mgpu_positions_host(self._h)
        if not p:
            raise SphError("sph_mgpu_positions_host failed: " + self._L.sph_mgpu_last_error(self._h).decode())
        return np.ctypeslib.as_array(p, shape=(self.n, 3)) if self.n else np.zeros((0, 3), np.float32)

    def download_state(self):
        """Rows of locally owned particles (particle-id order); the others are NaN."""
        n = self.n
        pos = np.full((n, 3), np.nan, np.float32)
        vel = np.full((n, 3), np.nan, np.float32)
        rho = np.full(n, np.nan, np.float32)
        w = C.c_int(0)
        self._check(self._L.sph_mgpu_download_state(self._h, _fp(pos), _fp(vel), _fp(rho), C.byref(w)),
                    "sph_mgpu_download_state")
        return dict(pos=pos, vel=vel, rho=rho, written=w.value)

    def sync(self):
        self._check(self._L.sph_mgpu_sync(self._h), "sph_mgpu_sync")

    def stats(self, reset=False):
        s = SphMgpuStats()
        self._check(self._L.sph_mgpu_get_stats(self._h, C.byref(s), 1 if reset else 0), "sph_mgpu_get_stats")
        return s
