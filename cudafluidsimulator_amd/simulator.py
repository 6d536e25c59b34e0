"""Python mirror of the reference's `Simulator` (src/simulator.h:53-74).

Same method names and semantics -- setup(), simulate(), simulateAndTime(times),
getPosition(), moveParticles((x, y)) -- over the C-ABI of libsph_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (SphError, SphKernelTimes, SphOptions, SphSettings, SphTimes,
                   load_library)


class Times(SphTimes):
    """times.h:5-10.  display() prints the reference's table (times.h:12-35)."""

    def display(self):
        it = self.iters
        rows = [("%-12s%18s%12s" % ("Operation", "Per frame", "Total")), "-" * 45]
        g = self.buildGrid / it if it else 0.0
        s = self.sphUpdate / it if it else 0.0
        m = self.memcpy / it if it else 0.0
        rows.append("%-11s%11.5f%15.5f" % ("Grid construction", g, self.buildGrid))
        rows.append("%-12s%16.5f%15.5f" % ("SPH update", s, self.sphUpdate))
        rows.append("%-12s%15.5f%15.5f" % ("Data transfer", m, self.memcpy))
        return "\n".join(rows)


Settings = SphSettings


def default_settings(num_particles, random_init):
    """The constants main() derives (main.cpp:57-63)."""
    s = SphSettings()
    rc = load_library().sph_default_settings(C.byref(s), int(num_particles),
                                             1 if random_init else 0)
    if rc:
        raise SphError("sph_default_settings failed")
    return s


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class Simulator:
    def __init__(self, settings, sweep="list", flags=0, device=-1, capacity=0, math="strict", key_order="flattened"):
        self.settings = settings
        self._L = load_library()
        self._h = C.c_void_p()
        self._opt = SphOptions()
        self._opt.struct_size = C.sizeof(SphOptions)
        self._opt.device = device
        self._opt.math_mode = _lib.SPH_MATH_FAST if math == "fast" else _lib.SPH_MATH_STRICT
        self._opt.sweep = _lib.SWEEPS[sweep]
        self._opt.flags = flags
        self._opt.capacity = capacity
        self._opt.key_order = 1 if key_order == "morton" else 0
        rc = self._L.sph_create(C.byref(settings), C.byref(self._opt), C.byref(self._h))
        if rc:
            msg = self._L.sph_last_error(None).decode()
            self._h = C.c_void_p()
            raise SphError(f"sph_create failed ({rc}): {msg}")
        self.mouseClicked = False
        self.clickCoords = (0, 0)

    # -- plumbing --
    def _check(self, rc, what):
        if rc:
            raise SphError(f"{what} failed ({rc}): {self._L.sph_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.sph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n(self):
        return self.settings.numParticles

    # -- the reference's methods --
    def setup(self):
        self._check(self._L.sph_setup(self._h), "sph_setup")

    def simulate(self):
        self._check(self._L.sph_step(self._h, None), "sph_step")
        if self.mouseClicked:  # simulator.cu:482-489
            self.moveParticles(self.clickCoords)
            self.mouseClicked = False

    def simulateAndTime(self, times):
        self._check(self._L.sph_step(self._h, C.byref(times)), "sph_step")

    def moveParticles(self, mouse_pos):
        self._check(self._L.sph_apply_click(self._h, int(mouse_pos[0]), int(mouse_pos[1])),
                    "sph_apply_click")

    def getPosition(self):
        """(n, 3) float32 view of the handle's host buffer, original-id order,
        valid until the next step (simulator.cu:407-409)."""
        p = self._L.sph_positions_host(self._h)
        if not p:
            raise SphError("sph_positions_host failed")
        if self.n == 0:
            return np.zeros((0, 3), np.float32)
        return np.ctypeslib.as_array(p, shape=(self.n, 3))

    # -- extras (tests, bench) --
    def upload_state(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, dtype=np.float32)
        if vel is not None:
            vel = np.ascontiguousarray(vel, dtype=np.float32)
        self._check(self._L.sph_upload_state(self._h, _fp(pos), _fp(vel), len(pos)),
                    "sph_upload_state")

    def download_state(self):
        n = self.n
        pos = np.zeros((n, 3), np.float32)
        vel = np.zeros((n, 3), np.float32)
        rho = np.zeros(n, np.float32)
        prs = np.zeros(n, np.float32)
        self._check(self._L.sph_download_state(self._h, _fp(pos), _fp(vel), _fp(rho),
                                               _fp(prs)), "sph_download_state")
        return dict(pos=pos, vel=vel, rho=rho, prs=prs)

    def download_force(self):
        f = np.zeros((self.n, 3), np.float32)
        self._check(self._L.sph_download_force(self._h, _fp(f)), "sph_download_force")
        return f

    def download_grid(self):
        n = self.n
        ids = np.zeros(n, np.uint32)
        keys = np.zeros(n, np.uint32)
        cells = np.zeros((self._L.sph_num_table_cells(self._h), 2), np.int32)
        self._check(self._L.sph_download_grid(
            self._h, ids.ctypes.data_as(C.POINTER(C.c_uint32)),
            keys.ctypes.data_as(C.POINTER(C.c_uint32)),
            cells.ctypes.data_as(C.POINTER(C.c_int32))), "sph_download_grid")
        return dict(ids=ids, keys=keys, cells=cells)

    def debug_counters(self):
        out = (C.c_uint64 * 16)()
        self._check(self._L.sph_debug_counters(self._h, out), "sph_debug_counters")
        return list(out)

    def save_state(self, path):
        self._check(self._L.sph_save_state(self._h, str(path).encode()), "sph_save_state")

    def load_state(self, path):
        self._check(self._L.sph_load_state(self._h, str(path).encode()), "sph_load_state")

    def phase(self, name):
        self._check(getattr(self._L, "sph_phase_" + name)(self._h), "sph_phase_" + name)

    def sync(self):
        self._check(self._L.sph_sync(self._h), "sph_sync")

    def kernel_times(self, reset=False):
        kt = SphKernelTimes()
        self._check(self._L.sph_get_kernel_times(self._h, C.byref(kt), 1 if reset else 0),
                    "sph_get_kernel_times")
        return kt


def sort_check(keys, key_bits=20, device=-1):
    """Run the grid build's radix sort alone (tests)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    n = len(keys)
    perm = np.zeros(n, np.uint32)
    sk = np.zeros(n, np.uint32)
    u32p = C.POINTER(C.c_uint32)
    rc = load_library().sph_sort_check(device, keys.ctypes.data_as(u32p), n, key_bits,
                                       perm.ctypes.data_as(u32p), sk.ctypes.data_as(u32p))
    if rc:
        raise SphError(f"sph_sort_check failed ({rc})")
    return perm, sk
