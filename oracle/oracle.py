"""ctypes binding of the CPU oracle (oracle/sph_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (cudafluidsimulator_amd) never
imports this module.  Parity is unpinned by reference tests (the reference has
none); see sph_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsph_oracle.so")


class OracleSettings(C.Structure):
    # layout-identical to the reference's struct Settings (simulator.h:19-31)
    _fields_ = [
        ("randomInit", C.c_uint8),
        ("pad_", C.c_uint8 * 3),
        ("numParticles", C.c_int32),
        ("h", C.c_float),
        ("v_kernel_coeff", C.c_float),
        ("d_kernel_coeff", C.c_float),
        ("boxDim", C.c_float),
        ("numCellsPerDim", C.c_float),
        ("timestep", C.c_float),
    ]


def build(force=False):
    src = os.path.join(_HERE, "sph_oracle.c")
    hdr = os.path.join(_HERE, "sph_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsph_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def use_native_build():
    """bench.py's cpu_baseline leg: switch this module to the -O3 -march=native build
    of the same source (oracle/Makefile target `native`), compiled on THIS machine.
    Must be called before the library is first used.  Returns the flags string."""
    global _LIB_PATH
    if _lib is not None:
        raise RuntimeError("oracle library already loaded")
    subprocess.check_call(["make", "-C", _HERE, "native"], stdout=subprocess.DEVNULL)
    _LIB_PATH = os.path.join(_HERE, "_native", "libsph_oracle_native.so")
    return "gcc -O3 -march=native -ffp-contract=off -fno-fast-math -fopenmp"


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.endswith("_native.so"):
        build()
    # A GPU box hands a job a CPU *share* (about 16 cores of a 128-core host) that
    # neither os.cpu_count() nor the affinity mask shows.  A 128-thread OpenMP team
    # spinning at every barrier on that share can make a one-second test take minutes,
    # so the team is capped (ORACLE_MAX_THREADS, default 16) and waits passively.
    # Results do not depend on the thread count.
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    i32p = C.POINTER(C.c_int32)
    sp = C.POINTER(OracleSettings)
    L.oracle_make_settings.argtypes = [sp, C.c_int, C.c_int]
    L.oracle_init_positions.argtypes = [sp, fp]
    L.oracle_init_positions.restype = C.c_int
    L.oracle_init_positions_dense.argtypes = [sp, fp]
    L.oracle_init_positions_dense.restype = C.c_int
    L.oracle_cell_keys.argtypes = [sp, fp, C.c_int, u32p]
    L.oracle_stable_sort.argtypes = [u32p, C.c_int, C.c_int, u32p]
    L.oracle_cell_table.argtypes = [u32p, C.c_int, C.c_int, i32p, i32p]
    L.oracle_density.argtypes = [sp, fp, C.c_int, i32p, i32p, C.c_int, C.c_int, fp, fp]
    L.oracle_force.argtypes = [sp, fp, fp, fp, fp, C.c_int, i32p, i32p, C.c_int,
                               C.c_int, fp]
    L.oracle_integrate.argtypes = [sp, fp, fp, fp, fp, C.c_int, C.c_int]
    L.oracle_pair_tests.argtypes = [sp, fp, C.c_int, i32p, i32p, C.c_int, C.c_int]
    L.oracle_pair_tests.restype = C.c_uint64
    L.oracle_sim_create.argtypes = [sp]
    L.oracle_sim_create.restype = C.c_void_p
    L.oracle_sim_destroy.argtypes = [C.c_void_p]
    L.oracle_sim_setup.argtypes = [C.c_void_p]
    L.oracle_sim_upload.argtypes = [C.c_void_p, fp, fp]
    L.oracle_sim_step.argtypes = [C.c_void_p]
    L.oracle_sim_click.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.oracle_sim_download.argtypes = [C.c_void_p, fp, fp, fp, fp, fp]
    L.oracle_sim_sorted.argtypes = [C.c_void_p, u32p, u32p, fp, fp]
    L.oracle_sim_sorted.restype = C.c_int
    L.oracle_sim_last_pair_tests.argtypes = [C.c_void_p]
    L.oracle_sim_last_pair_tests.restype = C.c_uint64
    L.oracle_set_key_order.argtypes = [C.c_int]
    L.oracle_num_keys.argtypes = [C.c_int]
    L.oracle_num_keys.restype = C.c_int
    L.oracle_num_threads.restype = C.c_int
    L.oracle_set_num_threads.argtypes = [C.c_int]
    cap = int(os.environ.get("ORACLE_MAX_THREADS", "16"))
    if cap > 0 and L.oracle_num_threads() > cap:
        L.oracle_set_num_threads(cap)
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def make_settings(n, random_init):
    s = OracleSettings()
    lib().oracle_make_settings(C.byref(s), int(n), 1 if random_init else 0)
    return s


def init_positions(s):
    pos = np.zeros((s.numParticles, 3), dtype=np.float32)
    written = lib().oracle_init_positions(C.byref(s), _fp(pos))
    if written < s.numParticles:
        lib().oracle_init_positions_dense(C.byref(s), _fp(pos))
    return pos


def cell_keys(s, pos):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    keys = np.zeros(len(pos), dtype=np.uint32)
    lib().oracle_cell_keys(C.byref(s), _fp(pos), len(pos), _u32(keys))
    return keys


def stable_sort(keys, num_cells=1000000):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    perm = np.zeros(len(keys), dtype=np.uint32)
    lib().oracle_stable_sort(_u32(keys), len(keys), num_cells, _u32(perm))
    return perm


def cell_table(sorted_keys, num_cells=1000000):
    sk = np.ascontiguousarray(sorted_keys, dtype=np.uint32)
    cs = np.zeros(num_cells, dtype=np.int32)
    ce = np.zeros(num_cells, dtype=np.int32)
    lib().oracle_cell_table(_u32(sk), len(sk), num_cells, _i32(cs), _i32(ce))
    return cs, ce


def density(s, pos, cs, ce, i_begin=0, i_end=None):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    n = len(pos)
    i_end = n if i_end is None else i_end
    rho = np.zeros(n, dtype=np.float32)
    prs = np.zeros(n, dtype=np.float32)
    lib().oracle_density(C.byref(s), _fp(pos), n, _i32(cs), _i32(ce), i_begin, i_end,
                         _fp(rho), _fp(prs))
    return rho, prs


def force(s, pos, vel, rho, prs, cs, ce, i_begin=0, i_end=None):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    vel = np.ascontiguousarray(vel, dtype=np.float32)
    n = len(pos)
    i_end = n if i_end is None else i_end
    f = np.zeros((n, 3), dtype=np.float32)
    lib().oracle_force(C.byref(s), _fp(pos), _fp(vel), _fp(rho), _fp(prs), n, _i32(cs),
                       _i32(ce), i_begin, i_end, _fp(f))
    return f


def integrate(s, pos, vel, frc, rho, i_begin=0, i_end=None):
    """In place on pos/vel (must be C-contiguous float32)."""
    n = len(pos)
    i_end = n if i_end is None else i_end
    lib().oracle_integrate(C.byref(s), _fp(pos), _fp(vel), _fp(frc), _fp(rho), i_begin,
                           i_end)


def pair_tests(s, pos, cs, ce, i_begin=0, i_end=None):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    n = len(pos)
    i_end = n if i_end is None else i_end
    return int(lib().oracle_pair_tests(C.byref(s), _fp(pos), n, _i32(cs), _i32(ce),
                                       i_begin, i_end))


class OracleSim:
    """Single-domain simulation object; mirrors Simulator (simulator.h:53-74)."""

    def __init__(self, n, random_init):
        self.settings = make_settings(n, random_init)
        self.n = int(n)
        self._h = lib().oracle_sim_create(C.byref(self.settings))

    def close(self):
        if self._h:
            lib().oracle_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setup(self):
        lib().oracle_sim_setup(self._h)

    def upload(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, dtype=np.float32)
        assert pos.shape == (self.n, 3)
        if vel is not None:
            vel = np.ascontiguousarray(vel, dtype=np.float32)
        lib().oracle_sim_upload(self._h, _fp(pos), _fp(vel))

    def step(self, k=1):
        for _ in range(k):
            lib().oracle_sim_step(self._h)

    def click(self, x, y):
        lib().oracle_sim_click(self._h, int(x), int(y))

    def download(self, want_force=False):
        n = self.n
        pos = np.zeros((n, 3), np.float32)
        vel = np.zeros((n, 3), np.float32)
        rho = np.zeros(n, np.float32)
        prs = np.zeros(n, np.float32)
        frc = np.zeros((n, 3), np.float32) if want_force else None
        lib().oracle_sim_download(self._h, _fp(pos), _fp(vel), _fp(rho), _fp(prs),
                                  _fp(frc))
        out = dict(pos=pos, vel=vel, rho=rho, prs=prs)
        if want_force:
            out["force"] = frc
        return out

    def sorted_state(self):
        n = self.n
        ids = np.zeros(n, np.uint32)
        keys = np.zeros(n, np.uint32)
        pos = np.zeros((n, 3), np.float32)
        vel = np.zeros((n, 3), np.float32)
        lib().oracle_sim_sorted(self._h, _u32(ids), _u32(keys), _fp(pos), _fp(vel))
        return dict(ids=ids, keys=keys, pos=pos, vel=vel)

    def last_pair_tests(self):
        return int(lib().oracle_sim_last_pair_tests(self._h))


def set_key_order(order):
    """0 = flattened cell index (reference), 1 = Morton.  Set before creating a sim."""
    lib().oracle_set_key_order(1 if order in (1, "morton") else 0)


def num_keys(cells_per_dim=100):
    return int(lib().oracle_num_keys(int(cells_per_dim)))


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(t):
    lib().oracle_set_num_threads(int(t))
