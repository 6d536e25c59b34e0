"""The C-ABI library loads and exports every symbol include/sph_c_api.h declares;
struct layouts match the reference's Settings / Times (no compute without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sph_c_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = sph.load_library()
    for name in declared_symbols():
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", sph.library_path()], text=True)
    exported = set(re.findall(r" T (sph_\w+)", out))
    assert set(declared_symbols()) <= exported


def test_struct_layouts_match_reference():
    # Settings: bool,int,6 floats = 32 B (simulator.h:19-31); Times: 3 doubles + int = 32 B
    assert C.sizeof(sph.SphSettings) == 32
    assert sph.SphSettings.numParticles.offset == 4 and sph.SphSettings.h.offset == 8
    assert sph.SphSettings.timestep.offset == 28
    assert C.sizeof(sph.SphTimes) == 32 and sph.SphTimes.iters.offset == 24
    assert C.sizeof(sph.SphOptions) == 28


def test_default_settings_match_oracle():
    from oracle import oracle as O
    a = sph.default_settings(12345, True)
    b = O.make_settings(12345, True)
    assert bytes(a) == bytes(b)


def test_no_cpu_fallback(have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    with pytest.raises(sph.SphError, match="no HIP device"):
        sph.Simulator(sph.default_settings(16, False))


def test_product_does_not_link_or_import_oracle():
    out = subprocess.check_output(["ldd", sph.library_path()], text=True)
    assert "oracle" not in out
    pkg = os.path.join(ROOT, "cudafluidsimulator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libsph_oracle" not in text and '#include "sph_oracle' not in text, f


def test_times_table_format():
    t = sph.Times()
    t.buildGrid, t.sphUpdate, t.memcpy, t.iters = 0.12346, 1.5, 0.0421, 100
    lines = t.display().split("\n")
    assert lines[0] == "Operation            Per frame       Total"
    assert lines[1] == "-" * 45
    assert lines[2] == "Grid construction    0.00123        0.12346"
    assert lines[3] == "SPH update           0.01500        1.50000"
    assert lines[4] == "Data transfer        0.00042        0.04210"
