"""Drop-in checks against the reference's own front-end sources.  These need
/root/reference (this container only; it does not exist on the GPU box) and are
compile-only: nothing from the reference is copied into the repo or executed
beyond its header-only times.h table printer."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
INC = ["-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include"]

needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted")

PROG = r"""
#include "times.h"
int main() { Times t; t.buildGrid = 0.123456; t.sphUpdate = 1.5; t.memcpy = 0.0421; t.iters = 100;
  displayTimes(&t); Times z; displayTimes(&z); return 0; }
"""


def _table(include_dir):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(PROG)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-I" + include_dir, src, "-o", exe])
        return subprocess.check_output([exe])


@needs_ref
def test_times_table_is_byte_identical_to_reference():
    assert _table(os.path.join(ROOT, "include")) == _table(REF)


def test_times_table_known_bytes():
    out = _table(os.path.join(ROOT, "include")).decode().split("\n")
    assert out[0] == "Operation            Per frame       Total"
    assert out[2] == "Grid construction    0.00123        0.12346"
    assert out[3] == "SPH update           0.01500        1.50000"
    assert out[4] == "Data transfer        0.00042        0.04210"


GLUT_STUB = r"""
// syntax-check stand-in for <GL/glut.h> (absent from the image); prototypes only
#pragma once
#define GLUT_RGB 0
#define GLUT_DOUBLE 2
#define GLUT_LEFT_BUTTON 0
#define GLUT_DOWN 0
#define GL_COLOR_BUFFER_BIT 0x4000
#define GL_DEPTH_BUFFER_BIT 0x100
#define GL_LINES 1
#define GL_POINTS 0
#define GL_POINT_SMOOTH 0xB10
#define GL_DEPTH_TEST 0xB71
#define GL_PROJECTION 0x1701
#define GL_MODELVIEW 0x1700
extern "C" {
void glutInit(int *, char **); void glutInitDisplayMode(unsigned); void glutInitWindowSize(int, int);
int glutCreateWindow(const char *); void glutDisplayFunc(void (*)()); void glutMouseFunc(void (*)(int, int, int, int));
void glutMainLoop(); void glutSwapBuffers(); void glutPostRedisplay();
void glClear(unsigned); void glLoadIdentity(); void glColor3f(float, float, float); void glBegin(unsigned); void glEnd();
void glVertex3fv(const float *); void glVertex3f(float, float, float); void glClearColor(float, float, float, float);
void glEnable(unsigned); void glPointSize(float); void glMatrixMode(unsigned);
void glFrustum(double, double, double, double, double, double); void glTranslatef(float, float, float);
}
"""


@needs_ref
@pytest.mark.parametrize("src", ["main.cpp", "display.cpp"])
def test_reference_front_end_compiles_against_our_headers(src):
    """The reference's unmodified main.cpp / display.cpp pass a syntax check with
    OUR simulator.h + times.h (the text is piped to g++ so its quoted includes
    resolve through -iquote, ours first) and the types-only cuda_runtime.h."""
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "GL"))
        open(os.path.join(d, "GL", "glut.h"), "w").write(GLUT_STUB)
        open(os.path.join(d, "GL", "glu.h"), "w").write("#pragma once\n")
        text = open(os.path.join(REF, src)).read()
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-x", "c++", "-fsyntax-only", "-",
               "-iquote", os.path.join(ROOT, "include"), "-iquote", REF,
               "-I" + os.path.join(ROOT, "include", "compat"), "-I" + d] + INC
        dep = subprocess.run(cmd + ["-M"], input=text, text=True, capture_output=True, check=True)
        assert os.path.join(ROOT, "include", "simulator.h") in dep.stdout
        assert os.path.join(REF, "simulator.h") not in dep.stdout
        subprocess.run(cmd, input=text, text=True, check=True)


@needs_ref
def test_reference_main_builds_and_links_against_library():
    """Build the reference's main.cpp (compiled from a temp copy of nothing: the
    file is #included from where it lies, with the current directory set so that
    its quoted includes find OUR headers) and link it with our simulator +
    headless stub + libsph_hip.so (+ libsph_mgpu.so, the SPH_GPUS=N path of the same
    Simulator class).  Link only -- running needs a GPU."""
    lib = os.path.join(ROOT, "cudafluidsimulator_amd", "libsph_hip.so")
    csrc = os.path.join(ROOT, "cudafluidsimulator_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "GL"))
        open(os.path.join(d, "GL", "glut.h"), "w").write(GLUT_STUB)
        open(os.path.join(d, "GL", "glu.h"), "w").write("#pragma once\n")
        # -x c++ from stdin: quoted includes then resolve via -iquote, ours first
        text = open(os.path.join(REF, "main.cpp")).read()
        obj = os.path.join(d, "refmain.o")
        cmd = ["g++", "-std=c++17", "-O2", "-x", "c++", "-c", "-", "-o", obj,
               "-iquote", os.path.join(ROOT, "include"), "-iquote", REF,
               "-I" + os.path.join(ROOT, "include", "compat"), "-I" + d,
               "-D__HIP_PLATFORM_AMD__"] + INC
        subprocess.run(cmd, input=text, text=True, check=True)
        exe = os.path.join(d, "sph_refmain")
        subprocess.check_call(["g++", "-o", exe, obj, os.path.join(csrc, "simulator.o"),
                               os.path.join(csrc, "headless.o"),
                               os.path.join(os.path.dirname(lib), "libsph_mgpu.so"), lib,
                               "-Wl,-rpath," + os.path.dirname(lib)])
        out = subprocess.run([exe, "-i", "bogus"], capture_output=True, text=True)
        assert out.returncode == 1 and "Invalid argument for option -i" in out.stdout
