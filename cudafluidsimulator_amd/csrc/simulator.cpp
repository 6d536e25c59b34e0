// `class Simulator` (include/simulator.h) over the C-ABI in sph_c_api.h.
// Mirrors the method semantics of the reference's Simulator
// (simulator.cu:370-546): void methods, caller-owned Settings kept by pointer,
// getPosition() stable for the simulator's lifetime and coherent as soon as
// simulate()/simulateAndTime() has returned.  The reference checks no CUDA
// status; here a failing call aborts with the library's message.
#include "simulator.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "sph_c_api.h"

// Defined by the front end (display.cpp:19-20) or by headless.cpp.
extern bool mouseClicked;
extern int2 clickCoords;

static_assert(sizeof(Settings) == sizeof(SphSettings), "Settings layout");
static_assert(offsetof(Settings, numParticles) == offsetof(SphSettings, numParticles), "Settings layout");
static_assert(offsetof(Settings, h) == offsetof(SphSettings, h), "Settings layout");
static_assert(offsetof(Settings, timestep) == offsetof(SphSettings, timestep), "Settings layout");
static_assert(sizeof(Times) == sizeof(SphTimes), "Times layout");
static_assert(offsetof(Times, iters) == offsetof(SphTimes, iters), "Times layout");
static_assert(sizeof(float3) == 12, "float3 must be 12 bytes");

static void check(sph_handle *h, int rc, const char *what) {
    if (rc == SPH_OK) return;
    fprintf(stderr, "sph: %s failed (%d): %s\n", what, rc, sph_last_error(h));
    abort();
}

Simulator::Simulator(Settings *settings) : impl(NULL), settings(settings) {}

Simulator::~Simulator() {
    if (impl) sph_destroy(impl);
}

void Simulator::setup() {
    if (impl) {
        sph_destroy(impl);
        impl = NULL;
    }
    SphSettings s;
    memcpy(&s, settings, sizeof s);
    s.randomInit = settings->randomInit ? 1 : 0;
    SphOptions o;
    memset(&o, 0, sizeof o);
    o.struct_size = (int32_t)sizeof o;
    o.device = -1;
    if (const char *e = getenv("SPH_SWEEP"))
        o.sweep = strcmp(e, "direct") == 0   ? SPH_SWEEP_DIRECT
                  : strcmp(e, "lds") == 0    ? SPH_SWEEP_LDS
                  : strcmp(e, "linked") == 0 ? SPH_SWEEP_LINKED
                                             : SPH_SWEEP_LIST;
    int rc = sph_create(&s, &o, &impl);
    check(NULL, rc, "sph_create");
    check(impl, sph_setup(impl), "sph_setup");
}

const float3 *Simulator::getPosition() {
    if (!impl) return NULL;
    return reinterpret_cast<const float3 *>(sph_positions_host(impl));
}

void Simulator::simulate() {
    check(impl, sph_step(impl, NULL), "sph_step");
    if (mouseClicked) { // simulator.cu:482-489
        check(impl, sph_apply_click(impl, clickCoords.x, clickCoords.y), "sph_apply_click");
        mouseClicked = false;
    }
}

void Simulator::simulateAndTime(Times *times) {
    check(impl, sph_step(impl, reinterpret_cast<SphTimes *>(times)), "sph_step");
}

void Simulator::moveParticles(int2 mouse_pos) {
    check(impl, sph_apply_click(impl, mouse_pos.x, mouse_pos.y), "sph_apply_click");
}
