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
#include "sph_mgpu.h"

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

static void mcheck(sph_mgpu *m, int rc, const char *what) {
    if (rc == SPH_OK) return;
    fprintf(stderr, "sph: %s failed (%d): %s\n", what, rc, sph_mgpu_last_error(m));
    abort();
}

Simulator::Simulator(Settings *settings) : impl(NULL), multi(NULL), settings(settings) {}

Simulator::~Simulator() {
    if (impl) sph_destroy(impl);
    if (multi) sph_mgpu_destroy(multi);
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
    // SPH_GPUS=N (N > 1): the same step over N z-slabs, one per GPU, halo layers by RCCL
    // send/recv, all driven from this thread (include/sph_mgpu.h).  SPH_TRANSPORT=loopback
    // runs the N slabs on one GPU (rehearsal on a one-GPU machine).
    int gpus = getenv("SPH_GPUS") ? atoi(getenv("SPH_GPUS")) : 1;
    if (multi) {
        sph_mgpu_destroy(multi);
        multi = NULL;
    }
    if (gpus > 1) {
        SphMgpuOptions mo;
        memset(&mo, 0, sizeof mo);
        mo.struct_size = (int32_t)sizeof mo;
        mo.world = gpus;
        mo.rank_begin = 0;
        mo.rank_count = gpus;
        const char *tr = getenv("SPH_TRANSPORT");
        mo.transport = (tr && strcmp(tr, "loopback") == 0) ? SPH_TRANSPORT_LOOPBACK
                       : (tr && strcmp(tr, "rccl_self") == 0) ? SPH_TRANSPORT_RCCL_SELF
                       : (tr && strcmp(tr, "streams") == 0)   ? SPH_TRANSPORT_STREAMS
                                                               : SPH_TRANSPORT_RCCL;
        for (int k = 0; k < SPH_MGPU_MAX_LOCAL; ++k) mo.devices[k] = mo.transport == SPH_TRANSPORT_RCCL ? k : 0;
        mo.sweep = o.sweep;
        if (const char *e = getenv("SPH_RECUT_EVERY")) mo.recut_every = atoi(e);
        int rc = sph_mgpu_create(&s, &mo, NULL, &multi);
        mcheck(NULL, rc, "sph_mgpu_create");
        mcheck(multi, sph_mgpu_setup(multi), "sph_mgpu_setup");
        return;
    }
    int rc = sph_create(&s, &o, &impl);
    check(NULL, rc, "sph_create");
    check(impl, sph_setup(impl), "sph_setup");
}

const float3 *Simulator::getPosition() {
    if (multi) return reinterpret_cast<const float3 *>(sph_mgpu_positions_host(multi));
    if (!impl) return NULL;
    return reinterpret_cast<const float3 *>(sph_positions_host(impl));
}

void Simulator::simulate() {
    if (multi) { // every slab applies the impulse to the layers it owns, after its force sweep
        if (mouseClicked) mcheck(multi, sph_mgpu_queue_click(multi, clickCoords.x, clickCoords.y), "sph_mgpu_queue_click");
        mcheck(multi, sph_mgpu_step(multi, NULL), "sph_mgpu_step");
        mouseClicked = false;
        return;
    }
    check(impl, sph_step(impl, NULL), "sph_step");
    if (mouseClicked) { // simulator.cu:482-489
        check(impl, sph_apply_click(impl, clickCoords.x, clickCoords.y), "sph_apply_click");
        mouseClicked = false;
    }
}

void Simulator::simulateAndTime(Times *times) {
    if (multi) {
        mcheck(multi, sph_mgpu_step(multi, reinterpret_cast<SphTimes *>(times)), "sph_mgpu_step");
        return;
    }
    check(impl, sph_step(impl, reinterpret_cast<SphTimes *>(times)), "sph_step");
}

void Simulator::moveParticles(int2 mouse_pos) {
    if (multi) { // multi-GPU: the impulse needs the slabs' grids of a step: it rides on the next simulate()
        mcheck(multi, sph_mgpu_queue_click(multi, mouse_pos.x, mouse_pos.y), "sph_mgpu_queue_click");
        return;
    }
    if (!impl) return;
    check(impl, sph_apply_click(impl, mouse_pos.x, mouse_pos.y), "sph_apply_click");
}
