// C-ABI implementation (include/sph_c_api.h) of the MI355X SPH step path.
// Host logic only; the kernels live in sort.hip / grid.hip / sweeps.hip.
//
// Step pipeline (replaces Simulator::simulate / simulateAndTime,
// simulator.cu:462-546):
//   compute stream: clear cell table -> hash -> 3-pass radix sort -> gather +
//                   cell ranges -> density -> force+integrate (+ scatter of
//                   positions into original-id order)
//   copy stream:    D2H of the id-ordered positions into pinned host memory,
//                   double-buffered on the device so step k+1 computes while
//                   step k's positions cross PCIe (the reference blocks on this
//                   copy every step, simulator.cu:479-480,532-533).
// Compile with -ffp-contract=off (host initialisers must round like the
// reference's g++ -O3 x86-64 build, Makefile:22-23).
#include "sph_c_api.h"
#include "sph_device.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

constexpr int kEventRing = 64;
constexpr int kCounterWords = 16 + 256 * 16; // 16 spare words, then 256 shards x 16: [0] pair tests,
                                             // [1..13] in-kernel stamps (diagnostic builds), [15] hits
constexpr size_t kCursorBytes = (size_t)SL_POOL_SHARDS * SL_CURSOR_STRIDE * sizeof(unsigned long long);

struct PairEvent { // one timed section of the slab path
    hipEvent_t a = nullptr, b = nullptr;
    double *target = nullptr;
    bool used = false;
};
constexpr int kPairRing = 48;

struct StepEvents {
    hipEvent_t e[6] = {}; // start, hash, sort, gather, density, force
    hipEvent_t c[2] = {}; // copy start / end
    bool used = false, hasCopy = false, counted = false;
};

} // namespace

struct sph_handle {
    SphSettings settings{};
    SphOptions opt{};
    DevParams P{};
    int n = 0, cap = 0, device = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    float4 *pos4[2] = {nullptr, nullptr};
    float4 *vel4[2] = {nullptr, nullptr};
    int cur = 0;     // buffers holding the current state
    int sorted = -1; // buffers holding the sorted streams of the last grid build
    SortWorkspace ws{};
    int sortedKeyBuf = 0;
    int2 *cellRange = nullptr;       // the cell table of the LAST grid build (= cellTable[cellCur])
    int2 *cellTable[2] = {nullptr, nullptr}; // two tables: a grid built ahead (below) must not clobber the last step's
    int cellCur = 0;
    // Step pipelining (timed steps): simulateAndTime() has to wait for its step, and between that wait and the
    // first launch of the next call the GPU idled ~0.12 ms per step (host: two event queries, the return to the
    // caller -- Python in bench.py --, the next call's first launches).  A timed step therefore queues the NEXT
    // step's grid build (which only needs the state this step leaves behind) BEFORE it waits, and waits for its
    // own force event instead of the whole stream; the next step finds its grid built.  Anything that changes
    // or replaces the state in between (click, upload, load) simply drops the grid built ahead.
    bool gridAhead = false;
    int2 *clickTable = nullptr;      // cell table of the last COMPLETED step (what sph_apply_click walks)
    bool clickValid = false;
    bool aheadEnabled = true;        // default: below 1.5 M particles; SPH_PIPELINE=0/1 forces it (same results)
    StepEvents *aheadEv = nullptr;
    float *devPos[2] = {nullptr, nullptr};
    float *hostPos = nullptr; // pinned, n*3
    bool hostPosIsInit = false; // setup() restored the initial state on the device: getPosition() fetches it on demand
    // Pinned staging for state uploads (two halves, ping-pong).  A hipMemcpy from pageable
    // memory makes the runtime pin and later unpin the caller's pages; the unpin is deferred
    // and stalls the GPU's queues for 6-28 ms some time AFTER the call returned -- inside the
    // first steps of the run that follows (measured).
    float4 *stage[2] = {nullptr, nullptr};
    hipEvent_t stageFree[2] = {nullptr, nullptr};
    bool mappedPos = false;   // SPH_FLAG_MAPPED_POSITIONS: devPos[] alias hostPos (host-mapped)
    // SPH_GRAPH=1: the three phases of a step replayed as hipGraphs (captured once per
    // read-back slot).  Measured (round 2): SLOWER than plain launches -- n = 262,144:
    // 0.238 vs 0.214 ms per step, n = 4,194,304: 2.37 vs 2.34 -- a graph replay costs
    // 10-16 us where the ~15 kernels of a step (>= 5 us each) are not host-bound; off by
    // default.  Any capture failure falls back to plain launches.
    hipGraphExec_t stepGraph[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}; // grid, density, force
    StepEvents graphEv[2];
    bool graphEvPending[2] = {false, false};
    bool useGraph = false, capturing = false;
    int graphKeyBuf = 0;
    int graphCellCur[2] = {0, 0};   // the cell table each slot's graphs were captured with (what a click after a replay walks)
    hipEvent_t computeDone[2] = {nullptr, nullptr}, copyDone[2] = {nullptr, nullptr};
    bool copyPending[2] = {false, false};
    // Read-back of a TIMED step through an SDMA engine (hsa_amd_memory_async_copy) instead of the HIP runtime's
    // copy, which on this platform is a blit KERNEL: beside it the first histogram pass takes 84 instead of
    // 27 us, the density sweep +30 us, the force sweep +50 us.  The engine moves the
    // same 56 GB/s and occupies no CU.  HIP offers no way to order an HSA copy behind a kernel, so the copy
    // is issued by the host right after it has seen the step's force sweep finish -- which a timed step
    // waits for anyway; untimed steps (simulate()) keep the stream-ordered HIP copy.  SPH_READBACK_SDMA=0: off.
    bool sdmaOk = false, stepTimed = false;
    hsa_agent_t hsaGpu{}, hsaCpu{};
    hsa_signal_t rbSig[2]{};
    bool rbPending[2] = {false, false};
    int rbDeferredSlot = -1;         // the read-back phase left this slot's copy to sph_step
    uint32_t sdmaEngine = 0;         // hsa_amd_sdma_engine_id_t picked by sdma_init (0: the HSA runtime's own choice)
    double hsaTickSeconds = 0;
    bool cursorClean = false; // the gather launch of this grid build cleared the hit-stream cursors
    long long stepIndex = 0;
    float4 *force4 = nullptr;
    unsigned long long *pairCounter = nullptr; // device
    unsigned long long *pairHost = nullptr;    // pinned
    StepEvents ring[kEventRing];
    int ringHead = 0;
    StepEvents *curEv = nullptr;
    SphKernelTimes kt{};
    float4 *pv8 = nullptr;
    uint32_t *maskPool = nullptr, *maskOff = nullptr; // SPH_SWEEP_LIST
    uint32_t *noneList = nullptr;    // SPH_SWEEP_LIST: waves without a stream this step (pool exhausted)
    uint32_t *hitCount = nullptr;    // SPH_SWEEP_LIST: recorded hits per sorted row
    int slabOwnedBegin = 0, slabOwnedEnd = 0; // rows of the last sph_slab_density (the rest of [0, n_all) is halo)
    uint32_t *quiet = nullptr;       // SPH_SWEEP_LIST: one bit per sorted row, the force sweep's zero-pair filter
    float4 *quietVref = nullptr;     // ... its reference velocity (device; picked by the first sort pass) ...
    unsigned long long *calm = nullptr; // ... and one bit per sorted row "moves with it" (written by the gather launch)
    float4 *initPos4 = nullptr;      // setup()'s initial positions (+ids), kept on the device for the next setup()
    SphOobLog *oobHost = nullptr;    // host-mapped: positions outside the grid met by the cell hash
    uint32_t oobSeen = 0;            // how many of them were already reported
    // SPH_STEP_TRACE=1 (diagnostic): where the HOST spends a timed step, printed by sph_destroy
    bool trace = false;
    double trEnqueue = 0, trSync = 0, trPost = 0, trBetween = 0, trPh[5] = {0, 0, 0, 0, 0};
    long long trSteps = 0;
    std::chrono::steady_clock::time_point trLastReturn{};
    hipEvent_t trBase = nullptr;     // first traced step's start: GPU-side timeline of every later step
    int initZLayers = 0;
    bool useQuiet = true;            // SPH_ZERO_PAIR_FILTER=0 switches the filter off (A/B; same results)
    uint64_t hitsRecorded = 0;       // SPH_FLAG_COUNT_PAIRS: hits in the stream, before the filter
    unsigned long long *maskCursor = nullptr;
    unsigned long long maskCapacity = 0; // quads (16 B)
    bool external = false;  // pos4/vel4 are caller-owned (sph_bind_buffers)
    hipStream_t ownCompute = nullptr;
    int *boundsDev = nullptr, *boundsHost = nullptr;
    int *partTiles = nullptr; // slab partition: class counts per 1024-particle tile
    PairEvent pairs[kPairRing];
    int pairHead = 0;
    int zLayers = 0;        // occupied z-layers of the (owned) particles: sizes xcd_tile()'s chunks
    int tileChunkEnv = -1;  // SPH_TILE_CHUNK: -1 auto, 0 contiguous eighths, >0 tiles per chunk
    int tileRotate = -1;    // SPH_XCD_ROTATE: xcd_tile()'s rotation period in groups (z-layers), 0 = off;
                            // -1 (default): off for the single domain, every layer for a slab (see slab_rotate)
    bool ready = false;     // state uploaded
    bool gridValid = false; // sorted streams + cell table match `sorted`
    int phase = 0;          // 0 idle, 1 grid done, 2 density done, 3 force done
    std::string err;
};

namespace {

#define HIPCHK(h, call)                                                               \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);            \
            return SPH_EHIP;                                                          \
        }                                                                             \
    } while (0)

// Every entry point that queues work makes the handle's device current first: a host
// thread that drives several GPUs (include/sph_mgpu.h) or switched devices since
// sph_create would otherwise launch on the wrong one.
#define SPH_ON_DEVICE(h)                                                              \
    do {                                                                              \
        int d__ = -1;                                                                 \
        if (hipGetDevice(&d__) != hipSuccess || d__ != (h)->device)                    \
            HIPCHK(h, hipSetDevice((h)->device));                                     \
    } while (0)

// The reference's out-of-grid diagnostic (getGridCell, simulator.cu:60-73), printed by the host after a
// synchronisation instead of by device printf: the first sort pass logs such positions and clamps their
// cell into the table (the reference indexes out of bounds there).
void report_oob(sph_handle *h) {
    if (!h || !h->oobHost) return;
    const uint32_t cnt = h->oobHost->count;
    if (cnt == h->oobSeen) return;
    const int D = h->P.D;
    const uint32_t shown = cnt < SPH_OOB_RECORDS ? cnt : SPH_OOB_RECORDS;
    for (uint32_t k = h->oobSeen < shown ? h->oobSeen : shown; k < shown; ++k) {
        const auto &r = h->oobHost->rec[k];
        const char axis[3] = {'x', 'y', 'z'};
        for (int a = 0; a < 3; ++a)
            if (r.cell[a] < 0 || r.cell[a] >= D)
                printf("OOB particle: %c = %d\n(%f, %f, %f)\n", axis[a], r.cell[a], r.pos[0], r.pos[1], r.pos[2]);
    }
    if (cnt > shown) printf("OOB particle: %u positions outside the grid so far (the first %u listed)\n", cnt, shown);
    fflush(stdout);
    h->oobSeen = cnt;
}

int fail(sph_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

// Largest dist2 for which pressureKernel (dist2 <= h*h) or viscosityKernel
// (sqrtf(dist2) <= h) can be non-zero (simulator.cu:105,125).
float force_cut2(float h) {
    float h2 = h * h;
    float x = h2;
    for (int k = 0; k < 64; ++k) {
        float nx = std::nextafterf(x, INFINITY);
        if (sqrtf(nx) <= h) x = nx;
        else break;
    }
    return x;
}

void fill_params(sph_handle *h) {
    const SphSettings &s = h->settings;
    DevParams &P = h->P;
    P.h = s.h;
    P.h2 = s.h * s.h;
    P.vcoef = s.v_kernel_coeff;
    P.dcoef = s.d_kernel_coeff;
    P.boxDim = s.boxDim;
    P.boxHi = s.boxDim - s.h;
    P.dt = s.timestep;
    P.cut2 = force_cut2(s.h);
    P.D = (int)s.numCellsPerDim;
    P.morton = h->opt.key_order == SPH_KEY_MORTON ? 1 : 0;
    {
        // The pair body's short divide / square-root chains are proven for the reference's constants
        // only (sweep_common.h): any other h or kernel coefficient selects the full IEEE expansions.
        // SPH_SLIM_DIV=0 forces the full expansions (A/B; same bits).
        SphSettings ref{};
        sph_default_settings(&ref, s.numParticles, s.randomInit);
        P.slimDiv = (s.h == ref.h && s.v_kernel_coeff == ref.v_kernel_coeff && s.d_kernel_coeff == ref.d_kernel_coeff) ? 1 : 0;
        if (const char *e = getenv("SPH_SLIM_DIV")) if (atoi(e) == 0) P.slimDiv = 0;
    }
    if (P.morton) {
        int b = 0;
        while ((1 << b) < P.D) ++b;
        P.numCells = 1 << (3 * b);
    } else {
        P.numCells = P.D * P.D * P.D;
    }
}

int key_bits(const sph_handle *h) {
    int bits = 1;
    while ((1ll << bits) < (long long)h->P.numCells) ++bits;
    return bits;
}

// Simulator::setup's initialisers (simulator.cu:430-453).
int init_positions_reference(const SphSettings &s, float *pos) {
    int n = s.numParticles;
    if (s.randomInit) {
        srand(1); // the state of a process that never called srand (the reference)
        for (int i = 0; i < n; i++) {
            float x = rand() / (float)RAND_MAX * (s.boxDim - 2.f) + 1.f;
            float y = rand() / (float)RAND_MAX * (s.boxDim - 2.f) + 1.f;
            float z = rand() / (float)RAND_MAX * (s.boxDim - 2.f) + 1.f;
            pos[3 * i + 0] = x;
            pos[3 * i + 1] = y;
            pos[3 * i + 2] = z;
        }
        return n;
    }
    float spacing = 0.9f * s.h;
    int nx = (int)(floor((s.boxDim - 2 * s.h) / spacing) + 1);
    int ny = nx, nz = nx;
    int count = 0;
    for (int x = 0; x < nx && count < n; x++)
        for (int y = 0; y < ny && count < n; y++)
            for (int z = 0; z < nz && count < n; z++) {
                pos[3 * count + 0] = s.h + spacing * x;
                pos[3 * count + 1] = s.h + spacing * y;
                pos[3 * count + 2] = s.h + spacing * z;
                count++;
            }
    return count;
}

// Extension for n beyond the reference lattice's 109^3 capacity.
void init_positions_dense(const SphSettings &s, float *pos) {
    int n = s.numParticles;
    int nx = (int)ceil(cbrt((double)n));
    while ((long long)nx * nx * nx < n) nx++;
    while (nx > 1 && (long long)(nx - 1) * (nx - 1) * (nx - 1) >= n) nx--;
    float spacing = nx > 1 ? (s.boxDim - 2 * s.h) / (float)(nx - 1) : 0.f;
    int count = 0;
    for (int x = 0; x < nx && count < n; x++)
        for (int y = 0; y < nx && count < n; y++)
            for (int z = 0; z < nx && count < n; z++) {
                pos[3 * count + 0] = s.h + spacing * x;
                pos[3 * count + 1] = s.h + spacing * y;
                pos[3 * count + 2] = s.h + spacing * z;
                count++;
            }
}

void sdma_init(sph_handle *h); // (below: the read-back through an SDMA engine)

int alloc_device(sph_handle *h) {
    const size_t cap = (size_t)(h->cap > 0 ? h->cap : 1);
    h->external = (h->opt.flags & SPH_FLAG_EXTERNAL_STATE) != 0;
    const size_t posCap = h->external ? 1 : cap; // id-ordered read-back is single-domain only
    for (int b = 0; b < 2; ++b) {
        if (!h->external) {
            HIPCHK(h, hipMalloc(&h->pos4[b], cap * sizeof(float4)));
            HIPCHK(h, hipMalloc(&h->vel4[b], cap * sizeof(float4)));
            // never hand uninitialised indices to a gather, whatever happens upstream
            HIPCHK(h, hipMemset(h->pos4[b], 0, cap * sizeof(float4)));
            HIPCHK(h, hipMemset(h->vel4[b], 0, cap * sizeof(float4)));
        }
        HIPCHK(h, hipMalloc(&h->ws.keys[b], cap * sizeof(uint32_t)));
        HIPCHK(h, hipMalloc(&h->ws.vals[b], cap * sizeof(uint32_t)));
        if (!(h->opt.flags & SPH_FLAG_MAPPED_POSITIONS))
            HIPCHK(h, hipMalloc(&h->devPos[b], posCap * 3 * sizeof(float)));
        HIPCHK(h, hipMemset(h->ws.keys[b], 0, cap * sizeof(uint32_t)));
        HIPCHK(h, hipMemset(h->ws.vals[b], 0, cap * sizeof(uint32_t)));
        HIPCHK(h, hipEventCreateWithFlags(&h->computeDone[b], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->copyDone[b], hipEventDisableTiming));
    }
    h->ws.capacity = (int)cap;
    h->ws.maxBlocks = (int)sph_sort_workspace_blocks((int)cap);
    HIPCHK(h, hipMalloc(&h->ws.blockHist,
                        (size_t)1024 * (size_t)(h->ws.maxBlocks > 0 ? h->ws.maxBlocks : 1) *
                            sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(&h->ws.digitTotal, 1024 * sizeof(uint32_t)));
    for (int b = 0; b < 2; ++b) {
        HIPCHK(h, hipMalloc(&h->cellTable[b], (size_t)h->P.numCells * sizeof(int2)));
        HIPCHK(h, hipMemset(h->cellTable[b], 0, (size_t)h->P.numCells * sizeof(int2)));
    }
    h->cellRange = h->cellTable[0];
    // measured (profiles/r03_experiments.md): n = 262,144: 0.170 -> 0.149 ms per step; n = 4,194,304: nothing
    // (the early steps are bound by the read-back, and beside more GPU work its blit kernel slows down)
    h->aheadEnabled = h->n < (3 << 19);
    if (const char *e = getenv("SPH_PIPELINE")) h->aheadEnabled = atoi(e) != 0;
    if (h->opt.flags & SPH_FLAG_MAPPED_POSITIONS) {
        // zero-copy: the force sweep's id-ordered scatter goes over PCIe into this buffer
        HIPCHK(h, hipHostMalloc(&h->hostPos, posCap * 3 * sizeof(float), hipHostMallocMapped));
        void *dp = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dp, h->hostPos, 0));
        h->devPos[0] = h->devPos[1] = static_cast<float *>(dp);
        h->mappedPos = true;
    } else {
        // (measured, round 3: a non-coherent buffer changes nothing for the runtime's copy)
        HIPCHK(h, hipHostMalloc(&h->hostPos, posCap * 3 * sizeof(float), hipHostMallocDefault));
    }
    memset(h->hostPos, 0, posCap * 3 * sizeof(float));
    if (h->opt.sweep == SPH_SWEEP_LIST) {
        // Hit-stream pool (sweeps_list.hip): a wave reserves Q quads (16 B = two (first
        // candidate, mask) pairs) for each of its 64 lanes, Q = half the largest number
        // of 32-candidate words any of its lanes can fill.  Candidates per particle =
        // 27 x particles per cell (about half the cells of the box hold particles with
        // the reference initialisers) and grow ~4.7x as the fluid settles (measured at
        // n = 4,194,304: 217 at step 1, 1011 at step 100, i.e. 9 -> 36 words per lane
        // plus up to two partly filled words per run).  The pool is sized for 5x the
        // initial fill + the per-run slack, never more than 40 % of the device memory
        // that is free now; a wave that finds it exhausted falls back to testing every
        // candidate again in the force sweep (k_force_fallback: same results, slower).
        const double ppc = (double)cap / (0.5 * (double)h->P.numCells);
        const double wordsPerLane = 5.0 * (27.0 * ppc / 32.0) + 24.0;
        unsigned long long quads = (unsigned long long)((double)cap * wordsPerLane * 0.5 * 1.25);
        if (quads < (1ull << 20)) quads = 1ull << 20;
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const unsigned long long lim = (unsigned long long)(0.4 * (double)freeB) / sizeof(uint4);
            if (quads > lim) quads = lim;
        }
        if (const char *e = getenv("SPH_MASK_POOL_WORDS")) quads = strtoull(e, nullptr, 10) / 4;
        if (quads > 0xFFFFFFF0ull) quads = 0xFFFFFFF0ull; // wave bases are 32-bit quad indices
        h->maskCapacity = quads;
        HIPCHK(h, hipMalloc(&h->maskPool, (size_t)(quads ? quads : 1) * sizeof(uint4)));
        HIPCHK(h, hipMalloc(&h->pv8, cap * 2 * sizeof(float4)));
        // per 64-particle wave: {base of its quads or ~0u, quads per lane}
        const size_t hdrWords = 2 * ((cap + 63) / 64 + 1);
        HIPCHK(h, hipMalloc(&h->maskOff, hdrWords * sizeof(uint32_t)));
        HIPCHK(h, hipMemset(h->maskOff, 0xFF, hdrWords * sizeof(uint32_t)));
        HIPCHK(h, hipMalloc(&h->noneList, hdrWords * sizeof(uint32_t))); // (>= one entry per wave)
        HIPCHK(h, hipMalloc(&h->maskCursor, kCursorBytes));
        HIPCHK(h, hipMemset(h->maskCursor, 0, kCursorBytes));
        HIPCHK(h, hipMalloc(&h->hitCount, (cap + 64) * sizeof(uint32_t)));
        HIPCHK(h, hipMemset(h->hitCount, 0, (cap + 64) * sizeof(uint32_t)));
        // one bit per sorted row + the word a 32-row window may reach into
        const size_t quietWords = 2 * ((cap + 63) / 64) + 2;
        HIPCHK(h, hipMalloc(&h->quiet, quietWords * sizeof(uint32_t)));
        HIPCHK(h, hipMemset(h->quiet, 0, quietWords * sizeof(uint32_t)));
        HIPCHK(h, hipMalloc(&h->quietVref, 2 * sizeof(float4))); // [0] the reference velocity, [1].x the all-quiet word, [1].y the halo rows'
        HIPCHK(h, hipMemset(h->quietVref, 0, 2 * sizeof(float4)));
        HIPCHK(h, hipMalloc(&h->calm, ((cap + 63) / 64 + 1) * sizeof(unsigned long long)));
        HIPCHK(h, hipMemset(h->calm, 0, ((cap + 63) / 64 + 1) * sizeof(unsigned long long)));
        if (const char *e = getenv("SPH_ZERO_PAIR_FILTER")) h->useQuiet = atoi(e) != 0;
    }
    HIPCHK(h, hipHostMalloc(&h->oobHost, sizeof(SphOobLog), hipHostMallocMapped));
    memset(h->oobHost, 0, sizeof(SphOobLog));
    {
        void *dp = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dp, h->oobHost, 0));
        h->ws.oob = static_cast<SphOobLog *>(dp);
    }
    HIPCHK(h, hipMalloc(&h->boundsDev, 16 * sizeof(int)));
    HIPCHK(h, hipMalloc(&h->partTiles, sph_partition_tiles((int)cap) * 9 * sizeof(int)));
    HIPCHK(h, hipHostMalloc(&h->boundsHost, 8 * sizeof(int), hipHostMallocDefault));
    for (auto &pe : h->pairs) {
        HIPCHK(h, hipEventCreate(&pe.a));
        HIPCHK(h, hipEventCreate(&pe.b));
    }
    if (h->opt.flags & SPH_FLAG_STORE_FORCE)
        HIPCHK(h, hipMalloc(&h->force4, cap * sizeof(float4)));
    HIPCHK(h, hipMalloc(&h->pairCounter, kCounterWords * sizeof(unsigned long long)));
    HIPCHK(h, hipMemset(h->pairCounter, 0, kCounterWords * sizeof(unsigned long long)));
    HIPCHK(h, hipHostMalloc(&h->pairHost, kCounterWords * sizeof(unsigned long long), hipHostMallocDefault));
    *h->pairHost = 0;
    for (auto &se : h->ring) {
        for (auto &e : se.e) HIPCHK(h, hipEventCreate(&e));
        for (auto &e : se.c) HIPCHK(h, hipEventCreate(&e));
    }
    for (auto &se : h->graphEv) {
        for (auto &e : se.e) HIPCHK(h, hipEventCreate(&e));
        for (auto &e : se.c) HIPCHK(h, hipEventCreate(&e));
    }
    if (const char *e = getenv("SPH_GRAPH")) h->useGraph = atoi(e) != 0;
    if (h->external) h->useGraph = false; // slab mode: the driver sizes every launch itself
    // Read-back pre-warm.  The runtime sets up its device-to-host copy path on the first copies
    // of a process: a one-off ~7 ms stall, which otherwise lands in the first steps of a run
    // (scripts/studies/early_stall.py).  A few small copies through the same stream and buffers here.
    if (h->hostPos && h->devPos[0] && !h->mappedPos) {
        int warm = 16;
        if (const char *e = getenv("SPH_PREWARM_COPIES")) warm = atoi(e);
        const size_t bytes = std::min<size_t>(posCap * 3 * sizeof(float), (size_t)1 << 20);
        for (int k = 0; k < warm; ++k) {
            HIPCHK(h, hipMemcpyAsync(h->hostPos, h->devPos[k & 1], bytes, hipMemcpyDeviceToHost, h->copy));
            HIPCHK(h, hipStreamSynchronize(h->copy));
        }
        memset(h->hostPos, 0, bytes);
    }
    HIPCHK(h, hipDeviceSynchronize()); // memsets above ran on the null stream
    sdma_init(h);
    return SPH_OK;
}

// Fold one finished step's events into the accumulated kernel times.
int resolve_events(sph_handle *h, StepEvents &se) {
    if (!se.used || se.counted) return SPH_OK;
    HIPCHK(h, hipEventSynchronize(se.e[5]));
    float ms[5];
    for (int k = 0; k < 5; ++k) HIPCHK(h, hipEventElapsedTime(&ms[k], se.e[k], se.e[k + 1]));
    h->kt.hash += ms[0] * 1e-3;
    h->kt.sort += ms[1] * 1e-3;
    h->kt.gather += ms[2] * 1e-3;
    h->kt.density += ms[3] * 1e-3;
    h->kt.force += ms[4] * 1e-3;
    if (se.hasCopy) {
        float cms;
        HIPCHK(h, hipEventSynchronize(se.c[1]));
        HIPCHK(h, hipEventElapsedTime(&cms, se.c[0], se.c[1]));
        h->kt.readback += cms * 1e-3;
    }
    h->kt.steps += 1;
    se.counted = true;
    se.used = false;
    if (h->trace && h->trBase && se.hasCopy) { // GPU-side timeline (ms since the first traced step began)
        float t0 = 0, t3 = 0, t5 = 0, c0 = 0, c1 = 0;
        if (hipEventElapsedTime(&t0, h->trBase, se.e[0]) == hipSuccess && hipEventElapsedTime(&t3, h->trBase, se.e[3]) == hipSuccess &&
            hipEventElapsedTime(&t5, h->trBase, se.e[5]) == hipSuccess && hipEventElapsedTime(&c0, h->trBase, se.c[0]) == hipSuccess &&
            hipEventElapsedTime(&c1, h->trBase, se.c[1]) == hipSuccess)
            fprintf(stderr, "sph timeline: grid %.3f..%.3f sweeps ..%.3f | copy %.3f..%.3f\n", t0, t3, t5, c0, c1);
        (void)hipGetLastError();
    }
    return SPH_OK;
}

// the captured launches carry tile_chunk(zLayers): a new state means new graphs
void drop_step_graphs(sph_handle *h) {
    for (auto &gs : h->stepGraph)
        for (auto &g : gs) {
            if (g) (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
    h->graphEvPending[0] = h->graphEvPending[1] = false;
}

// rows -> device through the handle's pinned staging halves; `fill(k, dst, count)` packs rows
// [k, k+count) into dst.  Returns when every row is on the device.
constexpr size_t kStageRows = (size_t)1 << 19; // 8 MB per half
template <class Fill>
int staged_upload(sph_handle *h, float4 *dev, size_t n, Fill fill) {
    for (int b = 0; b < 2; ++b) {
        if (!h->stage[b]) HIPCHK(h, hipHostMalloc(&h->stage[b], kStageRows * sizeof(float4), hipHostMallocDefault));
        if (!h->stageFree[b]) HIPCHK(h, hipEventCreateWithFlags(&h->stageFree[b], hipEventDisableTiming));
    }
    int b = 0;
    for (size_t k = 0; k < n; k += kStageRows, b ^= 1) {
        const size_t cnt = n - k < kStageRows ? n - k : kStageRows;
        HIPCHK(h, hipEventSynchronize(h->stageFree[b])); // (a never-recorded event is complete)
        fill(k, h->stage[b], cnt);
        HIPCHK(h, hipMemcpyAsync(dev + k, h->stage[b], cnt * sizeof(float4), hipMemcpyHostToDevice, h->compute));
        HIPCHK(h, hipEventRecord(h->stageFree[b], h->compute));
    }
    HIPCHK(h, hipStreamSynchronize(h->compute));
    return SPH_OK;
}

// host-side bookkeeping after the particle streams in buffer 0 were replaced
// ---- read-back through an SDMA engine (see sph_handle::sdmaOk) ----
struct AgentSearch { int wantBdf; hsa_agent_t gpu, cpu; bool haveGpu, haveCpu; };
hsa_status_t find_agents(hsa_agent_t a, void *data) {
    AgentSearch *S = static_cast<AgentSearch *>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !S->haveCpu) { S->cpu = a; S->haveCpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU && !S->haveGpu) {
        uint32_t bdf = 0;
        if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
            (S->wantBdf < 0 || (int)(bdf & 0xffff) == S->wantBdf)) { S->gpu = a; S->haveGpu = true; }
    }
    return HSA_STATUS_SUCCESS;
}

void sdma_init(sph_handle *h) {
    if (const char *e = getenv("SPH_READBACK_SDMA")) if (atoi(e) == 0) return;
    if (h->external || h->mappedPos || !h->hostPos || !h->devPos[0]) return;
    if (hsa_init() != HSA_STATUS_SUCCESS) return;
    AgentSearch S{};
    S.wantBdf = -1;
    int bus = 0, dev = 0;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, h->device) == hipSuccess &&
        hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, h->device) == hipSuccess)
        S.wantBdf = ((bus & 0xff) << 8) | ((dev & 0x1f) << 3);
    (void)hsa_iterate_agents(find_agents, &S);
    if (!S.haveGpu && S.wantBdf >= 0) { // no BDF match (virtualised ids): with ONE visible GPU there is no choice to make
        int count = 0;
        if (hipGetDeviceCount(&count) == hipSuccess && count == 1) {
            S.wantBdf = -1;
            (void)hsa_iterate_agents(find_agents, &S);
        }
    }
    uint64_t hz = 0;
    if (!S.haveGpu || !S.haveCpu || hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) != HSA_STATUS_SUCCESS || !hz ||
        hsa_signal_create(0, 0, nullptr, &h->rbSig[0]) != HSA_STATUS_SUCCESS ||
        hsa_signal_create(0, 0, nullptr, &h->rbSig[1]) != HSA_STATUS_SUCCESS) {
        (void)hsa_shut_down();
        return;
    }
    (void)hsa_amd_profiling_async_copy_enable(true);
    h->hsaGpu = S.gpu;
    h->hsaCpu = S.cpu;
    h->hsaTickSeconds = 1.0 / (double)hz;
    // The engines are not alike: on an MI355X four of them move 56 GB/s to the host and the rest 12.8
    // (scripts/microbench/sdma_d2h.cpp), and left to itself the HSA runtime sometimes hands out a slow one.
    // Time a few megabytes through every free engine once and keep the fastest.
    const size_t probe = std::min<size_t>((size_t)h->n * 3 * sizeof(float), (size_t)4 << 20);
    auto time_engine = [&](uint32_t engine) -> double { // seconds, or < 0
        double best = -1;
        for (int rep = 0; rep < 2; ++rep) {
            hsa_signal_store_relaxed(h->rbSig[0], 1);
            const hsa_status_t st = engine
                ? hsa_amd_memory_async_copy_on_engine(h->hostPos, S.cpu, h->devPos[0], S.gpu, probe, 0, nullptr, h->rbSig[0],
                                                      (hsa_amd_sdma_engine_id_t)engine, false)
                : hsa_amd_memory_async_copy(h->hostPos, S.cpu, h->devPos[0], S.gpu, probe, 0, nullptr, h->rbSig[0]);
            if (st != HSA_STATUS_SUCCESS) return -1;
            int tries = 0;
            while (hsa_signal_wait_scacquire(h->rbSig[0], HSA_SIGNAL_CONDITION_LT, 1, hz / 2, HSA_WAIT_STATE_BLOCKED) >= 1)
                if (++tries > 20) return -1;
            hsa_amd_profiling_async_copy_time_t t{};
            if (hsa_amd_profiling_get_async_copy_time(h->rbSig[0], &t) != HSA_STATUS_SUCCESS || t.end <= t.start) return -1;
            const double sec = (double)(t.end - t.start) * h->hsaTickSeconds;
            if (best < 0 || sec < best) best = sec;
        }
        return best;
    };
    if (probe >= ((size_t)1 << 16)) {
        double bestSec = time_engine(0);
        uint32_t mask = 0;
        if (hsa_amd_memory_copy_engine_status(S.cpu, S.gpu, &mask) == HSA_STATUS_SUCCESS)
            for (uint32_t bit = 1; bit && bit <= mask; bit <<= 1) {
                if (!(mask & bit)) continue;
                const double sec = time_engine(bit);
                if (sec > 0 && (bestSec < 0 || sec < 0.9 * bestSec)) { bestSec = sec; h->sdmaEngine = bit; }
            }
        if (bestSec < 0) { // no engine moved the probe: leave the read-back to the HIP runtime
            (void)hsa_signal_destroy(h->rbSig[0]);
            (void)hsa_signal_destroy(h->rbSig[1]);
            h->hsaTickSeconds = 0;
            (void)hsa_shut_down();
            return;
        }
        memset(h->hostPos, 0, probe);
    }
    h->sdmaOk = true;
    if (getenv("SPH_STEP_TRACE")) fprintf(stderr, "sph: read-back through SDMA engine id 0x%x (0 = the HSA runtime's choice)\n", h->sdmaEngine);
}

// wait for the SDMA copy out of devPos[slot] (if one is in flight); its duration goes to kt.readback
int sdma_wait(sph_handle *h, int slot) {
    if (!h->rbPending[slot]) return SPH_OK;
    for (int tries = 0;; ++tries) { // 60 x 0.5 s: a copy that never completes is an error, not a hang
        if (hsa_signal_wait_scacquire(h->rbSig[slot], HSA_SIGNAL_CONDITION_LT, 1, (uint64_t)(0.5 / h->hsaTickSeconds),
                                      HSA_WAIT_STATE_BLOCKED) < 1) break;
        if (tries >= 60) return fail(h, SPH_EHIP, "read-back copy did not complete");
    }
    hsa_amd_profiling_async_copy_time_t t{};
    if (hsa_amd_profiling_get_async_copy_time(h->rbSig[slot], &t) == HSA_STATUS_SUCCESS && t.end >= t.start)
        h->kt.readback += (double)(t.end - t.start) * h->hsaTickSeconds;
    h->rbPending[slot] = false;
    return SPH_OK;
}

// the host has seen the force sweep that filled devPos[slot] finish: copy it out
int sdma_issue(sph_handle *h, int slot) {
    for (int b = 0; b < 2; ++b) // (a HIP copy of an untimed step still writing the same host buffer)
        if (h->copyPending[b]) {
            HIPCHK(h, hipEventSynchronize(h->copyDone[b]));
            h->copyPending[b] = false;
        }
    int rc = sdma_wait(h, slot);
    if (rc) return rc;
    hsa_signal_t dep = h->rbSig[slot ^ 1];
    const bool haveDep = h->rbPending[slot ^ 1]; // copies land in one host buffer: one after the other
    hsa_signal_store_relaxed(h->rbSig[slot], 1);
    const size_t bytes = (size_t)h->n * 3 * sizeof(float);
    hsa_status_t st = HSA_STATUS_ERROR;
    if (h->sdmaEngine)
        st = hsa_amd_memory_async_copy_on_engine(h->hostPos, h->hsaCpu, h->devPos[slot], h->hsaGpu, bytes, haveDep ? 1 : 0,
                                                 haveDep ? &dep : nullptr, h->rbSig[slot], (hsa_amd_sdma_engine_id_t)h->sdmaEngine, false);
    if (st != HSA_STATUS_SUCCESS) // (no engine picked, or it is busy: the HSA runtime's own choice)
        st = hsa_amd_memory_async_copy(h->hostPos, h->hsaCpu, h->devPos[slot], h->hsaGpu, bytes, haveDep ? 1 : 0,
                                       haveDep ? &dep : nullptr, h->rbSig[slot]);
    if (st != HSA_STATUS_SUCCESS) {
        h->sdmaOk = false; // fall back to the runtime's copy, now and from here on
        rc = sdma_wait(h, slot ^ 1);
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(h->hostPos, h->devPos[slot], (size_t)h->n * 3 * sizeof(float), hipMemcpyDeviceToHost, h->copy));
        HIPCHK(h, hipEventRecord(h->copyDone[slot], h->copy));
        h->copyPending[slot] = true;
        return SPH_OK;
    }
    h->rbPending[slot] = true;
    return SPH_OK;
}

// forget a grid that was built ahead for a state that is no longer the current one
void drop_grid_ahead(sph_handle *h) {
    if (!h->gridAhead) return;
    h->gridAhead = false;
    if (h->aheadEv) h->aheadEv->used = false; // (its density / force events were never recorded)
    h->aheadEv = nullptr;
    h->gridValid = false;
    h->phase = 0;
}

void state_replaced(sph_handle *h) {
    drop_grid_ahead(h);
    (void)sdma_wait(h, 0);
    (void)sdma_wait(h, 1);
    h->rbDeferredSlot = -1;
    h->clickValid = false;
    h->cur = 0;
    drop_step_graphs(h);
    h->ready = true;
    h->gridValid = false;
    h->phase = 0;
    h->sorted = -1;
    h->stepIndex = 0;
    h->copyPending[0] = h->copyPending[1] = false;
}

int upload_common(sph_handle *h, const float *pos, const float *vel, int n) {
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state)");
    if (n != h->n) return fail(h, SPH_EINVAL, "particle count differs from settings");
    const float hh = h->settings.h;
    const int D = h->P.D;
    std::vector<float4> p4((size_t)n), v4((size_t)n);
    int zmin = D, zmax = -1;
    for (int i = 0; i < n; ++i) {
        float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        // (the range test comes BEFORE the conversion: float -> int of a NaN or of a value beyond INT_MAX is
        // undefined behaviour on the host -- found by the UBSan build under the fuzz tests' NaN upload)
        const float qx = x / hh, qy = y / hh, qz = z / hh, Df = (float)D;
        if (!(qx >= 0.f && qx < Df && qy >= 0.f && qy < Df && qz >= 0.f && qz < Df && x >= 0.f && y >= 0.f && z >= 0.f))
            return fail(h, SPH_EINVAL, "position outside the simulation box");
        const int cz = (int)qz;
        zmin = cz < zmin ? cz : zmin;
        zmax = cz > zmax ? cz : zmax;
        uint32_t id = (uint32_t)i;
        float idbits;
        memcpy(&idbits, &id, 4);
        p4[i] = make_float4(x, y, z, idbits);
        v4[i] = vel ? make_float4(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2], 0.f)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    HIPCHK(h, hipStreamSynchronize(h->compute));
    HIPCHK(h, hipStreamSynchronize(h->copy));
    h->cur = 0;
    if (n > 0) {
        int rc = staged_upload(h, h->pos4[0], (size_t)n,
                               [&](size_t k, float4 *dst, size_t cnt) { memcpy(dst, p4.data() + k, cnt * sizeof(float4)); });
        if (!rc)
            rc = staged_upload(h, h->vel4[0], (size_t)n,
                               [&](size_t k, float4 *dst, size_t cnt) { memcpy(dst, v4.data() + k, cnt * sizeof(float4)); });
        if (rc) return rc;
    }
    HIPCHK(h, hipDeviceSynchronize());
    h->zLayers = zmax >= zmin ? zmax - zmin + 1 : 0;
    state_replaced(h);
    // getPosition() right after an upload shows the uploaded state (like setup(): simulator.cu:430-456 fills the
    // host array it hands out)
    h->hostPosIsInit = false;
    if (h->hostPos && n > 0) memcpy(h->hostPos, pos, (size_t)n * 3 * sizeof(float));
    return SPH_OK;
}

// A slab is a dozen z-layers: with the fixed chunk -> XCD map the floor pile of every layer (the lowest y band
// = the first eighth of a layer's rows) lands on the same XCD and the launch waits for it -- N = 8 slabs of the
// headline run over 100 steps, slowest slab: density 0.196 -> 0.149, force 0.285 -> 0.187 ms per step with the map
// moved on by one XCD per layer.  (The single domain's 80 layers: within noise either way, default off.)
static int slab_rotate(const sph_handle *h) { return h->tileRotate >= 0 ? h->tileRotate : 1; }

// xcd_tile() chunk: an eighth of one z-layer's worth of 256-particle tiles.
int tile_chunk(const sph_handle *h, int count, int layers) {
    if (h->tileChunkEnv >= 0) return h->tileChunkEnv;
    if (layers <= 0) return 0;
    const long long tiles = ((long long)count + 255) / 256;
    return (int)(tiles / (8ll * layers)); // 0 (contiguous eighths) when a layer is under 8 tiles
}

SweepArgs make_sweep_args(sph_handle *h) {
    SweepArgs A{};
    const int s = h->sorted;
    A.pos4 = h->pos4[s];
    A.vel4 = h->vel4[s];
    A.cellRange = h->cellRange;
    A.keys = h->ws.keys[h->sortedKeyBuf];
    A.pos_out = h->pos4[s ^ 1];
    A.vel_out = h->vel4[s ^ 1];
    A.host_order_pos = nullptr;
    A.force_out = h->force4;
    A.pairCounter = nullptr;
    A.stampCounter = (h->opt.flags & SPH_FLAG_COUNT_PAIRS) ? h->pairCounter : nullptr;
    A.i_begin = 0;
    A.i_end = h->n;
    A.i_origin = 0;
    A.i_begin2 = A.i_end2 = 0;
    A.nblk1 = 0;
    A.patchHalo = 0;
    A.n_all = h->n;
    A.tileChunk = tile_chunk(h, h->n, h->zLayers);
    A.tileRotate = h->tileRotate > 0 ? h->tileRotate : 0;
    A.maskPool = h->maskPool;
    A.maskOff = h->maskOff;
    A.noneList = h->noneList;
    A.hitCount = h->hitCount;
    A.maskCursor = h->maskCursor;
    A.maskCapacity = h->maskCapacity;
    A.pv8 = h->pv8;
    // (slabs: the wave origin is rounded down to a multiple of 64, the gather launch clears the array,
    // so halo rows -- whose densities arrive after the density sweep -- stay "not quiet")
    A.quiet = (h->useQuiet && h->quiet) ? h->quiet : nullptr;
    A.calm = h->calm;
    A.quietAll = A.quiet ? reinterpret_cast<uint32_t *>(h->quietVref + 1) : nullptr;
    A.quietHalo = nullptr; // (slab launches next to a halo layer set it: slab_halo_quiet)
    A.rhoToVel4 = h->external ? 1 : 0;
    A.listHead = reinterpret_cast<const int *>(h->cellRange);
    A.listNext = reinterpret_cast<const int *>(h->ws.vals[0]);
    return A;
}

// what rides on the gather launch of a grid build: the hit-stream cursors are cleared there
GatherExtras gather_extras(sph_handle *h) {
    GatherExtras X;
    if (h->maskCursor) {
        X.cursor = h->maskCursor;
        X.cursorWords = (int)(kCursorBytes / sizeof(unsigned long long));
        h->cursorClean = true;
    }
    if (h->quiet && h->useQuiet) {
        X.vref = h->quietVref;
        X.calm = h->calm;
        X.quietAll = reinterpret_cast<uint32_t *>(h->quietVref + 1);
        if (h->external) { // single domain: the density sweep rewrites every word each step
            X.quietClear = h->quiet;
            X.quietWords = (int)(2 * (((size_t)h->cap + 63) / 64) + 2);
        }
    }
    return X;
}

int begin_step_events(sph_handle *h) {
    StepEvents &se = h->ring[h->ringHead];
    if (se.used) {
        int rc = resolve_events(h, se);
        if (rc) return rc;
    }
    se.used = true;
    se.counted = false;
    se.hasCopy = false;
    h->curEv = &se;
    h->ringHead = (h->ringHead + 1) % kEventRing;
    return SPH_OK;
}

int resolve_pair(sph_handle *h, PairEvent &pe) {
    if (!pe.used) return SPH_OK;
    float ms = 0.f;
    HIPCHK(h, hipEventSynchronize(pe.b));
    HIPCHK(h, hipEventElapsedTime(&ms, pe.a, pe.b));
    *pe.target += ms * 1e-3;
    pe.used = false;
    return SPH_OK;
}

// begin a timed section whose GPU time is added to *target when resolved
int pair_begin(sph_handle *h, double *target, PairEvent **out, hipStream_t stream = nullptr) {
    PairEvent &pe = h->pairs[h->pairHead];
    int rc = resolve_pair(h, pe);
    if (rc) return rc;
    h->pairHead = (h->pairHead + 1) % kPairRing;
    pe.target = target;
    pe.used = true;
    HIPCHK(h, hipEventRecord(pe.a, stream ? stream : h->compute));
    *out = &pe;
    return SPH_OK;
}

int slab_range_ok(sph_handle *h, int buf, int i_begin, int i_end, int n_all) {
    if (!h->external || !h->pos4[0]) return fail(h, SPH_ESTATE, "sph_bind_buffers first");
    if ((buf != 0 && buf != 1) || i_begin < 0 || i_end < i_begin || n_all < i_end || n_all > h->cap)
        return fail(h, SPH_EINVAL, "bad slab range");
    return SPH_OK;
}

} // namespace

extern "C" {

int sph_set_stream(sph_handle *h, void *hip_stream) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->compute));
    if (!h->ownCompute) h->ownCompute = h->compute;
    // NULL is HIP's default ("null") stream -- what torch.cuda.current_stream()
    // reports unless the caller switched streams.
    h->compute = (hipStream_t)hip_stream;
    return SPH_OK;
}

int sph_bind_buffers(sph_handle *h, void *pos4_a, void *vel4_a, void *pos4_b, void *vel4_b,
                     int capacity) {
    if (!h) return SPH_EINVAL;
    if (!h->external) return fail(h, SPH_ESTATE, "create with SPH_FLAG_EXTERNAL_STATE");
    if (!pos4_a || !vel4_a || !pos4_b || !vel4_b || capacity > h->cap || capacity < 0)
        return fail(h, SPH_EINVAL, "bad buffers / capacity exceeds options.capacity");
    h->pos4[0] = (float4 *)pos4_a;
    h->vel4[0] = (float4 *)vel4_a;
    h->pos4[1] = (float4 *)pos4_b;
    h->vel4[1] = (float4 *)vel4_b;
    return SPH_OK;
}

void *sph_get_stream(sph_handle *h) { return h ? (void *)h->compute : nullptr; }
void *sph_slab_records(sph_handle *h) { return (h && h->opt.sweep == SPH_SWEEP_LIST) ? (void *)h->pv8 : nullptr; }

int sph_slab_sort_async(sph_handle *h, int src_buf, int src_offset, int count,
                        const uint32_t *thresholds, int nthr, void *bounds_dev_out) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = slab_range_ok(h, src_buf, 0, 0, 0);
    if (rc) return rc;
    if (src_offset < 0 || count < 0 || (long long)src_offset + count > h->cap || nthr < 0 ||
        nthr > 8 || (nthr > 0 && !thresholds))
        return fail(h, SPH_EINVAL, "bad sort range");
    hipStream_t s = h->compute;
    PairEvent *pe = nullptr;
    if ((rc = pair_begin(h, &h->kt.sort, &pe))) return rc;
    h->ws.velSample = (h->quiet && h->useQuiet) ? h->vel4[src_buf] + src_offset : nullptr; // the filter's reference velocity
    h->ws.vrefOut = h->quietVref;
    int res = sph_sort_cells(h->ws, h->P, h->pos4[src_buf] + src_offset, count, key_bits(h), s, h->cellRange,
                             h->P.numCells); // (clears the cell table too)
    // the segment bounds (and the element count, [nthr]) and the clearing of the hit-stream
    // cursors ride on the gather launch
    GatherExtras X = gather_extras(h);
    if (nthr > 0) {
        for (int k = 0; k < nthr; ++k) X.thr.v[k] = thresholds[k];
        X.nthr = nthr;
        X.bounds = bounds_dev_out ? static_cast<int *>(bounds_dev_out) : h->boundsDev;
    }
    sph_launch_gather(h->pos4[src_buf] + src_offset, h->vel4[src_buf] + src_offset,
                      h->ws.vals[res], h->ws.keys[res], h->pos4[src_buf ^ 1],
                      h->vel4[src_buf ^ 1], h->pv8, h->cellRange, count, s, X);
    HIPCHK(h, hipEventRecord(pe->b, s));
    if (nthr == 4) // [zlo, zlo+1, zhi-1, zhi] * D*D: the slab's owned z-layers
        h->zLayers = (int)((thresholds[3] - thresholds[0]) / (uint32_t)(h->P.D * h->P.D));
    HIPCHK(h, hipGetLastError());
    h->sorted = src_buf ^ 1;
    h->sortedKeyBuf = res;
    h->gridValid = true;
    h->slabOwnedEnd = h->slabOwnedBegin = 0; // (a new sorted array: no density sweep has vouched for any row of it yet)
    return SPH_OK;
}

int sph_slab_sort(sph_handle *h, int src_buf, int src_offset, int count,
                  const uint32_t *thresholds, int nthr, int32_t *bounds_out) {
    if (!h) return SPH_EINVAL;
    if (nthr > 0 && !bounds_out) return fail(h, SPH_EINVAL, "bad sort range");
    int rc = sph_slab_sort_async(h, src_buf, src_offset, count, thresholds, nthr, nullptr);
    if (rc) return rc;
    if (nthr > 0) {
        HIPCHK(h, hipMemcpyAsync(h->boundsHost, h->boundsDev, nthr * sizeof(int),
                                 hipMemcpyDeviceToHost, h->compute));
        HIPCHK(h, hipStreamSynchronize(h->compute));
        for (int k = 0; k < nthr; ++k) bounds_out[k] = h->boundsHost[k];
    }
    return SPH_OK;
}

int sph_slab_partition_async(sph_handle *h, int src_buf, int src_offset, int count,
                             const uint32_t *thresholds, int nthr, void *bounds_dev_out) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = slab_range_ok(h, src_buf, 0, 0, 0);
    if (rc) return rc;
    if (src_offset < 0 || count < 0 || (long long)src_offset + count > h->cap || nthr < 1 ||
        nthr > 8 || !thresholds)
        return fail(h, SPH_EINVAL, "bad partition range");
    for (int k = 1; k < nthr; ++k)
        if (thresholds[k] < thresholds[k - 1]) return fail(h, SPH_EINVAL, "thresholds must ascend");
    hipStream_t s = h->compute;
    PairEvent *pe = nullptr;
    if ((rc = pair_begin(h, &h->kt.sort, &pe))) return rc;
    Thresholds T{};
    for (int k = 0; k < nthr; ++k) T.v[k] = thresholds[k];
    // two launches (count per tile, move): grid.hip
    sph_launch_partition(h->P, h->pos4[src_buf] + src_offset, h->vel4[src_buf] + src_offset,
                         h->pos4[src_buf ^ 1], h->vel4[src_buf ^ 1], T, nthr, count, h->partTiles,
                         h->boundsDev, s);
    HIPCHK(h, hipEventRecord(pe->b, s));
    if (bounds_dev_out)
        HIPCHK(h, hipMemcpyAsync(bounds_dev_out, h->boundsDev, (nthr + 1) * sizeof(int),
                                 hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipGetLastError());
    h->gridValid = false;
    return SPH_OK;
}

int sph_slab_partition(sph_handle *h, int src_buf, int src_offset, int count,
                       const uint32_t *thresholds, int nthr, int32_t *bounds_out,
                       void *bounds_dev_out) {
    if (!h) return SPH_EINVAL;
    if (!bounds_out) return fail(h, SPH_EINVAL, "bad partition range");
    int rc = sph_slab_partition_async(h, src_buf, src_offset, count, thresholds, nthr, bounds_dev_out);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->boundsHost, h->boundsDev, nthr * sizeof(int), hipMemcpyDeviceToHost, h->compute));
    HIPCHK(h, hipStreamSynchronize(h->compute));
    for (int k = 0; k < nthr; ++k) bounds_out[k] = h->boundsHost[k];
    return SPH_OK;
}

int sph_slab_copy_segments(sph_handle *h, int dst_buf, int nseg, const void *const *src_pos,
                           const void *const *src_vel, const int32_t *counts,
                           const int32_t *dst_offsets) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = slab_range_ok(h, dst_buf, 0, 0, 0);
    if (rc) return rc;
    if (nseg < 0 || nseg > 8 || (nseg > 0 && (!src_pos || !src_vel || !counts || !dst_offsets)))
        return fail(h, SPH_EINVAL, "bad segment list");
    SegmentTable T{};
    T.n = nseg;
    T.prefix[0] = 0;
    for (int k = 0; k < nseg; ++k) {
        if (counts[k] < 0 || dst_offsets[k] < 0 || (long long)dst_offsets[k] + counts[k] > h->cap ||
            (counts[k] > 0 && (!src_pos[k] || !src_vel[k])))
            return fail(h, SPH_EINVAL, "segment outside the bound buffers");
        T.spos[k] = static_cast<const float4 *>(src_pos[k]);
        T.svel[k] = static_cast<const float4 *>(src_vel[k]);
        T.dst[k] = dst_offsets[k];
        T.prefix[k + 1] = T.prefix[k] + counts[k];
    }
    sph_launch_copy_segments(T, h->pos4[dst_buf], h->vel4[dst_buf], h->compute);
    HIPCHK(h, hipGetLastError());
    return SPH_OK;
}

int sph_slab_density(sph_handle *h, int buf, int i_begin, int i_end, int n_all) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = slab_range_ok(h, buf, i_begin, i_end, n_all);
    if (rc) return rc;
    if (!h->gridValid || h->sorted != buf) return fail(h, SPH_ESTATE, "sph_slab_sort into this buffer first");
    SweepArgs A = make_sweep_args(h);
    A.i_begin = i_begin;
    A.i_end = i_end;
    A.i_origin = i_begin & ~63; // hit-stream waves = whole words of the zero-pair filter's bit array
    h->slabOwnedBegin = i_begin;
    h->slabOwnedEnd = i_end;
    A.n_all = n_all;
    A.tileChunk = tile_chunk(h, i_end - i_begin, h->zLayers);
    A.tileRotate = slab_rotate(h);
    A.force_out = nullptr;
    if (h->opt.flags & SPH_FLAG_COUNT_PAIRS) A.pairCounter = h->pairCounter;
    PairEvent *pe = nullptr;
    if (h->maskCursor && !h->cursorClean) HIPCHK(h, hipMemsetAsync(h->maskCursor, 0, kCursorBytes, h->compute));
    h->cursorClean = false;
    if ((rc = pair_begin(h, &h->kt.density, &pe))) return rc;
    sph_launch_density(h->P, A, h->opt.math_mode, h->opt.sweep, h->compute);
    HIPCHK(h, hipEventRecord(pe->b, h->compute));
    HIPCHK(h, hipGetLastError());
    return SPH_OK;
}

int sph_slab_force(sph_handle *h, int buf, int i_begin, int i_end, int n_all) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = slab_range_ok(h, buf, i_begin, i_end, n_all);
    if (rc) return rc;
    if (!h->gridValid || h->sorted != buf) return fail(h, SPH_ESTATE, "sph_slab_sort into this buffer first");
    SweepArgs A = make_sweep_args(h);
    A.i_begin = i_begin;
    A.i_end = i_end;
    A.i_origin = i_begin & ~63;
    A.quietAll = nullptr; // (rows next to the halo layers, whose quiet bits nobody computed here: no all-quiet skip)
    A.patchHalo = 1;
    A.n_all = n_all;
    A.tileChunk = tile_chunk(h, i_end - i_begin, h->zLayers);
    A.tileRotate = slab_rotate(h);
    A.force_out = nullptr;
    PairEvent *pe = nullptr;
    if ((rc = pair_begin(h, &h->kt.force, &pe))) return rc;
    sph_launch_force(h->P, A, h->opt.math_mode, h->opt.sweep, h->compute);
    HIPCHK(h, hipEventRecord(pe->b, h->compute));
    HIPCHK(h, hipGetLastError());
    h->kt.steps += 1;
    return SPH_OK;
}

int sph_slab_patch_halo(sph_handle *h, int buf, int i_begin, int i_end, int n_all, void *hip_stream) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = slab_range_ok(h, buf, i_begin, i_end, n_all);
    if (rc) return rc;
    if (!h->gridValid || h->sorted != buf) return fail(h, SPH_ESTATE, "sph_slab_sort into this buffer first");
    if (h->opt.sweep != SPH_SWEEP_LIST) return SPH_OK; // the other sweeps read vel4 directly
    SweepArgs A = make_sweep_args(h);
    A.i_begin = i_begin;
    A.i_end = i_end;
    A.n_all = n_all;
    sph_launch_patch_halo(A, hip_stream ? (hipStream_t)hip_stream : h->compute);
    HIPCHK(h, hipGetLastError());
    return SPH_OK;
}

int sph_slab_apply_click(sph_handle *h, int buf, int mx, int my, int z_lo, int z_hi) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (!h->external || !h->pos4[0]) return fail(h, SPH_ESTATE, "sph_bind_buffers first");
    if (buf != 0 && buf != 1) return fail(h, SPH_EINVAL, "bad buffer index");
    if (h->opt.sweep == SPH_SWEEP_LINKED)
        return fail(h, SPH_ESTATE, "the click impulse is not available with SPH_SWEEP_LINKED");
    if (!h->gridValid || h->sorted != (buf ^ 1))
        return fail(h, SPH_ESTATE, "click needs a completed slab step (it reuses that step's grid)");
    sph_launch_click(h->P, h->cellRange, h->vel4[buf], mx, my, h->compute, z_lo, z_hi);
    HIPCHK(h, hipGetLastError());
    return SPH_OK;
}

int sph_slab_force_ranges(sph_handle *h, int buf, int i_origin, int a0, int b0, int a1, int b1,
                          int n_all, int last, void *hip_stream) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = slab_range_ok(h, buf, a0, b0 > a0 ? b0 : a0, n_all);
    if (!rc) rc = slab_range_ok(h, buf, a1, b1 > a1 ? b1 : a1, n_all);
    if (rc) return rc;
    if (i_origin < 0 || (b0 > a0 && i_origin > a0) || (b1 > a1 && (i_origin > a1 || a1 < b0)))
        return fail(h, SPH_EINVAL, "bad wave origin / ranges must ascend");
    if (!h->gridValid || h->sorted != buf) return fail(h, SPH_ESTATE, "sph_slab_sort into this buffer first");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->compute;
    if (b0 > a0 || b1 > a1) {
        SweepArgs A = make_sweep_args(h);
        A.i_origin = i_origin & ~63; // (the same rounding as sph_slab_density)
        A.patchHalo = 0;
        A.n_all = n_all;
        A.tileRotate = slab_rotate(h);
        if (A.quietAll) {
            // the last launch of the step holds the rows next to the halo layers (exchange B is through on this
            // stream): "every row is quiet" needs the halo rows' word too.  Earlier launches are interior rows by
            // contract -- every neighbour an owned row.
            if (last && h->opt.sweep == SPH_SWEEP_LIST && h->slabOwnedEnd > h->slabOwnedBegin && h->slabOwnedEnd <= n_all) {
                uint32_t *halo = A.quietAll + 1;
                sph_launch_halo_quiet(h->pv8, h->slabOwnedBegin, h->slabOwnedEnd, n_all, h->quietVref, A.quietAll, halo, s);
                A.quietHalo = halo;
            } else if (last) {
                A.quietAll = nullptr;
            }
        }
        A.force_out = nullptr;
        PairEvent *pe = nullptr;
        if ((rc = pair_begin(h, &h->kt.force, &pe, s))) return rc;
        if (h->opt.sweep == SPH_SWEEP_LIST) { // both ranges in one launch: one grid, one tail
            A.i_begin = a0;
            A.i_end = b0 > a0 ? b0 : a0;
            A.i_begin2 = a1;
            A.i_end2 = b1 > a1 ? b1 : a1;
            A.tileChunk = tile_chunk(h, (b0 > a0 ? b0 - a0 : 0) + (b1 > a1 ? b1 - a1 : 0), h->zLayers);
            sph_launch_force(h->P, A, h->opt.math_mode, h->opt.sweep, s);
        } else {
            for (int k = 0; k < 2; ++k) {
                A.i_begin = k ? a1 : a0;
                A.i_end = k ? b1 : b0;
                if (A.i_end <= A.i_begin) continue;
                A.tileChunk = tile_chunk(h, A.i_end - A.i_begin, h->zLayers);
                sph_launch_force(h->P, A, h->opt.math_mode, h->opt.sweep, s);
            }
        }
        HIPCHK(h, hipEventRecord(pe->b, s));
        HIPCHK(h, hipGetLastError());
    }
    if (last) h->kt.steps += 1;
    return SPH_OK;
}

const char *sph_build_info(void) {
    return "libsph_hip gfx950 (MI355X/CDNA4), api v2, strict-fp32 sweeps, "
           "8/10-bit LSD radix grid build";
}

int sph_default_settings(SphSettings *out, int numParticles, int randomInit) {
    if (!out) return SPH_EINVAL;
    // main.cpp:57-63
    float h = .1f;
    float h_pow_6 = (float)pow((double)h, 6.0);
    float h_pow_9 = (float)pow((double)h, 9.0);
    float v_kernel_coeff = 45.f / (3.14159265f * h_pow_6);
    float d_kernel_coeff = 315.f / (64.f * 3.14159265f * h_pow_9);
    memset(out, 0, sizeof(*out));
    out->randomInit = randomInit ? 1 : 0;
    out->numParticles = numParticles;
    out->h = h;
    out->v_kernel_coeff = v_kernel_coeff;
    out->d_kernel_coeff = d_kernel_coeff;
    out->boxDim = 10.f;
    out->numCellsPerDim = 100;
    out->timestep = (float).01;
    return SPH_OK;
}

int sph_initial_positions(const SphSettings *settings, float *pos_xyz) {
    if (!settings || (!pos_xyz && settings->numParticles > 0) || settings->numParticles < 0)
        return SPH_EINVAL;
    int written = init_positions_reference(*settings, pos_xyz);
    if (written < settings->numParticles) {
        fprintf(stderr,
                "sph: -i grid holds at most %d lattice points in the reference "
                "(simulator.cu:439-452); n=%d uses the dense-lattice EXTENSION\n",
                written, settings->numParticles);
        init_positions_dense(*settings, pos_xyz);
    }
    return SPH_OK;
}

int sph_create(const SphSettings *settings, const SphOptions *options, sph_handle **out) {
    if (!settings || !out) return fail(nullptr, SPH_EINVAL, "null argument");
    *out = nullptr;
    if (settings->numParticles < 0) return fail(nullptr, SPH_EINVAL, "numParticles < 0");
    if (!(settings->h > 0.f) || !(settings->numCellsPerDim >= 1.f) ||
        settings->numCellsPerDim > 1024.f)
        return fail(nullptr, SPH_EINVAL, "bad h / numCellsPerDim");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, SPH_ENODEV,
                    "no HIP device: libsph_hip has no CPU fallback by design");
    sph_handle *h = new (std::nothrow) sph_handle();
    if (!h) return fail(nullptr, SPH_ENOMEM, "out of host memory");
    h->settings = *settings;
    if (options) {
        size_t sz = options->struct_size > 0 ? (size_t)options->struct_size : sizeof(SphOptions);
        memcpy(&h->opt, options, sz < sizeof(SphOptions) ? sz : sizeof(SphOptions));
    } else {
        h->opt.device = -1;
    }
    h->opt.struct_size = (int32_t)sizeof(SphOptions);
    if (h->opt.math_mode != SPH_MATH_STRICT && h->opt.math_mode != SPH_MATH_FAST) {
        delete h;
        return fail(nullptr, SPH_EINVAL, "unknown math_mode");
    }
    if (h->opt.sweep < SPH_SWEEP_LIST || h->opt.sweep > SPH_SWEEP_LINKED) {
        delete h;
        return fail(nullptr, SPH_EINVAL, "unknown sweep variant");
    }
    if (h->opt.math_mode == SPH_MATH_FAST &&
        (h->opt.sweep == SPH_SWEEP_DIRECT || h->opt.sweep == SPH_SWEEP_LINKED)) {
        delete h;
        return fail(nullptr, SPH_EINVAL, "SPH_MATH_FAST does not exist for SPH_SWEEP_DIRECT/LINKED");
    }
    if (h->opt.key_order != SPH_KEY_FLATTENED && h->opt.key_order != SPH_KEY_MORTON) {
        delete h;
        return fail(nullptr, SPH_EINVAL, "unknown key_order");
    }
    if (h->opt.key_order == SPH_KEY_MORTON &&
        (h->opt.sweep != SPH_SWEEP_DIRECT || (h->opt.flags & SPH_FLAG_EXTERNAL_STATE))) {
        delete h;
        return fail(nullptr, SPH_EINVAL, "SPH_KEY_MORTON is served by SPH_SWEEP_DIRECT, single domain, only");
    }
    if (h->opt.sweep == SPH_SWEEP_LINKED && (h->opt.flags & SPH_FLAG_EXTERNAL_STATE)) {
        delete h;
        return fail(nullptr, SPH_EINVAL, "SPH_SWEEP_LINKED is single-domain only");
    }
    h->n = settings->numParticles;
    // a slab handle (caller-owned streams) is sized by its capacity alone: a GPU of an N-GPU run
    // holds ~1/N of the numParticles its settings name
    if ((h->opt.flags & SPH_FLAG_EXTERNAL_STATE) && h->opt.capacity > 0) h->cap = h->opt.capacity;
    else h->cap = h->opt.capacity > h->n ? h->opt.capacity : h->n;
    if (const char *e = getenv("SPH_TILE_CHUNK")) h->tileChunkEnv = atoi(e); // tuning studies
    if (const char *e = getenv("SPH_XCD_ROTATE")) h->tileRotate = atoi(e);
    if (const char *e = getenv("SPH_STEP_TRACE")) h->trace = atoi(e) != 0;
    fill_params(h);
    int rc = SPH_OK;
    do {
        if (h->opt.device >= 0) {
            if (h->opt.device >= count) { rc = fail(nullptr, SPH_EINVAL, "device ordinal out of range"); break; }
            if (hipSetDevice(h->opt.device) != hipSuccess) { rc = fail(nullptr, SPH_EHIP, "hipSetDevice failed"); break; }
        }
        if (hipGetDevice(&h->device) != hipSuccess) { rc = fail(nullptr, SPH_EHIP, "hipGetDevice failed"); break; }
        if (hipStreamCreateWithFlags(&h->compute, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking) != hipSuccess) {
            rc = fail(nullptr, SPH_EHIP, "hipStreamCreate failed");
            break;
        }
        rc = alloc_device(h);
        if (rc) g_create_error = h->err;
    } while (0);
    if (rc) {
        sph_destroy(h);
        return rc;
    }
    *out = h;
    return SPH_OK;
}

void sph_destroy(sph_handle *h) {
    if (!h) return;
    if (h->trace && h->trSteps > 0)
        fprintf(stderr, "sph step trace (host, us per timed step over %lld steps): enqueue %.1f | wait for the compute stream %.1f | "
                        "after the wait %.1f | caller between two steps %.1f\n", h->trSteps, h->trEnqueue / h->trSteps * 1e6,
                h->trSync / h->trSteps * 1e6, h->trPost / h->trSteps * 1e6, h->trBetween / (h->trSteps > 1 ? h->trSteps - 1 : 1) * 1e6),
        fprintf(stderr, "  enqueue split: events %.1f | grid %.1f | density %.1f | force %.1f | read-back %.1f\n", h->trPh[0] / h->trSteps * 1e6,
                h->trPh[1] / h->trSteps * 1e6, h->trPh[2] / h->trSteps * 1e6, h->trPh[3] / h->trSteps * 1e6, h->trPh[4] / h->trSteps * 1e6);
    if (h->compute) (void)hipStreamSynchronize(h->compute);
    if (h->copy) (void)hipStreamSynchronize(h->copy);
    if (h->hsaTickSeconds > 0) { // (sdma_init got as far as creating the signals)
        (void)sdma_wait(h, 0);
        (void)sdma_wait(h, 1);
        (void)hsa_signal_destroy(h->rbSig[0]);
        (void)hsa_signal_destroy(h->rbSig[1]);
        (void)hsa_shut_down();
    }
    for (int b = 0; b < 2; ++b) {
        if (h->pos4[b] && !h->external) (void)hipFree(h->pos4[b]);
        if (h->vel4[b] && !h->external) (void)hipFree(h->vel4[b]);
        if (h->ws.keys[b]) (void)hipFree(h->ws.keys[b]);
        if (h->ws.vals[b]) (void)hipFree(h->ws.vals[b]);
        if (h->devPos[b] && !h->mappedPos) (void)hipFree(h->devPos[b]);
        if (h->computeDone[b]) (void)hipEventDestroy(h->computeDone[b]);
        if (h->copyDone[b]) (void)hipEventDestroy(h->copyDone[b]);
    }
    if (h->ws.blockHist) (void)hipFree(h->ws.blockHist);
    if (h->ws.digitTotal) (void)hipFree(h->ws.digitTotal);
    for (auto &t : h->cellTable) if (t) (void)hipFree(t);
    if (h->hostPos) (void)hipHostFree(h->hostPos);
    for (int b = 0; b < 2; ++b) {
        if (h->stage[b]) (void)hipHostFree(h->stage[b]);
        if (h->stageFree[b]) (void)hipEventDestroy(h->stageFree[b]);
    }
    if (h->force4) (void)hipFree(h->force4);
    if (h->pairCounter) (void)hipFree(h->pairCounter);
    if (h->pairHost) (void)hipHostFree(h->pairHost);
    for (auto &se : h->ring) {
        for (auto &e : se.e) if (e) (void)hipEventDestroy(e);
        for (auto &e : se.c) if (e) (void)hipEventDestroy(e);
    }
    for (auto &se : h->graphEv) {
        for (auto &e : se.e) if (e) (void)hipEventDestroy(e);
        for (auto &e : se.c) if (e) (void)hipEventDestroy(e);
    }
    for (auto &gs : h->stepGraph)
        for (auto &g : gs)
            if (g) (void)hipGraphExecDestroy(g);
    if (h->pv8) (void)hipFree(h->pv8);
    if (h->maskPool) (void)hipFree(h->maskPool);
    if (h->maskOff) (void)hipFree(h->maskOff);
    if (h->noneList) (void)hipFree(h->noneList);
    if (h->hitCount) (void)hipFree(h->hitCount);
    if (h->maskCursor) (void)hipFree(h->maskCursor);
    if (h->quiet) (void)hipFree(h->quiet);
    if (h->quietVref) (void)hipFree(h->quietVref);
    if (h->calm) (void)hipFree(h->calm);
    if (h->initPos4) (void)hipFree(h->initPos4);
    if (h->oobHost) (void)hipHostFree(h->oobHost);
    if (h->boundsDev) (void)hipFree(h->boundsDev);
    if (h->partTiles) (void)hipFree(h->partTiles);
    if (h->boundsHost) (void)hipHostFree(h->boundsHost);
    for (auto &pe : h->pairs) {
        if (pe.a) (void)hipEventDestroy(pe.a);
        if (pe.b) (void)hipEventDestroy(pe.b);
    }
    if (h->ownCompute) h->compute = h->ownCompute;
    if (h->compute) (void)hipStreamDestroy(h->compute);
    if (h->copy) (void)hipStreamDestroy(h->copy);
    delete h;
}

int sph_setup(sph_handle *h) {
    if (!h) return SPH_EINVAL;
    const int n = h->n;
    SPH_ON_DEVICE(h);
    if (h->initPos4 && !h->external) {
        // setup() again on the same handle (bench.py and the tests go back to the initial condition
        // after their warm-up steps): the initial streams are restored from a device-resident copy
        // instead of 12.6 M rand() calls, a validation pass and 134 MB over PCIe -- ~150 ms during
        // which the GPU idles and after which its clocks need ~20 steps to come back
        // (scripts/studies/clock_ramp.py: density sweep 0.63 -> 0.52 -> 0.47 ms over steps 1..40
        // right after an upload, 0.48 flat behind a busy GPU).
        HIPCHK(h, hipStreamSynchronize(h->compute));
        HIPCHK(h, hipStreamSynchronize(h->copy));
        HIPCHK(h, hipMemcpyAsync(h->pos4[0], h->initPos4, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, h->compute));
        HIPCHK(h, hipMemsetAsync(h->vel4[0], 0, (size_t)n * sizeof(float4), h->compute));
        HIPCHK(h, hipStreamSynchronize(h->compute));
        h->zLayers = h->initZLayers;
        state_replaced(h);
        h->hostPosIsInit = true; // (getPosition() before the next step: sph_positions_host copies them then)
        return SPH_OK;
    }
    std::vector<float> pos((size_t)(n > 0 ? n : 1) * 3, 0.f);
    int rc = sph_initial_positions(&h->settings, pos.data());
    if (rc) return fail(h, rc, "sph_initial_positions failed");
    rc = upload_common(h, pos.data(), nullptr, n);
    if (rc || n <= 0 || h->external) return rc;
    // keep the initial streams (best effort: without the copy the next setup() recomputes them)
    if (hipMalloc(&h->initPos4, (size_t)n * sizeof(float4)) == hipSuccess) {
        if (hipMemcpy(h->initPos4, h->pos4[0], (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice) == hipSuccess) {
            h->initZLayers = h->zLayers;
        } else {
            (void)hipFree(h->initPos4);
            h->initPos4 = nullptr;
        }
    } else {
        h->initPos4 = nullptr;
    }
    (void)hipGetLastError();
    return SPH_OK;
}

int sph_upload_state(sph_handle *h, const float *pos_xyz, const float *vel_xyz, int n) {
    if (!h || (!pos_xyz && n > 0)) return fail(h, SPH_EINVAL, "null argument");
    return upload_common(h, pos_xyz, vel_xyz, n);
}

int sph_phase_grid(sph_handle *h) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state): use the sph_slab_* entry points");
    if (!h->ready) return fail(h, SPH_ESTATE, "setup()/upload_state() must come first");
    if (h->gridAhead) { // built by the previous timed step for exactly this state
        h->gridAhead = false;
        // (a caller that does not go on with the build's events -- the phase API -- drops them: the
        // density / force events of this entry would never be recorded)
        if (h->aheadEv && h->curEv != h->aheadEv) h->aheadEv->used = false;
        h->aheadEv = nullptr;
        return SPH_OK;  // (phase is 1 already)
    }
    if (h->phase != 0 && h->phase != 3) return fail(h, SPH_ESTATE, "grid phase out of order");
    hipStream_t s = h->compute;
    StepEvents *ev = h->curEv;
    const int c = h->cur, n = h->n;
    if (ev) HIPCHK(h, hipEventRecord(ev->e[0], s));
    if (h->opt.sweep == SPH_SWEEP_LINKED) {
        // the reference's grid: list heads reset (kernelResetGrid :321-326), then one
        // atomic push per particle (kernelBuildGrid :133-147).  No sort, no gather:
        // the streams stay where they are, in particle-id order.
        int *head = reinterpret_cast<int *>(h->cellRange);
        int *next = reinterpret_cast<int *>(h->ws.vals[0]);
        HIPCHK(h, hipMemsetAsync(head, 0xFF, (size_t)h->P.numCells * sizeof(int), s));
        if (ev) HIPCHK(h, hipEventRecord(ev->e[1], s));
        if (ev) HIPCHK(h, hipEventRecord(ev->e[2], s));
        sph_launch_link_build(h->P, h->pos4[c], head, next, n, s);
        if (ev) HIPCHK(h, hipEventRecord(ev->e[3], s));
        HIPCHK(h, hipGetLastError());
        h->sorted = c;
        h->gridValid = false; // no cell-range table in this mode
        h->phase = 1;
        return SPH_OK;
    }
    // kernelResetGrid (simulator.cu:321-326,492-495) and the cell hash are both part of the
    // first sort pass: no launch of their own
    if (ev) HIPCHK(h, hipEventRecord(ev->e[1], s));
    h->cellCur ^= 1; // the previous build's table stays intact (a click after a step pipelined ahead needs it)
    h->cellRange = h->cellTable[h->cellCur];
    h->ws.velSample = (h->quiet && h->useQuiet) ? h->vel4[c] : nullptr; // the zero-pair filter's reference velocity
    h->ws.vrefOut = h->quietVref;
    int res = sph_sort_cells(h->ws, h->P, h->pos4[c], n, key_bits(h), s, h->cellRange, h->P.numCells);
    if (ev) HIPCHK(h, hipEventRecord(ev->e[2], s));
    // the list sweeps take velocities from the interleaved records: no sorted vel4 copy
    float4 *velSorted = (h->opt.sweep == SPH_SWEEP_LIST && h->pv8) ? nullptr : h->vel4[c ^ 1];
    sph_launch_gather(h->pos4[c], h->vel4[c], h->ws.vals[res], h->ws.keys[res],
                      h->pos4[c ^ 1], velSorted, h->pv8, h->cellRange, n, s, gather_extras(h));
    if (ev) HIPCHK(h, hipEventRecord(ev->e[3], s));
    HIPCHK(h, hipGetLastError());
    h->sorted = c ^ 1;
    h->sortedKeyBuf = res;
    h->gridValid = true;
    h->phase = 1;
    return SPH_OK;
}

int sph_phase_density(sph_handle *h) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->phase != 1) return fail(h, SPH_ESTATE, "density phase needs the grid phase first");
    SweepArgs A = make_sweep_args(h);
    if (h->opt.flags & SPH_FLAG_COUNT_PAIRS) A.pairCounter = h->pairCounter;
    if (h->maskCursor && !h->cursorClean) HIPCHK(h, hipMemsetAsync(h->maskCursor, 0, kCursorBytes, h->compute));
    h->cursorClean = false;
    sph_launch_density(h->P, A, h->opt.math_mode, h->opt.sweep, h->compute);
    if (h->curEv) HIPCHK(h, hipEventRecord(h->curEv->e[4], h->compute));
    HIPCHK(h, hipGetLastError());
    h->phase = 2;
    return SPH_OK;
}

int sph_phase_force(sph_handle *h) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->phase != 2) return fail(h, SPH_ESTATE, "force phase needs the density phase first");
    SweepArgs A = make_sweep_args(h);
    const int slot = (int)(h->stepIndex & 1);
    if (!(h->opt.flags & SPH_FLAG_NO_READBACK)) {
        // devPos[slot] was last read by the copy of step k-2
        if (h->copyPending[slot] && !h->capturing) { // (graph mode: waited for before the launch)
            HIPCHK(h, hipStreamWaitEvent(h->compute, h->copyDone[slot], 0));
            h->copyPending[slot] = false;
        }
        if (h->rbPending[slot]) { // an SDMA copy (timed step k-2) has no stream to wait on: the host waits
            int rc = sdma_wait(h, slot);
            if (rc) return rc;
        }
        A.host_order_pos = h->devPos[slot];
    }
    sph_launch_force(h->P, A, h->opt.math_mode, h->opt.sweep, h->compute);
    if (h->curEv) HIPCHK(h, hipEventRecord(h->curEv->e[5], h->compute));
    HIPCHK(h, hipGetLastError());
    h->cur = h->sorted ^ 1; // new state, still in this step's sorted order
    h->phase = 3;
    h->hostPosIsInit = false;
    h->clickTable = h->cellRange;
    h->clickValid = h->opt.sweep != SPH_SWEEP_LINKED;
    return SPH_OK;
}

int sph_phase_readback(sph_handle *h) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->phase != 3) return fail(h, SPH_ESTATE, "readback needs the force phase first");
    if (h->opt.flags & SPH_FLAG_NO_READBACK) {
        h->stepIndex++;
        h->phase = 0;
        return SPH_OK;
    }
    const int slot = (int)(h->stepIndex & 1);
    if (h->mappedPos) { // the force sweep already wrote the host buffer
        h->stepIndex++;
        h->phase = 0;
        return SPH_OK;
    }
    if (h->sdmaOk && h->stepTimed && !h->capturing && h->n > 0) {
        h->rbDeferredSlot = slot; // sph_step issues the copy once it has seen this step's force sweep finish
        h->stepIndex++;
        h->phase = 0;
        return SPH_OK;
    }
    for (int b = 0; b < 2; ++b) { // (SDMA copies of earlier timed steps write the same host buffer)
        int rc = sdma_wait(h, b);
        if (rc) return rc;
    }
    HIPCHK(h, hipEventRecord(h->computeDone[slot], h->compute));
    HIPCHK(h, hipStreamWaitEvent(h->copy, h->computeDone[slot], 0));
    if (h->curEv) HIPCHK(h, hipEventRecord(h->curEv->c[0], h->copy));
    // (measured, round 3: the copy in 4 / 16 / 64 pieces takes 0.93 / 1.04 / 1.50 ms instead of 0.90)
    if (h->n > 0)
        HIPCHK(h, hipMemcpyAsync(h->hostPos, h->devPos[slot], (size_t)h->n * 3 * sizeof(float),
                                 hipMemcpyDeviceToHost, h->copy));
    if (h->curEv) {
        HIPCHK(h, hipEventRecord(h->curEv->c[1], h->copy));
        h->curEv->hasCopy = true;
    }
    HIPCHK(h, hipEventRecord(h->copyDone[slot], h->copy));
    h->copyPending[slot] = true;
    h->stepIndex++;
    h->phase = 0;
    return SPH_OK;
}

namespace {

// Fold the per-kernel events of the graph replay of `slot` (two steps ago) into kt.
int fold_graph_events(sph_handle *h, int slot) {
    if (!h->graphEvPending[slot]) return SPH_OK;
    StepEvents &se = h->graphEv[slot];
    se.used = true;
    se.counted = false;
    h->graphEvPending[slot] = false;
    return resolve_events(h, se);
}

// Capture the three phases of one step (read-back slot `slot`) from the compute stream,
// one graph each: the timing events between them stay ordinary stream events (an event
// recorded INSIDE a graph cannot be timed: hipEventElapsedTime rejects it).
int capture_step_graph(sph_handle *h, int slot) {
    const int phase0 = h->phase, cur0 = h->cur, sorted0 = h->sorted, keybuf0 = h->sortedKeyBuf;
    const bool grid0 = h->gridValid;
    h->capturing = true;
    h->curEv = nullptr;
    bool ok = true;
    for (int ph = 0; ph < 3 && ok; ++ph) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(h->compute, hipStreamCaptureModeThreadLocal) != hipSuccess) { ok = false; break; }
        const int rc = ph == 0 ? sph_phase_grid(h) : ph == 1 ? sph_phase_density(h) : sph_phase_force(h);
        const hipError_t e = hipStreamEndCapture(h->compute, &g);
        ok = !rc && e == hipSuccess && g;
        if (ok) ok = hipGraphInstantiate(&h->stepGraph[slot][ph], g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
    }
    h->capturing = false;
    h->graphKeyBuf = h->sortedKeyBuf;
    h->graphCellCur[slot] = h->cellCur;
    // the captured calls only recorded work: restore the host-side state they advanced
    h->phase = phase0;
    h->cur = cur0;
    h->sorted = sorted0;
    h->sortedKeyBuf = keybuf0;
    h->gridValid = grid0;
    if (!ok) {
        (void)hipGetLastError();
        for (auto &x : h->stepGraph[slot]) {
            if (x) (void)hipGraphExecDestroy(x);
            x = nullptr;
        }
        return SPH_EHIP;
    }
    return SPH_OK;
}

} // namespace

int sph_step(sph_handle *h, SphTimes *times) {
    if (!h) return SPH_EINVAL;
    const auto trIn = std::chrono::steady_clock::now();
    if (h->trace && h->trSteps > 0) h->trBetween += std::chrono::duration<double>(trIn - h->trLastReturn).count();
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state): use the sph_slab_* entry points");
    if (!h->ready) return fail(h, SPH_ESTATE, "setup()/upload_state() must come first");
    if (h->phase != 0 && h->phase != 3 && !(h->gridAhead && h->phase == 1))
        return fail(h, SPH_ESTATE, "a step split into phases is still open");
    int rc;
    // slot of the PREVIOUS step's position copy (if any)
    const int prevSlot = (int)((h->stepIndex + 1) & 1);
    const bool prevCopy = h->stepIndex > 0 && (h->copyPending[prevSlot] || h->rbPending[prevSlot]);
    StepEvents *ev = nullptr;
    const int slot = (int)(h->stepIndex & 1);
    bool viaGraph = false;
    h->stepTimed = times != nullptr && !h->useGraph; // (the read-back phase picks the copy path by it)
    h->rbDeferredSlot = -1;
    if (h->useGraph && h->opt.sweep != SPH_SWEEP_LINKED && h->n > 0) {
        drop_grid_ahead(h); // (never set in graph mode; belt and braces)
        if ((rc = fold_graph_events(h, slot))) return rc;
        if (!h->stepGraph[slot][2] && capture_step_graph(h, slot) != SPH_OK) h->useGraph = false;
        if (h->stepGraph[slot][2]) {
            if (!(h->opt.flags & SPH_FLAG_NO_READBACK) && h->copyPending[slot]) {
                // devPos[slot] was last read by the copy of step k-2
                HIPCHK(h, hipStreamWaitEvent(h->compute, h->copyDone[slot], 0));
                h->copyPending[slot] = false;
            }
            ev = &h->graphEv[slot];
            hipStream_t cs = h->compute;
            // e[0] = e[1] | grid graph | e[2] = e[3] | density graph | e[4] | force graph | e[5]: the
            // whole grid build is booked as "sort" (its kernels are not timed one by one here)
            bool ok = hipEventRecord(ev->e[0], cs) == hipSuccess && hipEventRecord(ev->e[1], cs) == hipSuccess &&
                      hipGraphLaunch(h->stepGraph[slot][0], cs) == hipSuccess &&
                      hipEventRecord(ev->e[2], cs) == hipSuccess && hipEventRecord(ev->e[3], cs) == hipSuccess &&
                      hipGraphLaunch(h->stepGraph[slot][1], cs) == hipSuccess &&
                      hipEventRecord(ev->e[4], cs) == hipSuccess &&
                      hipGraphLaunch(h->stepGraph[slot][2], cs) == hipSuccess &&
                      hipEventRecord(ev->e[5], cs) == hipSuccess;
            if (ok) {
                viaGraph = true;
                ev->hasCopy = false;
                h->graphEvPending[slot] = true;
                h->curEv = ev; // the read-back below records its copy events here
                // what the three phase calls would have left behind
                h->sorted = h->cur ^ 1;
                h->sortedKeyBuf = h->graphKeyBuf;
                h->gridValid = true;
                h->cellCur = h->graphCellCur[slot];
                h->cellRange = h->cellTable[h->cellCur];
                h->clickTable = h->cellRange;
                h->clickValid = true;
                h->cur = h->sorted ^ 1;
                h->phase = 3;
                h->hostPosIsInit = false;
            } else {
                (void)hipGetLastError();
                h->useGraph = false;
            }
        }
    }
    auto trT = std::chrono::steady_clock::now();
    auto trLap = [&](int k) {
        if (!h->trace) return;
        const auto now = std::chrono::steady_clock::now();
        h->trPh[k] += std::chrono::duration<double>(now - trT).count();
        trT = now;
    };
    if (h->trace && !h->trBase && !viaGraph && !h->gridAhead) {
        if (hipEventCreate(&h->trBase) == hipSuccess) (void)hipEventRecord(h->trBase, h->compute);
    }
    if (!viaGraph) {
        if (h->gridAhead) { // the previous timed step queued this step's grid build (and recorded its events)
            ev = h->aheadEv;
            h->curEv = ev;
        } else {
            if ((rc = begin_step_events(h))) return rc;
            ev = h->curEv;
        }
        trLap(0);
        if ((rc = sph_phase_grid(h))) return rc; // (a grid built ahead is consumed here)
        trLap(1);
        if ((rc = sph_phase_density(h))) return rc;
        trLap(2);
        if ((rc = sph_phase_force(h))) return rc;
        trLap(3);
    }
    if ((rc = sph_phase_readback(h))) return rc; // ends the step
    h->stepTimed = false;
    trLap(4);
    h->curEv = nullptr;
    if (times) {
        const auto trA = std::chrono::steady_clock::now();
        if (h->aheadEnabled && !viaGraph && !h->useGraph && h->opt.sweep != SPH_SWEEP_LINKED && h->n > 0) {
            // queue the next step's grid build before waiting for this one (see sph_handle::gridAhead)
            if ((rc = begin_step_events(h))) return rc;
            StepEvents *nextEv = h->curEv;
            if ((rc = sph_phase_grid(h))) return rc;
            h->curEv = nullptr;
            h->aheadEv = nextEv;
            h->gridAhead = true;
            HIPCHK(h, hipEventSynchronize(ev->e[5])); // this step's force sweep (not the grid queued behind it)
        } else {
            HIPCHK(h, hipStreamSynchronize(h->compute));
        }
        const auto trB = std::chrono::steady_clock::now();
        if (h->trace) {
            h->trEnqueue += std::chrono::duration<double>(trA - trIn).count();
            h->trSync += std::chrono::duration<double>(trB - trA).count();
        }
        if (h->rbDeferredSlot >= 0) { // the force sweep is through: this step's positions leave through an SDMA engine
            const int s2 = h->rbDeferredSlot;
            h->rbDeferredSlot = -1;
            if ((rc = sdma_issue(h, s2))) return rc;
        }
        report_oob(h);
        float gridMs = 0.f, sphMs = 0.f;
        HIPCHK(h, hipEventElapsedTime(&gridMs, ev->e[0], ev->e[3]));
        HIPCHK(h, hipEventElapsedTime(&sphMs, ev->e[3], ev->e[5]));
        times->buildGrid += gridMs * 1e-3;
        times->sphUpdate += sphMs * 1e-3;
        // "Data transfer" = the part of the previous step's D2H that this
        // step's compute did not hide (the reference blocks on every copy,
        // simulator.cu:532-533; here step k's copy overlaps step k+1).
        if (prevCopy) {
            auto t0 = std::chrono::steady_clock::now();
            if (h->copyPending[prevSlot]) HIPCHK(h, hipEventSynchronize(h->copyDone[prevSlot]));
            if ((rc = sdma_wait(h, prevSlot))) return rc;
            times->memcpy +=
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        times->iters += 1;
        if (h->trace) {
            h->trLastReturn = std::chrono::steady_clock::now();
            h->trPost += std::chrono::duration<double>(h->trLastReturn - trB).count();
            h->trSteps++;
        }
    }
    return SPH_OK;
}

int sph_apply_click(sph_handle *h, int mx, int my) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state): use the sph_slab_* entry points");
    if (h->opt.sweep == SPH_SWEEP_LINKED)
        return fail(h, SPH_ESTATE, "the click impulse is not available with SPH_SWEEP_LINKED");
    if (!h->clickValid || (h->phase != 0 && !h->gridAhead) || h->stepIndex == 0)
        return fail(h, SPH_ESTATE, "click needs a completed step (it reuses that step's grid)");
    drop_grid_ahead(h); // a grid built ahead gathered the velocities this impulse is about to change
    sph_launch_click(h->P, h->clickTable, h->vel4[h->cur], mx, my, h->compute);
    HIPCHK(h, hipGetLastError());
    return SPH_OK;
}

const float *sph_positions_host(sph_handle *h) {
    if (!h) return nullptr;
    if (hipStreamSynchronize(h->compute) != hipSuccess ||
        hipStreamSynchronize(h->copy) != hipSuccess) {
        h->err = "stream synchronize failed";
        return nullptr;
    }
    if (sdma_wait(h, 0) || sdma_wait(h, 1)) return nullptr;
    if (h->hostPosIsInit && h->initPos4 && h->hostPos && h->n > 0) {
        // the initial streams are in id order: x, y, z of every 16-byte row
        if (hipMemcpy2D(h->hostPos, 3 * sizeof(float), h->initPos4, sizeof(float4), 3 * sizeof(float), (size_t)h->n,
                        hipMemcpyDeviceToHost) != hipSuccess) {
            h->err = "copy of the initial positions failed";
            return nullptr;
        }
    }
    h->hostPosIsInit = false;
    report_oob(h);
    return h->hostPos;
}

namespace {
struct SnapshotHeader { // 64 bytes
    char magic[8];      // "SPHSNAP1"
    int32_t n;
    int32_t reserved;
    int64_t stepIndex;
    SphSettings settings; // 32 bytes
    char pad[8];
};
static_assert(sizeof(SnapshotHeader) == 64, "snapshot header");
} // namespace

int sph_save_state(sph_handle *h, const char *path) {
    if (!h || !path) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state)");
    if (!h->ready || (h->phase != 0 && !h->gridAhead)) return fail(h, SPH_ESTATE, "no complete state to save");
    int rc = sph_sync(h);
    if (rc) return rc;
    const size_t n = (size_t)h->n;
    std::vector<float4> p4(n ? n : 1), v4(n ? n : 1);
    if (n) {
        HIPCHK(h, hipMemcpy(p4.data(), h->pos4[h->cur], n * sizeof(float4), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(v4.data(), h->vel4[h->cur], n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    SnapshotHeader hd{};
    memcpy(hd.magic, "SPHSNAP1", 8);
    hd.n = h->n;
    hd.stepIndex = h->stepIndex;
    hd.settings = h->settings;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(h, SPH_EINVAL, std::string("cannot open ") + path);
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && (n == 0 || (fwrite(p4.data(), sizeof(float4), n, f) == n &&
                                                                fwrite(v4.data(), sizeof(float4), n, f) == n));
    ok = (fclose(f) == 0) && ok;
    return ok ? SPH_OK : fail(h, SPH_EINVAL, "short write");
}

int sph_load_state(sph_handle *h, const char *path) {
    if (!h || !path) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state)");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(h, SPH_EINVAL, std::string("cannot open ") + path);
    SnapshotHeader hd{};
    bool ok = fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, "SPHSNAP1", 8) == 0;
    if (ok && (hd.n != h->n || memcmp(&hd.settings.h, &h->settings.h, 24) != 0)) {
        fclose(f);
        return fail(h, SPH_EINVAL, "snapshot does not match this simulator's settings");
    }
    const size_t n = (size_t)h->n;
    std::vector<float4> p4(n ? n : 1), v4(n ? n : 1);
    ok = ok && (n == 0 || (fread(p4.data(), sizeof(float4), n, f) == n && fread(v4.data(), sizeof(float4), n, f) == n));
    fclose(f);
    if (!ok) return fail(h, SPH_EINVAL, "not a snapshot / truncated");
    std::vector<char> seen(n ? n : 1, 0);
    int zmin = h->P.D, zmax = -1;
    for (size_t i = 0; i < n; ++i) { // ids must be a permutation: they index devicePosition
        uint32_t id;
        memcpy(&id, &p4[i].w, 4);
        if (id >= n || seen[id]) return fail(h, SPH_EINVAL, "corrupt snapshot (ids)");
        seen[id] = 1;
        // the same box / NaN check as sph_upload_state: a corrupt file must not inject NaNs
        const float x = p4[i].x, y = p4[i].y, z = p4[i].z, hh = h->settings.h;
        const float qx = x / hh, qy = y / hh, qz = z / hh, Df = (float)h->P.D; // (range test before any conversion)
        if (!(qx >= 0.f && qx < Df && qy >= 0.f && qy < Df && qz >= 0.f && qz < Df && x >= 0.f && y >= 0.f && z >= 0.f) ||
            !(v4[i].x == v4[i].x && v4[i].y == v4[i].y && v4[i].z == v4[i].z))
            return fail(h, SPH_EINVAL, "corrupt snapshot (position outside the simulation box / NaN)");
        const int cz = (int)qz;
        zmin = cz < zmin ? cz : zmin;
        zmax = cz > zmax ? cz : zmax;
    }
    h->zLayers = zmax >= zmin ? zmax - zmin + 1 : 0;
    HIPCHK(h, hipStreamSynchronize(h->compute));
    HIPCHK(h, hipStreamSynchronize(h->copy));
    h->cur = 0;
    if (n) {
        int rc = staged_upload(h, h->pos4[0], (size_t)n,
                               [&](size_t k, float4 *dst, size_t cnt) { memcpy(dst, p4.data() + k, cnt * sizeof(float4)); });
        if (!rc)
            rc = staged_upload(h, h->vel4[0], (size_t)n,
                               [&](size_t k, float4 *dst, size_t cnt) { memcpy(dst, v4.data() + k, cnt * sizeof(float4)); });
        if (rc) return rc;
    }
    HIPCHK(h, hipDeviceSynchronize());
    drop_step_graphs(h);
    drop_grid_ahead(h);
    h->clickValid = false;
    h->ready = true;
    h->gridValid = false;
    h->phase = 0;
    h->sorted = -1;
    h->stepIndex = hd.stepIndex;
    h->copyPending[0] = h->copyPending[1] = false;
    (void)sdma_wait(h, 0);
    (void)sdma_wait(h, 1);
    h->rbDeferredSlot = -1;
    h->hostPosIsInit = false;
    if (h->hostPos) // getPosition() shows the loaded state (id order)
        for (size_t i = 0; i < n; ++i) {
            uint32_t id;
            memcpy(&id, &p4[i].w, 4);
            h->hostPos[3 * (size_t)id] = p4[i].x;
            h->hostPos[3 * (size_t)id + 1] = p4[i].y;
            h->hostPos[3 * (size_t)id + 2] = p4[i].z;
        }
    return SPH_OK;
}

int sph_sync(sph_handle *h) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->compute));
    HIPCHK(h, hipStreamSynchronize(h->copy));
    int rc = sdma_wait(h, 0);
    if (!rc) rc = sdma_wait(h, 1);
    if (rc) return rc;
    report_oob(h);
    return SPH_OK;
}

int sph_num_particles(const sph_handle *h) { return h ? h->n : SPH_EINVAL; }
int sph_num_table_cells(const sph_handle *h) { return h ? h->P.numCells : SPH_EINVAL; }

int sph_download_state(sph_handle *h, float *pos, float *vel, float *rho, float *prs) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state): use the sph_slab_* entry points");
    if (!h->ready) return fail(h, SPH_ESTATE, "no state");
    int rc = sph_sync(h);
    if (rc) return rc;
    const int n = h->n;
    std::vector<float4> p4((size_t)(n > 0 ? n : 1)), v4((size_t)(n > 0 ? n : 1));
    if (n > 0) {
        HIPCHK(h, hipMemcpy(p4.data(), h->pos4[h->cur], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(v4.data(), h->vel4[h->cur], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < n; ++i) {
        uint32_t id;
        memcpy(&id, &p4[i].w, 4);
        if (id >= (uint32_t)n) return fail(h, SPH_EHIP, "corrupt particle id in device state");
        if (pos) { pos[3 * id] = p4[i].x; pos[3 * id + 1] = p4[i].y; pos[3 * id + 2] = p4[i].z; }
        if (vel) { vel[3 * id] = v4[i].x; vel[3 * id + 1] = v4[i].y; vel[3 * id + 2] = v4[i].z; }
        float r = v4[i].w;
        if (rho) rho[id] = r;
        // same expression as simulator.cu:188-189
        if (prs) prs[id] = fmaxf(0.f, SPH_GAS_CONSTANT * (r - SPH_REST_DENSITY));
    }
    return SPH_OK;
}

int sph_download_force(sph_handle *h, float *force_xyz) {
    if (!h || !force_xyz) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state): use the sph_slab_* entry points");
    if (!h->force4) return fail(h, SPH_ESTATE, "create with SPH_FLAG_STORE_FORCE");
    int rc = sph_sync(h);
    if (rc) return rc;
    const int n = h->n;
    std::vector<float4> f4((size_t)(n > 0 ? n : 1)), p4((size_t)(n > 0 ? n : 1));
    if (n > 0) {
        HIPCHK(h, hipMemcpy(f4.data(), h->force4, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(p4.data(), h->pos4[h->cur], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < n; ++i) {
        uint32_t id;
        memcpy(&id, &p4[i].w, 4);
        if (id >= (uint32_t)n) return fail(h, SPH_EHIP, "corrupt particle id in device state");
        force_xyz[3 * id] = f4[i].x;
        force_xyz[3 * id + 1] = f4[i].y;
        force_xyz[3 * id + 2] = f4[i].z;
    }
    return SPH_OK;
}

int sph_download_grid(sph_handle *h, uint32_t *ids, uint32_t *keys, int32_t *cell_ranges) {
    if (!h) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    if (h->external) return fail(h, SPH_ESTATE, "handle is in slab mode (external state): use the sph_slab_* entry points");
    if (!h->gridValid) return fail(h, SPH_ESTATE, "no grid built yet");
    int rc = sph_sync(h);
    if (rc) return rc;
    const int n = h->n;
    if (ids && n > 0) {
        std::vector<float4> p4((size_t)n);
        HIPCHK(h, hipMemcpy(p4.data(), h->pos4[h->sorted], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) memcpy(&ids[i], &p4[i].w, 4);
    }
    if (keys && n > 0)
        HIPCHK(h, hipMemcpy(keys, h->ws.keys[h->sortedKeyBuf], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cell_ranges)
        HIPCHK(h, hipMemcpy(cell_ranges, h->cellRange, (size_t)h->P.numCells * sizeof(int2), hipMemcpyDeviceToHost));
    return SPH_OK;
}

int sph_get_kernel_times(sph_handle *h, SphKernelTimes *out, int reset) {
    if (!h || !out) return SPH_EINVAL;
    SPH_ON_DEVICE(h);
    int rc = sph_sync(h);
    if (rc) return rc;
    for (auto &se : h->ring) {
        if (h->gridAhead && &se == h->aheadEv) continue; // a grid built ahead: its step has not run yet
        if ((rc = resolve_events(h, se))) return rc;
    }
    for (int slot = 0; slot < 2; ++slot) {
        if (!h->graphEvPending[slot]) continue;
        h->graphEv[slot].used = true;
        h->graphEv[slot].counted = false;
        h->graphEvPending[slot] = false;
        if ((rc = resolve_events(h, h->graphEv[slot]))) return rc;
    }
    for (auto &pe : h->pairs)
        if ((rc = resolve_pair(h, pe))) return rc;
    if (h->opt.flags & SPH_FLAG_COUNT_PAIRS) {
        HIPCHK(h, hipMemcpy(h->pairHost, h->pairCounter, kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        h->kt.pair_tests = h->pairHost[0];
        h->kt.pair_hits = 0;
        h->hitsRecorded = 0;
        for (int sh = 0; sh < 256; ++sh) { // sharded counters (one address would serialise the waves)
            h->kt.pair_tests += h->pairHost[16 + sh * 16];
            h->kt.pair_hits += h->pairHost[16 + sh * 16 + 14];   // bodies evaluated (after the zero-pair filter)
            h->hitsRecorded += h->pairHost[16 + sh * 16 + 15];   // hits in the stream
        }
    }
    *out = h->kt;
    if (reset) {
        h->kt = SphKernelTimes{};
        // (not hipMemset: the first use of the null stream makes the runtime create another
        // hardware queue, and that stalled the GPU's queues for ~7 ms a step or two later)
        HIPCHK(h, hipMemsetAsync(h->pairCounter, 0, kCounterWords * sizeof(unsigned long long), h->compute));
        HIPCHK(h, hipStreamSynchronize(h->compute));
    }
    return SPH_OK;
}

int sph_debug_counters(sph_handle *h, uint64_t *out16) {
    if (!h || !out16) return SPH_EINVAL;
    int rc = sph_sync(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(h->pairHost, h->pairCounter, kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 16; ++k) out16[k] = h->pairHost[k];
    for (int sh = 0; sh < 256; ++sh)
        for (int k = 0; k < 16; ++k) out16[k] += h->pairHost[16 + sh * 16 + k];
    return SPH_OK;
}

const char *sph_last_error(const sph_handle *h) {
    return h ? h->err.c_str() : g_create_error.c_str();
}

int sph_sort_check(int device, const uint32_t *keys, int n, int key_bits_, uint32_t *perm_out,
                   uint32_t *sorted_keys_out) {
    if (n < 0 || (n > 0 && !keys)) return SPH_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SPH_ENODEV;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return SPH_EHIP;
    if (n == 0) return SPH_OK;
    SortWorkspace ws{};
    int rc = SPH_OK;
    std::vector<uint32_t> iota((size_t)n);
    for (int i = 0; i < n; ++i) iota[i] = (uint32_t)i;
    size_t nb = sph_sort_workspace_blocks(n);
    bool ok = true;
    for (int b = 0; b < 2 && ok; ++b) {
        ok = ok && hipMalloc(&ws.keys[b], (size_t)n * 4) == hipSuccess;
        ok = ok && hipMalloc(&ws.vals[b], (size_t)n * 4) == hipSuccess;
    }
    ok = ok && hipMalloc(&ws.blockHist, 1024 * nb * 4) == hipSuccess;
    ok = ok && hipMalloc(&ws.digitTotal, 1024 * 4) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b) {
        ok = ok && hipMemset(ws.keys[b], 0, (size_t)n * 4) == hipSuccess;
        ok = ok && hipMemset(ws.vals[b], 0, (size_t)n * 4) == hipSuccess;
    }
    if (ok) {
        ws.capacity = n;
        ws.maxBlocks = (int)nb;
        ok = ok && hipMemcpy(ws.keys[0], keys, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(ws.vals[0], iota.data(), (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
        int res = sph_sort_pairs(ws, n, key_bits_, nullptr);
        ok = ok && hipDeviceSynchronize() == hipSuccess;
        if (ok && perm_out)
            ok = hipMemcpy(perm_out, ws.vals[res], (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && sorted_keys_out)
            ok = hipMemcpy(sorted_keys_out, ws.keys[res], (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) rc = SPH_EHIP;
    for (int b = 0; b < 2; ++b) {
        if (ws.keys[b]) (void)hipFree(ws.keys[b]);
        if (ws.vals[b]) (void)hipFree(ws.vals[b]);
    }
    if (ws.blockHist) (void)hipFree(ws.blockHist);
    if (ws.digitTotal) (void)hipFree(ws.digitTotal);
    return rc;
}

} // extern "C"
