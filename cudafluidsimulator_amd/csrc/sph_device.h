// Internal declarations shared by the HIP translation units of libsph_hip.so.
// gfx950 (MI355X / CDNA4) only: wavefront = 64 lanes is hard-coded throughout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPH_WAVE 64

// Physics constants of the reference (simulator.h:6-12, simulator.cu:13-14).
#define SPH_MASS 0.02f
#define SPH_GAS_CONSTANT 1.f
#define SPH_REST_DENSITY 1000.f
#define SPH_VISCOSITY 1.f
#define SPH_GRAVITY -9.8f
#define SPH_ELASTICITY 0.5f
#define SPH_EPS_F (1e-4f)
#define SPH_PUSH_STRENGTH (5.f)
#define SPH_BOX_MAX_X (600)
#define SPH_BOX_MIN_X (200)
#define SPH_BOX_MAX_Y (450)
#define SPH_BOX_MIN_Y (150)

// Run constants, passed to every kernel by value (they land in SGPRs; this
// replaces the reference's `__constant__ Settings deviceSettings`,
// simulator.cu:19,459).
struct DevParams {
    float h;       // smoothing radius = cell size
    float h2;      // h*h, rounded once in fp32 (simulator.cu:89,105)
    float vcoef;   // v_kernel_coeff
    float dcoef;   // d_kernel_coeff
    float boxDim;
    float boxHi;   // boxDim - h in fp32 (simulator.cu:283)
    float dt;      // timestep
    float cut2;    // largest dist2 for which ANY force term can be non-zero
    int D;         // cells per dimension (numCellsPerDim as int)
    int numCells;  // entries of the cell table: D^3 (flattened keys) or 8^ceil(log2 D) (Morton)
    int morton;    // key function: 0 = flattened cell index (simulator.cu:78-82), 1 = Morton
    int slimDiv;   // 1: h and the kernel coefficients are the reference's (main.cpp:57-63): the pair body may use
                   // the divide / square-root chains without range fix-ups (sweep_common.h); 0: full IEEE expansions
};

// The neighbour grid's key function.  Flattened = the reference's x + y D + z D^2; Morton =
// bits of x, y, z interleaved (x lowest), the ordering the reference's README.md:5 names
// for its `z_index_sort` branch (not in the checkout).  SPH_KEY_MORTON exists for the A/B
// of the two orderings and is served by the direct sweep only.
__device__ __forceinline__ uint32_t sph_spread3(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t sph_cell_key(const DevParams &P, int cx, int cy, int cz) {
    if (P.morton) return sph_spread3((uint32_t)cx) | (sph_spread3((uint32_t)cy) << 1) | (sph_spread3((uint32_t)cz) << 2);
    return (uint32_t)(cx + cy * P.D + cz * P.D * P.D);
}

// Particle state lives in two float4 streams, both in cell-sorted order:
//   pos4[i] = (x, y, z, bits(original particle id))
//   vel4[i] = (vx, vy, vz, rho_i)      rho filled by the density sweep
// 32 B per particle instead of the reference's 56-B AoS with a list pointer
// (simulator.h:33-51); a neighbour test touches 16 B.

// Out-of-grid positions met by the cell hash (the reference printf's "OOB particle" from the kernel,
// simulator.cu:60-73): host-mapped, written by the first sort pass, printed by the host.
#define SPH_OOB_RECORDS 8
struct SphOobLog {
    uint32_t count;
    struct { int cell[3]; float pos[3]; } rec[SPH_OOB_RECORDS];
};

// ---- radix sort (sort.hip) ----
struct SortWorkspace {
    uint32_t *keys[2];
    uint32_t *vals[2];
    uint32_t *blockHist;  // [digits <= 1024][numBlocks], digit-major
    uint32_t *digitTotal; // [digits]
    int capacity;         // elements
    int maxBlocks;
    SphOobLog *oob;       // device view of the handle's out-of-grid log (may be null)
    // zero-pair filter (optional): the first pass of a grid build samples 64 rows of velSample[0..n) and leaves the
    // most common velocity in *vrefOut -- the gather launch of the same build marks the rows that move with it
    const float4 *velSample;
    float4 *vrefOut;
};
size_t sph_sort_workspace_blocks(int n);
// Stable LSD sort of (keys[0], vals[0]) on `bits` key bits; returns the index
// (0/1) of the buffer pair that holds the result.
int sph_sort_pairs(const SortWorkspace &ws, int n, int bits, hipStream_t s);
struct DevParams;
// Same, for the grid build: the keys are the flattened cell indices of pos4[0..n), computed
// inside the first histogram pass (no separate hash kernel, no iota of values in memory);
// that pass also zeroes cellRange[0..numCells) (kernelResetGrid) for k_gather_cells.
int sph_sort_cells(const SortWorkspace &ws, const DevParams &P, const float4 *pos4, int n, int bits,
                   hipStream_t s, int2 *cellRange, int numCells);

// ---- grid build (grid.hip) ----
void sph_launch_hash(const DevParams &P, const float4 *pos4, uint32_t *keys,
                     uint32_t *vals, int n, hipStream_t s);
// bounds[k] = #keys < thr[k] over sorted keys (one binary search per lane)
struct Thresholds { uint32_t v[8]; };
// work that rides on the gather launch instead of a launch of its own (all optional)
struct GatherExtras {
    unsigned long long *cursor = nullptr; // hit-stream allocation cursors to clear ...
    int cursorWords = 0;                  // ... this many 8-byte words
    int *bounds = nullptr;                // slab path: bounds[t] = #keys < thr.v[t], bounds[nthr] = n
    const float4 *vref = nullptr;         // zero-pair filter: the reference velocity (picked by the first sort pass) ...
    unsigned long long *calm = nullptr;   // ... and one bit per sorted row "moves with it", a 64-bit word per 64 rows
    uint32_t *quietAll = nullptr;         // the "every (owned) row is quiet" word and, next to it, the "every halo row is
                                          // quiet" word (slabs) are set to 1 here
    uint32_t *quietClear = nullptr;       // slab path: the filter's bit array is cleared here (halo rows stay "not quiet") ...
    int quietWords = 0;                   // ... this many 32-bit words
    Thresholds thr{};
    int nthr = 0;
};
void sph_launch_gather(const float4 *pos_in, const float4 *vel_in,
                       const uint32_t *perm, const uint32_t *sorted_keys,
                       float4 *pos_out, float4 *vel_out, float4 *pv8, int2 *cellRange, int n,
                       hipStream_t s, const GatherExtras &X = GatherExtras());
void sph_launch_lower_bounds(const uint32_t *sorted_keys, int n, Thresholds thr, int nthr,
                             int *bounds_dev, hipStream_t s);
void sph_launch_classify(const DevParams &P, const float4 *pos4, Thresholds thr, int nthr,
                         uint32_t *keys, uint32_t *vals, int n, hipStream_t s);
// stable partition of [0, n) by key class in two launches (tileCount: sph_partition_tiles(n) x 9 ints)
void sph_launch_partition(const DevParams &P, const float4 *pos_in, const float4 *vel_in, float4 *pos_out,
                          float4 *vel_out, Thresholds thr, int nthr, int n, int *tileCount, int *bounds_dev,
                          hipStream_t s);
size_t sph_partition_tiles(int n);
void sph_launch_gather_plain(const float4 *pos_in, const float4 *vel_in, const uint32_t *perm,
                             float4 *pos_out, float4 *vel_out, int n, hipStream_t s);
struct SegmentTable {
    const float4 *spos[8];
    const float4 *svel[8];
    int dst[8];
    int prefix[9]; // rows before segment k; prefix[n] = total
    int n;
};
void sph_launch_copy_segments(const SegmentTable &T, float4 *dpos, float4 *dvel, hipStream_t s);
void sph_launch_click(const DevParams &P, const int2 *cellRange, float4 *vel4,
                      int mx, int my, hipStream_t s, int zlo = 0, int zhi = 1 << 30);

// Hit-stream pool of the list sweep: cut into sub-pools with one allocation cursor each
// (one cursor per 256 bytes), see k_density_mask_lds.
#define SL_POOL_SHARDS 64
#define SL_CURSOR_STRIDE 32 // unsigned long long per cursor slot

// ---- sweeps (sweeps.hip) ----
struct SweepArgs {
    const float4 *pos4;       // sorted positions (+id)
    float4 *vel4;             // sorted velocities (+rho): density writes .w
    const int2 *cellRange;    // {start,end} per flattened cell
    const uint32_t *keys;     // sorted flattened cell keys
    float4 *pos_out;          // force+integrate outputs (same sorted index)
    float4 *vel_out;
    float *host_order_pos;    // n x 3 floats by original id (may be null)
    float4 *force_out;        // optional (SPH_FLAG_STORE_FORCE)
    unsigned long long *pairCounter; // optional (SPH_FLAG_COUNT_PAIRS)
    unsigned long long *stampCounter; // diagnostic builds only (same buffer)
    int i_begin, i_end;       // owned range (whole array for one domain)
    int i_origin;             // list sweep: particle 0 of wave 0 of the hit stream (the density
                              // sweep's i_begin); a force launch may cover a sub-range of it
    int i_begin2, i_end2;     // list sweep force launch: an optional SECOND row range in the same
                              // launch (a slab's two boundary layers: one grid, one tail); empty = {0,0}
    int nblk1;                // ... blocks of the launch that belong to the first range
    int n_all;
    int patchHalo;            // list sweep force launch: copy the halo rows' vel4 into pv8 first
    int tileChunk;            // xcd_tile(): 256-particle tiles per chunk (1/8 z-layer), 0 = eighths
    int tileRotate;           // xcd_tile(): the chunk -> XCD assignment moves on by one every so many groups (0 = fixed)
    // SPH_SWEEP_LIST: hit bit streams handed from the density to the force sweep
    uint32_t *maskPool;               // pool of quads: two (first candidate, 32-bit mask) pairs each
    uint32_t *maskOff;                // per 64-particle wave: {first quad or ~0u, quads per lane}
    uint32_t *hitCount;               // per sorted row: hits the density sweep recorded (the force sweep deals rows to lanes by it)
    unsigned long long *maskCursor;   // quads handed out this step, per sub-pool: [SL_POOL_SHARDS][SL_CURSOR_STRIDE];
                                      // word 1 (cleared with them): waves that found their sub-pool exhausted ...
    uint32_t *noneList;               // ... and their numbers, in arrival order (k_force_fallback walks this list)
    unsigned long long maskCapacity;  // pool size in quads
    float4 *pv8;                      // interleaved (pos4, vel4) copy of the sorted streams
    uint32_t *quiet;                  // zero-pair filter (may be null = off): bit j of this array is set when sorted row j
                                      // has no pressure and moves with the reference velocity *quietVref; a hit between
                                      // two such rows adds exactly +-0 to the force and is dropped unread
    const unsigned long long *calm;   // bit j: sorted row j moves with the reference velocity (written by the gather launch)
    uint32_t *quietAll;               // 1 while EVERY row of this step's density sweep (slabs: every owned row) is quiet -- set by the
                                      // gather launch, cleared by the density sweep's first non-quiet row; the force
                                      // sweep then has no pair to evaluate and does not read its hit stream at all
    const uint32_t *quietHalo;        // slabs, launches that hold rows next to a halo layer (else null): 1 while every
                                      // HALO row of this step is quiet too (k_halo_quiet, after exchange B)
    int rhoToVel4;                    // list sweep: also store rho in vel4.w (slab halo exchange B)
    // SPH_SWEEP_LINKED: per-cell linked lists over the UNSORTED streams
    const int *listHead;              // [numCells] first particle of the cell or -1
    const int *listNext;              // [n] next particle of the same cell or -1
};
void sph_launch_halo_quiet(const float4 *pv8, int lo_end, int hi_begin, int n_all, const float4 *vref,
                           const uint32_t *ownedQuiet, uint32_t *haloQuiet, hipStream_t s);
// ---- linked-list backend (sweeps_linked.hip) ----
void sph_launch_link_build(const DevParams &P, const float4 *pos4, int *head, int *next, int n,
                           hipStream_t s);
void sph_launch_density_linked(const DevParams &P, const SweepArgs &A, hipStream_t s);
void sph_launch_force_linked(const DevParams &P, const SweepArgs &A, hipStream_t s);
void sph_launch_density_list(const DevParams &P, const SweepArgs &A, int mathMode, hipStream_t s);
void sph_launch_force_list(const DevParams &P, const SweepArgs &A, int mathMode, hipStream_t s);
void sph_launch_patch_halo(const SweepArgs &A, hipStream_t s);
void sph_launch_density(const DevParams &P, const SweepArgs &A, int mathMode,
                        int sweep, hipStream_t s);
void sph_launch_force(const DevParams &P, const SweepArgs &A, int mathMode,
                      int sweep, hipStream_t s);
