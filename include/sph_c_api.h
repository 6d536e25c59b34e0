/*
 * sph_c_api.h -- C-ABI drop-in boundary of the MI355X SPH step path.
 *
 * Plain C types only (no HIP, no torch).  This is what a host program in any
 * language binds (cgo / JNI / ctypes / the C++ `Simulator` class in
 * include/simulator.h).  The reference has no FFI of its own -- its boundary is
 * the C++ class in src/simulator.h:53-74 -- so each entry point cites the
 * reference method or code it replaces.  All functions return 0 on success and
 * a negative SPH_E* code on failure; sph_last_error() gives the message.
 *
 * Threading: a handle is used from one host thread at a time (the reference is
 * single-threaded, simulator.cu:462-546).  Work is queued on the handle's own
 * HIP streams; sph_positions_host()/sph_download_state()/sph_sync() block until
 * the data they expose is complete, so callers observe the reference's
 * "synchronous on return" behaviour.
 */
#ifndef SPH_C_API_H
#define SPH_C_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_API_VERSION 2 /* 2: + sph_slab_apply_click, sph_slab_records (additive; every v1 entry point unchanged) */

#define SPH_OK 0
#define SPH_EINVAL (-1)  /* bad argument / state outside the box       */
#define SPH_EHIP (-2)    /* a HIP runtime call failed                   */
#define SPH_ENOMEM (-3)  /* host or device allocation failed            */
#define SPH_ESTATE (-4)  /* call order violated (e.g. step before setup) */
#define SPH_ENODEV (-5)  /* no usable GPU                               */

/* Layout-identical to the reference's `struct Settings` (simulator.h:19-31):
 * bool, int, 6 floats = 32 bytes.  numCellsPerDim is a float there too. */
typedef struct SphSettings {
    uint8_t randomInit;
    uint8_t pad_[3];
    int32_t numParticles;
    float h;
    float v_kernel_coeff;
    float d_kernel_coeff;
    float boxDim;
    float numCellsPerDim;
    float timestep;
} SphSettings;

/* Layout-identical to the reference's `struct Times` (times.h:5-10). */
typedef struct SphTimes {
    double buildGrid;
    double sphUpdate;
    double memcpy;
    int32_t iters;
} SphTimes;

enum {
    SPH_MATH_STRICT = 0, /* every op individually rounded; bit-identical to oracle */
    SPH_MATH_FAST = 1    /* FMA contraction + approximate rcp/sqrt (tolerance-checked) */
};
enum {
    SPH_SWEEP_LIST = 0,   /* production: LDS-staged density sweep records one hit bit per
                             candidate, the force sweep walks the recorded hits */
    SPH_SWEEP_DIRECT = 1, /* check path: one thread per particle, direct global loads */
    SPH_SWEEP_LDS = 2,    /* LDS-staged window in both sweeps, hit FIFO in the force sweep */
    SPH_SWEEP_LINKED = 3  /* the reference's own neighbour structure: no sort, one atomically
                             built linked list per cell (simulator.cu:44-55,133-147), walked
                             as in :163-189/:207-251.  Summation order is a race, so results
                             match the oracle to rounding only, and differ run to run.  Kept to
                             time "the reference's algorithm on MI355X"; strict math, single
                             domain only; sph_apply_click / sph_download_grid / the slab entry
                             points return SPH_ESTATE. */
};
enum {
    SPH_FLAG_COUNT_PAIRS = 1, /* accumulate the candidate pair-test count per step */
    SPH_FLAG_STORE_FORCE = 2, /* keep per-particle force of the last step (tests)  */
    SPH_FLAG_NO_READBACK = 4, /* skip the per-step D2H of positions (kernel studies) */
    SPH_FLAG_EXTERNAL_STATE = 8, /* particle streams are caller-owned device buffers
                                   (sph_bind_buffers); used by the multi-GPU slab
                                   driver so halo send/recv is zero-copy */
    SPH_FLAG_MAPPED_POSITIONS = 16 /* the id-ordered positions of getPosition() are written by
                                   the force sweep STRAIGHT into host-mapped pinned memory
                                   (zero-copy: no per-step device->host copy); SURVEY.md 8f
                                   rank 3.  Measured slower than the overlapped copy at
                                   n = 4 M: kept as an option, not the default */
};

enum {
    SPH_KEY_FLATTENED = 0, /* x + y D + z D^2 (simulator.cu:78-82): the 27-cell walk is nine
                              contiguous runs of the sorted stream -- what every sweep is built on */
    SPH_KEY_MORTON = 1     /* bits of x, y, z interleaved (the ordering the reference's README.md:5
                              names for its z_index_sort branch; BASELINE config 3).  For the A/B of
                              the two orderings: SPH_SWEEP_DIRECT, strict math, single domain only */
};

typedef struct SphOptions {
    int32_t struct_size; /* = sizeof(SphOptions) */
    int32_t device;      /* HIP device ordinal; -1 = current */
    int32_t math_mode;   /* SPH_MATH_* */
    int32_t sweep;       /* SPH_SWEEP_* */
    int32_t flags;       /* SPH_FLAG_* */
    int32_t capacity;    /* particle slots to allocate (0 = numParticles); slabs
                            need room for halo + migrants */
    int32_t key_order;   /* SPH_KEY_* (callers built against the 24-byte struct get FLATTENED) */
} SphOptions;

/* Per-kernel GPU time, accumulated from HIP events recorded on the handle's
 * compute stream (seconds).  `steps` = number of steps accumulated. */
typedef struct SphKernelTimes {
    double hash, sort, gather, density, force, readback;
    uint64_t pair_tests; /* sum over steps, if SPH_FLAG_COUNT_PAIRS */
    int64_t steps;
    uint64_t pair_hits;  /* SPH_FLAG_COUNT_PAIRS + SPH_SWEEP_LIST: pair bodies the force sweep
                            evaluates = candidates inside the support radius (popcount of the
                            recorded hit masks) minus the pairs its zero-pair filter drops
                            (neither row under pressure, same velocity: the pair adds exactly
                            +-0), summed over steps; 0 for the other sweeps.  The unfiltered
                            count is word 15 of sph_debug_counters() */
} SphKernelTimes;

typedef struct sph_handle sph_handle;

/* main.cpp:57-63 -- the constants main() derives before constructing Settings. */
int sph_default_settings(SphSettings *out, int numParticles, int randomInit);

/* The host half of Simulator::setup (simulator.cu:430-453) on its own: fills
 * numParticles x (x,y,z) with the reference initial condition.  Needs no GPU
 * (the slab driver splits this array across ranks). */
int sph_initial_positions(const SphSettings *settings, float *pos_xyz);

/* Simulator::Simulator (simulator.cu:370-375).  Copies *settings. */
int sph_create(const SphSettings *settings, const SphOptions *options,
               sph_handle **out);
/* Simulator::~Simulator (simulator.cu:377-405). */
void sph_destroy(sph_handle *h);

/* Simulator::setup (simulator.cu:411-460): allocate, reference initial
 * conditions (random: glibc rand() as if never seeded; grid: 0.09 lattice),
 * upload.  n > 109^3 in grid mode is outside the reference's domain and uses
 * the labelled "dense lattice" extension. */
int sph_setup(sph_handle *h);
/* Replaces setup()'s initialiser with caller state in particle-id order
 * (xyz interleaved; vel may be NULL = zero).  Positions must lie in the box. */
int sph_upload_state(sph_handle *h, const float *pos_xyz, const float *vel_xyz,
                     int n);

/* Simulator::simulate (times == NULL, simulator.cu:462-497) and
 * Simulator::simulateAndTime (times != NULL, simulator.cu:499-546). */
int sph_step(sph_handle *h, SphTimes *times);
/* kernelMoveParticles (simulator.cu:329-367) on the last step's grid; this is
 * what simulate() runs when `mouseClicked` is set (simulator.cu:482-489). */
int sph_apply_click(sph_handle *h, int mouse_x, int mouse_y);

/* Simulator::getPosition (simulator.cu:407-409): numParticles x (x,y,z),
 * original particle-id order, owned by the handle, valid until the next step.
 * Blocks until the last step's device->host copy has landed. */
const float *sph_positions_host(sph_handle *h);
/* Full state in particle-id order; any pointer may be NULL. rho/prs are the
 * values kernelUpdatePressureAndDensity produced in the last step. */
int sph_download_state(sph_handle *h, float *pos_xyz, float *vel_xyz,
                       float *rho, float *prs);
int sph_download_force(sph_handle *h, float *force_xyz); /* SPH_FLAG_STORE_FORCE */
/* Sorted-order view of the last grid build: ids, flattened cell keys
 * (pre-integration), and the {start,end} cell table (numCells x 2 ints). */
int sph_download_grid(sph_handle *h, uint32_t *ids, uint32_t *keys,
                      int32_t *cell_ranges);

/* Checkpoint / resume (no reference counterpart).  The file
 * is a raw snapshot of the two cell-sorted float4 streams (ids included) plus a
 * 64-byte header, so a resumed run continues BIT-IDENTICALLY (the canonical
 * summation order depends on the stored order, not only on positions). */
int sph_save_state(sph_handle *h, const char *path);
int sph_load_state(sph_handle *h, const char *path);

int sph_sync(sph_handle *h);
int sph_num_particles(const sph_handle *h);
/* entries of the cell table sph_download_grid fills: D^3, or 8^ceil(log2 D) with Morton keys */
int sph_num_table_cells(const sph_handle *h);
int sph_get_kernel_times(sph_handle *h, SphKernelTimes *out, int reset);
const char *sph_last_error(const sph_handle *h); /* h may be NULL: create errors */
/* Diagnostics: [0] = pair tests; [1..] = in-kernel phase stamps, filled only by a
 * -DSW_STAMPS=1 build of the library (see sweeps.hip), zero otherwise. */
int sph_debug_counters(sph_handle *h, uint64_t *out16);

/* ---- the step split into its phases (tests, profiling, slab driver) ---- */
int sph_phase_grid(sph_handle *h);     /* kernelBuildGrid + kernelResetGrid */
int sph_phase_density(sph_handle *h);  /* kernelUpdatePressureAndDensity    */
int sph_phase_force(sph_handle *h);    /* kernelUpdateForces + UpdatePositions */
int sph_phase_readback(sph_handle *h); /* the per-step D2H (simulator.cu:479) */

/* Stand-alone check of the radix sort used by the grid build: stable sort of
 * n (key, index) pairs; writes the permutation (host pointers). */
int sph_sort_check(int device, const uint32_t *keys, int n, int key_bits,
                   uint32_t *perm_out, uint32_t *sorted_keys_out);

/* ---- z-slab decomposition (one handle per GPU; cudafluidsimulator_amd/slab.py) ----
 * No reference counterpart: the reference is single-GPU.  The slab
 * driver owns four float4 device buffers per rank -- pos4[2], vel4[2], `capacity`
 * particles each -- and exchanges halo / migrant ranges of them with RCCL
 * send/recv; the library only runs kernels on ranges of those buffers, on the
 * caller's stream (so it is stream-ordered with the collectives). */
int sph_set_stream(sph_handle *h, void *hip_stream); /* a hipStream_t; NULL = HIP's default stream */
int sph_bind_buffers(sph_handle *h, void *pos4_a, void *vel4_a, void *pos4_b,
                     void *vel4_b, int capacity);
/* Hash + stable radix sort + gather of particles [src_offset, src_offset+count)
 * of buffer pair `src_buf` into [0, count) of the other pair, and the cell table
 * over the result.  bounds_out[k] = number of sorted keys < thresholds[k]
 * (k < nthr <= 8); blocks until those counts are on the host. */
int sph_slab_sort(sph_handle *h, int src_buf, int src_offset, int count,
                  const uint32_t *thresholds, int nthr, int32_t *bounds_out);
/* Stable PARTITION of particles [src_offset, src_offset+count) of buffer pair
 * `src_buf` into [0, count) of the other pair by key class (class = number of
 * thresholds <= the particle's new cell key; previous order kept inside a class).
 * What a rank needs before the exchange: [migrants down | lower boundary layer |
 * interior | upper boundary layer | migrants up] as contiguous ranges, at a third of
 * the launches of a full sort.  bounds_out[k] = number of particles with key <
 * thresholds[k].  Builds no cell table (sph_slab_sort of the combined array does).
 * bounds_dev_out (may be NULL): DEVICE pointer that receives nthr+1 int32 -- the
 * bounds, then `count` -- stream-ordered, so the driver can send them to the
 * neighbours as a message header without a round trip through the host. */
int sph_slab_partition(sph_handle *h, int src_buf, int src_offset, int count,
                       const uint32_t *thresholds, int nthr, int32_t *bounds_out,
                       void *bounds_dev_out);
/* Assemble the combined array: copy nseg <= 8 row ranges -- src_pos[k]/src_vel[k] are
 * DEVICE pointers to counts[k] float4 rows each (ranges of the bound buffers or of the
 * driver's receive buffers) -- to rows dst_offsets[k].. of buffer pair `dst_buf`, in one
 * kernel launch on the handle's stream. */
int sph_slab_copy_segments(sph_handle *h, int dst_buf, int nseg, const void *const *src_pos,
                           const void *const *src_vel, const int32_t *counts,
                           const int32_t *dst_offsets);
/* kernelUpdatePressureAndDensity for particles [i_begin, i_end) of the n_all
 * sorted particles in buffer pair `buf` (halo particles are candidates only). */
int sph_slab_density(sph_handle *h, int buf, int i_begin, int i_end, int n_all);
/* kernelUpdateForces + kernelUpdatePositions for [i_begin, i_end); new state is
 * written to the same indices of the other buffer pair. */
int sph_slab_force(sph_handle *h, int buf, int i_begin, int i_end, int n_all);

/* ---- the same phases WITHOUT a host round trip (include/sph_mgpu.h's in-process driver) ----
 * Everything is queued on the handle's stream (sph_get_stream: a hipStream_t) and
 * nothing blocks: the bounds are only written to DEVICE memory (bounds_dev_out:
 * nthr+1 int32 = the bounds, then `count`), from where the driver sends them to the
 * neighbours / copies them to pinned memory.  sph_slab_force_ranges runs the force +
 * integration sweep for row ranges of the owned range whose hit stream was recorded
 * by sph_slab_density(buf, i_origin, ...): the interior rows can run while the halo
 * densities are still in flight, the two boundary layers after sph_slab_patch_halo.
 * Every launch before the one flagged last_launch_of_the_step must hold INTERIOR rows only
 * (every neighbour an owned row): the last one also tests the halo rows for the zero-pair
 * filter's "every row is quiet" shortcut, the earlier ones rely on the owned rows alone. */
void *sph_get_stream(sph_handle *h);
int sph_slab_partition_async(sph_handle *h, int src_buf, int src_offset, int count,
                             const uint32_t *thresholds, int nthr, void *bounds_dev_out);
int sph_slab_sort_async(sph_handle *h, int src_buf, int src_offset, int count,
                        const uint32_t *thresholds, int nthr, void *bounds_dev_out);
int sph_slab_patch_halo(sph_handle *h, int buf, int i_begin, int i_end, int n_all, void *hip_stream);
/* SPH_SWEEP_LIST: the interleaved (pos4, vel4) records of the sorted rows (32 bytes per row, row r at
 * byte 32 r; device pointer, NULL for the other sweeps).  The density sweep leaves rho in the record, so
 * the driver can send a boundary layer's records straight into the neighbour's halo rows (exchange B)
 * instead of sending vel4 rows and patching them in (sph_slab_patch_halo). */
void *sph_slab_records(sph_handle *h);
/* kernelMoveParticles (simulator.cu:329-367) for one slab: the impulse of sph_apply_click on the
 * z-layers [z_lo, z_hi) this slab owns, applied to the NEW state (buffer `buf`: the rows the last
 * sph_slab_force* launch wrote, still in that step's sorted order) through the cell table of that
 * step's sort -- the pre-integration grid, like the reference (simulator.cu:482-489). */
int sph_slab_apply_click(sph_handle *h, int buf, int mouse_x, int mouse_y, int z_lo, int z_hi);
/* rows [a0, b0) and [a1, b1) (either may be empty; a1 >= b0) of the owned range, in ONE
 * launch; hip_stream NULL = the handle's stream. */
int sph_slab_force_ranges(sph_handle *h, int buf, int i_origin, int a0, int b0, int a1, int b1,
                          int n_all, int last_launch_of_the_step, void *hip_stream);

const char *sph_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
