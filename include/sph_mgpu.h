/*
 * sph_mgpu.h -- C-ABI of the in-process multi-GPU driver (libsph_mgpu.so): the SPH
 * step of include/sph_c_api.h cut into z-slabs of whole cell layers, one slab per
 * MI355X, halo layers exchanged with RCCL send/recv over xGMI.
 *
 * No reference counterpart: the reference is single-GPU (simulator.cu:462-546 is the
 * step being distributed).  Host code is C++ (csrc/mgpu.cpp): one host
 * thread drives every local slab; the data path never goes through Python.  Two ways
 * to run it:
 *   - one process drives all GPUs of the node (`./sph` with SPH_GPUS=N):
 *     rank_begin = 0, rank_count = world, ncclCommInitAll;
 *   - one process per GPU (bench.py under torch.distributed.run): rank_count = 1,
 *     ncclCommInitRank with a unique id made by sph_mgpu_unique_id() on rank 0 and
 *     handed to the other ranks by the launcher.
 * SPH_TRANSPORT_LOOPBACK steps `world` slabs on ONE device with device-to-device copies
 * in place of RCCL (tests and the one-GPU box); SPH_TRANSPORT_RCCL_SELF does the same
 * through a one-rank RCCL communicator sending to itself (exercises the RCCL calls).
 *
 * Per step and slab (buffers `s` and `t` swap roles):
 *   1. stable partition of the owned rows by the z-range their NEW cell falls in
 *      [far down | down | lower boundary layer | interior | upper boundary layer | up | far up]
 *   2. exchange A: header (the partition bounds) + a FIXED number of rows per face
 *      (migrants + boundary layer), no count exchange first
 *   3. the ONE host synchronisation of the step: own bounds + the neighbours' headers
 *   4. assemble [halo | migrants | mine | migrants | halo], stable sort by cell key,
 *      cell table, density sweep of the owned rows
 *   5. exchange B (rho of the boundary layers) on a second stream WHILE the force sweep
 *      runs over the interior layers; then the two boundary layers
 * N slabs reproduce the single-domain result bit for bit (tests/test_mgpu.py).
 * Limit: a particle may cross into a neighbour slab and land anywhere short of that
 * slab's far boundary layer in one step; beyond that the step after reports SPH_ESTATE.
 */
#ifndef SPH_MGPU_H
#define SPH_MGPU_H

#include "sph_c_api.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_MGPU_MAX_LOCAL 8

enum {
    SPH_TRANSPORT_LOOPBACK = 0,  /* device-to-device copies; every slab on devices[0] */
    SPH_TRANSPORT_RCCL = 1,      /* ncclSend/ncclRecv between the slabs' GPUs */
    SPH_TRANSPORT_RCCL_SELF = 2, /* one-rank communicator, every message sent to itself */
    SPH_TRANSPORT_MAILBOX = 3,   /* TEST HOOK for the one-process-per-GPU code path on a one-GPU box:
                                    `world` driver objects of one slab each live in ONE process on one
                                    device and stand for the ranks; a message is a note in a process-wide
                                    table.  Every object must be stepped phase by phase
                                    (sph_mgpu_step_phase 1..4 on all ranks before the next phase). */
    SPH_TRANSPORT_STREAMS = 4    /* the RCCL transport's stream layout on ONE device: every slab has its
                                    own compute, exchange and boundary streams, exchange B overlaps the
                                    interior force sweep, the boundary layers run on a stream of their
                                    own -- and a message is a device-to-device copy ordered by events the
                                    way a grouped ncclSend/ncclRecv orders it (receiver after the
                                    sender's stream reached the send, sender held until the copy is done).
                                    Exercises everything of the RCCL path but RCCL itself on a one-GPU box. */
};

typedef struct SphMgpuOptions {
    int32_t struct_size;   /* = sizeof(SphMgpuOptions) */
    int32_t world;         /* z-slabs of the domain (= GPUs) */
    int32_t rank_begin;    /* first slab this process drives */
    int32_t rank_count;    /* how many (world, or 1 with one process per GPU) */
    int32_t transport;     /* SPH_TRANSPORT_* */
    int32_t devices[SPH_MGPU_MAX_LOCAL]; /* HIP device of local slab k */
    int32_t sweep;         /* SPH_SWEEP_* */
    int32_t math_mode;     /* SPH_MATH_* */
    int32_t face_capacity; /* rows per fixed-size exchange message; 0 = 1.25 x fullest layer */
    int32_t slab_capacity; /* rows per slab buffer; 0 = 1.6 x largest slab */
    int32_t recut_every;   /* re-balance the z cuts from the layer histogram every K steps
                              (0 = never; single-process runs only) */
} SphMgpuOptions;

typedef struct SphMgpuStats {
    int64_t steps;
    int64_t host_syncs;       /* blocking host<->device synchronisations inside sph_mgpu_step */
    int64_t overflow_rounds;  /* steps in which a face outgrew its fixed-size message */
    int64_t recuts;
    int32_t local_slabs;
    int32_t owned[SPH_MGPU_MAX_LOCAL];     /* particles per local slab now */
    double kernel_s[SPH_MGPU_MAX_LOCAL];   /* GPU time of the slab's kernels: the sum of the next three */
    double grid_s[SPH_MGPU_MAX_LOCAL];     /*   partition + combined sort + gather / cell table */
    double density_s[SPH_MGPU_MAX_LOCAL];  /*   kernelUpdatePressureAndDensity (computeDensity) */
    double force_s[SPH_MGPU_MAX_LOCAL];    /*   force + integrate, all ranges */
} SphMgpuStats;

typedef struct sph_mgpu sph_mgpu;

/* 128 opaque bytes (ncclUniqueId) for the one-process-per-GPU mode. */
int sph_mgpu_unique_id(void *out128);
/* unique_id128 may be NULL unless transport == RCCL and rank_count < world. */
int sph_mgpu_create(const SphSettings *settings, const SphMgpuOptions *options,
                    const void *unique_id128, sph_mgpu **out);
void sph_mgpu_destroy(sph_mgpu *m);

/* Simulator::setup (simulator.cu:411-460): the reference initial condition, cut into
 * slabs by the z-layer histogram.  upload_state: caller state, particle-id order (every
 * process passes the whole state and keeps its slabs' particles). */
int sph_mgpu_setup(sph_mgpu *m);
int sph_mgpu_upload_state(sph_mgpu *m, const float *pos_xyz, const float *vel_xyz, int n);
/* Simulator::simulate / simulateAndTime (simulator.cu:462-546) over all local slabs. */
int sph_mgpu_step(sph_mgpu *m, SphTimes *times);
/* `mouseClicked` of Simulator::simulate (simulator.cu:482-489): the NEXT step that completes
 * applies kernelMoveParticles (simulator.cu:329-367) after its force sweep, every slab on the
 * z-layers it owns, through that step's grid.  One process per GPU: every rank queues the same
 * click.  Queue it between steps. */
int sph_mgpu_queue_click(sph_mgpu *m, int mouse_x, int mouse_y);
/* The four phases of a step on their own (1: partition + exchange A, 2: headers + host sync,
 * 3: assemble/sort/density + exchange B, 4: force + read-back), in this order. */
int sph_mgpu_step_phase(sph_mgpu *m, int phase, SphTimes *times);
/* Simulator::getPosition: numParticles x (x,y,z), particle-id order; rows of particles
 * owned by other processes keep their last known value (one process per GPU). */
const float *sph_mgpu_positions_host(sph_mgpu *m);
/* particle-id order; only rows of locally owned particles are written.  Returns the
 * number of rows written through *written (may be NULL). */
int sph_mgpu_download_state(sph_mgpu *m, float *pos_xyz, float *vel_xyz, float *rho, int *written);
int sph_mgpu_sync(sph_mgpu *m);
int sph_mgpu_get_stats(sph_mgpu *m, SphMgpuStats *out, int reset);
const char *sph_mgpu_last_error(const sph_mgpu *m); /* m may be NULL: create errors */

#ifdef __cplusplus
}
#endif
#endif
