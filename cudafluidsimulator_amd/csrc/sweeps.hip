// Density/pressure sweep and fused force + integration sweep over the key-sorted
// float4 particle streams -- the gfx950 replacements for
// kernelUpdatePressureAndDensity (simulator.cu:149-190), kernelUpdateForces
// (simulator.cu:192-256) and kernelUpdatePositions (simulator.cu:258-318).
//
// THIS FILE MUST BE COMPILED WITH -ffp-contract=off: in strict mode every fp32
// operation rounds on its own, in the order the reference's expressions are
// written, so results are bit-identical to oracle/sph_oracle.c.
//
// Neighbour walk.  With the flattened key (x fastest) the reference's 27-cell
// walk (dz outer, dy, dx inner; simulator.cu:163-176) is nine contiguous index
// ranges of the sorted stream ("runs": cells x-1..x+1 of row (y+dy, z+dz)),
// visited in ascending order -- which is the canonical summation order.
//
// Production variant (SPH_SWEEP_LDS): every 64-lane wave is autonomous (no
// workgroup barrier).  It owns 64 consecutive sorted particles, groups its
// lanes by grid row, and for each of the nine runs stages the UNION of its
// lanes' ranges into its private LDS slice in chunks of SW_CAP float4
// (coalesced 16-B loads -> ds_write_b128), then each lane walks ITS OWN
// sub-range with ds_read_b128.  Chunking makes the window independent of the
// per-cell particle count (the reference's lists are unbounded too).
//
// Force sweep: only ~15 % of the candidates of a 27-cell box lie inside the
// support radius, and the in-radius body is ~10x the cost of the distance test
// (2 sqrt + 3 IEEE divides).  Running the body under a divergent mask would
// cost the full body on every candidate, so the test loop only PUSHES hits
// into a per-lane FIFO in LDS; whenever some lane's FIFO is full the whole
// wave pops one entry each and runs the body once.  Hits are consumed in push
// order, so every particle still accumulates its neighbours in canonical
// order, and skipped candidates contribute exactly +-0 in the reference, which
// never changes an accumulator that is not -0 (it never is: sums start at +0).
#include "sweep_common.h"

// =====================  DIRECT variant (check path)  =========================
__global__ __launch_bounds__(SW_THREADS) void k_density_direct(DevParams P,
                                                               SweepArgs A) {
    int i = A.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = i < A.i_end;
    float4 pi = valid ? A.pos4[i] : make_float4(0, 0, 0, 0);
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    float rho = 0.f;
    uint32_t pairs = 0;
    if (P.morton) {
        // Morton keys: the three x-neighbours are not adjacent in the sorted stream, so the
        // walk is the reference's 27 cells one by one (simulator.cu:163-176: z outer, x inner)
        for (int dz = -1; dz < 2; ++dz)
            for (int dy = -1; dy < 2; ++dy)
                for (int dx = -1; dx < 2; ++dx) {
                    const int x = c.x + dx, y = c.y + dy, z = c.z + dz;
                    if (!valid || x < 0 || x >= P.D || y < 0 || y >= P.D || z < 0 || z >= P.D) continue;
                    const int2 r = A.cellRange[sph_cell_key(P, x, y, z)];
                    pairs += (uint32_t)(r.y - r.x);
                    for (int j = r.x; j < r.y; ++j) density_pair(P, pi.x, pi.y, pi.z, A.pos4[j], rho);
                }
    } else {
        int js[9], je[9];
        load_runs(P, A.cellRange, c, valid, js, je);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            pairs += (uint32_t)(je[r] - js[r]);
            for (int j = js[r]; j < je[r]; ++j) density_pair(P, pi.x, pi.y, pi.z, A.pos4[j], rho);
        }
    }
    if (A.pairCounter) {
        uint32_t s = wave_sum_u32(pairs);
        if ((threadIdx.x & 63) == 0) atomicAdd(A.pairCounter, (unsigned long long)s);
    }
    if (valid) {
        rho = fmaxf(rho, SPH_EPS_F);
        A.vel4[i].w = rho;
    }
}

__global__ __launch_bounds__(SW_THREADS) void k_force_direct(DevParams P, SweepArgs A) {
    int i = A.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.i_end) return;
    float4 pi = A.pos4[i];
    float4 vi = A.vel4[i];
    float prs_i = fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY));
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    ForceAcc F = {0.f, 0.f, 0.f};
    if (P.morton) {
        for (int dz = -1; dz < 2; ++dz)
            for (int dy = -1; dy < 2; ++dy)
                for (int dx = -1; dx < 2; ++dx) {
                    const int x = c.x + dx, y = c.y + dy, z = c.z + dz;
                    if (x < 0 || x >= P.D || y < 0 || y >= P.D || z < 0 || z >= P.D) continue;
                    const int2 r = A.cellRange[sph_cell_key(P, x, y, z)];
                    for (int j = r.x; j < r.y; ++j)
                        force_pair<false>(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, A.pos4[j], A.vel4[j], F);
                }
    } else {
        int js[9], je[9];
        load_runs(P, A.cellRange, c, true, js, je);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            for (int j = js[r]; j < je[r]; ++j)
                force_pair<false>(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, A.pos4[j], A.vel4[j],
                           F);
        }
    }
    float vx = vi.x, vy = vi.y, vz = vi.z;
    integrate_particle(P, pi, vx, vy, vz, F, vi.w);
    store_particle(A, i, pi, vx, vy, vz, vi.w, F);
}

// =====================  LDS variant (production)  ============================
// Walks the nine runs of the lanes of one wave.  The inner loop is branch-free
// and unrolled by SW_UNROLL: each trip issues SW_UNROLL ds_read_b128 up front and
// calls V.candidate(j, pj) for every lane; a lane that has run out of candidates
// reads the SENTINEL slot (a point 1e18 away: it fails every radius test), so no
// per-lane exec masking is needed.  V.poll() is called wave-uniformly once per
// trip.
// In-kernel phase stamps (diagnostic builds only: -DSW_STAMPS=1).  Sums of
// s_memtime deltas per phase go to counters 1.. of A.pairCounter; the shipped
// build executes none of this.
#ifndef SW_STAMPS
#define SW_STAMPS 0
#endif
#if SW_STAMPS
__device__ __forceinline__ unsigned long long sw_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define SW_STAMP(var) unsigned long long var = sw_stamp()
#else
#define SW_STAMP(var)
#endif
struct WalkStamps {
    unsigned long long stage = 0, test = 0;
};

#ifndef SW_PIPELINE
#define SW_PIPELINE 1
#endif
// Negative results kept out of the code (measured on MI355X, n = 4,194,304):
//  * streaming the next run with global_load_lds (LDS-DMA) while the current one
//    is tested: density 1.29 ms vs 0.98 ms -- hipcc drains lgkmcnt(0) on every LDS
//    read while a global_load_lds is outstanding;
//  * staging three runs per round trip (3 sub-buffers): staging stamps fell from
//    32k to 20k cycles per wave but the extra registers cost a wave per SIMD and
//    the sweeps got slower (density 1.02 ms, force 2.62 ms vs 0.98 / 2.40);
//  * a per-CELL table of the nine runs (72 MB, [run][cell]) read with 9 loads per
//    particle instead of 27 gathers from the 8 MB cell table: slower (density
//    1.11 -> 1.13 ms, force 1.57 -> 1.67 ms) -- the small table stays in L2, the
//    big one does not.
#define SW_SENTINEL SW_CAP // index of the far-away point inside the stage slice

template <class Visitor>
__device__ __forceinline__ void wave_walk(const SweepArgs &A, float4 *__restrict__ stage,
                                          int lane, bool valid, int rowId,
                                          const int (&js)[9], const int (&je)[9],
                                          Visitor &V, WalkStamps &W) {
    (void)W;
    if (lane < SW_UNROLL) stage[SW_SENTINEL + lane] = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int rowL = __builtin_amdgcn_readlane(rowId, leader);
        const bool act = valid && rowId == rowL;
        todo &= ~__ballot(act);
        // one copy of the loop body (not 9): r is wave-uniform, so picking this
        // run's bounds is 16 scalar-conditioned moves, and the instruction
        // footprint stays well inside the I-cache
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
            int jsr = js[0], jer = je[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) {
                jsr = (r == q) ? js[q] : jsr;
                jer = (r == q) ? je[q] : jer;
            }
            const bool nonempty = act && jer > jsr;
            const unsigned long long m = __ballot(nonempty);
            if (!m) continue;
            // lanes are key-sorted, so run bounds are monotone over the lanes
            const int lo = __ffsll((long long)m) - 1;
            const int hi = 63 - __clzll((long long)m);
            const int u0 = __builtin_amdgcn_readlane(jsr, lo);
            const int u1 = __builtin_amdgcn_readlane(jer, hi);
            for (int cs = u0; cs < u1; cs += SW_CAP) {
                SW_STAMP(tA);
                const int len = min(SW_CAP, u1 - cs);
                for (int k = lane; k < len; k += SPH_WAVE) stage[k] = A.pos4[cs + k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#if SW_STAMPS
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
                SW_STAMP(tB);
                const int a = nonempty ? max(jsr, cs) - cs : 0;      // first local slot
                const int b = nonempty ? min(jer, cs + len) - cs : 0; // one past last
                // per-lane cursor: LDS slot, candidates left, global index
                const float4 *cur = stage + a;
                const float4 *const sent = stage + SW_SENTINEL;
                int rem = max(b - a, 0);
                int jcur = cs + a;
#if SW_PIPELINE
                // Software-pipelined by one trip: the ds_read_b128 of trip t+1 are
                // in flight while trip t is evaluated, so the wave does not park on
                // LDS latency once per trip (two register sets, A and B).
                float4 pa[SW_UNROLL], pb[SW_UNROLL];
#define SW_ISSUE(dst, remv, curv)                                              \
    _Pragma("unroll") for (int u = 0; u < SW_UNROLL; ++u) {                    \
        const float4 *p_ = ((remv) > u) ? (curv) : sent;                       \
        dst[u] = p_[u];                                                        \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0); /* keep the reads ahead of the math */
#define SW_CONSUME(src)                                                        \
    _Pragma("unroll") for (int u = 0; u < SW_UNROLL; ++u)                      \
        V.candidate(jcur + u, src[u]);                                         \
    _Pragma("unroll") for (int u = 0; u < SW_UNROLL; ++u)                      \
        asm volatile("" ::"v"(src[u].w));                                      \
    V.poll();                                                                  \
    rem -= SW_UNROLL;                                                          \
    cur += SW_UNROLL;                                                          \
    jcur += SW_UNROLL;
                SW_ISSUE(pa, rem, cur)
                while (__ballot(rem > 0)) {
                    SW_ISSUE(pb, rem - SW_UNROLL, cur + SW_UNROLL)
                    SW_CONSUME(pa)
                    if (!__ballot(rem > 0)) break;
                    SW_ISSUE(pa, rem - SW_UNROLL, cur + SW_UNROLL)
                    SW_CONSUME(pb)
                }
#undef SW_ISSUE
#undef SW_CONSUME
#else
                for (; __ballot(rem > 0); rem -= SW_UNROLL, cur += SW_UNROLL, jcur += SW_UNROLL) {
                    float4 pj[SW_UNROLL];
#pragma unroll
                    for (int u = 0; u < SW_UNROLL; ++u) {
                        // exhausted lanes read sentinel slot u (the +u folds into the
                        // ds_read offset field either way)
                        const float4 *p = (rem > u) ? cur : sent;
                        pj[u] = p[u];
                    }
#pragma unroll
                    for (int u = 0; u < SW_UNROLL; ++u) V.candidate(jcur + u, pj[u]);
                    // keep .w live so each read is one ds_read_b128 (4 LDS cycles per
                    // wave) instead of the ds_read_b96 (8 cycles) hipcc would pick
#pragma unroll
                    for (int u = 0; u < SW_UNROLL; ++u) asm volatile("" ::"v"(pj[u].w));
                    V.poll();
                }
#endif
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#if SW_STAMPS
                SW_STAMP(tC);
                W.stage += tB - tA;
                W.test += tC - tB;
#endif
            }
        }
    }
}

template <bool FAST>
struct DensityVisitor {
    const DevParams &P;
    float pix, piy, piz;
    float rho;
    // densityKernel (simulator.cu:84-97) + `density += MASS * W` (:179), branch-free:
    // diff = max(h2 - dist2, 0) makes W exactly +0 outside the radius, and adding
    // +0 never changes rho (rho >= +0), so this equals the reference's early return.
    __device__ __forceinline__ void candidate(int, float4 pj) {
        if (FAST) {
            float dx, dy, dz;
            const float dist2 = fast_dist2(pix, piy, piz, pj, dx, dy, dz);
            const float diff = fmaxf(P.h2 - dist2, 0.f);
            rho = __builtin_fmaf((SPH_MASS * P.dcoef) * (diff * diff), diff, rho);
            return;
        }
        float dx = pix - pj.x;
        float dy = piy - pj.y;
        float dz = piz - pj.z;
        float dist2 = dx * dx + dy * dy + dz * dz;
        float diff = fmaxf(P.h2 - dist2, 0.f);
        rho += SPH_MASS * (P.dcoef * diff * diff * diff);
    }
    __device__ __forceinline__ void poll() {}
};

template <bool FAST>
__global__ __launch_bounds__(SW_THREADS) void k_density_lds(DevParams P, SweepArgs A) {
    __shared__ float4 stageAll[SW_WAVES][SW_CAP + SW_UNROLL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 *stage = stageAll[w];
    SW_STAMP(t0);
    const int i = A.i_begin + xcd_tile(blockIdx.x, gridDim.x, A.tileChunk, A.tileRotate) * blockDim.x + threadIdx.x;
    const bool valid = i < A.i_end;
    float4 pi = valid ? A.pos4[i] : make_float4(0, 0, 0, 0);
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    int js[9], je[9];
    load_runs(P, A.cellRange, c, valid, js, je);
    if (A.pairCounter) {
        uint32_t pairs = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) pairs += (uint32_t)(je[r] - js[r]);
        uint32_t s = wave_sum_u32(pairs);
        if (lane == 0) atomicAdd(A.pairCounter, (unsigned long long)s);
    }
    DensityVisitor<FAST> V{P, pi.x, pi.y, pi.z, 0.f};
    WalkStamps W;
#if SW_STAMPS
    asm volatile("" ::"v"(js[0] + je[8]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    SW_STAMP(t1);
    wave_walk(A, stage, lane, valid, c.y + c.z * P.D, js, je, V, W);
    if (valid) {
        float rho = fmaxf(V.rho, SPH_EPS_F);
        A.vel4[i].w = rho;
    }
#if SW_STAMPS
    SW_STAMP(t2);
    if (lane == 0 && A.pairCounter) { // 256 shards: same-address atomics would distort
        unsigned long long *S = A.pairCounter + 16 + (blockIdx.x & 255) * 16;
        atomicAdd(S + 1, t1 - t0);  // prologue
        atomicAdd(S + 2, W.stage);  // staging (global -> LDS) incl. waits
        atomicAdd(S + 3, W.test);   // test loops
        atomicAdd(S + 4, t2 - t0);  // whole wave
        atomicAdd(S + 5, 1ull);     // waves
    }
#endif
}

#ifndef SW_DRAIN
#define SW_DRAIN 4 // FIFO entries evaluated per drain (loads of all issued first)
#endif
#ifndef SW_PUSH_BRANCHFREE
#define SW_PUSH_BRANCHFREE 1 // measured: force sweep 2.47 ms vs 2.55 ms (n = 4,194,304)
#endif

template <bool FAST, bool SLIM>
struct ForceVisitor {
    const DevParams &P;
    const SweepArgs &A;
    uint32_t *queue; // this wave's FIFO storage: [SW_QCAP][64]
    int lane;
    uint32_t selfIdx; // a candidate that contributes exactly nothing (dist = 0)
    float pix, piy, piz, vix, viy, viz, prs_i;
    uint32_t head, tail;
    ForceAcc F;
    unsigned long long drains = 0;
    uint32_t items = 0; // diagnostic: FIFO entries really evaluated by this lane

    __device__ __forceinline__ void candidate(int j, float4 pj) {
        float dx = pix - pj.x;
        float dy = piy - pj.y;
        float dz = piz - pj.z;
        float dist2 = FAST ? __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx))
                           : dx * dx + dy * dy + dz * dz;
#if SW_PUSH_BRANCHFREE
        // the slot at `tail` is always free here (poll() keeps SW_UNROLL slots
        // spare), so store unconditionally and only advance on a hit
        queue[(tail & (SW_QCAP - 1)) * SPH_WAVE + lane] = (uint32_t)j;
        tail += !(dist2 > P.cut2) ? 1u : 0u;
#else
        if (!(dist2 > P.cut2)) {
            queue[(tail & (SW_QCAP - 1)) * SPH_WAVE + lane] = (uint32_t)j;
            ++tail;
        }
#endif
    }
    // Pop up to SW_DRAIN hits per lane, issue all their loads, then evaluate them
    // in FIFO order.  An empty slot is replaced by the particle itself, whose
    // pair terms are gated off by dist < EPS_F -- an exact no-op.
    __device__ __forceinline__ void drain() {
        const uint32_t have = tail - head;
#if SW_STAMPS
        ++drains;
        items += min(have, (uint32_t)SW_DRAIN);
#endif
        uint32_t j[SW_DRAIN];
        float4 pj[SW_DRAIN], vj[SW_DRAIN];
#pragma unroll
        for (int u = 0; u < SW_DRAIN; ++u) {
            uint32_t q = queue[((head + u) & (SW_QCAP - 1)) * SPH_WAVE + lane];
            j[u] = ((uint32_t)u < have) ? q : selfIdx;
        }
#pragma unroll
        for (int u = 0; u < SW_DRAIN; ++u) {
            pj[u] = A.pos4[j[u]];
            vj[u] = A.vel4[j[u]];
        }
        head += min(have, (uint32_t)SW_DRAIN);
#pragma unroll
        for (int u = 0; u < SW_DRAIN; ++u) {
            if (FAST) force_pair_fast(P, pix, piy, piz, vix, viy, viz, prs_i, pj[u], vj[u], F);
            else force_pair<SLIM>(P, pix, piy, piz, vix, viy, viz, prs_i, pj[u], vj[u], F);
        }
    }
    __device__ __forceinline__ void poll() {
        // the next trip pushes at most SW_UNROLL entries per lane
        while (__ballot((tail - head) > (uint32_t)(SW_QCAP - SW_UNROLL))) drain();
    }
    __device__ __forceinline__ void flush() {
        while (__ballot(tail != head)) drain();
    }
};

template <bool FAST, bool SLIM>
__global__ __launch_bounds__(SW_THREADS) void k_force_lds(DevParams P, SweepArgs A) {
    __shared__ float4 stageAll[SW_WAVES][SW_CAP + SW_UNROLL];
    __shared__ uint32_t queueAll[SW_WAVES][SW_QCAP * SPH_WAVE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    SW_STAMP(t0);
    const int i = A.i_begin + xcd_tile(blockIdx.x, gridDim.x, A.tileChunk, A.tileRotate) * blockDim.x + threadIdx.x;
    const bool valid = i < A.i_end;
    const int iSafe = valid ? i : A.i_begin; // i_end > i_begin whenever we are launched
    float4 pi = A.pos4[iSafe];
    float4 vi = A.vel4[iSafe];
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    int js[9], je[9];
    load_runs(P, A.cellRange, c, valid, js, je);
    ForceVisitor<FAST, SLIM> V{P, A, queueAll[w], lane, (uint32_t)iSafe, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z,
                   fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY)),
                   0u, 0u, {0.f, 0.f, 0.f}};
    WalkStamps W;
#if SW_STAMPS
    asm volatile("" ::"v"(js[0] + je[8]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    SW_STAMP(t1);
    wave_walk(A, stageAll[w], lane, valid, c.y + c.z * P.D, js, je, V, W);
    SW_STAMP(t2);
    V.flush();
    SW_STAMP(t3);
    if (valid) {
        float vx = vi.x, vy = vi.y, vz = vi.z;
        integrate_particle(P, pi, vx, vy, vz, V.F, vi.w);
        store_particle(A, i, pi, vx, vy, vz, vi.w, V.F);
    }
#if SW_STAMPS
    SW_STAMP(t4);
    if (lane == 0 && A.stampCounter) {
        unsigned long long *S = A.stampCounter + 16 + (blockIdx.x & 255) * 16;
        atomicAdd(S + 6, t1 - t0);   // prologue
        atomicAdd(S + 7, W.stage);   // staging
        atomicAdd(S + 8, W.test);    // test loops incl. drains they trigger
        atomicAdd(S + 9, t3 - t2);   // final flush
        atomicAdd(S + 10, t4 - t0);  // whole wave
        atomicAdd(S + 11, V.drains); // drain() calls
    }
    if (A.stampCounter) {
        uint32_t it = wave_sum_u32(V.items);
        if (lane == 0) atomicAdd(A.stampCounter + 16 + (blockIdx.x & 255) * 16 + 12, (unsigned long long)it);
    }
#endif
}

void sph_launch_density(const DevParams &P, const SweepArgs &A, int mathMode, int sweep,
                        hipStream_t s) {
    int cnt = A.i_end - A.i_begin;
    if (cnt <= 0) return;
    int blocks = (cnt + SW_THREADS - 1) / SW_THREADS;
    if (sweep == 0)
        sph_launch_density_list(P, A, mathMode, s);
    else if (sweep == 3)
        sph_launch_density_linked(P, A, s); // strict only (reference structure)
    else if (sweep == 1)
        k_density_direct<<<blocks, SW_THREADS, 0, s>>>(P, A); // strict only (check path)
    else if (mathMode == 1)
        k_density_lds<true><<<blocks, SW_THREADS, 0, s>>>(P, A);
    else
        k_density_lds<false><<<blocks, SW_THREADS, 0, s>>>(P, A);
}

void sph_launch_force(const DevParams &P, const SweepArgs &A, int mathMode, int sweep,
                      hipStream_t s) {
    if (sweep == 0) { // (may carry a second row range: sizes its own launch)
        sph_launch_force_list(P, A, mathMode, s);
        return;
    }
    int cnt = A.i_end - A.i_begin;
    if (cnt <= 0) return;
    int blocks = (cnt + SW_THREADS - 1) / SW_THREADS;
    if (sweep == 3)
        sph_launch_force_linked(P, A, s);
    else if (sweep == 1)
        k_force_direct<<<blocks, SW_THREADS, 0, s>>>(P, A); // strict only (check path)
    else if (mathMode == 1)
        k_force_lds<true, true><<<blocks, SW_THREADS, 0, s>>>(P, A);
    else if (P.slimDiv)
        k_force_lds<false, true><<<blocks, SW_THREADS, 0, s>>>(P, A);
    else
        k_force_lds<false, false><<<blocks, SW_THREADS, 0, s>>>(P, A);
}
