// SPH_SWEEP_LIST: density sweep that RECORDS which candidates are inside the
// support radius, and a force sweep that walks only those.
//
// kernelUpdatePressureAndDensity (simulator.cu:149-190) and kernelUpdateForces
// (simulator.cu:192-256) test exactly the same candidate pairs against the same
// radius, one kernel after the other, and ~85 % of those tests fail.  Here the
// density sweep leaves one bit per candidate (bit b of word w of run r <=> sorted
// index js[r] + 32 w + b) in a per-particle bit stream; the force sweep pops set
// bits in ascending order -- the canonical summation order -- and evaluates the
// pair body for hits only.  It needs no distance tests, no LDS and few
// registers, so it runs at high occupancy, and its wave iterations are the
// maximum hit COUNT over the wave's lanes rather than the maximum candidate
// count.
//
// Bit streams live in a pool (mask words) carved up per wave with one atomicAdd;
// a wave that finds the pool exhausted marks its particles and the force sweep
// falls back to testing every candidate for them (same results, slower).
//
// MUST be compiled with -ffp-contract=off (see sweep_common.h).
#include "sweep_common.h"

#define SL_NONE 0xFFFFFFFFu

// In-kernel phase stamps (diagnostic builds only: -DSW_STAMPS=1), as in sweeps.hip.
#ifndef SW_STAMPS
#define SW_STAMPS 0
#endif
#if SW_STAMPS
__device__ __forceinline__ unsigned long long sl_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define SL_STAMP(var) unsigned long long var = sl_stamp()
#else
#define SL_STAMP(var)
#endif
#ifndef SL_WINDOW
#define SL_WINDOW 160 // records around the wave's own particles cached in LDS by the force sweep
                      // (0 = off).  Measured: 1.60 -> 1.49 ms with 128; 160 or 192 another 1 %
                      // (5 KiB per wave still leaves six waves per SIMD); 256 records or three
                      // rows (12 KiB per wave) lose more to occupancy than they save.
#endif
// Workgroup sizes of the two sweeps.  Their waves are autonomous (no workgroup barrier),
// but a workgroup's LDS and wave slots are only handed back when its LAST wave ends, and
// wave times differ with the hit counts: one wave per workgroup measured -1.3 % (density)
// and -2.7 % (force) against four.
#ifndef SL_K1_THREADS
#define SL_K1_THREADS 64
#endif
#ifndef SL_K2_THREADS
#define SL_K2_THREADS 64
#endif
#ifndef SL_PAIRQ
#define SL_PAIRQ 2 // 0, 2 or 4: the force sweep refills that many pairs at a time (aligned lane streams)
#endif
#ifndef SL_VCONST
#define SL_VCONST 0
#endif
#ifndef SL_WBUF
#define SL_WBUF 8 // mask words buffered per lane before they are stored (2 KiB per wave);
                  // measured: density 1.17 ms with 8, 1.26 ms with 16 (one resident wave fewer)
#endif
#ifndef SL_EXP_LDSONLY
#define SL_EXP_LDSONLY 0
#endif
#ifndef SL_PV8
#define SL_PV8 1 // gather one interleaved 32-B (pos4, vel4) record per hit: measured
                 // force sweep 1.80 -> ~1.55 ms (two loads, ONE cache line per lane)
#endif

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < SPH_WAVE; off <<= 1) {
        uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// ---------------------------------------------------------------------------
// density + hit masks, LDS-staged (production).  Same wave-autonomous walk as
// k_density_lds: the wave stages the union of its lanes' ranges of one run in
// LDS and every lane walks its own range from its first candidate, four per
// trip.  All lanes of the wave are at the same candidate ORDINAL k at any time,
// so the mask bit position (k & 31) is wave-uniform: recording a hit costs a
// compare and an add-with-carry (m = 2m + hit), and whole words are bit-reversed
// and flushed every eighth trip.  A run whose union does not fit the slice (dense cells) is walked
// in the same lock-step straight from global memory.
// ---------------------------------------------------------------------------
#ifndef SL_ADDC
#define SL_ADDC 1
#endif
#ifndef SL_K1_WAVES
#define SL_K1_WAVES 0 // >0: ask for that many resident waves per SIMD (caps the VGPR budget)
#endif
template <bool FAST>
__global__
#if SL_K1_WAVES
__launch_bounds__(SL_K1_THREADS, SL_K1_WAVES)
#else
__launch_bounds__(SL_K1_THREADS)
#endif
void k_density_mask_lds(DevParams P, SweepArgs A) {
    __shared__ float4 stageAll[SL_K1_THREADS / SPH_WAVE][SW_CAP + SW_UNROLL];
    // Finished mask words wait here ([slot][lane]: conflict-free) until a lane has
    // SL_WBUF of them, then leave as 16-byte stores: single-word stores to 64
    // different streams made every word a partial-line write (measured 2.0 GB of
    // WRITE_SIZE per launch for 0.28 GB of masks).
    __shared__ uint32_t wbufAll[SL_K1_THREADS / SPH_WAVE][SL_WBUF * SPH_WAVE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 *stage = stageAll[w];
    uint32_t *wbuf = wbufAll[w] + lane;
    SL_STAMP(t0);
#if SW_STAMPS
    unsigned long long accStage = 0, accTest = 0;
#endif
    const int i = A.i_begin + xcd_tile(blockIdx.x, gridDim.x, A.tileChunk * (256 / SL_K1_THREADS)) * blockDim.x + threadIdx.x;
    const bool valid = i < A.i_end;
    float4 pi = valid ? A.pos4[i] : make_float4(0, 0, 0, 0);
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    int js[9], je[9];
    load_runs(P, A.cellRange, c, valid, js, je);

    // pool space for the worst case: one (first candidate, mask) pair per 32 candidates
    uint32_t words = 0, pairs = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        words += 2u * ((uint32_t)(je[r] - js[r] + 31) >> 5);
        pairs += (uint32_t)(je[r] - js[r]);
    }
#if SL_PAIRQ
    words = (words + 2u * SL_PAIRQ - 1u) & ~(2u * SL_PAIRQ - 1u); // lane streams start on a load boundary
#endif
    const uint32_t incl = wave_incl_scan_u32(words, lane);
    const uint32_t total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(A.maskCursor, (unsigned long long)total);
    base = (unsigned long long)__shfl((unsigned)(base >> 32), 0) << 32 |
           (unsigned long long)__shfl((unsigned)base, 0);
    const bool ok = base + total <= A.maskCapacity; // wave-uniform
    uint32_t woff = ok ? (uint32_t)base + (incl - words) : SL_NONE;
    const uint32_t woff0 = woff;
    if (A.pairCounter) {
        uint32_t s = wave_sum_u32(pairs);
        if (lane == 0) atomicAdd(A.pairCounter, (unsigned long long)s);
    }

    if (lane < SW_UNROLL) stage[SW_CAP + lane] = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    const float4 *const sent = stage + SW_CAP;
    float rho = 0.f;
#if SW_STAMPS
    asm volatile("" ::"v"(js[0] + je[8] + (int)woff));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    SL_STAMP(t1);
#if SL_VCONST
    float h2v = P.h2, dcv = P.dcoef, cut2v = P.cut2;
    asm volatile("" : "+v"(h2v), "+v"(dcv), "+v"(cut2v));
#endif
#if SL_ADDC
    float cut2r = P.cut2; // in a VGPR once: the compare below cannot take an SGPR there
    asm volatile("" : "+v"(cut2r));
#endif
    int pend = 0; // words of this lane waiting in wbuf
    // write this lane's pending words (4 at a time) and empty its buffer
    auto flush_words = [&]() {
#pragma unroll
        for (int q = 0; q < SL_WBUF; q += 4) {
            if (q < pend) {
                uint4 v;
                v.x = wbuf[(q + 0) * SPH_WAVE];
                v.y = wbuf[(q + 1) * SPH_WAVE];
                v.z = wbuf[(q + 2) * SPH_WAVE];
                v.w = wbuf[(q + 3) * SPH_WAVE];
                uint32_t *dst = A.maskPool + woff + q;
                if (q + 4 <= pend) {
                    *reinterpret_cast<uint4 *>(dst) = v; // global_store_dwordx4 (dword aligned)
                } else {
                    dst[0] = v.x;
                    if (q + 1 < pend) dst[1] = v.y;
                    if (q + 2 < pend) dst[2] = v.z;
                }
            }
        }
        woff += (uint32_t)pend;
        pend = 0;
    };
    const int rowId = c.y + c.z * P.D;
    unsigned long long todo = __ballot(valid);
    // Lanes are grouped by grid row only to keep a run's union inside the LDS slice:
    // run ranges are monotone in the lane's cell key, so ANY set of lanes can share a
    // staged range.  When the wave spans several rows but every run's union over all
    // its lanes still fits (sparse fills: a thin sheet puts ~50 rows in one wave), one
    // pass serves them all instead of one pass per row (n = 8192 -i grid: 0.33 -> 0.03 ms).
    bool unify = false;
    if (todo) {
        const int row0 = __builtin_amdgcn_readlane(rowId, __ffsll((long long)todo) - 1);
        if (__ballot(valid && rowId != row0)) {
            unify = true;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const unsigned long long mm = __ballot(valid && je[r] > js[r]);
                if (mm) {
                    const int u0 = __builtin_amdgcn_readlane(js[r], __ffsll((long long)mm) - 1);
                    const int u1 = __builtin_amdgcn_readlane(je[r], 63 - __clzll((long long)mm));
                    unify = unify && (u1 - u0) <= SW_CAP;
                }
            }
        }
    }
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int rowL = __builtin_amdgcn_readlane(rowId, leader);
        const bool act = valid && (unify || rowId == rowL);
        todo &= ~__ballot(act);
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
            int jsr = js[0], jer = je[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) { // r is wave-uniform: scalar-conditioned moves
                jsr = (r == q) ? js[q] : jsr;
                jer = (r == q) ? je[q] : jer;
            }
            const bool nonempty = act && jer > jsr;
            const unsigned long long mm = __ballot(nonempty);
            if (!mm) continue;
            const int lo = __ffsll((long long)mm) - 1;
            const int hi = 63 - __clzll((long long)mm);
            const int u0 = __builtin_amdgcn_readlane(jsr, lo);
            const int u1 = __builtin_amdgcn_readlane(jer, hi);
            const int len = nonempty ? jer - jsr : 0;          // this lane's candidates

            const bool staged = (u1 - u0) <= SW_CAP;           // wave-uniform
            SL_STAMP(tA);
            if (staged) {
                for (int k = lane; k < u1 - u0; k += SPH_WAVE) stage[k] = A.pos4[u0 + k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#if SW_STAMPS
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
            }
            SL_STAMP(tB);
            const float4 *cur = stage + (nonempty ? jsr - u0 : 0);
            const float4 *gcur = A.pos4 + (nonempty ? jsr : 0);
            uint32_t m = 0;
            int k = 0;
            // four candidates of one trip: density terms, hit bits, word hand-over
            auto trip = [&](const float4 (&pj)[SW_UNROLL]) {
#if SL_VCONST
                // VALU ops with an SGPR source issue at half rate on gfx950
                // (scripts/microbench/valu_rate.hip): keep the constants in VGPRs
                const float h2 = h2v, dcoef = dcv, cut2 = cut2v;
#else
                const float h2 = P.h2, dcoef = P.dcoef, cut2 = P.cut2;
#endif
#pragma unroll
                for (int u = 0; u < SW_UNROLL; ++u) {
                    float dx = pi.x - pj[u].x;
                    float dy = pi.y - pj[u].y;
                    float dz = pi.z - pj[u].z;
                    float dist2;
                    if (FAST) {
                        dist2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                        const float diff = fmaxf(h2 - dist2, 0.f);
                        rho = __builtin_fmaf((SPH_MASS * dcoef) * (diff * diff), diff, rho);
                    } else {
                        dist2 = dx * dx + dy * dy + dz * dz;
                        const float diff = fmaxf(h2 - dist2, 0.f);
                        rho += SPH_MASS * (dcoef * diff * diff * diff);
                    }
#if SL_ADDC
                    // hit bit shifted in through the carry: m = 2m + !(dist2 > cut2), two
                    // VALU ops per candidate instead of mov + cmp + cndmask + or.  The word
                    // fills from the top, so it is bit-reversed once when it is handed over.
                    asm("v_cmp_ngt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
                        : "+v"(m)
                        : "v"(dist2), "v"(cut2r)
                        : "vcc");
#else
                    const uint32_t bit = 1u << ((k + u) & 31); // wave-uniform
                    m |= !(dist2 > cut2) ? bit : 0u;
#endif
                }
#pragma unroll
                for (int u = 0; u < SW_UNROLL; ++u) asm volatile("" ::"v"(pj[u].w));
                if (((k + SW_UNROLL) & 31) == 0) { // a whole word is complete (wave-uniform)
                    if (ok && m != 0) { // empty words are not stored
                        wbuf[pend++ * SPH_WAVE] = (uint32_t)(jsr + (k & ~31));
#if SL_ADDC
                        wbuf[pend++ * SPH_WAVE] = __builtin_bitreverse32(m);
#else
                        wbuf[pend++ * SPH_WAVE] = m;
#endif
                    }
                    m = 0;
                    if (__ballot(pend >= SL_WBUF)) flush_words();
                }
            };
            // (measured: a select-free first phase -- trips in which every lane still has
            // four candidates, found with a DPP wave-min -- saves 8 of ~78 VALU ops in
            // about half the trips and 0.5 % of the kernel: not kept)
            // (two copies of the loop rather than one with a select in it: joining
            // the LDS and the global candidates cost 20 register moves per trip)
            if (staged) {
                for (; __ballot(k < len); k += SW_UNROLL) {
                    float4 pj[SW_UNROLL];
#pragma unroll
                    for (int u = 0; u < SW_UNROLL; ++u) {
                        const float4 *p = (k + u < len) ? cur + k : sent;
                        pj[u] = p[u];
                    }
                    trip(pj);
                }
            } else {
                for (; __ballot(k < len); k += SW_UNROLL) {
                    float4 pj[SW_UNROLL];
#pragma unroll
                    for (int u = 0; u < SW_UNROLL; ++u) {
                        const bool in = k + u < len;
                        pj[u] = gcur[in ? k + u : 0];
                        pj[u].x = in ? pj[u].x : 1e18f; // out of range: fails every radius test
                    }
                    trip(pj);
                }
            }
            if ((k & 31) != 0) { // last, partial word
                if (ok && m != 0) {
                    wbuf[pend++ * SPH_WAVE] = (uint32_t)(jsr + (k & ~31));
#if SL_ADDC
                    wbuf[pend++ * SPH_WAVE] = __builtin_bitreverse32(m) >> (32 - (k & 31));
#else
                    wbuf[pend++ * SPH_WAVE] = m;
#endif
                }
                if (__ballot(pend >= SL_WBUF)) flush_words();
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#if SW_STAMPS
            {
                SL_STAMP(tC);
                accStage += tB - tA;
                accTest += tC - tB;
            }
#endif
        }
    }
    SL_STAMP(t2);
    if (ok) flush_words();
    if (valid) { // where this particle's pairs start and how many dwords they take
        A.maskOff[2 * (size_t)i] = woff0;
        A.maskOff[2 * (size_t)i + 1] = ok ? woff - woff0 : 0u;
    }
    if (valid) {
        rho = fmaxf(rho, SPH_EPS_F);
        A.vel4[i].w = rho;
        A.pv8[2 * (size_t)i + 1].w = rho;
    }
#if SW_STAMPS
    {
        SL_STAMP(t3);
        if (lane == 0 && A.stampCounter) {
            unsigned long long *S = A.stampCounter + 16 + (blockIdx.x & 255) * 16;
            atomicAdd(S + 1, t1 - t0);   // prologue (pos, cell, 27 table reads, pool slice)
            atomicAdd(S + 2, accStage);  // staging (global -> LDS) incl. waits
            atomicAdd(S + 3, accTest);   // test loops incl. mask hand-over
            atomicAdd(S + 4, t3 - t0);   // whole wave
            atomicAdd(S + 5, 1ull);      // waves
            atomicAdd(S + 13, t3 - t2);  // final flush + stores
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// force + integrate over the recorded hits
// ---------------------------------------------------------------------------
// Measured dead end: non-temporal (`nt`) loads of the hit stream and stores of it in the
// density sweep, meant to keep the stream from pushing neighbour records out of L2:
// force sweep 1.30 -> 1.66 ms, density 1.12 -> 1.17 ms (an nt load does not keep the
// lane's line for its next 8-byte pair either).
#ifndef SL_K2_WAVES
#define SL_K2_WAVES 0 // >0: ask for that many resident waves per SIMD (caps the VGPR budget)
#endif
template <bool FAST>
__global__
#if SL_K2_WAVES
__launch_bounds__(SL_K2_THREADS, SL_K2_WAVES)
#else
__launch_bounds__(SL_K2_THREADS)
#endif
void k_force_list(DevParams P, SweepArgs A) {
    const int tileIdx = xcd_tile(blockIdx.x, gridDim.x, A.tileChunk * (256 / SL_K2_THREADS));
    const int i = A.i_begin + tileIdx * blockDim.x + threadIdx.x;
    const bool valid = i < A.i_end;
    const int iSafe = valid ? i : A.i_begin;
    float4 pi = A.pos4[iSafe];
    const float4 vi = A.vel4[iSafe];
    const float prs_i = fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY));
    // this particle's stream of (first candidate, 32-bit hit mask) pairs
    const uint32_t off = valid ? A.maskOff[2 * (size_t)i] : 0u;
    const int total = valid ? (int)A.maskOff[2 * (size_t)i + 1] : 0; // dwords
    ForceAcc F = {0.f, 0.f, 0.f};

#if SL_WINDOW
    // LDS copy of the records around the wave's own particles.  The kernel is bound
    // by the texture addresser's gather rate (PMC: TA busy 89 %), and ~40 % of all
    // hits are neighbours in the particle's own grid row, i.e. within a few dozen
    // slots of the wave's 64 particles in the sorted stream: those are served by
    // ds_read_b128 instead of a 64-address global gather.
    __shared__ float4 winAll[SL_K2_THREADS / SPH_WAVE][2 * SL_WINDOW];
    float4 *win = winAll[threadIdx.x >> 6];
    const int tile0 = A.i_begin + tileIdx * blockDim.x + (threadIdx.x & ~63);
    const int w0 = max(tile0 - (SL_WINDOW - SPH_WAVE) / 2, 0);
    const int wlen = max(min(SL_WINDOW, A.n_all - w0), 0);
    {
        const int lane = threadIdx.x & 63;
        for (int k = lane; k < 2 * wlen; k += SPH_WAVE) win[k] = A.pv8[2 * (size_t)w0 + k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#endif

    // A wave that found the mask pool exhausted has no stream: its particles are
    // handled by k_force_fallback (kept out of this kernel: its 27 table reads and
    // run arrays would cost two resident waves per SIMD here).
    if (__ballot(valid && off == SL_NONE)) return;
    {
        const uint2 *stream = reinterpret_cast<const uint2 *>(A.maskPool + off);
        const int npairs = total >> 1;
        // Bit cursor.  (jb, m): first candidate and remaining bits of the current
        // pair; (jbn, mn): the next pair, already in flight.  pop() returns the next
        // hit's sorted index, or the particle itself once the stream is exhausted
        // (dist = 0 gates every term: exact no-op).
#if SL_PAIRQ
        // SL_PAIRQ pairs per refill (16-byte loads): fewer stream loads and cache-line touches
        const uint4 *stream4 = reinterpret_cast<const uint4 *>(A.maskPool + off);
        const int nq = (npairs + SL_PAIRQ - 1) / SL_PAIRQ;
        int wq = 0;
        uint32_t m = 0, mq[SL_PAIRQ];
        int jb = 0, jq[SL_PAIRQ];
#pragma unroll
        for (int u = 0; u < SL_PAIRQ; ++u) { mq[u] = 0; jq[u] = 0; }
        bool live = true;
        auto fetch = [&]() {
            if (wq < nq) {
#pragma unroll
                for (int u = 0; u < SL_PAIRQ; u += 2) {
                    const uint4 t = stream4[wq * (SL_PAIRQ / 2) + u / 2];
                    jq[u] = (int)t.x;
                    mq[u] = (SL_PAIRQ * wq + u < npairs) ? t.y : 0u; // beyond the count: padding
                    jq[u + 1] = (int)t.z;
                    mq[u + 1] = (SL_PAIRQ * wq + u + 1 < npairs) ? t.w : 0u;
                }
                ++wq;
            }
        };
        fetch();
        auto pop = [&]() -> int {
            if (m == 0) { // next pair; stored masks are never 0, so mq[0] == 0 means "queue empty"
                m = mq[0];
                jb = jq[0];
#pragma unroll
                for (int u = 0; u + 1 < SL_PAIRQ; ++u) { mq[u] = mq[u + 1]; jq[u] = jq[u + 1]; }
                mq[SL_PAIRQ - 1] = 0;
                if (mq[0] == 0) fetch();
            }
#else
        int wi = 0;
        uint32_t m = 0, mn = 0;
        int jb = 0, jbn = 0;
        bool live = true;
        if (wi < npairs) { const uint2 t = stream[wi]; jbn = (int)t.x; mn = t.y; ++wi; }
        auto pop = [&]() -> int {
            if (m == 0) { // take the prefetched pair, start fetching the one after
                m = mn;
                jb = jbn;
                mn = 0;
                if (wi < npairs) { const uint2 t = stream[wi]; jbn = (int)t.x; mn = t.y; ++wi; }
            }
#endif
#if SL_PAIRQ
            live = (m | mq[0]) != 0;
#else
            live = (m | mn) != 0;
#endif
            const bool has = m != 0;
            const int b = has ? __builtin_ctz(m) : 0;
            m &= m - 1u; // (0 stays 0)
            return has ? jb + b : iSafe;
        };
        auto body = [&](const float4 &pj, const float4 &vj) {
            if (FAST) force_pair_fast(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, pj, vj, F);
            else force_pair(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, pj, vj, F);
        };
        // Two gathers are always in flight while a pair body is evaluated: the
        // loop is unrolled by two so the pipeline registers never move.  (Three in
        // flight: 99 VGPRs, four resident waves, 1.24 -> 1.44 ms.)
#if SL_EXP_LDSONLY
        // PERF-ONLY experiment (results wrong by construction): every hit is read from the
        // wave's LDS slice at (j mod window) -- what the sweep would cost if all records
        // came from LDS at this LDS footprint.
#define SL_FETCH(j, p, v)
#define SL_USE(j, p, v)                                                        \
    p = win[2 * ((j) & (SL_WINDOW - 1))];                                      \
    v = win[2 * ((j) & (SL_WINDOW - 1)) + 1];                                  \
    body(p, v);
#elif SL_WINDOW
        // fetch: issue the global gather only for lanes whose hit is outside the
        // window (fewer active lanes = fewer addresses for the TA); the LDS copy is
        // read when the hit is consumed.
#define SL_FETCH(j, p, v)                                                      \
    if ((unsigned)((j)-w0) >= (unsigned)wlen) {                                \
        p = A.pv8[2 * (size_t)(j)];                                            \
        v = A.pv8[2 * (size_t)(j) + 1];                                        \
    }
#define SL_USE(j, p, v)                                                        \
    if ((unsigned)((j)-w0) < (unsigned)wlen) {                                 \
        p = win[2 * ((j)-w0)];                                                 \
        v = win[2 * ((j)-w0) + 1];                                             \
    }                                                                          \
    body(p, v);
#else
#define SL_FETCH(j, p, v)                                                      \
    p = A.pv8[2 * (size_t)(j)];                                                \
    v = A.pv8[2 * (size_t)(j) + 1];
#define SL_USE(j, p, v) body(p, v);
#endif
        float4 p0 = make_float4(0, 0, 0, 0), v0 = p0, p1 = p0, v1 = p0;
        int j0 = pop();
        SL_FETCH(j0, p0, v0)
        int j1 = pop();
        SL_FETCH(j1, p1, v1)
        for (;;) {
            SL_USE(j0, p0, v0)
            j0 = pop();
            if (!__ballot(live)) { SL_USE(j1, p1, v1) SL_FETCH(j0, p0, v0) SL_USE(j0, p0, v0) break; }
            SL_FETCH(j0, p0, v0)
            SL_USE(j1, p1, v1)
            j1 = pop();
            if (!__ballot(live)) { SL_USE(j0, p0, v0) SL_FETCH(j1, p1, v1) SL_USE(j1, p1, v1) break; }
            SL_FETCH(j1, p1, v1)
        }
#undef SL_FETCH
#undef SL_USE
    }
    if (valid) {
        float vx = vi.x, vy = vi.y, vz = vi.z;
        integrate_particle(P, pi, vx, vy, vz, F, vi.w);
        store_particle(A, i, pi, vx, vy, vz, vi.w, F);
    }
}

void sph_launch_density_list(const DevParams &P, const SweepArgs &A, int mathMode, hipStream_t s) {
    int cnt = A.i_end - A.i_begin;
    if (cnt <= 0) return;
    int blocks = (cnt + SL_K1_THREADS - 1) / SL_K1_THREADS;
    if (mathMode == 1) k_density_mask_lds<true><<<blocks, SL_K1_THREADS, 0, s>>>(P, A);
    else k_density_mask_lds<false><<<blocks, SL_K1_THREADS, 0, s>>>(P, A);
}

// Particles whose wave found the mask pool exhausted in the density sweep: test
// every candidate, like the check path.  Launched after k_force_list every step;
// waves with nothing to do leave after one load.
template <bool FAST>
__global__ __launch_bounds__(SW_THREADS) void k_force_fallback(DevParams P, SweepArgs A) {
    const int i = A.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = i < A.i_end && A.maskOff[2 * (size_t)i] == SL_NONE;
    if (!__ballot(mine)) return;
    const int iSafe = mine ? i : A.i_begin;
    float4 pi = A.pos4[iSafe];
    const float4 vi = A.vel4[iSafe];
    const float prs_i = fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY));
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    int js[9], je[9];
    load_runs(P, A.cellRange, c, mine, js, je);
    ForceAcc F = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 9; ++r)
        for (int j = js[r]; j < je[r]; ++j) {
            if (FAST) force_pair_fast(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, A.pos4[j], A.vel4[j], F);
            else force_pair(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, A.pos4[j], A.vel4[j], F);
        }
    if (mine) {
        float vx = vi.x, vy = vi.y, vz = vi.z;
        integrate_particle(P, pi, vx, vy, vz, F, vi.w);
        store_particle(A, i, pi, vx, vy, vz, vi.w, F);
    }
}

// Slab path: rho (and velocity) of HALO particles arrive in vel4 through exchange
// B after the density sweep; mirror them into the interleaved records.
__global__ void k_patch_pv8(const float4 *__restrict__ vel4, float4 *__restrict__ pv8, int i_begin,
                            int i_end, int n_all) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int j = t < i_begin ? t : i_end + (t - i_begin);
    if (j < n_all) pv8[2 * (size_t)j + 1] = vel4[j];
}

void sph_launch_force_list(const DevParams &P, const SweepArgs &A, int mathMode, hipStream_t s) {
    int cnt = A.i_end - A.i_begin;
    if (cnt <= 0) return;
    int blocks = (cnt + SW_THREADS - 1) / SW_THREADS;
    const int halo = A.i_begin + (A.n_all - A.i_end);
    if (halo > 0) k_patch_pv8<<<(halo + 255) / 256, 256, 0, s>>>(A.vel4, A.pv8, A.i_begin, A.i_end, A.n_all);
    if (mathMode == 1) {
        k_force_list<true><<<(cnt + SL_K2_THREADS - 1) / SL_K2_THREADS, SL_K2_THREADS, 0, s>>>(P, A);
        k_force_fallback<true><<<blocks, SW_THREADS, 0, s>>>(P, A);
    } else {
        k_force_list<false><<<(cnt + SL_K2_THREADS - 1) / SL_K2_THREADS, SL_K2_THREADS, 0, s>>>(P, A);
        k_force_fallback<false><<<blocks, SW_THREADS, 0, s>>>(P, A);
    }
}
