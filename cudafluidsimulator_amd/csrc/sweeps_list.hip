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
// SL_POOL_SHARDS / SL_CURSOR_STRIDE live in sph_device.h (the host sizes the cursor array)

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
#define SL_K2_THREADS 64 // k_force_list (the one-wave-per-workgroup sweep; production is k_force_dealt below)
#endif
#ifndef SL_EXP_LDSONLY
#define SL_EXP_LDSONLY 0
#endif
#ifndef SL_TRIVIAL_TEST
#define SL_TRIVIAL_TEST 0
#endif
#ifndef SL_FETCH_UNIFORM
#define SL_FETCH_UNIFORM 0
#endif
#ifndef SL_LDSDMA
#define SL_LDSDMA 0
#endif
#ifndef SL_PV8
#define SL_PV8 1 // gather one interleaved 32-B (pos4, vel4) record per hit: measured
                 // force sweep 1.80 -> ~1.55 ms (two loads, ONE cache line per lane)
#endif

// ---------------------------------------------------------------------------
// density + hit masks, LDS-staged (production).  Same wave-autonomous walk as
// k_density_lds: the wave stages the union of its lanes' ranges of one run in
// LDS and every lane walks its own range from its first candidate, four per
// trip.  All lanes of the wave are at the same candidate ORDINAL k at any time,
// so the mask bit position (k & 31) is wave-uniform: recording a hit costs a
// compare and an add-with-carry (m = 2m + hit), and a whole word is bit-reversed
// and handed over every eighth trip.  A run whose union does not fit the slice
// (dense cells) is walked in the same lock-step straight from global memory.
//
// Hit stream layout ("wave-transposed"): the wave owns Q quads per lane,
// quad q of lane l at maskPool[(base + q*64 + l) * 4 .. +4) = {j0, m0, j1, m1}:
// two (first candidate, 32-bit hit mask) pairs.  A lane writes only its
// NON-EMPTY words, compacted, and ends its sequence with a zero mask (unless
// it fills all 2Q pairs), so the force sweep reads pairs until it meets m == 0.
// Lanes complete quads at about the same time, so a quad store is (mostly) one
// contiguous kilobyte per wave and every cache line is filled by neighbouring
// lanes within a few hundred cycles -- round 1's per-lane contiguous streams
// cost 0.93 GB of partial-line WRITE_SIZE per launch for 0.45 GB of pairs
// and an LDS staging buffer per wave (profiles/r01_pmc_hbm_v10.csv).
// Per wave header: maskOff[2w] = base in quads (or SL_NONE: pool exhausted ->
// k_force_fallback), maskOff[2w+1] = Q.
// ---------------------------------------------------------------------------
#ifndef SL_ADDC
#define SL_ADDC 1
#endif
#ifndef SL_K1_WAVES
#define SL_K1_WAVES 6 // resident waves per SIMD asked for (caps the VGPR budget at 80); measured: 5..8 the same
#endif
// ---- zero-pair filter ("quiet" rows) ----
// kernelUpdateForces adds, for a neighbour j of i, a pressure term proportional to
// (p_i + p_j) and a viscosity term proportional to (v_j - v_i) (simulator.cu:223-251).
// When neither row is under pressure and both move with the same velocity, both terms are
// exactly +-0 and the accumulators (never -0) do not change: the pair can be dropped without
// reading j at all.  That is every pair of a body of fluid in free fall -- all of the
// reference's `-i random` run until the cloud reaches the floor, and the part of it still
// falling afterwards.  The density sweep leaves one bit per sorted row: "no pressure, and the
// velocity equals the reference velocity" (any reference is correct; the first sort pass picks the
// most common velocity among 64 sampled rows and the gather launch compares every row with it --
// the last sorted row, the top of the highest z-layer, was the first choice and went wrong at step
// 57 of the headline run, when splashes from the floor opened a z-layer of their own); the force
// sweep of a quiet row clears the quiet candidates out of its hit masks, 32 at a time.
// one 64-bit word per 64-row wave (rows [i - lane, i - lane + 64), i - lane a multiple of 64); the gather
// launch left the velocity half of the test ("calm": the row moves with the reference velocity) as a word of
// the same shape, so the sweep adds the pressure half without reading a velocity
__device__ __forceinline__ void sl_store_quiet(const SweepArgs &A, int i, bool valid, float rho, int lane) {
    const unsigned long long calm = A.calm[(i - lane) >> 6]; // (wave-uniform address)
    const float prs = fmaxf(0.f, SPH_GAS_CONSTANT * (rho - SPH_REST_DENSITY));
    const bool q = valid && prs == 0.f && ((calm >> lane) & 1ull);
    const unsigned long long qb = __ballot(q), vb = __ballot(valid);
    if (lane == 0) {
        reinterpret_cast<unsigned long long *>(A.quiet)[(i - lane) >> 6] = qb;
        if (A.quietAll && qb != vb) *A.quietAll = 0u; // (plain store: every writer writes the same value)
    }
}
// quiet bits of rows [j, j + 32)
__device__ __forceinline__ uint32_t sl_quiet_window(const uint32_t *__restrict__ quiet, uint32_t j) {
    const uint32_t lo = quiet[j >> 5], hi = quiet[(j >> 5) + 1];
    return __builtin_amdgcn_alignbit(hi, lo, j & 31u);
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// SAMECUT: the force cut-off P.cut2 (largest dist2 with sqrtf(dist2) <= h) equals h*h --
// true for the reference's h = 0.1f.  The hit bit is then the sign of h2 - dist2, which the
// density term needs anyway, and one v_alignbit_b32 shifts it in (miss = 1; the word is
// complemented when it is handed over) instead of a compare and an add-with-carry.
template <bool FAST, bool SAMECUT>
__global__
#if SL_K1_WAVES
__launch_bounds__(SL_K1_THREADS, SL_K1_WAVES * SL_K1_THREADS / 64)
#else
__launch_bounds__(SL_K1_THREADS)
#endif
void k_density_mask_lds(DevParams P, SweepArgs A) {
    __shared__ float4 stageAll[SL_K1_THREADS / SPH_WAVE][SW_CAP + SW_UNROLL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 *stage = stageAll[w];
    SL_STAMP(t0);
#if SW_STAMPS
    unsigned long long accStage = 0, accTest = 0;
#endif
    const int wv = xcd_tile(blockIdx.x, gridDim.x, A.tileChunk * (256 / SL_K1_THREADS), A.tileRotate) * (SL_K1_THREADS / SPH_WAVE) + w;
    // waves are numbered from i_origin (<= i_begin, a multiple of 64 in slab mode so that a wave's
    // 64 rows are one word of the zero-pair filter's bit array); rows below i_begin are not this launch's
    const int i = A.i_origin + wv * SPH_WAVE + lane;
    const bool valid = i >= A.i_begin && i < A.i_end;
    const bool anyValid = __ballot(valid) != 0ull;
    float4 pi = valid ? A.pos4[i] : make_float4(0, 0, 0, 0);
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    int js[9], je[9];
    load_runs(P, A.cellRange, c, valid, js, je);

    // pool space: a lane stores at most one pair per 32 candidates of each run
    int words = 0;
    uint32_t pairs = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        words += (je[r] - js[r] + 31) >> 5;
        pairs += (uint32_t)(je[r] - js[r]);
    }
    const int Q = __builtin_amdgcn_readfirstlane((wave_max_i32(words) + 1) >> 1); // quads per lane
    // The pool is cut into SL_POOL_SHARDS equal sub-pools, each with its own cursor on a
    // cache line of its own: 65,536 waves per launch bumping ONE address is a rate limit
    // of its own (a returning atomic on one word saturates at ~88 per microsecond on this
    // chip: 0.74 ms per launch, which is what bounded this kernel's early steps in round 1).
    const int shard = wv & (SL_POOL_SHARDS - 1);
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(A.maskCursor + shard * SL_CURSOR_STRIDE, (unsigned long long)Q * SPH_WAVE);
    base = (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32 |
           (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)base);
    const unsigned long long subCap = A.maskCapacity / SL_POOL_SHARDS;    // quads per sub-pool
    const bool ok = base + (unsigned long long)Q * SPH_WAVE <= subCap;    // wave-uniform
    base += (unsigned long long)shard * subCap;
    if (lane == 0 && anyValid) {
        A.maskOff[2 * (size_t)wv] = ok ? (uint32_t)base : SL_NONE;
        A.maskOff[2 * (size_t)wv + 1] = (uint32_t)Q;
        if (!ok) A.noneList[atomicAdd(A.maskCursor + 1, 1ull)] = (uint32_t)wv; // (for k_force_fallback)
    }
    uint4 *const myq = reinterpret_cast<uint4 *>(A.maskPool) + (ok ? base : 0ull) + lane;
    if (A.pairCounter) { // sharded like the stamps: one address would serialise the waves
        uint32_t s = wave_sum_u32(pairs);
        if (lane == 0) atomicAdd(A.pairCounter + 16 + (wv & 255) * 16, (unsigned long long)s);
    }

    if (lane < SW_UNROLL) stage[SW_CAP + lane] = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    const float4 *const sent = stage + SW_CAP;
    float rho = 0.f;
#if SW_STAMPS
    asm volatile("" ::"v"(js[0] + je[8]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    SL_STAMP(t1);
    // VALU ops with an SGPR or literal source issue at half rate on gfx950
    // (scripts/microbench/valu_rate.hip): the loop constants live in VGPRs
    float h2v = P.h2, dcv = P.dcoef, cut2r = P.cut2, massv = SPH_MASS;
    asm volatile("" : "+v"(h2v), "+v"(dcv), "+v"(cut2r), "+v"(massv));
    // this lane's pending pair and the number of quads it has stored
    uint32_t pj0 = 0, pm0 = 0;
    bool pending = false;
    int qidx = 0;
    uint32_t hcount = 0; // hits recorded for this row (the force sweep deals rows to lanes by it)
    // hand a finished word over: non-empty words only, two per 16-byte store
    auto emit = [&](uint32_t jbase, uint32_t mask) {
        if (ok && mask != 0) {
            hcount += (uint32_t)__builtin_popcount(mask);
            if (!pending) {
                pj0 = jbase;
                pm0 = mask;
                pending = true;
            } else {
#if defined(SL_DBG_SAMEADDR) // (perf-only experiments: what the hit stream's stores cost the density sweep)
                myq[0] = make_uint4(pj0, pm0, jbase, mask);
#elif !defined(SL_DBG_NOEMIT) // (measured, round 3: a non-temporal store here: density 1.44 vs 1.35 ms at steps 81..100)
                myq[(size_t)qidx * SPH_WAVE] = make_uint4(pj0, pm0, jbase, mask);
#endif
                ++qidx;
                pending = false;
            }
        }
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
    // One run of one pass: this lane's range, the wave's union, staged or not.
    struct Run {
        int jsr, len, u0, ulen;
        bool any, staged;
    };
    auto describe = [&](int r, bool act) -> Run {
        int jsr = js[0], jer = je[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) { // r is wave-uniform: scalar-conditioned moves
            jsr = (r == q) ? js[q] : jsr;
            jer = (r == q) ? je[q] : jer;
        }
        Run R;
        const bool nonempty = act && jer > jsr;
        const unsigned long long mm = __ballot(nonempty);
        R.any = mm != 0;
        const int lo = mm ? __ffsll((long long)mm) - 1 : 0;
        const int hi = mm ? 63 - __clzll((long long)mm) : 0;
        R.u0 = __builtin_amdgcn_readlane(jsr, lo);
        R.ulen = R.any ? __builtin_amdgcn_readlane(jer, hi) - R.u0 : 0;
        R.staged = R.ulen <= SW_CAP; // wave-uniform
        R.jsr = nonempty ? jsr : R.u0;
        R.len = nonempty ? jer - jsr : 0; // this lane's candidates
        return R;
    };
    // (measured, round 2: fetching the union of run r+1 into registers while run r is
    // tested changes nothing -- 1.138 vs 1.139 ms -- and neither does the resident wave
    // count between 5 and 8 per SIMD: the sweep is bound by VALU issue at the ~3.1
    // cycles per non-FMA wave-instruction that scripts/microbench/valu_rate.hip measures,
    // not by the latency of the nine staging round trips.)
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int rowL = __builtin_amdgcn_readlane(rowId, leader);
        const bool act = valid && (unify || rowId == rowL);
        todo &= ~__ballot(act);
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
            const Run R = describe(r, act);
            SL_STAMP(tA);
            if (R.any && R.staged) { // the union of the wave's ranges of this run -> LDS slice
                for (int k = lane; k < R.ulen; k += SPH_WAVE) stage[k] = A.pos4[R.u0 + k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#if SW_STAMPS
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
            }
            SL_STAMP(tB);
            if (R.any) {
                const int jsr = R.jsr, len = R.len;
                const float4 *cur = stage + (jsr - R.u0);
                const float4 *gcur = A.pos4 + jsr;
                uint32_t m = 0;
                int k = 0;
                // four candidates of one trip: density terms, hit bits, word hand-over
                auto trip = [&](const float4 (&pj)[SW_UNROLL]) {
#pragma unroll
                    for (int u = 0; u < SW_UNROLL; ++u) {
                        float dx = pi.x - pj[u].x;
                        float dy = pi.y - pj[u].y;
                        float dz = pi.z - pj[u].z;
                        float dist2;
                        float draw;
                        if (FAST) {
                            dist2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                            draw = h2v - dist2;
                            const float diff = fmaxf(draw, 0.f);
                            rho = __builtin_fmaf((massv * dcv) * (diff * diff), diff, rho);
                        } else {
                            dist2 = dx * dx + dy * dy + dz * dz;
                            draw = h2v - dist2;
                            // (fmaxf() of a value that is also used as bits costs a second,
                            // canonicalising v_max: the difference of two finite floats needs none)
                            float diff;
                            asm("v_max_f32_e32 %0, 0, %1" : "=v"(diff) : "v"(draw));
                            rho += massv * (dcv * diff * diff * diff);
                        }
                        if (SAMECUT) { // m = 2m + (dist2 > h2): one op
                            m = __builtin_amdgcn_alignbit(m, __float_as_uint(draw), 31);
                            continue;
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
                        m |= !(dist2 > cut2r) ? bit : 0u;
#endif
                    }
#pragma unroll
                    for (int u = 0; u < SW_UNROLL; ++u) asm volatile("" ::"v"(pj[u].w));
                    if (((k + SW_UNROLL) & 31) == 0) { // a whole word is complete (wave-uniform)
                        if (SAMECUT) emit((uint32_t)(jsr + (k & ~31)), ~__builtin_bitreverse32(m));
                        else
#if SL_ADDC
                            emit((uint32_t)(jsr + (k & ~31)), __builtin_bitreverse32(m));
#else
                            emit((uint32_t)(jsr + (k & ~31)), m);
#endif
                        m = 0;
                    }
                };
                // (measured: a select-free first phase -- trips in which every lane still has
                // four candidates, found with a DPP wave-min -- saves 8 of ~78 VALU ops in
                // about half the trips and 0.5 % of the kernel: not kept)
                // (two copies of the loop rather than one with a select in it: joining
                // the LDS and the global candidates cost 20 register moves per trip)
                // (measured, round 2: staging a union longer than the slice chunk by chunk --
                // chunks aligned to absolute multiples of 32, every lane walking every chunk --
                // instead of the global-memory walk below: bit-identical, but the extra trips
                // (union vs own range) cost what the gathers cost: density 0.786 -> 0.884 ms over
                // the 100 steps, 1.72 -> 1.69 at steps 81..100.  A larger slice helps instead:
                // SW_CAP 256 / 384 / 512 / 640 / 1024: 0.786 / 0.718 / 0.750 / 0.827 / 1.128.)
                // (measured, round 3: a select-free form of the trips in which no lane is in its LAST, partial
                // trip -- 60 % of the trips at step 1, 92 % at step 100; one pointer select per lane instead of a
                // compare and a select per candidate, 6 of 73 VALU instructions -- is bit-identical and SLOWER:
                // density 0.753 vs 0.70 ms over the 100 steps; the extra ballot and branch per trip cost more
                // than the selects.  The same trips as a first phase of FIXED trip count per run -- no vote either, a
                // lane without a range parked on the sentinel: 80 VGPRs + 24 bytes of scratch at six waves, 0.688-0.706 vs
                // 0.704-0.709; at five waves 0.711-0.718.  Round 2 measured that form at +0.5 % too: the loop is not
                // bound by the instructions these forms remove)
                if (R.staged) {
                    for (; __ballot(k < len); k += SW_UNROLL) {
                        float4 pj[SW_UNROLL];
#pragma unroll
                        for (int u = 0; u < SW_UNROLL; ++u) {
                            const float4 *p = (k + u < len) ? cur + k : sent;
#if defined(SL_DBG_ONE_READ) // (perf-only: one LDS read per trip instead of four -- is the loop bound by LDS traffic?)
                            pj[u] = u == 0 ? p[0] : make_float4(pj[0].x + (float)u, pj[0].y, pj[0].z, 0.f);
#else
                            pj[u] = p[u];
#endif
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
                    if (SAMECUT) emit((uint32_t)(jsr + (k & ~31)), (~__builtin_bitreverse32(m)) >> (32 - (k & 31)));
                    else
#if SL_ADDC
                        emit((uint32_t)(jsr + (k & ~31)), __builtin_bitreverse32(m) >> (32 - (k & 31)));
#else
                        emit((uint32_t)(jsr + (k & ~31)), m);
#endif
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
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
    if (ok && valid) { // end of this lane's sequence: a zero mask, unless all 2Q pairs are used
        if (pending) myq[(size_t)qidx * SPH_WAVE] = make_uint4(pj0, pm0, 0u, 0u);
        else if (qidx < Q) myq[(size_t)qidx * SPH_WAVE] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (valid) {
        rho = fmaxf(rho, SPH_EPS_F);
        // the force sweep of this variant reads every record (its own too) from pv8; the
        // separate velocity stream only needs rho when halo layers are exchanged (slabs)
        if (A.rhoToVel4) A.vel4[i].w = rho;
        A.pv8[2 * (size_t)i + 1].w = rho;
    }
    if (A.quiet) sl_store_quiet(A, i, valid, rho, lane);
    if (A.hitCount && valid) A.hitCount[i] = hcount;
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
            atomicAdd(S + 13, t3 - t2);  // final stores
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// force + integrate over the recorded hits, one wave per workgroup, rows in place
// (-DSL_DEAL=0: the A/B partner of k_force_dealt, and the home of the round-2/3 experiments)
// ---------------------------------------------------------------------------
// Measured dead end: non-temporal (`nt`) loads of the hit stream and stores of it in the
// density sweep, meant to keep the stream from pushing neighbour records out of L2:
// force sweep 1.30 -> 1.66 ms, density 1.12 -> 1.17 ms (an nt load does not keep the
// lane's line for its next 8-byte pair either).
#ifndef SL_K2_WAVES
#define SL_K2_WAVES 0 // >0: ask for that many resident waves per SIMD (caps the VGPR budget)
#endif
template <bool FAST, bool SLIM>
__global__
#if SL_K2_WAVES
__launch_bounds__(SL_K2_THREADS, SL_K2_WAVES)
#else
__launch_bounds__(SL_K2_THREADS)
#endif
void k_force_list(DevParams P, SweepArgs A) {
    // wave w of the hit stream owns particles [i_origin + 64 w, +64); this launch covers
    // the waves that intersect [i_begin, i_end) (a sub-range when the slab driver runs the
    // interior while the halo densities are still in flight)
    const bool second = (int)blockIdx.x >= A.nblk1; // wave-uniform: blocks past nblk1 serve range 2
    const int rb = second ? A.i_begin2 : A.i_begin, re = second ? A.i_end2 : A.i_end;
    const int tileIdx = ((rb - A.i_origin) >> 6) / (SL_K2_THREADS / SPH_WAVE) +
                        xcd_tile(second ? (int)blockIdx.x - A.nblk1 : (int)blockIdx.x,
                                 second ? (int)gridDim.x - A.nblk1 : A.nblk1,
                                 A.tileChunk * (256 / SL_K2_THREADS), A.tileRotate);
    const int i = A.i_origin + tileIdx * blockDim.x + threadIdx.x;
    const bool valid = i >= rb && i < re;
    const int iSafe = valid ? i : rb;
    float4 pi = A.pv8[2 * (size_t)iSafe];
    const float4 vi = A.pv8[2 * (size_t)iSafe + 1];
    const float prs_i = fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY));
    // this wave's hit stream (layout: k_density_mask_lds): Q quads per lane, quad q of
    // this lane at stream4[q * 64] = two (first candidate, 32-bit hit mask) pairs
    const int wv = tileIdx * (SL_K2_THREADS / SPH_WAVE) + (threadIdx.x >> 6);
    const uint32_t baseq = __builtin_amdgcn_readfirstlane(A.maskOff[2 * (size_t)wv]);
    const int Q = __builtin_amdgcn_readfirstlane((int)A.maskOff[2 * (size_t)wv + 1]);
    ForceAcc F = {0.f, 0.f, 0.f};

#if SL_WINDOW
    // LDS copy of the records around the wave's own particles.  The kernel is bound
    // by the texture addresser's gather rate (PMC: TA busy 89 %), and ~40 % of all
    // hits are neighbours in the particle's own grid row, i.e. within a few dozen
    // slots of the wave's 64 particles in the sorted stream: those are served by
    // ds_read_b128 instead of a 64-address global gather.
    __shared__ float4 winAll[SL_K2_THREADS / SPH_WAVE][2 * SL_WINDOW];
    float4 *win = winAll[threadIdx.x >> 6];
#if SL_LDSDMA
    __shared__ float4 ringAll[SL_K2_THREADS / SPH_WAVE][2][2][SPH_WAVE];
    float4(*ring)[2][SPH_WAVE] = ringAll[threadIdx.x >> 6];
#endif
    const int tile0 = A.i_origin + tileIdx * blockDim.x + (threadIdx.x & ~63);
    const int w0 = max(tile0 - (SL_WINDOW - SPH_WAVE) / 2, 0);
    const int wlen = max(min(SL_WINDOW, A.n_all - w0), 0);
#endif

    // A wave that found the mask pool exhausted has no stream: its particles are
    // handled by k_force_fallback (kept out of this kernel: its 27 table reads and
    // run arrays would cost two resident waves per SIMD here).
    if (baseq == SL_NONE) return;
    // Every row of the domain quiet (fluid in free fall): no pair adds anything -- the hit stream is
    // not even read, the sweep is the integration alone.
    const bool allQuiet = A.quietAll && __builtin_amdgcn_readfirstlane(*A.quietAll) != 0u &&
                          (!A.quietHalo || __builtin_amdgcn_readfirstlane(*A.quietHalo) != 0u);
    if (!allQuiet) {
        // Bit cursor.  (jb, m): first candidate and remaining bits of the current
        // pair; (jq, mq): the pairs of the last quad loaded.  A lane's sequence ends
        // with a zero mask (or after Q quads).  pop() returns the next hit's sorted
        // index, or the particle itself once the stream is exhausted (dist = 0 gates
        // every term: exact no-op).
        // (uniform base + one 32-bit per-lane quad index: a per-lane 64-bit pointer, a quad counter and a
        // per-lane end cost three more VGPRs, and the 73rd costs the seventh resident wave)
        const uint4 *const sbase = reinterpret_cast<const uint4 *>(A.maskPool) + baseq;
        const uint32_t send = (uint32_t)Q * SPH_WAVE;                           // uniform: end of the wave's quads
        uint32_t sidx = valid ? (threadIdx.x & 63u) : send;                      // this lane's next quad
        uint32_t m = 0, mq[2] = {0u, 0u};
        int jb = 0, jq[2] = {0, 0};
        bool live = true;
        // zero-pair filter: a quiet row drops its quiet candidates (see sl_is_quiet)
        const bool qi = A.quiet && valid && ((A.quiet[i >> 5] >> (i & 31)) & 1u);
        // post-condition: mq[0] != 0, or the lane's sequence is exhausted
        auto fetch = [&]() {
            while (sidx < send) {
                uint4 t = sbase[sidx];
                sidx = (t.w == 0u) ? send : sidx + SPH_WAVE; // a zero mask ends the sequence
                if (qi) {
                    t.y &= ~sl_quiet_window(A.quiet, t.x);
                    t.w &= ~sl_quiet_window(A.quiet, t.z);
                }
                if (t.y == 0u) { // first pair empty (filtered, or the terminator): the second moves up
                    t.x = t.z;
                    t.y = t.w;
                    t.w = 0u;
                }
                jq[0] = (int)t.x;
                mq[0] = t.y;
                jq[1] = (int)t.z;
                mq[1] = t.w;
                if (t.y != 0u) break;
            }
        };
        fetch();
        // A wave whose lanes have nothing left after the filter (fluid in free fall) skips the sweep:
        // no window, no gathers, straight to the integration.
        if (__ballot(mq[0] != 0u)) {
#if SL_WINDOW
        {
            const int lane = threadIdx.x & 63;
            for (int k = lane; k < 2 * wlen; k += SPH_WAVE) win[k] = A.pv8[2 * (size_t)w0 + k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
#endif
        auto pop = [&]() -> int {
            if (m == 0) { // next pair; queued masks are never 0, so mq[0] == 0 means "queue empty"
                m = mq[0];
                jb = jq[0];
                mq[0] = mq[1];
                jq[0] = jq[1];
                mq[1] = 0;
                if (mq[0] == 0) fetch();
            }
            live = (m | mq[0]) != 0;
            const bool has = m != 0;
            const int b = has ? __builtin_ctz(m) : 0;
            m &= m - 1u; // (0 stays 0)
            return has ? jb + b : iSafe;
        };
        auto body = [&](const float4 &pj, const float4 &vj) {
            if (FAST) force_pair_fast(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, pj, vj, F);
            else force_pair<SLIM>(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, pj, vj, F);
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
#elif SL_TRIVIAL_TEST
        // experiment: a pair with no pressure on either side and no relative velocity adds
        // exactly +-0 to the force (fPressure = -0, dv = +0): fetch the velocity half first and
        // fetch the position half / run the body only for the other pairs (here: synchronously)
#define SL_FETCH(j, p, v)                                                      \
    if ((unsigned)((j)-w0) >= (unsigned)wlen) v = A.pv8[2 * (size_t)(j) + 1];
#define SL_USE(j, p, v)                                                        \
    {                                                                          \
        const bool inw = (unsigned)((j)-w0) < (unsigned)wlen;                  \
        if (inw) v = win[2 * ((j)-w0) + 1];                                    \
        const float prs_j = fmaxf(0.f, SPH_GAS_CONSTANT * (v.w - SPH_REST_DENSITY)); \
        const bool triv = (prs_i + prs_j == 0.f) && v.x == vi.x && v.y == vi.y && v.z == vi.z; \
        if (!triv) {                                                           \
            p = inw ? win[2 * ((j)-w0)] : A.pv8[2 * (size_t)(j)];              \
            body(p, v);                                                        \
        }                                                                      \
    }
#elif SL_WINDOW && SL_LDSDMA
        // experiment: the gathers go through the LDS-DMA path -- per-lane
        // `global_load_lds_dwordx4` into a two-slot ring in LDS (2 x 2 KB per wave), consumed by
        // ds_read_b128 like the window's records: no VGPRs hold records in flight.
#define SL_SLOT_p0 0
#define SL_SLOT_p1 1
#define SL_LDS_PTR(x) ((__attribute__((address_space(3))) void *)(x))
#define SL_FETCH(j, p, v)                                                      \
    if ((unsigned)((j)-w0) >= (unsigned)wlen) {                                \
        __builtin_amdgcn_global_load_lds((const void *)(A.pv8 + 2 * (size_t)(j)), SL_LDS_PTR(&ring[SL_SLOT_##p][0][0]), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((const void *)(A.pv8 + 2 * (size_t)(j) + 1), SL_LDS_PTR(&ring[SL_SLOT_##p][1][0]), 16, 0, 0); \
    }
#define SL_USE(j, p, v)                                                        \
    {                                                                          \
        const bool inw_ = (unsigned)((j)-w0) < (unsigned)wlen;                 \
        const float4 *a_ = inw_ ? &win[2 * ((j)-w0)] : &ring[SL_SLOT_##p][0][threadIdx.x & 63]; \
        const float4 *b_ = inw_ ? a_ + 1 : a_ + SPH_WAVE;                      \
        p = *a_;                                                               \
        v = *b_;                                                               \
        body(p, v);                                                            \
    }
#elif SL_WINDOW && SL_FETCH_UNIFORM
        // experiment: every lane issues both loads of every fetch, so the gathers are straight-line
        // code and the compiler can wait with exact vmcnt counts (two gathers really in flight);
        // a lane whose hit is inside the LDS window loads the window's FIRST record instead (one
        // wave-uniform address: one line for all such lanes, not a lane-request each).
#define SL_FETCH(j, p, v)                                                      \
    {                                                                          \
        const size_t a_ = ((unsigned)((j)-w0) < (unsigned)wlen) ? (size_t)w0 : (size_t)(j); \
        p = A.pv8[2 * a_];                                                     \
        v = A.pv8[2 * a_ + 1];                                                 \
    }
#define SL_USE(j, p, v)                                                        \
    if ((unsigned)((j)-w0) < (unsigned)wlen) {                                 \
        p = win[2 * ((j)-w0)];                                                 \
        v = win[2 * ((j)-w0) + 1];                                             \
    }                                                                          \
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
        }
#undef SL_FETCH
#undef SL_USE
    }
    if (valid) {
        // (measured, round 3: re-reading the id and the density here instead of keeping them alive saves
        // two VGPRs -- 68 -- and nothing else; forcing 64 VGPRs for an eighth wave spills 8: 1.14 vs 0.83 ms)
        float vx = vi.x, vy = vi.y, vz = vi.z;
        integrate_particle(P, pi, vx, vy, vz, F, vi.w);
        store_particle(A, i, pi, vx, vy, vz, vi.w, F);
    }
}

// ---------------------------------------------------------------------------
// force + integrate over the recorded hits, rows DEALT to lanes (production)
// ---------------------------------------------------------------------------
// A wave's trips are its longest lane's hit count; among 64 consecutive rows the counts differ by a
// factor of two (floor pile next to falling cloud, cell to cell).  A workgroup of four waves therefore
// takes SL_DEAL_ROWS consecutive rows and deals them to lanes SORTED by the hit counts the density sweep
// recorded (hitCount[], 4 B per row; an LDS counting sort over 128 buckets), so that the 64 lanes of a
// wave run out of hits together.  Every lane still walks its OWN row's stream in canonical order
// (per-lane stream base and end): same bits.  Lane efficiency on oracle states
// (scripts/studies/sorted_lanes.py): 0.75-0.83 -> 0.89-0.93, 10-17 % fewer trips.
//  * Quiet rows (zero-pair filter) sort first whatever they recorded: the filter leaves them only their
//    pairs with not-quiet neighbours, and quiet / busy rows come in patches -- a wave that mixes them runs
//    as long as its busy rows (n = 16,777,216: 4.61 -> 3.89 ms).  A group whose rows are ALL quiet is not
//    dealt at all: a wave that reads the streams of its own 64 rows reads whole lines.
//  * Measured and not kept: groups of 128 / 512 rows (0.792 / 0.805 vs 0.777); 512-row groups whose wave w
//    sweeps sorted virtual waves w and 7 - w one after the other, a light and a heavy one, so that the four
//    waves end together (0.92); buckets of 2 .. 64 hits (the same).
//  * One LDS window per workgroup: the records of its own rows +- 48 (own grid row: ~40 % of all hits),
//    staged only if one of those rows is not quiet.
// Measured, n = 4,194,304 -i random, 100 steps: force sweep 0.812 (one wave per workgroup, rows in place)
// -> 0.777 ms per step with 256-row groups (profiles/r03_experiments.md has every A/B).
#ifndef SL_DEAL
#define SL_DEAL 1
#endif
#ifndef SL_SORT_SHIFT
#define SL_SORT_SHIFT 3 // 128 buckets of (1 << SL_SORT_SHIFT) hits (1 .. 6 measured the same)
#endif
#define SL_DEAL_THREADS 256
#define SL_DEAL_ROWS SL_DEAL_THREADS
#define SL_DEAL_WINDOW (SL_DEAL_ROWS + SL_WINDOW - SPH_WAVE)
static_assert(SL_WINDOW >= SPH_WAVE, "the dealt sweep needs the window");

template <bool FAST, bool SLIM>
__global__ __launch_bounds__(SL_DEAL_THREADS) void k_force_dealt(DevParams P, SweepArgs A) {
    __shared__ uint32_t sortBase[128];
    __shared__ uint16_t rowOf[SL_DEAL_ROWS];
    __shared__ float4 win[2 * SL_DEAL_WINDOW];
    const int t = threadIdx.x, lane = t & 63;
    // this launch covers the groups that hold rows of [i_begin, i_end) and, past block nblk1, of [i_begin2,
    // i_end2) (the slab driver runs the interior while the halo densities are still in flight); groups are
    // numbered from i_origin like the density sweep's waves
    const bool second = (int)blockIdx.x >= A.nblk1;
    const int rb = second ? A.i_begin2 : A.i_begin, re = second ? A.i_end2 : A.i_end;
    const int chunk = A.tileChunk > 0 ? max(A.tileChunk * 256 / SL_DEAL_ROWS, 1) : 0;
    const int group = ((rb - A.i_origin) >> 6) / (SL_DEAL_ROWS / SPH_WAVE) +
                      xcd_tile(second ? (int)blockIdx.x - A.nblk1 : (int)blockIdx.x,
                               second ? (int)gridDim.x - A.nblk1 : A.nblk1, chunk, A.tileRotate);
    const int R0 = A.i_origin + group * SL_DEAL_ROWS;
    const unsigned long long *const quiet64 = reinterpret_cast<const unsigned long long *>(A.quiet);

    // ---- the deal ----
    unsigned long long qw = 0ull; // quiet bits of this wave's 64 rows in place
    bool inPlace = false;
    if (A.quiet) {
        const int r0 = R0 + (t & ~63);
        const int lo = max(rb - r0, 0), hi = min(re - r0, 64); // the rows of this launch among them
        const unsigned long long vm = hi > lo ? ((hi - lo == 64 ? ~0ull : (1ull << (hi - lo)) - 1ull) << lo) : 0ull;
        if (vm) qw = quiet64[r0 >> 6];
        inPlace = __syncthreads_and((qw & vm) == vm) != 0;
    }
    if (!inPlace) {
        for (int b = t; b < 128; b += SL_DEAL_THREADS) sortBase[b] = 0u;
        __syncthreads();
        const int row = R0 + t;
        const bool counted = row >= rb && row < re && !((qw >> lane) & 1ull);
        const uint32_t bucket = min((counted ? A.hitCount[row] : 0u) >> SL_SORT_SHIFT, 127u);
        const uint32_t slot = atomicAdd(&sortBase[bucket], 1u);
        __syncthreads();
        if (t < SPH_WAVE) { // exclusive scan of the bucket counts, two per lane of the first wave
            const uint32_t a = sortBase[2 * t], b = sortBase[2 * t + 1];
            uint32_t incl = a + b;
#pragma unroll
            for (int off = 1; off < SPH_WAVE; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                incl += t >= off ? up : 0u;
            }
            const uint32_t excl = incl - (a + b);
            sortBase[2 * t] = excl;
            sortBase[2 * t + 1] = excl + a;
        }
        __syncthreads();
        rowOf[sortBase[bucket] + slot] = (uint16_t)t;
    }

    // ---- the window ----
    // Every row of the domain quiet (fluid in free fall): no pair adds anything -- the hit stream is not even
    // read, the sweep is the integration alone.
    const bool allQuiet = A.quietAll && __builtin_amdgcn_readfirstlane(*A.quietAll) != 0u &&
                          (!A.quietHalo || __builtin_amdgcn_readfirstlane(*A.quietHalo) != 0u);
    const int w0 = max(R0 - (SL_WINDOW - SPH_WAVE) / 2, 0);
    int wlen = max(min(SL_DEAL_WINDOW, A.n_all - w0), 0);
    {
        // a quiet row's pairs with quiet rows are dropped: a window of quiet rows serves nobody
        bool wanted = !allQuiet && wlen > 0;
        if (wanted && A.quiet) {
            const int k0 = w0 >> 6, k1 = (w0 + wlen - 1) >> 6; // (whole words: some rows beyond either end count too)
            wanted = k0 + t <= k1 && quiet64[k0 + t] != ~0ull;
        }
        if (__syncthreads_or(wanted)) { // (also orders the deal's rowOf[] writes before their reads)
            for (int k = t; k < 2 * wlen; k += SL_DEAL_THREADS) win[k] = A.pv8[2 * (size_t)w0 + k];
            __syncthreads();
        } else {
            wlen = 0;
        }
    }

    // ---- the sweep of this lane's row ----
    {
        const int i = R0 + (inPlace ? t : (int)rowOf[t]);
        const bool inRange = i >= rb && i < re;
        const int iSafe = inRange ? i : rb;
        float4 pi = A.pv8[2 * (size_t)iSafe];
        const float4 vi = A.pv8[2 * (size_t)iSafe + 1];
        const float prs_i = fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY));
        // the row's hit stream (layout: k_density_mask_lds): wave v of the density sweep owns rows [i_origin + 64 v,
        // +64) and Q quads per lane; quad q of its lane l at (base + 64 q + l) = two (first candidate, mask) pairs
        const int rel = iSafe - A.i_origin;
        const uint32_t baseq = inRange ? A.maskOff[2 * (size_t)(rel >> 6)] : SL_NONE;
        const int Q = inRange ? (int)A.maskOff[2 * (size_t)(rel >> 6) + 1] : 0;
        // (a row whose density wave found the pool exhausted has no stream: k_force_fallback integrates it)
        const bool valid = inRange && baseq != SL_NONE;
        ForceAcc F = {0.f, 0.f, 0.f};
        if (!allQuiet) {
            // Bit cursor.  (jb, m): first candidate and remaining bits of the current pair; (jq, mq): the pairs of
            // the last quad loaded.  A lane's sequence ends with a zero mask (or after Q quads).  pop() returns the
            // next hit's sorted index, or the particle itself once the stream is exhausted (dist = 0 gates every
            // term: exact no-op).
            const uint4 *const sbase = reinterpret_cast<const uint4 *>(A.maskPool);
            const uint32_t send = valid ? baseq + (uint32_t)Q * SPH_WAVE : 0u; // end of the row's quads
            uint32_t sidx = valid ? baseq + (uint32_t)(rel & 63) : 0u;         // this lane's next quad
            uint32_t m = 0, mq[2] = {0u, 0u};
            int jb = 0, jq[2] = {0, 0};
            bool live = true;
            // zero-pair filter: a quiet row drops its quiet candidates
            const bool qi = A.quiet && valid && ((A.quiet[i >> 5] >> (i & 31)) & 1u);
            // post-condition: mq[0] != 0, or the lane's sequence is exhausted
            auto fetch = [&]() {
                while (sidx < send) {
                    uint4 q = sbase[sidx];
                    sidx = (q.w == 0u) ? send : sidx + SPH_WAVE; // a zero mask ends the sequence
                    if (qi) {
                        q.y &= ~sl_quiet_window(A.quiet, q.x);
                        q.w &= ~sl_quiet_window(A.quiet, q.z);
                    }
                    if (q.y == 0u) { // first pair empty (filtered, or the terminator): the second moves up
                        q.x = q.z;
                        q.y = q.w;
                        q.w = 0u;
                    }
                    jq[0] = (int)q.x;
                    mq[0] = q.y;
                    jq[1] = (int)q.z;
                    mq[1] = q.w;
                    if (q.y != 0u) break;
                }
            };
            fetch();
            // A wave whose lanes have nothing left after the filter skips the sweep: no gathers, straight to
            // the integration.
            if (__ballot(mq[0] != 0u)) {
                auto pop = [&]() -> int {
                    if (m == 0) { // next pair; queued masks are never 0, so mq[0] == 0 means "queue empty"
                        m = mq[0];
                        jb = jq[0];
                        mq[0] = mq[1];
                        jq[0] = jq[1];
                        mq[1] = 0;
                        if (mq[0] == 0) fetch();
                    }
                    live = (m | mq[0]) != 0;
                    const bool has = m != 0;
                    const int b = has ? __builtin_ctz(m) : 0;
                    m &= m - 1u; // (0 stays 0)
                    return has ? jb + b : iSafe;
                };
                auto body = [&](const float4 &pj, const float4 &vj) {
                    if (FAST) force_pair_fast(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, pj, vj, F);
                    else force_pair<SLIM>(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, pj, vj, F);
                };
                // Two gathers are always in flight while a pair body is evaluated; the loop is unrolled by two so
                // the pipeline registers never move.  A record comes EITHER from the gather (issued only by the
                // lanes whose hit is outside the window: fewer addresses for the texture path) or from the window
                // (read when the hit is consumed): the registers are declared undefined before the gather, or the
                // compiler keeps "the old value where no load was issued" alive through both conditionals -- 14
                // v_mov per pair body and a second set of record registers (70 -> 57 VGPRs).
#define SD_UNDEF4(q) asm volatile("" : "=v"(q.x), "=v"(q.y), "=v"(q.z), "=v"(q.w));
#ifdef SD_EXP_LDSONLY // PERF-ONLY experiment (results wrong by construction): every hit is read from the window
#define SD_FETCH(j, p, v) SD_UNDEF4(p) SD_UNDEF4(v)
#define SD_USE(j, p, v)                                                        \
    p = win[2 * ((unsigned)(j) % (unsigned)SL_DEAL_WINDOW)];                   \
    v = win[2 * ((unsigned)(j) % (unsigned)SL_DEAL_WINDOW) + 1];               \
    body(p, v);
#else
#define SD_FETCH(j, p, v)                                                      \
    SD_UNDEF4(p)                                                               \
    SD_UNDEF4(v)                                                               \
    if ((unsigned)((j)-w0) >= (unsigned)wlen) {                                \
        p = A.pv8[2 * (size_t)(j)];                                            \
        v = A.pv8[2 * (size_t)(j) + 1];                                        \
    }
#define SD_USE(j, p, v)                                                        \
    if ((unsigned)((j)-w0) < (unsigned)wlen) {                                 \
        p = win[2 * ((j)-w0)];                                                 \
        v = win[2 * ((j)-w0) + 1];                                             \
    }                                                                          \
    body(p, v);
#endif
                float4 p0 = make_float4(0, 0, 0, 0), v0 = p0, p1 = p0, v1 = p0;
                int j0 = pop();
                SD_FETCH(j0, p0, v0)
                int j1 = pop();
                SD_FETCH(j1, p1, v1)
                // (`live` after a pop: some lane has a hit beyond the ones popped so far)
                for (;;) {
                    SD_USE(j0, p0, v0)
                    if (!__ballot(live)) { SD_USE(j1, p1, v1) break; }
                    j0 = pop();
                    SD_FETCH(j0, p0, v0)
                    SD_USE(j1, p1, v1)
                    if (!__ballot(live)) { SD_USE(j0, p0, v0) break; }
                    j1 = pop();
                    SD_FETCH(j1, p1, v1)
                }
#undef SD_FETCH
#undef SD_USE
#undef SD_UNDEF4
            }
        }
        if (valid) {
            float vx = vi.x, vy = vi.y, vz = vi.z;
            integrate_particle(P, pi, vx, vy, vz, F, vi.w);
            store_particle(A, i, pi, vx, vy, vz, vi.w, F);
        }
    }
}

// Slabs: the halo rows' half of "every row is quiet".  A slab's density sweep clears *quietAll at its first
// owned row that is not quiet; the halo rows' densities arrive with exchange B, so their test -- the same one:
// no pressure, the slab's reference velocity -- runs afterwards, on the stream of the force launch that
// holds the rows next to the halo layers.  (Free fall, N = 8 slabs of the headline run: that launch then
// skips its hit stream like the single domain's.)
__global__ __launch_bounds__(256) void k_halo_quiet(const float4 *__restrict__ pv8, int lo_end, int hi_begin, int n_all,
                                                    const float4 *__restrict__ vref, const uint32_t *__restrict__ ownedQuiet,
                                                    uint32_t *__restrict__ haloQuiet) {
    if (*ownedQuiet == 0u) return; // (nobody will ask)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lo_end + (n_all - hi_begin)) return;
    const int j = t < lo_end ? t : hi_begin + (t - lo_end);
    const float4 v = pv8[2 * (size_t)j + 1], r = *vref;
    const float prs = fmaxf(0.f, SPH_GAS_CONSTANT * (v.w - SPH_REST_DENSITY));
    if (!(prs == 0.f && v.x == r.x && v.y == r.y && v.z == r.z)) *haloQuiet = 0u; // (every writer writes the same value)
}
void sph_launch_halo_quiet(const float4 *pv8, int lo_end, int hi_begin, int n_all, const float4 *vref,
                           const uint32_t *ownedQuiet, uint32_t *haloQuiet, hipStream_t s) {
    const int rows = lo_end + (n_all - hi_begin);
    if (rows > 0) k_halo_quiet<<<(rows + 255) / 256, 256, 0, s>>>(pv8, lo_end, hi_begin, n_all, vref, ownedQuiet, haloQuiet);
}

// SPH_FLAG_COUNT_PAIRS only (bench.py's untimed counting replay): pair bodies the force
// sweep will evaluate = set bits of the recorded masks.  A kernel of its own so that the
// production kernels carry no counting code.
__global__ __launch_bounds__(256) void k_count_hits(SweepArgs A) {
    const int i = A.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 0, dropped = 0;
    if (i < A.i_end) {
        const int wv = (i - A.i_origin) >> 6;
        const uint32_t baseq = A.maskOff[2 * (size_t)wv];
        const int Q = (int)A.maskOff[2 * (size_t)wv + 1];
        if (baseq != SL_NONE) {
            const uint4 *q4 = reinterpret_cast<const uint4 *>(A.maskPool) + baseq + (threadIdx.x & 63);
            const bool qi = A.quiet && ((A.quiet[i >> 5] >> (i & 31)) & 1u);
            for (int q = 0; q < Q; ++q) {
                const uint4 t = q4[(size_t)q * SPH_WAVE];
                n += (uint32_t)__builtin_popcount(t.y) + (uint32_t)__builtin_popcount(t.w);
                if (qi) // what the zero-pair filter drops (k_force_list)
                    dropped += (uint32_t)__builtin_popcount(t.y & sl_quiet_window(A.quiet, t.x)) +
                               (uint32_t)__builtin_popcount(t.w & sl_quiet_window(A.quiet, t.z));
                if (t.w == 0u) break;
            }
        }
    }
    n = wave_sum_u32(n);
    dropped = wave_sum_u32(dropped);
    if ((threadIdx.x & 63) == 0 && n) {
        atomicAdd(A.pairCounter + 16 + (blockIdx.x & 255) * 16 + 15, (unsigned long long)n);
        atomicAdd(A.pairCounter + 16 + (blockIdx.x & 255) * 16 + 14, (unsigned long long)(n - dropped));
    }
}

void sph_launch_density_list(const DevParams &P, const SweepArgs &A, int mathMode, hipStream_t s) {
    int cnt = A.i_end - A.i_begin;
    if (cnt <= 0) return;
    int blocks = (A.i_end - A.i_origin + SL_K1_THREADS - 1) / SL_K1_THREADS; // waves are numbered from i_origin
    const bool same = P.cut2 == P.h2;
    if (mathMode == 1) {
        if (same) k_density_mask_lds<true, true><<<blocks, SL_K1_THREADS, 0, s>>>(P, A);
        else k_density_mask_lds<true, false><<<blocks, SL_K1_THREADS, 0, s>>>(P, A);
    } else {
        if (same) k_density_mask_lds<false, true><<<blocks, SL_K1_THREADS, 0, s>>>(P, A);
        else k_density_mask_lds<false, false><<<blocks, SL_K1_THREADS, 0, s>>>(P, A);
    }
    if (A.pairCounter) k_count_hits<<<(cnt + 255) / 256, 256, 0, s>>>(A);
}

// Particles whose wave found the mask pool exhausted in the density sweep: test
// every candidate, like the check path.  Launched after the force sweep every step with a
// small fixed grid that walks the density sweep's list of such waves (none in the runs
// measured: the launch is a handful of waves reading one word -- a launch over every row
// that checked maskOff[] cost 8 us per step at n = 4 M, 4-7 us of a 220 us slab step).
#define SL_FALLBACK_BLOCKS 512
template <bool FAST, bool SLIM>
__global__ __launch_bounds__(SPH_WAVE) void k_force_fallback(DevParams P, SweepArgs A) {
    const unsigned count = (unsigned)A.maskCursor[1];
    for (unsigned idx = blockIdx.x; idx < count; idx += gridDim.x) {
    // the rows of this force launch (up to two ranges) among the wave's 64
    const int i = A.i_origin + (int)A.noneList[idx] * SPH_WAVE + (int)threadIdx.x;
    const bool mine = (i >= A.i_begin && i < A.i_end) || (i >= A.i_begin2 && i < A.i_end2);
    if (!__ballot(mine)) continue;
    const int iSafe = mine ? i : A.i_begin;
    float4 pi = A.pv8[2 * (size_t)iSafe];
    const float4 vi = A.pv8[2 * (size_t)iSafe + 1];
    const float prs_i = fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY));
    int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    int js[9], je[9];
    load_runs(P, A.cellRange, c, mine, js, je);
    ForceAcc F = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 9; ++r)
        for (int j = js[r]; j < je[r]; ++j) {
            if (FAST) force_pair_fast(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, A.pv8[2 * (size_t)j], A.pv8[2 * (size_t)j + 1], F);
            else force_pair<SLIM>(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, A.pv8[2 * (size_t)j], A.pv8[2 * (size_t)j + 1], F);
        }
    if (mine) {
        float vx = vi.x, vy = vi.y, vz = vi.z;
        integrate_particle(P, pi, vx, vy, vz, F, vi.w);
        store_particle(A, i, pi, vx, vy, vz, vi.w, F);
    }
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

// rows outside [i_begin, i_end) of the n_all sorted rows are halo rows (A.i_begin/i_end
// = the OWNED range here, whatever sub-range the force launches cover)
void sph_launch_patch_halo(const SweepArgs &A, hipStream_t s) {
    const int halo = A.i_begin + (A.n_all - A.i_end);
    if (halo > 0) k_patch_pv8<<<(halo + 255) / 256, 256, 0, s>>>(A.vel4, A.pv8, A.i_begin, A.i_end, A.n_all);
}

void sph_launch_force_list(const DevParams &P, const SweepArgs &A, int mathMode, hipStream_t s) {
    SweepArgs B = A;
    if (B.i_end2 <= B.i_begin2) B.i_begin2 = B.i_end2 = 0;
    if (B.i_end <= B.i_begin) { // only the second range holds rows
        B.i_begin = B.i_begin2;
        B.i_end = B.i_end2;
        B.i_begin2 = B.i_end2 = 0;
    }
    if (B.i_end <= B.i_begin) return;
    if (B.patchHalo) sph_launch_patch_halo(B, s);
    // groups of the stream's waves (numbered from i_origin) that hold rows of a range
#if SL_DEAL
    const int wpb = SL_DEAL_ROWS / SPH_WAVE, threads = SL_DEAL_THREADS;
#define SL_FORCE_KERNEL k_force_dealt
#else
    const int wpb = SL_K2_THREADS / SPH_WAVE, threads = SL_K2_THREADS;
#define SL_FORCE_KERNEL k_force_list
#endif
    auto blocks_of = [&](int a, int b) {
        if (b <= a) return 0;
        const int w0 = (a - B.i_origin) >> 6, w1 = (b - B.i_origin + 63) >> 6;
        return (w1 + wpb - 1) / wpb - w0 / wpb;
    };
    B.nblk1 = blocks_of(B.i_begin, B.i_end);
    const int fblocks = B.nblk1 + blocks_of(B.i_begin2, B.i_end2);
    const int blocks = SL_FALLBACK_BLOCKS;
    if (mathMode == 1) {
        SL_FORCE_KERNEL<true, true><<<fblocks, threads, 0, s>>>(P, B);
        k_force_fallback<true, true><<<blocks, SPH_WAVE, 0, s>>>(P, B);
    } else if (P.slimDiv) { // the reference's h and kernel coefficients (sweep_common.h)
        SL_FORCE_KERNEL<false, true><<<fblocks, threads, 0, s>>>(P, B);
        k_force_fallback<false, true><<<blocks, SPH_WAVE, 0, s>>>(P, B);
    } else {
        SL_FORCE_KERNEL<false, false><<<fblocks, threads, 0, s>>>(P, B);
        k_force_fallback<false, false><<<blocks, SPH_WAVE, 0, s>>>(P, B);
    }
#undef SL_FORCE_KERNEL
}
