// Stable LSD radix sort of (cell key, particle slot) pairs -- the sort-based
// replacement for the reference's lock-free linked-list grid build
// (kernelBuildGrid + insertList, simulator.cu:44-55,133-147).  Written for
// gfx950: 8- or 10-bit digits (the 20-bit flattened cell key sorts in TWO 10-bit
// passes), one 4096-key tile per 256-thread workgroup, each of the
// four 64-lane waves owns a CONTIGUOUS 1024-key chunk so that (wave, round,
// lane) order is index order and ranks are stable.  Equal digits inside a wave
// are found with 8 wave-wide ballots (a 64-bit match mask), the prefix popcount
// of that mask is the in-wave rank -- no per-key LDS atomics, no serialisation
// when a whole tile shares one digit (the usual case for the top digit of an
// almost-sorted key stream).
//
// Per pass: k_radix_hist (tile histograms, digit-major) -> k_radix_rowscan
// (one workgroup per digit scans its row of tile counts) -> k_radix_scatter
// (re-rank in the tile, add digit base + tile base, scatter).
// HBM traffic per pass: read 4 B (hist) + read 8 B + write 8 B per pair.
#include "sph_device.h"

#define RS_THREADS 256
// Keys per thread: 16 (4096-key tiles) at large n, 4 (1024-key tiles) below RS_SMALL_N keys,
// where a launch of 4096-key tiles leaves most of the 256 CUs idle and the per-thread loop of
// 16 ballot rounds IS the kernel's latency (n = 262,144: 64 tiles).
#ifndef RS_ITEMS_BIG
#define RS_ITEMS_BIG 16
#endif
#define RS_ITEMS_SMALL 4
#ifdef RS_SMALL_N_OVERRIDE
#define RS_SMALL_N RS_SMALL_N_OVERRIDE
#else
#define RS_SMALL_N (3 << 19)
#endif
#define RS_WAVES (RS_THREADS / SPH_WAVE)
#ifndef RS_HIST_BALLOT
#define RS_HIST_BALLOT 0 // 1: count with the scatter's ballot match (A/B)
#endif

// Lanes of this wave whose digit equals mine (among valid lanes).
template <int BITS>
__device__ __forceinline__ unsigned long long match_digit(uint32_t digit,
                                                          bool valid) {
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        bool bit = (digit >> b) & 1u;
        unsigned long long v = __ballot(valid && bit);
        m &= bit ? v : ~v;
    }
    return m;
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    // popcount of m restricted to lanes lower than mine
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// BITS = 8 or 10 bits per pass: the 20-bit flattened cell key sorts in TWO 10-bit
// passes (1024 digits, four per thread) instead of three 8-bit ones.
// HASH: first pass of the grid build -- the key is computed from the particle's position
// here (getGridCell + flattenGridCoord, simulator.cu:57-82) and stored for the scatter
// passes, instead of a separate hash kernel writing keys and an iota of values.
//
// One workgroup per scatter tile (RS_THREADS * RS_ITEMS keys), but with a thread per FOUR keys
// (1024 threads for a 4096-key tile) and all four loads in flight before the first is used:
// round 2's form -- 256 threads walking 16 keys each, every load behind a branch -- was the
// latency of 16 dependent HBM round trips with one load in flight per wave (93-106 us for 92 MB).
// Counting needs no ranks, only run lengths: the input is the previous step's cell-sorted order, so
// consecutive lanes mostly share a digit.  A lane whose left neighbour holds another digit is a
// run HEAD (one DPP compare + ballot); it adds the distance to the next head to the tile
// histogram: one LDS atomic per run instead of one per key, and no two lanes of a run on one
// address (the "wave64 ballot boundary detection" of the north star).
// Positions whose cell lies outside the table: counted (and the first few recorded) in host-mapped
// memory; the host prints the reference's diagnostic at its next synchronisation (sph_api.hip).
__device__ __forceinline__ void oob_report(SphOobLog *log, float4 p, int ux, int uy, int uz) {
    if (!log) return;
    const uint32_t k = __hip_atomic_fetch_add(&log->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (k < SPH_OOB_RECORDS) {
        log->rec[k].cell[0] = ux;
        log->rec[k].cell[1] = uy;
        log->rec[k].cell[2] = uz;
        log->rec[k].pos[0] = p.x;
        log->rec[k].pos[1] = p.y;
        log->rec[k].pos[2] = p.z;
    }
}

#define RS_HIST_ITEMS 4
template <int BITS, bool HASH, int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS *RS_ITEMS / RS_HIST_ITEMS) void k_radix_hist(
    const uint32_t *__restrict__ keys, uint32_t *__restrict__ blockHist, int n,
    int shift, int numBlocks, DevParams P, const float4 *__restrict__ pos4,
    uint32_t *__restrict__ keysOut, int2 *__restrict__ zeroTable, int zeroCount, SphOobLog *oob,
    const float4 *__restrict__ velSample, float4 *__restrict__ vrefOut) {
    constexpr int DIG = 1 << BITS;
    constexpr int HT = RS_THREADS * RS_ITEMS / RS_HIST_ITEMS, RS_TILE = RS_THREADS * RS_ITEMS;
    constexpr int WAVE_KEYS = SPH_WAVE * RS_HIST_ITEMS;
    __shared__ uint32_t hist[DIG];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int d = t; d < DIG; d += HT) hist[d] = 0;
    if (HASH) // kernelResetGrid (simulator.cu:321-326): the cell table is cleared here, not by a launch of its own
        for (int k = blockIdx.x * HT + t; k < zeroCount; k += numBlocks * HT)
            zeroTable[k] = make_int2(0, 0);
    // rider of the grid build's first pass: the reference velocity of the force sweep's zero-pair filter = the
    // most common velocity among 64 rows sampled evenly from the (unsorted) input -- a body of fluid in free fall
    // shares one velocity bit for bit; any choice is correct, a popular one drops the most pairs
    if (HASH && velSample && blockIdx.x == 0 && t < SPH_WAVE) {
        const float4 v = velSample[(long long)lane * n / SPH_WAVE];
        int cnt = 0;
        for (int k = 0; k < SPH_WAVE; ++k) { // (readlane: k is wave-uniform)
            const bool same = __builtin_amdgcn_readlane(__float_as_int(v.x), k) == __float_as_int(v.x) &&
                              __builtin_amdgcn_readlane(__float_as_int(v.y), k) == __float_as_int(v.y) &&
                              __builtin_amdgcn_readlane(__float_as_int(v.z), k) == __float_as_int(v.z);
            cnt += same ? 1 : 0;
        }
        // most matches, then the lowest sample: wave-wide max of cnt * 64 + (63 - lane)
        int best = cnt * SPH_WAVE + (SPH_WAVE - 1 - lane);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off));
        if (best == cnt * SPH_WAVE + (SPH_WAVE - 1 - lane)) *vrefOut = v;
    }
    __syncthreads();
    // a wave owns WAVE_KEYS consecutive keys, RS_HIST_ITEMS rounds of 64
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)w * WAVE_KEYS + lane;
    uint32_t key[RS_HIST_ITEMS];
    if (HASH) {
        float4 p[RS_HIST_ITEMS];
#pragma unroll
        for (int r = 0; r < RS_HIST_ITEMS; ++r) {
            const long long idx = base + r * SPH_WAVE;
            p[r] = pos4[idx < n ? idx : (long long)n - 1]; // (n > 0; the clamped load is discarded)
        }
#pragma unroll
        for (int r = 0; r < RS_HIST_ITEMS; ++r) {
            const long long idx = base + r * SPH_WAVE;
            // IEEE divide like the reference; cells clamped into the table (grid.hip)
            const int ux = (int)(p[r].x / P.h), uy = (int)(p[r].y / P.h), uz = (int)(p[r].z / P.h);
            const int cx = min(max(ux, 0), P.D - 1);
            const int cy = min(max(uy, 0), P.D - 1);
            const int cz = min(max(uz, 0), P.D - 1);
            // the reference's diagnostic (getGridCell, simulator.cu:60-73), once per step here instead of in
            // every kernel that hashes a position; the cell is then clamped into the table where the
            // reference would index out of bounds.  Never taken for states that came through
            // setup()/upload_state() and the integrator's wall clamp (caller-owned slab buffers can).
            if (idx < n && ((ux != cx) | (uy != cy) | (uz != cz))) oob_report(oob, p[r], ux, uy, uz);
            key[r] = sph_cell_key(P, cx, cy, cz);
            if (idx < n) keysOut[idx] = key[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < RS_HIST_ITEMS; ++r) {
            const long long idx = base + r * SPH_WAVE;
            key[r] = keys[idx < n ? idx : (long long)n - 1];
        }
    }
#pragma unroll
    for (int r = 0; r < RS_HIST_ITEMS; ++r) {
        const long long idx = base + r * SPH_WAVE;
        const bool valid = idx < n;
        const uint32_t d = (key[r] >> shift) & (DIG - 1);
#if RS_HIST_BALLOT
        unsigned long long mm = match_digit<BITS>(d, valid);
        if (valid && lanes_below(mm) == 0) atomicAdd(&hist[d], (uint32_t)__popcll(mm));
#else
        const uint32_t dprev = __shfl_up(d, 1);
        const unsigned long long live = __ballot(valid); // a prefix of the wave: lanes [0, count)
        const unsigned long long heads = __ballot(valid && (lane == 0 || d != dprev));
        if (valid && (lane == 0 || d != dprev)) {
            // the next head above this lane, or the end of the valid lanes
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
            const int end = above ? __builtin_ctzll(above) : (int)__popcll(live);
            atomicAdd(&hist[d], (uint32_t)(end - lane));
        }
#endif
    }
    __syncthreads();
    for (int d = t; d < DIG; d += HT) blockHist[(size_t)d * numBlocks + blockIdx.x] = hist[d];
}

// Block-wide inclusive scan of one value per thread (256 threads).
__device__ __forceinline__ uint32_t block_inclusive_scan_256(uint32_t v,
                                                             uint32_t *tmp) {
    const int t = threadIdx.x;
    tmp[t] = v;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < RS_THREADS; off <<= 1) {
        uint32_t add = (t >= off) ? tmp[t - off] : 0u;
        __syncthreads();
        tmp[t] += add;
        __syncthreads();
    }
    return tmp[t];
}

// grid = one workgroup per digit; exclusive scan of that digit's row of tile
// counts, in place, and the digit's total.
__global__ __launch_bounds__(RS_THREADS) void k_radix_rowscan(
    uint32_t *__restrict__ blockHist, uint32_t *__restrict__ digitTotal,
    int numBlocks) {
    __shared__ uint32_t tmp[RS_THREADS];
    uint32_t *row = blockHist + (size_t)blockIdx.x * numBlocks;
    const int t = threadIdx.x;
    const int chunk = (numBlocks + RS_THREADS - 1) / RS_THREADS;
    const int b0 = min(t * chunk, numBlocks), b1 = min(b0 + chunk, numBlocks);
    uint32_t s = 0;
    for (int b = b0; b < b1; ++b) s += row[b];
    uint32_t incl = block_inclusive_scan_256(s, tmp);
    if (t == RS_THREADS - 1) digitTotal[blockIdx.x] = incl;
    uint32_t run = incl - s;
    for (int b = b0; b < b1; ++b) {
        uint32_t c = row[b];
        row[b] = run;
        run += c;
    }
}

// IOTA: the values of this pass are the element indices themselves (first pass of the grid
// build: nothing wrote an iota to memory)
template <int BITS, bool IOTA, int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(
    const uint32_t *__restrict__ keysIn, const uint32_t *__restrict__ valsIn,
    uint32_t *__restrict__ keysOut, uint32_t *__restrict__ valsOut,
    const uint32_t *__restrict__ blockHist,
    const uint32_t *__restrict__ digitTotal, int n, int shift, int numBlocks) {
    constexpr int DIG = 1 << BITS, PER = DIG / RS_THREADS;
    constexpr int RS_WAVE_TILE = SPH_WAVE * RS_ITEMS, RS_TILE = RS_THREADS * RS_ITEMS;
    __shared__ uint32_t waveCount[RS_WAVES][DIG];
    __shared__ uint32_t digitOff[DIG];
    __shared__ uint32_t tmp[RS_THREADS];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
    for (int q = 0; q < RS_WAVES; ++q)
#pragma unroll
        for (int e = 0; e < PER; ++e) waveCount[q][t * PER + e] = 0;
    // thread t owns digits [t*PER, t*PER+PER).  Global base of a digit for this
    // tile = (sum of totals of smaller digits) + (its count in earlier tiles).
    uint32_t tot[PER], sum = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        tot[e] = digitTotal[t * PER + e];
        sum += tot[e];
    }
    uint32_t run = block_inclusive_scan_256(sum, tmp) - sum; // ends with a barrier
    uint32_t myBase[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        myBase[e] = run + blockHist[(size_t)(t * PER + e) * numBlocks + blockIdx.x];
        run += tot[e];
    }

    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
    const long long base =
        (long long)blockIdx.x * RS_TILE + (long long)w * RS_WAVE_TILE + lane;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        long long idx = base + r * SPH_WAVE;
        bool valid = idx < n;
        key[r] = valid ? keysIn[idx] : 0xFFFFFFFFu;
        val[r] = IOTA ? (uint32_t)idx : (valid ? valsIn[idx] : 0u);
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        long long idx = base + r * SPH_WAVE;
        bool valid = idx < n;
        uint32_t d = (key[r] >> shift) & (DIG - 1);
        unsigned long long m = match_digit<BITS>(d, valid);
        uint32_t below = lanes_below(m);
        uint32_t old = waveCount[w][d]; // every lane reads before the leader adds
        rank[r] = old + below;
        if (valid && below == 0) waveCount[w][d] = old + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int d = t * PER + e;
        uint32_t c0 = waveCount[0][d], c1 = waveCount[1][d], c2 = waveCount[2][d];
        waveCount[0][d] = 0;
        waveCount[1][d] = c0;
        waveCount[2][d] = c0 + c1;
        waveCount[3][d] = c0 + c1 + c2;
        digitOff[d] = myBase[e];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        long long idx = base + r * SPH_WAVE;
        if (idx < n) {
            uint32_t d = (key[r] >> shift) & (DIG - 1);
            uint32_t dst = digitOff[d] + waveCount[w][d] + rank[r];
            keysOut[dst] = key[r];
            valsOut[dst] = val[r];
        }
    }
}

size_t sph_sort_workspace_blocks(int n) { // for the smallest tile any launch may use
    return (size_t)((n + RS_THREADS * RS_ITEMS_SMALL - 1) / (RS_THREADS * RS_ITEMS_SMALL));
}

template <int BITS, int ITEMS>
static void radix_pass(const SortWorkspace &ws, int cur, int n, int shift, hipStream_t s,
                       const DevParams *P = nullptr, const float4 *pos4 = nullptr, int2 *zeroTable = nullptr,
                       int zeroCount = 0) {
    const int numBlocks = (n + RS_THREADS * ITEMS - 1) / (RS_THREADS * ITEMS);
    if (pos4) { // first pass of the grid build: hash fused in, values = iota
        k_radix_hist<BITS, true, ITEMS><<<numBlocks, RS_THREADS * ITEMS / RS_HIST_ITEMS, 0, s>>>(nullptr, ws.blockHist, n, shift,
                                                                         numBlocks, *P, pos4, ws.keys[cur],
                                                                         zeroTable, zeroCount, ws.oob, ws.velSample, ws.vrefOut);
        k_radix_rowscan<<<1 << BITS, RS_THREADS, 0, s>>>(ws.blockHist, ws.digitTotal, numBlocks);
        k_radix_scatter<BITS, true, ITEMS><<<numBlocks, RS_THREADS, 0, s>>>(
            ws.keys[cur], nullptr, ws.keys[cur ^ 1], ws.vals[cur ^ 1], ws.blockHist, ws.digitTotal, n, shift,
            numBlocks);
        return;
    }
    k_radix_hist<BITS, false, ITEMS><<<numBlocks, RS_THREADS * ITEMS / RS_HIST_ITEMS, 0, s>>>(ws.keys[cur], ws.blockHist, n, shift,
                                                                      numBlocks, DevParams{}, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
    k_radix_rowscan<<<1 << BITS, RS_THREADS, 0, s>>>(ws.blockHist, ws.digitTotal, numBlocks);
    k_radix_scatter<BITS, false, ITEMS><<<numBlocks, RS_THREADS, 0, s>>>(
        ws.keys[cur], ws.vals[cur], ws.keys[cur ^ 1], ws.vals[cur ^ 1], ws.blockHist,
        ws.digitTotal, n, shift, numBlocks);
}

// fewest passes with 8- or 10-bit digits: <=8: 8 | <=10: 10 | <=16: 8+8 |
// <=20: 10+10 (the 100^3 grid) | <=24: 8+8+8 | <=30: 10+10+10 | else 8-bit passes
static int digit_bits(int bits) {
    return ((bits > 8 && bits <= 10) || (bits > 16 && bits <= 20) || (bits > 24 && bits <= 30)) ? 10 : 8;
}

static int sort_impl(const SortWorkspace &ws, const DevParams *P, const float4 *pos4, int n, int bits,
                     hipStream_t s, int2 *zeroTable = nullptr, int zeroCount = 0) {
    if (n <= 0) {
        if (zeroTable && zeroCount > 0) (void)hipMemsetAsync(zeroTable, 0, (size_t)zeroCount * sizeof(int2), s);
        return 0;
    }
    const int digit = digit_bits(bits);
    const bool small = n < RS_SMALL_N;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += digit) {
        const bool first = shift == 0 && pos4;
        const DevParams *p = first ? P : nullptr;
        const float4 *q = first ? pos4 : nullptr;
        if (digit == 10) {
            if (small) radix_pass<10, RS_ITEMS_SMALL>(ws, cur, n, shift, s, p, q, zeroTable, zeroCount);
            else radix_pass<10, RS_ITEMS_BIG>(ws, cur, n, shift, s, p, q, zeroTable, zeroCount);
        } else {
            if (small) radix_pass<8, RS_ITEMS_SMALL>(ws, cur, n, shift, s, p, q, zeroTable, zeroCount);
            else radix_pass<8, RS_ITEMS_BIG>(ws, cur, n, shift, s, p, q, zeroTable, zeroCount);
        }
        cur ^= 1;
    }
    return cur;
}

// The grid build's sort: cell keys computed from pos4 inside the first histogram pass,
// which also clears the cell table (cellRange[0..numCells)) for k_gather_cells.
int sph_sort_cells(const SortWorkspace &ws, const DevParams &P, const float4 *pos4, int n, int bits,
                   hipStream_t s, int2 *cellRange, int numCells) {
    return sort_impl(ws, &P, pos4, n, bits, s, cellRange, numCells);
}

int sph_sort_pairs(const SortWorkspace &ws, int n, int bits, hipStream_t s) {
    return sort_impl(ws, nullptr, nullptr, n, bits, s);
}
