// Neighbour-grid build around the radix sort: particle -> cell hash, and the
// fused "apply permutation + detect cell boundaries" pass that replaces the
// reference's per-cell linked-list heads (simulator.cu:133-147) with a
// {start,end} range per flattened cell over the key-sorted particle streams.
// Also the click impulse (kernelMoveParticles, simulator.cu:329-367).
#include "sph_device.h"

#include <algorithm>

// getGridCell + flattenGridCoord (simulator.cu:57-82).  The division is the
// IEEE fp32 divide of the reference (hipcc's default `/` is correctly rounded);
// the flattening is done in integers, which equals the reference's float
// evaluation exactly below 2^24.  Cells are clamped into the table so a
// position outside the box can never index out of bounds (the reference only
// printf's in that case, simulator.cu:60-73).
__device__ __forceinline__ int3 grid_cell(const DevParams &P, float x, float y,
                                          float z) {
    int3 c;
    c.x = (int)(x / P.h);
    c.y = (int)(y / P.h);
    c.z = (int)(z / P.h);
    c.x = min(max(c.x, 0), P.D - 1);
    c.y = min(max(c.y, 0), P.D - 1);
    c.z = min(max(c.z, 0), P.D - 1);
    return c;
}

__global__ __launch_bounds__(256) void k_hash(DevParams P,
                                              const float4 *__restrict__ pos4,
                                              uint32_t *__restrict__ keys,
                                              uint32_t *__restrict__ vals, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pos4[i];
    int3 c = grid_cell(P, p.x, p.y, p.z);
    keys[i] = sph_cell_key(P, c.x, c.y, c.z);
    vals[i] = (uint32_t)i;
}

void sph_launch_hash(const DevParams &P, const float4 *pos4, uint32_t *keys,
                     uint32_t *vals, int n, hipStream_t s) {
    if (n <= 0) return;
    k_hash<<<(n + 255) / 256, 256, 0, s>>>(P, pos4, keys, vals, n);
}

// One pass over the sorted (key, source slot) pairs: move both float4 streams
// into sorted order (16-B gathers, 16-B coalesced stores) and write cell
// boundaries.  A lane compares its key with its wave neighbours through DPP
// shuffles; only lanes 0 and 63 touch memory for the key next door.
// cellRange must have been cleared beforehand (the first sort pass does it): empty cells
// keep {0,0}.
void sph_launch_lower_bounds(const uint32_t *sorted_keys, int n, Thresholds thr, int nthr,
                             int *bounds_dev, hipStream_t s);

__global__ __launch_bounds__(256) void k_gather_cells(
    const float4 *__restrict__ pos_in, const float4 *__restrict__ vel_in,
    const uint32_t *__restrict__ perm, const uint32_t *__restrict__ skeys,
    float4 *__restrict__ pos_out, float4 *__restrict__ vel_out,
    float4 *__restrict__ pv8, int2 *__restrict__ cellRange, int n, GatherExtras X) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // riders of this launch (each was a launch of its own): the hit-stream pool's allocation
    // cursors are cleared for the density sweep that follows ...
    if (i < X.cursorWords) X.cursor[i] = 0ull;
    if (i < X.quietWords) X.quietClear[i] = 0u;
    if (X.quietAll && i == 0) X.quietAll[0] = X.quietAll[1] = 1u; // ("every owned row", "every halo row" is quiet)
    bool valid = i < n;
    uint32_t k = valid ? skeys[i] : 0xFFFFFFFFu;
    uint32_t kprev = __shfl_up(k, 1);
    uint32_t knext = __shfl_down(k, 1);
    if (lane == 0) kprev = (valid && i > 0) ? skeys[i - 1] : 0xFFFFFFFFu;
    if (lane == 63) knext = (i + 1 < n) ? skeys[i + 1] : 0xFFFFFFFFu;
    if (!valid) return;
    if (i + 1 >= n) knext = 0xFFFFFFFFu;
    uint32_t src = perm[i];
    const float4 p = pos_in[src], v = vel_in[src];
    // ... and the zero-pair filter's "moves with the reference velocity" bit of every row (the reference was
    // picked by the first sort pass of this build); one 64-bit word per 64 rows
    if (X.calm) {
        const float4 vr = *X.vref;
        const unsigned long long cb = __ballot(v.x == vr.x && v.y == vr.y && v.z == vr.z);
        if (lane == 0) X.calm[i >> 6] = cb;
    }
    pos_out[i] = p;
    if (vel_out) vel_out[i] = v; // (null: the list sweeps read velocities from pv8 only)
    if (pv8) { // interleaved copy for the list sweep's force gathers (one line per hit)
        pv8[2 * (size_t)i] = p;
        pv8[2 * (size_t)i + 1] = v;
    }
    if (k != kprev) cellRange[k].x = i;
    if (k != knext) cellRange[k].y = i + 1;
    // ... and the slab path's segment bounds: bounds[t] = first index whose key is >= thr[t]
    // (n if there is none), bounds[nthr] = n
    if (X.bounds) {
        const bool first = i == 0, last = i + 1 >= n;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= X.nthr) break;
            const uint32_t v = X.thr.v[t];
            if (k >= v && (first || kprev < v)) X.bounds[t] = i;
            if (last && k < v) X.bounds[t] = n;
        }
        if (last) X.bounds[X.nthr] = n;
    }
}

void sph_launch_gather(const float4 *pos_in, const float4 *vel_in,
                       const uint32_t *perm, const uint32_t *sorted_keys,
                       float4 *pos_out, float4 *vel_out, float4 *pv8, int2 *cellRange, int n,
                       hipStream_t s, const GatherExtras &X) {
    if (n <= 0) { // nothing to gather: the riders still have to happen (no rows: no quiet bits to clear)
        if (X.cursor && X.cursorWords > 0)
            (void)hipMemsetAsync(X.cursor, 0, (size_t)X.cursorWords * sizeof(unsigned long long), s);
        if (X.bounds) sph_launch_lower_bounds(sorted_keys, 0, X.thr, X.nthr, X.bounds, s);
        return;
    }
    const int threads = std::max(n, std::max(X.cursorWords, X.quietWords));
    k_gather_cells<<<(threads + 255) / 256, 256, 0, s>>>(pos_in, vel_in, perm, sorted_keys,
                                                         pos_out, vel_out, pv8, cellRange, n, X);
}

// ---- slab path: stable partition by key class (which segment of the slab's key
// range a particle's NEW cell falls in) instead of a full sort.  class = number of
// thresholds <= key; the one-pass radix sort on that small key keeps the previous
// order inside every class, which is all the exchange needs (the combined array is
// sorted by the full key afterwards).
__global__ __launch_bounds__(256) void k_classify(DevParams P, const float4 *__restrict__ pos4,
                                                  Thresholds thr, int nthr,
                                                  uint32_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pos4[i];
    int3 c = grid_cell(P, p.x, p.y, p.z);
    const uint32_t key = sph_cell_key(P, c.x, c.y, c.z);
    uint32_t cls = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cls += (k < nthr && key >= thr.v[k]) ? 1u : 0u;
    keys[i] = cls;
    vals[i] = (uint32_t)i;
}

void sph_launch_classify(const DevParams &P, const float4 *pos4, Thresholds thr, int nthr,
                         uint32_t *keys, uint32_t *vals, int n, hipStream_t s) {
    if (n <= 0) return;
    k_classify<<<(n + 255) / 256, 256, 0, s>>>(P, pos4, thr, nthr, keys, vals, n);
}

// ---- slab path: the same stable partition in TWO launches instead of seven (classify, one
// radix pass = histogram + row scan + scatter, gather, segment bounds, header copy).  At the
// slab sizes this path sees (a few 10^5 particles per GPU) every launch costs its latency, not
// its bytes.  Tiles of 1024 particles; a wave owns 256 consecutive ones (4 rounds of 64), so
// (tile, wave, round, lane) order is index order and ranks are stable.
//   k_partition_count : class of every particle, per-tile class counts        -> tileCount[tile][8]
//   k_partition_move  : every tile sums the counts of the tiles before it (a few KB from L2),
//                       ranks its particles by wave ballots and moves pos4/vel4 to their place;
//                       tile 0 also writes the bounds (and n) for the message header.
#define PT_THREADS 256
#define PT_ITEMS 4
#define PT_TILE (PT_THREADS * PT_ITEMS)

__device__ __forceinline__ uint32_t partition_class(const DevParams &P, const Thresholds &thr, int nthr, float4 p) {
    int3 c = grid_cell(P, p.x, p.y, p.z);
    const uint32_t key = sph_cell_key(P, c.x, c.y, c.z);
    uint32_t cls = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cls += (k < nthr && key >= thr.v[k]) ? 1u : 0u;
    return cls;
}

__global__ __launch_bounds__(PT_THREADS) void k_partition_count(DevParams P, const float4 *__restrict__ pos4,
                                                                Thresholds thr, int nthr, int n,
                                                                int *__restrict__ tileCount) {
    __shared__ int cnt[9];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 9) cnt[t] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * PT_TILE + (long long)w * (SPH_WAVE * PT_ITEMS) + lane;
#pragma unroll
    for (int r = 0; r < PT_ITEMS; ++r) {
        const long long idx = base + r * SPH_WAVE;
        const bool valid = idx < n;
        const uint32_t cls = valid ? partition_class(P, thr, nthr, pos4[idx]) : 0xFFu;
#pragma unroll
        for (int c = 0; c < 9; ++c) { // wave-uniform trip count; nthr <= 8 => classes 0..8
            const unsigned long long m = __ballot(valid && cls == (uint32_t)c);
            if (lane == 0 && m) atomicAdd(&cnt[c], (int)__popcll(m));
        }
    }
    __syncthreads();
    if (t < 9) tileCount[(size_t)blockIdx.x * 9 + t] = cnt[t];
}

__global__ __launch_bounds__(PT_THREADS) void k_partition_move(DevParams P, const float4 *__restrict__ pos_in,
                                                               const float4 *__restrict__ vel_in,
                                                               float4 *__restrict__ pos_out,
                                                               float4 *__restrict__ vel_out, Thresholds thr,
                                                               int nthr, int n, const int *__restrict__ tileCount,
                                                               int numTiles, int *__restrict__ bounds_out) {
    __shared__ int before[9], total[9], classBase[10];
    __shared__ int waveCnt[PT_THREADS / SPH_WAVE][9];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 9) before[t] = total[t] = 0;
    if (t < 36) waveCnt[t / 9][t % 9] = 0;
    __syncthreads();
    // class c = t % 9 (threads 0..251: 28 rows of 9), tiles strided by 28
    if (t < 252) {
        const int c = t % 9;
        int b = 0, a = 0;
        for (int q = t / 9; q < numTiles; q += 28) {
            const int v = tileCount[(size_t)q * 9 + c];
            a += v;
            b += q < (int)blockIdx.x ? v : 0;
        }
        atomicAdd(&before[c], b);
        atomicAdd(&total[c], a);
    }
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int c = 0; c < 9; ++c) {
            classBase[c] = run;
            run += total[c];
        }
        classBase[9] = run;
        if (blockIdx.x == 0 && bounds_out) { // bounds[k] = #particles of class <= k; then n
            for (int k = 0; k < nthr; ++k) bounds_out[k] = classBase[k + 1];
            bounds_out[nthr] = n;
        }
    }
    __syncthreads();
    const long long base = (long long)blockIdx.x * PT_TILE + (long long)w * (SPH_WAVE * PT_ITEMS) + lane;
    float4 p[PT_ITEMS], v[PT_ITEMS];
    uint32_t cls[PT_ITEMS];
    int rank[PT_ITEMS]; // rank inside this wave's 256 particles, among its class
#pragma unroll
    for (int r = 0; r < PT_ITEMS; ++r) {
        const long long idx = base + r * SPH_WAVE;
        const bool valid = idx < n;
        p[r] = valid ? pos_in[idx] : make_float4(0, 0, 0, 0);
        v[r] = valid ? vel_in[idx] : make_float4(0, 0, 0, 0);
        cls[r] = valid ? partition_class(P, thr, nthr, p[r]) : 0xFFu;
        rank[r] = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const unsigned long long m = __ballot(valid && cls[r] == (uint32_t)c);
            if (cls[r] == (uint32_t)c)
                rank[r] = waveCnt[w][c] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            __builtin_amdgcn_wave_barrier();
            if (lane == 0 && m) waveCnt[w][c] += (int)__popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PT_ITEMS; ++r) {
        const long long idx = base + r * SPH_WAVE;
        if (idx < n) {
            const int c = (int)cls[r];
            int off = 0; // particles of class c in the waves of this tile before mine
            for (int q = 0; q < w; ++q) off += waveCnt[q][c];
            const int dst = classBase[c] + before[c] + off + rank[r];
            pos_out[dst] = p[r];
            vel_out[dst] = v[r];
        }
    }
}

void sph_launch_partition(const DevParams &P, const float4 *pos_in, const float4 *vel_in, float4 *pos_out,
                          float4 *vel_out, Thresholds thr, int nthr, int n, int *tileCount, int *bounds_dev,
                          hipStream_t s) {
    const int tiles = n > 0 ? (n + PT_TILE - 1) / PT_TILE : 1;
    if (n > 0) k_partition_count<<<tiles, PT_THREADS, 0, s>>>(P, pos_in, thr, nthr, n, tileCount);
    // (n == 0: one tile that finds no particle still writes the bounds: all zero)
    if (n <= 0) (void)hipMemsetAsync(tileCount, 0, 9 * sizeof(int), s);
    k_partition_move<<<tiles, PT_THREADS, 0, s>>>(P, pos_in, vel_in, pos_out, vel_out, thr, nthr, n, tileCount, tiles,
                                                  bounds_dev);
}

size_t sph_partition_tiles(int n) { return (size_t)((n + PT_TILE - 1) / PT_TILE + 1); }

__global__ __launch_bounds__(256) void k_gather_plain(const float4 *__restrict__ pos_in,
                                                      const float4 *__restrict__ vel_in,
                                                      const uint32_t *__restrict__ perm,
                                                      float4 *__restrict__ pos_out,
                                                      float4 *__restrict__ vel_out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = perm[i];
    pos_out[i] = pos_in[src];
    vel_out[i] = vel_in[src];
}

void sph_launch_gather_plain(const float4 *pos_in, const float4 *vel_in, const uint32_t *perm,
                             float4 *pos_out, float4 *vel_out, int n, hipStream_t s) {
    if (n <= 0) return;
    k_gather_plain<<<(n + 255) / 256, 256, 0, s>>>(pos_in, vel_in, perm, pos_out, vel_out, n);
}

// Assembly of a slab's combined array: up to 8 (pos4, vel4) row ranges copied to their
// places in ONE launch (they are small -- a boundary layer, a handful of migrants --
// and fourteen separate copies were launch-bound).
__global__ __launch_bounds__(256) void k_copy_segments(SegmentTable T, float4 *__restrict__ dpos,
                                                       float4 *__restrict__ dvel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.prefix[T.n]) return;
    int k = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) k += (q < T.n && i >= T.prefix[q]) ? 1 : 0;
    const int r = i - T.prefix[k];
    dpos[T.dst[k] + r] = T.spos[k][r];
    dvel[T.dst[k] + r] = T.svel[k][r];
}

void sph_launch_copy_segments(const SegmentTable &T, float4 *dpos, float4 *dvel, hipStream_t s) {
    const int total = T.prefix[T.n];
    if (total <= 0) return;
    k_copy_segments<<<(total + 255) / 256, 256, 0, s>>>(T, dpos, dvel);
}

__global__ void k_lower_bounds(const uint32_t *__restrict__ keys, int n, Thresholds thr,
                               int nthr, int *__restrict__ out) {
    int t = threadIdx.x;
    if (t == nthr) out[nthr] = n; // the element count rides along (slab message headers)
    if (t >= nthr) return;
    uint32_t v = thr.v[t];
    int lo = 0, hi = n; // first index with keys[idx] >= v
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    out[t] = lo;
}

void sph_launch_lower_bounds(const uint32_t *sorted_keys, int n, Thresholds thr, int nthr,
                             int *bounds_dev, hipStream_t s) {
    k_lower_bounds<<<1, 64, 0, s>>>(sorted_keys, n, thr, nthr, bounds_dev);
}

// kernelMoveParticles (simulator.cu:329-367).  The reference launches <<<1, numCellsPerDim>>>: thread t
// owns the z-layer (int)((float)t*h/h) and walks the 5 x 5 cells around the click through their
// linked lists, one particle after the other; two t that round to the same layer race on the
// velocities (the oracle applies them one after the other, sph_oracle.c).  Here: one 64-lane wave
// per (cell of the 5 x 5 footprint, z-layer), the cell's particles in parallel through the cell
// table of the last grid build (the pre-integration grid, like the reference); a wave first counts
// how many of the reference's threads own its layer (0, 1 or 2) and applies the impulse that many
// times in sequence -- the serial order's result, without the race.
// zlo/zhi: only the layers [zlo, zhi) act (a slab applies the impulse to the layers it owns).
__global__ __launch_bounds__(SPH_WAVE) void k_click(DevParams P, const int2 *__restrict__ cellRange,
                                                    float4 *__restrict__ vel4, int mx, int my, int zlo, int zhi) {
    const int cz = blockIdx.y, lane = threadIdx.x;
    if (cz < zlo || cz >= zhi) return;
    int owners = 0; // reference threads t with (int)((float)t * h / h) == cz
    for (int t = lane; t < P.D; t += SPH_WAVE) owners += ((int)(((float)t * P.h) / P.h) == cz) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) owners += __shfl_xor(owners, off);
    if (owners == 0) return;
    const float x = ((float)(mx - SPH_BOX_MIN_X) / (float)(SPH_BOX_MAX_X - SPH_BOX_MIN_X)) * P.boxDim;
    const float y = ((float)(my - SPH_BOX_MIN_Y) / (float)(SPH_BOX_MAX_Y - SPH_BOX_MIN_Y)) * P.boxDim;
    const int cx = (int)(x / P.h);
    const int cy = (int)((float)P.D - (float)(int)(y / P.h));
    const int dy = (int)blockIdx.x / 5 - 2, dx = (int)blockIdx.x % 5 - 2;
    const int sy = cy + dy, sx = cx + dx;
    if (sy < 0 || sy >= P.D || sx < 0 || sx >= P.D) return;
    const int2 r = cellRange[sph_cell_key(P, sx, sy, cz)];
    for (int j = r.x + lane; j < r.y; j += SPH_WAVE) {
        float4 v = vel4[j];
        for (int k = 0; k < owners; ++k) {
            if (dx != 0) v.x += (1.f / dx) * SPH_PUSH_STRENGTH;
            if (dy != 0) v.y += (1.f / dy) * SPH_PUSH_STRENGTH;
            if (dx == 0 && dy == 0) v.z -= SPH_PUSH_STRENGTH;
        }
        vel4[j] = v;
    }
}

void sph_launch_click(const DevParams &P, const int2 *cellRange, float4 *vel4, int mx,
                      int my, hipStream_t s, int zlo, int zhi) {
    k_click<<<dim3(25, P.D), SPH_WAVE, 0, s>>>(P, cellRange, vel4, mx, my, zlo, zhi);
}
