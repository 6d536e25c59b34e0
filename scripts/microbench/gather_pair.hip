// Does the texture addresser charge a divergent 16-B gather per LANE or per distinct LINE?
// 64-lane waves fetch one 32-B record per lane and trip from a 134 MB table, the record
// index drawn from a 2,304-record neighbourhood of the wave's own position (the force
// sweep's access pattern), two gathers in flight:
//   A: every lane loads both halves of its own record (2 instructions, 64 distinct lines each)
//   B: lane pairs load the two halves of ONE record per instruction (2 instructions, 32 distinct
//      lines each, 32 contiguous bytes per pair), then swap halves with DPP
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off gather_pair.hip -o gather_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define TRIPS 46

__device__ __forceinline__ unsigned lcg(unsigned &s) {
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

template <int MODE>
__global__ __launch_bounds__(64) void k(const float4 *__restrict__ tab, float *__restrict__ out, int nrec, int HOOD) {
    __shared__ float pad_lds[1536]; // 6 KB per one-wave workgroup: the same 26 waves per CU in every mode
    const int lane = threadIdx.x, wave = blockIdx.x;
    if (HOOD < 0) pad_lds[lane] = 1.f;
    const int base = min(max(wave * 64 - HOOD / 2, 0), nrec - HOOD);
    unsigned s = wave * 64u + lane + 12345u;
    float acc = 0.f;
    if (MODE == 3) { // the record as four 8-byte loads
        const float2 *tab2 = reinterpret_cast<const float2 *>(tab);
        int j0 = base + lcg(s) % HOOD, j1 = base + lcg(s) % HOOD;
        float2 a0 = tab2[4 * (size_t)j0], b0 = tab2[4 * (size_t)j0 + 1], c0 = tab2[4 * (size_t)j0 + 2], d0 = tab2[4 * (size_t)j0 + 3];
        float2 a1 = tab2[4 * (size_t)j1], b1 = tab2[4 * (size_t)j1 + 1], c1 = tab2[4 * (size_t)j1 + 2], d1 = tab2[4 * (size_t)j1 + 3];
        for (int t = 0; t < TRIPS; t += 2) {
            acc += a0.x * c0.y + b0.x * d0.y;
            j0 = base + lcg(s) % HOOD;
            a0 = tab2[4 * (size_t)j0]; b0 = tab2[4 * (size_t)j0 + 1]; c0 = tab2[4 * (size_t)j0 + 2]; d0 = tab2[4 * (size_t)j0 + 3];
            acc += a1.x * c1.y + b1.x * d1.y;
            j1 = base + lcg(s) % HOOD;
            a1 = tab2[4 * (size_t)j1]; b1 = tab2[4 * (size_t)j1 + 1]; c1 = tab2[4 * (size_t)j1 + 2]; d1 = tab2[4 * (size_t)j1 + 3];
        }
        acc += a0.x + c0.x + a1.x + c1.x;
    } else if (MODE == 2) { // position half only: one gather per trip
        int j0 = base + lcg(s) % HOOD, j1 = base + lcg(s) % HOOD;
        float4 p0 = tab[2 * (size_t)j0], p1 = tab[2 * (size_t)j1];
        for (int t = 0; t < TRIPS; t += 2) {
            acc += p0.x * p0.y + p0.z * p0.w;
            j0 = base + lcg(s) % HOOD;
            p0 = tab[2 * (size_t)j0];
            acc += p1.x * p1.y + p1.z * p1.w;
            j1 = base + lcg(s) % HOOD;
            p1 = tab[2 * (size_t)j1];
        }
        acc += p0.x + p1.x;
    } else if (MODE == 0) {
        int j0 = base + lcg(s) % HOOD, j1 = base + lcg(s) % HOOD;
        float4 p0 = tab[2 * (size_t)j0], v0 = tab[2 * (size_t)j0 + 1];
        float4 p1 = tab[2 * (size_t)j1], v1 = tab[2 * (size_t)j1 + 1];
        for (int t = 0; t < TRIPS; t += 2) {
            acc += p0.x * v0.y + p0.z * v0.w;
            j0 = base + lcg(s) % HOOD;
            p0 = tab[2 * (size_t)j0];
            v0 = tab[2 * (size_t)j0 + 1];
            acc += p1.x * v1.y + p1.z * v1.w;
            j1 = base + lcg(s) % HOOD;
            p1 = tab[2 * (size_t)j1];
            v1 = tab[2 * (size_t)j1 + 1];
        }
        acc += p0.x + v0.x;
        acc += p1.x + v1.x;
    } else {
        const bool odd = lane & 1;
        auto fetch = [&](int j, float4 &a, float4 &b) {
            // partner's index: quad_perm [1,0,3,2]
            const int jp = __builtin_amdgcn_mov_dpp(j, 0xB1, 0xF, 0xF, true);
            const int je = odd ? jp : j, jo = odd ? j : jp; // record of the even / odd lane of the pair
            a = tab[2 * (size_t)je + (odd ? 1 : 0)];       // even: A.pos, odd: A.vel
            b = tab[2 * (size_t)jo + (odd ? 1 : 0)];       // even: B.pos, odd: B.vel
        };
        auto fix = [&](float4 a, float4 b, float4 &p, float4 &v) {
            const float4 send = odd ? a : b; // odd sends A.vel, even sends B.pos
            float4 got;
            got.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.x), 0xB1, 0xF, 0xF, true));
            got.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.y), 0xB1, 0xF, 0xF, true));
            got.z = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.z), 0xB1, 0xF, 0xF, true));
            got.w = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.w), 0xB1, 0xF, 0xF, true));
            p = odd ? got : a; // even keeps A.pos; odd receives B.pos
            v = odd ? b : got; // odd keeps B.vel; even receives A.vel
        };
        int j0 = base + lcg(s) % HOOD, j1 = base + lcg(s) % HOOD;
        float4 a0, b0, a1, b1, p, v;
        fetch(j0, a0, b0);
        fetch(j1, a1, b1);
        for (int t = 0; t < TRIPS; t += 2) {
            fix(a0, b0, p, v);
            acc += p.x * v.y + p.z * v.w;
            j0 = base + lcg(s) % HOOD;
            fetch(j0, a0, b0);
            fix(a1, b1, p, v);
            acc += p.x * v.y + p.z * v.w;
            j1 = base + lcg(s) % HOOD;
            fetch(j1, a1, b1);
        }
        fix(a0, b0, p, v);
        acc += p.x + v.x;
        fix(a1, b1, p, v);
        acc += p.x + v.x;
    }
    out[wave * 64 + lane] = acc;
}

int main() {
    const int nrec = 4194304, waves = 65536;
    float4 *tab;
    float *out;
    (void)hipMalloc(&tab, (size_t)nrec * 32);
    (void)hipMalloc(&out, (size_t)waves * 64 * 4);
    std::vector<float> h((size_t)nrec * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 0.001f;
    (void)hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> r0((size_t)waves * 64), r1((size_t)waves * 64);
    // neighbourhood sizes: 64 records = 2 KB (L1-resident), 2,304 = 74 KB (the force sweep's), 65,536 = 2 MB (L2 only)
    for (int hood : {64, 512, 2304, 65536}) {
        float ms[4];
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) { // second timing counts
                (void)hipEventRecord(e0);
                for (int q = 0; q < 5; ++q) {
                    if (mode == 0) k<0><<<waves, 64>>>(tab, out, nrec, hood);
                    else if (mode == 1) k<1><<<waves, 64>>>(tab, out, nrec, hood);
                    else if (mode == 2) k<2><<<waves, 64>>>(tab, out, nrec, hood);
                    else k<3><<<waves, 64>>>(tab, out, nrec, hood);
                }
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms[mode], e0, e1);
            }
            if (mode < 2) (void)hipMemcpy(mode ? r1.data() : r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost);
        }
        size_t bad = 0;
        for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
        printf("neighbourhood %6d records: own-record gathers %.3f ms/launch | lane-pair gathers %.3f (results differ in %zu lanes) | "
               "position half only %.3f | four 8-byte loads %.3f\n", hood, ms[0] / 5, ms[1] / 5, bad, ms[2] / 5, ms[3] / 5);
    }
    return 0;
}
