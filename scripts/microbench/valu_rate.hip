// Microbenchmark: achievable wave64 VALU issue rate on gfx950 for the instruction
// mixes of the SPH sweeps (plain v_mul/v_add/v_sub, no FMA contraction), as a
// function of waves per SIMD.  Prints cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// density body fed from LDS (ds_read_b128 / b96 per candidate, per-lane address)
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float *out, int iters, float a, float b) {
    __shared__ float4 tile[4][260];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = lane; k < 260; k += 64) tile[w][k] = make_float4(k * 1e-3f, k * 2e-3f, k * 3e-3f, 1.f);
    __syncthreads();
    float px = lane * 1e-3f, py = 0.1f, pz = 0.2f, rho = 0.f;
    float hv = a, dv = b;
    asm volatile("" : "+v"(hv), "+v"(dv));
    int idx = lane >> 3; // 8 lanes share a candidate, like 8 particles per cell
    for (int i = 0; i < iters; ++i) {
        const float4 *cur = &tile[w][idx];
        float4 pj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 1) { // predicated address select like the real loop
                const float4 *p = (i + u < iters) ? cur : &tile[w][256];
                pj[u] = p[u];
            } else pj[u] = cur[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float dx = px - pj[u].x, dy = py - pj[u].y, dz = pz - pj[u].z;
            float d2 = dx * dx + dy * dy + dz * dz;
            float diff = fmaxf(hv - d2, 0.f);
            rho += 0.02f * (dv * diff * diff * diff);
        }
        if (MODE != 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" ::"v"(pj[u].w));
        }
        idx = (idx + 4) & 127;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = rho;
}

template <int MODE>
void run_lds(const char *name, int wavesPerSimd) {
    int blocks = 256 * wavesPerSimd;
    float *out;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_lds<MODE><<<blocks, 256>>>(out, 100, 0.01f, 1e3f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_lds<MODE><<<blocks, 256>>>(out, iters, 0.01f, 1e3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double tests = (double)iters * 4;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s waves/SIMD %d : %.3f ms, %.1f cycles(@2.4GHz) per wave-test per SIMD\n", name, wavesPerSimd, ms,
           cyc / (tests * wavesPerSimd));
    (void)hipFree(out);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    float y0 = a, y1 = b, y2 = a + b, y3 = a - b;
    float av = a, bv = b;
    asm volatile("" : "+v"(av), "+v"(bv));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { // independent mul/add pairs (8 chains)
                x0 = x0 * a; y0 = y0 + b; x1 = x1 * a; y1 = y1 + b;
                x2 = x2 * a; y2 = y2 + b; x3 = x3 * a; y3 = y3 + b;
            } else if (MODE == 1) { // the density pair body shape
                float dx = x0 - y0, dy = x1 - y1, dz = x2 - y2;
                float d2 = dx * dx + dy * dy + dz * dz;
                float diff = fmaxf(a - d2, 0.f);
                x3 += 0.02f * (b * diff * diff * diff);
                y0 += 1e-7f; y1 += 1e-7f; y2 += 1e-7f;
            } else if (MODE == 3) { // mul/add with the constants held in VGPRs
                x0 = x0 * av; y0 = y0 + bv; x1 = x1 * av; y1 = y1 + bv;
                x2 = x2 * av; y2 = y2 + bv; x3 = x3 * av; y3 = y3 + bv;
            } else if (MODE == 4) { // independent (non-chained) ops, VGPR operands only
                x0 = y0 * av; x1 = y1 * av; x2 = y2 * av; x3 = y3 * av;
                y0 = x0 + bv; y1 = x1 + bv; y2 = x2 + bv; y3 = x3 + bv;
            } else { // fma chains
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                y0 = fmaf(y0, a, b); y1 = fmaf(y1, a, b); y2 = fmaf(y2, a, b); y3 = fmaf(y3, a, b);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3;
}


// Packed variant: candidates staged SoA (x[], y[], z[]), four per trip through three
// ds_read_b128, arithmetic on float2 pairs (v_pk_add_f32 / v_pk_mul_f32), hit bits
// through the carry; per-candidate validity applied as a +0/+inf penalty on dist2.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_lds_pk(float *out, int iters, float a, float b) {
    __shared__ f4 tx[4][68], ty[4][68], tz[4][68];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = lane; k < 68; k += 64) {
        tx[w][k] = (f4){k * 1e-3f, k * 1.1e-3f, k * 1.2e-3f, k * 1.3e-3f};
        ty[w][k] = (f4){k * 2e-3f, k * 2.1e-3f, k * 2.2e-3f, k * 2.3e-3f};
        tz[w][k] = (f4){k * 3e-3f, k * 3.1e-3f, k * 3.2e-3f, k * 3.3e-3f};
    }
    __syncthreads();
    const float px = lane * 1e-3f, py = 0.1f, pz = 0.2f;
    const f2 PX = {px, px}, PY = {py, py}, PZ = {pz, pz};
    float rho = 0.f;
    float hv = a, dv = b, cut = a * 1.0001f;
    asm volatile("" : "+v"(hv), "+v"(dv), "+v"(cut));
    const f2 H2 = {hv, hv}, DC = {dv, dv}, MS = {0.02f, 0.02f};
    unsigned m = 0, acc = 0;
    int idx = lane >> 3;
    const int lo = lane & 3, hi = iters * 4 - (lane & 1); // a lane-dependent valid range
    for (int i = 0; i < iters; ++i) {
        const f4 X = tx[w][idx & 63], Y = ty[w][idx & 63], Z = tz[w][idx & 63];
        // validity of the four candidates of this trip as penalties (0 or +inf)
        const int k0 = 4 * i;
        f4 pen;
        pen.x = (k0 + 0 >= lo && k0 + 0 < hi) ? 0.f : __builtin_inff();
        pen.y = (k0 + 1 >= lo && k0 + 1 < hi) ? 0.f : __builtin_inff();
        pen.z = (k0 + 2 >= lo && k0 + 2 < hi) ? 0.f : __builtin_inff();
        pen.w = (k0 + 3 >= lo && k0 + 3 < hi) ? 0.f : __builtin_inff();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f2 xj = h ? X.zw : X.xy, yj = h ? Y.zw : Y.xy, zj = h ? Z.zw : Z.xy;
            const f2 pn = h ? pen.zw : pen.xy;
            const f2 dx = PX - xj, dy = PY - yj, dz = PZ - zj;
            f2 d2 = dx * dx + dy * dy + dz * dz;
            d2 = d2 + pn;
            f2 diff = H2 - d2;
            diff.x = fmaxf(diff.x, 0.f);
            diff.y = fmaxf(diff.y, 0.f);
            const f2 t = MS * (DC * diff * diff * diff);
            rho += t.x;
            rho += t.y;
            asm("v_cmp_ngt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2.x), "v"(cut) : "vcc");
            asm("v_cmp_ngt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2.y), "v"(cut) : "vcc");
        }
        if ((i & 7) == 7) { acc ^= m; m = 0; }
        idx += 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = rho + (float)acc;
}

void run_lds_pk(int wavesPerSimd) {
    int blocks = 256 * wavesPerSimd;
    float *out;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_lds_pk<<<blocks, 256>>>(out, 100, 0.01f, 1e3f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_lds_pk<<<blocks, 256>>>(out, iters, 0.01f, 1e3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double tests = (double)iters * 4;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s waves/SIMD %d : %.3f ms, %.1f cycles(@2.4GHz) per wave-test per SIMD\n", "lds-soa-packed", wavesPerSimd, ms,
           cyc / (tests * wavesPerSimd));
    (void)hipFree(out);
}

template <int MODE>
void run(const char *name, int instrPerInner, int wavesPerSimd) {
    int blocksPerCU = wavesPerSimd; // 256 threads = 4 waves = 1 wave per SIMD
    int blocks = 256 * blocksPerCU;
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_valu<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_valu<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instrPerWave = (double)iters * 8 * instrPerInner;
    double wavesPerSimdTotal = (double)blocks * 4 / (256 * 4);
    double cyc = ms * 1e-3 * 2.4e9; // at nominal 2.4 GHz
    printf("%-10s waves/SIMD %d : %.3f ms, %.2f cycles(@2.4GHz)/wave-instr/SIMD\n", name, wavesPerSimd, ms,
           cyc / (instrPerWave * wavesPerSimdTotal));
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) run_lds<0>("lds-b128", w);
    for (int w : {1, 2, 4, 8}) run_lds<1>("lds-b128-sel", w);
    for (int w : {1, 2, 4, 8}) run_lds<2>("lds-b96", w);
    for (int w : {1, 2, 4, 6, 8}) run_lds_pk(w);
    for (int w : {1, 2, 4, 8}) run<0>("mul/add", 8, w);
    for (int w : {1, 2, 4, 8}) run<1>("density", 18, w);
    for (int w : {1, 2, 4, 8}) run<2>("fma", 8, w);
    for (int w : {1, 2, 4, 8}) run<3>("mul/add-v", 8, w);
    for (int w : {1, 2, 4, 8}) run<4>("indep-v", 8, w);
    return 0;
}
