// Are packed fp32 VALU ops (v_pk_mul_f32 / v_pk_add_f32) a throughput win on gfx950?
// Compares 8 independent scalar mul/add chains with 4 packed float2 chains doing
// the same arithmetic.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize pk_rate.hip
// (without -fno-slp-vectorize hipcc packs the "scalar" chains into v_pk_* too)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float av = a, bv = b;
    asm volatile("" : "+v"(av), "+v"(bv));
    if (MODE == 0) {
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = x0 * av + bv; x1 = x1 * av + bv; x2 = x2 * av + bv; x3 = x3 * av + bv;
                x4 = x4 * av + bv; x5 = x5 * av + bv; x6 = x6 * av + bv; x7 = x7 * av + bv;
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else {
        float2v A = {av, av}, B = {bv, bv};
        float2v y0 = {(float)threadIdx.x, 1.f}, y1 = y0 + 2.f, y2 = y0 + 4.f, y3 = y0 + 6.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                y0 = y0 * A + B; y1 = y1 * A + B; y2 = y2 * A + B; y3 = y3 * A + B;
            }
        }
        float2v s = y0 + y1 + y2 + y3;
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    }
}
template <int MODE> void run(const char *name, int wps) {
    int blocks = 256 * wps; float *out; (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 1e-6f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, 20000, 1.0001f, 1e-6f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 256 * 20000.0 * 8 * 16; // 8 unroll x 8 chains x (mul+add)
    printf("%-8s waves/SIMD %d: %.3f ms  %.1f TFLOP/s (unfused mul+add)\n", name, wps, ms, flops / ms * 1e-9);
    (void)hipFree(out);
}
int main() { for (int w : {2, 4, 8}) { run<0>("scalar", w); run<1>("packed", w); } return 0; }
