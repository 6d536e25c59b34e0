// SPH_SWEEP_LINKED: the reference's OWN neighbour structure -- one lock-free
// linked list per grid cell, rebuilt every step (insertList simulator.cu:44-55,
// kernelBuildGrid :133-147, list walks :163-189 and :207-251) -- kept as an
// alternative backend so that "the reference's algorithm on MI355X" can be timed
// next to the sorted-stream design.  It is NOT the product
// path and never `value` in bench.py.
//
// What is the same as the reference: no sort; particles stay in particle-id order;
// a cell's list is built by atomic pushes, so the order of a list -- and with it
// the order of every fp32 sum -- depends on which lane wins the atomic: results
// agree with the oracle to rounding (a few ulp per sum), not bit for bit, and
// differ from run to run.  What is not: state is the two float4 streams of the
// rest of the library instead of the 56-byte AoS record, `next` is a 4-byte index
// instead of an 8-byte pointer, the push is one atomic exchange instead of a
// compare-and-swap loop, and the sums are kept in registers instead of being
// re-read from global memory per neighbour (simulator.cu:179,231-250).
#include "sweep_common.h"

#define LK_THREADS 256

__global__ __launch_bounds__(LK_THREADS) void k_link_build(DevParams P, const float4 *__restrict__ pos4,
                                                          int *__restrict__ head, int *__restrict__ next, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos4[i];
    const int3 c = sweep_cell(P, p.x, p.y, p.z);
    const int cell = c.x + c.y * P.D + c.z * P.D * P.D;
    next[i] = atomicExch(&head[cell], i); // LIFO push; the list ends at -1
}

__global__ __launch_bounds__(LK_THREADS) void k_density_linked(DevParams P, SweepArgs A) {
    const int i = A.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.i_end) return;
    const float4 pi = A.pos4[i];
    const int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    float rho = 0.f;
    uint32_t tests = 0;
    for (int z = max(c.z - 1, 0); z <= min(c.z + 1, P.D - 1); ++z)
        for (int y = max(c.y - 1, 0); y <= min(c.y + 1, P.D - 1); ++y)
            for (int x = max(c.x - 1, 0); x <= min(c.x + 1, P.D - 1); ++x)
                for (int j = A.listHead[x + y * P.D + z * P.D * P.D]; j >= 0; j = A.listNext[j]) {
                    density_pair(P, pi.x, pi.y, pi.z, A.pos4[j], rho);
                    ++tests;
                }
    if (A.pairCounter) atomicAdd(A.pairCounter, (unsigned long long)tests);
    A.vel4[i].w = fmaxf(rho, SPH_EPS_F);
}

__global__ __launch_bounds__(LK_THREADS) void k_force_linked(DevParams P, SweepArgs A) {
    const int i = A.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.i_end) return;
    float4 pi = A.pos4[i];
    const float4 vi = A.vel4[i];
    const float prs_i = fmaxf(0.f, SPH_GAS_CONSTANT * (vi.w - SPH_REST_DENSITY));
    const int3 c = sweep_cell(P, pi.x, pi.y, pi.z);
    ForceAcc F = {0.f, 0.f, 0.f};
    for (int z = max(c.z - 1, 0); z <= min(c.z + 1, P.D - 1); ++z)
        for (int y = max(c.y - 1, 0); y <= min(c.y + 1, P.D - 1); ++y)
            for (int x = max(c.x - 1, 0); x <= min(c.x + 1, P.D - 1); ++x)
                for (int j = A.listHead[x + y * P.D + z * P.D * P.D]; j >= 0; j = A.listNext[j])
                    force_pair<false>(P, pi.x, pi.y, pi.z, vi.x, vi.y, vi.z, prs_i, A.pos4[j], A.vel4[j], F);
    float vx = vi.x, vy = vi.y, vz = vi.z;
    integrate_particle(P, pi, vx, vy, vz, F, vi.w);
    store_particle(A, i, pi, vx, vy, vz, vi.w, F);
}

void sph_launch_link_build(const DevParams &P, const float4 *pos4, int *head, int *next, int n,
                           hipStream_t s) {
    if (n <= 0) return;
    k_link_build<<<(n + LK_THREADS - 1) / LK_THREADS, LK_THREADS, 0, s>>>(P, pos4, head, next, n);
}

void sph_launch_density_linked(const DevParams &P, const SweepArgs &A, hipStream_t s) {
    const int cnt = A.i_end - A.i_begin;
    if (cnt <= 0) return;
    k_density_linked<<<(cnt + LK_THREADS - 1) / LK_THREADS, LK_THREADS, 0, s>>>(P, A);
}

void sph_launch_force_linked(const DevParams &P, const SweepArgs &A, hipStream_t s) {
    const int cnt = A.i_end - A.i_begin;
    if (cnt <= 0) return;
    k_force_linked<<<(cnt + LK_THREADS - 1) / LK_THREADS, LK_THREADS, 0, s>>>(P, A);
}
