// Device helpers shared by the sweep translation units (sweeps.hip,
// sweeps_list.hip): per-pair arithmetic in the reference's operation order, the
// nine-run neighbour ranges, integration and the XCD-aware tile mapping.
// Every TU that includes this MUST be compiled with -ffp-contract=off.
#pragma once
#include "sph_device.h"

#define SW_THREADS 256
#define SW_WAVES (SW_THREADS / SPH_WAVE)
#ifndef SW_UNROLL
#define SW_UNROLL 4 // candidates per trip of a lock-step walk
#endif
#ifndef SW_CAP
#define SW_CAP 384  // staged candidates per run per wave (6 KiB).  Measured over the 100-step headline
                    // run (density sweep, ms): 256: 0.786, 384: 0.718, 512: 0.750, 640: 0.827, 1024: 1.128 --
                    // late in the run 40-60 % of the candidates sit in runs longer than 256 and are walked
                    // from global memory; beyond 384 the lost resident waves cost more than that
#endif
#ifndef SW_QCAP
// hit-FIFO entries per lane (8 KiB per wave).  Measured at n = 4,194,304: 16 entries
// -> 41 drains per wave at 41 % slot efficiency, 32 -> 32 drains at 53 %; force
// sweep 2.40 -> 2.34 ms even though the extra LDS costs a resident wave per SIMD.
#define SW_QCAP 32
#endif

__device__ __forceinline__ int3 sweep_cell(const DevParams &P, float x, float y,
                                           float z) {
    int3 c;
    c.x = min(max((int)(x / P.h), 0), P.D - 1);
    c.y = min(max((int)(y / P.h), 0), P.D - 1);
    c.z = min(max((int)(z / P.h), 0), P.D - 1);
    return c;
}

// The nine runs of one particle: [js[r], je[r]) in sorted-stream indices,
// r = (dz+1)*3 + (dy+1).  Empty runs are {0,0}.
__device__ __forceinline__ void load_runs(const DevParams &P,
                                          const int2 *__restrict__ cellRange,
                                          int3 c, bool valid, int (&js)[9],
                                          int (&je)[9]) {
    const int x0 = max(c.x - 1, 0), x1 = min(c.x + 1, P.D - 1);
    const int xm = min(x0 + 1, x1);
    // All 27 table reads are issued before any is used (rows outside the grid are
    // clamped for the load and discarded afterwards): one L2 round trip per
    // particle instead of nine dependent ones.
    int2 r0[9], r1[9], r2[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int sz = min(max(c.z + (r / 3 - 1), 0), P.D - 1);
        const int sy = min(max(c.y + (r % 3 - 1), 0), P.D - 1);
        const int base = sy * P.D + sz * P.D * P.D;
        r0[r] = cellRange[base + x0];
        r1[r] = cellRange[base + xm];
        r2[r] = cellRange[base + x1];
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int sz = c.z + (r / 3 - 1), sy = c.y + (r % 3 - 1);
        const bool inGrid = valid && sy >= 0 && sy < P.D && sz >= 0 && sz < P.D;
        const bool n0 = r0[r].y > r0[r].x, n1 = r1[r].y > r1[r].x, n2 = r2[r].y > r2[r].x;
        const bool any = inGrid && (n0 | n1 | n2);
        const int s = n0 ? r0[r].x : (n1 ? r1[r].x : r2[r].x);
        const int e = n2 ? r2[r].y : (n1 ? r1[r].y : r0[r].y);
        js[r] = any ? s : 0;
        je[r] = any ? e : 0;
    }
}

// ---- per-pair arithmetic (strict: individually rounded, reference order) ----

// densityKernel (simulator.cu:84-97) folded with `density += MASS * W` (:179).
__device__ __forceinline__ void density_pair(const DevParams &P, float pix, float piy,
                                             float piz, float4 pj, float &rho) {
    float dx = pix - pj.x;
    float dy = piy - pj.y;
    float dz = piz - pj.z;
    float dist2 = dx * dx + dy * dy + dz * dz;
    if (!(dist2 > P.h2)) {
        float diff = P.h2 - dist2;
        rho += SPH_MASS * (P.dcoef * diff * diff * diff);
    }
}

struct ForceAcc {
    float fx, fy, fz;
};

// ---- correctly rounded divide and square root without the range fix-ups ----
// hipcc expands `a / b` to v_div_scale x2, v_rcp, a Newton chain of five fmas, v_div_fmas and
// v_div_fixup (11 instructions), and sqrtf() to a scaled v_sqrt with two one-ulp corrections and
// a class check (17).  The scale / fix-up instructions only act on operands near the ends of the
// exponent range, on denormals, infinities and NaNs.  With the REFERENCE's constants (h = 0.1,
// main.cpp:57-63) the pair body's operands are nowhere near: densities 1e-4..2e9 (at most 31.3 per
// coincident neighbour), distances 1e-4..0.1, numerators 0 or 8e-10..1e8 -- so the bare Newton
// chain, the same fmas in the same order, returns the same correctly rounded bits, and the two
// divisions by rho_j share the refined reciprocal (x / (2 rho) == (x / rho) / 2 exactly).
// SLIM is a template parameter chosen per launch: the host selects it only when the handle's
// settings ARE the reference's (DevParams::slimDiv); any other h / kernel coefficients get the
// compiler's full IEEE expansions (tests: h = 0.01 and h = 1 boxes, dense and coincident states).
// The check path (SPH_SWEEP_DIRECT) and the linked-list backend always use the full expansions, so
// `list == direct` bit for bit is also a test of this argument.
__device__ __forceinline__ float sw_recip_refined(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}
__device__ __forceinline__ float sw_div_with(float a, float b, float r1) { // a / b, r1 = sw_recip_refined(b)
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}
// sqrtf(x) for x = 0 or x >= 2^-96 (a smaller x gives some value far below SPH_EPS_F, which is
// all the callers' `dist < SPH_EPS_F` gate needs)
__device__ __forceinline__ float sw_sqrt(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1);
    const float sp = __int_as_float(__float_as_int(s) + 1);
    const float tm = __builtin_fmaf(-sm, s, x);
    const float tp = __builtin_fmaf(-sp, s, x);
    float r = (0.f >= tm) ? sm : s;
    r = (0.f < tp) ? sp : r;
    return r;
}

// One neighbour of kernelUpdateForces (simulator.cu:223-251) with
// pressureKernel (:99-117) and viscosityKernel (:119-130) inlined.
template <bool SLIM>
__device__ __forceinline__ void force_pair(const DevParams &P, float pix, float piy,
                                           float piz, float vix, float viy, float viz,
                                           float prs_i, float4 pj, float4 vj,
                                           ForceAcc &F) {
    float dx = pix - pj.x;
    float dy = piy - pj.y;
    float dz = piz - pj.z;
    float dist2 = dx * dx + dy * dy + dz * dz;
    float rho_j = vj.w;
    float prs_j = fmaxf(0.f, SPH_GAS_CONSTANT * (rho_j - SPH_REST_DENSITY));
    if constexpr (SLIM) {
        // One basic block: the three Newton chains and the square root's corrections interleave, the
        // gates become selects.  Adding +-0 for a gated-out term is exact: F starts at +0 and an IEEE
        // sum is -0 only if both operands are.  A coincident pair (dist = 0) makes NaNs / infinities
        // in `scale` that the selects discard.
        const float dist = sw_sqrt(dist2);
        const bool tiny = dist < SPH_EPS_F;
        const bool inP = !(dist2 > P.h2) && !tiny, inV = !(dist > P.h) && !tiny;
        const float rrho = sw_recip_refined(rho_j);
        const float hd = P.h - dist;
        const float fPressure = 0.5f * sw_div_with(-SPH_MASS * (prs_i + prs_j), rho_j, rrho);
        const float scale = sw_div_with((-P.vcoef) * hd * hd, dist, sw_recip_refined(dist));
        const float fViscosity = sw_div_with(SPH_VISCOSITY * SPH_MASS * (P.vcoef * hd), rho_j, rrho);
        // the gates act on the two scalar factors (two selects instead of six): a gated-out term is
        // then (finite) * 0 = +-0, which leaves the accumulators unchanged like the skipped addition
        // (dx, dv and fPressure are finite: rho_j >= SPH_EPS_F)
        const float scaleG = inP ? scale : 0.f, fViscG = inV ? fViscosity : 0.f;
        float kx = dx * scaleG, ky = dy * scaleG, kz = dz * scaleG;
        kx *= fPressure;
        ky *= fPressure;
        kz *= fPressure;
        float dvx = vj.x - vix, dvy = vj.y - viy, dvz = vj.z - viz;
        dvx *= fViscG;
        dvy *= fViscG;
        dvz *= fViscG;
        F.fx += kx;
        F.fy += ky;
        F.fz += kz;
        F.fx += dvx;
        F.fy += dvy;
        F.fz += dvz;
    } else {
        float dist = sqrtf(dist2);
        bool tiny = dist < SPH_EPS_F;
        if (!(dist2 > P.h2) && !tiny) {
            float fPressure = -SPH_MASS * (prs_i + prs_j) / (2.f * rho_j);
            float scale = (-P.vcoef) * (P.h - dist) * (P.h - dist) / dist;
            float kx = dx * scale, ky = dy * scale, kz = dz * scale;
            kx *= fPressure;
            ky *= fPressure;
            kz *= fPressure;
            F.fx += kx;
            F.fy += ky;
            F.fz += kz;
        }
        if (!(dist > P.h) && !tiny) {
            float fViscosity =
                SPH_VISCOSITY * SPH_MASS * (P.vcoef * (P.h - dist)) / rho_j;
            float dvx = vj.x - vix, dvy = vj.y - viy, dvz = vj.z - viz;
            dvx *= fViscosity;
            dvy *= fViscosity;
            dvz *= fViscosity;
            F.fx += dvx;
            F.fy += dvy;
            F.fz += dvz;
        }
    }
}

// ---- SPH_MATH_FAST variants: same formulas, FMA-contracted, with the hardware's
// approximate reciprocal / reciprocal square root (~1 ulp) instead of the
// correctly rounded divide and sqrt.  Not bit-identical to the oracle; checked
// against it at the north star's 1e-5 relative tolerance (tests).
__device__ __forceinline__ float fast_dist2(float pix, float piy, float piz, float4 pj,
                                            float &dx, float &dy, float &dz) {
    dx = pix - pj.x;
    dy = piy - pj.y;
    dz = piz - pj.z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ void force_pair_fast(const DevParams &P, float pix, float piy,
                                                float piz, float vix, float viy, float viz,
                                                float prs_i, float4 pj, float4 vj,
                                                ForceAcc &F) {
    float dx, dy, dz;
    const float dist2 = fast_dist2(pix, piy, piz, pj, dx, dy, dz);
    const float rho_j = vj.w;
    const float inv_rho = __builtin_amdgcn_rcpf(rho_j);
    const float prs_j = fmaxf(0.f, rho_j - SPH_REST_DENSITY);
    const float inv_dist = __builtin_amdgcn_rsqf(dist2);
    const float dist = dist2 * inv_dist;
    const bool ok = !(dist2 > P.h2) && !(dist < SPH_EPS_F) && dist2 > 0.f;
    if (ok) {
        const float hd = P.h - dist;
        // fPressure * scale = (-MASS (p_i+p_j) / (2 rho_j)) * (-vcoef (h-r)^2 / r)
        const float s = (0.5f * SPH_MASS * P.vcoef) * (prs_i + prs_j) * inv_rho * (hd * hd) * inv_dist;
        // fViscosity = VISCOSITY MASS vcoef (h-r) / rho_j
        const float fv = (SPH_VISCOSITY * SPH_MASS * P.vcoef) * hd * inv_rho;
        F.fx = __builtin_fmaf(vj.x - vix, fv, __builtin_fmaf(dx, s, F.fx));
        F.fy = __builtin_fmaf(vj.y - viy, fv, __builtin_fmaf(dy, s, F.fy));
        F.fz = __builtin_fmaf(vj.z - viz, fv, __builtin_fmaf(dz, s, F.fz));
    }
}

// kernelUpdatePositions (simulator.cu:258-318).
__device__ __forceinline__ void integrate_particle(const DevParams &P, float4 &p,
                                                   float &vx, float &vy, float &vz,
                                                   const ForceAcc &F, float density) {
    const float timestep = P.dt;
    vx += timestep * F.fx / density;
    vy += timestep * (F.fy / density + SPH_GRAVITY);
    vz += timestep * F.fz / density;

    p.x += timestep * vx;
    p.y += timestep * vy;
    p.z += timestep * vz;

    if (p.x < P.h) { p.x = P.h; vx *= -SPH_ELASTICITY; }
    else if (p.x > P.boxHi) { p.x = P.boxHi; vx *= -SPH_ELASTICITY; }
    if (p.y < P.h) { p.y = P.h; vy *= -SPH_ELASTICITY; }
    else if (p.y > P.boxHi) { p.y = P.boxHi; vy *= -SPH_ELASTICITY; }
    if (p.z < P.h) { p.z = P.h; vz *= -SPH_ELASTICITY; }
    else if (p.z > P.boxHi) { p.z = P.boxHi; vz *= -SPH_ELASTICITY; }

    if (fabsf(vx) < SPH_EPS_F) vx = 0;
    if (fabsf(vy) < SPH_EPS_F) vy = 0;
    if (fabsf(vz) < SPH_EPS_F) vz = 0;
}

__device__ __forceinline__ void store_particle(const SweepArgs &A, int i, float4 p,
                                               float vx, float vy, float vz,
                                               float rho, const ForceAcc &F) {
    A.pos_out[i] = p;
    A.vel_out[i] = make_float4(vx, vy, vz, rho);
    if (A.host_order_pos) {
        // devicePosition[pIdx] = position (simulator.cu:317): original-id order
        uint32_t id = __float_as_uint(p.w);
        float *o = A.host_order_pos + 3 * (size_t)id;
        o[0] = p.x;
        o[1] = p.y;
        o[2] = p.z;
    }
    if (A.force_out) A.force_out[i] = make_float4(F.fx, F.fy, F.fz, 0.f);
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2),
// so consecutive blockIdx values land on different XCDs, and an XCD keeps about
// 170 workgroups (43 K particles) in flight.  A tile of the cell-sorted stream
// gathers from three z-layers (its own and the two next to it), each 1/L of the
// stream.  Two placements (speed only -- any placement gives the same result):
//  * C == 0: every XCD walks ONE contiguous eighth of the stream.  Its resident
//    tiles then span most of a layer and its L2 has to hold three whole layers
//    (5 MB of (pos4, vel4) records at n = 4 M): it does not, and every record is
//    fetched from memory about three times.
//  * C > 0 (tiles per chunk, = 1/8 of a z-layer): the stream is cut into groups
//    of 8 chunks and chunk x of every group goes to XCD x.  All XCDs then move
//    through the layers together, each on its own y-band, and an XCD's working
//    set is that band of the ~9 layers it has in flight (1.8 MB): the records of
//    the layers above and below are still in its L2 when it reaches them.
//  * rot > 0: chunk x of group g goes to XCD (x - g / rot) mod 8 instead of XCD x.  A group is
//    about one z-layer, sorted y-major: once the fluid has reached the floor, the dense rows of
//    EVERY layer sit in its first chunks, and with a fixed chunk -> XCD assignment the same one to
//    four XCDs would carry all of them (they are most of the work late in a run) while the rest
//    idle through the sparse cloud.  Rotating every `rot` layers spreads the floor over all eight
//    and keeps `rot` consecutive layers of a band on one XCD (the L2 reuse above).
__device__ __forceinline__ int xcd_tile(int b, int nb, int C, int rot = 0) {
    int base = 0, R = nb, r = b;
    if (C > 0) {
        const int G = 8 * C;
        const int ng = nb / G, g = b / G;
        if (g < ng) {
            r = b - g * G;
            const int x = rot > 0 ? ((r & 7) + g / rot) & 7 : (r & 7);
            return g * G + x * C + (r >> 3);
        }
        base = ng * G; // the tail (< one group) is dealt in contiguous eighths
        R = nb - base;
        r = b - base;
    }
    const int xcd = r & 7, idx = r >> 3;
    const int q = R >> 3, rem = R & 7;
    return base + xcd * q + min(xcd, rem) + idx;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

