/*
 * sph_oracle.c -- CPU oracle for the SPH step path.  TEST INFRASTRUCTURE ONLY.
 * See sph_oracle.h for scope, the canonical-order decision and the statement
 * that parity is UNPINNED by any reference test or fixture (there are none).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: every operation below must round on its own.
 * Results do not depend on the OpenMP thread count (each particle's sums are
 * sequential; threads only split the particle range).
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* simulator.h:6-12, simulator.cu:13-14 */
#define O_PI 3.14159265f
#define O_MASS 0.02f
#define O_GAS_CONSTANT 1.f
#define O_REST_DENSITY 1000.f
#define O_VISCOSITY 1.f
#define O_GRAVITY -9.8f
#define O_ELASTICITY 0.5f
#define O_EPS_F (1e-4f)
#define O_PUSH_STRENGTH (5.f)
/* simulator.h:14-17 */
#define O_BOX_MAX_X (600)
#define O_BOX_MIN_X (200)
#define O_BOX_MAX_Y (450)
#define O_BOX_MIN_Y (150)

void oracle_set_num_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* main.cpp:57-63 */
void oracle_make_settings(OracleSettings *s, int numParticles, int randomInit) {
    float h = .1f;
    float h_pow_6 = (float)pow((double)h, 6.0);
    float h_pow_9 = (float)pow((double)h, 9.0);
    float v_kernel_coeff = 45.f / (O_PI * h_pow_6);
    float d_kernel_coeff = 315.f / (64.f * O_PI * h_pow_9);
    memset(s, 0, sizeof(*s));
    s->randomInit = randomInit ? 1 : 0;
    s->numParticles = numParticles;
    s->h = h;
    s->v_kernel_coeff = v_kernel_coeff;
    s->d_kernel_coeff = d_kernel_coeff;
    s->boxDim = 10.f;
    s->numCellsPerDim = 100;
    s->timestep = (float).01;
}

/* simulator.cu:430-453 */
int oracle_init_positions(const OracleSettings *s, float *pos) {
    int n = s->numParticles;
    if (s->randomInit) {
        srand(1); /* == the never-seeded state of a fresh process (glibc) */
        for (int i = 0; i < n; i++) {
            float x = rand() / (float)RAND_MAX * (s->boxDim - 2.f) + 1.f;
            float y = rand() / (float)RAND_MAX * (s->boxDim - 2.f) + 1.f;
            float z = rand() / (float)RAND_MAX * (s->boxDim - 2.f) + 1.f;
            pos[3 * i + 0] = x;
            pos[3 * i + 1] = y;
            pos[3 * i + 2] = z;
        }
        return n;
    }
    float spacing = 0.9f * s->h;
    int nx = (int)(floor((s->boxDim - 2 * s->h) / spacing) + 1);
    int ny = nx, nz = nx;
    int count = 0;
    for (int x = 0; x < nx && count < n; x++) {
        for (int y = 0; y < ny && count < n; y++) {
            for (int z = 0; z < nz && count < n; z++) {
                pos[3 * count + 0] = s->h + spacing * x;
                pos[3 * count + 1] = s->h + spacing * y;
                pos[3 * count + 2] = s->h + spacing * z;
                count++;
            }
        }
    }
    return count; /* < n means the reference leaves the tail uninitialised */
}

/* Extension beyond the reference's domain, labelled as such. */
int oracle_init_positions_dense(const OracleSettings *s, float *pos) {
    int n = s->numParticles;
    int nx = (int)ceil(cbrt((double)n));
    while ((long long)nx * nx * nx < n) nx++;
    while (nx > 1 && (long long)(nx - 1) * (nx - 1) * (nx - 1) >= n) nx--;
    float spacing = nx > 1 ? (s->boxDim - 2 * s->h) / (float)(nx - 1) : 0.f;
    int count = 0;
    for (int x = 0; x < nx && count < n; x++)
        for (int y = 0; y < nx && count < n; y++)
            for (int z = 0; z < nx && count < n; z++) {
                pos[3 * count + 0] = s->h + spacing * x;
                pos[3 * count + 1] = s->h + spacing * y;
                pos[3 * count + 2] = s->h + spacing * z;
                count++;
            }
    return count;
}

/* simulator.cu:57-76 */
static inline void get_grid_cell(const OracleSettings *s, const float *p,
                                 int *cx, int *cy, int *cz) {
    *cx = (int)(p[0] / s->h);
    *cy = (int)(p[1] / s->h);
    *cz = (int)(p[2] / s->h);
}

/* Key function of the neighbour grid (the oracle takes it as a parameter
 * because it changes the tie order of the stable sort).  0 = the reference's flattened
 * index (simulator.cu:78-82); 1 = Morton / z-order (the reference's README.md:5 names a
 * `z_index_sort` branch that is not in the checkout: bits of x, y, z interleaved, x lowest).
 * Process-wide switch: test infrastructure, one simulation at a time. */
static int g_key_order = 0;
void oracle_set_key_order(int order) { g_key_order = order ? 1 : 0; }
int oracle_key_order(void) { return g_key_order; }

static inline uint32_t spread3(uint32_t v) { /* 10 bits -> every third bit */
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

/* number of table entries the key function needs for d cells per axis */
int oracle_num_keys(int d) {
    if (!g_key_order) return d * d * d;
    int b = 0;
    while ((1 << b) < d) b++;
    return 1 << (3 * b);
}

/* simulator.cu:78-82 (evaluated in float there; exact below 2^24) */
static inline int flatten_grid_coord(const OracleSettings *s, int x, int y,
                                     int z) {
    if (g_key_order)
        return (int)(spread3((uint32_t)x) | (spread3((uint32_t)y) << 1) | (spread3((uint32_t)z) << 2));
    return (int)(x + y * s->numCellsPerDim +
                 z * s->numCellsPerDim * s->numCellsPerDim);
}

void oracle_cell_keys(const OracleSettings *s, const float *pos, int n,
                      uint32_t *keys) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        int cx, cy, cz;
        get_grid_cell(s, pos + 3 * i, &cx, &cy, &cz);
        keys[i] = (uint32_t)flatten_grid_coord(s, cx, cy, cz);
    }
}

void oracle_stable_sort(const uint32_t *keys, int n, int numCells,
                        uint32_t *perm) {
    int32_t *count = (int32_t *)calloc((size_t)numCells + 1, sizeof(int32_t));
    for (int i = 0; i < n; i++) count[keys[i] + 1]++;
    for (int c = 0; c < numCells; c++) count[c + 1] += count[c];
    for (int i = 0; i < n; i++) perm[count[keys[i]]++] = (uint32_t)i;
    free(count);
}

void oracle_cell_table(const uint32_t *sk, int n, int numCells,
                       int32_t *cellStart, int32_t *cellEnd) {
    memset(cellStart, 0, sizeof(int32_t) * (size_t)numCells);
    memset(cellEnd, 0, sizeof(int32_t) * (size_t)numCells);
    for (int i = 0; i < n; i++) {
        if (i == 0 || sk[i] != sk[i - 1]) cellStart[sk[i]] = i;
        if (i == n - 1 || sk[i] != sk[i + 1]) cellEnd[sk[i]] = i + 1;
    }
}

/* simulator.cu:84-97 */
static inline float density_kernel(const OracleSettings *s, const float *pi,
                                   const float *pj) {
    float dx = pi[0] - pj[0];
    float dy = pi[1] - pj[1];
    float dz = pi[2] - pj[2];
    float dist2 = dx * dx + dy * dy + dz * dz;
    float h2 = s->h * s->h;
    if (dist2 > h2) return 0.f;
    float diff = h2 - dist2;
    return s->d_kernel_coeff * diff * diff * diff;
}

/* simulator.cu:99-117 */
static inline void pressure_kernel(const OracleSettings *s, const float *pi,
                                   const float *pj, float out[3]) {
    float dx = pi[0] - pj[0];
    float dy = pi[1] - pj[1];
    float dz = pi[2] - pj[2];
    float dist2 = dx * dx + dy * dy + dz * dz;
    out[0] = out[1] = out[2] = 0.f;
    if (dist2 > s->h * s->h) return;
    float dist = sqrtf(dist2);
    if (dist < O_EPS_F) return;
    float scale =
        (-s->v_kernel_coeff) * (s->h - dist) * (s->h - dist) / dist;
    out[0] = dx * scale;
    out[1] = dy * scale;
    out[2] = dz * scale;
}

/* simulator.cu:119-130 */
static inline float viscosity_kernel(const OracleSettings *s, const float *pi,
                                     const float *pj) {
    float dx = pi[0] - pj[0];
    float dy = pi[1] - pj[1];
    float dz = pi[2] - pj[2];
    float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    if ((dist > s->h) || (dist < O_EPS_F)) return 0.f;
    return s->v_kernel_coeff * (s->h - dist);
}

/* simulator.cu:149-190 */
void oracle_density(const OracleSettings *s, const float *pos, int n_all,
                    const int32_t *cellStart, const int32_t *cellEnd,
                    int i_begin, int i_end, float *rho, float *prs) {
    (void)n_all;
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = i_begin; i < i_end; i++) {
        const float *pi = pos + 3 * i;
        int cx, cy, cz;
        get_grid_cell(s, pi, &cx, &cy, &cz);
        float density = 0.f;
        for (int dz = -1; dz < 2; dz++) {
            int sz = cz + dz;
            if (sz < 0 || sz >= s->numCellsPerDim) continue;
            for (int dy = -1; dy < 2; dy++) {
                int sy = cy + dy;
                if (sy < 0 || sy >= s->numCellsPerDim) continue;
                for (int dx = -1; dx < 2; dx++) {
                    int sx = cx + dx;
                    if (sx < 0 || sx >= s->numCellsPerDim) continue;
                    int c = flatten_grid_coord(s, sx, sy, sz);
                    for (int j = cellStart[c]; j < cellEnd[c]; j++) {
                        density += O_MASS * density_kernel(s, pi, pos + 3 * j);
                    }
                }
            }
        }
        density = fmaxf(density, O_EPS_F);
        rho[i] = density;
        prs[i] = fmaxf(0.f, O_GAS_CONSTANT * (density - O_REST_DENSITY));
    }
}

/* simulator.cu:192-256 */
void oracle_force(const OracleSettings *s, const float *pos, const float *vel,
                  const float *rho, const float *prs, int n_all,
                  const int32_t *cellStart, const int32_t *cellEnd,
                  int i_begin, int i_end, float *force) {
    (void)n_all;
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = i_begin; i < i_end; i++) {
        const float *pi = pos + 3 * i;
        int cx, cy, cz;
        get_grid_cell(s, pi, &cx, &cy, &cz);
        float fx = 0.f, fy = 0.f, fz = 0.f;
        for (int dz = -1; dz < 2; dz++) {
            int sz = cz + dz;
            if (sz < 0 || sz >= s->numCellsPerDim) continue;
            for (int dy = -1; dy < 2; dy++) {
                int sy = cy + dy;
                if (sy < 0 || sy >= s->numCellsPerDim) continue;
                for (int dx = -1; dx < 2; dx++) {
                    int sx = cx + dx;
                    if (sx < 0 || sx >= s->numCellsPerDim) continue;
                    int c = flatten_grid_coord(s, sx, sy, sz);
                    for (int j = cellStart[c]; j < cellEnd[c]; j++) {
                        const float *pj = pos + 3 * j;
                        float fPressure =
                            -O_MASS * (prs[i] + prs[j]) / (2.f * rho[j]);
                        float k1[3];
                        pressure_kernel(s, pi, pj, k1);
                        k1[0] *= fPressure;
                        k1[1] *= fPressure;
                        k1[2] *= fPressure;
                        fx += k1[0];
                        fy += k1[1];
                        fz += k1[2];

                        float dvx = vel[3 * j + 0] - vel[3 * i + 0];
                        float dvy = vel[3 * j + 1] - vel[3 * i + 1];
                        float dvz = vel[3 * j + 2] - vel[3 * i + 2];
                        float fViscosity = O_VISCOSITY * O_MASS *
                                           viscosity_kernel(s, pi, pj) /
                                           rho[j];
                        dvx *= fViscosity;
                        dvy *= fViscosity;
                        dvz *= fViscosity;
                        fx += dvx;
                        fy += dvy;
                        fz += dvz;
                    }
                }
            }
        }
        force[3 * i + 0] = fx;
        force[3 * i + 1] = fy;
        force[3 * i + 2] = fz;
    }
}

/* simulator.cu:258-318 */
void oracle_integrate(const OracleSettings *s, float *pos, float *vel,
                      const float *force, const float *rho, int i_begin,
                      int i_end) {
#pragma omp parallel for schedule(static)
    for (int i = i_begin; i < i_end; i++) {
        float timestep = s->timestep;
        float *p = pos + 3 * i;
        float *v = vel + 3 * i;
        const float *f = force + 3 * i;
        float density = rho[i];

        v[0] += timestep * f[0] / density;
        v[1] += timestep * (f[1] / density + O_GRAVITY);
        v[2] += timestep * f[2] / density;

        p[0] += timestep * v[0];
        p[1] += timestep * v[1];
        p[2] += timestep * v[2];

        for (int a = 0; a < 3; a++) {
            if (p[a] < s->h) {
                p[a] = s->h;
                v[a] *= -O_ELASTICITY;
            } else if (p[a] > s->boxDim - s->h) {
                p[a] = s->boxDim - s->h;
                v[a] *= -O_ELASTICITY;
            }
        }
        for (int a = 0; a < 3; a++) {
            if (fabsf(v[a]) < O_EPS_F) v[a] = 0;
        }
    }
}

uint64_t oracle_pair_tests(const OracleSettings *s, const float *pos, int n_all,
                           const int32_t *cellStart, const int32_t *cellEnd,
                           int i_begin, int i_end) {
    (void)n_all;
    uint64_t total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int i = i_begin; i < i_end; i++) {
        int cx, cy, cz;
        get_grid_cell(s, pos + 3 * i, &cx, &cy, &cz);
        for (int dz = -1; dz < 2; dz++) {
            int sz = cz + dz;
            if (sz < 0 || sz >= s->numCellsPerDim) continue;
            for (int dy = -1; dy < 2; dy++) {
                int sy = cy + dy;
                if (sy < 0 || sy >= s->numCellsPerDim) continue;
                for (int dx = -1; dx < 2; dx++) {
                    int sx = cx + dx;
                    if (sx < 0 || sx >= s->numCellsPerDim) continue;
                    int c = flatten_grid_coord(s, sx, sy, sz);
                    total += (uint64_t)(cellEnd[c] - cellStart[c]);
                }
            }
        }
    }
    return total;
}

/* ------------------------------------------------------------------------ */
struct OracleSim {
    OracleSettings s;
    int n, numCells;
    /* key-sorted state (order of the last step's grid build) */
    float *pos, *vel, *rho, *prs, *force;
    uint32_t *id, *keys;
    int32_t *cellStart, *cellEnd;
    /* scratch */
    float *tpos, *tvel;
    uint32_t *tid, *tkeys, *perm;
    uint64_t lastPairs;
};

OracleSim *oracle_sim_create(const OracleSettings *s) {
    OracleSim *m = (OracleSim *)calloc(1, sizeof(OracleSim));
    m->s = *s;
    m->n = s->numParticles;
    int d = (int)s->numCellsPerDim;
    m->numCells = oracle_num_keys(d);
    size_t n = (size_t)(m->n > 0 ? m->n : 1);
    m->pos = (float *)calloc(3 * n, sizeof(float));
    m->vel = (float *)calloc(3 * n, sizeof(float));
    m->force = (float *)calloc(3 * n, sizeof(float));
    m->rho = (float *)calloc(n, sizeof(float));
    m->prs = (float *)calloc(n, sizeof(float));
    m->id = (uint32_t *)calloc(n, sizeof(uint32_t));
    m->keys = (uint32_t *)calloc(n, sizeof(uint32_t));
    m->tpos = (float *)calloc(3 * n, sizeof(float));
    m->tvel = (float *)calloc(3 * n, sizeof(float));
    m->tid = (uint32_t *)calloc(n, sizeof(uint32_t));
    m->tkeys = (uint32_t *)calloc(n, sizeof(uint32_t));
    m->perm = (uint32_t *)calloc(n, sizeof(uint32_t));
    m->cellStart = (int32_t *)calloc((size_t)m->numCells, sizeof(int32_t));
    m->cellEnd = (int32_t *)calloc((size_t)m->numCells, sizeof(int32_t));
    for (int i = 0; i < m->n; i++) m->id[i] = (uint32_t)i;
    return m;
}

void oracle_sim_destroy(OracleSim *m) {
    if (!m) return;
    free(m->pos); free(m->vel); free(m->force); free(m->rho); free(m->prs);
    free(m->id); free(m->keys); free(m->tpos); free(m->tvel); free(m->tid);
    free(m->tkeys); free(m->perm); free(m->cellStart); free(m->cellEnd);
    free(m);
}

void oracle_sim_setup(OracleSim *m) {
    int written = oracle_init_positions(&m->s, m->pos);
    if (written < m->n) {
        /* n > 109^3 in grid mode is outside the reference's domain
         * (simulator.cu:425-452 leaves the tail uninitialised): extension. */
        oracle_init_positions_dense(&m->s, m->pos);
    }
    memset(m->vel, 0, sizeof(float) * 3 * (size_t)m->n);
    /* a fresh particle array: simulator.cu:414-422 cudaMemset()s it, density and pressure are 0 until a step */
    memset(m->rho, 0, sizeof(float) * (size_t)m->n);
    memset(m->prs, 0, sizeof(float) * (size_t)m->n);
    for (int i = 0; i < m->n; i++) m->id[i] = (uint32_t)i;
}

void oracle_sim_upload(OracleSim *m, const float *pos, const float *vel) {
    memcpy(m->pos, pos, sizeof(float) * 3 * (size_t)m->n);
    if (vel) memcpy(m->vel, vel, sizeof(float) * 3 * (size_t)m->n);
    else memset(m->vel, 0, sizeof(float) * 3 * (size_t)m->n);
    memset(m->rho, 0, sizeof(float) * (size_t)m->n); /* (a new state, like setup()) */
    memset(m->prs, 0, sizeof(float) * (size_t)m->n);
    for (int i = 0; i < m->n; i++) m->id[i] = (uint32_t)i;
}

void oracle_sim_step(OracleSim *m) {
    int n = m->n;
    /* grid build: stable re-sort of the previous order by cell key */
    oracle_cell_keys(&m->s, m->pos, n, m->tkeys);
    oracle_stable_sort(m->tkeys, n, m->numCells, m->perm);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        uint32_t src = m->perm[i];
        m->keys[i] = m->tkeys[src];
        m->tid[i] = m->id[src];
        for (int a = 0; a < 3; a++) {
            m->tpos[3 * i + a] = m->pos[3 * src + a];
            m->tvel[3 * i + a] = m->vel[3 * src + a];
        }
    }
    { float *t = m->pos; m->pos = m->tpos; m->tpos = t; }
    { float *t = m->vel; m->vel = m->tvel; m->tvel = t; }
    { uint32_t *t = m->id; m->id = m->tid; m->tid = t; }
    oracle_cell_table(m->keys, n, m->numCells, m->cellStart, m->cellEnd);
    m->lastPairs = oracle_pair_tests(&m->s, m->pos, n, m->cellStart,
                                     m->cellEnd, 0, n);
    oracle_density(&m->s, m->pos, n, m->cellStart, m->cellEnd, 0, n, m->rho,
                   m->prs);
    oracle_force(&m->s, m->pos, m->vel, m->rho, m->prs, n, m->cellStart,
                 m->cellEnd, 0, n, m->force);
    oracle_integrate(&m->s, m->pos, m->vel, m->force, m->rho, 0, n);
}

/* simulator.cu:329-367; launched <<<1, numCellsPerDim>>> (:483-486): thread t
 * owns "z-layer" (int)((float)t*h / h).  Threads are applied in ascending t;
 * should two t map to one layer the reference has an unsynchronised RMW race
 * there and this order is one legal outcome. */
void oracle_sim_click(OracleSim *m, int mx, int my) {
    const OracleSettings *s = &m->s;
    int nthreads = (int)s->numCellsPerDim;
    for (int t = 0; t < nthreads; t++) {
        float x = ((float)(mx - O_BOX_MIN_X) /
                   (float)(O_BOX_MAX_X - O_BOX_MIN_X)) * s->boxDim;
        float y = ((float)(my - O_BOX_MIN_Y) /
                   (float)(O_BOX_MAX_Y - O_BOX_MIN_Y)) * s->boxDim;
        float z = (float)t * s->h;
        float p[3] = {x, y, z};
        int cx, cy, cz;
        get_grid_cell(s, p, &cx, &cy, &cz);
        cy = (int)(s->numCellsPerDim - cy);
        if (cz < 0 || cz >= s->numCellsPerDim) continue; /* would be OOB */
        for (int dy = -2; dy < 3; dy++) {
            int sy = cy + dy;
            if (sy < 0 || sy >= s->numCellsPerDim) continue;
            for (int dx = -2; dx < 3; dx++) {
                int sx = cx + dx;
                if (sx < 0 || sx >= s->numCellsPerDim) continue;
                int c = flatten_grid_coord(s, sx, sy, cz);
                for (int j = m->cellStart[c]; j < m->cellEnd[c]; j++) {
                    if (dx != 0) m->vel[3 * j + 0] += (1.f / dx) * O_PUSH_STRENGTH;
                    if (dy != 0) m->vel[3 * j + 1] += (1.f / dy) * O_PUSH_STRENGTH;
                    if (dx == 0 && dy == 0) m->vel[3 * j + 2] -= O_PUSH_STRENGTH;
                }
            }
        }
    }
}

void oracle_sim_download(const OracleSim *m, float *pos, float *vel, float *rho,
                         float *prs, float *force) {
    for (int i = 0; i < m->n; i++) {
        uint32_t id = m->id[i];
        for (int a = 0; a < 3; a++) {
            if (pos) pos[3 * id + a] = m->pos[3 * i + a];
            if (vel) vel[3 * id + a] = m->vel[3 * i + a];
            if (force) force[3 * id + a] = m->force[3 * i + a];
        }
        if (rho) rho[id] = m->rho[i];
        if (prs) prs[id] = m->prs[i];
    }
}

int oracle_sim_sorted(const OracleSim *m, uint32_t *ids, uint32_t *keys,
                      float *pos, float *vel) {
    size_t n = (size_t)m->n;
    if (ids) memcpy(ids, m->id, n * sizeof(uint32_t));
    if (keys) memcpy(keys, m->keys, n * sizeof(uint32_t));
    if (pos) memcpy(pos, m->pos, 3 * n * sizeof(float));
    if (vel) memcpy(vel, m->vel, 3 * n * sizeof(float));
    return m->n;
}

uint64_t oracle_sim_last_pair_tests(const OracleSim *m) { return m->lastPairs; }
