/*
 * sph_oracle.h -- CPU oracle for the SPH step path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch, strict-IEEE-fp32 restatement of the algorithm in the
 * reference's src/simulator.cu (cell hash :57-82, smoothing kernels :84-130,
 * density/pressure :149-190, forces :192-256, integration :258-318, initial
 * conditions :430-453) and of the constants in src/main.cpp:57-63 and
 * src/simulator.h:6-12.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (libsph_hip.so, ./sph) never does.
 *
 * PARITY UNPINNED BY REFERENCE TESTS: the reference ships no tests, fixtures or
 * golden vectors for this path and its CUDA source cannot be compiled or run
 * here (no nvcc, no NVIDIA device).  The oracle is pinned only by hand-derived
 * known-answer values (tests/test_oracle_known_answers.py).
 *
 * The reference's neighbour lists are built by an atomicCAS race
 * (simulator.cu:44-55), so its summation order -- and therefore its last-ulp
 * result -- is not defined.  The oracle fixes ONE legal order ("canonical
 * order"): particles are kept sorted by flattened cell index with a STABLE
 * sort of the previous step's order (step 0: particle-id order); cells are
 * visited z-outer / y / x-inner exactly as simulator.cu:163-176, and within a
 * cell in ascending sorted position.  Every arithmetic operation is rounded
 * individually (compile with -ffp-contract=off), in the order the source
 * expression is written.
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout-identical to the reference's struct Settings (simulator.h:19-31). */
typedef struct OracleSettings {
    uint8_t randomInit; /* bool */
    uint8_t pad_[3];
    int32_t numParticles;
    float h;
    float v_kernel_coeff;
    float d_kernel_coeff;
    float boxDim;
    float numCellsPerDim; /* a float in the reference too */
    float timestep;
} OracleSettings;

/* main.cpp:57-63: h, pow(h,6), pow(h,9), the two kernel coefficients, box. */
void oracle_make_settings(OracleSettings *s, int numParticles, int randomInit);

/* simulator.cu:430-453.  pos_xyz: n*3 floats.  Random mode re-seeds with
 * srand(1), which is glibc's state in a fresh process (the reference never
 * seeds).  Grid mode fills at most 109^3 lattice points; the rest (outside the
 * reference's defined domain) are set to the "dense lattice" extension only if
 * n > 109^3 -- see oracle_init_positions_dense. Returns #points written. */
int oracle_init_positions(const OracleSettings *s, float *pos_xyz);

/* Documented extension for n > 1,295,029 in grid mode:
 * nx = ceil(cbrt(n)), spacing = (boxDim - 2h)/(nx-1), same x-major fill. */
int oracle_init_positions_dense(const OracleSettings *s, float *pos_xyz);

/* simulator.cu:57-82: cell = (int)(p/h) per axis; key = x + y*100 + z*100*100. */
void oracle_cell_keys(const OracleSettings *s, const float *pos_xyz, int n,
                      uint32_t *keys);

/* Stable counting sort by key: perm[i] = source index of the i-th element. */
void oracle_stable_sort(const uint32_t *keys, int n, int numCells,
                        uint32_t *perm);

/* cellStart/cellEnd (numCells entries each) over keys already sorted. */
void oracle_cell_table(const uint32_t *sorted_keys, int n, int numCells,
                       int32_t *cellStart, int32_t *cellEnd);

/* Sweeps over a key-sorted array of n_all particles; results are produced for
 * i in [i_begin, i_end) only (the "owned" range of a z-slab; the whole array
 * for a single domain).  pos/vel are xyz-interleaved. */
void oracle_density(const OracleSettings *s, const float *pos, int n_all,
                    const int32_t *cellStart, const int32_t *cellEnd,
                    int i_begin, int i_end, float *rho, float *prs);

void oracle_force(const OracleSettings *s, const float *pos, const float *vel,
                  const float *rho, const float *prs, int n_all,
                  const int32_t *cellStart, const int32_t *cellEnd,
                  int i_begin, int i_end, float *force);

/* simulator.cu:258-318, in place on pos/vel for i in [i_begin, i_end). */
void oracle_integrate(const OracleSettings *s, float *pos, float *vel,
                      const float *force, const float *rho, int i_begin,
                      int i_end);

/* Number of candidate pair tests of one 27-cell sweep (SURVEY.md 8d "P"). */
uint64_t oracle_pair_tests(const OracleSettings *s, const float *pos, int n_all,
                           const int32_t *cellStart, const int32_t *cellEnd,
                           int i_begin, int i_end);

/* ---- whole-simulation object (single domain) ---- */
typedef struct OracleSim OracleSim;

OracleSim *oracle_sim_create(const OracleSettings *s);
void oracle_sim_destroy(OracleSim *sim);
/* Reference initial conditions (simulator.cu:411-460). */
void oracle_sim_setup(OracleSim *sim);
/* Arbitrary initial state, particle-id order (tests / dense goldens). */
void oracle_sim_upload(OracleSim *sim, const float *pos_xyz,
                       const float *vel_xyz);
/* One simulate() step (simulator.cu:462-497 without the click impulse). */
void oracle_sim_step(OracleSim *sim);
/* Click impulse (simulator.cu:329-367) applied to the current state using the
 * cell table of the last step's (pre-integration) grid, as the reference does. */
void oracle_sim_click(OracleSim *sim, int mouse_x, int mouse_y);
/* Outputs in ORIGINAL particle-id order (simulator.cu:317). Any may be NULL.
 * rho/prs/force are the values computed during the last step. */
void oracle_sim_download(const OracleSim *sim, float *pos_xyz, float *vel_xyz,
                         float *rho, float *prs, float *force_xyz);
/* Sorted-order views of the last step (for slab / kernel-level checks). */
int oracle_sim_sorted(const OracleSim *sim, uint32_t *ids, uint32_t *keys,
                      float *pos_xyz, float *vel_xyz);
uint64_t oracle_sim_last_pair_tests(const OracleSim *sim);
int oracle_num_threads(void);
void oracle_set_num_threads(int t); /* OpenMP threads for the sweeps (results do not change) */

/* Key function of the neighbour grid: 0 flattened (reference), 1 Morton.  Set BEFORE
 * oracle_sim_create; process-wide (test infrastructure). */
void oracle_set_key_order(int order);
int oracle_key_order(void);
int oracle_num_keys(int cells_per_dim);

#ifdef __cplusplus
}
#endif
#endif
