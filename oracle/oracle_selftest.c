/* Sanitizer self-test of the CPU oracle (test infrastructure; SURVEY.md section 5: "run the CPU oracle under
 * -fsanitize=address,undefined").  Built by `make -C oracle sanitize` together with sph_oracle.c, both with the
 * sanitizers on; drives every entry point the tests use on small, nasty inputs -- particles on the box faces and in
 * the corner cells (the 27-cell walk at the grid's edge), coincident particles, one crowded cell, clicks at the
 * window's corners, both key orders, both initialisers, n = 1 -- and exits 0 if nothing was reported. */
#include "sph_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static void run(int n, int randomInit, int order, unsigned seed) {
    OracleSettings s;
    oracle_make_settings(&s, n, randomInit);
    oracle_set_key_order(order);
    OracleSim *m = oracle_sim_create(&s);
    oracle_sim_setup(m);
    for (int k = 0; k < 2; ++k) oracle_sim_step(m);
    float *pos = malloc(sizeof(float) * 3 * (size_t)n), *vel = malloc(sizeof(float) * 3 * (size_t)n);
    float *rho = malloc(sizeof(float) * (size_t)n), *prs = malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const unsigned kind = lcg(&seed) % 6;
        for (int a = 0; a < 3; ++a) {
            float v = 0.1f + 9.8f * (float)(lcg(&seed) % 100000) / 100000.f;
            if (kind == 0) v = (lcg(&seed) & 1) ? 0.1f : 9.9f;          /* box faces / corner cells */
            if (kind == 1) v = 5.0f;                                     /* coincident */
            if (kind == 2) v = 3.0f + 0.09f * (float)(lcg(&seed) % 1000) / 1000.f; /* one crowded cell */
            pos[3 * i + a] = v;
            vel[3 * i + a] = (float)((int)(lcg(&seed) % 2001) - 1000) / 100.f;
        }
    }
    oracle_sim_upload(m, pos, vel);
    for (int k = 0; k < 4; ++k) {
        oracle_sim_step(m);
        if (k == 1) oracle_sim_click(m, 0, 0);
        if (k == 2) oracle_sim_click(m, 799, 599);
    }
    oracle_sim_download(m, pos, vel, rho, prs, NULL);
    uint32_t *ids = malloc(sizeof(uint32_t) * (size_t)n), *keys = malloc(sizeof(uint32_t) * (size_t)n);
    (void)oracle_sim_sorted(m, ids, keys, NULL, NULL);
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += pos[3 * i] + rho[i];
    printf("n=%d init=%s key=%s: ok (checksum %.6g, %llu pair tests)\n", n, randomInit ? "random" : "grid",
           order ? "morton" : "flattened", sum, (unsigned long long)oracle_sim_last_pair_tests(m));
    free(pos); free(vel); free(rho); free(prs); free(ids); free(keys);
    oracle_sim_destroy(m);
}

int main(void) {
    oracle_set_num_threads(2);
    run(1, 0, 0, 1u);
    run(2, 1, 0, 2u);
    run(777, 0, 0, 3u);
    run(5000, 1, 0, 4u);
    run(3000, 1, 1, 5u);
    oracle_set_key_order(0);
    puts("oracle self-test: done");
    return 0;
}
