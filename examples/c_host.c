/* c_host.c -- the drop-in boundary used from plain C (no Python, no torch): fits a small synthetic
 * alignment with plm_fit and prints the strongest long-range coupling, then checks that the planted pair is
 * found.  Build:  gcc -std=c11 -O2 -Iinclude examples/c_host.c -Levcouplings_amd -lplm_hip -Wl,-rpath,$PWD/evcouplings_amd -o c_host
 * This is what a C/C++ host replacing the plmc child process (evcouplings/couplings/tools.py:266) links against. */
#include "plm_hip.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long rng_state = 88172645463325252ull;
static unsigned rnd(void) {   /* xorshift64 */
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (unsigned)(rng_state >> 32);
}

static int on_iteration(int32_t iter, double secs, double cond, double fx, double nll, double norm_h, double norm_e,
                         void *user) {
    (void)secs; (void)nll; (void)norm_h; (void)norm_e;
    if (iter % 10 == 0) fprintf((FILE *)user, "iter %d  fx %.4f  |g|/|x| %.3e\n", iter, fx, cond);
    return 0;   /* non-zero would cancel the fit */
}

int main(void) {
    enum { N = 600, L = 40, Q = 21, PI = 7, PJ = 29 };
    if (plm_version() != PLM_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
    if (plm_device_count() < 1) { fprintf(stderr, "no gfx950 device: %s\n", plm_last_error()); return 3; }
    int8_t *msa = malloc((size_t)N * L);
    for (int s = 0; s < N; s++) {
        for (int i = 0; i < L; i++) msa[s * L + i] = (int8_t)(1 + rnd() % 20);
        if (rnd() % 10 < 8) msa[s * L + PJ] = (int8_t)(1 + (msa[s * L + PI] + 6) % 20);   /* planted coupling */
    }
    plm_problem_t p;
    memset(&p, 0, sizeof p);
    p.n_seqs = N; p.n_sites = L; p.n_states = Q; p.msa = msa;
    p.theta_id = 0.8; p.scale = 1.0; p.lambda_h = 0.01; p.lambda_j = 0.01 * (Q - 1) * (L - 1);
    p.max_iter = 60; p.epsilon = 1e-6; p.lbfgs_m = 0; p.n_shards = 1; p.shard = 0; p.flags = 0;
    plm_result_t r;
    memset(&r, 0, sizeof r);
    r.cn = malloc(sizeof(float) * L * L);
    r.hi = malloc(sizeof(float) * L * Q);
    int rc = plm_fit(&p, &r, 0, NULL, on_iteration, stderr, NULL, NULL);
    if (rc != PLM_OK) { fprintf(stderr, "plm_fit failed (%d): %s\n", rc, plm_last_error()); return 1; }
    int bi = 0, bj = 6;
    for (int i = 0; i < L; i++)
        for (int j = i + 6; j < L; j++)
            if (r.cn[i * L + j] > r.cn[bi * L + bj]) { bi = i; bj = j; }
    printf("iterations %d  evaluations %d  n_eff %.1f  status \"%s\"\n", r.iters_done, r.n_evals, r.n_eff, r.status_msg);
    printf("top long-range pair %d %d  cn %.4f\n", bi, bj, r.cn[bi * L + bj]);
    free(msa); free(r.cn); free(r.hi);
    return (bi == PI && bj == PJ) ? 0 : 4;
}
