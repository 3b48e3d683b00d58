/*
 * plm_oracle.c -- CPU restatement of the pseudo-likelihood Potts inference path
 * that EVcouplings delegates to the external `plmc` binary.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, link or call it.  The product path
 * (evcouplings_amd/ + libplm_hip.so) never does.
 *
 * PARITY STATUS
 *   pinned (against the reference's own code, see tests/golden/make_golden.py):
 *     - sequence reweighting      <- evcouplings/align/alignment.py:1193-1233
 *     - single / pair frequencies <- evcouplings/align/alignment.py:1079-1153
 *     - zero-sum gauge, FN, APC   <- evcouplings/couplings/model.py:179-233, 744-827
 *     - statistical energies, single-mutant matrix (rows N2)
 *                                 <- evcouplings/couplings/model.py:25-109 (tests/golden/energies_L12.npz, exact)
 *     (mean-field DCA, row N4, is restated in oracle/meanfield_ref.py and pinned by
 *      tests/golden/meanfield_{a,d}.npz <- evcouplings/couplings/mean_field.py:717-1014)
 *   PARITY UNPINNED (no reference code, test or golden vector exists for it; the
 *   arithmetic lives in the un-vendored, un-versioned third-party `plmc` program,
 *   reached only through evcouplings/couplings/tools.py:202-266):
 *     - the symmetric L2-regularised PLM objective + gradient
 *     - the L-BFGS driver
 *   For those two the oracle is pinned only by first principles: brute-force
 *   enumeration, finite differences and convexity (tests/test_oracle.py) -- with one
 *   exception the reference does provide: the scaling convention of the field part
 *   (unnormalised N_eff-weighted log-likelihood + lambda_h |h|^2, no 1/2) is the one
 *   CouplingsModel.to_independent_model minimises (couplings/model.py:894-910); its
 *   optimum is checked to be a stationary point of this objective at J = 0
 *   (tests/golden/independent_model_a.npz).
 *
 * Build:  gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC            -> double (parity)
 *         gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC -DPLMO_F32 -> float  (timed
 *         "plmc-equivalent OpenMP restatement")
 *
 * Parameter vector layout (same as the product's, and the order of the `.model`
 * file, evcouplings/couplings/model.py:355-389):
 *     x = [ h_i(a) : i<L, a<q ]  ++  [ J_ij(a,b) : i<j row-major pairs, a<q, b<q ]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef PLMO_F32
typedef float real;
#define PLMO_NAME(n) plmo32_##n
#else
typedef double real;
#define PLMO_NAME(n) plmo_##n
#endif

#define PLMO_OK 0
#define PLMO_EINVAL -1
#define PLMO_ENOMEM -2
#define PLMO_ELINESEARCH -3

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static inline size_t pair_index(int i, int j, int L) { /* i < j */
    return (size_t)i * (size_t)(2 * L - i - 1) / 2 + (size_t)(j - i - 1);
}

int PLMO_NAME(sizeof_real)(void) { return (int)sizeof(real); }

void PLMO_NAME(set_num_threads)(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int PLMO_NAME(num_threads)(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Integer identity threshold, SURVEY.md App. C.1 / D-1: ident >= ceil(theta*L - 1e-9).
 * Equal to the in-repo rule `pair_id / L >= identity_threshold`
 * (align/alignment.py:1229) for every L the tests sweep. */
/* Convention switches (the PLM_CONV_* bits of include/plm_hip.h; same numbers): conventions of plmc that cannot be
 * verified here are selectable, identically in the oracle and in the HIP library.  Process-global (the oracle is
 * test infrastructure): 32 float32 threshold, 64 -g gap-gap identities, 128 -g threshold on the jointly ungapped
 * length, 256 -g frequencies normalised by N_eff, 512 Frobenius norm without the gap state. */
static int g_conv = 0;
void PLMO_NAME(set_conventions)(int conv) { g_conv = conv; }
/* Group regulariser (run_plmc's lambda_g -> plmc -lg, evcouplings/couplings/tools.py:252-253): process-global like the
 * conventions, 0 = off.  PARITY UNPINNED: plmc's implementation cannot be read here; the term restated is the group
 * lasso its option text names, smoothed at the origin so that L-BFGS can run on it:
 *     R_g = lambda_g * sum_{i<j} sqrt(|J_ij|_F^2 + delta^2),  delta = 1e-4      (PLM_GROUP_DELTA in include/plm_hip.h)
 * dR_g/dJ_ij(a,b) = lambda_g J_ij(a,b) / sqrt(|J_ij|_F^2 + delta^2). */
static double g_lambda_group = 0.0;
#define PLMO_GROUP_DELTA 1e-4
void PLMO_NAME(set_lambda_group)(double lg) { g_lambda_group = lg; }
/* adds the group term of every pair block to the gradient (blocks of qm x qm at J, gradient at gj) and returns its value */
static double group_term(const real *J, real *gj, int L, int qm) {
    if (!(g_lambda_group > 0)) return 0.0;
    const size_t qq = (size_t)qm * qm;
    double tot = 0;
#pragma omp parallel for schedule(static) reduction(+ : tot)
    for (long p = 0; p < (long)L * (L - 1) / 2; p++) {
        double n2 = 0;
        for (size_t k = 0; k < qq; k++) n2 += (double)J[(size_t)p * qq + k] * (double)J[(size_t)p * qq + k];
        const double nrm = sqrt(n2 + PLMO_GROUP_DELTA * PLMO_GROUP_DELTA);
        tot += g_lambda_group * nrm;
        for (size_t k = 0; k < qq; k++) gj[(size_t)p * qq + k] += (real)(g_lambda_group * (double)J[(size_t)p * qq + k] / nrm);
    }
    return tot;
}
int PLMO_NAME(get_conventions)(void) { return g_conv; }

int PLMO_NAME(threshold)(int L, double theta_id) {
    if (g_conv & 32) {   /* what a float32 plmc makes of `-t 1-theta` (tools.py:236-239) */
        const float t = (float)(1.0 - theta_id);
        const float need = (1.0f - t) * (float)L;
        return (int)ceilf(need);
    }
    return (int)ceil(theta_id * (double)L - 1e-9);
}

/* align/alignment.py:1193-1233 (num_cluster_members): counts[s] = number of
 * sequences t (self included) with at least T identical positions. */
int PLMO_NAME(reweight)(const int8_t *msa, int N, int L, double theta_id, int32_t *counts) {
    if (!msa || !counts || N <= 0 || L <= 0) return PLMO_EINVAL;
    const int T = PLMO_NAME(threshold)(L, theta_id);
#pragma omp parallel for schedule(dynamic, 16)
    for (int s = 0; s < N; s++) {
        const int8_t *a = msa + (size_t)s * L;
        int c = 0;
        for (int t = 0; t < N; t++) {
            const int8_t *b = msa + (size_t)t * L;
            int id = 0;
            for (int k = 0; k < L; k++) id += (a[k] == b[k]);
            c += (id >= T);
        }
        counts[s] = c;
    }
    return PLMO_OK;
}

/* align/alignment.py:1079-1106 and 1110-1153: weighted single / pair frequencies,
 * normalised by sum of weights.  fij holds only i<j blocks, [a][b], a at site i. */
int PLMO_NAME(marginals)(const int8_t *msa, const real *w, int N, int L, int q,
                         real *fi, real *fij) {
    if (!msa || !w || !fi || N <= 0 || L <= 0 || q <= 0) return PLMO_EINVAL;
    double neff = 0;
    for (int s = 0; s < N; s++) neff += w[s];
    const size_t qq = (size_t)q * q;
    memset(fi, 0, sizeof(real) * (size_t)L * q);
    for (int s = 0; s < N; s++)
        for (int i = 0; i < L; i++) fi[(size_t)i * q + msa[(size_t)s * L + i]] += w[s];
    for (size_t k = 0; k < (size_t)L * q; k++) fi[k] = (real)(fi[k] / neff);
    if (fij) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int i = 0; i < L - 1; i++) {
            real *blk = fij + pair_index(i, i + 1, L) * qq;
            memset(blk, 0, sizeof(real) * qq * (size_t)(L - 1 - i));
            for (int s = 0; s < N; s++) {
                const int8_t *row = msa + (size_t)s * L;
                const int a = row[i];
                for (int j = i + 1; j < L; j++) blk[(size_t)(j - i - 1) * qq + (size_t)a * q + row[j]] += w[s];
            }
            for (size_t k = 0; k < qq * (size_t)(L - 1 - i); k++) blk[k] = (real)(blk[k] / neff);
        }
    }
    return PLMO_OK;
}

/* Symmetric L2-regularised negative log pseudo-likelihood and its gradient,
 * SURVEY.md section 8 row a6 / App. C.3 (PARITY UNPINNED, see header).
 *   fx  = -sum_s w_s sum_i log P(x_si | x_s,-i) + lh |h|^2 + lj sum_{i<j} |J_ij|^2
 *   P_si(a) ~ exp(h_i(a) + sum_{j!=i} J_ij(a, x_sj))
 * nll_out receives the unregularised part. */
int PLMO_NAME(eval)(const int8_t *msa, const real *w, int N, int L, int q, double lambda_h,
                    double lambda_j, const real *x, real *g, double *fx_out, double *nll_out) {
    if (!msa || !w || !x || !g || N <= 0 || L <= 1 || q <= 1 || q > 64) return PLMO_EINVAL;
    const size_t qq = (size_t)q * q;
    const size_t nh = (size_t)L * q;
    const size_t npair = (size_t)L * (L - 1) / 2;
    const real *J = x + nh;
    /* asymmetric accumulator: slab[i][j][a][b'] (a at i, b' at j) */
    real *slab = (real *)calloc((size_t)L * L * qq, sizeof(real));
    if (!slab) return PLMO_ENOMEM;
    double nll = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : nll)
    for (int i = 0; i < L; i++) {
        real H[64], Pr[64];
        real *si = slab + (size_t)i * L * qq;
        real *gh = g + (size_t)i * q;
        for (int a = 0; a < q; a++) gh[a] = 0;
        double site = 0;
        for (int s = 0; s < N; s++) {
            const int8_t *row = msa + (size_t)s * L;
            const real ws = w[s];
            for (int a = 0; a < q; a++) H[a] = x[(size_t)i * q + a];
            for (int j = 0; j < i; j++) {
                const real *blk = J + pair_index(j, i, L) * qq + (size_t)row[j] * q; /* J_ji(x_sj, a) */
                for (int a = 0; a < q; a++) H[a] += blk[a];
            }
            for (int j = i + 1; j < L; j++) {
                const real *blk = J + pair_index(i, j, L) * qq + row[j]; /* J_ij(a, x_sj) */
                for (int a = 0; a < q; a++) H[a] += blk[(size_t)a * q];
            }
            real mx = H[0];
            for (int a = 1; a < q; a++) mx = H[a] > mx ? H[a] : mx;
            double Z = 0;
            for (int a = 0; a < q; a++) {
                Pr[a] = (real)exp((double)(H[a] - mx));
                Z += Pr[a];
            }
            const int xi = row[i];
            site -= (double)ws * ((double)(H[xi] - mx) - log(Z));
            for (int a = 0; a < q; a++) Pr[a] = (real)(ws * (Pr[a] / Z));
            Pr[xi] -= ws; /* residual r_si(a) = w_s (P_si(a) - [x_si = a]) */
            for (int a = 0; a < q; a++) gh[a] += Pr[a];
            for (int j = 0; j < L; j++) {
                if (j == i) continue;
                real *dst = si + (size_t)j * qq + row[j];
                for (int a = 0; a < q; a++) dst[(size_t)a * q] += Pr[a];
            }
        }
        nll += site;
    }
    double reg = 0;
    for (size_t k = 0; k < nh; k++) {
        reg += lambda_h * (double)x[k] * (double)x[k];
        g[k] += (real)(2.0 * lambda_h * x[k]);
    }
    double regj = 0;
#pragma omp parallel for schedule(static) reduction(+ : regj)
    for (int i = 0; i < L - 1; i++)
        for (int j = i + 1; j < L; j++) {
            const size_t p = pair_index(i, j, L) * qq;
            const real *sij = slab + ((size_t)i * L + j) * qq;
            const real *sji = slab + ((size_t)j * L + i) * qq;
            for (int a = 0; a < q; a++)
                for (int b = 0; b < q; b++) {
                    const real xv = J[p + (size_t)a * q + b];
                    regj += lambda_j * (double)xv * (double)xv;
                    g[nh + p + (size_t)a * q + b] =
                        sij[(size_t)a * q + b] + sji[(size_t)b * q + a] + (real)(2.0 * lambda_j * xv);
                }
        }
    (void)npair;
    free(slab);
    regj += group_term(J, g + nh, L, q);
    if (fx_out) *fx_out = nll + reg + regj;
    if (nll_out) *nll_out = nll;
    return PLMO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Gap-ignoring mode (plmc -g; run_plmc's ignore_gaps, evcouplings/couplings/tools.py:222-224,
 * protocol.py:159-165 "if we ignore gaps, there is one character less").  PARITY UNPINNED and
 * partly a design decision: plmc's exact -g semantics cannot be checked here.  The spec used by
 * both this oracle and the HIP path:
 *   - the model has qn = q-1 states (alphabet without its first, gap, character); x, g, fi, fij
 *     below are in that qn-state layout; msa still holds 0..q-1 with 0 = gap
 *   - reweighting: a position counts as identical only if both residues are equal AND not gaps;
 *     the threshold stays T = ceil(theta*L - 1e-9) on the full model length
 *   - site i of sequence s contributes a conditional only if x_si is not a gap; gapped
 *     neighbours contribute no coupling; the softmax runs over the qn non-gap states
 *   - f_i(a) = sum_s w_s [x_si=a] / sum_s w_s [x_si != gap]; f_ij normalised by the weight of
 *     sequences ungapped at both sites
 * ------------------------------------------------------------------------------------------ */
int PLMO_NAME(reweight_gaps)(const int8_t *msa, int N, int L, double theta_id, int32_t *counts) {
    if (!msa || !counts || N <= 0 || L <= 0) return PLMO_EINVAL;
    const int T = PLMO_NAME(threshold)(L, theta_id);
#pragma omp parallel for schedule(dynamic, 16)
    for (int s = 0; s < N; s++) {
        const int8_t *a = msa + (size_t)s * L;
        int c = 0;
        for (int t = 0; t < N; t++) {
            const int8_t *b = msa + (size_t)t * L;
            int id = 0, nb = 0;
            if ((g_conv & 64) && !(g_conv & 128)) {          /* gap-gap positions are identities, as without -g */
                for (int k = 0; k < L; k++) id += (a[k] == b[k]);
                c += (id >= T) || (t == s);
                continue;
            }
            for (int k = 0; k < L; k++) {
                id += (a[k] == b[k]) && (a[k] != 0);
                nb += (a[k] != 0) && (b[k] != 0);
            }
            if (g_conv & 128)                                /* threshold on the jointly ungapped length */
                c += (id >= (int)ceil(theta_id * (double)nb - 1e-9)) || (t == s);
            else
                c += (id >= T) || (t == s);   /* a sequence always belongs to its own cluster */
        }
        counts[s] = c;
    }
    return PLMO_OK;
}

int PLMO_NAME(marginals_gaps)(const int8_t *msa, const real *w, int N, int L, int q, real *fi, real *fij) {
    if (!msa || !w || !fi || N <= 0 || L <= 0 || q <= 1) return PLMO_EINVAL;
    const int qn = q - 1;
    const size_t qq = (size_t)qn * qn;
    double neff = 0;
    for (int s = 0; s < N; s++) neff += w[s];
    const int by_total = (g_conv & 256) != 0;    /* normalise by N_eff instead of the ungapped weight */
    for (int i = 0; i < L; i++) {
        double c[64] = {0}, tot = 0;
        for (int s = 0; s < N; s++) {
            const int a = msa[(size_t)s * L + i];
            if (a > 0) { c[a - 1] += w[s]; tot += w[s]; }
        }
        if (by_total) tot = neff;
        for (int a = 0; a < qn; a++) fi[(size_t)i * qn + a] = (real)(tot > 0 ? c[a] / tot : 0);
    }
    if (fij) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int i = 0; i < L - 1; i++)
            for (int j = i + 1; j < L; j++) {
                real *blk = fij + pair_index(i, j, L) * qq;
                double *c = (double *)calloc(qq, sizeof(double)), tot = 0;
                for (int s = 0; s < N; s++) {
                    const int a = msa[(size_t)s * L + i], b = msa[(size_t)s * L + j];
                    if (a > 0 && b > 0) { c[(size_t)(a - 1) * qn + (b - 1)] += w[s]; tot += w[s]; }
                }
                if (by_total) tot = neff;
                for (size_t k = 0; k < qq; k++) blk[k] = (real)(tot > 0 ? c[k] / tot : 0);
                free(c);
            }
    }
    return PLMO_OK;
}

int PLMO_NAME(eval_gaps)(const int8_t *msa, const real *w, int N, int L, int q, double lambda_h,
                         double lambda_j, const real *x, real *g, double *fx_out, double *nll_out) {
    if (!msa || !w || !x || !g || N <= 0 || L <= 1 || q <= 2 || q > 64) return PLMO_EINVAL;
    const int qn = q - 1;
    const size_t qq = (size_t)qn * qn, nh = (size_t)L * qn;
    const real *J = x + nh;
    real *slab = (real *)calloc((size_t)L * L * qq, sizeof(real));
    if (!slab) return PLMO_ENOMEM;
    double nll = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : nll)
    for (int i = 0; i < L; i++) {
        real H[64], Pr[64];
        real *si = slab + (size_t)i * L * qq;
        real *gh = g + (size_t)i * qn;
        for (int a = 0; a < qn; a++) gh[a] = 0;
        double site = 0;
        for (int s = 0; s < N; s++) {
            const int8_t *row = msa + (size_t)s * L;
            const int xi = row[i] - 1;
            if (xi < 0) continue;                       /* gapped site: no conditional */
            const real ws = w[s];
            for (int a = 0; a < qn; a++) H[a] = x[(size_t)i * qn + a];
            for (int j = 0; j < L; j++) {
                const int xj = row[j] - 1;
                if (j == i || xj < 0) continue;         /* gapped neighbour: no coupling */
                if (j < i) {
                    const real *blk = J + pair_index(j, i, L) * qq + (size_t)xj * qn;
                    for (int a = 0; a < qn; a++) H[a] += blk[a];
                } else {
                    const real *blk = J + pair_index(i, j, L) * qq + xj;
                    for (int a = 0; a < qn; a++) H[a] += blk[(size_t)a * qn];
                }
            }
            real mx = H[0];
            for (int a = 1; a < qn; a++) mx = H[a] > mx ? H[a] : mx;
            double Z = 0;
            for (int a = 0; a < qn; a++) {
                Pr[a] = (real)exp((double)(H[a] - mx));
                Z += Pr[a];
            }
            site -= (double)ws * ((double)(H[xi] - mx) - log(Z));
            for (int a = 0; a < qn; a++) Pr[a] = (real)(ws * (Pr[a] / Z));
            Pr[xi] -= ws;
            for (int a = 0; a < qn; a++) gh[a] += Pr[a];
            for (int j = 0; j < L; j++) {
                const int xj = row[j] - 1;
                if (j == i || xj < 0) continue;
                real *dst = si + (size_t)j * qq + xj;
                for (int a = 0; a < qn; a++) dst[(size_t)a * qn] += Pr[a];
            }
        }
        nll += site;
    }
    double reg = 0;
    for (size_t k = 0; k < nh; k++) {
        reg += lambda_h * (double)x[k] * (double)x[k];
        g[k] += (real)(2.0 * lambda_h * x[k]);
    }
    double regj = 0;
#pragma omp parallel for schedule(static) reduction(+ : regj)
    for (int i = 0; i < L - 1; i++)
        for (int j = i + 1; j < L; j++) {
            const size_t p = pair_index(i, j, L) * qq;
            const real *sij = slab + ((size_t)i * L + j) * qq;
            const real *sji = slab + ((size_t)j * L + i) * qq;
            for (int a = 0; a < qn; a++)
                for (int b = 0; b < qn; b++) {
                    const real xv = J[p + (size_t)a * qn + b];
                    regj += lambda_j * (double)xv * (double)xv;
                    g[nh + p + (size_t)a * qn + b] =
                        sij[(size_t)a * qn + b] + sji[(size_t)b * qn + a] + (real)(2.0 * lambda_j * xv);
                }
        }
    free(slab);
    regj += group_term(J, g + nh, L, qn);
    if (fx_out) *fx_out = nll + reg + regj;
    if (nll_out) *nll_out = nll;
    return PLMO_OK;
}

/* Zero-sum gauge -> Frobenius norm -> APC, following
 * evcouplings/couplings/model.py:179-233 (_zero_sum_gauge), :790-793 (FN over all q
 * states) and :744-775 (apc: column means over off-diagonal entries, factor L/(L-1),
 * diagonal blanked).  jij = i<j blocks; fn, cn = dense L x L (symmetric, zero diag). */
static int scores_impl(const real *jij, int L, int q, int a_lo, double *fn, double *cn);
int PLMO_NAME(scores)(const real *jij, int L, int q, double *fn, double *cn) {
    return scores_impl(jij, L, q, (g_conv & 512) ? 1 : 0, fn, cn);
}
/* a_lo = 1: the gap state is left out of the Frobenius norm (the gauge is still taken over all q states) */
static int scores_impl(const real *jij, int L, int q, int a_lo, double *fn, double *cn) {
    if (!jij || !fn || !cn || L <= 1 || q <= 0) return PLMO_EINVAL;
    const size_t qq = (size_t)q * q;
    memset(fn, 0, sizeof(double) * (size_t)L * L);
    for (int i = 0; i < L - 1; i++)
        for (int j = i + 1; j < L; j++) {
            const real *blk = jij + pair_index(i, j, L) * qq;
            double rm[64] = {0}, cm[64] = {0}, m = 0;
            for (int a = 0; a < q; a++)
                for (int b = 0; b < q; b++) {
                    const double v = blk[(size_t)a * q + b];
                    rm[a] += v;
                    cm[b] += v;
                    m += v;
                }
            m /= (double)qq;
            double ss = 0;
            for (int a = 0; a < q; a++)
                for (int b = 0; b < q; b++) {
                    const double v = blk[(size_t)a * q + b] - rm[a] / q - cm[b] / q + m;
                    if (a >= a_lo && b >= a_lo) ss += v * v;
                }
            fn[(size_t)i * L + j] = fn[(size_t)j * L + i] = sqrt(ss);
        }
    double *col = (double *)calloc((size_t)L, sizeof(double));
    if (!col) return PLMO_ENOMEM;
    double tot = 0;
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++) {
            col[j] += fn[(size_t)i * L + j];
            tot += fn[(size_t)i * L + j];
        }
    /* np.mean(axis=0) * L/(L-1) == column sum / (L-1);  np.mean() * L/(L-1) == tot / (L (L-1)) */
    const double mean = tot / ((double)L * (L - 1));
    for (int j = 0; j < L; j++) col[j] /= (double)(L - 1);
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++)
            cn[(size_t)i * L + j] = (i == j) ? 0.0 : fn[(size_t)i * L + j] - col[i] * col[j] / mean;
    free(col);
    return PLMO_OK;
}

/* Statistical energies of sequences under a model, following
 * evcouplings/couplings/model.py:25-60 (_hamiltonians): for every sequence the sums
 * H_h = sum_i h_i(x_i), H_J = sum_{i<j} J_ij(x_i, x_j), H = H_J + H_h, accumulated in double in
 * the reference's loop order.  x = canonical vector (h [L][q], then i<j blocks [q][q]);
 * out = n x 3 doubles (H, H_J, H_h) -- the reference's column order (FULL, COUPLINGS, FIELDS). */
int PLMO_NAME(hamiltonians)(const int8_t *seqs, int n, int L, int q, const real *x, double *out) {
    if (!seqs || !x || !out || L <= 1 || q <= 0) return PLMO_EINVAL;
    const real *h = x, *jij = x + (size_t)L * q;
    const size_t qq = (size_t)q * q;
    for (int s = 0; s < n; s++) {
        const int8_t *A = seqs + (size_t)s * L;
        double hs = 0, js = 0;
        for (int i = 0; i < L; i++) {
            hs += h[(size_t)i * q + A[i]];
            for (int j = i + 1; j < L; j++) js += jij[pair_index(i, j, L) * qq + (size_t)A[i] * q + A[j]];
        }
        out[(size_t)s * 3 + 0] = js + hs;
        out[(size_t)s * 3 + 1] = js;
        out[(size_t)s * 3 + 2] = hs;
    }
    return PLMO_OK;
}

/* All single-site substitutions of a target sequence, following
 * evcouplings/couplings/model.py:63-109 (_single_mutant_hamiltonians):
 * dh(i,a) = h_i(a) - h_i(t_i); dJ(i,a) = sum_{j != i} [J_ij(a, t_j) - J_ij(t_i, t_j)];
 * out = L x q x 3 doubles (dJ + dh, dJ, dh). */
int PLMO_NAME(single_mutants)(const int8_t *target, int L, int q, const real *x, double *out) {
    if (!target || !x || !out || L <= 1 || q <= 0) return PLMO_EINVAL;
    const real *h = x, *jij = x + (size_t)L * q;
    const size_t qq = (size_t)q * q;
    for (int i = 0; i < L; i++)
        for (int a = 0; a < q; a++) {
            const double dh = (double)h[(size_t)i * q + a] - (double)h[(size_t)i * q + target[i]];
            double dj = 0;
            for (int j = 0; j < L; j++) {
                if (j == i) continue;
                /* J_ij(a, b) for i > j is the transposed block of the stored pair (j, i) */
                const real *blk = (i < j) ? jij + pair_index(i, j, L) * qq : jij + pair_index(j, i, L) * qq;
                const double v_new = (i < j) ? blk[(size_t)a * q + target[j]] : blk[(size_t)target[j] * q + a];
                const double v_old = (i < j) ? blk[(size_t)target[i] * q + target[j]]
                                             : blk[(size_t)target[j] * q + target[i]];
                dj += v_new - v_old;
            }
            double *o = out + ((size_t)i * q + a) * 3;
            o[0] = dj + dh;
            o[1] = dj;
            o[2] = dh;
        }
    return PLMO_OK;
}

/* ------------------------------------------------------------------------- */
/* L-BFGS with a More'-Thuente line search (the published  */
/* algorithms: Nocedal 1980 two-loop recursion, More' & Thuente 1994).          */
/* PARITY UNPINNED: plmc's bundled optimiser is not available to compare with.  */
/* ------------------------------------------------------------------------- */

typedef void (*plmo_iter_cb)(int iter, double secs, double cond, double fx, double nll,
                             double norm_h, double norm_e, void *user);

typedef struct {
    const int8_t *msa;
    const real *w;
    int N, L, q;
    double lh, lj;
    int nevals;
    int gaps;        /* 1: gap-ignoring mode, model has q-1 states */
} evalctx_t;

static int ctx_eval(evalctx_t *c, const real *x, real *g, double *fx, double *nll) {
    c->nevals++;
    if (c->gaps) return PLMO_NAME(eval_gaps)(c->msa, c->w, c->N, c->L, c->q, c->lh, c->lj, x, g, fx, nll);
    return PLMO_NAME(eval)(c->msa, c->w, c->N, c->L, c->q, c->lh, c->lj, x, g, fx, nll);
}

static double vdot(const real *a, const real *b, size_t n) {
    double s = 0;
#pragma omp parallel for reduction(+ : s)
    for (size_t k = 0; k < n; k++) s += (double)a[k] * (double)b[k];
    return s;
}

/* One safeguarded trial-step update (More' & Thuente 1994, section 4). */
static int mt_update(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy,
                     double *stp, double fp, double dp, double tmin, double tmax, int *brackt) {
    double stpf, stpc, stpq, gamma, p, qq, r, s, theta;
    int bound;
    const double sgnd = dp * (*dx / fabs(*dx));
    if (*brackt && (*stp <= fmin(*stx, *sty) || *stp >= fmax(*stx, *sty))) return -1;
    if (*dx * (*stp - *stx) >= 0.0) return -1;
    if (tmax < tmin) return -1;
    if (fp > *fx) { /* case 1: higher function value -> minimum bracketed */
        bound = 1;
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma = -gamma;
        p = (gamma - *dx) + theta;
        qq = ((gamma - *dx) + gamma) + dp;
        r = p / qq;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
        stpf = (fabs(stpc - *stx) < fabs(stpq - *stx)) ? stpc : stpc + (stpq - stpc) / 2.0;
        *brackt = 1;
    } else if (sgnd < 0.0) { /* case 2: derivatives of opposite sign -> bracketed */
        bound = 0;
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        qq = ((gamma - dp) + gamma) + *dx;
        r = p / qq;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        stpf = (fabs(stpc - *stp) > fabs(stpq - *stp)) ? stpc : stpq;
        *brackt = 1;
    } else if (fabs(dp) < fabs(*dx)) { /* case 3: derivative magnitude decreases */
        bound = 1;
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
        gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        qq = (gamma + (*dx - dp)) + gamma;
        r = p / qq;
        if (r < 0.0 && gamma != 0.0)
            stpc = *stp + r * (*stx - *stp);
        else if (*stp > *stx)
            stpc = tmax;
        else
            stpc = tmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt)
            stpf = (fabs(*stp - stpc) < fabs(*stp - stpq)) ? stpc : stpq;
        else
            stpf = (fabs(*stp - stpc) > fabs(*stp - stpq)) ? stpc : stpq;
    } else { /* case 4: derivative magnitude does not decrease */
        bound = 0;
        if (*brackt) {
            theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            s = fmax(fabs(theta), fmax(fabs(*dy), fabs(dp)));
            gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            qq = ((gamma - dp) + gamma) + *dy;
            r = p / qq;
            stpf = *stp + r * (*sty - *stp);
        } else if (*stp > *stx)
            stpf = tmax;
        else
            stpf = tmin;
    }
    if (fp > *fx) {
        *sty = *stp;
        *fy = fp;
        *dy = dp;
    } else {
        if (sgnd < 0.0) {
            *sty = *stx;
            *fy = *fx;
            *dy = *dx;
        }
        *stx = *stp;
        *fx = fp;
        *dx = dp;
    }
    stpf = fmin(tmax, stpf);
    stpf = fmax(tmin, stpf);
    *stp = stpf;
    if (*brackt && bound) {
        const double lim = *stx + 0.66 * (*sty - *stx);
        if (*sty > *stx)
            *stp = fmin(lim, *stp);
        else
            *stp = fmax(lim, *stp);
    }
    return 0;
}

typedef struct {
    int max_iter;    /* 0 = until converged */
    double epsilon;  /* |g| / max(1,|x|) */
    int m;           /* history */
    int max_ls;      /* function evaluations per line search */
    double ftol, gtol, xtol, stpmin, stpmax;
    double epsf;     /* relative noise allowance on f in the sufficient-decrease test */
} lbfgs_opt_t;

/* returns number of iterations done (>=0) or a negative error; status: 0 converged,
 * 1 max iterations, 2 line search gave up (x holds the best point found). */
static int lbfgs_run(evalctx_t *c, size_t n, real *x, const lbfgs_opt_t *o, plmo_iter_cb cb,
                     void *user, double *fx_final, int *status) {
    const int m = o->m;
    const size_t nh = (size_t)c->L * (c->q - c->gaps);
    real *g = malloc(sizeof(real) * n), *xp = malloc(sizeof(real) * n), *gp = malloc(sizeof(real) * n);
    real *d = malloc(sizeof(real) * n);
    real *S = malloc(sizeof(real) * n * m), *Y = malloc(sizeof(real) * n * m);
    double *alpha = malloc(sizeof(double) * m), *ys = malloc(sizeof(double) * m);
    if (!g || !xp || !gp || !d || !S || !Y || !alpha || !ys) return PLMO_ENOMEM;
    const double t0 = now_s();
    double fx, nll;
    int rc = ctx_eval(c, x, g, &fx, &nll);
    if (rc) return rc;
    double xnorm = sqrt(vdot(x, x, n)), gnorm = sqrt(vdot(g, g, n));
    int k = 0, end = 0, stored = 0;
    *status = 0;
    if (gnorm / fmax(1.0, xnorm) <= o->epsilon) goto done;
    for (size_t t = 0; t < n; t++) d[t] = -g[t];
    double step = 1.0 / sqrt(vdot(d, d, n));
    for (k = 1;; k++) {
        memcpy(xp, x, sizeof(real) * n);
        memcpy(gp, g, sizeof(real) * n);
        /* ---- More'-Thuente line search along d ---- */
        {
            const double dginit = vdot(g, d, n);
            if (dginit >= 0) { *status = 2; k--; break; }
            const double finit = fx, dgtest = o->ftol * dginit;
            int brackt = 0, stage1 = 1, count = 0, uinfo = 0, lsrc = 1;
            double width = o->stpmax - o->stpmin, prev_width = 2.0 * width;
            double stx = 0, fxx = finit, dgx = dginit, sty = 0, fy = finit, dgy = dginit;
            double stp = step, stmin, stmax;
            for (;;) {
                if (brackt) {
                    stmin = fmin(stx, sty);
                    stmax = fmax(stx, sty);
                } else {
                    stmin = stx;
                    stmax = stp + 4.0 * (stp - stx);
                }
                if (stp < o->stpmin) stp = o->stpmin;
                if (stp > o->stpmax) stp = o->stpmax;
                if ((brackt && (stp <= stmin || stmax <= stp || count >= o->max_ls - 1 || uinfo)) ||
                    (brackt && stmax - stmin <= o->xtol * stmax))
                    stp = stx;
#pragma omp parallel for
                for (size_t t = 0; t < n; t++) x[t] = (real)(xp[t] + stp * d[t]);
                rc = ctx_eval(c, x, g, &fx, &nll);
                if (rc) return rc;
                const double dg = vdot(g, d, n);
                const double ftest1 = finit + stp * dgtest;
                count++;
                if (brackt && (stp <= stmin || stmax <= stp || uinfo)) { lsrc = -1; break; }
                if (stp == o->stpmax && fx <= ftest1 && dg <= dgtest) { lsrc = -2; break; }
                if (stp == o->stpmin && (ftest1 < fx || dgtest <= dg)) { lsrc = -3; break; }
                if (brackt && stmax - stmin <= o->xtol * stmax) { lsrc = -4; break; }
                if (count >= o->max_ls) { lsrc = -5; break; }
                /* accept on strong Wolfe; once the decrease drowns in rounding noise of f, fall back
                 * to "f did not rise beyond noise" + the curvature condition (the approximate Wolfe
                 * idea of Hager & Zhang 2005) so the gradient test can still be reached */
                if ((fx <= ftest1 || fx <= finit + o->epsf * fabs(finit)) && fabs(dg) <= o->gtol * (-dginit)) {
                    lsrc = 1;
                    break;
                }
                if (stage1 && fx <= ftest1 && fmin(o->ftol, o->gtol) * dginit <= dg) stage1 = 0;
                if (stage1 && ftest1 < fx && fx <= fxx) {
                    double fm = fx - stp * dgtest, fxm = fxx - stx * dgtest, fym = fy - sty * dgtest;
                    double dgm = dg - dgtest, dgxm = dgx - dgtest, dgym = dgy - dgtest;
                    uinfo = mt_update(&stx, &fxm, &dgxm, &sty, &fym, &dgym, &stp, fm, dgm, stmin, stmax, &brackt);
                    fxx = fxm + stx * dgtest;
                    fy = fym + sty * dgtest;
                    dgx = dgxm + dgtest;
                    dgy = dgym + dgtest;
                } else {
                    uinfo = mt_update(&stx, &fxx, &dgx, &sty, &fy, &dgy, &stp, fx, dg, stmin, stmax, &brackt);
                }
                if (brackt) {
                    if (0.66 * prev_width <= fabs(sty - stx)) stp = stx + 0.5 * (sty - stx);
                    prev_width = width;
                    width = fabs(sty - stx);
                }
            }
            if (lsrc < 0) {
                /* give up: accept the point only if it did not increase f */
                if (fx > finit) {
                    memcpy(x, xp, sizeof(real) * n);
                    memcpy(g, gp, sizeof(real) * n);
                    fx = finit;
                    rc = ctx_eval(c, x, g, &fx, &nll);
                    if (rc) return rc;
                }
                *status = 2;
                k--;
                break;
            }
            step = stp;
        }
        xnorm = sqrt(vdot(x, x, n));
        gnorm = sqrt(vdot(g, g, n));
        if (cb)
            cb(k, now_s() - t0, gnorm / fmax(1.0, xnorm), fx, nll, sqrt(vdot(x, x, nh)),
               sqrt(vdot(x + nh, x + nh, n - nh)), user);
        if (gnorm / fmax(1.0, xnorm) <= o->epsilon) { *status = 0; break; }
        if (o->max_iter > 0 && k >= o->max_iter) { *status = 1; break; }
        /* ---- update history, two-loop recursion ---- */
        real *s = S + (size_t)end * n, *y = Y + (size_t)end * n;
#pragma omp parallel for
        for (size_t t = 0; t < n; t++) {
            s[t] = x[t] - xp[t];
            y[t] = g[t] - gp[t];
        }
        const double ysv = vdot(y, s, n), yy = vdot(y, y, n);
        ys[end] = ysv;
        if (stored < m) stored++;
        end = (end + 1) % m;
        for (size_t t = 0; t < n; t++) d[t] = -g[t];
        int j = end;
        for (int i = 0; i < stored; i++) {
            j = (j + m - 1) % m;
            const real *sj = S + (size_t)j * n, *yj = Y + (size_t)j * n;
            alpha[j] = vdot(sj, d, n) / ys[j];
            const double aj = alpha[j];
#pragma omp parallel for
            for (size_t t = 0; t < n; t++) d[t] = (real)(d[t] - aj * yj[t]);
        }
        const double sc = ysv / yy;
#pragma omp parallel for
        for (size_t t = 0; t < n; t++) d[t] = (real)(d[t] * sc);
        for (int i = 0; i < stored; i++) {
            const real *sj = S + (size_t)j * n, *yj = Y + (size_t)j * n;
            const double beta = vdot(yj, d, n) / ys[j];
            const double cf = alpha[j] - beta;
#pragma omp parallel for
            for (size_t t = 0; t < n; t++) d[t] = (real)(d[t] + cf * sj[t]);
            j = (j + 1) % m;
        }
        step = 1.0;
    }
done:
    *fx_final = fx;
    free(g); free(xp); free(gp); free(d); free(S); free(Y); free(alpha); free(ys);
    return k;
}

/* Full fit: reweight -> marginals -> L-BFGS from (h = centred log-frequency, J = 0)
 * -> scores.  All output buffers caller-allocated; any may be NULL except x_out.
 *   weights N, fi L*qm, fij npair*qm*qm, x_out L*qm + npair*qm*qm, fn/cn L*L (double),
 *   qm = q (ignore_gaps = 0) or q-1 (ignore_gaps = 1, see eval_gaps). */
int PLMO_NAME(fit2)(const int8_t *msa, int N, int L, int q, double theta_id, double scale,
                    double lambda_h, double lambda_j, int max_iter, double epsilon, int lbfgs_m,
                    int ignore_gaps, real *weights, double *neff_out, real *fi, real *fij, real *x_out,
                    double *fn, double *cn, int *iters_out, int *status_out, int *nevals_out,
                    double *fx_out, plmo_iter_cb cb, void *user) {
    if (!msa || !x_out || N <= 0 || L <= 1 || q <= 1 + (ignore_gaps ? 1 : 0) || q > 64) return PLMO_EINVAL;
    for (size_t k = 0; k < (size_t)N * L; k++)
        if (msa[k] < 0 || msa[k] >= q) return PLMO_EINVAL;
    const int gaps = ignore_gaps ? 1 : 0, qm = q - gaps;
    const size_t qq = (size_t)qm * qm, nh = (size_t)L * qm, npair = (size_t)L * (L - 1) / 2;
    const size_t n = nh + npair * qq;
    int32_t *counts = malloc(sizeof(int32_t) * N);
    real *w = weights ? weights : malloc(sizeof(real) * N);
    real *fi_l = fi ? fi : malloc(sizeof(real) * nh);
    int rc = (!counts || !w || !fi_l) ? PLMO_ENOMEM : PLMO_OK;
    int iters = 0, status = 0;
    double fx = 0, neff = 0;
    evalctx_t c = {msa, w, N, L, q, lambda_h, lambda_j, 0, gaps};
    if (!rc)
        rc = gaps ? PLMO_NAME(reweight_gaps)(msa, N, L, theta_id, counts)
                  : PLMO_NAME(reweight)(msa, N, L, theta_id, counts);
    if (!rc) {
        for (int s = 0; s < N; s++) {
            w[s] = (real)(scale / counts[s]);
            neff += w[s];
        }
        if (neff_out) *neff_out = neff;
        rc = gaps ? PLMO_NAME(marginals_gaps)(msa, w, N, L, q, fi_l, fij)
                  : PLMO_NAME(marginals)(msa, w, N, L, q, fi_l, fij);
    }
    if (!rc) {
        /* start point: h_i(a) = log(f_i(a) + 1/N_eff) minus its site mean, J = 0 */
        memset(x_out, 0, sizeof(real) * n);
        for (int i = 0; i < L; i++) {
            double mean = 0;
            for (int a = 0; a < qm; a++) {
                const double v = log((double)fi_l[(size_t)i * qm + a] + 1.0 / neff);
                x_out[(size_t)i * qm + a] = (real)v;
                mean += v;
            }
            mean /= qm;
            for (int a = 0; a < qm; a++) x_out[(size_t)i * qm + a] -= (real)mean;
        }
        lbfgs_opt_t o = {max_iter, epsilon, lbfgs_m > 0 ? lbfgs_m : 6, 20, 1e-4, 0.9, 1e-16, 1e-20, 1e20,
                         sizeof(real) == 4 ? 1e-6 : 1e-13};
        iters = lbfgs_run(&c, n, x_out, &o, cb, user, &fx, &status);
        if (iters < 0) rc = iters;
    }
    if (!rc) {
        if (iters_out) *iters_out = iters;
        if (status_out) *status_out = status;
        if (nevals_out) *nevals_out = c.nevals;
        if (fx_out) *fx_out = fx;
        if (fn && cn) rc = scores_impl(x_out + nh, L, qm, (!gaps && (g_conv & 512)) ? 1 : 0, fn, cn);
    }
    free(counts);
    if (!weights) free(w);
    if (!fi) free(fi_l);
    return rc;
}

int PLMO_NAME(fit)(const int8_t *msa, int N, int L, int q, double theta_id, double scale,
                   double lambda_h, double lambda_j, int max_iter, double epsilon, int lbfgs_m,
                   real *weights, double *neff_out, real *fi, real *fij, real *x_out, double *fn,
                   double *cn, int *iters_out, int *status_out, int *nevals_out, double *fx_out,
                   plmo_iter_cb cb, void *user) {
    return PLMO_NAME(fit2)(msa, N, L, q, theta_id, scale, lambda_h, lambda_j, max_iter, epsilon, lbfgs_m, 0,
                           weights, neff_out, fi, fij, x_out, fn, cn, iters_out, status_out, nevals_out,
                           fx_out, cb, user);
}
