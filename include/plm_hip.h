/*
 * plm_hip.h -- C ABI of libplm_hip.so, the MI355X (gfx950) pseudo-likelihood Potts solver
 * that replaces the external `plmc` process behind EVcouplings' couplings stage.
 *
 * Nothing like this ABI exists in the reference: its boundary for this path is a
 * subprocess (evcouplings/couplings/tools.py:202-266 builds the argv, :266 launches it,
 * :286 parses stderr).  Each entry point below names the piece of that boundary it
 * replaces.  All functions return PLM_OK (0) or a negative PLM_E* code, never throw and
 * never abort; plm_last_error() holds a message for the calling thread.  Host buffers are
 * caller-owned; the library owns its device memory.  No torch types cross this boundary.
 *
 * Parameter-vector layout ("canonical", identical to the order of the plmc_v2 `.model`
 * file, evcouplings/couplings/model.py:355-389):
 *     x = [ h_i(a) : i<L, a<q ] ++ [ J_ij(a,b) : pairs i<j row-major, a<q (site i), b<q (site j) ]
 */
#ifndef PLM_HIP_H
#define PLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libplm_hip.so is built with -fvisibility=hidden: the declarations between this push and the pop at the end of the
 * file are its whole dynamic symbol table (tests/test_abi.py compares `nm -D` with this header). */
#pragma GCC visibility push(default)

#define PLM_ABI_VERSION 2   /* 2: plm_iter_cb returns int (cancellation), PLM_STATUS_INTERRUPTED, lambda_group */

#define PLM_OK 0
#define PLM_EINVAL (-1)      /* bad argument (NULL, size <= 0, state outside 0..q-1, ...) */
#define PLM_ENOMEM (-2)      /* host or device allocation failed, or the problem needs more device memory than is free (checked up front) */
#define PLM_EDEVICE (-3)     /* HIP runtime error / no gfx950 device */
#define PLM_EUNSUPPORTED (-4)/* alphabet size outside 2..32 (any size in that range runs, padded to 4 / 5 / 20 / 21 / 32) */
#define PLM_ENUMERIC (-5)    /* NaN/Inf met in objective */
#define PLM_ECALLBACK (-6)   /* exchange / collective callback or an RCCL call reported failure */

/* optimisation end states (plm_result_t.status) -> "Gradient optimization: (.+)" line that
 * tools.py:56,99 parses */
#define PLM_STATUS_CONVERGED 0
#define PLM_STATUS_MAXITER 1
#define PLM_STATUS_LINESEARCH 2
#define PLM_STATUS_INTERRUPTED 3   /* the iteration callback asked to stop; the result holds the point reached so far */

typedef struct plm_ctx plm_ctx_t;

/* One inference problem.  Mirrors the plmc options run_plmc() can set
 * (tools.py:126-130, 222-259): -t theta, -s scale, -lh, -le, -m iterations. */
typedef struct {
    int32_t n_seqs;      /* N  valid sequences */
    int32_t n_sites;     /* L  model columns */
    int32_t n_states;    /* q  alphabet size, gap first (alignment.py:25-26) */
    const int8_t *msa;   /* host, row-major N x L, values 0..q-1 */
    double theta_id;     /* identity threshold (0.8); cluster if ident >= ceil(theta*L - 1e-9) */
    double scale;        /* cluster weight scale (plmc -s), w_s = scale / cluster size */
    double lambda_h;     /* L2 strength on fields */
    double lambda_j;     /* L2 strength on couplings, as passed to plmc -le (already scaled
                            by (q-1)(L-1), protocol.py:179) */
    int32_t max_iter;    /* L-BFGS iterations; 0 = until converged */
    double epsilon;      /* stop when |g| / max(1,|x|) < epsilon.  Also selects the precision of the backward GEMM's
                            fixed-point residuals at context creation: 24 bits (three int8 digit planes), or 32 bits
                            (four planes, 4/3 of the cost) when 0 < epsilon < 1e-4 */
    int32_t lbfgs_m;     /* history length; 0 = default (6) */
    int32_t n_shards;    /* site shards (GPUs); 1 = single GPU */
    int32_t shard;       /* this process' shard index */
    int32_t flags;       /* PLM_FLAG_* */
    double lambda_group; /* run_plmc's lambda_g (plmc -lg, tools.py:252-253): group regulariser on the coupling blocks,
                            lambda_group * sum_{i<j} sqrt(|J_ij|_F^2 + PLM_GROUP_DELTA^2); 0 = none */
} plm_problem_t;
/* smoothing of the group norm at the origin (the start point is J = 0, where the bare norm has no gradient) */
#define PLM_GROUP_DELTA 1e-4

#define PLM_FLAG_NONE 0
#define PLM_FLAG_VERBOSE 1
/* plmc -g / run_plmc(ignore_gaps=True) (tools.py:222-224): state 0 (the gap) is excluded from the
 * model.  Sites where a sequence has a gap contribute no conditional, gapped neighbours no coupling,
 * identities for reweighting count only non-gap matches, frequencies are normalised over ungapped
 * sequences.  All arrays keep the q-state layout of the alphabet; every entry that involves state 0
 * is zero (the Python host drops them and writes a (q-1)-state model file).  DESIGN.md section 2b. */
#define PLM_FLAG_IGNORE_GAPS 2
/* n_shards > 1 only (at most 16): parameters, gradient and L-BFGS state are split over the shards instead of
 * replicated: a shard's fields, the block pairs inside its site blocks and half of every
 * rectangle of block pairs it shares with another shard.  Needs a plm_collective_cb; per evaluation two all-to-alls
 * (couplings, gradient fragments; every rank exchanges with every other) and one scalar all-reduce replace the
 * all-gather of whole gradient slabs; the fitted parameters are assembled with one PLM_COLL_ALLREDUCE_F32 of the
 * canonical vector. */
#define PLM_FLAG_SHARDED_STATE 4
/* L-BFGS with a diagonal initial Hessian (inverse Hessian diagonal of the independent-site model) instead of the
 * textbook scalar one.  Same optimum (strictly convex objective), different path.  Measured on MI355X: fewer
 * iterations on short alignments, MORE at L = 300 -- opt-in. */
#define PLM_FLAG_PRECOND 8
/* Optimise fields and couplings jointly with L-BFGS, as libLBFGS-based plmc does, instead of the default variable
 * projection (fields solved exactly by Newton for every trial couplings, L-BFGS over the couplings only:
 * DESIGN.md section 2c).  Same objective, same optimum; the joint path needs ~10-20x more iterations to reach
 * |g|/|x| < epsilon.  With max_iter far below convergence (the reference default 100) the two paths stop at
 * different points.  The Python boundary exposes it as solver="joint" / PLM_HIP_SOLVER=joint / plmc_hip --solver joint.
 * (lambda_h = 0 always runs this path: the per-site Hessians of the field solver are then singular.) */
#define PLM_FLAG_JOINT_LBFGS 16
/* plm_fit / plm_fit_sharded / plm_fit_sharded_rccl with PLM_FLAG_IGNORE_GAPS only: the result arrays come back in the
 * (q-1)-state layout plmc -g itself writes -- fi, hi: [L][q-1]; fij, jij: [L(L-1)/2][q-1][q-1] -- instead of the q-state
 * layout with zero entries for the gap state.  The entries are dropped on the device before the download (the Python host
 * used to drop them with two strided copies of the pair arrays: 30 ms of a -g fit at the headline). */
#define PLM_FLAG_COMPACT_GAPS 1024
/* ---- convention switches ---------------------------------------------------------------------------------------
 * plmc is not available to this project, so a few of its conventions cannot be checked; each
 * is a switch here (same bits in the oracle, oracle/plm_oracle.c), so that a plmc binary on a future host pins the
 * path by choosing bits, not by changing code.  Defaults (bit clear) are the documented ones of DESIGN.md section 2.
 *   PLM_CONV_THRESHOLD_F32      SURVEY App. D-1: the cluster threshold as a float32 plmc would evaluate what run_plmc
 *                               sends it (-t 1-theta): ident >= (float)(1 - (float)(1 - theta)) * L, instead of the
 *                               integer rule ceil(theta * L - 1e-9).  Identical for theta = 0.8 and every L <= 2000.
 *   PLM_CONV_G_GAPS_IDENTICAL   -g: two gaps at a position count as identical in reweighting, as without -g (plmc's
 *                               usage text describes -g as excluding the gap from the POTENTIAL calculations only);
 *                               default: only non-gap matches count.
 *   PLM_CONV_G_UNGAPPED_LENGTH  -g: the identity threshold applies to the positions where both sequences are
 *                               ungapped, ident >= ceil(theta * n_both - 1e-9), instead of the full length.
 *   PLM_CONV_G_FREQ_TOTAL       -g: f_i and f_ij are normalised by N_eff (all sequences), so they sum to the ungapped
 *                               fraction of a site / pair; default: normalised over the ungapped sequences.
 *   PLM_CONV_FN_NO_GAP          SURVEY App. D-3: the Frobenius norm of a coupling block (zero-sum gauge over all q
 *                               states) sums the non-gap states only; default: all q states, as the reference's
 *                               CouplingsModel does (couplings/model.py:792).  No effect with -g (no gap state). */
#define PLM_CONV_THRESHOLD_F32 32
#define PLM_CONV_G_GAPS_IDENTICAL 64
#define PLM_CONV_G_UNGAPPED_LENGTH 128
#define PLM_CONV_G_FREQ_TOTAL 256
#define PLM_CONV_FN_NO_GAP 512
#define PLM_CONV_MASK (32 | 64 | 128 | 256 | 512)

/* Per-iteration progress: the 7 columns of plmc's stderr table that
 * parse_plmc_log() collects (tools.py:59-83): iter time cond fx -loglk ||h|| ||e||.
 * Return 0 to go on; any other value cancels the fit after this iteration (status PLM_STATUS_INTERRUPTED, the result
 * arrays hold the point reached).  This is how the signal handlers a pipeline installs around the stage
 * (evcouplings/utils/pipeline.py:476-545: SIGTERM / SIGINT -> sys.exit) reach a running fit: the plmc child process
 * was killed by the signal itself, an in-process solver has to be told. */
typedef int (*plm_iter_cb)(int32_t iter, double secs, double cond, double fx, double nll,
                           double norm_h, double norm_e, void *user);

/* Exchange step of the site-sharded evaluation: every shard has
 * written `bytes_per_shard` bytes at  dev_buf + shard * bytes_per_shard ; on return the
 * whole buffer (n_shards * bytes_per_shard) must hold every shard's part.  The Python host
 * implements it with torch.distributed.all_gather_into_tensor (RCCL).  Called on the
 * context's stream after a stream synchronise; return 0 on success. */
typedef int (*plm_exchange_cb)(void *dev_buf, size_t bytes_per_shard, int32_t n_shards,
                               int32_t shard, void *user);

/* Collectives of the sharded-state mode, implemented by the host with torch.distributed (RCCL).
 * All buffers are device pointers of this library; counts are BYTES per rank (arrays of n_shards).
 *   PLM_COLL_ALLTOALL      send -> recv with per-rank byte counts, messages in rank order
 *   PLM_COLL_ALLREDUCE_F64 in-place sum over ranks of send_counts[0] bytes of doubles at `send`
 *   PLM_COLL_ALLREDUCE_F32 in-place sum over ranks of send_counts[0] bytes of floats at `send`
 * Called after the context's stream has been synchronised; must return 0 on success with the result
 * visible to later work on that stream. */
#define PLM_COLL_ALLTOALL 1
#define PLM_COLL_ALLREDUCE_F64 2
#define PLM_COLL_ALLREDUCE_F32 3
/*   PLM_COLL_BROADCAST     send_counts[0] bytes at `send` travel from rank recv_counts[0] to every rank (in place).
 *                          Rounds 2-5 assembled the fitted parameters with two broadcasts per shard; since round 6 (a
 *                          shard's entries are no longer contiguous in the canonical order) that is one
 *                          PLM_COLL_ALLREDUCE_F32 of the canonical vector.  The operation stays part of the contract. */
#define PLM_COLL_BROADCAST 4
typedef int (*plm_collective_cb)(int32_t op, void *send, void *recv, const int64_t *send_counts,
                                 const int64_t *recv_counts, int32_t n_shards, int32_t shard, void *user);

typedef struct {
    float *weights;      /* [N]            or NULL */
    float *fi;           /* [L*q]          or NULL */
    float *fij;          /* [L(L-1)/2*q*q] or NULL (i<j blocks, [a][b]) */
    float *hi;           /* [L*q]          or NULL */
    float *jij;          /* [L(L-1)/2*q*q] or NULL */
    float *fn;           /* [L*L]          or NULL */
    float *cn;           /* [L*L]          or NULL */
    float n_eff;
    int32_t iters_done;
    int32_t n_evals;
    int32_t status;      /* PLM_STATUS_* */
    double fx;           /* final objective */
    double seconds_reweight, seconds_marginals, seconds_optimize, seconds_total;
    char status_msg[128];
} plm_result_t;

/* -- library ---------------------------------------------------------------------------- */
int plm_version(void);                 /* PLM_ABI_VERSION */
int plm_device_count(void);            /* visible gfx950 devices, or PLM_EDEVICE */
const char *plm_strerror(int code);
const char *plm_last_error(void);      /* thread-local message of the last failure */

/* -- whole stage: replaces the `plmc` child process (tools.py:266) ----------------------- */
/* Reweight -> marginals -> L-BFGS on the symmetric L2-regularised pseudo-likelihood ->
 * zero-sum gauge / Frobenius / APC scores.  `stream` is a hipStream_t (0 = null stream);
 * `device` the HIP device ordinal.  exchange/exchange_user are only used when
 * problem->n_shards > 1. */
int plm_fit(const plm_problem_t *problem, plm_result_t *result, int device, void *stream,
            plm_iter_cb iter_cb, void *iter_user, plm_exchange_cb exchange, void *exchange_user);

/* Same stage with problem->flags & PLM_FLAG_SHARDED_STATE: every rank calls it with its shard index;
 * all ranks return the same full result arrays. */
int plm_fit_sharded(const plm_problem_t *problem, plm_result_t *result, int device, void *stream,
                    plm_iter_cb iter_cb, void *iter_user, plm_collective_cb collective, void *collective_user);

/* The same with the collectives issued by the library itself: RCCL calls on the context's stream (one process per
 * GPU, ranks = shards), no host callback and no stream synchronisation around a collective.  Rank 0 creates an id
 * (plm_rccl_unique_id), the host hands its PLM_RCCL_ID_BYTES bytes to every rank by its own means (MPI, a file,
 * torch.distributed), then every rank calls plm_fit_sharded_rccl -- or plm_ctx_attach_rccl on a resident context
 * (a collective call: all ranks, each with its GPU current).  librccl is resolved at run time (a copy already in
 * the process first, then /opt/rocm's; PLM_RCCL_LIB overrides).  plm_rccl_selftest runs every collective on a
 * one-rank communicator. */
#define PLM_RCCL_ID_BYTES 128
int plm_rccl_unique_id(void *id_out);
int plm_rccl_runtime_version(void);          /* NCCL version code of the resolved library, 0 if none */
int plm_rccl_selftest(int device, void *stream);
/* Every rank of a prospective communicator: form it from `rccl_id`, exchange one all-to-all and one all-reduce, check
 * the data, destroy it.  PLM_OK on this rank = the library-issued transport works here (dist.py negotiates the ranks'
 * verdicts and falls back to the callback transport if any of them failed). */
int plm_rccl_probe(const void *rccl_id, int32_t nranks, int32_t rank, int device, void *stream);
/* The rank-local half of the probe (device, librccl, buffer, upload) with no communicator call in it.  A rank that
 * failed these steps inside plm_rccl_probe would leave its peers blocked in the communicator call: run this on every
 * rank and agree on the verdicts first. */
int plm_rccl_probe_local(int32_t nranks, int device, void *stream);
int plm_fit_sharded_rccl(const plm_problem_t *problem, plm_result_t *result, int device, void *stream,
                         plm_iter_cb iter_cb, void *iter_user, const void *rccl_id);

/* -- fine-grained, host buffers (parity tests) ------------------------------------------- */
/* plmc sequence reweighting; twin: align/alignment.py:1193-1233.  counts[s] = cluster size. */
int plm_reweight(const int8_t *msa, int32_t n_seqs, int32_t n_sites, double theta_id,
                 int32_t *counts_out);
/* the same with plmc -g semantics and / or PLM_CONV_* switches: flags = PLM_FLAG_IGNORE_GAPS | PLM_CONV_... */
int plm_reweight_ex(const int8_t *msa, int32_t n_seqs, int32_t n_sites, double theta_id, int32_t flags,
                    int32_t *counts_out);
/* plmc marginals; twins: align/alignment.py:1079-1153.  weights need not be normalised. */
int plm_marginals(const int8_t *msa, const float *weights, int32_t n_seqs, int32_t n_sites,
                  int32_t n_states, float *fi_out, float *fij_out);
/* plmc objective + gradient at x (canonical layout). */
int plm_eval(const int8_t *msa, const float *weights, int32_t n_seqs, int32_t n_sites,
             int32_t n_states, double lambda_h, double lambda_j, const float *x, double *fx_out,
             double *nll_out, float *g_out);
/* plmc EC scoring; twins: couplings/model.py:179-233, 744-775, 790-793.  fn/cn dense LxL. */
int plm_scores(const float *jij, int32_t n_sites, int32_t n_states, float *fn_out,
               float *cn_out);
/* the same with PLM_CONV_FN_NO_GAP in flags: state 0 left out of the Frobenius norm */
int plm_scores_ex(const float *jij, int32_t n_sites, int32_t n_states, int32_t flags, float *fn_out, float *cn_out);

/* ---- statistical energies under a fitted model -------------------
 * Replace the numba loops of evcouplings/couplings/model.py that the mutate stage and
 * CouplingsModel.hamiltonians / .single_mut_mat call:
 *   plm_hamiltonians  <->  _hamiltonians(sequences, J_ij, h_i)              model.py:25-60
 *   plm_potentials    <->  the sums of _single_mutant_hamiltonians           model.py:63-109
 * seqs: n x n_sites int8 states 0..q-1 (row-major); x_canonical: the model in the canonical
 * layout (h [L][q], then J pair blocks i<j row-major [q][q] -- the plmc_v2 .model order).
 * energies_out: n x 3 doubles (H, H_J, H_h) with H_J = sum_{i<j} J_ij(x_i,x_j), H_h = sum_i h_i(x_i).
 * potentials_out: n x n_sites x q floats, HJ[s][i][a] = sum_{j != i} J_ij(a, x_sj); the single-mutant
 * matrix of a sequence is dJ(i,a) = HJ[i][a] - HJ[i][x_i], dh(i,a) = h_i(a) - h_i(x_i).            */
int plm_hamiltonians(const int8_t *seqs, int32_t n, int32_t n_sites, int32_t n_states,
                     const float *x_canonical, int device, void *stream, double *energies_out);
int plm_potentials(const int8_t *seqs, int32_t n, int32_t n_sites, int32_t n_states,
                   const float *x_canonical, int device, void *stream, float *potentials_out);

/* ---- mean-field direct coupling analysis ---------------------------
 * Replaces the arithmetic of evcouplings/couplings/mean_field.py:163-222 (MeanFieldDCA.fit: weights,
 * frequencies, pseudo-count regularisation :717-790, covariance matrix :897-940, J = -C^-1 :204-210 and
 * :943-975, fields :977-1014) and :792-893 (direct_information).  All outputs are caller-allocated host
 * buffers; any pointer may be NULL to skip that output.                                                */
typedef struct {
    float *weights;     /* n_seqs                    1 / cluster size                                    */
    float n_eff;
    float *fi;          /* n_sites * q               raw frequencies                                     */
    float *fij;         /* pairs * q * q             raw pair frequencies, i<j blocks row-major           */
    double *hi;         /* n_sites * q               fields                                              */
    double *jij_full;   /* n_sites^2 * q^2           dense couplings [i][j][a][b], diagonal blocks included
                                                     (what reshape_invC_to_4d returns)                   */
    float *jij;         /* pairs * q * q             the i<j blocks in the .model file's precision        */
    double *di;         /* n_sites^2                 direct information, symmetric, zero diagonal         */
} plm_mf_result_t;
int plm_meanfield(const int8_t *msa, int32_t n_seqs, int32_t n_sites, int32_t n_states, double theta_id,
                  double pseudo_count, int device, void *stream, plm_mf_result_t *out);
/* Direct information of every pair from given couplings and frequencies: replaces
 * evcouplings/couplings/mean_field.py:842-893 direct_information(J_ij, f_i).  jij_full: dense
 * [L][L][q][q] doubles (only i<j blocks are read); fi: [L][q] doubles; di_out: [L][L] doubles.       */
int plm_direct_information(const double *jij_full, const double *fi, int32_t n_sites, int32_t n_states,
                           int device, void *stream, double *di_out);

/* ---- alignment statistics of the upstream align stage -----------------------------
 * One pass over an integer-coded alignment (n x n_sites, row-major, states 0..126) for the quantities
 * evcouplings/align/protocol.py:806-1016 (modify_alignment) filters and reports on:
 *   seq_gaps[s]   = #{i : msa[s][i] == gap_state}        Alignment.count("-", axis="seq")  (alignment.py:707-747)
 *   col_gaps[i]   = #{s : msa[s][i] == gap_state}        Alignment.count(gap, axis="pos")
 *   ident[s]      = #{i : msa[s][i] == query[i]}         identities_to_seq(seq, matrix)    (alignment.py:1157-1190)
 * Raw integer counts (the reference normalises on the host); any output pointer may be NULL, query may be NULL
 * when ident is.                                                                                              */
int plm_alignment_stats(const int8_t *msa, int32_t n_seqs, int32_t n_sites, int32_t gap_state, const int8_t *query,
                        int32_t *seq_gaps, int32_t *col_gaps, int32_t *ident, int device, void *stream);

/* ---- alignment input (host code, no device): what plmc does when it reads the file run_plmc names (tools.py:202-262
 * passes only the path).  evcouplings_amd/alignment_io.py states the rules and keeps a pure-Python twin (fallback and
 * test oracle); these two single passes replace its per-line loop and fancy indexing (0.23 -> 0.03 s at the headline).
 *   plm_fasta_split     FASTA / A2M framing of a file image: lines stripped of ASCII whitespace at both ends, empty lines
 *                       skipped, '>' opens a record, a record's data lines are concatenated.  seq_out == NULL: only
 *                       *n_records and *seq_bytes are set (call once to size the buffers, once to fill them);
 *                       hdr_off / hdr_len delimit each id inside buf.  PLM_EINVAL: data before the first header.
 *   plm_encode_columns  out[r][k] = lut256[mat[r][cols[k]]] for a row-major byte matrix (n_rows x width), valid[r] = 1
 *                       iff no entry of the row is negative (a symbol outside the alphabet in a kept column).        */
int plm_fasta_split(const char *buf, int64_t n, int64_t *n_records, int64_t *seq_bytes, int64_t *hdr_off,
                    int32_t *hdr_len, int64_t *seq_len, char *seq_out);
int plm_encode_columns(const uint8_t *mat, int64_t n_rows, int64_t width, const int64_t *cols, int64_t n_cols,
                       const int8_t *lut256, int8_t *out, uint8_t *valid);
/* ... and output (host code): the raw EC file plmc writes and couplings/pairs.py:55-58 reads -- one line per site pair
 * i < j, "index_i A_i index_j A_j 0 cn" with cn as "%.6f"; cn = dense row-major [n_sites][n_sites] doubles.  One buffer, one
 * write instead of 44 850 Python format operations (L = 300); model_io.write_raw_ec_file keeps the Python twin. */
int plm_write_raw_ec_file(const char *path, int32_t n_sites, const int32_t *index_list, const char *target_seq,
                          const double *cn);

/* -- resident-context API (bench / multi-GPU host) ---------------------------------------- */
/* Uploads the alignment once; everything below runs on data resident in HBM. */
int plm_ctx_create(const plm_problem_t *problem, int device, void *stream, plm_ctx_t **out);
void plm_ctx_destroy(plm_ctx_t *ctx);
int plm_ctx_set_exchange(plm_ctx_t *ctx, plm_exchange_cb exchange, void *user);
int plm_ctx_set_collective(plm_ctx_t *ctx, plm_collective_cb collective, void *user);
int plm_ctx_attach_rccl(plm_ctx_t *ctx, const void *rccl_id);   /* ranks = problem.n_shards, rank = problem.shard */
/* change the stop rule / history of later plm_ctx_optimize calls (negative = keep) */
int plm_ctx_set_options(plm_ctx_t *ctx, int32_t max_iter, double epsilon, int32_t lbfgs_m);
/* number of floats of the solver's internal ("native", 16-site blocked) parameter vector */
int64_t plm_ctx_native_size(const plm_ctx_t *ctx);
int plm_ctx_reweight(plm_ctx_t *ctx);                       /* fills device weights, N_eff */
int plm_ctx_set_weights(plm_ctx_t *ctx, const float *weights_host);
int plm_ctx_get_weights(plm_ctx_t *ctx, float *weights_host, int32_t *counts_host, float *n_eff);
int plm_ctx_marginals(plm_ctx_t *ctx, float *fi_host, float *fij_host);
int plm_ctx_set_x(plm_ctx_t *ctx, const float *x_canonical_host);   /* NULL = start point */
int plm_ctx_get_x(plm_ctx_t *ctx, float *x_canonical_host);
int plm_ctx_get_g(plm_ctx_t *ctx, float *g_canonical_host);
/* one objective+gradient evaluation at the resident x; asynchronous unless fx_out != NULL */
int plm_ctx_eval(plm_ctx_t *ctx, double *fx_out, double *nll_out);
int plm_ctx_optimize(plm_ctx_t *ctx, plm_iter_cb cb, void *user, plm_result_t *result);
int plm_ctx_scores(plm_ctx_t *ctx, float *fn_host, float *cn_host);
/* average HIP-event duration (ms) of each kernel of the evaluation pipeline over `reps`
 * evaluations: out_ms[PLM_K_*]; used by bench.py for the roofline block. */
#define PLM_K_EXPAND 0
#define PLM_K_FORWARD 1
#define PLM_K_BACKWARD 2
#define PLM_K_ASSEMBLE 3
#define PLM_K_TOTAL 4
#define PLM_K_REWEIGHT 5
#define PLM_K_FIELDS 6      /* variable-projection fit: Newton passes on the fields + residual pass (0 otherwise) */
#define PLM_K_FORWARD_ACCURATE 7   /* the forward GEMM's accurate instantiation (f64 outer sums): last iterations, plm_eval */
#define PLM_K_LBFGS_VECTOR 8       /* the L-BFGS vector kernels of one iteration (m = 6) on this context's share of the state */
#define PLM_K_COUNT 9
int plm_ctx_time_kernels(plm_ctx_t *ctx, int32_t reps, float *out_ms /* [PLM_K_COUNT] */);
/* The two kinds of position of the field solver's chain, timed alone on this context's site blocks (after
 * plm_ctx_time_kernels, whose forward GEMM left the potentials): out_ms[0] = a Hessian position (pass with gradient,
 * diagonal and sampled Hessian sums + per-site Newton step), out_ms[1] = the closing position (pass that writes the
 * residual planes + gradient sums, norm check).  scripts/shard_compute.py prices a chain of p passes as
 * (p - 1) out_ms[0] + out_ms[1]. */
int plm_ctx_time_field_positions(plm_ctx_t *ctx, int32_t reps, float *out_ms /* [2] */);
/* Field-solver statistics of the last plm_ctx_optimize on this context (variable-projection fits; zeros otherwise),
 * measured with HIP events on the context's stream around the solver of every evaluation: bench.py reports the average
 * field-solver time per evaluation of its timed window from these. */
#define PLM_S_EVALS 0          /* evaluations whose field solver was timed */
#define PLM_S_FIELD_MS 1       /* total milliseconds between the forward GEMM and the backward GEMM of those evaluations */
#define PLM_S_PASSES 2         /* passes over the stored potentials made by the solver's chains (the last, residual-only pass not counted) */
#define PLM_S_CHAIN_SHORT 3    /* evaluations whose chain ran out of positions and had to be continued by the host */
#define PLM_S_GEMM_EVALS 4     /* evaluations (plain arithmetic, chain done at the first look) whose two GEMMs were timed */
#define PLM_S_FWD_MS 5         /* total milliseconds of their forward GEMMs (HIP events on the context's stream, inside the fit) */
#define PLM_S_BWD_MS 6         /* total milliseconds of their backward GEMMs */
#define PLM_S_COUNT 7
int plm_ctx_solver_stats(plm_ctx_t *ctx, double *out /* [PLM_S_COUNT] */);

/* -- host arithmetic exposed for tests (no device) ------------------------------------------ */
/* Two-loop recursion of L-BFGS in coefficient space over a ring of m history slots, of which the `stored` slots
 * before `end` (ring order, newest = end - 1) are live: direction p = sum_j cs[j] s_j + D^-1 (sum_j cy[j] y_j + cg g).
 * All arrays are indexed by PHYSICAL slot: SY[i*m+j] = s_i.y_j, YDY[i*m+j] = y_i.D^-1 y_j, Sg[i] = s_i.g,
 * YDg[i] = y_i.D^-1 g, gDg = g.D^-1 g; *dg = g.p.  This is the code plm_ctx_optimize runs every iteration. */
void plm_lbfgs_coefficients(int m, int stored, int end, const double *SY, const double *YDY, const double *Sg,
                            const double *YDg, double gDg, double *cs, double *cy, double *cg, double *dg);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PLM_HIP_H */
