// plm_host.cpp -- context, evaluation pipeline, L-BFGS driver and the C ABI of libplm_hip.so
// (include/plm_hip.h).  Replaces the plmc child process of evcouplings/couplings/tools.py:266.
// No CPU fallback exists: every entry point needs a gfx950 device and fails with
// PLM_EDEVICE otherwise.
#include "../../include/plm_hip.h"
#include "plm_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace
// message recorder for the other translation units of the library
int plm_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return fail(code, "%s", buf);
}
int plm_meanfield_device(const float *fi, const float *fij, int L, int q, double pseudo_count, hipStream_t st,
                         double *hi, double *jfull, float *jpairs, double *di);   // plm_meanfield.hip
int plm_direct_information_device(const double *jdense, const double *rfi, int L, int q, hipStream_t st, double *di);
namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(e__ == hipErrorOutOfMemory ? PLM_ENOMEM : PLM_EDEVICE, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e__), __FILE__, __LINE__);                        \
    } while (0)
#define PLM_TRY(expr)                  \
    do {                               \
        int rc__ = (expr);             \
        if (rc__ != PLM_OK) return rc__; \
    } while (0)

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int check_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PLM_EDEVICE, "no HIP device visible (libplm_hip has no CPU path)");
    if (device < 0 || device >= n) return fail(PLM_EINVAL, "device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PLM_EDEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    return PLM_OK;
}

// identity threshold of the reweighting: the integer rule, or what a float32 plmc makes of
// the `-t 1-theta` that run_plmc sends it (tools.py:236-239)
int cluster_threshold(double theta_id, int L, int conv) {
    if (conv & PLM_CONV_THRESHOLD_F32) {
        const float t = (float)(1.0 - theta_id);
        const float need = (1.0f - t) * (float)L;
        return (int)std::ceil(need);
    }
    return (int)std::ceil(theta_id * (double)L - 1e-9);
}

}  // namespace
// every environment knob of the library, read once per context (nothing on the evaluation path calls getenv)
PlmOptions plm_options_from_env() {
    PlmOptions o;
    if (const char *e = getenv("PLM_BWD_PLANES")) o.bwd_planes = atoi(e) == 4 ? 4 : 3;
    if (const char *e = getenv("PLM_KSPLIT")) o.ksplit = std::max(1, atoi(e));
    if (const char *e = getenv("PLM_JEXP_BIAS")) o.jexp_bias = atoi(e);
    if (const char *e = getenv("PLM_BWD_KERNEL")) o.bwd_kernel = atoi(e) ? 1 : 0;
    if (const char *e = getenv("PLM_FWD_KERNEL")) o.fwd_kernel = atoi(e) ? 1 : 0;
    if (const char *e = getenv("PLM_FWD_ACCURATE")) o.fwd_mode = atoi(e) ? 1 : 0;
    if (const char *e = getenv("PLM_VP_FLOOR")) o.vp_floor = atof(e);
    if (const char *e = getenv("PLM_VP_REL")) o.vp_rel = atof(e);
    if (const char *e = getenv("PLM_VP_HESS_POS")) o.vp_hess_pos = atoi(e);
    if (const char *e = getenv("PLM_ACC_FACTOR")) o.acc_factor = atof(e);
    o.debug = getenv("PLM_DEBUG") != nullptr;
    o.debug_vp = getenv("PLM_DEBUG_VP") != nullptr;
    return o;
}
namespace {

// K split of the backward GEMM for a given number of digit planes.  A launch has tiles * nplanes * ks workgroups (one
// digit plane per workgroup), each over 1/ks of the K steps; with integer accumulation the split changes no result, only
// the balance: the launch runs in ceil(tiles * nplanes * ks / 256) rounds of 1/ks of a full-K workgroup each (XCD
// granularity makes it slightly worse), and every extra set of partial slabs costs k_assemble two more reads of it.
// Cost in units of one full-K workgroup (a 128-sequence K step is ~1.3 us; the partial slabs reach k_assemble largely
// through L2 / the 256 MB Infinity Cache -- measured at the headline: ks 1 -> 2 adds 0.11 ms to k_assemble for 0.98 GB
// more reads and takes 0.20 ms off k_bwd -- hence the 12 TB/s of `beta`).
int pick_ksplit(const PlmDims &d, const PlmOptions &opt, int nplanes, int *out) {
    // int32 accumulators: a K range of ceil(nst128 / ks) steps of 128 sequences contributes at most 128 (|one-hot value|) *
    // 128 (|digit|: signed base-256 digits reach -128) per sequence, which must stay below 2^31 -- K ranges of at most
    // 1023 steps (130 944 sequences).  More sequences than 16 such ranges simply get more ranges.
    const int ks_min = std::max(1, (d.nst128 + 1022) / 1023);
    const int ks_max = std::max(ks_min, std::max(1, std::min(16, d.nst128 / 8)));
    if ((int64_t)((d.nst128 + ks_min - 1) / ks_min) * PLM_BWD_KSTEP * 128 * 128 >= ((int64_t)1 << 31))
        return fail(PLM_EUNSUPPORTED, "%d sequences: no admissible K split of the backward GEMM", d.N);
    const double beta = (2.0 * nplanes * (double)d.nmf * d.nnfl * 1024.0 / 12e12) / ((double)d.nst128 * 1.3e-6);
    int best = ks_min;
    double best_cost = 1e300;
    for (int ks = ks_min; ks <= ks_max; ks++) {
        const int groups_per_xcd = (d.ncol_tiles * nplanes * ks + 7) / 8;     // busiest XCD
        const double cost = (double)((groups_per_xcd * d.nrow_tiles + 31) / 32) / ks + beta * ks;
        if (cost < best_cost * 0.98) { best_cost = cost; best = ks; }
    }
    // measurement knob: force the K split of the backward GEMM (results are identical for every value)
    if (opt.ksplit) best = std::max(ks_min, std::min(ks_max, opt.ksplit));
    *out = best;
    return PLM_OK;
}

int make_dims(const plm_problem_t &p, const PlmOptions &opt, PlmDims *out) {
    PlmDims d;
    memset(&d, 0, sizeof d);
    if (p.n_seqs <= 0 || p.n_sites <= 1) return fail(PLM_EINVAL, "need n_seqs > 0 and n_sites > 1");
    if (!plm_q_supported(p.n_states))
        return fail(PLM_EUNSUPPORTED, "alphabet size %d outside 2..32", p.n_states);
    const int nshards = p.n_shards > 0 ? p.n_shards : 1;
    if (p.shard < 0 || p.shard >= nshards) return fail(PLM_EINVAL, "shard %d outside 0..%d", p.shard, nshards - 1);
    d.N = p.n_seqs;
    d.L = p.n_sites;
    d.Qc = p.n_states;                 // the problem's alphabet: stride of the canonical arrays at the API
    d.Q = plm_q_template(p.n_states);  // the size the kernels run at; states Qc..Q-1 are dead padding
    d.Np = (d.N + PLM_SEQ_TILE - 1) / PLM_SEQ_TILE * PLM_SEQ_TILE;
    d.nb16 = (d.L + 15) / 16;
    d.Lp16 = d.nb16 * 16;
    d.nu = (d.L + 31) / 32;
    d.Lp32 = d.nu * 32;
    d.nksteps = d.nu * PLM_FWD_TILES(d.Q);
    d.nssteps = d.Np / 32;
    d.nst128 = d.Np / PLM_BWD_KSTEP;
    // digit planes of the backward GEMM: 24-bit residuals by default, 32-bit for fits that must converge below 1e-4
    // (the quantisation noise of three planes is ~5e-5 |x| at the headline); PLM_BWD_PLANES = 3 | 4 overrides
    d.nplanes = (p.epsilon > 0 && p.epsilon < 1e-4) ? 4 : 3;
    if (opt.bwd_planes) d.nplanes = opt.bwd_planes;
    d.jexp_bias = opt.jexp_bias;
    const float qmax = d.nplanes == 4 ? PLM_R_QMAX4 : PLM_R_QMAX3;
    d.rscale = qmax;                    // for unit weights; plm_ctx_set_weights sets the pair for the weights in use
    d.gscale = 1.0f / (qmax * (float)PLM_BWD_ONEHOT_VALUE);
    d.nstiles = d.Np / PLM_SEQ_TILE;
    plm_pick_tile(d.Q, &d.FM, &d.FN);
    d.nmf = d.nb16 * d.Q + d.FM;
    d.nshards = nshards;
    d.shard = p.shard;
    d.blk_per_shard = (d.nb16 + nshards - 1) / nshards;
    d.shard_base = d.nb16 / nshards;
    d.shard_rem = d.nb16 % nshards;
    d.b16_lo = plm_shard_lo(d, d.shard);
    d.b16_hi = d.b16_lo + plm_shard_cnt(d, d.shard);
    d.nnfl = d.blk_per_shard * d.Q;
    d.nrow_tiles = (d.nmf + 4 * d.FM - 1) / (4 * d.FM);
    d.ncol_tiles = (d.nnfl + 2 * d.FN - 1) / (2 * d.FN);
    d.fwd_w = (d.Q == 21 && opt.fwd_kernel != 0) ? 1 : 0;   // PLM_FWD_KERNEL=0: k_fwd (A/B runs)
    d.bwd_w = (d.Q == 21 && opt.bwd_kernel != 0) ? 1 : 0;   // PLM_BWD_KERNEL=0: the compiler-allocated k_bwd (A/B runs)
    if (d.bwd_w) d.ncol_tiles = (d.nnfl + PLM_BWDW_COLS - 1) / PLM_BWDW_COLS;
    PLM_TRY(pick_ksplit(d, opt, d.nplanes, &d.ksplit));
    d.nbp = (int64_t)d.nb16 * (d.nb16 + 1) / 2;
    d.nh_pad = ((int64_t)d.L * d.Q + 255) / 256 * 256;
    d.n_native = d.nh_pad + d.nbp * d.Q * d.Q * 256;
    d.n_canon = (int64_t)d.L * d.Qc + (int64_t)d.L * (d.L - 1) / 2 * d.Qc * d.Qc;
    d.gap_mode = (p.flags & PLM_FLAG_IGNORE_GAPS) ? 1 : 0;
    d.conv = p.flags & PLM_CONV_MASK;
    d.theta = p.theta_id;
    d.sharded = ((p.flags & PLM_FLAG_SHARDED_STATE) && nshards > 1) ? 1 : 0;
    d.own_lo = d.sharded ? d.b16_lo : 0;
    d.own_hi = d.sharded ? d.b16_hi : d.nb16;
    d.nblk_own = d.own_hi - d.own_lo;
    d.h_site0 = d.own_lo * 16;
    d.bp_base = 0;
    for (int p = 0; p < PLM_MAX_SHARDS; p++) d.own_base[p] = d.oth_base[p] = -1;
    d.nx_halo = d.ng_halo = 0;
    if (d.sharded) {
        if (nshards > PLM_MAX_SHARDS) return fail(PLM_EUNSUPPORTED, "sharded-state mode takes at most %d shards", PLM_MAX_SHARDS);
        d.ntri = d.nblk_own * (d.nblk_own + 1) / 2;
        // the two halves of the rectangle shared with every other shard, each kind numbered in partner order (plm_internal.h)
        for (int p = 0; p < nshards; p++) {
            if (p == d.shard) continue;
            const int64_t mine = plm_half_blocks(d, p), theirs = (int64_t)d.nblk_own * plm_shard_cnt(d, p) - mine;
            d.own_base[p] = (int)d.ng_halo;
            d.ng_halo += mine;
            d.oth_base[p] = (int)d.nx_halo;
            d.nx_halo += theirs;
        }
        d.np_own = d.ntri + d.ng_halo;
    } else {
        d.np_own = d.nbp;
        d.ntri = (int)d.nbp;
    }
    {
        const int site_end = std::min(d.L, d.own_hi * 16);
        const int64_t nhl = (int64_t)std::max(0, site_end - d.h_site0) * d.Q;
        d.nh_pad_l = d.sharded ? std::max<int64_t>(256, (nhl + 255) / 256 * 256) : d.nh_pad;
    }
    d.n_local = d.nh_pad_l + d.np_own * d.Q * d.Q * 256;
    if (d.gap_mode && d.Qc < 3) return fail(PLM_EINVAL, "ignore_gaps needs at least 2 non-gap states");
    *out = d;
    return PLM_OK;
}

}  // namespace

// Two-loop recursion of L-BFGS (Nocedal 1980) in coefficient space: with the Gram matrix of {s_j, y_j, g} the direction
//     p = sum_j cs[j] s_j + D^-1 (sum_j cy[j] y_j + cg g)      (D^-1 = I without preconditioning)
// needs no pass over the vectors.  The history is a ring of m physical slots; the LIVE pairs are the `stored` slots before
// `end` in ring order (newest = end - 1).  That is not always the physical range 0..stored-1: when a noise-dominated pair
// is skipped on a full ring (plm_ctx_optimize), the dead slot is `end` -- wherever the ring stands.  Inputs and outputs
// are indexed by PHYSICAL slot: SY[i*m+j] = s_i.y_j, YDY[i*m+j] = y_i.D^-1 y_j, Sg[i] = s_i.g, YDg[i] = y_i.D^-1 g,
// gDg = g.D^-1 g; cs / cy get zeros in dead slots; *dg = g.p.  Exported for the host-side test (tests/test_host_layer.py).
extern "C" void plm_lbfgs_coefficients(int m, int stored, int end, const double *SY, const double *YDY, const double *Sg,
                                       const double *YDg, double gDg, double *cs, double *cy, double *cg_out,
                                       double *dg_out) {
    std::vector<double> alpha(m, 0.0);
    std::vector<int> live(stored);                       // newest first
    for (int i = 0; i < stored; i++) live[i] = (end + m - 1 - i) % m;
    for (int j = 0; j < m; j++) cs[j] = cy[j] = 0.0;
    double cg = -1.0;
    // first loop (newest -> oldest): q = cg g + sum cy y lives in the plain space, only s_i.q is needed (cs is still zero)
    for (int i = 0; i < stored; i++) {
        const int j = live[i];
        double v = cg * Sg[j];
        for (int k : live) v += cy[k] * SY[j * m + k];
        alpha[j] = v / SY[j * m + j];
        cy[j] -= alpha[j];
    }
    if (stored > 0) {
        const int newest = live[0];
        const double gamma = SY[newest * m + newest] / YDY[newest * m + newest];
        cg *= gamma;
        for (int k : live) cy[k] *= gamma;
    }
    // second loop (oldest -> newest): r = sum cs s + D^-1 (cg g + sum cy y)
    for (int i = stored - 1; i >= 0; i--) {
        const int j = live[i];
        double v = cg * YDg[j];
        for (int k : live) v += cs[k] * SY[k * m + j] + cy[k] * YDY[j * m + k];
        cs[j] += alpha[j] - v / SY[j * m + j];
    }
    double dg = cg * gDg;
    for (int k : live) dg += cs[k] * Sg[k] + cy[k] * YDg[k];
    *cg_out = cg;
    *dg_out = dg;
}

struct plm_ctx {
    plm_problem_t prob;
    PlmOptions opt;            // environment knobs, read once at creation
    PlmDims d;
    int device = 0;
    hipStream_t st = nullptr;
    plm_exchange_cb exchange = nullptr;
    void *exchange_user = nullptr;
    // device buffers
    int8_t *msa_rm = nullptr, *msa_cm = nullptr;
    float *w = nullptr;
    int32_t *counts = nullptr;
    void *Bt = nullptr, *Rt = nullptr;
    int32_t *G = nullptr;      // digit-plane / K-range partial slabs of the backward GEMM (local, exact integer sums)
    float *gather = nullptr;   // exchange buffer [nshards][slab] (replicated multi-shard mode only)
    plm_collective_cb collective = nullptr;   // sharded-state mode: collectives through the host ...
    void *collective_user = nullptr;
    PlmRccl *rccl = nullptr;                  // ... or issued here, on st (plm_ctx_attach_rccl)
    float *xhalo = nullptr, *ghalo = nullptr, *xsend = nullptr, *gsend = nullptr;
    std::vector<int64_t> x_send, x_recv, g_send, g_recv;   // all-to-all byte counts per rank
    double *fx_part = nullptr, *reg_part = nullptr, *dot_scratch = nullptr, *scal = nullptr;
    uint32_t *maxbits = nullptr;
    int32_t *jexp = nullptr;
    float *x = nullptr, *g = nullptr, *xp = nullptr, *gp = nullptr, *dir = nullptr, *hist = nullptr;
    float *xa = nullptr, *ga = nullptr;   // anchor point of the next curvature pair when pairs had to be skipped (lazy)
    float *canon = nullptr;    // canonical-layout staging (n_canon floats, + L*L for fn)
    float *pair_n2 = nullptr;  // squared norms of the coupling blocks (group regulariser only)
    float *dinv = nullptr;     // H0 diagonal of the preconditioned L-BFGS (n_local floats), built by plm_ctx_optimize
    // variable-projection fit: coupling part of the conditionals, Newton statistics, per-site gradient norms
    float *hj = nullptr, *hpart = nullptr;
    double *gpart = nullptr, *dpart = nullptr;
    double *hg2 = nullptr, *hinv = nullptr, *h64 = nullptr;   // h64: the field solver's f64 copies of the fields (two buffers)
    double *hcnt = nullptr;    // [local site][Q] weighted state counts (plm_launch_site_counts), refreshed with the weights
    bool hcnt_valid = false;
    int *vp_flag = nullptr;    // device-side state of the field solver's chain (PlmVpState)
    int vp_hess_age = -1;      // evaluations since the cached inverse Hessians were refreshed (-1: none exist yet)
    // host side of the chain (ctx_eval_vp_enqueue / ctx_eval_vp_finish)
    int vp_c_prev = 3;         // Newton steps the previous evaluation's chain took before it converged
    int vp_pos = 0;            // chain positions enqueued for the current evaluation
    unsigned vp_hess_mask = 0; // ... and which of them (bit = position) were enqueued with Hessian sums
    int vp_extra = 0;          // continuations of the current evaluation (a chain that ran out of positions)
    double vp_last_gh2 = 0;    // squared field-gradient norm when the current evaluation's chain was last looked at
    double vp_floor2 = 0;      // squared noise floor of the field gradient's f32 sums (set by plm_ctx_optimize)
    hipEvent_t vp_ev[2] = {nullptr, nullptr};   // around the field solver of the last enqueued evaluation
    hipEvent_t gemm_ev[3] = {nullptr, nullptr, nullptr};   // in front of the forward GEMM / in front of / behind the backward GEMM
    bool vp_ev_pending = false;
    // statistics of the field solver since the last plm_ctx_optimize began (plm_ctx_solver_stats)
    double stat_field_ms = 0, stat_passes = 0, stat_fwd_ms = 0, stat_bwd_ms = 0;
    int stat_gemm_evals = 0;
    int stat_field_evals = 0, stat_chain_short = 0;
    int hist_m = 0;
    double *h_scal = nullptr;  // pinned host scalars
    bool have_weights = false;
    float wmax = 0;            // largest weight in use (scale of the fixed-point residuals)
    // (x, g) and these values belong together: set when an optimisation ends, cleared by everything that
    // changes x, the weights or the scratch use of g -- lets a follow-up plm_ctx_optimize (a resumed fit)
    // start from the known point instead of re-evaluating it
    bool eval_valid = false;
    bool hj_at_x = false;    // ... and the stored potentials HJ are those of that point (not of a rejected trial)
    double last_fx = 0, last_nll = 0, last_gh2 = 0;
    bool eval_vp = false;      // ... and g is the gradient of the reduced (variable-projection) objective
    // Forward GEMM of the next evaluation: the plain instantiation (f32 accumulation over the whole K range) or the
    // accurate one (f64 outer sums, ~1.5x the time).  plm_ctx_eval always asks for the accurate one; a fit switches to it
    // for its last iterations (plm_ctx_optimize); PLM_FWD_ACCURATE = 0 | 1 forces one of them (measurements).
    bool fwd_accurate = false;
    bool eval_accurate = false;   // the valid (x, g, f) above came from the accurate instantiation
    // digit planes of the residuals: the plain evaluation runs planes_base (3, or 4 when the stop rule is below 1e-4),
    // the accurate one always 4 (with three, residuals below 2^-24 of the largest weight round to zero -- for every such
    // sequence alike: 1.7e-4 |x| at N = 100 000 against 0.5e-4 with four).  Rt and G are allocated for the larger.
    int planes_base = 3, ksplit_base = 1, ksplit4 = 1;
    double n_eff = 0;
    int n_evals = 0;
    std::vector<float> h_fi;   // L*q, kept for the start point

    int n_fx_part() const { return d.nstiles * (d.b16_hi - d.b16_lo); }
};

namespace {

// Precision of the next evaluations: plain (f16 forward GEMM, __expf, planes_base digit planes) or accurate (exact forward
// GEMM, exact softmax arguments, four digit planes).  PLM_BWD_PLANES pins the plane count (measurements).
void ctx_set_accurate(plm_ctx *c, bool on) {
    c->fwd_accurate = on;
    const int planes = c->opt.bwd_planes ? c->opt.bwd_planes : (on ? 4 : c->planes_base);
    c->d.nplanes = planes;
    c->d.ksplit = planes == 4 ? c->ksplit4 : c->ksplit_base;
    const float qmax = planes == 4 ? PLM_R_QMAX4 : PLM_R_QMAX3, wmax = c->wmax > 0 ? c->wmax : 1.f;
    c->d.rscale = qmax / wmax;
    c->d.gscale = wmax / (qmax * (float)PLM_BWD_ONEHOT_VALUE);   // the one-hot operand of k_bwd carries -128
}

template <typename T> int dalloc(T **p, size_t n_elems) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n_elems, 1) * sizeof(T));
    if (e != hipSuccess)
        return fail(PLM_ENOMEM, "hipMalloc of %zu bytes failed: %s", n_elems * sizeof(T), hipGetErrorString(e));
    *p = (T *)q;
    return PLM_OK;
}

int ctx_alloc_lbfgs(plm_ctx *c, int m) {
    if (c->xp && c->hist_m >= m) return PLM_OK;
    const size_t n = (size_t)c->d.n_local;
    if (!c->xp) {
        PLM_TRY(dalloc(&c->xp, n));
        PLM_TRY(dalloc(&c->gp, n));
        PLM_TRY(dalloc(&c->dir, n));
    }
    if (c->hist) hipFree(c->hist);
    c->hist = nullptr;
    PLM_TRY(dalloc(&c->hist, n * 2 * (size_t)m));
    c->hist_m = m;
    return PLM_OK;
}

// enqueue one objective+gradient evaluation at c->x -> c->g, scal[0] = fx, scal[1] = nll
// one collective of the sharded-state mode (stream is synchronised first: the host runs it with RCCL)
int ctx_collective(plm_ctx *c, int op, void *send, void *recv, const int64_t *scounts, const int64_t *rcounts) {
    if (c->rccl) {   // stream-ordered: nothing to wait for
        if (plm_rccl_collective(c->rccl, op, send, recv, scounts, rcounts, c->st) != 0)
            return fail(PLM_ECALLBACK, "RCCL collective failed (op %d): %s", op, plm_rccl_error());
        return PLM_OK;
    }
    if (!c->collective) return fail(PLM_EINVAL, "sharded-state mode needs a collective callback or an RCCL communicator");
    HIP_TRY(hipStreamSynchronize(c->st));
    if (c->collective(op, send, recv, scounts, rcounts, c->d.nshards, c->d.shard, c->collective_user) != 0)
        return fail(PLM_ECALLBACK, "collective callback failed (op %d)", op);
    return PLM_OK;
}
// in-place sum over the shards of c->scal[first .. first+count)
int ctx_allreduce_scalars(plm_ctx *c, int first, int count) {
    if (!c->d.sharded) return PLM_OK;
    const int64_t bytes = (int64_t)sizeof(double) * count;
    return ctx_collective(c, PLM_COLL_ALLREDUCE_F64, c->scal + first, c->scal + first, &bytes, &bytes);
}

// forward half of an evaluation at the fields and couplings of c->x (joint L-BFGS, plm_ctx_eval): the GEMM stores the
// coupling potentials, one pass of the field kernel turns them into residuals (Rt) and -log P partials
// group regulariser (lambda_group > 0): squared norms of the coupling blocks of the current x, for k_assemble
int group_norms(plm_ctx *c) {
    if (!(c->prob.lambda_group > 0)) return PLM_OK;
    if (!c->pair_n2) PLM_TRY(dalloc(&c->pair_n2, (size_t)std::max<int64_t>(1, c->d.np_own) * 256));
    HIP_TRY(plm_launch_pair_norms(c->d, c->x, c->pair_n2, c->st));
    return PLM_OK;
}
int vp_alloc(plm_ctx *c);
int forward_at_x(plm_ctx *c) {
    const PlmDims &d = c->d;
    PLM_TRY(vp_alloc(c));
    HIP_TRY(plm_launch_forward_store(d, c->msa_rm, c->Bt, c->jexp, c->hj, c->fwd_accurate, c->st));
    HIP_TRY(plm_launch_h64_init(d, c->x, c->h64, c->st));
    HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 1, 0, c->fwd_accurate, c->Rt, c->fx_part, nullptr, nullptr, nullptr, nullptr, PLM_VP_ALWAYS, c->st));
    return PLM_OK;
}

// The coupling message of the sharded-state mode needs no packing: the rectangles this shard owns lie behind the triangle
// of its local vector, in partner order, each exactly as the partner expects it in its halo.
float *x_rectangles(plm_ctx *c) { return c->x + c->d.nh_pad_l + (size_t)c->d.ntri * PLM_BLOCK_FLOATS(c->d); }

// sharded-state evaluation: local x (+ halo from the partners) -> local g; scal[0..1] = this shard's
// part of fx and nll (summed over shards by the caller together with its dot products)
int ctx_eval_enqueue_sharded(plm_ctx *c) {
    const PlmDims &d = c->d;
    PLM_TRY(ctx_collective(c, PLM_COLL_ALLTOALL, x_rectangles(c), c->xhalo, c->x_send.data(), c->x_recv.data()));
    HIP_TRY(plm_launch_maxabs2(c->x + d.nh_pad_l, d.n_local - d.nh_pad_l, c->xhalo,
                               d.nx_halo * (int64_t)PLM_BLOCK_FLOATS(d), c->maxbits, c->jexp, d.jexp_bias, c->st));
    HIP_TRY(plm_launch_expand(d, c->x, c->xhalo, c->jexp, c->Bt, c->fwd_accurate ? 1 : 0, c->st));
    PLM_TRY(forward_at_x(c));
    if (d.nblk_own > 0) HIP_TRY(plm_launch_backward(d, c->msa_cm, c->Rt, c->G, nullptr, c->st));
    HIP_TRY(plm_launch_pack_g(d, c->G, c->gsend, c->st));
    PLM_TRY(ctx_collective(c, PLM_COLL_ALLTOALL, c->gsend, c->ghalo, c->g_send.data(), c->g_recv.data()));
    PLM_TRY(group_norms(c));
    HIP_TRY(plm_launch_assemble(d, c->G, d.ksplit, c->ghalo, c->x, c->g, c->prob.lambda_h, c->prob.lambda_j,
                                c->reg_part, 0, 0.f, c->pair_n2, (float)c->prob.lambda_group, c->st));
    HIP_TRY(plm_launch_finish_fx(d, c->fx_part, c->n_fx_part(), nullptr, 0, c->reg_part, plm_reg_parts(d), c->scal,
                                 c->st));
    c->n_evals++;
    return PLM_OK;
}

int ctx_eval_enqueue(plm_ctx *c) {
    const PlmDims &d = c->d;
    if (d.sharded) return ctx_eval_enqueue_sharded(c);
    HIP_TRY(plm_launch_maxabs(d, c->x, c->maxbits, c->jexp, c->st));
    HIP_TRY(plm_launch_expand(d, c->x, nullptr, c->jexp, c->Bt, c->fwd_accurate ? 1 : 0, c->st));
    PLM_TRY(forward_at_x(c));
    HIP_TRY(plm_launch_backward(d, c->msa_cm, c->Rt, c->G, nullptr, c->st));
    const void *Gsrc = c->G;
    int ks_count = d.ksplit, n_shard_nll = 0;
    const double *shard_nll = nullptr;
    if (d.nshards > 1) {
        const size_t slab = plm_slab_bytes(d);
        char *mine = (char *)c->gather + (size_t)d.shard * slab;
        HIP_TRY(plm_launch_slab_reduce(d, c->G, (float *)mine, c->st));
        HIP_TRY(plm_launch_partial_sum(c->fx_part, c->n_fx_part(), (double *)(mine + slab - 256), c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        if (!c->exchange) return fail(PLM_EINVAL, "n_shards > 1 but no exchange callback set");
        if (c->exchange(c->gather, slab, d.nshards, d.shard, c->exchange_user) != 0)
            return fail(PLM_ECALLBACK, "exchange callback failed");
        Gsrc = c->gather;
        ks_count = 0;            // float slabs, planes and K ranges already combined by k_slab_reduce
        n_shard_nll = d.nshards;
        shard_nll = (const double *)((char *)c->gather + slab - 256);
    }
    PLM_TRY(group_norms(c));
    HIP_TRY(plm_launch_assemble(d, Gsrc, ks_count, nullptr, c->x, c->g, c->prob.lambda_h, c->prob.lambda_j,
                                c->reg_part, 0, 0.f, c->pair_n2, (float)c->prob.lambda_group, c->st));
    HIP_TRY(plm_launch_finish_fx(d, c->fx_part, c->n_fx_part(), shard_nll, n_shard_nll, c->reg_part,
                                 plm_reg_parts(d), c->scal, c->st));
    c->n_evals++;
    return PLM_OK;
}

// ---- variable projection ------------------------------------------------------------
// The fields enter the objective only through per-site, strictly convex subproblems (for fixed couplings), so
// the fit runs L-BFGS on  F(J) = min_h f(h, J):  an evaluation computes the coupling part HJ of every conditional
// with the forward GEMM, solves the fields by Newton (k_hpass / k_hsolve, HJ streamed from HBM once per step),
// and returns dF/dJ = df/dJ at (h*(J), J) (envelope theorem).  x's field part is overwritten with h*(J); the
// field part of g is zero.  scal[SL_GH2 = 5] = squared gradient norm of the field subproblems at the returned point.
bool vp_enabled(const plm_ctx *c) {
    // lambda_h = 0: the per-site Hessians are singular along the softmax gauge direction (the Newton solver has no
    // pivoting) -- such a problem runs the joint path
    return !(c->prob.flags & PLM_FLAG_JOINT_LBFGS) && (c->d.nshards == 1 || c->d.sharded) && c->prob.lambda_h > 0;
}
int vp_alloc(plm_ctx *c) {
    if (c->hj) return PLM_OK;
    // per-site buffers are indexed by i - h_site0 over the LOCAL FIELD PART: the shard's own sites in sharded-state
    // mode, but all L sites in the replicated multi-shard mode (own_lo = 0, own_hi = nb16), where the shard's own
    // column blocks are only a slice of them (h64 is filled and read for every site by forward_at_x)
    const size_t nsites = (size_t)std::max(std::max(1, (c->d.b16_hi - c->d.b16_lo) * 16),
                                           std::min(c->d.L, c->d.own_hi * 16) - c->d.h_site0);
    PLM_TRY(dalloc((char **)&c->hj, plm_hj_bytes(c->d)));
    PLM_TRY(dalloc((char **)&c->hpart, plm_hpart_bytes(c->d)));
    PLM_TRY(dalloc((char **)&c->gpart, plm_gpart_bytes(c->d)));
    PLM_TRY(dalloc((char **)&c->dpart, plm_gpart_bytes(c->d)));      // diagonal second-order sums: same shape
    PLM_TRY(dalloc(&c->hg2, nsites));
    PLM_TRY(dalloc(&c->hinv, nsites * c->d.Q * c->d.Q));
    PLM_TRY(dalloc(&c->h64, 2 * plm_h64_stride(c->d)));
    PLM_TRY(dalloc(&c->hcnt, plm_h64_stride(c->d)));
    c->hcnt_valid = false;
    PLM_TRY(dalloc((char **)&c->vp_flag, sizeof(PlmVpState)));
    HIP_TRY(hipMemsetAsync(c->vp_flag, 0, sizeof(PlmVpState), c->st));
    for (auto &e : c->vp_ev) HIP_TRY(hipEventCreate(&e));
    for (auto &e : c->gemm_ev) HIP_TRY(hipEventCreate(&e));
    c->vp_hess_age = -1;
    return PLM_OK;
}
// stage 1: forward GEMM (couplings of x) -> HJ
int vp_counts(plm_ctx *c) {      // weighted state counts per site: once per set of weights
    if (!c->hcnt_valid) {
        HIP_TRY(plm_launch_site_counts(c->d, c->msa_cm, c->w, c->hcnt, c->st));
        c->hcnt_valid = true;
    }
    return PLM_OK;
}
int vp_stage1(plm_ctx *c) {
    const PlmDims &d = c->d;
    PLM_TRY(vp_counts(c));
    if (d.sharded) {
        PLM_TRY(ctx_collective(c, PLM_COLL_ALLTOALL, x_rectangles(c), c->xhalo, c->x_send.data(), c->x_recv.data()));
        HIP_TRY(plm_launch_maxabs2(c->x + d.nh_pad_l, d.n_local - d.nh_pad_l, c->xhalo,
                                   d.nx_halo * (int64_t)PLM_BLOCK_FLOATS(d), c->maxbits, c->jexp, d.jexp_bias, c->st));
        HIP_TRY(plm_launch_expand(d, c->x, c->xhalo, c->jexp, c->Bt, c->fwd_accurate ? 1 : 0, c->st));
    } else {
        HIP_TRY(plm_launch_maxabs(d, c->x, c->maxbits, c->jexp, c->st));
        HIP_TRY(plm_launch_expand(d, c->x, nullptr, c->jexp, c->Bt, c->fwd_accurate ? 1 : 0, c->st));
    }
    if (c->gemm_ev[0]) HIP_TRY(hipEventRecord(c->gemm_ev[0], c->st));
    HIP_TRY(plm_launch_forward_store(d, c->msa_rm, c->Bt, c->jexp, c->hj, c->fwd_accurate, c->st));
    HIP_TRY(plm_launch_h64_init(d, c->x, c->h64, c->st));
    return PLM_OK;
}
// One Newton step on the fields with the cached (or, refresh: recomputed) inverse Hessians + the residual pass at the
// result: the "fields" leg of plm_ctx_time_kernels.  (The fit's evaluations run the chain below.)
int vp_counts(plm_ctx *c);
int vp_step_and_residuals(plm_ctx *c, bool refresh) {
    const PlmDims &d = c->d;
    PLM_TRY(vp_counts(c));
    const int full = (refresh || c->vp_hess_age < 0) ? 1 : 0;
    HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 0, full ? 2 : 1, c->fwd_accurate, nullptr, nullptr, c->hpart,
                             c->gpart, c->dpart, nullptr, PLM_VP_ALWAYS, c->st));
    HIP_TRY(plm_launch_hsolve(d, c->hpart, c->gpart, full, c->x, c->h64, c->prob.lambda_h, 1, c->hinv, c->hg2, c->scal + 5, 0.0,
                              0.0, nullptr, 0, c->hcnt, c->dpart, c->st));
    if (full) c->vp_hess_age = 0;
    HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 1, 1, c->fwd_accurate, c->Rt, c->fx_part, c->hpart, c->gpart,
                             nullptr, nullptr, PLM_VP_ALWAYS, c->st));
    HIP_TRY(plm_launch_hsolve(d, c->hpart, c->gpart, 0, c->x, c->h64, c->prob.lambda_h, 0, c->hinv, c->hg2, c->scal + 5, 0.0,
                              0.0, nullptr, 0, c->hcnt, c->dpart, c->st));
    return PLM_OK;
}
// stage 2: the field solver as ONE chain of launches (round 5; rounds 2-4 ran it in rounds with a host round trip
// between them).  A chain position = a pass over the stored potentials in one of two roles -- statistics (gradient
// sums, at the positions `hess` says also the sampled Hessian sums), or, when the chain predicts that this pass will be
// the last, gradient sums + residual planes + -log P partials -- then the per-site Newton step (k_hsolve: sites within
// their share of the tolerance stay where they are) and k_vp_check (chain done? role of the next pass?).  Launches
// behind the end of the chain return at once (~2 us each).  What the CPU lab (tests/probes/field_solver_lab.py) and the
// PLM_DEBUG_VP traces of round 4 showed: with the Hessians refreshed only at the first step of a round the early
// evaluations of a fit needed 9-13 passes and a third of them ended stalled far from the tolerance; a fresh (sampled)
// Hessian at EVERY step converges at ~0.04 per step from anywhere the cap allows, and a Hessian pass costs 0.41 ms
// against 0.33 ms.
int vp_chain(plm_ctx *c, int npos, int hess_upto, int expected_last, double tol2) {
    const PlmDims &d = c->d;
    for (int k = 0; k < npos; k++) {
        const int pos = c->vp_pos++;
        // Hessian sums: while the previous evaluation says the solver is still far (pos < hess_upto), whenever no
        // inverse exists yet, and at every position past the expected end (the cached inverses were not good enough)
        const bool hess = pos < hess_upto || pos > expected_last || c->vp_hess_age < 0;
        HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 0, hess ? 2 : 1, c->fwd_accurate, nullptr, nullptr, c->hpart,
                                 c->gpart, c->dpart, c->vp_flag, PLM_VP_PASS, c->st));
        HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 1, 1, c->fwd_accurate, c->Rt, c->fx_part, c->hpart, c->gpart,
                                 nullptr, c->vp_flag, PLM_VP_PASS_RT, c->st));
        HIP_TRY(plm_launch_hsolve(d, c->hpart, c->gpart, hess ? 2 : 0, c->x, c->h64, c->prob.lambda_h, 1, c->hinv, c->hg2,
                                  c->scal + 5, tol2, c->vp_floor2, c->vp_flag, 1, c->hcnt, c->dpart, c->st));
        // (whether the device RUNS this position is known only after the chain: ctx_eval_vp_finish resets the age of the
        // cached inverses if a Hessian position was among the passes executed -- ADVICE r5: resetting it here made the
        // periodic refresh dead code)
        if (hess && pos < 32) c->vp_hess_mask |= 1u << pos;
        if (hess && c->vp_hess_age < 0) c->vp_hess_age = 0;
    }
    return PLM_OK;
}
// stage 3: backward GEMM, gradient of the reduced objective, objective value
// gout / mode: where the gradient goes and which one -- mode 2 into c->g for the fit (reduced objective: field part
// zero), mode 0 into a scratch vector for the certificate of the shipped point (joint gradient, fields included)
int vp_stage3(plm_ctx *c, bool conditional, float *gout = nullptr, int mode = 2) {
    const PlmDims &d = c->d;
    if (!gout) gout = c->g;
    // conditional: the backward GEMM runs only if the chain in front of it is done (PlmVpState::done is the first int
    // of the state) -- otherwise the host continues the chain and enqueues stage 3 again.  Not when sharded: every rank
    // takes part in the gradient-halo exchange of every stage 3, and the ranks' chains end independently.
    if (c->gemm_ev[1]) HIP_TRY(hipEventRecord(c->gemm_ev[1], c->st));
    if (d.nblk_own > 0 || !d.sharded)
        HIP_TRY(plm_launch_backward(d, c->msa_cm, c->Rt, c->G, (conditional && !d.sharded) ? c->vp_flag : nullptr, c->st));
    if (c->gemm_ev[2]) HIP_TRY(hipEventRecord(c->gemm_ev[2], c->st));
    if (d.sharded) {
        HIP_TRY(plm_launch_pack_g(d, c->G, c->gsend, c->st));
        PLM_TRY(ctx_collective(c, PLM_COLL_ALLTOALL, c->gsend, c->ghalo, c->g_send.data(), c->g_recv.data()));
    }
    // mode 2: gradient of the reduced objective (field part zero), regulariser sums as usual
    PLM_TRY(group_norms(c));
    HIP_TRY(plm_launch_assemble(d, c->G, d.ksplit, d.sharded ? c->ghalo : nullptr, c->x, gout, c->prob.lambda_h,
                                c->prob.lambda_j, c->reg_part, mode, 0.f, c->pair_n2, (float)c->prob.lambda_group, c->st));
    if (mode == 2)
        HIP_TRY(plm_launch_finish_fx(d, c->fx_part, c->n_fx_part(), nullptr, 0, c->reg_part, plm_reg_parts(d), c->scal,
                                     c->st));
    return PLM_OK;
}
// The whole evaluation, enqueued without a host round trip: forward GEMM, the field solver's chain, the residual pass
// (skipped on the device when the pass that ended the chain wrote the planes), backward GEMM (only if the chain is
// done), assemble.  The caller appends what it needs of the result (dot products, Gram rows), synchronises ONCE with
// scalar slots 5..7 among the values it fetches (sharded: all-reduces them), and calls ctx_eval_vp_finish.
int ctx_eval_vp_enqueue(plm_ctx *c, double tol2) {
    const PlmDims &d = c->d;
    PLM_TRY(vp_stage1(c));
    if (c->vp_ev_pending) c->vp_ev_pending = false;     // (an evaluation nobody waited for: its events are re-recorded)
    HIP_TRY(hipEventRecord(c->vp_ev[0], c->st));
    // a chain expected to end on its first pass (late in a fit: the L-BFGS extrapolation of the fields is within the
    // tolerance) starts in the residual-writing role -- unless the cached inverse Hessians are due for their periodic
    // refresh (every 32 evaluations: a pass in that role carries no Hessian sums)
    const int c_prev = c->vp_c_prev;
    if (c->vp_hess_age >= 0) c->vp_hess_age++;
    const bool stale = c->vp_hess_age < 0 || c->vp_hess_age >= 32;
    HIP_TRY(plm_launch_vp_reset(c->vp_flag, (c_prev == 0 && !stale) ? 1 : 0, c->scal + 5, c->st));
    c->vp_pos = 0;
    c->vp_extra = 0;
    c->vp_hess_mask = 0;
    c->vp_last_gh2 = INFINITY;
    // fresh Hessian sums at every position before the expected last one (measured, gpurun_out/r5c9: Hessians at the
    // first position only -> 9.8 passes per evaluation in the bench window instead of 4.6, at the first two -> 5.9)
    int hess_upto = c_prev >= 2 ? c_prev : (stale ? 1 : 0);
    if (c->opt.vp_hess_pos >= 0) hess_upto = std::min(hess_upto, std::max(c->opt.vp_hess_pos, c->vp_hess_age < 0 ? 1 : 0));
    PLM_TRY(vp_chain(c, std::min(14, c_prev + 3), hess_upto, c_prev, tol2));
    HIP_TRY(plm_launch_fields_to_x(d, c->h64, c->vp_flag, c->x, c->st));
    HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 1, 0, c->fwd_accurate, c->Rt, c->fx_part, nullptr, nullptr, nullptr,
                             c->vp_flag, PLM_VP_FINAL, c->st));
    HIP_TRY(hipEventRecord(c->vp_ev[1], c->st));
    c->vp_ev_pending = true;
    PLM_TRY(vp_stage3(c, true));
    c->n_evals++;
    return PLM_OK;
}
// After the caller's synchronisation (h_scal[5..7]: squared field-gradient norm, passes, verdict -- summed over the
// shards).  *again = false: the evaluation is complete, *gh2_out = the norm left by the solver.  *again = true: the chain
// ran out of positions before it was done (or, sharded, some rank's did): more of it, the residual pass and stage 3 have
// been enqueued again -- the caller repeats its own launches and synchronises once more.
int ctx_eval_vp_finish(plm_ctx *c, double tol2, bool *again, double *gh2_out) {
    const PlmDims &d = c->d;
    const int nsh = d.sharded ? d.nshards : 1;
    const double gh2 = c->h_scal[5];
    const int passes = (int)std::lround(c->h_scal[6] / nsh);
    const bool done = c->h_scal[7] > nsh - 0.5;
    if (c->vp_ev_pending) {      // the events lie in front of the synchronisation the caller just made
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->vp_ev[0], c->vp_ev[1]) == hipSuccess) {
            c->stat_field_ms += ms;
            if (c->vp_extra == 0) c->stat_field_evals++;
        }
        // the two GEMMs of this evaluation, as the fit ran them: only evaluations in the plain arithmetic whose chain was
        // done at the first look (the conditional backward GEMM did its work; h64_init rides with the forward GEMM: 5 us)
        float f = 0, b = 0;
        if (done && c->vp_extra == 0 && !c->fwd_accurate && hipEventElapsedTime(&f, c->gemm_ev[0], c->vp_ev[0]) == hipSuccess &&
            hipEventElapsedTime(&b, c->gemm_ev[1], c->gemm_ev[2]) == hipSuccess) {
            c->stat_fwd_ms += f;
            c->stat_bwd_ms += b;
            c->stat_gemm_evals++;
        }
        c->vp_ev_pending = false;
    }
    *gh2_out = gh2;
    *again = false;
    if (c->opt.debug_vp) {
        fprintf(stderr, "[plm vp] eval %d%s: %d passes (expected %d steps), done=%d |g_h|=%.3e tol=%.3e |", c->n_evals - 1,
                c->vp_extra ? " (continued)" : "", passes, c->vp_c_prev, (int)done, std::sqrt(gh2), std::sqrt(tol2));
        PlmVpState S;      // per-pass history (debug only: one more small copy)
        if (hipMemcpy(&S, c->vp_flag, sizeof S, hipMemcpyDeviceToHost) == hipSuccess)
            for (int k = 0; k < std::min(S.passes, PLM_VP_HIST); k++) {
                const double open_sites = std::floor(S.hist[k] / 1e9);
                fprintf(stderr, " %.2e/%d/%d", std::sqrt(S.hist[k] - open_sites * 1e9), (int)open_sites, S.hist_loud[k]);
            }
        fprintf(stderr, "\n");
    }
    if (c->vp_extra == 0) c->stat_passes += passes;
    // the cached inverse Hessians are fresh if one of the passes that RAN carried Hessian sums (a position in the
    // residual-writing role does not: at most one per chain, the last)
    if (passes > 0 && (c->vp_hess_mask & ((passes >= 32 ? 0xffffffffu : ((1u << passes) - 1u))))) c->vp_hess_age = 0;
    if (done && c->vp_extra == 0) {
        c->vp_c_prev = std::max(0, passes - 1);
        return PLM_OK;
    }
    if (c->vp_extra > 0) {
        // a continuation always ends with an unconditional residual pass + stage 3: the evaluation is complete.  Go on
        // only while the solver still makes progress towards a tolerance it has not met (bounded)
        const bool stalled = !(gh2 > tol2) || gh2 > 0.25 * c->vp_last_gh2 || c->vp_extra >= 3 || !std::isfinite(gh2);
        if (done || stalled) {
            c->vp_c_prev = std::min(11, std::max(c->vp_pos - 1, 0));
            return PLM_OK;
        }
    } else {
        c->stat_chain_short++;
    }
    // the chain ran out of positions: six more with fresh Hessians, then the residual pass with the gradient norm at its
    // fields (unconditional) and stage 3 (unconditional)
    c->vp_last_gh2 = gh2;
    c->vp_extra++;
    HIP_TRY(hipEventRecord(c->vp_ev[0], c->st));
    HIP_TRY(hipMemsetAsync(c->scal + 5, 0, 3 * sizeof(double), c->st));
    if (gh2 > tol2) PLM_TRY(vp_chain(c, 6, 1 << 30, -1, tol2));
    HIP_TRY(plm_launch_fields_to_x(d, c->h64, c->vp_flag, c->x, c->st));
    HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 1, 1, c->fwd_accurate, c->Rt, c->fx_part, c->hpart, c->gpart,
                             nullptr, c->vp_flag, PLM_VP_ALWAYS, c->st));
    // norm only (update = 0, not a chain position: k_vp_check writes the sum alone -- passes / verdict of the continued
    // chain stay), at the chain's current fields
    HIP_TRY(plm_launch_hsolve(d, c->hpart, c->gpart, 0, c->x, c->h64, c->prob.lambda_h, 0, c->hinv, c->hg2, c->scal + 5, 0.0,
                              0.0, c->vp_flag, 0, c->hcnt, c->dpart, c->st));
    HIP_TRY(hipEventRecord(c->vp_ev[1], c->st));
    c->vp_ev_pending = true;
    PLM_TRY(vp_stage3(c, false));
    *again = true;
    return PLM_OK;
}
int fetch_scalars(plm_ctx *c, int first, int count) {
    HIP_TRY(hipMemcpyAsync(c->h_scal + first, c->scal + first, sizeof(double) * count, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    return PLM_OK;
}

// scal slots: 0 fx, 1 nll, 2.. dot results
int dots(plm_ctx *c, int npairs, const float *const *a, const float *const *b, int64_t n, int slot) {
    HIP_TRY(plm_launch_dots(npairs, a, b, n, c->dot_scratch, c->scal + slot, c->st));
    return PLM_OK;
}

// ---- More'-Thuente trial-step update (More' & Thuente 1994, sec. 4); scalar host code -------
int mt_update(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy, double *stp, double fp,
              double dp, double tmin, double tmax, int *brackt) {
    if (*brackt && (*stp <= std::min(*stx, *sty) || *stp >= std::max(*stx, *sty))) return -1;
    if (*dx * (*stp - *stx) >= 0.0 || tmax < tmin) return -1;
    const double sgnd = dp * (*dx / std::fabs(*dx));
    double stpf, stpc, stpq, gamma, p, q, r, s, theta;
    bool bound;
    if (fp > *fx) {
        bound = true;
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = std::max(std::fabs(theta), std::max(std::fabs(*dx), std::fabs(dp)));
        gamma = s * std::sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma = -gamma;
        p = (gamma - *dx) + theta;
        q = ((gamma - *dx) + gamma) + dp;
        r = p / q;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
        stpf = (std::fabs(stpc - *stx) < std::fabs(stpq - *stx)) ? stpc : stpc + (stpq - stpc) / 2.0;
        *brackt = 1;
    } else if (sgnd < 0.0) {
        bound = false;
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = std::max(std::fabs(theta), std::max(std::fabs(*dx), std::fabs(dp)));
        gamma = s * std::sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + *dx;
        r = p / q;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        stpf = (std::fabs(stpc - *stp) > std::fabs(stpq - *stp)) ? stpc : stpq;
        *brackt = 1;
    } else if (std::fabs(dp) < std::fabs(*dx)) {
        bound = true;
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = std::max(std::fabs(theta), std::max(std::fabs(*dx), std::fabs(dp)));
        gamma = s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = (gamma + (*dx - dp)) + gamma;
        r = p / q;
        if (r < 0.0 && gamma != 0.0) stpc = *stp + r * (*stx - *stp);
        else stpc = (*stp > *stx) ? tmax : tmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt) stpf = (std::fabs(*stp - stpc) < std::fabs(*stp - stpq)) ? stpc : stpq;
        else stpf = (std::fabs(*stp - stpc) > std::fabs(*stp - stpq)) ? stpc : stpq;
    } else {
        bound = false;
        if (*brackt) {
            theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            s = std::max(std::fabs(theta), std::max(std::fabs(*dy), std::fabs(dp)));
            gamma = s * std::sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + *dy;
            r = p / q;
            stpf = *stp + r * (*sty - *stp);
        } else {
            stpf = (*stp > *stx) ? tmax : tmin;
        }
    }
    if (fp > *fx) {
        *sty = *stp; *fy = fp; *dy = dp;
    } else {
        if (sgnd < 0.0) { *sty = *stx; *fy = *fx; *dy = *dx; }
        *stx = *stp; *fx = fp; *dx = dp;
    }
    stpf = std::max(tmin, std::min(tmax, stpf));
    *stp = stpf;
    if (*brackt && bound) {
        const double lim = *stx + 0.66 * (*sty - *stx);
        *stp = (*sty > *stx) ? std::min(lim, *stp) : std::max(lim, *stp);
    }
    return 0;
}

const char *status_text(int status) {
    switch (status) {
    case PLM_STATUS_CONVERGED: return "converged (|g|/max(1,|x|) below epsilon)";
    case PLM_STATUS_MAXITER: return "maximum number of iterations reached";
    case PLM_STATUS_INTERRUPTED: return "interrupted by the caller";
    default: return "line search could not improve further (treated as converged to precision)";
    }
}

int set_start_point(plm_ctx *c) {
    // h_i(a) = log(f_i(a) + 1/N_eff) minus the site mean, J = 0
    const PlmDims &d = c->d;
    if (c->h_fi.empty()) return fail(PLM_EINVAL, "start point needs single-site frequencies: run marginals first");
    std::vector<float> h((size_t)d.nh_pad_l, 0.f);
    const int a0 = d.gap_mode;   // gap mode: state 0 is not a model state, its field stays 0
    for (int i = d.h_site0; i < std::min(d.L, d.own_hi * 16); i++) {
        double mean = 0;
        std::vector<double> v(d.Q);
        for (int a = a0; a < d.Qc; a++) {      // h_fi is a canonical array (Qc states), h the native field part (Q)
            v[a] = std::log((double)c->h_fi[(size_t)i * d.Qc + a] + 1.0 / c->n_eff);
            mean += v[a];
        }
        mean /= (d.Qc - a0);
        for (int a = a0; a < d.Qc; a++) h[(size_t)(i - d.h_site0) * d.Q + a] = (float)(v[a] - mean);
    }
    c->eval_valid = false;
    HIP_TRY(hipMemsetAsync(c->x, 0, sizeof(float) * d.n_local, c->st));
    HIP_TRY(hipMemcpyAsync(c->x, h.data(), sizeof(float) * d.nh_pad_l, hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    return PLM_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int plm_version(void) { return PLM_ABI_VERSION; }

int plm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return fail(PLM_EDEVICE, "hipGetDeviceCount failed");
    return n;
}

const char *plm_strerror(int code) {
    switch (code) {
    case PLM_OK: return "ok";
    case PLM_EINVAL: return "invalid argument";
    case PLM_ENOMEM: return "out of memory";
    case PLM_EDEVICE: return "HIP device error";
    case PLM_EUNSUPPORTED: return "unsupported alphabet size";
    case PLM_ENUMERIC: return "non-finite value in objective";
    case PLM_ECALLBACK: return "exchange callback failed";
    default: return "unknown error";
    }
}

const char *plm_last_error(void) { return g_err.c_str(); }

void plm_ctx_destroy(plm_ctx_t *c) {
    if (!c) return;
    hipSetDevice(c->device);
    for (auto &e : c->gemm_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->vp_ev)
        if (e) (void)hipEventDestroy(e);
    void *bufs[] = {c->msa_rm, c->msa_cm, c->w, c->counts, c->Bt, c->Rt, c->G, c->gather, c->fx_part, c->reg_part,
                    c->dot_scratch, c->scal, c->maxbits, c->jexp, c->x, c->g, c->xp, c->gp, c->dir, c->hist,
                    c->canon, c->xhalo, c->ghalo, c->xsend, c->gsend, c->dinv, c->hj, c->hpart, c->gpart, c->dpart, c->hg2, c->hinv, c->h64, c->hcnt, c->vp_flag,
                    c->xa, c->ga, c->pair_n2};
    for (void *b : bufs)
        if (b) hipFree(b);
    if (c->h_scal) hipHostFree(c->h_scal);
    if (c->rccl) {
        hipStreamSynchronize(c->st);
        plm_rccl_destroy(c->rccl);
    }
    delete c;
}

int plm_ctx_create(const plm_problem_t *prob, int device, void *stream, plm_ctx_t **out) {
    if (!prob || !out || !prob->msa) return fail(PLM_EINVAL, "NULL problem / msa / out");
    *out = nullptr;
    PLM_TRY(check_device(device));
    const PlmOptions opt = plm_options_from_env();
    PlmDims d;
    PLM_TRY(make_dims(*prob, opt, &d));
    if (!(prob->theta_id >= 0.0 && prob->theta_id <= 1.0)) return fail(PLM_EINVAL, "theta_id must be in [0,1]");
    if (prob->lambda_h < 0 || prob->lambda_j < 0 || prob->lambda_group < 0)
        return fail(PLM_EINVAL, "negative regularisation strength");
    for (size_t k = 0; k < (size_t)d.N * d.L; k++)
        if (prob->msa[k] < 0 || prob->msa[k] >= d.Qc)
            return fail(PLM_EINVAL, "msa[%zu] = %d outside 0..%d", k, (int)prob->msa[k], d.Qc - 1);
    plm_ctx *c = new plm_ctx();
    c->prob = *prob;
    c->prob.msa = nullptr;  // host pointer not retained
    c->opt = opt;
    c->d = d;
    c->planes_base = d.nplanes;
    c->ksplit_base = d.ksplit;
    {
        int ks4 = 1;
        const int rc4 = pick_ksplit(d, opt, 4, &ks4);
        if (rc4 != PLM_OK) { delete c; return rc4; }
        c->ksplit4 = ks4;
    }
    // Rt and G for the larger of the two precisions (accurate evaluations run four digit planes)
    PlmDims dmax = d;
    dmax.nplanes = opt.bwd_planes ? opt.bwd_planes : 4;
    dmax.ksplit = std::max(c->ksplit_base, c->ksplit4);
    c->device = device;
    c->st = (hipStream_t)stream;
    int rc = PLM_OK;
    auto bail = [&](int code) {
        plm_ctx_destroy(c);
        return code;
    };
    // padded host image of the alignment, row-major; the column-major image (+ the "ones" block) is made from it on the
    // device (round 5: the strided host transposition was 25 ms of every context at the headline)
    const size_t rm_rows = (size_t)d.Np + 32, cm_rows = (size_t)(d.nb16 + 1) * 16;
    std::vector<int8_t> rm(rm_rows * d.Lp32, (int8_t)PLM_PAD_STATE);
    for (int s = 0; s < d.N; s++) memcpy(&rm[(size_t)s * d.Lp32], prob->msa + (size_t)s * d.L, d.L);
    const size_t cm_size = cm_rows * d.Np;
    {
        // Refuse a problem the device cannot hold BEFORE allocating: the runtime grants allocations beyond the HBM and the
        // failure would surface at some later launch.  What a fit holds: the operands of the two GEMMs, the stored
        // potentials, x, g, the canonical image and the optimiser's 2 m + 3 vectors (m = 6 unless the caller asks for more).
        const int m = std::min(20, prob->lbfgs_m > 0 ? prob->lbfgs_m : 6);
        const double need = (double)rm.size() + (double)cm_size + (double)plm_bt_bytes(d) + (double)plm_rt_bytes(dmax) +
                            (double)plm_g_bytes(dmax) + (vp_enabled(c) ? (double)plm_hj_bytes(d) : 0.0) +
                            4.0 * ((double)d.n_local * (2 * m + 5) + (double)d.n_canon + (double)d.L * d.L);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (double)free_b)
            return bail(fail(PLM_ENOMEM, "the problem needs %.1f GB of device memory, %.1f GB are free (of %.1f GB)", need / 1e9,
                             free_b / 1e9, total_b / 1e9));
    }
    if ((rc = dalloc(&c->msa_rm, rm.size())) || (rc = dalloc(&c->msa_cm, cm_size)) ||
        (rc = dalloc(&c->w, (size_t)d.Np)) || (rc = dalloc(&c->counts, (size_t)d.Np)) ||
        (rc = dalloc((char **)&c->Bt, plm_bt_bytes(d))) || (rc = dalloc((char **)&c->Rt, plm_rt_bytes(dmax))) ||
        (rc = dalloc((char **)&c->G, plm_g_bytes(dmax))) || (rc = dalloc(&c->fx_part, (size_t)c->n_fx_part())) ||
        (rc = dalloc(&c->reg_part, (size_t)plm_reg_parts(d))) ||
        (rc = dalloc(&c->dot_scratch, (size_t)4 * PLM_MAX_BASIS * PLM_DOT_BLOCKS)) ||
        (rc = dalloc(&c->scal, (size_t)256)) ||
        (rc = dalloc(&c->maxbits, (size_t)1)) || (rc = dalloc(&c->jexp, (size_t)1)) ||
        (rc = dalloc(&c->x, (size_t)d.n_local)) || (rc = dalloc(&c->g, (size_t)d.n_local)) ||
        (rc = dalloc(&c->canon, (size_t)d.n_canon + (size_t)d.L * d.L)))
        return bail(rc);
    if (d.nshards > 1 && !d.sharded && (rc = dalloc((char **)&c->gather, plm_slab_bytes(d) * d.nshards)))
        return bail(rc);
    if (d.sharded) {
        const size_t blk = PLM_BLOCK_FLOATS(d);
        if ((rc = dalloc(&c->xhalo, (size_t)d.nx_halo * blk)) || (rc = dalloc(&c->gsend, (size_t)d.nx_halo * blk)) ||
            (rc = dalloc(&c->ghalo, (size_t)d.ng_halo * blk)))
            return bail(rc);
        // all-to-all byte counts: with shard r this shard shares the rectangle (own blocks) x (blocks of r); each of the
        // two owns half of its rows (plm_pair_owner), sends the couplings of its half and receives the partner's gradient
        // fragments for it
        c->x_send.assign(d.nshards, 0); c->x_recv.assign(d.nshards, 0);
        c->g_send.assign(d.nshards, 0); c->g_recv.assign(d.nshards, 0);
        for (int r = 0; r < d.nshards; r++) {
            if (r == d.shard) continue;
            const int64_t mine = plm_half_blocks(d, r), theirs = (int64_t)d.nblk_own * plm_shard_cnt(d, r) - mine;
            c->x_send[r] = c->g_recv[r] = mine * (int64_t)blk * 4;
            c->x_recv[r] = c->g_send[r] = theirs * (int64_t)blk * 4;
        }
    }
    hipError_t e;
    if ((e = hipHostMalloc((void **)&c->h_scal, sizeof(double) * 256)) != hipSuccess)
        return bail(fail(PLM_ENOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e)));
#define CT(expr)                                                                                  \
    if ((e = (expr)) != hipSuccess)                                                               \
        return bail(fail(e == hipErrorOutOfMemory ? PLM_ENOMEM : PLM_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e)));
    // (a problem beyond the device may get its hipMallocs granted and meet hipErrorOutOfMemory only here, at the first
    // launch: tests/test_gpu_parity.py::test_a_problem_that_does_not_fit_the_device_is_refused_with_enomem)
    CT(hipMemcpyAsync(c->msa_rm, rm.data(), rm.size(), hipMemcpyHostToDevice, c->st));
    CT(plm_launch_msa_columns(d, c->msa_rm, c->msa_cm, (int)cm_rows, c->st));
    CT(hipMemsetAsync(c->w, 0, sizeof(float) * d.Np, c->st));
    CT(hipMemsetAsync(c->Rt, 0, plm_rt_bytes(dmax), c->st));
    CT(hipMemsetAsync(c->x, 0, sizeof(float) * d.n_local, c->st));
    CT(hipMemsetAsync(c->g, 0, sizeof(float) * d.n_local, c->st));
    CT(hipMemsetAsync(c->scal, 0, sizeof(double) * 256, c->st));   // slots nobody writes still travel in all-reduces
    if (c->gather) CT(hipMemsetAsync(c->gather, 0, plm_slab_bytes(d) * d.nshards, c->st));
    CT(hipStreamSynchronize(c->st));
#undef CT
    *out = c;
    return PLM_OK;
}

int plm_ctx_set_exchange(plm_ctx_t *c, plm_exchange_cb exchange, void *user) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    c->exchange = exchange;
    c->exchange_user = user;
    return PLM_OK;
}

int plm_ctx_set_collective(plm_ctx_t *c, plm_collective_cb collective, void *user) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    c->collective = collective;
    c->collective_user = user;
    return PLM_OK;
}

int plm_ctx_attach_rccl(plm_ctx_t *c, const void *rccl_id) {
    if (!c || !rccl_id) return fail(PLM_EINVAL, "NULL ctx / id");
    if (c->rccl) return fail(PLM_EINVAL, "context already has a communicator");
    HIP_TRY(hipSetDevice(c->device));
    if (plm_rccl_init(rccl_id, c->d.nshards, c->d.shard, &c->rccl) != 0)
        return fail(PLM_ECALLBACK, "RCCL communicator of %d ranks: %s", c->d.nshards, plm_rccl_error());
    return PLM_OK;
}

int plm_rccl_unique_id(void *id_out) {
    if (!id_out) return fail(PLM_EINVAL, "NULL id");
    if (plm_rccl_id(id_out) != 0) return fail(PLM_ECALLBACK, "%s", plm_rccl_error());
    return PLM_OK;
}
int plm_rccl_runtime_version(void) { return plm_rccl_version(); }

// every collective of the sharded-state mode on a one-rank communicator (the only kind a single GPU can form)
int plm_rccl_selftest(int device, void *stream) {
    PLM_TRY(check_device(device));
    hipStream_t st = (hipStream_t)stream;
    unsigned char id[PLM_RCCL_ID_BYTES];
    PLM_TRY(plm_rccl_unique_id(id));
    PlmRccl *r = nullptr;
    if (plm_rccl_init(id, 1, 0, &r) != 0) return fail(PLM_ECALLBACK, "%s", plm_rccl_error());
    const int n = 1000;
    double *buf = nullptr;
    int rc = dalloc(&buf, (size_t)2 * n);
    std::vector<double> h(2 * n, 0.0), back(2 * n, -1.0);
    for (int k = 0; k < n; k++) h[k] = 0.5 * k - 7.0;
    const int64_t bytes = (int64_t)sizeof(double) * n, root = 0;
    auto run = [&](int op, void *s, void *d) {
        if (rc == PLM_OK && plm_rccl_collective(r, op, s, d, &bytes, op == PLM_COLL_BROADCAST ? &root : &bytes, st) != 0)
            rc = fail(PLM_ECALLBACK, "op %d: %s", op, plm_rccl_error());
    };
    if (rc == PLM_OK && hipMemcpyAsync(buf, h.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, st) != hipSuccess)
        rc = fail(PLM_EDEVICE, "upload failed");
    run(PLM_COLL_ALLREDUCE_F64, buf, buf);
    run(PLM_COLL_ALLREDUCE_F32, buf, buf);
    run(PLM_COLL_BROADCAST, buf, nullptr);
    run(PLM_COLL_ALLTOALL, buf, buf + n);
    if (rc == PLM_OK && (hipMemcpyAsync(back.data(), buf, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, st) != hipSuccess ||
                         hipStreamSynchronize(st) != hipSuccess))
        rc = fail(PLM_EDEVICE, "download failed");
    if (rc == PLM_OK)
        for (int k = 0; k < n; k++)
            if (back[k] != h[k] || back[n + k] != h[k]) {
                rc = fail(PLM_ECALLBACK, "one-rank collectives changed the data at %d", k);
                break;
            }
    if (buf) hipFree(buf);
    plm_rccl_destroy(r);
    return rc;
}

// A communicator of `nranks` formed from `rccl_id`, one all-reduce and one all-to-all (1 KB to every peer) across it, the
// communicator destroyed again: every rank calls this before a multi-GPU job commits to the library-issued transport
// (evcouplings_amd/dist.py negotiates the outcome over torch.distributed and falls back to host callbacks otherwise).
int plm_rccl_probe(const void *rccl_id, int32_t nranks, int32_t rank, int device, void *stream) {
    if (!rccl_id || nranks < 1 || rank < 0 || rank >= nranks) return fail(PLM_EINVAL, "bad communicator shape");
    PLM_TRY(check_device(device));
    hipStream_t st = (hipStream_t)stream;
    // Everything that can fail on this rank ALONE comes first: a rank that returned from here before the
    // communicator call would leave its peers blocked inside ncclCommInitRank.  The host agrees on the outcome of
    // plm_rccl_probe_local (the same steps) before any rank enters this function; here they can only fail again if the
    // device was lost in between.
    const int per = 128;                                   // doubles per peer message
    double *buf = nullptr;
    PLM_TRY(dalloc(&buf, (size_t)(2 * nranks + 1) * per));
    std::vector<double> h((size_t)(2 * nranks + 1) * per, 0.0);
    for (int k = 0; k < nranks; k++)
        for (int j = 0; j < per; j++) h[(size_t)k * per + j] = 1000.0 * rank + k;          // message to rank k
    h[(size_t)2 * nranks * per] = 1.0 + rank;
    const std::vector<int64_t> counts((size_t)nranks, (int64_t)sizeof(double) * per);
    const int64_t one = sizeof(double);
    if (hipMemcpyAsync(buf, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        hipFree(buf);
        return fail(PLM_EDEVICE, "upload failed");
    }
    PlmRccl *r = nullptr;
    if (plm_rccl_init(rccl_id, nranks, rank, &r) != 0) {
        hipFree(buf);
        return fail(PLM_ECALLBACK, "RCCL communicator of %d ranks: %s", nranks, plm_rccl_error());
    }
    int rc = PLM_OK;
    if (plm_rccl_collective(r, PLM_COLL_ALLTOALL, buf, buf + (size_t)nranks * per, counts.data(), counts.data(), st) != 0)
        rc = fail(PLM_ECALLBACK, "all-to-all: %s", plm_rccl_error());
    if (rc == PLM_OK && plm_rccl_collective(r, PLM_COLL_ALLREDUCE_F64, buf + (size_t)2 * nranks * per, nullptr, &one, &one, st) != 0)
        rc = fail(PLM_ECALLBACK, "all-reduce: %s", plm_rccl_error());
    if (rc == PLM_OK && (hipMemcpyAsync(h.data(), buf, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
                         hipStreamSynchronize(st) != hipSuccess))
        rc = fail(PLM_EDEVICE, "download failed");
    if (rc == PLM_OK) {
        if (h[(size_t)2 * nranks * per] != 0.5 * nranks * (nranks + 1)) rc = fail(PLM_ECALLBACK, "all-reduce returned a wrong sum");
        for (int k = 0; k < nranks && rc == PLM_OK; k++)
            if (h[(size_t)(nranks + k) * per] != 1000.0 * k + rank) rc = fail(PLM_ECALLBACK, "all-to-all delivered a wrong message from rank %d", k);
    }
    hipStreamSynchronize(st);
    hipFree(buf);
    plm_rccl_destroy(r);
    return rc;
}

// The rank-local half of plm_rccl_probe -- device, RCCL library, buffer, upload -- with no communicator call in it: hosts
// run it on every rank and agree on the verdicts BEFORE any rank enters plm_rccl_probe.
int plm_rccl_probe_local(int32_t nranks, int device, void *stream) {
    if (nranks < 1) return fail(PLM_EINVAL, "bad communicator shape");
    PLM_TRY(check_device(device));
    if (plm_rccl_version() < 20000) return fail(PLM_ECALLBACK, "librccl not loadable: %s", plm_rccl_error());
    hipStream_t st = (hipStream_t)stream;
    double *buf = nullptr;
    PLM_TRY(dalloc(&buf, (size_t)(2 * nranks + 1) * 128));
    std::vector<double> h((size_t)(2 * nranks + 1) * 128, 1.0);
    const bool ok = hipMemcpyAsync(buf, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, st) == hipSuccess &&
                    hipStreamSynchronize(st) == hipSuccess;
    hipFree(buf);
    return ok ? PLM_OK : fail(PLM_EDEVICE, "upload failed");
}

int plm_ctx_set_options(plm_ctx_t *c, int32_t max_iter, double epsilon, int32_t lbfgs_m) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    if (max_iter >= 0) c->prob.max_iter = max_iter;
    if (lbfgs_m >= 0) c->prob.lbfgs_m = lbfgs_m;
    if (epsilon >= 0) {
        c->prob.epsilon = epsilon;
        // The stop rule also selects the precision of the plain evaluation's residuals (24 bits above 1e-4, 32 bits
        // below: the quantisation noise of three digit planes, ~5e-5 |x|, would make a tighter rule unreachable); the
        // buffers hold four planes in any case.
        PlmDims d2;
        PLM_TRY(make_dims(c->prob, c->opt, &d2));
        if (d2.nplanes != c->planes_base) {
            c->planes_base = d2.nplanes;
            c->ksplit_base = d2.ksplit;
            ctx_set_accurate(c, c->fwd_accurate);
            c->eval_valid = false;
        }
    }
    return PLM_OK;
}

int64_t plm_ctx_native_size(const plm_ctx_t *c) { return c ? c->d.n_local : 0; }

int plm_ctx_set_weights(plm_ctx_t *c, const float *weights_host) {
    if (!c || !weights_host) return fail(PLM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const PlmDims &d = c->d;
    double neff = 0;
    float wmax = 0;
    for (int s = 0; s < d.N; s++) {
        if (!(weights_host[s] >= 0.f) || !std::isfinite(weights_host[s]))
            return fail(PLM_EINVAL, "weight %d is negative or not finite", s);
        neff += weights_host[s];
        wmax = std::max(wmax, weights_host[s]);
    }
    // (rounds 1-2 limited the weights to < 4: the f16 planes of the residuals had a fixed 2^14 pre-scale.  The fixed-point
    // residuals of round 3 are scaled by the largest weight in use, any positive magnitude works.)
    if (!(neff > 0)) return fail(PLM_EINVAL, "sum of weights is zero");
    HIP_TRY(hipMemsetAsync(c->w, 0, sizeof(float) * d.Np, c->st));
    HIP_TRY(hipMemcpyAsync(c->w, weights_host, sizeof(float) * d.N, hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    c->n_eff = neff;
    c->have_weights = true;
    c->hcnt_valid = false;
    c->wmax = wmax;
    c->eval_valid = false;
    // residual quantisation of the backward GEMM: |r_s(i,a)| <= w_s, so the largest weight maps to the largest
    // magnitude whose signed digits fit int8
    ctx_set_accurate(c, c->fwd_accurate);
    return PLM_OK;
}

int plm_ctx_reweight(plm_ctx_t *c) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    HIP_TRY(hipSetDevice(c->device));
    const PlmDims &d = c->d;
    const int thresh = cluster_threshold(c->prob.theta_id, d.L, d.conv);
    std::vector<int32_t> counts(d.N);
    if (thresh <= 0) {
        std::fill(counts.begin(), counts.end(), d.N);  // every pair is a neighbour
        HIP_TRY(hipMemcpyAsync(c->counts, counts.data(), sizeof(int32_t) * d.N, hipMemcpyHostToDevice, c->st));
    } else {
        HIP_TRY(plm_launch_reweight(d, c->msa_rm, thresh, c->counts, c->st));
        HIP_TRY(hipMemcpyAsync(counts.data(), c->counts, sizeof(int32_t) * d.N, hipMemcpyDeviceToHost, c->st));
    }
    HIP_TRY(hipStreamSynchronize(c->st));
    std::vector<float> w(d.N);
    const float scale = c->prob.scale > 0 ? (float)c->prob.scale : 1.f;
    for (int s = 0; s < d.N; s++) {
        if (counts[s] < 1) return fail(PLM_ENUMERIC, "sequence %d has cluster size %d", s, counts[s]);
        w[s] = scale / (float)counts[s];
    }
    return plm_ctx_set_weights(c, w.data());
}

int plm_ctx_get_weights(plm_ctx_t *c, float *weights_host, int32_t *counts_host, float *n_eff) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (weights_host) HIP_TRY(hipMemcpy(weights_host, c->w, sizeof(float) * c->d.N, hipMemcpyDeviceToHost));
    if (counts_host) HIP_TRY(hipMemcpy(counts_host, c->counts, sizeof(int32_t) * c->d.N, hipMemcpyDeviceToHost));
    if (n_eff) *n_eff = (float)c->n_eff;
    return PLM_OK;
}

// compact: device scratch for PLM_FLAG_COMPACT_GAPS (gap mode: fij_host receives (Q-1)-state blocks); nullptr = q-state layout
static int ctx_marginals(plm_ctx_t *c, float *fi_host, float *fij_host, float *compact);
int plm_ctx_marginals(plm_ctx_t *c, float *fi_host, float *fij_host) { return ctx_marginals(c, fi_host, fij_host, nullptr); }
static int ctx_marginals(plm_ctx_t *c, float *fi_host, float *fij_host, float *compact) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    if (!c->have_weights) return fail(PLM_EINVAL, "weights not set: call plm_ctx_reweight / plm_ctx_set_weights");
    if (c->d.nshards > 1) return fail(PLM_EUNSUPPORTED, "marginals run unsharded (create a 1-shard context)");
    c->eval_valid = false;   // g is used as scratch below
    HIP_TRY(hipSetDevice(c->device));
    const PlmDims &d = c->d;
    // weighted one-hot Gram matrix through the backward GEMM: G = X^T diag(w) X
    HIP_TRY(plm_launch_onehot_rt(d, c->msa_rm, c->w, c->Rt, c->st));
    HIP_TRY(plm_launch_backward(d, c->msa_cm, c->Rt, c->G, nullptr, c->st));
    // gap mode: raw weighted counts come back (factor 1) and are normalised per site / per pair over
    // the ungapped sequences on the host
    HIP_TRY(plm_launch_assemble(d, c->G, d.ksplit, nullptr, c->g, c->g, 0.f, 0.f, c->reg_part, 1,
                                d.gap_mode ? 1.f : (float)(1.0 / c->n_eff), nullptr, 0.f, c->st));
    HIP_TRY(plm_launch_native_to_canon(d, c->g, c->canon, c->st));
    c->h_fi.resize((size_t)d.L * d.Qc);
    HIP_TRY(hipMemcpyAsync(c->h_fi.data(), c->canon, sizeof(float) * d.L * d.Qc, hipMemcpyDeviceToHost, c->st));
    // PLM_CONV_G_FREQ_TOTAL: normalised by N_eff (all sequences) instead of the ungapped ones
    const bool by_total = d.conv & PLM_CONV_G_FREQ_TOTAL;
    if (fij_host) {
        // gap mode: the pair blocks are normalised over the jointly ungapped sequences on the device (round 5; the host
        // loop over 44 850 blocks was 15 ms of a -g fit), same arithmetic: total in f64 in block order, (float)(f / total)
        if (d.gap_mode)
            HIP_TRY(plm_launch_gap_normalise_pairs(d, c->canon + (size_t)d.L * d.Qc, by_total ? (double)c->n_eff : 0.0, c->st));
        if (compact && d.gap_mode) {
            const size_t npair = (size_t)d.L * (d.L - 1) / 2, qn = (size_t)d.Qc - 1;
            HIP_TRY(plm_launch_compact_gap_blocks(d, c->canon + (size_t)d.L * d.Qc, compact, c->st));
            HIP_TRY(hipMemcpyAsync(fij_host, compact, sizeof(float) * npair * qn * qn, hipMemcpyDeviceToHost, c->st));
        } else {
            HIP_TRY(hipMemcpyAsync(fij_host, c->canon + (size_t)d.L * d.Qc,
                                   sizeof(float) * (d.n_canon - (int64_t)d.L * d.Qc), hipMemcpyDeviceToHost, c->st));
        }
    }
    HIP_TRY(hipStreamSynchronize(c->st));
    if (d.gap_mode) {
        const int Q = d.Qc;
        for (int i = 0; i < d.L; i++) {
            float *f = &c->h_fi[(size_t)i * Q];
            double tot = 0;
            for (int a = 1; a < Q; a++) tot += f[a];
            if (by_total) tot = c->n_eff;
            f[0] = 0.f;
            for (int a = 1; a < Q; a++) f[a] = tot > 0 ? (float)(f[a] / tot) : 0.f;
        }
    }
    if (fi_host) memcpy(fi_host, c->h_fi.data(), sizeof(float) * d.L * d.Qc);
    return PLM_OK;
}

int plm_ctx_set_x(plm_ctx_t *c, const float *x_canonical_host) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (!x_canonical_host) return set_start_point(c);
    c->eval_valid = false;
    const PlmDims &d = c->d;
    HIP_TRY(hipMemcpyAsync(c->canon, x_canonical_host, sizeof(float) * d.n_canon, hipMemcpyHostToDevice, c->st));
    HIP_TRY(plm_launch_canon_to_native(d, c->canon, c->x, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    return PLM_OK;
}

// c->canon <- the FULL canonical vector of a local (native-layout) vector; in sharded-state mode every
// shard contributes its own entries and an all-reduce (sum) puts the whole vector on every rank
static int canon_full(plm_ctx_t *c, const float *native) {
    const PlmDims &d = c->d;
    if (d.sharded) HIP_TRY(hipMemsetAsync(c->canon, 0, sizeof(float) * d.n_canon, c->st));
    HIP_TRY(plm_launch_native_to_canon(d, native, c->canon, c->st));
    if (d.sharded) {
        // all-gather of the parameter slices.  Every shard has written the entries IT owns into a zeroed canonical vector;
        // the sum over the shards is the whole vector on every rank (exact: every entry has one non-zero term).  (Round
        // 5 broadcast two contiguous ranges per shard; with the rectangles split between the shards the own entries are
        // no longer contiguous in the canonical order.)
        const int64_t bytes = (int64_t)sizeof(float) * d.n_canon;
        PLM_TRY(ctx_collective(c, PLM_COLL_ALLREDUCE_F32, c->canon, c->canon, &bytes, &bytes));
    }
    return PLM_OK;
}
static int get_vec(plm_ctx_t *c, const float *native, float *out_host) {
    if (!c || !out_host) return fail(PLM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    PLM_TRY(canon_full(c, native));
    HIP_TRY(hipMemcpyAsync(out_host, c->canon, sizeof(float) * c->d.n_canon, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    return PLM_OK;
}
// the same vector as two host arrays (fields [L][q], couplings [pairs][q][q]): what plm_fit hands back, without a
// staging copy of the whole vector (79 MB at the headline: allocation, zero fill and a second memcpy were 25 ms)
// compact (gap mode, PLM_FLAG_COMPACT_GAPS): device scratch; both arrays then come back in the (Q-1)-state layout
static int get_vec_split(plm_ctx_t *c, const float *native, float *h_host, float *j_host, float *compact) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    HIP_TRY(hipSetDevice(c->device));
    PLM_TRY(canon_full(c, native));
    const PlmDims &d = c->d;
    const size_t nh = (size_t)d.L * d.Qc;
    const bool cut = compact && d.gap_mode;
    std::vector<float> hfull(cut && h_host ? nh : 0);
    if (h_host) HIP_TRY(hipMemcpyAsync(cut ? hfull.data() : h_host, c->canon, sizeof(float) * nh, hipMemcpyDeviceToHost, c->st));
    if (j_host && cut) {
        const size_t npair = (size_t)d.L * (d.L - 1) / 2, qn = (size_t)d.Qc - 1;
        HIP_TRY(plm_launch_compact_gap_blocks(d, c->canon + nh, compact, c->st));
        HIP_TRY(hipMemcpyAsync(j_host, compact, sizeof(float) * npair * qn * qn, hipMemcpyDeviceToHost, c->st));
    } else if (j_host) {
        HIP_TRY(hipMemcpyAsync(j_host, c->canon + nh, sizeof(float) * ((size_t)d.n_canon - nh), hipMemcpyDeviceToHost, c->st));
    }
    HIP_TRY(hipStreamSynchronize(c->st));
    if (cut && h_host)
        for (int i = 0; i < d.L; i++) memcpy(h_host + (size_t)i * (d.Qc - 1), &hfull[(size_t)i * d.Qc + 1], sizeof(float) * (d.Qc - 1));
    return PLM_OK;
}
int plm_ctx_get_x(plm_ctx_t *c, float *x_canonical_host) { return get_vec(c, c ? c->x : nullptr, x_canonical_host); }
int plm_ctx_get_g(plm_ctx_t *c, float *g_canonical_host) { return get_vec(c, c ? c->g : nullptr, g_canonical_host); }

int plm_ctx_eval(plm_ctx_t *c, double *fx_out, double *nll_out) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    if (!c->have_weights) return fail(PLM_EINVAL, "weights not set");
    HIP_TRY(hipSetDevice(c->device));
    ctx_set_accurate(c, c->opt.fwd_mode != 0);    // a single evaluation is asked for its value: the accurate arithmetic
    PLM_TRY(ctx_eval_enqueue(c));
    if (c->d.sharded) PLM_TRY(ctx_allreduce_scalars(c, 0, 2));   // every shard must call eval together
    c->eval_valid = false;
    if (fx_out || nll_out) {
        PLM_TRY(fetch_scalars(c, 0, 2));
        if (fx_out) *fx_out = c->h_scal[0];
        if (nll_out) *nll_out = c->h_scal[1];
        c->eval_valid = true;
        c->eval_vp = false;
        c->eval_accurate = c->fwd_accurate;
        c->last_fx = c->h_scal[0];
        c->last_nll = c->h_scal[1];
    }
    return PLM_OK;
}

// L-BFGS with a More'-Thuente line search; vectors stay in HBM, only scalars cross PCIe.
// Defaults follow libLBFGS (m = 6, ftol 1e-4, gtol 0.9, <= 20 trial steps), which plmc bundles
// [recollection, SURVEY.md App. C.4].  The two-loop recursion (Nocedal 1980) is evaluated in
// coefficient space ("vector-free" L-BFGS, Chen et al. 2014): every inner product it needs is an
// entry of the Gram matrix of {s_j, y_j, g}, refreshed by ONE pass over the history per
// iteration (k_multidot), and the direction is ONE fused linear combination (k_multiaxpy).
// Two host synchronisations per iteration: after the line-search evaluation and after the pass.
int plm_ctx_optimize(plm_ctx_t *c, plm_iter_cb cb, void *user, plm_result_t *res) {
    if (!c) return fail(PLM_EINVAL, "NULL ctx");
    if (!c->have_weights) return fail(PLM_EINVAL, "weights not set");
    HIP_TRY(hipSetDevice(c->device));
    const double eps = c->prob.epsilon > 0 ? c->prob.epsilon : 1e-3;
    const double t0 = now_s();
    const PlmDims &d = c->d;
    const int64_t n = d.n_local;
    const int m = std::min(20, c->prob.lbfgs_m > 0 ? c->prob.lbfgs_m : 6);
    const int max_iter = c->prob.max_iter;
    const int max_ls = 20;
    const double ftol = 1e-4, gtol = 0.9, xtol = 1e-7, stpmin = 1e-20, stpmax = 1e20;
    // f is an f64 sum of ~N L f32 terms: its rounding noise is ~1e-10 |f| (measured: 2e-3 at f = 2.8e7, N = 100 000),
    // while the gradient -- exact integer sums of 24-bit residuals since round 3 -- is good to a few 1e-4 of |x|.  Near
    // the optimum of a large problem the decrease of an iteration (stp |g.d| / 2) falls below that noise before the stop
    // rule is met; a More'-Thuente search fed with such values interpolates on noise (seen at config 3: trials that
    // wander between two step lengths until the bracket collapses, accepted steps that raise f).  Two devices:
    //  * approximate Wolfe (Hager & Zhang 2005), as in the f32 oracle build: a trial whose f did not rise beyond
    //    epsf |f0| passes the sufficient-decrease test;
    //  * where |f(trial) - f(0)| is inside flat_rel |f(0)| AND the evaluated value contradicts what the derivatives
    //    imply, f(0) + stp (g0.d + g.d) / 2 (trapezoid: exact for a quadratic), by more than half of that implied change,
    //    the search works with the implied value -- then the sufficient-decrease test is H&Z's derivative form
    //    g.d <= (2 ftol - 1) g0.d and the cubic steps see consistent data.  Small problems never get there (their f is
    //    good to ~1e-10 |f| too, and that is far below their decreases).
    const double epsf = 1e-6, flat_rel = 4e-9;
    PLM_TRY(ctx_alloc_lbfgs(c, m));
    float *S = c->hist, *Y = c->hist + (size_t)m * n;
    // H0 = gamma * D^-1 with D the Hessian diagonal of the independent-site model at the start point (closed form
    // from the single-site frequencies).  The recursion only ever needs s.y, s.g and the D^-1-weighted products of
    // {y_j, g}.  Opt-in (PLM_FLAG_PRECOND): measured to need MORE iterations than the scalar H0 at L = 300.
    const bool precond = (c->prob.flags & PLM_FLAG_PRECOND) && !c->h_fi.empty();
    if (precond) {
        if (!c->dinv) PLM_TRY(dalloc(&c->dinv, (size_t)n));
        const size_t lq = (size_t)d.L * d.Q;
        std::vector<float> fv(2 * lq, 0.f);
        const int a0 = d.gap_mode;
        for (int i = 0; i < d.L; i++) {
            double tot = 0;
            for (int a = a0; a < d.Qc; a++) tot += (double)c->h_fi[i * d.Qc + a] + 1.0 / c->n_eff;
            for (int a = a0; a < d.Qc; a++) {
                const double pa = ((double)c->h_fi[i * d.Qc + a] + 1.0 / c->n_eff) / tot;   // the start point's P_i(a)
                fv[i * d.Q + a] = c->h_fi[i * d.Qc + a];
                fv[lq + i * d.Q + a] = (float)(pa * (1.0 - pa));
            }
        }
        // staged through the gradient buffer's tail?  no: a small dedicated upload into canon (free during optimize)
        HIP_TRY(hipMemcpyAsync(c->canon, fv.data(), sizeof(float) * fv.size(), hipMemcpyHostToDevice, c->st));
        HIP_TRY(plm_launch_precond(d, c->canon, (float)c->n_eff, (float)c->prob.lambda_h, (float)c->prob.lambda_j,
                                   c->dinv, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));   // fv leaves scope
    }
    const float *dinv = precond ? c->dinv : nullptr;
    // Gram matrix pieces, indexed by history slot: SY[i][j] = s_i.y_j, YDY[i][j] = y_i.D^-1 y_j, Sg = s_i.g,
    // YDg = y_i.D^-1 g, gDg = g.D^-1 g, gg = g.g (stop rule)
    std::vector<double> SY(m * m, 0.0), YDY(m * m, 0.0), Sg(m, 0.0), YDg(m, 0.0);
    double gg = 0, gDg = 0, xx = 0, hh = 0;
    std::vector<double> cs(m), cy(m);
    c->n_evals = 0;
    // device scalar slots; in sharded-state mode each fetch is preceded by a sum over the shards of
    // exactly the slots that were just written ([FX..DG] after an evaluation, [XX..MD+..] after the pass)
    enum { SL_FX = 0, SL_NLL = 1, SL_DG = 2, SL_XX = 3, SL_HH = 4, SL_GH2 = 5, SL_MD = 8 };

    auto norm_dots = [&]() -> int {   // x.x (all) and x.x (fields only)
        const float *a[1] = {c->x};
        PLM_TRY(dots(c, 1, a, a, n, SL_XX));
        PLM_TRY(dots(c, 1, a, a, d.nh_pad_l, SL_HH));
        return PLM_OK;
    };
    // trial != nullptr: the launch also writes the first trial point of the line search, trial = xacc + stp0 * p
    auto direction = [&](int stored, int end, double *dginit, const float *xacc, float stp0, float *trial) -> int {
        // p = sum cs[j] s_j + D^-1 (sum cy[j] y_j + cg g): two-loop recursion in coefficient space over the LIVE pairs,
        // the `stored` ring slots before `end` (after a skipped pair on a full ring the dead slot is `end` itself, not
        // the highest physical slot)
        double cg;
        plm_lbfgs_coefficients(m, stored, end, SY.data(), YDY.data(), Sg.data(), YDg.data(), gDg, cs.data(), cy.data(), &cg,
                               dginit);
        PlmVecList B;
        PlmCoefList C;
        B.n = 0;
        for (int i = 0; i < stored; i++) { const int j = (end + m - 1 - i) % m; B.v[B.n] = S + (size_t)j * n; C.c[B.n++] = (float)cs[j]; }
        const int first_weighted = B.n;
        for (int i = 0; i < stored; i++) { const int j = (end + m - 1 - i) % m; B.v[B.n] = Y + (size_t)j * n; C.c[B.n++] = (float)cy[j]; }
        B.v[B.n] = c->g;
        C.c[B.n++] = (float)cg;
        if (trial) HIP_TRY(plm_launch_multiaxpy_trial(c->dir, B, C, n, dinv, first_weighted, xacc, stp0, trial, c->st));
        else HIP_TRY(plm_launch_multiaxpy(c->dir, B, C, n, dinv, first_weighted, c->st));
        return PLM_OK;
    };

    // variable projection: the fields are solved exactly for every trial couplings (the chain of ctx_eval_vp_enqueue);
    // L-BFGS then only sees the couplings (field part of g is zero, field part of s = the change of the optimal fields)
    const bool vp = vp_enabled(c);
    if (vp) PLM_TRY(vp_alloc(c));
    c->stat_field_ms = c->stat_passes = c->stat_fwd_ms = c->stat_bwd_ms = 0;
    c->stat_field_evals = c->stat_chain_short = c->stat_gemm_evals = 0;
    double gh2 = 0;                          // |grad_h|^2 left by the field solver at the current point
    // field-solver tolerance: a fraction of what the stop rule allows the whole gradient, and never below what the
    // solver can reach.  The fields themselves are iterated in f64 (k_hsolve); what is left is the random f32
    // rounding of the stored potentials and of the softmax, ~1e-7 per (sequence, site) term, i.e.
    // ~sqrt(N_eff) per entry (measured with scripts/vp_floor_probe.py: 3e-8 ... 3e-7 sqrt(N_eff L q) in norm).  Below
    // it an evaluation only burns passes until the stall test of ctx_eval_vp_finish ends it.
    // (A tolerance relative to the current gradient of the couplings -- inexact field solves far from the optimum --
    // was measured: 0.3 % of |g| costs 15 % more iterations at the headline and stalls config 2 at |g|/|x| = 0.1.)
    const double vp_floor = c->opt.vp_floor;   // 2e-7 unless a probe set PLM_VP_FLOOR
    // Far from the optimum that absolute tolerance is ~1e-7 of the gradient the fields are solved FOR: the solve may stop
    // at vp_rel (1e-4) of the norm of the reduced gradient at the last accepted point -- the error it leaves in a trial's
    // gradient is of that relative size, far below what a quasi-Newton step notices, and it only binds while
    // |g|/|x| > 0.1 eps / vp_rel (= 1 at the default stop rule).  What it saves is the tail of every early chain: a
    // handful of sites whose sampled Hessians are poor converge at 0.6 per pass and kept the chain going for 5-9 passes
    // between |g_h| = 0.5 and the absolute tolerance (PLM_DEBUG_VP traces of round 5).  Rounds 2-4 ended those
    // evaluations by a stall rule instead, at |g_h| up to 1e5; a tolerance of 3e-3 |g| was measured then (15 % more
    // iterations, config 2 stalling at |g|/|x| = 0.1) -- 30 times looser than this one.
    // The floor itself is not part of the tolerance any more (round 5): the chain ends ON the pass that meets the
    // tolerance, so a tolerance of the floor's size would leave |g_h| at that size -- 2e-6 |x| on a small problem, the
    // whole of a tight stop rule; the chain goes below the floor while a pass still gains a factor 2 (k_vp_check).
    const double vp_rel = c->opt.vp_rel;
    c->vp_floor2 = vp_floor * vp_floor * c->n_eff * (double)d.L * d.Q;
    auto vp_tol2 = [&](double xnorm2) {
        const double t = std::max(0.1 * eps * std::max(1.0, std::sqrt(xnorm2)), vp_rel * std::sqrt(gg));
        return t * t;
    };
    // Forward-GEMM mode.  The plain instantiation's f32 accumulation leaves an error of ~3e-11 N L |x| in the gradient
    // (4.5e-4 |x| at the headline, 1e-3 |x| at N = 100 000 -- the size of the default stop rule).
    // Far from the optimum that is irrelevant; the last iterations run the accurate instantiation (f64 outer sums), so the
    // stop rule is decided on a gradient whose error is several times smaller.  The switch happens at an accepted point,
    // which is evaluated once more so that f, g, the pair of the step and the Gram rows all come from one arithmetic.
    const double fwd_noise = 3e-11 * (double)d.N * (double)d.L;
    const double acc_thr = std::max(3.0 * eps, c->opt.acc_factor * fwd_noise);
    // (a problem whose plain-kernel error is below a twentieth of the stop rule never needs the switch)
    auto want_accurate = [&](double cond) {
        return c->opt.fwd_mode == 1 || (c->opt.fwd_mode != 0 && fwd_noise > 0.05 * eps && cond < acc_thr);
    };
    // objective and gradient at the start point -- unless this context still holds them (a resumed fit)
    const bool resume = c->eval_valid && c->eval_vp == vp;
    ctx_set_accurate(c, c->opt.fwd_mode == 1 || (c->opt.fwd_mode != 0 && resume && c->eval_accurate));
    auto start_eval = [&](bool have) -> int {
        const double tol2 = vp_tol2((double)d.L);
        if (!have) {
            if (!vp) PLM_TRY(ctx_eval_enqueue(c));
            else PLM_TRY(ctx_eval_vp_enqueue(c, tol2));
        }
        for (;;) {
            PlmVecList Qg, Bg;
            Qg.n = 1; Qg.v[0] = c->g;
            Bg.n = 2; Bg.v[0] = c->g; Bg.v[1] = c->g;
            HIP_TRY(plm_launch_multidot(Qg, Bg, n, c->dot_scratch, c->scal + SL_MD, dinv, 1u, 1ull, c->st));   // gDg, gg
            PLM_TRY(norm_dots());
            PLM_TRY(ctx_allreduce_scalars(c, have ? SL_DG : 0, (have ? 8 - SL_DG : 8) + 2));
            PLM_TRY(fetch_scalars(c, 0, 10));
            bool again = false;
            if (vp && !have) PLM_TRY(ctx_eval_vp_finish(c, tol2, &again, &gh2));
            if (!again) break;
        }
        gDg = c->h_scal[SL_MD];
        gg = c->h_scal[SL_MD + 1];
        xx = c->h_scal[SL_XX];
        hh = c->h_scal[SL_HH];
        return PLM_OK;
    };
    PLM_TRY(start_eval(resume));
    if (resume) gh2 = c->last_gh2;
    double fx = resume ? c->last_fx : c->h_scal[SL_FX], nll = resume ? c->last_nll : c->h_scal[SL_NLL];
    c->eval_valid = false;
    if (!std::isfinite(fx)) return fail(PLM_ENUMERIC, "objective is not finite at the start point");
    if (!c->fwd_accurate && want_accurate(std::sqrt(gg + gh2) / std::max(1.0, std::sqrt(xx)))) {
        ctx_set_accurate(c, true);   // a start point this close to the optimum: its gradient decides the stop rule
        PLM_TRY(start_eval(false));
        fx = c->h_scal[SL_FX];
        nll = c->h_scal[SL_NLL];
    }
    // Certificate of the point that SHIPS.  The field solver iterates the fields in f64; the parameter vector (and the
    // .model file) holds them rounded to f32.  A field off by 2^-25 |h| changes the gradient by (N-proportional
    // curvature) x that: at N = 100 000 the float64 oracle sees 4.7e-4 |x| more at the rounded point than the solver at
    // its own (round 4: reported 0.94e-3, oracle 1.05e-3).  So when the reduced gradient meets the stop rule, the joint
    // gradient -- fields included -- is evaluated once at the rounded point (the stored potentials are reused: residual
    // pass, backward GEMM, assemble into the free direction vector); the fit ends only if THAT meets the rule, otherwise
    // it goes on with a target lowered by what the rounding costs.
    auto shipped_cond = [&](double *out) -> int {
        HIP_TRY(plm_launch_h64_init(d, c->x, c->h64, c->st));
        HIP_TRY(plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 1, 0, c->fwd_accurate, c->Rt, c->fx_part, nullptr, nullptr,
                                 nullptr, nullptr, PLM_VP_ALWAYS, c->st));
        PLM_TRY(vp_stage3(c, false, c->dir, 0));
        const float *a[1] = {c->dir};
        PLM_TRY(dots(c, 1, a, a, n, SL_DG));
        PLM_TRY(ctx_allreduce_scalars(c, SL_DG, 1));
        PLM_TRY(fetch_scalars(c, SL_DG, 1));
        *out = std::sqrt(c->h_scal[SL_DG]) / std::max(1.0, std::sqrt(xx));
        return PLM_OK;
    };
    double eps_eff = eps;      // target of the reduced gradient: the stop rule minus what the f32 rounding of the fields costs
    double rounding_note = 0;  // > 0: the rule was met on the f64-field point only; what rounding the fields to f32 adds
    int n_cert = 0;
    int k = 0, end = 0, stored = 0, status = PLM_STATUS_CONVERGED, ls_reason = 0, restarts = 0;
    // The next pair is (x - anchor, g - g(anchor)).  The anchor is the previous accepted point (xp, gp) -- unless the
    // pair(s) since were skipped as noise (below): then it stays where the last STORED pair ended, in its own buffers,
    // so that the difference is taken over a longer baseline, where H s outgrows the evaluation error again.
    bool anchored = false;
    double last_cond = std::sqrt(gg + gh2) / std::max(1.0, std::sqrt(xx));   // |g|/max(1,|x|) at the last accepted point
    // (a resumed fit as well -- its previous run certified the point for ITS epsilon -- when the potentials of the
    // point are still there)
    if (last_cond <= eps && vp && (!resume || c->hj_at_x)) {   // a start point that meets the rule must meet it as it ships, too
        double cj = 0;
        PLM_TRY(shipped_cond(&cj));
        const double r2 = std::max(0.0, cj * cj - last_cond * last_cond);
        if (cj > eps && r2 >= 0.75 * eps * eps) rounding_note = std::sqrt(r2);       // not representable in float32 fields
        else if (cj > eps) { eps_eff = std::sqrt(eps * eps - r2); last_cond = cj; }   // go on until the shipped point meets it
    }
    if (last_cond > eps) {
        // first step: unit displacement along the plain gradient; the D^-1-scaled direction is Newton-like for the
        // diagonal part of the Hessian, so it starts from min(1, that)
        auto first_step = [&]() { return precond ? std::min(1.0, 1.0 / std::sqrt(gDg)) : 1.0 / std::sqrt(gg); };
        double step = first_step();
        for (k = 1;; k++) {
            double dginit;
            // the direction and, in the same pass, the first trial point x + stp0 p -- written where the trial points are
            // built after the swap below (today's xp: the point before the accepted one, not needed any more unless it is
            // still the anchor of the next pair, which lives in its own buffers then)
            const float stp0 = (float)std::max(stpmin, std::min(stpmax, step));
            PLM_TRY(direction(stored, end, &dginit, c->x, stp0, c->xp));
            if (!(dginit < 0)) { status = PLM_STATUS_LINESEARCH; ls_reason = 10; k--; break; }
            // the accepted point moves to (xp, gp); trial points are built in (x, g)
            std::swap(c->x, c->xp);
            std::swap(c->g, c->gp);
            const double finit = fx, nllinit = nll, dgtest = ftol * dginit;
            // the new pair goes into slot `end`; queries and basis of the Gram pass (rows for s, y, g)
            float *s_new = S + (size_t)end * n, *y_new = Y + (size_t)end * n;
            const int nst = std::min(m, stored + 1);
            PlmVecList Qv, B;
            Qv.n = 3;
            Qv.v[0] = s_new; Qv.v[1] = y_new; Qv.v[2] = c->g;       // c->g: the buffer the trial gradients land in
            B.n = 0;
            for (int i = 0; i < nst; i++) B.v[B.n++] = S + (size_t)i * n;
            unsigned long long wb = 0;                    // basis vectors that enter products in the H0 metric: Y, g
            for (int i = 0; i < nst; i++) { wb |= 1ull << B.n; B.v[B.n++] = Y + (size_t)i * n; }
            wb |= 1ull << B.n;
            B.v[B.n++] = c->g;
            B.v[B.n++] = c->g;                            // once more, unweighted: g.g for the stop rule
            const unsigned wq = 6u;                       // queries y_new and g
            double gh2_trial = 0;
            // objective + gradient at the trial point c->x, and with it everything the NEXT direction needs
            auto evaluate_trial = [&]() -> int {
                const double tol2 = vp_tol2(xx);
                if (!vp) PLM_TRY(ctx_eval_enqueue(c));
                else PLM_TRY(ctx_eval_vp_enqueue(c, tol2));
              for (;;) {
                // Speculate that this trial point is accepted (it is, 97 % of the time): form its (s, y) pair
                // in slot `end` -- the slot the next pair goes to anyway; a rejected trial is simply
                // overwritten by the next one -- and run the Gram pass now, so that ONE host
                // synchronisation (and, sharded, one all-reduce) per trial brings back f, the directional
                // derivative and everything the next direction needs.
                // (round 6: pair, Gram rows, g.p, x.x and the fields' x.x in ONE pass over the vectors)
                HIP_TRY(plm_launch_sy_multidot(s_new, y_new, c->x, anchored ? c->xa : c->xp, c->g, anchored ? c->ga : c->gp,
                                               c->dir, B, n, d.nh_pad_l, c->dot_scratch, c->scal + SL_MD, c->scal + SL_DG, dinv,
                                               wq, wb, c->st));
                PLM_TRY(ctx_allreduce_scalars(c, SL_FX, SL_MD + 3 * B.n + 1 - SL_FX));
                PLM_TRY(fetch_scalars(c, SL_FX, SL_MD + 3 * B.n + 1 - SL_FX));
                bool again = false;      // the field solver's chain ran out of positions: more was enqueued (rare)
                if (vp) PLM_TRY(ctx_eval_vp_finish(c, tol2, &again, &gh2_trial));
                if (!again) break;
              }
                return PLM_OK;
            };
            int brackt = 0, stage1 = 1, count = 0, uinfo = 0, lsrc = 1;
            double width = stpmax - stpmin, prev_width = 2.0 * width;
            double stx = 0, fxx = finit, dgx = dginit, sty = 0, fy = finit, dgy = dginit, stp = step, stmin, stmax;
            double trace[64][3];
            for (;;) {
                if (brackt) { stmin = std::min(stx, sty); stmax = std::max(stx, sty); }
                else { stmin = stx; stmax = stp + 4.0 * (stp - stx); }
                stp = std::max(stpmin, std::min(stpmax, stp));
                if ((brackt && (stp <= stmin || stmax <= stp || count >= max_ls - 1 || uinfo)) ||
                    (brackt && stmax - stmin <= xtol * stmax))
                    stp = stx;
                if (!(count == 0 && (float)stp == stp0))      // the first trial point came with the direction
                    HIP_TRY(plm_launch_lincomb(c->x, 1.f, c->xp, (float)stp, c->dir, n, c->st));
                PLM_TRY(evaluate_trial());
                double dg = c->h_scal[SL_DG];
                fx = c->h_scal[SL_FX];
                nll = c->h_scal[SL_NLL];
                if (!std::isfinite(fx) || !std::isfinite(dg)) {
                    // the trial left the region where the objective is representable: the interpolation formulas
                    // of mt_update are meaningless on it (NaN steps).  Make the trial the far end of the bracket
                    // with a finite stand-in value and bisect towards the best point found so far.
                    if (count < 64) { trace[count][0] = stp; trace[count][1] = INFINITY; trace[count][2] = 0; }
                    if (++count >= max_ls) { lsrc = -5; break; }
                    sty = stp;
                    fy = fxx + std::fabs(fxx) + 1.0;
                    dgy = std::fabs(dgx);
                    brackt = 1;
                    stp = stx + 0.5 * (stp - stx);
                    width = std::fabs(sty - stx);
                    prev_width = 2.0 * width;
                    continue;
                }
                const double ftest1 = finit + stp * dgtest;
                // the value the search works with: the evaluated one, or inside the noise band the derivative-implied one
                const double f_implied = finit + 0.5 * stp * (dginit + dg);
                const bool noisy = std::fabs(fx - finit) <= flat_rel * std::fabs(finit) &&
                                   std::fabs(fx - f_implied) > 0.5 * std::fabs(f_implied - finit) + 1e-12 * std::fabs(finit);
                const double fl = noisy ? f_implied : fx;
                if (count < 64) { trace[count][0] = stp; trace[count][1] = fx - finit; trace[count][2] = dg; }
                count++;
                if (brackt && (stp <= stmin || stmax <= stp || uinfo)) { lsrc = -1; break; }
                if (stp == stpmax && fl <= ftest1 && dg <= dgtest) { lsrc = -2; break; }
                if (stp == stpmin && (ftest1 < fl || dgtest <= dg)) { lsrc = -3; break; }
                if (brackt && stmax - stmin <= xtol * stmax) { lsrc = -4; break; }
                if (count >= max_ls) { lsrc = -5; break; }
                if ((fl <= ftest1 || (!noisy && fx <= finit + epsf * std::fabs(finit))) && std::fabs(dg) <= gtol * (-dginit)) {
                    lsrc = 1;
                    break;
                }
                if (stage1 && fl <= ftest1 && std::min(ftol, gtol) * dginit <= dg) stage1 = 0;
                if (stage1 && ftest1 < fl && fl <= fxx) {
                    double fm = fl - stp * dgtest, fxm = fxx - stx * dgtest, fym = fy - sty * dgtest;
                    double dgm = dg - dgtest, dgxm = dgx - dgtest, dgym = dgy - dgtest;
                    uinfo = mt_update(&stx, &fxm, &dgxm, &sty, &fym, &dgym, &stp, fm, dgm, stmin, stmax, &brackt);
                    fxx = fxm + stx * dgtest; fy = fym + sty * dgtest;
                    dgx = dgxm + dgtest; dgy = dgym + dgtest;
                } else {
                    uinfo = mt_update(&stx, &fxx, &dgx, &sty, &fy, &dgy, &stp, fl, dg, stmin, stmax, &brackt);
                }
                if (brackt) {
                    if (0.66 * prev_width <= std::fabs(sty - stx)) stp = stx + 0.5 * (sty - stx);
                    prev_width = width;
                    width = std::fabs(sty - stx);
                }
            }
            if (lsrc < 0 && c->opt.debug) {
                fprintf(stderr, "[plm] line search failed at iteration %d: code %d, finit=%.6f dginit=%.6e step0=%.3e brackt=%d stx=%.6e sty=%.6e\n", k, -lsrc, finit, dginit, step, brackt, stx, sty);
                for (int t = 0; t < count && t < 64; t++)
                    fprintf(stderr, "[plm]   trial %d: stp=%.9e  f-f0=%.6e  dg=%.6e\n", t, trace[t][0], trace[t][1], trace[t][2]);
            }
            if (lsrc < 0) {
                // the last accepted point is still in (xp, gp): make it current again
                std::swap(c->x, c->xp);
                std::swap(c->g, c->gp);
                fx = finit;
                nll = nllinit;
                k--;
                if (stored > 0 && restarts < 2) {
                    // a stale quasi-Newton model is the usual reason near the f32 floor: drop the history
                    // and retry once from steepest descent before giving up
                    restarts++;
                    stored = 0;
                    end = 0;
                    anchored = false;
                    step = first_step();
                    continue;
                }
                status = PLM_STATUS_LINESEARCH;
                ls_reason = -lsrc;
                break;
            }
            restarts = 0;
            if (c->opt.debug)
                fprintf(stderr, "[plm] it %d: %d trial(s), step %.4e (first %.4e), f - f0 = %.6e, dg0 = %.4e, dg = %.4e%s\n", k, count, stp,
                        trace[0][0], fx - finit, dginit, c->h_scal[SL_DG], c->fwd_accurate ? " [accurate]" : "");
            step = stp;
            // the pair of the accepted point already sits in slot `end` and its Gram rows in h_scal (see above)
            const double *md = c->h_scal + SL_MD;
            const int nbv = B.n, e = end;
            bool straddle = false;              // this step's pair would difference gradients of two arithmetics
            if (!c->fwd_accurate &&
                want_accurate(std::sqrt(md[2 * nbv + 2 * nst + 1] + gh2_trial) / std::max(1.0, std::sqrt(c->h_scal[SL_XX])))) {
                ctx_set_accurate(c, true);      // from here on: the accurate arithmetic, starting with this very point
                straddle = true;
                PLM_TRY(evaluate_trial());
                fx = c->h_scal[SL_FX];
                nll = c->h_scal[SL_NLL];
                if (!std::isfinite(fx)) return fail(PLM_ENUMERIC, "objective is not finite after the switch of the forward GEMM");
            }
            for (int j = 0; j < nst; j++) {
                SY[e * m + j] = md[0 * nbv + nst + j];            // s_e . y_j
                SY[j * m + e] = md[1 * nbv + j];                  // s_j . y_e
                YDY[e * m + j] = YDY[j * m + e] = md[1 * nbv + nst + j];
                Sg[j] = md[2 * nbv + j];
                YDg[j] = md[2 * nbv + nst + j];
            }
            gDg = md[2 * nbv + 2 * nst];
            gg = md[2 * nbv + 2 * nst + 1];
            xx = c->h_scal[SL_XX];
            hh = c->h_scal[SL_HH];
            gh2 = gh2_trial;
            const double xnorm = std::sqrt(xx), gnorm = std::sqrt(gg + gh2);
            last_cond = gnorm / std::max(1.0, xnorm);
            // a non-zero return cancels the fit (a pipeline's SIGTERM / SIGINT handler, evcouplings/utils/pipeline.py:476-545,
            // raised inside a Python callback): the point reached so far stays in the context
            if (cb && cb(k, now_s() - t0, last_cond, fx, nll, std::sqrt(hh), std::sqrt(std::max(0.0, xx - hh)), user) != 0) {
                status = PLM_STATUS_INTERRUPTED;
                break;
            }
            if (last_cond <= eps_eff) {
                if (!vp) { status = PLM_STATUS_CONVERGED; break; }
                double cj = 0;
                PLM_TRY(shipped_cond(&cj));
                if (!(cj > eps)) { last_cond = cj; status = PLM_STATUS_CONVERGED; break; }
                const double r2 = std::max(0.0, cj * cj - last_cond * last_cond);     // the rounding's share
                if (r2 >= 0.75 * eps * eps || ++n_cert > 8) {     // the target would fall below eps / 2
                    // float32 fields cannot carry this stop rule (a rule far below 1e-3, or a very deep alignment): the
                    // fit ends by the rule on its own (f64-field) point and says what the rounding adds
                    status = PLM_STATUS_CONVERGED;
                    rounding_note = std::sqrt(r2);
                    break;
                }
                eps_eff = std::min(0.9 * eps_eff, std::sqrt(eps * eps - r2));
            }
            if (max_iter > 0 && k >= max_iter) { status = PLM_STATUS_MAXITER; break; }
            // Curvature pair of the accepted step (slot e).  s.y > 0 always holds under the Wolfe conditions -- for exact
            // gradients.  Near the noise floor of the evaluation (config 3: error 1e-3 |x| at a stop rule of 1e-3) steps
            // get so short that y = H s + (difference of two evaluation errors) is mostly the latter: s.y is then a tiny
            // number of either sign, 1 / s.y blows the two-loop coefficients up (seen with PLM_DEBUG: directions whose
            // g.d disagreed in sign with the value the coefficients imply, a unit step raising f by 8e14) and the fit
            // ends in a line-search failure.  For a convex quadratic cos(s, H s) >= 2 sqrt(k) / (1 + k) with k the
            // condition number (0.12 at the headline's 120 : 3.5e4; the field part of s lowers it to ~0.04), a pair of
            // pure noise has cos ~ 1 / sqrt(P) = 2e-4: pairs below 1e-3 are not stored (the history keeps its other
            // pairs; the slot is written again by the next iteration -- with a pair over the longer baseline from the
            // anchor, see `anchored`).
            // Variable projection: y has no field part (the reduced gradient's is zero), the curvature information of the pair
            // lives in the couplings -- but s carries the CHANGE OF THE OPTIMAL FIELDS as well, which enters no product of the
            // two-loop recursion and only lowers cos(s, y) (0.12 -> 0.04 at the headline; at N = 500 000 below the admission
            // rule: every pair of the accurate phase was skipped and the history went stale, NOTES_r06.md).  The rule looks at
            // the coupling part of s.
            const double ss_full = md[0 * nbv + e], ss_h = vp ? std::min(ss_full, std::max(0.0, md[3 * nbv])) : 0.0;
            const double sy = SY[e * m + e], ss = std::max(ss_full - ss_h, 1e-300), yy = md[1 * nbv + nst + e];
            if (c->opt.debug)
                fprintf(stderr, "[plm]        pair: s.y = %.4e, |s| = %.4e (couplings %.4e), |y| = %.4e, cos = %.3e; history %d pair(s)%s%s\n",
                        sy, std::sqrt(ss_full), std::sqrt(ss), std::sqrt(yy), sy / std::sqrt(std::max(1e-300, ss * yy)), stored,
                        anchored ? ", anchored" : "", straddle ? ", straddle" : "");
            if (straddle) {
                // y = g_accurate(x) - g_plain(previous x) carries the DIFFERENCE of the two evaluations' errors (~3e-11 N L
                // |x|): not a curvature pair.  It is dropped (slot e, the oldest pair once the ring is full, goes with it);
                // the next pair starts from this point, whose gradient is the accurate one.
                stored = std::min(stored, m - 1);
                anchored = false;
            } else if (sy > 0 && sy * sy >= 1e-6 * ss * yy) {
                stored = nst;
                end = (end + 1) % m;
                anchored = false;
            } else if (sy > 0) {       // noise-dominated pair: skipped; slot e (the oldest pair once the ring is full) is gone
                stored = std::min(stored, m - 1);
                if (!anchored) {       // keep the point the pair started from: (xp, gp) is overwritten by the next swap
                    if (!c->xa) {
                        PLM_TRY(dalloc(&c->xa, (size_t)n));
                        PLM_TRY(dalloc(&c->ga, (size_t)n));
                    }
                    HIP_TRY(hipMemcpyAsync(c->xa, c->xp, sizeof(float) * n, hipMemcpyDeviceToDevice, c->st));
                    HIP_TRY(hipMemcpyAsync(c->ga, c->gp, sizeof(float) * n, hipMemcpyDeviceToDevice, c->st));
                    anchored = true;
                }
            } else {                   // slot e now holds a rejected pair: restart the history
                stored = 0;
                end = 0;
                anchored = false;
            }
            step = 1.0;
            // (Rounds 4-5 kept a stagnation watch here -- no new best |g|/|x| for 12 iterations next to the stop rule ->
            // history dropped.  It did not fire in any run since the accurate evaluation took over the last iterations of a
            // fit and had no test: removed in round 6.)
        }
    }
    HIP_TRY(hipStreamSynchronize(c->st));
    c->eval_valid = true;   // every exit path above leaves the last accepted point in (x, g)
    c->hj_at_x = vp && status != PLM_STATUS_LINESEARCH;   // a failed line search ends on a rejected trial's potentials
    c->eval_vp = vp;
    c->eval_accurate = c->fwd_accurate;
    c->last_fx = fx;
    c->last_nll = nll;
    c->last_gh2 = gh2;
    if (res) {
        res->iters_done = k;
        res->n_evals = c->n_evals;
        res->status = status;
        res->fx = fx;
        res->n_eff = (float)c->n_eff;
        res->seconds_optimize = now_s() - t0;
        // a fit that did not meet the stop rule says how far it got: the reference records this string as
        // `optimization_status` (couplings/tools.py:99), the only place a pipeline user sees it
        if (status == PLM_STATUS_CONVERGED && rounding_note > 0)
            snprintf(res->status_msg, sizeof res->status_msg,
                     "%s; float32 rounding of the fields adds %.1e", status_text(status), rounding_note);
        else if (status == PLM_STATUS_LINESEARCH)
            snprintf(res->status_msg, sizeof res->status_msg, "%s [code %d]; |g|/max(1,|x|) = %.3e", status_text(status),
                     ls_reason, last_cond);
        else if (status == PLM_STATUS_MAXITER || status == PLM_STATUS_INTERRUPTED)
            snprintf(res->status_msg, sizeof res->status_msg, "%s; |g|/max(1,|x|) = %.3e (epsilon %.1e)",
                     status_text(status), last_cond, eps);
        else
            snprintf(res->status_msg, sizeof res->status_msg, "%s", status_text(status));
    }
    return PLM_OK;
}

int plm_ctx_scores(plm_ctx_t *c, float *fn_host, float *cn_host) {
    if (!c || !fn_host || !cn_host) return fail(PLM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const PlmDims &d = c->d;
    float *fn_dev = c->canon + d.n_canon;
    PLM_TRY(canon_full(c, c->x));
    PlmDims dc = d;
    dc.Q = d.Qc;             // k_fn walks the canonical blocks: the gauge is taken over the problem's states only
    if (d.gap_mode) {
        // the zero-sum gauge and the norm are taken over the model's (Q-1) states only: k_fn leaves out row and column 0
        // of every block (round 5; rounds 2-4 repacked the blocks on the host: 54 ms of a -g fit at the headline)
        HIP_TRY(plm_launch_fn(dc, c->canon + (size_t)d.L * d.Qc, fn_dev, 1, 1, c->st));
    } else {
        HIP_TRY(plm_launch_fn(dc, c->canon + (size_t)d.L * d.Qc, fn_dev, (d.conv & PLM_CONV_FN_NO_GAP) ? 1 : 0, 0, c->st));
    }
    HIP_TRY(hipMemcpyAsync(fn_host, fn_dev, sizeof(float) * d.L * d.L, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    // APC (couplings/model.py:744-775): means over off-diagonal entries, diagonal blanked
    const int L = d.L;
    std::vector<double> col(L, 0.0);
    double tot = 0;
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++) {
            col[j] += fn_host[(size_t)i * L + j];
            tot += fn_host[(size_t)i * L + j];
        }
    const double mean = tot / ((double)L * (L - 1));
    for (int j = 0; j < L; j++) col[j] /= (double)(L - 1);
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++)
            cn_host[(size_t)i * L + j] =
                (i == j) ? 0.f : (float)((double)fn_host[(size_t)i * L + j] - col[i] * col[j] / mean);
    return PLM_OK;
}

int plm_ctx_time_field_positions(plm_ctx_t *c, int32_t reps, float *out_ms) {
    if (!c || !out_ms || reps <= 0) return fail(PLM_EINVAL, "bad argument");
    if (!c->have_weights) return fail(PLM_EINVAL, "weights not set");
    if (!vp_enabled(c) || !c->hj) return fail(PLM_EINVAL, "no stored potentials: plm_ctx_time_kernels on a variable-projection context first");
    HIP_TRY(hipSetDevice(c->device));
    const PlmDims &d = c->d;
    c->eval_valid = false;
    ctx_set_accurate(c, false);
    PLM_TRY(vp_counts(c));
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    double acc[2] = {0, 0};
    int rc = PLM_OK;
    for (int r = 0; r < reps && rc == PLM_OK; r++) {
        float ms;
        hipError_t e = hipEventRecord(ev[0], c->st);
        if (e == hipSuccess) e = plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 0, 2, 0, nullptr, nullptr, c->hpart, c->gpart, c->dpart,
                                                  nullptr, PLM_VP_ALWAYS, c->st);
        if (e == hipSuccess) e = plm_launch_hsolve(d, c->hpart, c->gpart, 1, c->x, c->h64, c->prob.lambda_h, 1, c->hinv, c->hg2, c->scal + 5,
                                                   0.0, 0.0, nullptr, 0, c->hcnt, c->dpart, c->st);
        if (e == hipSuccess) e = hipEventRecord(ev[1], c->st);
        if (e == hipSuccess) e = plm_launch_hpass(d, c->hj, c->msa_rm, c->w, c->h64, 1, 1, 0, c->Rt, c->fx_part, c->hpart, c->gpart, nullptr,
                                                  nullptr, PLM_VP_ALWAYS, c->st);
        if (e == hipSuccess) e = plm_launch_hsolve(d, c->hpart, c->gpart, 0, c->x, c->h64, c->prob.lambda_h, 0, c->hinv, c->hg2, c->scal + 5,
                                                   0.0, 0.0, nullptr, 0, c->hcnt, c->dpart, c->st);
        if (e == hipSuccess) e = hipEventRecord(ev[2], c->st);
        if (e == hipSuccess) e = hipEventSynchronize(ev[2]);
        if (e == hipSuccess && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) acc[0] += ms;
        if (e == hipSuccess && hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) acc[1] += ms;
        if (e != hipSuccess) rc = fail(PLM_EDEVICE, "field position timing: %s", hipGetErrorString(e));
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    c->vp_hess_age = 0;
    out_ms[0] = (float)(acc[0] / reps);
    out_ms[1] = (float)(acc[1] / reps);
    return rc;
}

int plm_ctx_time_kernels(plm_ctx_t *c, int32_t reps, float *out_ms) {
    if (!c || !out_ms || reps <= 0) return fail(PLM_EINVAL, "bad argument");
    if (!c->have_weights) return fail(PLM_EINVAL, "weights not set");
    // one shard of the sharded-state mode can be timed alone (its kernels over its own site blocks; the halo buffers are
    // read as they are -- the collectives that fill them are not part of this measurement): scripts/shard_compute.py
    if (c->d.nshards > 1 && !c->d.sharded) return fail(PLM_EUNSUPPORTED, "kernel timing runs on 1-shard or sharded-state contexts");
    HIP_TRY(hipSetDevice(c->device));
    c->eval_valid = false;
    ctx_set_accurate(c, false);   // the evaluation pipeline is timed with the plain arithmetic (the exact forward GEMM separately)
    const PlmDims &d = c->d;
    struct Events {   // destroyed on every exit path (HIP_TRY returns early)
        hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Events() { for (auto &x : e) if (x) (void)hipEventDestroy(x); }
    } evs;
    hipEvent_t (&ev)[6] = evs.e;
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    double acc[PLM_K_COUNT] = {0};
    const bool vp = vp_enabled(c);
    if (vp) PLM_TRY(vp_alloc(c));
    for (int r = 0; r < reps; r++) {
        float ms;
        HIP_TRY(hipEventRecord(ev[0], c->st));
        if (d.sharded) {
            HIP_TRY(plm_launch_maxabs2(c->x + d.nh_pad_l, d.n_local - d.nh_pad_l, c->xhalo,
                                       d.nx_halo * (int64_t)PLM_BLOCK_FLOATS(d), c->maxbits, c->jexp, d.jexp_bias, c->st));
            HIP_TRY(plm_launch_expand(d, c->x, c->xhalo, c->jexp, c->Bt, c->fwd_accurate ? 1 : 0, c->st));
        } else {
            HIP_TRY(plm_launch_maxabs(d, c->x, c->maxbits, c->jexp, c->st));
            HIP_TRY(plm_launch_expand(d, c->x, nullptr, c->jexp, c->Bt, c->fwd_accurate ? 1 : 0, c->st));
        }
        HIP_TRY(hipEventRecord(ev[1], c->st));
        if (vp) {   // the fit's pipeline: forward GEMM -> HJ, 2 Newton steps on the fields, residual pass
            HIP_TRY(plm_launch_forward_store(d, c->msa_rm, c->Bt, c->jexp, c->hj, 0, c->st));
            HIP_TRY(hipEventRecord(ev[2], c->st));
            PLM_TRY(vp_step_and_residuals(c, r == 0));   // one Newton step + the residual pass
            HIP_TRY(hipEventRecord(ev[5], c->st));
        } else {
            PLM_TRY(forward_at_x(c));
            HIP_TRY(hipEventRecord(ev[2], c->st));
            HIP_TRY(hipEventRecord(ev[5], c->st));
        }
        if (d.nblk_own > 0 || !d.sharded) HIP_TRY(plm_launch_backward(d, c->msa_cm, c->Rt, c->G, nullptr, c->st));
        HIP_TRY(hipEventRecord(ev[3], c->st));
        if (d.sharded) HIP_TRY(plm_launch_pack_g(d, c->G, c->gsend, c->st));
        HIP_TRY(plm_launch_assemble(d, c->G, d.ksplit, d.sharded ? c->ghalo : nullptr, c->x, c->g, c->prob.lambda_h, c->prob.lambda_j,
                                    c->reg_part, vp ? 2 : 0, 0.f, nullptr, 0.f, c->st));
        HIP_TRY(plm_launch_finish_fx(d, c->fx_part, c->n_fx_part(), nullptr, 0, c->reg_part, plm_reg_parts(d),
                                     c->scal, c->st));
        HIP_TRY(hipEventRecord(ev[4], c->st));
        HIP_TRY(hipEventSynchronize(ev[4]));
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1])); acc[PLM_K_EXPAND] += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2])); acc[PLM_K_FORWARD] += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[5])); acc[PLM_K_FIELDS] += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[5], ev[3])); acc[PLM_K_BACKWARD] += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[3], ev[4])); acc[PLM_K_ASSEMBLE] += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[4])); acc[PLM_K_TOTAL] += ms;
    }
    {   // the L-BFGS vector kernels of one iteration with a full history of m = 6: trial point, pair + Gram pass, direction
        const int m = 6;
        PLM_TRY(ctx_alloc_lbfgs(c, m));
        const int64_t n = d.n_local;
        float *S = c->hist, *Y = c->hist + (size_t)m * n;
        PlmVecList Qv, B;
        PlmCoefList C;
        Qv.n = 3; Qv.v[0] = S; Qv.v[1] = Y; Qv.v[2] = c->g;
        B.n = 0;
        for (int i = 0; i < m; i++) B.v[B.n++] = S + (size_t)i * n;
        for (int i = 0; i < m; i++) B.v[B.n++] = Y + (size_t)i * n;
        B.v[B.n++] = c->g;
        B.v[B.n++] = c->g;
        for (int i = 0; i < B.n; i++) C.c[i] = 0.f;
        HIP_TRY(hipMemsetAsync(c->hist, 0, sizeof(float) * 2 * (size_t)m * n, c->st));
        HIP_TRY(hipMemsetAsync(c->dir, 0, sizeof(float) * n, c->st));
        HIP_TRY(hipMemcpyAsync(c->xp, c->x, sizeof(float) * n, hipMemcpyDeviceToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->gp, c->g, sizeof(float) * n, hipMemcpyDeviceToDevice, c->st));
        HIP_TRY(hipEventRecord(ev[0], c->st));
        for (int r = 0; r < reps; r++) {
            // as the optimiser issues them: pair + Gram rows + scalar products in one pass, direction + first trial point
            HIP_TRY(plm_launch_sy_multidot(S, Y, c->x, c->xp, c->g, c->gp, c->dir, B, n, d.nh_pad_l, c->dot_scratch, c->scal + 8,
                                           c->scal + 2, nullptr, 6u, 0ull, c->st));
            B.n -= 1;
            HIP_TRY(plm_launch_multiaxpy_trial(c->dir, B, C, n, nullptr, m, c->xp, 0.f, c->x, c->st));
            B.n += 1;
        }
        HIP_TRY(hipEventRecord(ev[1], c->st));
        HIP_TRY(hipEventSynchronize(ev[1]));
        float ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
        acc[PLM_K_LBFGS_VECTOR] = ms;
        HIP_TRY(hipMemsetAsync(c->hist, 0, sizeof(float) * 2 * (size_t)m * n, c->st));
    }
    if (vp && !d.sharded) {   // the exact forward GEMM (last iterations of a fit, plm_eval), on its own operand
        float ms;
        HIP_TRY(plm_launch_expand(d, c->x, nullptr, c->jexp, c->Bt, 1, c->st));
        HIP_TRY(hipEventRecord(ev[0], c->st));
        for (int r = 0; r < reps; r++) HIP_TRY(plm_launch_forward_store(d, c->msa_rm, c->Bt, c->jexp, c->hj, 1, c->st));
        HIP_TRY(hipEventRecord(ev[1], c->st));
        HIP_TRY(hipEventSynchronize(ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
        acc[PLM_K_FORWARD_ACCURATE] = ms;
    }
    {
        const int thresh = cluster_threshold(c->prob.theta_id, d.L, d.conv);
        HIP_TRY(hipEventRecord(ev[0], c->st));
        HIP_TRY(plm_launch_reweight(d, c->msa_rm, std::max(1, thresh), c->counts, c->st));
        HIP_TRY(hipEventRecord(ev[1], c->st));
        HIP_TRY(hipEventSynchronize(ev[1]));
        float ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
        acc[PLM_K_REWEIGHT] = ms * reps;
    }
    for (int k = 0; k < PLM_K_COUNT; k++) out_ms[k] = (float)(acc[k] / reps);
    return PLM_OK;
}

int plm_ctx_solver_stats(plm_ctx_t *c, double *out) {
    if (!c || !out) return fail(PLM_EINVAL, "NULL argument");
    out[PLM_S_EVALS] = c->stat_field_evals;
    out[PLM_S_FIELD_MS] = c->stat_field_ms;
    out[PLM_S_PASSES] = c->stat_passes;
    out[PLM_S_CHAIN_SHORT] = c->stat_chain_short;
    out[PLM_S_GEMM_EVALS] = c->stat_gemm_evals;
    out[PLM_S_FWD_MS] = c->stat_fwd_ms;
    out[PLM_S_BWD_MS] = c->stat_bwd_ms;
    return PLM_OK;
}

// ---------------------------------------------------------------------------------------------
// host-buffer conveniences built on the context API
// ---------------------------------------------------------------------------------------------
static plm_problem_t basic_problem(const int8_t *msa, int n, int l, int q) {
    plm_problem_t p;
    memset(&p, 0, sizeof p);
    p.n_seqs = n; p.n_sites = l; p.n_states = q; p.msa = msa;
    p.theta_id = 0.8; p.scale = 1.0; p.n_shards = 1;
    return p;
}

int plm_reweight(const int8_t *msa, int32_t n_seqs, int32_t n_sites, double theta_id, int32_t *counts_out) {
    return plm_reweight_ex(msa, n_seqs, n_sites, theta_id, 0, counts_out);
}

int plm_reweight_ex(const int8_t *msa, int32_t n_seqs, int32_t n_sites, double theta_id, int32_t flags,
                    int32_t *counts_out) {
    if (!msa || !counts_out) return fail(PLM_EINVAL, "NULL argument");
    // reweighting compares raw bytes; any state value 0..126 is legal here, so borrow q = 21
    // only for the context's tiling and validate the range ourselves
    for (size_t k = 0; k < (size_t)n_seqs * n_sites; k++)
        if (msa[k] < 0 || msa[k] > 126) return fail(PLM_EINVAL, "msa[%zu] outside 0..126", k);
    plm_problem_t p = basic_problem(msa, n_seqs, n_sites, 21);
    p.theta_id = theta_id;
    p.flags = flags & (PLM_FLAG_IGNORE_GAPS | PLM_CONV_MASK);
    PLM_TRY(check_device(0));
    PlmDims d;
    PLM_TRY(make_dims(p, plm_options_from_env(), &d));
    const size_t rm_rows = (size_t)d.Np + 32;
    std::vector<int8_t> rm(rm_rows * d.Lp32, (int8_t)PLM_PAD_STATE);
    for (int s = 0; s < d.N; s++) memcpy(&rm[(size_t)s * d.Lp32], msa + (size_t)s * d.L, d.L);
    int8_t *dev = nullptr;
    int32_t *cnt = nullptr;
    PLM_TRY(dalloc(&dev, rm.size()));
    int rc = dalloc(&cnt, (size_t)d.Np);
    if (rc) { hipFree(dev); return rc; }
    const int thresh = cluster_threshold(theta_id, d.L, d.conv);
    hipError_t e = hipMemcpy(dev, rm.data(), rm.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && thresh > 0) e = plm_launch_reweight(d, dev, thresh, cnt, nullptr);
    if (e == hipSuccess && thresh > 0) e = hipMemcpy(counts_out, cnt, sizeof(int32_t) * d.N, hipMemcpyDeviceToHost);
    if (thresh <= 0) std::fill(counts_out, counts_out + d.N, d.N);
    hipFree(dev);
    hipFree(cnt);
    if (e != hipSuccess) return fail(PLM_EDEVICE, "reweight failed: %s", hipGetErrorString(e));
    return PLM_OK;
}

int plm_marginals(const int8_t *msa, const float *weights, int32_t n_seqs, int32_t n_sites, int32_t n_states,
                  float *fi_out, float *fij_out) {
    if (!msa || !weights || !fi_out) return fail(PLM_EINVAL, "NULL argument");
    plm_problem_t p = basic_problem(msa, n_seqs, n_sites, n_states);
    plm_ctx_t *c = nullptr;
    PLM_TRY(plm_ctx_create(&p, 0, nullptr, &c));
    int rc = plm_ctx_set_weights(c, weights);
    if (!rc) rc = plm_ctx_marginals(c, fi_out, fij_out);
    plm_ctx_destroy(c);
    return rc;
}

int plm_eval(const int8_t *msa, const float *weights, int32_t n_seqs, int32_t n_sites, int32_t n_states,
             double lambda_h, double lambda_j, const float *x, double *fx_out, double *nll_out, float *g_out) {
    if (!msa || !weights || !x) return fail(PLM_EINVAL, "NULL argument");
    plm_problem_t p = basic_problem(msa, n_seqs, n_sites, n_states);
    p.lambda_h = lambda_h;
    p.lambda_j = lambda_j;
    plm_ctx_t *c = nullptr;
    PLM_TRY(plm_ctx_create(&p, 0, nullptr, &c));
    int rc = plm_ctx_set_weights(c, weights);
    if (!rc) rc = plm_ctx_set_x(c, x);
    double fx = 0, nll = 0;
    if (!rc) rc = plm_ctx_eval(c, &fx, &nll);
    if (!rc && g_out) rc = plm_ctx_get_g(c, g_out);
    if (fx_out) *fx_out = fx;
    if (nll_out) *nll_out = nll;
    plm_ctx_destroy(c);
    return rc;
}

int plm_scores(const float *jij, int32_t n_sites, int32_t n_states, float *fn_out, float *cn_out) {
    return plm_scores_ex(jij, n_sites, n_states, 0, fn_out, cn_out);
}

int plm_scores_ex(const float *jij, int32_t n_sites, int32_t n_states, int32_t flags, float *fn_out, float *cn_out) {
    if (!jij || !fn_out || !cn_out) return fail(PLM_EINVAL, "NULL argument");
    if (n_sites < 2 || n_states < 1 || n_states > 32) return fail(PLM_EINVAL, "bad L / q");
    std::vector<int8_t> dummy((size_t)n_sites, 0);
    plm_problem_t p = basic_problem(dummy.data(), 1, n_sites, plm_q_supported(n_states) ? n_states : 21);
    PLM_TRY(check_device(0));
    PlmDims d;
    PLM_TRY(make_dims(p, plm_options_from_env(), &d));
    d.Q = n_states;
    const size_t nj = (size_t)n_sites * (n_sites - 1) / 2 * n_states * n_states;
    float *dj = nullptr, *dfn = nullptr;
    PLM_TRY(dalloc(&dj, nj));
    int rc = dalloc(&dfn, (size_t)n_sites * n_sites);
    if (rc) { hipFree(dj); return rc; }
    hipError_t e = hipMemcpy(dj, jij, sizeof(float) * nj, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = plm_launch_fn(d, dj, dfn, (flags & PLM_CONV_FN_NO_GAP) ? 1 : 0, 0, nullptr);
    if (e == hipSuccess) e = hipMemcpy(fn_out, dfn, sizeof(float) * n_sites * n_sites, hipMemcpyDeviceToHost);
    hipFree(dj);
    hipFree(dfn);
    if (e != hipSuccess) return fail(PLM_EDEVICE, "scores failed: %s", hipGetErrorString(e));
    const int L = n_sites;
    std::vector<double> col(L, 0.0);
    double tot = 0;
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++) {
            col[j] += fn_out[(size_t)i * L + j];
            tot += fn_out[(size_t)i * L + j];
        }
    const double mean = tot / ((double)L * (L - 1));
    for (int j = 0; j < L; j++) col[j] /= (double)(L - 1);
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++)
            cn_out[(size_t)i * L + j] =
                (i == j) ? 0.f : (float)((double)fn_out[(size_t)i * L + j] - col[i] * col[j] / mean);
    return PLM_OK;
}

// statistical energies / potentials of n sequences under a model: upload, canonical -> native, expand,
// forward GEMM with the energy epilogue (no weights, no backward pass, no optimiser state)
static int energies_impl(const int8_t *seqs, int32_t n, int32_t L, int32_t q, const float *x_canon, int device,
                         void *stream, int potentials, void *out_host) {
    if (!seqs || !x_canon || !out_host) return fail(PLM_EINVAL, "NULL argument");
    PLM_TRY(check_device(device));
    plm_problem_t p = basic_problem(seqs, n, L, q);
    const PlmOptions opt = plm_options_from_env();
    // PLM_FWD_ACCURATE=1 (measurements, tests/probes/potentials_probe.py): the potentials from the accurate forward GEMM
    const int accurate = (potentials && opt.fwd_mode == 1) ? 1 : 0;
    PlmDims d;
    PLM_TRY(make_dims(p, opt, &d));
    if ((int64_t)(d.Np + 32) * d.Lp32 >= (int64_t)1 << 31) return fail(PLM_EINVAL, "too many sequences for one call");
    for (size_t k = 0; k < (size_t)n * L; k++)
        if (seqs[k] < 0 || seqs[k] >= q) return fail(PLM_EINVAL, "seqs[%zu] = %d outside 0..%d", k, (int)seqs[k], q - 1);
    hipStream_t st = (hipStream_t)stream;
    const size_t rm_rows = (size_t)d.Np + 32;
    std::vector<int8_t> rm(rm_rows * d.Lp32, (int8_t)PLM_PAD_STATE);
    for (int s = 0; s < d.N; s++) memcpy(&rm[(size_t)s * d.Lp32], seqs + (size_t)s * d.L, d.L);
    const int nblk = d.b16_hi - d.b16_lo;
    const int ngrp = plm_fwd_groups(d.Q, 0);     // energy partials per (sequence, site block, state group)
    const size_t out_dev_floats = potentials ? (size_t)d.N * d.L * d.Qc : (size_t)d.Np * nblk * ngrp * 2;
    int8_t *msa_rm = nullptr;
    float *canon = nullptr, *x = nullptr, *outd = nullptr;
    char *Bt = nullptr;
    uint32_t *maxbits = nullptr;
    int32_t *jexp = nullptr;
    double *en = nullptr;
    auto cleanup = [&](int code) {
        void *all[] = {msa_rm, canon, x, outd, Bt, maxbits, jexp, en};
        for (void *b : all)
            if (b) (void)hipFree(b);
        return code;
    };
    int rc;
    if ((rc = dalloc(&msa_rm, rm.size())) || (rc = dalloc(&canon, (size_t)d.n_canon)) ||
        (rc = dalloc(&x, (size_t)d.n_native)) || (rc = dalloc(&outd, out_dev_floats)) ||
        (rc = dalloc(&Bt, plm_bt_bytes(d))) || (rc = dalloc(&maxbits, (size_t)1)) || (rc = dalloc(&jexp, (size_t)1)) ||
        (rc = dalloc(&en, (size_t)d.N * 3)))
        return cleanup(rc);
    hipError_t e;
#define ET(expr)                                                                                   \
    if ((e = (expr)) != hipSuccess)                                                                \
        return cleanup(fail(e == hipErrorOutOfMemory ? PLM_ENOMEM : PLM_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e)));
    ET(hipMemcpyAsync(msa_rm, rm.data(), rm.size(), hipMemcpyHostToDevice, st));
    ET(hipMemcpyAsync(canon, x_canon, sizeof(float) * d.n_canon, hipMemcpyHostToDevice, st));
    ET(hipMemsetAsync(x, 0, sizeof(float) * d.n_native, st));
    ET(plm_launch_canon_to_native(d, canon, x, st));
    ET(plm_launch_maxabs(d, x, maxbits, jexp, st));
    ET(plm_launch_expand(d, x, nullptr, jexp, Bt, accurate, st));
    ET(plm_launch_forward_energy(d, msa_rm, Bt, x, jexp, potentials, accurate, outd, st));
    if (potentials) {
        ET(hipMemcpyAsync(out_host, outd, sizeof(float) * out_dev_floats, hipMemcpyDeviceToHost, st));
    } else {
        ET(plm_launch_energy_sum(d, outd, en, st));
        ET(hipMemcpyAsync(out_host, en, sizeof(double) * (size_t)d.N * 3, hipMemcpyDeviceToHost, st));
    }
    ET(hipStreamSynchronize(st));
#undef ET
    return cleanup(PLM_OK);
}

int plm_hamiltonians(const int8_t *seqs, int32_t n, int32_t n_sites, int32_t n_states, const float *x_canonical,
                     int device, void *stream, double *energies_out) {
    return energies_impl(seqs, n, n_sites, n_states, x_canonical, device, stream, 0, energies_out);
}
int plm_potentials(const int8_t *seqs, int32_t n, int32_t n_sites, int32_t n_states, const float *x_canonical,
                   int device, void *stream, float *potentials_out) {
    return energies_impl(seqs, n, n_sites, n_states, x_canonical, device, stream, 1, potentials_out);
}

int plm_meanfield(const int8_t *msa, int32_t n_seqs, int32_t n_sites, int32_t n_states, double theta_id,
                  double pseudo_count, int device, void *stream, plm_mf_result_t *out) {
    if (!msa || !out) return fail(PLM_EINVAL, "NULL argument");
    if (!(pseudo_count > 0.0 && pseudo_count < 1.0)) return fail(PLM_EINVAL, "pseudo_count must be in (0, 1)");
    plm_problem_t p = basic_problem(msa, n_seqs, n_sites, n_states);
    p.theta_id = theta_id;
    plm_ctx_t *c = nullptr;
    PLM_TRY(plm_ctx_create(&p, device, stream, &c));
    const PlmDims d = c->d;
    const size_t lq = (size_t)d.L * d.Qc, pq = (size_t)d.L * (d.L - 1) / 2 * d.Qc * d.Qc, llqq = lq * lq;
    double *hi = nullptr, *jfull = nullptr, *di = nullptr;
    float *jp = nullptr;
    auto done = [&](int code) {
        void *all[] = {hi, jfull, di, jp};
        for (void *b : all)
            if (b) (void)hipFree(b);
        plm_ctx_destroy(c);
        return code;
    };
    int rc = plm_ctx_reweight(c);
    if (!rc) rc = plm_ctx_get_weights(c, out->weights, nullptr, &out->n_eff);
    if (!rc) rc = plm_ctx_marginals(c, out->fi, out->fij);      // leaves [f_i | f_ij blocks] in c->canon
    if (rc) return done(rc);
    if ((out->hi && (rc = dalloc(&hi, lq))) || (out->jij_full && (rc = dalloc(&jfull, llqq))) ||
        (out->di && (rc = dalloc(&di, (size_t)d.L * d.L))) || (out->jij && (rc = dalloc(&jp, pq))))
        return done(rc);
    rc = plm_meanfield_device(c->canon, c->canon + lq, d.L, d.Qc, pseudo_count, c->st, hi, jfull, jp, di);
    if (rc) return done(rc);
    hipError_t e = hipSuccess;
    if (hi && e == hipSuccess) e = hipMemcpy(out->hi, hi, sizeof(double) * lq, hipMemcpyDeviceToHost);
    if (jfull && e == hipSuccess) e = hipMemcpy(out->jij_full, jfull, sizeof(double) * llqq, hipMemcpyDeviceToHost);
    if (di && e == hipSuccess) e = hipMemcpy(out->di, di, sizeof(double) * (size_t)d.L * d.L, hipMemcpyDeviceToHost);
    if (jp && e == hipSuccess) e = hipMemcpy(out->jij, jp, sizeof(float) * pq, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return done(fail(PLM_EDEVICE, "download of the mean-field result failed: %s", hipGetErrorString(e)));
    return done(PLM_OK);
}

int plm_direct_information(const double *jij_full, const double *fi, int32_t n_sites, int32_t n_states, int device,
                           void *stream, double *di_out) {
    if (!jij_full || !fi || !di_out || n_sites < 2) return fail(PLM_EINVAL, "NULL argument or fewer than 2 sites");
    PLM_TRY(check_device(device));
    const size_t L = (size_t)n_sites, q = (size_t)n_states;
    double *J = nullptr, *f = nullptr, *di = nullptr;
    auto done = [&](int code) {
        void *all[] = {J, f, di};
        for (void *b : all)
            if (b) (void)hipFree(b);
        return code;
    };
    int rc;
    if ((rc = dalloc(&J, L * L * q * q)) || (rc = dalloc(&f, L * q)) || (rc = dalloc(&di, L * L))) return done(rc);
    hipError_t e = hipMemcpy(J, jij_full, sizeof(double) * L * L * q * q, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(f, fi, sizeof(double) * L * q, hipMemcpyHostToDevice);
    if (e != hipSuccess) return done(fail(PLM_EDEVICE, "upload failed: %s", hipGetErrorString(e)));
    rc = plm_direct_information_device(J, f, n_sites, n_states, (hipStream_t)stream, di);
    if (rc) return done(rc);
    e = hipMemcpy(di_out, di, sizeof(double) * L * L, hipMemcpyDeviceToHost);
    return done(e == hipSuccess ? PLM_OK : fail(PLM_EDEVICE, "download failed: %s", hipGetErrorString(e)));
}

int plm_alignment_stats(const int8_t *msa, int32_t n, int32_t L, int32_t gap_state, const int8_t *query,
                        int32_t *seq_gaps, int32_t *col_gaps, int32_t *ident, int device, void *stream) {
    if (!msa || n <= 0 || L <= 0) return fail(PLM_EINVAL, "NULL alignment or empty shape");
    if (ident && !query) return fail(PLM_EINVAL, "identities need a query sequence");
    if (gap_state < 0 || gap_state > 126) return fail(PLM_EINVAL, "gap state outside 0..126");
    for (size_t k = 0; k < (size_t)n * L; k++)
        if (msa[k] < 0) return fail(PLM_EINVAL, "msa[%zu] is negative", k);     // the packed compare needs bytes < 0x80
    if (query)
        for (int k = 0; k < L; k++)
            if (query[k] < 0) return fail(PLM_EINVAL, "query[%d] is negative (states must be 0..127)", k);
    PLM_TRY(check_device(device));
    hipStream_t st = (hipStream_t)stream;
    int8_t *dm = nullptr, *dq = nullptr;
    int32_t *dsg = nullptr, *dcg = nullptr, *did = nullptr;
    auto done = [&](int code) {
        void *all[] = {dm, dq, dsg, dcg, did};
        for (void *b : all)
            if (b) (void)hipFree(b);
        return code;
    };
    int rc;
    if ((rc = dalloc(&dm, (size_t)n * L + 16)) || (query && (rc = dalloc(&dq, (size_t)L))) ||
        (seq_gaps && (rc = dalloc(&dsg, (size_t)n))) || (col_gaps && (rc = dalloc(&dcg, (size_t)L))) ||
        (ident && (rc = dalloc(&did, (size_t)n))))
        return done(rc);
    hipError_t e = hipMemcpyAsync(dm, msa, (size_t)n * L, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && query) e = hipMemcpyAsync(dq, query, (size_t)L, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = plm_launch_align_stats(dm, n, L, gap_state, dq, dsg, dcg, did, st);
    if (e == hipSuccess && seq_gaps) e = hipMemcpyAsync(seq_gaps, dsg, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && col_gaps) e = hipMemcpyAsync(col_gaps, dcg, sizeof(int32_t) * L, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && ident) e = hipMemcpyAsync(ident, did, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(fail(PLM_EDEVICE, "alignment statistics failed: %s", hipGetErrorString(e)));
    return done(PLM_OK);
}

static int fit_impl(const plm_problem_t *problem, plm_result_t *result, int device, void *stream, plm_iter_cb iter_cb,
                    void *iter_user, plm_exchange_cb exchange, void *exchange_user, plm_collective_cb collective,
                    void *collective_user, const void *rccl_id = nullptr) {
    if (!problem || !result) return fail(PLM_EINVAL, "NULL problem / result");
    const double t0 = now_s();
    const int nshards = problem->n_shards > 0 ? problem->n_shards : 1;
    // reweighting and marginals are cheap and run unsharded on every rank (identical inputs ->
    // identical weights everywhere); the optimisation runs on the sharded context
    plm_problem_t p1 = *problem;
    p1.n_shards = 1;
    p1.shard = 0;
    plm_ctx_t *c1 = nullptr;
    PLM_TRY(plm_ctx_create(&p1, device, stream, &c1));
    int rc = PLM_OK;
    const int N = problem->n_seqs, L = problem->n_sites, q = problem->n_states;
    const size_t npq = (size_t)L * (L - 1) / 2 * q * q;
    std::vector<float> w(N), fi((size_t)L * q);
    double t1 = now_s();
    rc = plm_ctx_reweight(c1);
    float neff = 0;
    if (!rc) rc = plm_ctx_get_weights(c1, w.data(), nullptr, &neff);
    result->seconds_reweight = now_s() - t1;
    t1 = now_s();
    // PLM_FLAG_COMPACT_GAPS: the pair arrays lose the gap state's row and column on the device, before the download
    const bool compact = (problem->flags & PLM_FLAG_COMPACT_GAPS) && (problem->flags & PLM_FLAG_IGNORE_GAPS);
    float *cbuf = nullptr;
    if (!rc && compact && (result->fij || result->jij) && q > 1 &&
        hipMalloc((void **)&cbuf, sizeof(float) * (size_t)L * (L - 1) / 2 * (q - 1) * (q - 1) + 16) != hipSuccess)
        rc = fail(PLM_ENOMEM, "out of device memory (gap compaction buffer)");
    if (!rc) rc = ctx_marginals(c1, fi.data(), result->fij, cbuf);
    result->seconds_marginals = now_s() - t1;
    plm_ctx_t *c = c1;
    if (!rc && nshards > 1) {
        rc = plm_ctx_create(problem, device, stream, &c);
        if (!rc) rc = plm_ctx_set_weights(c, w.data());
        if (!rc) {
            c->h_fi = c1->h_fi;
            plm_ctx_set_exchange(c, exchange, exchange_user);
            plm_ctx_set_collective(c, collective, collective_user);
            if (rccl_id) rc = plm_ctx_attach_rccl(c, rccl_id);
        }
        plm_ctx_destroy(c1);
        c1 = nullptr;
    }
    if (!rc) rc = plm_ctx_set_x(c, nullptr);
    if (!rc) rc = plm_ctx_optimize(c, iter_cb, iter_user, result);
    if (!rc && (result->hi || result->jij)) rc = get_vec_split(c, c->x, result->hi, result->jij, cbuf);
    if (cbuf) (void)hipFree(cbuf);
    if (!rc && (result->fn || result->cn)) {   // either score matrix may be asked for on its own
        std::vector<float> spare;
        float *fn = result->fn, *cn = result->cn;
        if (!fn || !cn) {
            spare.resize((size_t)c->d.L * c->d.L);
            (fn ? cn : fn) = spare.data();
        }
        rc = plm_ctx_scores(c, fn, cn);
    }
    if (!rc) {
        if (result->weights) memcpy(result->weights, w.data(), sizeof(float) * N);
        if (result->fi && compact)
            for (int i = 0; i < L; i++) memcpy(result->fi + (size_t)i * (q - 1), &fi[(size_t)i * q + 1], sizeof(float) * (q - 1));
        else if (result->fi)
            memcpy(result->fi, fi.data(), sizeof(float) * L * q);
        result->n_eff = neff;
    }
    if (c) plm_ctx_destroy(c);
    result->seconds_total = now_s() - t0;
    return rc;
}

int plm_fit(const plm_problem_t *problem, plm_result_t *result, int device, void *stream, plm_iter_cb iter_cb,
            void *iter_user, plm_exchange_cb exchange, void *exchange_user) {
    if (problem && (problem->flags & PLM_FLAG_SHARDED_STATE) && problem->n_shards > 1)
        return fail(PLM_EINVAL, "PLM_FLAG_SHARDED_STATE needs plm_fit_sharded (collective callback)");
    return fit_impl(problem, result, device, stream, iter_cb, iter_user, exchange, exchange_user, nullptr, nullptr);
}

int plm_fit_sharded(const plm_problem_t *problem, plm_result_t *result, int device, void *stream,
                    plm_iter_cb iter_cb, void *iter_user, plm_collective_cb collective, void *collective_user) {
    if (!problem || !result) return fail(PLM_EINVAL, "NULL problem / result");
    if (problem->n_shards > 1 && !collective) return fail(PLM_EINVAL, "n_shards > 1 needs a collective callback");
    plm_problem_t p = *problem;
    p.flags |= PLM_FLAG_SHARDED_STATE;
    return fit_impl(&p, result, device, stream, iter_cb, iter_user, nullptr, nullptr, collective, collective_user);
}

int plm_fit_sharded_rccl(const plm_problem_t *problem, plm_result_t *result, int device, void *stream,
                         plm_iter_cb iter_cb, void *iter_user, const void *rccl_id) {
    if (!problem || !result) return fail(PLM_EINVAL, "NULL problem / result");
    if (problem->n_shards > 1 && !rccl_id) return fail(PLM_EINVAL, "n_shards > 1 needs the communicator id");
    plm_problem_t p = *problem;
    p.flags |= PLM_FLAG_SHARDED_STATE;
    return fit_impl(&p, result, device, stream, iter_cb, iter_user, nullptr, nullptr, nullptr, nullptr,
                    problem->n_shards > 1 ? rccl_id : nullptr);
}

}  // extern "C"
