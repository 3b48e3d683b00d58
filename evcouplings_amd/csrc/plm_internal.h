// plm_internal.h -- data layout in HBM and launch-wrapper declarations shared by
// plm_kernels.hip (device code) and plm_host.cpp (context, L-BFGS, C ABI).
//
// HBM layout (all sizes for N sequences, L sites, q states):
//   msa_rm  int8 [Np][Lp32]        row-major alignment, pad value 127 (never a state)
//   msa_cm  int8 [(nb16+1)*16][Np] column-major copy; the extra 16-site block holds the
//                                  "ones" column (site 0 = state 0 for s < N) that turns the
//                                  field gradient into one more row fragment of the GEMM
//   x, g    f32  [n_native]        "native" parameter vector: h[L][q] padded to 256 floats,
//                                  then per block pair (I<=J of 16-site blocks), per (a,b):
//                                  a 16x16 tile [ii][jj] of J_{16I+ii,16J+jj}(a,b)
//   Bt      f16  [b16][kstep][2][q][64][8]   one-hot GEMM B operand of the forward pass:
//                                  expanded couplings split hi/lo, stored as ready-made MFMA
//                                  B fragments (lane-linear 16 B per lane)
//   Rt      i8   [3 planes][step128][nf][2][64][16]   residuals w_s (P_si(a) - [x_si=a]) in 24-bit fixed point
//                                  (R = rint(r * rscale), |R| <= 8 355 711), split into three SIGNED base-256 digits
//                                  (R = d0 + 256 d1 + 65536 d2, each digit in [-128, 127]); plane p holds digit p as
//                                  ready-made B fragments of v_mfma_i32_16x16x64_i8: per 128-sequence K step and column
//                                  fragment two 1 KB fragments (sequences 0-63 / 64-127), lane (G = lane / 16, site =
//                                  lane % 16) holding the 16 consecutive sequences 16 G .. 16 G + 15
//   G       i32  [3 planes][ksplit][mf][nfl][64][4]   asymmetric gradient slab per digit plane and K range: EXACT
//                                  integer sums (one 16x16 MFMA accumulator tile per (row fragment, col fragment));
//                                  consumers combine g = gscale * sum_k (G0 + 256 G1 + 65536 G2)
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime.h>

#define PLM_MAX_SHARDS 16     // sharded-state mode: shards of one problem (PlmDims carries a table per partner)
#define PLM_PAD_STATE 127
#define PLM_SEQ_TILE 256      // sequences per forward workgroup (8 waves x 32)
#define PLM_R_EXP 14          // forward operand: couplings are stored scaled by 2^(14 - exponent of max|J|)
// Backward GEMM on the int8 matrix cores: residuals in 24-bit fixed point, three signed base-256 digit
// planes, ONE plane per workgroup (the plane index takes the place of most of the split-K factor), int32 accumulators.
// The largest magnitude whose digits all fit int8 is 127 * (65536 + 256 + 1) = 8 355 711; the scale maps the largest
// weight a little below it, so that the float rounding of w * rscale * (P - delta) (|P - delta| <= 1) cannot cross it.
// Three planes (24 bits) are the default: their quantisation noise in the gradient, ~2^-24 w_max sqrt(2 L^2 q N / 12) in
// norm, is ~5e-5 |x| at the headline, far inside the stop rule 1e-3.  A fit asked to converge below 1e-4 (tests that
// compare optima at tight tolerance) runs FOUR planes (32 bits: every bit of the f32 residuals) at 4/3 of the cost.
#define PLM_BWD_MAXPLANES 4
#define PLM_R_QMAX3 8355000.0f          // < 127 * (65536 + 256 + 1)
#define PLM_R_QMAX4 2138000000.0f       // < 127 * (16777216 + 65536 + 256 + 1) = 2 139 062 143
#define PLM_BWD_KSTEP 128     // sequences per K step of k_bwd (two v_mfma_i32_16x16x64_i8 sub-steps)
#define PLM_BWDW_COLS 9        // column fragments of a k_bwd_w workgroup tile (= PLM_BWDW_FN of plm_bwd_asm.inc)
#define PLM_BWD_ONEHOT_VALUE (-128)   // the one-hot operand of k_bwd holds -128 for a match (two VALU ops per 4 sites)

// Forward GEMM on the 2:4 sparse MFMA (v_smfmac_f32_16x16x64_f16): the K index is ordered (site, state) with the
// alphabet padded to a multiple of 4, so that every group of 4 dense K slots holds 4 states of ONE site -- at most
// one non-zero of the one-hot operand, the pattern the instruction requires, by construction.
// State 0 is the reference state and has no K slots: sum_j J_ij(a, x_sj) = sum_j J_ij(a, 0) + sum_{j: x_sj != 0}
// (J_ij(a, x_sj) - J_ij(a, 0)) -- the first sum is a constant per (i, a) (k_fwd_ref, added in k_fwd's epilogues), the
// GEMM runs on the differences and on Q - 1 states: 20 = 5 groups of 4 for the protein alphabet, no padding.
// A K step covers, for the 32 sites of a block u, one of 2 * NG instruction slices (NG = ceil((Q - 1) / 4) state
// groups per site, 4 (site, group) pairs per lane and instruction) and one of the two f16 planes of the operand (hi +
// lo: 22 bits of every coupling difference): 4 NG tiles per block.  The exact forward GEMM (k_fwd_x: int8 matrix cores,
// five signed base-256 digit planes of the 39-bit fixed-point differences) has equally large tiles, per block u NG
// state groups x 5 planes; Bt is allocated for those.
#define PLM_FWD_NG(Q) (((Q) + 2) / 4)
#define PLM_FWD_SPU(Q) (4 * PLM_FWD_NG(Q))        // K-step tiles per 32-site block, k_fwd
#define PLM_FWD_TILES(Q) (5 * PLM_FWD_NG(Q))      // ... of the larger of the two operands (allocation)

struct PlmDims {
    int N, L, Q;       // Q: alphabet size the kernels are instantiated for (4, 5, 20, 21) -- the native layout's stride
    int Qc;            // alphabet size of the problem (2..Q): stride of every canonical-layout array at the API;
                       // states Qc..Q-1 exist only as structurally-zero padding of the native layout
    int Np;        // N padded to PLM_SEQ_TILE
    int nb16;      // 16-site blocks covering L
    int Lp16;      // nb16 * 16
    int nu;        // 32-site K blocks covering L
    int Lp32;      // nu * 32
    int nksteps;   // nu * PLM_FWD_TILES(Q): K-step tiles (2 Q KB each) Bt has room for per column block
    int nssteps;   // Np / 32     (32-sequence groups: the wave granularity of the forward-side kernels)
    int nst128;    // Np / 128    backward K steps (PLM_BWD_KSTEP sequences)
    int nplanes;   // digit planes of the residuals: 3 (24-bit fixed point) or 4 (32-bit)
    float rscale;  // residual quantisation: R = rint(r * rscale), rscale = PLM_R_QMAX{3,4} / max_s w_s (set with the weights)
    float gscale;  // 1 / rscale: turns the integer gradient sums back into residual units
    int nstiles;   // Np / PLM_SEQ_TILE
    int FM, FN;    // backward wave tile in fragments
    int nmf;       // row fragments of the backward GEMM: nb16*Q + FM (last FM = "ones" block)
    int nshards, shard;
    int blk_per_shard;  // ceil(nb16 / nshards): the most column blocks a shard owns (slab width)
    int shard_base, shard_rem;   // balanced partition: the LAST shard_rem shards own shard_base + 1 blocks, the others shard_base
    int b16_lo, b16_hi; // this shard's column blocks [lo, hi)
    int nnfl;      // local col fragments = blk_per_shard * Q (slab width, padded)
    int ksplit;    // split-K factor of the backward GEMM
    int nrow_tiles, ncol_tiles; // backward workgroup grid
    int fwd_w;         // plain forward GEMM of the fit by k_fwd_w (512 sequences per workgroup, K loop in assembly)
    int bwd_w;         // backward GEMM by k_bwd_w (wave tile 7 x 9 in AccVGPRs, K step in assembly) instead of k_bwd
    int64_t nbp;       // block pairs I<=J
    int64_t nh_pad;    // L*Q rounded up to 256
    int64_t n_native;  // nh_pad + nbp*Q*Q*256
    int64_t n_canon;   // L*Q + L(L-1)/2*Q*Q
    int gap_mode;      // 1: state 0 (gap) excluded from the model (plmc -g)
    int jexp_bias;     // measurement knob (PlmOptions::jexp_bias): added to the scale exponent of the forward operand
    int conv;          // PLM_CONV_* convention switches (include/plm_hip.h)
    double theta;      // identity threshold (PLM_CONV_G_UNGAPPED_LENGTH evaluates it per pair)
    // ---- sharded-state mode (PLM_FLAG_SHARDED_STATE): parameters, gradient and L-BFGS vectors are split over the shards;
    // the "local" vector is [h of own sites | own block pairs].  Own block pairs (round 6: balanced): the TRIANGLE of
    // pairs (I <= J) with both blocks own, then this shard's HALF of every rectangle of pairs it shares with another
    // shard p: of the rectangle of shards s < t (rows = blocks of s) the even rows belong to s, the odd rows to t
    // (plm_pair_owner) -- every shard owns about half of each of its rectangles, for any number of shards.  (Round 5: all
    // of them went to the lower shard -- shard 0 of 8 held 23 % of the state at L = 500, the last one 2 %.)  A shard's
    // rows of a rectangle are numbered row-major, (row >> 1) * n_t + (J - lo_t); the rectangles follow each other in
    // partner order.  In every other mode the local vector IS the native vector (own_lo = 0, own_hi = nb16).
    int sharded;       // 1 in sharded-state mode
    int own_lo, own_hi;   // blocks whose parameters live in the local vector
    int nblk_own;      // own_hi - own_lo
    int h_site0;       // first site of the local field part
    int64_t bp_base;   // plm_bp_index(own_lo, own_lo): first own block pair
    int64_t np_own;    // own block pairs
    int64_t nh_pad_l;  // local field part, padded to 256 floats
    int64_t n_local;   // nh_pad_l + np_own*Q*Q*256
    int64_t nx_halo;   // coupling blocks received per evaluation: the partners' halves of the shared rectangles
    int64_t ng_halo;   // gradient blocks received per evaluation: this shard's halves
    int ntri;          // own pairs with both blocks own: nblk_own (nblk_own + 1) / 2 (all of them outside sharded-state mode)
    // per partner shard p, in partner order: own_base[p] = first block of THIS shard's half of the rectangle shared with p
    // among its halves -- its place behind the triangle in the local vector (ntri + own_base[p]) = in the coupling message
    // sent to p = in ghalo (p's gradient fragments for it); oth_base[p] = first block of p's half among the partners'
    // halves -- its place in xhalo (p's couplings) = in the gradient message sent to p.
    int own_base[PLM_MAX_SHARDS], oth_base[PLM_MAX_SHARDS];
};

// Environment knobs of the library, read ONCE per context (plm_options_from_env, plm_host.cpp) -- nothing on the
// evaluation path calls getenv.  All of them are measurement / debugging aids; the product behaviour is the default.
struct PlmOptions {
    int bwd_planes = 0;     // PLM_BWD_PLANES = 3 | 4: digit planes of the backward GEMM (0: chosen from epsilon)
    int ksplit = 0;         // PLM_KSPLIT: K split of the backward GEMM (0: cost model); results are identical for every value
    int fwd_kernel = -1;    // PLM_FWD_KERNEL: 0 = k_fwd everywhere, otherwise k_fwd_w where it exists (21 states, the fit's store mode)
    int bwd_kernel = -1;    // PLM_BWD_KERNEL: 0 = k_bwd everywhere, otherwise k_bwd_w where it exists (21 states)
    int jexp_bias = 0;      // PLM_JEXP_BIAS: added to the scale exponent of the forward operand (tests/probes/noise_probe.py)
    int fwd_mode = -1;      // PLM_FWD_ACCURATE = 0 | 1: force the plain / the exact forward GEMM (-1: the solver decides)
    double acc_factor = 8.0;   // PLM_ACC_FACTOR: the fit switches to the accurate evaluation below max(3 eps, this x 3e-11 N L)
    double vp_rel = 1e-4;   // PLM_VP_REL: field-solver tolerance relative to the reduced gradient of the last accepted point
    int vp_hess_pos = -1;   // PLM_VP_HESS_POS: chain positions that take fresh Hessian sums at most (-1: every one before the expected last)
    double vp_floor = 2e-7; // PLM_VP_FLOOR: noise floor of the field solver's tolerance (scripts/vp_floor_probe.py)
    bool debug = false;     // PLM_DEBUG: line-search failures are traced to stderr
    bool debug_vp = false;  // PLM_DEBUG_VP: every round of the field solver is traced to stderr
};
PlmOptions plm_options_from_env();

// balanced partition of the nb16 column blocks over the shards: the LAST nb16 % nshards shards own one block more (19
// blocks on 8 GPUs: 2,2,2,2,2,3,3,3 -- no idle GPU).  The surplus blocks sit at the high end because a shard also owns
// the block pairs (I own, J >= I): the low shards hold the long rows of the triangle (their share of the L-BFGS vectors
// and of k_assemble), the high shards almost none -- at the headline the busiest rank then owns 3 blocks and 24 block
// pairs instead of 3 blocks and 54 (round 6).
static inline __host__ __device__ int plm_shard_lo(const PlmDims &d, int r) {
    const int small = d.nshards - d.shard_rem;          // shards with shard_base blocks come first
    return r * d.shard_base + (r > small ? r - small : 0);
}
static inline __host__ __device__ int plm_shard_cnt(const PlmDims &d, int r) {
    return d.shard_base + (r >= d.nshards - d.shard_rem ? 1 : 0);
}
static inline __host__ __device__ int plm_shard_of(const PlmDims &d, int b) {
    const int small = d.nshards - d.shard_rem, nsmall = small * d.shard_base;
    if (b < nsmall) return d.shard_base > 0 ? b / d.shard_base : 0;
    return small + (b - nsmall) / (d.shard_base + 1);
}
static inline __host__ __device__ int64_t plm_bp_index(int I, int J, int nb16) { // I <= J
    return (int64_t)I * nb16 - (int64_t)I * (I - 1) / 2 + (J - I);
}
// sharded-state mode: the block pair (I < J) of two different shards sI < sJ belongs to sI when I is an even row of the
// rectangle (counted from sI's first block), else to sJ
static inline __host__ __device__ int plm_pair_owner(const PlmDims &d, int I, int sI, int sJ) {
    return ((I - plm_shard_lo(d, sI)) & 1) ? sJ : sI;
}
// blocks in this shard's half of the rectangle it shares with shard p
static inline __host__ __device__ int64_t plm_half_blocks(const PlmDims &d, int p) {
    return d.shard < p ? (int64_t)((d.nblk_own + 1) / 2) * plm_shard_cnt(d, p) : (int64_t)(plm_shard_cnt(d, p) / 2) * d.nblk_own;
}
// Local number of the block pair (I <= J) among this context's own pairs, or -1; for a pair this shard does not own but
// whose couplings it needs (one of the blocks is an own column block) *halo = its place in xhalo (else -1).
static inline __host__ __device__ int64_t plm_pair_local(const PlmDims &d, int I, int J, int *halo) {
    if (halo) *halo = -1;
    if (!d.sharded) return plm_bp_index(I, J, d.nb16);
    const int sI = plm_shard_of(d, I), sJ = plm_shard_of(d, J);
    if (sI == d.shard && sJ == d.shard) {
        const int r = I - d.own_lo;
        return (int64_t)r * d.nblk_own - (int64_t)r * (r - 1) / 2 + (J - I);
    }
    if (sI != d.shard && sJ != d.shard) return -1;
    const int p = sI == d.shard ? sJ : sI;
    const int idx = ((I - plm_shard_lo(d, sI)) >> 1) * plm_shard_cnt(d, sJ) + (J - plm_shard_lo(d, sJ));
    if (plm_pair_owner(d, I, sI, sJ) == d.shard) return (int64_t)d.ntri + d.own_base[p] + idx;
    if (halo) *halo = d.oth_base[p] + idx;
    return -1;
}
// the blocks (I < J) of pair k of one half of the rectangle shared with shard p: this shard's half (mine) or p's
static inline __host__ __device__ void plm_half_pair(const PlmDims &d, int p, bool mine, int64_t k, int *I, int *J) {
    const int np = plm_shard_cnt(d, p);
    if (d.shard < p) {      // rows = own blocks; even rows are this shard's
        *I = d.own_lo + 2 * (int)(k / np) + (mine ? 0 : 1);
        *J = plm_shard_lo(d, p) + (int)(k % np);
    } else {                // rows = p's blocks; odd rows are this shard's
        *I = plm_shard_lo(d, p) + 2 * (int)(k / d.nblk_own) + (mine ? 1 : 0);
        *J = d.own_lo + (int)(k % d.nblk_own);
    }
}
// ... and back: the blocks (I <= J) of own pair number k
static inline __host__ __device__ void plm_pair_of_local(const PlmDims &d, int64_t k, int *I, int *J) {
    if (k < d.ntri) {                                   // the triangle over the own blocks, row major
        int r = 0;
        while (k >= d.nblk_own - r) { k -= d.nblk_own - r; r++; }
        *I = d.own_lo + r;
        *J = d.own_lo + r + (int)k;
        return;
    }
    k -= d.ntri;
    for (int p = 0; p < d.nshards; p++) {
        if (p == d.shard) continue;
        const int64_t n = plm_half_blocks(d, p);
        if (k < n) {
            plm_half_pair(d, p, true, k, I, J);
            return;
        }
        k -= n;
    }
    *I = *J = 0;      // not reached for k < np_own
}
static inline __host__ __device__ int64_t plm_pair_index(int i, int j, int L) { // i < j
    return (int64_t)i * (2 * L - i - 1) / 2 + (j - i - 1);
}

// ---- launch wrappers (plm_kernels.hip) --------------------------------------------------
// every wrapper enqueues on `st` and returns the hipError_t of the launch
hipError_t plm_launch_reweight(const PlmDims &d, const int8_t *msa_rm, int thresh, int32_t *counts,
                               hipStream_t st);
hipError_t plm_launch_onehot_rt(const PlmDims &d, const int8_t *msa_rm, const float *w, void *Rt,
                                hipStream_t st);
hipError_t plm_launch_maxabs(const PlmDims &d, const float *x, uint32_t *maxbits, int32_t *jexp,
                             hipStream_t st);
// exact: the int8 digit planes of k_fwd_x (accurate evaluation) instead of the f16 planes of k_fwd
hipError_t plm_launch_expand(const PlmDims &d, const float *x, const float *xhalo, const int32_t *jexp,
                             void *Bt, int exact, hipStream_t st);
// statistical energies of sequences under a model (k_fwd modes 1/2)
// exact (potentials only): from k_fwd_x (Bt expanded with exact = 1)
hipError_t plm_launch_forward_energy(const PlmDims &d, const int8_t *msa_rm, const void *Bt, const float *x,
                                     const int32_t *jexp, int potentials, int accurate, float *out, hipStream_t st);
hipError_t plm_launch_energy_sum(const PlmDims &d, const float *part, double *out, hipStream_t st);
// ---- variable-projection fit (fields eliminated by an inner Newton solve) ------------
// forward GEMM only: HJ[s,i,a] = sum_{j != i} J_ij(a, x_sj) in accumulator order (plm_hj_bytes)
// exact: k_fwd_x, integer arithmetic; plm_fwd_groups = state groups per workgroup of it
hipError_t plm_launch_forward_store(const PlmDims &d, const int8_t *msa_rm, const void *Bt, const int32_t *jexp,
                                    float *hj, int exact, hipStream_t st);
int plm_fwd_groups(int q, int exact);
// one pass over HJ with the fields of x: per-workgroup per-site sums for the field solver (stats 1: gradient sums
// into gpart (f64), 2: also Hessian sums -- exact diagonal, sampled off-diagonal -- into hpart (f32)) and, with write_rt, the residual fragments (Rt) and -log P partials
// (fx_part) of the solver's forward epilogue.  state / cond: the launch's role in the field solver's device-side chain
// (PlmVpState below; NULL / PLM_VP_ALWAYS: unconditional).
// exact: the passes of an accurate evaluation (exact-argument exponentials, see exp_softmax)
hipError_t plm_launch_hpass(const PlmDims &d, const float *hj, const int8_t *msa_rm, const float *w,
                            const double *h64, int write_rt, int stats, int exact, void *Rt, double *fx_part, float *hpart,
                            double *gpart, double *dpart, const int *state, int cond, hipStream_t st);
// The field solver of one evaluation is ONE chain of launches without host round trips: per chain
// position a statistics pass over the stored potentials OR (when the chain predicts that this pass is the last) a pass
// that also writes the residual planes, then the per-site Newton step, then k_vp_check.  The state lives in HBM; every
// launch of the chain looks at it first and returns at once when its role is not wanted.
#define PLM_VP_HIST 24
#define PLM_VP_MAXBLK 256   // local 16-site blocks the quiet flags cover (more: no block is ever quiet)
struct PlmVpState {
    int done;         // every site is within its share of the tolerance: the rest of the chain does nothing
    int want_rt;      // the next pass is predicted to be the last: it runs in its residual-writing role
    int final_skip;   // the pass that ended the chain wrote the residual planes: the separate last pass is not needed
    int passes;       // passes executed
    int cur;          // which of the two field buffers (h64 + cur * plm_h64_stride) holds the current fields: a step is
                      // written to the other one and becomes current only if the chain goes on (k_vp_check)
    int pad_;
    double g2_prev;   // squared gradient norm of the previous pass (contraction estimate)
    double g2_prev2;  // ... and of the pass before it (stall detection over two passes)
    double hist[PLM_VP_HIST];   // squared gradient norm and open sites (x 1e-6 in the fraction... see k_vp_check) per pass: PLM_DEBUG_VP
    int hist_loud[PLM_VP_HIST]; // 16-site blocks still active (not quiet) after the pass: PLM_DEBUG_VP
    // Quiet blocks (round 6): a 16-site block whose squared field-gradient norm is within its share of a quarter of the
    // tolerance stops moving for the rest of the chain -- its gradient is then a constant (fixed potentials, fixed
    // fields), so the statistics passes that follow skip it (k_hpass workgroups of the block return at once, k_hsolve
    // keeps its norm and copies its fields).  Passes in the residual-writing role and the final residual pass cover every
    // block.  The late passes of a chain work for a handful of slowly converging sites: PLM_DEBUG_VP traces of the
    // headline fit show 19, 19, 19, 10, 2 active blocks over the five passes of a typical evaluation.
    unsigned char quiet[PLM_VP_MAXBLK];
};
#define PLM_VP_ALWAYS 0     // unconditional launch
#define PLM_VP_PASS 1       // chain pass in its statistics role: runs while !done && !want_rt
#define PLM_VP_PASS_RT 2    // chain pass in its residual-writing role: runs while !done && want_rt
#define PLM_VP_FINAL 3      // residual pass after the chain: runs unless final_skip
hipError_t plm_launch_vp_reset(int *state, int want_rt, double *zero3, hipStream_t st);   // zero3: three scalar slots cleared with the state (or nullptr)
// Per-site gradient norms of the last pass (their sum -> g2_out[0]); update = 1: sites above their share of tol2 take a
// Newton step on the field part of x (full = 1: that pass carried Hessian sums: inverse recomputed and cached in hinv
// [sites][Q][Q]; 0: the cached inverse; 2: a chain position with Hessian sums -- fresh unless the pass ran in its
// residual-writing role).  state (device PlmVpState, may be NULL): the chain's bookkeeping (k_vp_check: done, the
// prediction for the next pass, g2_out[1] = passes, g2_out[2] = verdict): the chain ends when the squared norm over all
// sites is within tol2, or when it is below floor2 (the noise floor of the f32 gradient sums) and no longer falls by 4x
// per pass; tol2 = 0 and no state give the plain "step everywhere" behaviour.
// plm_rccl.cpp: RCCL resolved at run time
struct PlmRccl;
int plm_rccl_id(void *id128);
int plm_rccl_version();
int plm_rccl_init(const void *id128, int nranks, int rank, PlmRccl **out);
void plm_rccl_destroy(PlmRccl *p);
int plm_rccl_collective(PlmRccl *p, int op, void *send, void *recv, const int64_t *scounts, const int64_t *rcounts,
                        hipStream_t st);
const char *plm_rccl_error();
hipError_t plm_launch_h64_init(const PlmDims &d, const float *x, double *h64, hipStream_t st);
// chain = 1: a position of the chain -- returns at once when the chain is done, the step goes to the OTHER field buffer
// (committed by k_vp_check unless this pass ends the chain: the fields the pass -- and its residual planes -- saw stay
// current), x is left alone (plm_launch_fields_to_x after the chain); chain = 0: in place.
hipError_t plm_launch_hsolve(const PlmDims &d, const float *hpart, const double *gpart, int full, float *x, double *h64,
                             double lambda_h, int update, double *hinv, double *g2_site, double *g2_out, double tol2,
                             double floor2, int *state, int chain, const double *cnt, const double *dpart, hipStream_t st);
// cnt[local site][Q] = sum_s w_s [x_si = a] (f64, fixed order): the constant part of the exact first-order sums k_hsolve
// rescales the sampled Hessian with (cnt = NULL: the sampled row sums, rounds 2-4)
hipError_t plm_launch_site_counts(const PlmDims &d, const int8_t *msa_cm, const float *w, double *cnt, hipStream_t st);
size_t plm_h64_stride(const PlmDims &d);       // doubles per field buffer (h64 holds two)
// field part of x <- the chain's current fields, rounded to f32
hipError_t plm_launch_fields_to_x(const PlmDims &d, const double *h64, const int *state, float *x, hipStream_t st);
size_t plm_hj_bytes(const PlmDims &d);
size_t plm_hpart_bytes(const PlmDims &d);   // Hessian sums (f32)
size_t plm_gpart_bytes(const PlmDims &d);   // gradient sums (f64)
// run (device int, may be NULL): the kernel returns at once while *run == 0
hipError_t plm_launch_backward(const PlmDims &d, const int8_t *msa_cm, const void *Rt, int32_t *G, const int *run,
                               hipStream_t st);
// replicated multi-shard mode: planes and K ranges combined into this shard's float slab (residual units / gscale)
hipError_t plm_launch_slab_reduce(const PlmDims &d, const int32_t *G, float *slab, hipStream_t st);
// g = gscale * (G + G^T) + 2 lambda x ; mode 1: marginals (out = G / neff, no symmetrisation).  G: the int32 plane /
// K-range partials of k_bwd (ks_count = d.ksplit), or with ks_count = 0 a float slab already combined (the gathered
// slabs of the replicated multi-shard mode)
// pair_n2 / lambda_g: the group regulariser (plm_launch_pair_norms of the same x first; lambda_g = 0: none)
hipError_t plm_launch_assemble(const PlmDims &d, const void *G, int ks_count, const float *ghalo,
                               const float *x, float *g, float lambda_h, float lambda_j, double *reg_part,
                               int mode, float inv_neff, const float *pair_n2, float lambda_g, hipStream_t st);
hipError_t plm_launch_pair_norms(const PlmDims &d, const float *x, float *n2, hipStream_t st);
// sharded-state exchange staging: blocks of Q*Q*256 floats
#define PLM_BLOCK_FLOATS(d) ((size_t)(d).Q * (d).Q * 256)
hipError_t plm_launch_pack_g(const PlmDims &d, const int32_t *G, float *sendbuf, hipStream_t st);
hipError_t plm_launch_maxabs2(const float *a, int64_t na, const float *b, int64_t nb, uint32_t *maxbits,
                              int32_t *jexp, int jexp_bias, hipStream_t st);
hipError_t plm_launch_finish_fx(const PlmDims &d, const double *fx_part, int n_fx_part,
                                const double *shard_nll, int n_shard_nll, const double *reg_part,
                                int n_reg_part, double *out2 /* fx, nll */, hipStream_t st);
hipError_t plm_launch_partial_sum(const double *part, int n, double *out, hipStream_t st);
// out[k] = sum a_k[i]*b_k[i] for k < npairs (<= 4); scratch holds npairs*PLM_DOT_BLOCKS doubles
#define PLM_DOT_BLOCKS 1024
hipError_t plm_launch_dots(int npairs, const float *const *a, const float *const *b, int64_t n,
                           double *scratch, double *out, hipStream_t st);
hipError_t plm_launch_lincomb(float *out, float ca, const float *a, float cb, const float *b,
                              int64_t n, hipStream_t st);
hipError_t plm_launch_canon_to_native(const PlmDims &d, const float *xc, float *xn, hipStream_t st);
hipError_t plm_launch_native_to_canon(const PlmDims &d, const float *xn, float *xc, hipStream_t st);
// a_lo = 1: state 0 is left out of the norm (PLM_CONV_FN_NO_GAP), the gauge is still taken over all Q states
hipError_t plm_launch_fn(const PlmDims &d, const float *jij_canon, float *fn, int a_lo, int g_lo, hipStream_t st);
hipError_t plm_launch_msa_columns(const PlmDims &d, const int8_t *msa_rm, int8_t *msa_cm, int cm_rows, hipStream_t st);
hipError_t plm_launch_compact_gap_blocks(const PlmDims &d, const float *blocks, float *out, hipStream_t st);
hipError_t plm_launch_gap_normalise_pairs(const PlmDims &d, float *fij_canon, double fixed_total, hipStream_t st);
// alignment statistics (row N3): per-sequence gap counts + identities to a query, per-column gap counts
hipError_t plm_launch_align_stats(const int8_t *msa, int n, int L, int gap_state, const int8_t *query, int32_t *seq_gaps,
                                  int32_t *col_gaps, int32_t *ident, hipStream_t st);
size_t plm_bt_bytes(const PlmDims &d);
size_t plm_rt_bytes(const PlmDims &d);
size_t plm_g_bytes(const PlmDims &d);      // [nplanes][ksplit][nmf][nnfl][256] int32
size_t plm_slab_bytes(const PlmDims &d);   // [nmf][nnfl][256] floats + 256 B tail (shard nll)
int plm_reg_parts(const PlmDims &d);       // number of double partials assemble writes
bool plm_q_supported(int q);     // 2..21
int plm_q_template(int q);       // the instantiated alphabet size a problem with q symbols runs on
void plm_pick_tile(int q, int *fm, int *fn);

// ---- vector-free L-BFGS kernels (plm_kernels.hip) ----------------------------------------
#define PLM_MAX_BASIS 42   // 2*m + 2 with m <= 20 (S, Y, g in the H0 metric, g)
struct PlmVecList {
    const float *v[PLM_MAX_BASIS];
    int n;
};
// out[q * basis.n + k] = <queries.v[q], basis.v[k]>  (queries.n <= 4), f64 accumulation;
// scratch holds queries.n * basis.n * PLM_DOT_BLOCKS doubles
// dinv != nullptr: products of a query flagged in `wq` (bit q) with a basis vector flagged in `wb` (bit k) carry the
// diagonal weight dinv[i] (the H0 metric of the preconditioned L-BFGS)
hipError_t plm_launch_multidot(const PlmVecList &queries, const PlmVecList &basis, int64_t n, double *scratch,
                               double *out, const float *dinv, unsigned wq, unsigned long long wb, hipStream_t st);
// out = sum_k coef[k] * basis.v[k]   (coefficients passed by value, f32)
struct PlmCoefList {
    float c[PLM_MAX_BASIS];
};
// dinv != nullptr: the basis vectors from index `first_weighted` on are summed separately and multiplied by dinv
hipError_t plm_launch_multiaxpy(float *out, const PlmVecList &basis, const PlmCoefList &coef, int64_t n,
                                const float *dinv, int first_weighted, hipStream_t st);
// inverse Hessian diagonal of the independent-site model in the native layout; fv = [f | p(1-p)], 2*L*Q floats
hipError_t plm_launch_precond(const PlmDims &d, const float *fv, float neff, float lambda_h, float lambda_j, float *dinv,
                              hipStream_t st);
// s = x - xp ; y = g - gp in one pass
hipError_t plm_launch_sy(float *s, float *y, const float *x, const float *xp, const float *g, const float *gp,
                         int64_t n, hipStream_t st);
// The vector work behind a trial point in ONE pass (round 6): the pair s = x - xp, y = g - gp is formed in registers, written
// to its ring slot and used at once as the queries (s, y, g) of the Gram pass over `basis` -- the entries of `basis` that
// ARE s_new / y_new / g are taken from the registers, the others read once -- together with g.dir, x.x and x.x over the first
// nh entries (the fields).  Same accumulation per product as plm_launch_sy + plm_launch_multidot + three plm_launch_dots
// (same grid, same reduction tree: the same bits), 33 instead of 41 vector transfers per iteration.
//   out_md[q * basis.n + k] as plm_launch_multidot (queries s_new, y_new, g), out_md[3 * basis.n] = s_new.s_new over the
//   fields; out_ex[0..2] = g.dir, x.x, x.x (fields)
hipError_t plm_launch_sy_multidot(float *s_new, float *y_new, const float *x, const float *xp, const float *g, const float *gp,
                                  const float *dir, const PlmVecList &basis, int64_t n, int64_t nh, double *scratch,
                                  double *out_md, double *out_ex, const float *dinv, unsigned wq, unsigned long long wb,
                                  hipStream_t st);
// plm_launch_multiaxpy that also writes the first trial point of the line search, trial = xacc + stp * out
hipError_t plm_launch_multiaxpy_trial(float *out, const PlmVecList &basis, const PlmCoefList &coef, int64_t n,
                                      const float *dinv, int first_weighted, const float *xacc, float stp, float *trial,
                                      hipStream_t st);
