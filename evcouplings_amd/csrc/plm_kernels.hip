// plm_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the pseudo-likelihood Potts solver.
//
// Hot path, replacing the arithmetic of the external plmc
// process that evcouplings/couplings/tools.py:266 launches:
//   k_reweight   N x N Hamming identity counts on the packed int8 alignment (VALU, integer)
//   k_expand     parameters -> forward B operand (f16 hi/lo MFMA fragments)
//   k_fwd        one-hot(MSA) x J on the 2:4 sparse MFMA (K ordered (site, state): 4 states of one site per group
//                of 4 slots, state 0 as reference state) -> the coupling part of every conditional (HJ), stored for
//                k_hpass; two more epilogues turn the same GEMM into statistical energies / potentials of sequences
//                under a fitted model (row N2); an ACCURATE instantiation (state groups per workgroup, f64 outer sums)
//                serves the last evaluations of a fit and plm_eval.
//   k_bwd        one-hot(MSA)^T x residuals on MFMA -> asymmetric gradient slab
//   k_assemble   slab + slab^T + L2 term -> gradient, regulariser partial sums
//   k_hpass      per-site softmax over HJ + fields -> residuals (the backward operand), -log P, and the gradient /
//                Hessian sums of the field subproblems; k_hsolve: per-site Newton step (variable-projection fit,
//                DESIGN.md 2c, 4.8).  (Rounds 1-2 also had the softmax fused into k_fwd: 71 spilled registers; the
//                split costs the same time, spills nothing and serves both optimisers.)
//   k_align_rows / k_align_cols   gap counts and identities of the align stage (row N3)
// plus small streaming kernels for L-BFGS (dots / linear combinations) and scoring.  Mean-field DCA
// (covariance inverse, fields, direct information) lives in plm_meanfield.hip.
// Compile-time experiment switches (neutral in the product build; DESIGN.md 4.3 has the measurements):
// PLM_DMA_STAGGER_*, PLM_ASYNC_A, PLM_PROBE.  (The dense forward formulation, the three-buffer pipeline and the ablation
// masks of rounds 1-3 are gone: git history and profiles/r0[1-3]_* hold what they measured.)
//
// The alignment is int8 in HBM; one-hot MFMA A fragments are expanded from 8 packed bytes
// in registers (never materialised in memory); the dense operand (couplings / residuals)
// is split into two f16 planes (22-bit significand, power-of-two pre-scaled) and
// accumulated in f32 by v_mfma_f32_16x16x32_f16 (backward) / v_smfmac_f32_16x16x64_f16 (forward).  Written for
// wave64 / gfx950 only.
#include "../../include/plm_hip.h"
#include "plm_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <utility>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

// PLM_DMA_STAGGER_FWD / _BWD (sixteenths of a K step): waves 0-3 issue their LDS-DMA pieces at the start of
// the step, waves 4-7 (the SIMD partners of 0-3) this far into it.  Issuing a piece blocks a wave for
// ~100-200 cycles; when both waves of a SIMD do that at the same time the MFMA pipe idles, but pieces issued
// late land late.  Measured on MI355X (ms, stagger 0/4/8/12): k_fwd 5.42/5.27/5.59/5.69, k_bwd 5.19/5.46/5.50/5.46.
#ifndef PLM_DMA_STAGGER_FWD
#define PLM_DMA_STAGGER_FWD 4
#endif
#ifndef PLM_DMA_STAGGER_BWD
#define PLM_DMA_STAGGER_BWD 0
#endif
typedef unsigned long long u64;
// PLM_ASYNC_A: the alignment bytes of the next K step are fetched by a load the compiler does not track.
// hipcc cannot count the LDS-DMA pieces issued under branches after an ordinary load, so it waits
// vmcnt(0) where the loaded value is first used and schedules around that wait; the untracked load lands
// under the kernel's own vmcnt(0) before the next barrier instead.  The destination is a loop-carried
// variable updated IN PLACE ("+v": input and output share the register), so no copy of a not-yet-landed
// value can be emitted; the consumer copies it out only after the wait (vm_landed).
#ifndef PLM_ASYNC_A
#define PLM_ASYNC_A 1
#endif
// address = wave-uniform base (SGPR pair) + 32-bit per-lane byte offset: one VGPR instead of a 64-bit pointer
__device__ __forceinline__ void load_b64_inplace(u64 &v, const void *base, u32 off) {
#if PLM_ASYNC_A
    asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base) : "memory");
#else
    v = *(const u64 *)((const char *)base + off);
#endif
}
__device__ __forceinline__ void vm_landed(u64 &v) { asm volatile("" : "+v"(v)); }
// hipcc does NOT drain the LDS-DMA queue at a barrier (only lgkmcnt): every wave waits for its own
// global_load_lds pieces explicitly before the barrier that publishes a tile (without it a late piece
// is read before it lands -- seen as run-to-run noise at N = 50k)
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// bare s_barrier: unlike __syncthreads() it does not wait for this wave's LDS reads in flight
__device__ __forceinline__ void barrier_raw() { asm volatile("s_barrier" ::: "memory"); }
// PLM_PROBE=1 (debug build only): per-wave cycle totals summed into plm_probe_acc[kernel][phase]:
// 0 wait (vmcnt + barrier), 3 whole kernel, 4 epilogue, 5 waves
#ifndef PLM_PROBE
#define PLM_PROBE 0
#endif
#if PLM_PROBE
__device__ unsigned long long plm_probe_acc[2][8];
#define PROBE_NOW() __builtin_readcyclecounter()
extern "C" __attribute__((visibility("default"))) void plm_probe_read(unsigned long long *out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(plm_probe_acc), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(plm_probe_acc), z, sizeof z);
    }
}
#endif
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
// The stored potentials (1.28 GB per evaluation at the headline) are written once and read back from HBM by the field
// solver: a streaming store keeps them from evicting the forward operand, which every workgroup of a site block shares
// through L2 (round 6; -DPLM_HJ_NT=0 for the A/B build)
#ifndef PLM_HJ_NT
#define PLM_HJ_NT 1
#endif
__device__ __forceinline__ void store_hj(float4 *p, float x, float y, float z, float w) {
#if PLM_HJ_NT
    __builtin_nontemporal_store((f32x4){x, y, z, w}, (f32x4 *)p);
#else
    *p = make_float4(x, y, z, w);
#endif
}
#define LDS_FPTR(p) ((__attribute__((address_space(3))) float *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// half index e of an 8-wide MFMA k-chunk <-> byte PERM8[e] of the 8 packed alignment bytes
// (pairing (0,2),(1,3) lets one multiply turn two match bits into two f16 1.0 values)
__host__ __device__ __forceinline__ constexpr int perm8(int e) { return (e & 4) | ((e & 1) << 1) | ((e & 2) >> 1); }

// 8 packed states (two dwords) vs state b -> 8 f16 values in {0,1}.  States are < 128, so
// (x ^ b) + 0x7f sets bit 7 of a byte exactly when it differs from b, with no carries.
__device__ __forceinline__ half8 onehot8(u32 lo, u32 hi, u32 bb) {
    const u32 y0 = (lo ^ bb) + 0x7f7f7f7fu;
    const u32 y1 = (hi ^ bb) + 0x7f7f7f7fu;
    union {
        u32 w[4];
        half8 h;
    } r;
    r.w[0] = 0x3C003C00u - ((y0 >> 7) & 0x00010001u) * 0x3C00u;   // bytes 0,2
    r.w[1] = 0x3C003C00u - ((y0 >> 15) & 0x00010001u) * 0x3C00u;  // bytes 1,3
    r.w[2] = 0x3C003C00u - ((y1 >> 7) & 0x00010001u) * 0x3C00u;   // bytes 4,6
    r.w[3] = 0x3C003C00u - ((y1 >> 15) & 0x00010001u) * 0x3C00u;  // bytes 5,7
    return r.h;
}

// ---- hand-scheduled LDS reads ---------------------------------------------------------------
// hipcc sinks every ds_read next to its first use and waits lgkmcnt(0) there (LDS latency exposed
// once per fragment).  These helpers issue the read early as inline asm and tie the counted wait
// to the destination registers by data dependence, so nothing that uses them can be hoisted
// above the wait (cdna_hip_programming.md section 5.4 rule 18).  LDS returns in order, so
// "N newer reads may still be in flight" is s_waitcnt lgkmcnt(N).
__device__ __forceinline__ u32 lds_addr(const void *p) {
    return (u32)(size_t)(__attribute__((address_space(3))) const void *)p;
}
template <int OFF> __device__ __forceinline__ half8 lds_read_b128(u32 addr) {
    half8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void lds_wait(half8 &a, half8 &b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// the same for the int8 fragments of the backward GEMM (one 16-byte fragment slot per read)
template <int OFF> __device__ __forceinline__ i32x4 lds_read_b128_i(u32 addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void lds_wait_i(i32x4 &a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

// ---- residuals in fixed point, signed base-256 digits (backward GEMM on the int8 matrix cores) -------------------
// Three digits: R in [-8355711, 8355711], R + 0x808080 lies in [0, 2^24) and its bytes are the digits + 128
// (R = sum_k (u_k - 128) 256^k  <=>  sum_k u_k 256^k = R + 128 (1 + 256 + 65536)); xor 0x80 per byte turns u_k into the
// two's-complement digit d_k = u_k - 128.  Four digits: the same with 0x80808080 in 32-bit wrap-around arithmetic
// (|R| <= 2 139 062 143).  Returns the digits in bytes 0-2 (0-3).
__device__ __forceinline__ u32 digits_of(float v, bool four) {
    const u32 bias = four ? 0x80808080u : 0x00808080u;
    const int R = (int)__builtin_rintf(v);       // ties to even: unbiased (|v| <= PLM_R_QMAX by construction)
    return ((u32)R + bias) ^ bias;
}
// bytes p of four digit words -> the plane words [x0.p, x1.p, x2.p, x3.p] (v_perm_b32: selector byte 0-3 picks a byte
// of the SECOND source, 4-7 of the first)
__device__ __forceinline__ void planes_of4(u32 x0, u32 x1, u32 x2, u32 x3, u32 (&pl)[PLM_BWD_MAXPLANES]) {
    const u32 t01 = __builtin_amdgcn_perm(x1, x0, 0x05010400u);   // x0.b0 x1.b0 x0.b1 x1.b1
    const u32 t23 = __builtin_amdgcn_perm(x3, x2, 0x05010400u);
    const u32 u01 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);   // x0.b2 x1.b2 x0.b3 x1.b3
    const u32 u23 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
    pl[0] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
    pl[1] = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
    pl[2] = __builtin_amdgcn_perm(u23, u01, 0x05040100u);
    pl[3] = __builtin_amdgcn_perm(u23, u01, 0x07060302u);
}
// byte address of the 16-byte fragment slot of (plane, 64-sequence group s64, column fragment nf, lane slot) in Rt
__device__ __forceinline__ size_t rt_slot(const PlmDims &d, int plane, int s64, int nf, int slot) {
    return ((((size_t)plane * d.nst128 + (s64 >> 1)) * d.nnfl + nf) * 2 + (s64 & 1)) * 1024 + (size_t)slot * 16;
}

// ---- LDS-DMA staging of the next tile -----------------------------------------------------------
// A tile is copied global -> LDS in 1 KB pieces (one global_load_lds per wave-instruction); wave w copies
// pieces w, w+8, ...  (the wave number is passed through readfirstlane so the piece tests are scalar
// branches).  Measured alternatives on MI355X, all slower: spreading a wave's pieces over the K step
// (k_bwd +35 %: late pieces stall the vmcnt(0) before the next barrier), leaving the copy to 4 of the 8
// waves (k_fwd 5.3 -> 6.2 ms, k_bwd 5.8 -> 9.7 ms), register staging instead of LDS-DMA.
struct DmaPlan {
    const char *src;   // global address of piece 0 (wave-uniform: lives in SGPRs)
    char *dst;         // LDS base of the target buffer (wave-uniform)
    int first;         // this wave's first piece (wave-uniform)
    int limit;         // number of pieces to copy (0 = nothing to stage)
    u32 lane_off;      // lane * 16: the only per-lane part of the address
    bool late;         // this wave issues in the second slot of the step (PLM_DMA_STAGGER_*)
    // k_bwd: one more piece per wave, the alignment bytes of its row group for one half of the next K step (gathered:
    // every lane has its own global address, the LDS side is lane-linear like every piece)
    const char *a_src = nullptr;   // wave-uniform base (nullptr: no such piece)
    u32 a_off = 0;                 // per-lane byte offset
    char *a_dst = nullptr;         // LDS destination (wave-uniform)
    // k_fwd with state groups: a workgroup copies the fragments of ITS states only -- two runs of a tile (slots 0-7 and
    // slots 8-15 of every state's fragment).  Pieces from `split` on lie `gap` bytes further in the source.
    int split = 1 << 30;
    int gap = 0;
};
// NW = waves of the workgroup: wave w copies pieces w, w + NW, ...
template <int NP, int NW = 8> __device__ __forceinline__ void dma_issue_all(const DmaPlan &P) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const int p = P.first + NW * k;
        if (p < P.limit)
            __builtin_amdgcn_global_load_lds(GLB_PTR(P.src + p * 1024 + (p >= P.split ? P.gap : 0) + P.lane_off),
                                             LDS_PTR(P.dst + p * 1024), 16, 0, 0);
    }
    if (P.a_src) __builtin_amdgcn_global_load_lds(GLB_PTR(P.a_src + P.a_off), LDS_PTR(P.a_dst), 16, 0, 0);
}
// the two issue slots of a step with NF fragments: fragment 0 for the early waves, fragment NF * STG / 16 for the late ones
template <int NF, int A, int NP, int STG, int NW = 8> __device__ __forceinline__ void dma_at(const DmaPlan &P) {
    constexpr int D = (NF * STG) / 16;
    constexpr int S1 = (D < NF) ? D : NF - 1;
    if constexpr (S1 == 0) {
        if constexpr (A == 0) dma_issue_all<NP, NW>(P);
    } else {
        if constexpr (A == 0) { if (!P.late) dma_issue_all<NP, NW>(P); }
        if constexpr (A == S1) { if (P.late) dma_issue_all<NP, NW>(P); }
    }
}

// the value lane `src` (compile-time constant after unrolling) holds, in every lane: two v_readlane_b32
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double block_reduce_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += sh[k];
    return t;  // valid in thread 0
}

#define PLM_MAX_DEVICES 64
static int plm_current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (dev >= 0 && dev < PLM_MAX_DEVICES) ? dev : 0;
}
// Any alphabet of 2..32 symbols runs on the next instantiated size (4, 5, 20, 21, 32): the surplus states are dead
// padding of the native layout (never observed, masked out of every softmax, parameters structurally zero), the
// canonical arrays at the API keep the problem's own size (PlmDims::Qc).  The 32-state instantiation (alphabets of
// 22..32 symbols: proteins with ambiguity codes, extended nucleotide alphabets) runs the forward GEMM with two state
// groups per workgroup (2 x 32 accumulator fragments would not fit a wave) and has no field solver: such problems take
// the joint L-BFGS path (plm_host.cpp vp_enabled).
bool plm_q_supported(int q) { return q >= 2 && q <= 32; }
int plm_q_template(int q) { return q <= 4 ? 4 : q == 5 ? 5 : q <= 20 ? 20 : q == 21 ? 21 : 32; }
void plm_pick_tile(int q, int *fm, int *fn) {
    if (q == 32) { *fm = 8; *fn = 5; }
    else if (q == 21) { *fm = 7; *fn = 7; }
    else if (q == 20) { *fm = 5; *fn = 5; }
    else if (q == 5) { *fm = 5; *fn = 5; }
    else { *fm = 4; *fn = 4; }
}
// forward operand: the K-step tiles of every local column block, then (sparse formulation) the reference-state
// constants C[local site][Q] as floats
size_t plm_bt_bytes(const PlmDims &d) {
    return (size_t)d.blk_per_shard * d.nksteps * 2 * d.Q * 1024 +
           (size_t)d.blk_per_shard * 16 * d.Q * (sizeof(float) + sizeof(double));
}
// the reference-state constants behind the tiles: C[local site][Q] as floats (energy / potential epilogues of k_fwd),
// then the same in f64 (the fit: k_hpass adds them together with the fields, as a hi + lo pair)
static inline const float *bt_cref32(const PlmDims &d, const void *Bt) {
    return (const float *)((const char *)Bt + (size_t)d.blk_per_shard * d.nksteps * 2 * d.Q * 1024);
}
static inline const double *bt_cref64(const PlmDims &d, const void *Bt) {
    return (const double *)(bt_cref32(d, Bt) + (size_t)d.blk_per_shard * 16 * d.Q);
}
// (+ slack: k_bwd_w copies whole tiles of PLM_BWDW_COLS column fragments, also where the last tile of a row is narrower)
size_t plm_rt_bytes(const PlmDims &d) {
    return (size_t)d.nplanes * d.nst128 * d.nnfl * 2 * 1024 + (size_t)PLM_BWDW_COLS * 2 * 1024;
}
size_t plm_g_bytes(const PlmDims &d) { return (size_t)d.nplanes * d.ksplit * d.nmf * d.nnfl * 1024; }
size_t plm_slab_bytes(const PlmDims &d) { return (size_t)d.nmf * d.nnfl * 1024 + 256; }
int plm_reg_parts(const PlmDims &d) { return (int)(d.np_own * d.Q) + (int)((d.nh_pad_l + 255) / 256); }

// =========================================================================================
// K1  sequence reweighting (row a4; twin: align/alignment.py:1193-1233)
//   counts[s] += #{ t in this block's range : ident(s,t) >= thresh }
// One lane per sequence s; the other sequence t is wave-uniform and read through the scalar
// cache.  4 alignment bytes per VALU triple: xor+add, and, popcount-accumulate.
// =========================================================================================
#define RW_TT 32   // t-rows per register tile
#define RW_CW 16   // dwords (64 sites) per column chunk
// zero bytes (gaps) of a packed word -> 0x7c, a value no alignment byte takes (states < 32, pad 127)
// 0x80 in every zero byte of v, exactly (all bytes < 0x80: (b & 0x7f) + 0x7f carries into bit 7 iff b != 0, and never
// into the next byte).  The shorter (v - 0x01010101) & ~v & 0x80808080 is NOT exact per byte: the borrow of a zero byte
// also flags a byte of value 1 above it -- a residue of state 1 behind a gap would be taken for a gap.
__device__ __forceinline__ u32 zero_bytes(u32 v) { return ~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u; }
__device__ __forceinline__ u32 gaps_to_sentinel(u32 v) {
    return v + (zero_bytes(v) >> 7) * 0x7cu;
}
// UNGAPPED (PLM_CONV_G_UNGAPPED_LENGTH, gap mode only): the threshold of a pair applies to the n_both positions where
// both sequences are ungapped, ident >= ceil(theta * n_both - 1e-9); npad = padded columns (they look like matches)
template <bool UNGAPPED>
__global__ __launch_bounds__(256) void k_reweight(const u32 *__restrict__ msa32, int Lw, int N,
                                                 int thresh_padded, int t_per_block,
                                                 int32_t *__restrict__ counts, int gap_mode, double theta, int npad) {
    const int s = blockIdx.x * 256 + threadIdx.x;  // < Np always (rows exist, padded)
    const int tb0 = blockIdx.y * t_per_block;
    const int tb1 = min(N, tb0 + t_per_block);
    const u32 *__restrict__ myrow = msa32 + (size_t)s * Lw;
    int cnt = 0;
    for (int t0 = tb0; t0 < tb1; t0 += RW_TT) {
        int ident[RW_TT], both[UNGAPPED ? RW_TT : 1];
#pragma unroll
        for (int k = 0; k < RW_TT; k++) ident[k] = 0;
        if constexpr (UNGAPPED) {
#pragma unroll
            for (int k = 0; k < RW_TT; k++) both[k] = 0;
        }
        for (int c0 = 0; c0 < Lw; c0 += RW_CW) {   // Lw is a multiple of 8
            u32 mine[RW_CW], mgap[UNGAPPED ? RW_CW : 1];
            const int cw = min(RW_CW, Lw - c0);
#pragma unroll
            for (int k = 0; k < RW_CW; k++) {
                mine[k] = k < cw ? myrow[c0 + k] : 0x7e7e7e7eu;
                if constexpr (UNGAPPED) mgap[k] = zero_bytes(mine[k]);
                if (gap_mode) mine[k] = gaps_to_sentinel(mine[k]);
            }
#pragma unroll
            for (int tt = 0; tt < RW_TT; tt++) {
                // rows t >= N are padding rows (all 127): they exist in memory, never reach thresh
                const u32 *__restrict__ trow = msa32 + (size_t)(t0 + tt) * Lw + c0;
                int acc = ident[tt];
#pragma unroll
                for (int k = 0; k < RW_CW; k++) {
                    const u32 other = k < cw ? trow[k] : 0x7d7d7d7du;
                    const u32 y = (mine[k] ^ other) + 0x7f7f7f7fu;      // bit7 set <=> mismatch
                    acc += 4 - __builtin_popcount(y & 0x80808080u);
                    if constexpr (UNGAPPED)
                        if (k < cw) both[tt] += 4 - __builtin_popcount(mgap[k] | zero_bytes(other));
                }
                ident[tt] = acc;
            }
            // early exit as in k_reweight_reg: when no (lane, partner) pair of the wave can still reach the threshold
            // with every remaining site a match, the rest of the columns changes no count
            if constexpr (!UNGAPPED) {
                const int rest = 4 * max(0, Lw - (c0 + RW_CW));
                bool alive = false;
#pragma unroll
                for (int tt = 0; tt < RW_TT; tt++) alive |= ident[tt] + rest >= thresh_padded;
                if (rest > 0 && !__any(alive)) break;
            }
        }
#pragma unroll
        for (int tt = 0; tt < RW_TT; tt++) {
            bool hit;
            if constexpr (UNGAPPED) {
                const int nb = both[tt] - npad, id = ident[tt] - npad;
                hit = id >= (int)ceil(theta * (double)nb - 1e-9);
            } else {
                hit = ident[tt] >= thresh_padded;
            }
            cnt += (t0 + tt < tb1 && (hit || (gap_mode && t0 + tt == s))) ? 1 : 0;
        }
    }
    if (s < N && cnt) atomicAdd(&counts[s], cnt);
}

// Register-resident variant (the one normally used): the lane's whole row (LW dwords) stays in
// VGPRs, partner rows t stream through the scalar cache one at a time (wave-uniform address ->
// s_load_dwordx16), 3 VALU per 4 sites: v_xad (xor+add), v_and, v_bcnt (popcount-accumulate).
// LW is the row length in dwords rounded up to 8; rows longer than 192 dwords (L > 768) fall
// back to the column-chunked kernel above.
template <int LW>
__global__ __launch_bounds__(256) void k_reweight_reg(const u32 *__restrict__ msa32, int Lw, int N,
                                                     int thresh_padded, int t_per_block,
                                                     int32_t *__restrict__ counts, int gap_mode) {
    // Symmetric: only pairs t > s are compared; a hit is credited to s (per-lane counter) and to t
    // (wave ballot -> one LDS atomic per wave and t, flushed to HBM once per block).
    extern __shared__ int tcount[];
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int tb0 = blockIdx.y * t_per_block;
    const int tb1 = min(N, tb0 + t_per_block);
    if (tb1 <= (int)blockIdx.x * 256 + 1) return;          // whole T range at or below this S tile
    for (int k = threadIdx.x; k < tb1 - tb0; k += 256) tcount[k] = 0;
    __syncthreads();
    const u32 *__restrict__ myrow = msa32 + (size_t)s * Lw;
    u32 mine[LW];
#pragma unroll
    for (int k = 0; k < LW; k += 4) {
        if (k < Lw) {   // Lw is a multiple of 8: whole uint4 loads
            const uint4 v = *(const uint4 *)(myrow + k);
            mine[k] = v.x; mine[k + 1] = v.y; mine[k + 2] = v.z; mine[k + 3] = v.w;
            if (gap_mode) {   // gap-gap is not an identity: my gaps become a value no partner byte has
#pragma unroll
                for (int e = 0; e < 4; e++) mine[k + e] = gaps_to_sentinel(mine[k + e]);
            }
        } else {
            mine[k] = mine[k + 1] = mine[k + 2] = mine[k + 3] = 0x7e7e7e7eu;   // never matches the partner's pad
        }
    }
    // mismatches are counted (popcount of the per-byte "differs" flags); ident = 4*Lw - mism.
    // The partner row is consumed in 16-dword chunks (one s_load_dwordx16 each, the next chunk in
    // flight while this one is compared); words past Lw in the last chunk meet 0x7e in `mine`
    // and always count as 4 mismatches, which the threshold absorbs.
    struct Chunk { u32 v[16]; };
    u32 c7f = 0x7f7f7f7fu;
    asm volatile("" : "+v"(c7f));   // keep the constant in a VGPR (VOP3 takes one SGPR, no literal)
    const int nch = (Lw + 15) / 16;
    const int max_mism = 4 * Lw - thresh_padded + 4 * (16 * nch - Lw);
    int cnt = 0;   // the sequence itself is pre-counted: the launcher fills counts[] with 1
    const int t_first = max(tb0, (int)blockIdx.x * 256 + 1);
    // Early exit (round 5): mismatch counts only grow along a row, so once EVERY lane of the wave is past max_mism the
    // partner cannot be a neighbour of any of its 64 sequences and the rest of the row is skipped -- the same counts,
    // fewer compares.  Unrelated sequences (identity 0.2-0.3) pass 1 - theta = 0.2 L mismatches after a quarter to a
    // half of the row: 2.1 of 5 chunks per (wave, partner) at the headline.  The first chunk of the NEXT partner row is
    // requested before this row's compares start (rows are short now: its latency would be exposed once per row).
    Chunk head = *(const Chunk *)(msa32 + (size_t)t_first * Lw);
    for (int t = t_first; t < tb1; ++t) {
        const Chunk *__restrict__ trow = (const Chunk *)(msa32 + (size_t)t * Lw);
        int mism = 0;
        Chunk cur = head;
        head = *(const Chunk *)(msa32 + (size_t)(t + 1) * Lw);      // (rows up to Np + 32 exist: t + 1 <= N is readable)
        bool far = false;
#pragma unroll
        for (int c = 0; c < LW / 16; c++) {
            if (c < nch && !far) {
                Chunk nxt = cur;
                if (c + 1 < nch) nxt = trow[c + 1];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    if (!far) {
#pragma unroll
                        for (int k = 8 * hf; k < 8 * hf + 8; k++) {
                            u32 y;   // (mine ^ partner) + 0x7f7f7f7f in one VALU op: bit 7 of a byte set <=> sites differ
                            asm("v_xad_u32 %0, %1, %2, %3" : "=v"(y) : "v"(mine[16 * c + k]), "s"(cur.v[k]), "v"(c7f));
                            mism += __builtin_popcount(y & 0x80808080u);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        far = __all(mism > max_mism) != 0;     // wave-uniform; looked at every 32 sites
                    }
                }
                cur = nxt;
            }
        }
        if (far) continue;
        const bool hit = t > s && s < N && mism <= max_mism;
        cnt += hit ? 1 : 0;
        const unsigned long long m = __ballot(hit);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&tcount[t - tb0], __popcll(m));
    }
    if (s < N && cnt) atomicAdd(&counts[s], cnt);
    __syncthreads();
    for (int k = threadIdx.x; k < tb1 - tb0; k += 256)
        if (tcount[k]) atomicAdd(&counts[tb0 + k], tcount[k]);
}

hipError_t plm_launch_reweight(const PlmDims &d, const int8_t *msa_rm, int thresh, int32_t *counts,
                               hipStream_t st) {
    const int Lw = d.Lp32 / 4;
    // -g conventions (include/plm_hip.h): gap-gap identities, threshold on the jointly ungapped length
    const bool ungapped = d.gap_mode && (d.conv & PLM_CONV_G_UNGAPPED_LENGTH);    // implies: gaps are no identities
    const int gap_mode = (d.gap_mode && (ungapped || !(d.conv & PLM_CONV_G_GAPS_IDENTICAL))) ? 1 : 0;
    const bool fast = Lw <= 192 && !ungapped;
    // symmetric kernel: every sequence starts with itself counted; fallback kernel: self is a match
    hipError_t e = fast ? hipMemsetD32Async((hipDeviceptr_t)counts, 1, (size_t)d.Np, st)
                        : hipMemsetAsync(counts, 0, sizeof(int32_t) * d.Np, st);
    if (e != hipSuccess) return e;
    // padded columns (value 127 in every row) always match: shift the threshold instead
    const int thr = thresh + (d.Lp32 - d.L);
    if (fast) {
        int tsplit = std::max(1, (8192 + d.nstiles - 1) / d.nstiles);   // ~half the blocks exit at once
        tsplit = std::max(tsplit, (d.N + 8191) / 8192);                    // LDS column counters <= 32 KB
        int tper = (d.N + tsplit - 1) / tsplit;
        tsplit = (d.N + tper - 1) / tper;
        const dim3 grid(d.nstiles, tsplit), block(256);
        const size_t lds = sizeof(int) * (size_t)tper;
        const u32 *m32 = (const u32 *)msa_rm;
        if (Lw <= 32) hipLaunchKernelGGL(k_reweight_reg<32>, grid, block, lds, st, m32, Lw, d.N, thr, tper, counts, gap_mode);
        else if (Lw <= 64) hipLaunchKernelGGL(k_reweight_reg<64>, grid, block, lds, st, m32, Lw, d.N, thr, tper, counts, gap_mode);
        else if (Lw <= 96) hipLaunchKernelGGL(k_reweight_reg<96>, grid, block, lds, st, m32, Lw, d.N, thr, tper, counts, gap_mode);
        else if (Lw <= 128) hipLaunchKernelGGL(k_reweight_reg<128>, grid, block, lds, st, m32, Lw, d.N, thr, tper, counts, gap_mode);
        else hipLaunchKernelGGL(k_reweight_reg<192>, grid, block, lds, st, m32, Lw, d.N, thr, tper, counts, gap_mode);
        return hipGetLastError();
    }
    int tsplit = (2048 + d.nstiles - 1) / d.nstiles;
    int tper = (d.N + tsplit - 1) / tsplit;
    tper = ((tper + RW_TT - 1) / RW_TT) * RW_TT;
    tsplit = (d.N + tper - 1) / tper;
    // the padded rows t in [N, Np) are readable; rows beyond Np are not: cap the tile walk
    // (tb1 <= N and t0+tt < t0+RW_TT <= Np + RW_TT) -> msa_rm is allocated with RW_TT spare rows
    if (ungapped)   // the threshold is per pair; a sequence with itself always passes (ident = n_both)
        hipLaunchKernelGGL(k_reweight<true>, dim3(d.nstiles, tsplit), dim3(256), 0, st, (const u32 *)msa_rm, Lw, d.N,
                           thr, tper, counts, gap_mode, d.theta, d.Lp32 - d.L);
    else
        hipLaunchKernelGGL(k_reweight<false>, dim3(d.nstiles, tsplit), dim3(256), 0, st, (const u32 *)msa_rm, Lw,
                           d.N, thr, tper, counts, gap_mode, d.theta, d.Lp32 - d.L);
    return hipGetLastError();
}

// =========================================================================================
// column-major image of the alignment for the backward GEMM's A operand: cm[i][s] = rm[s][i] for the sites, pad 127 in
// the rows past Lp32 and the sequences past N, and the "ones" row nb16 * 16 = state 0 for every real sequence (its
// one-hot row against state 0 is all ones: the field gradient / the single-site counts come out of the same GEMM).
// 64 x 64 byte tiles through LDS, both sides coalesced.
// =========================================================================================
__global__ __launch_bounds__(256) void k_msa_columns(PlmDims d, const int8_t *__restrict__ rm, int8_t *__restrict__ cm,
                                                    int cm_rows) {
    __shared__ int8_t tile[64][65];
    const int s0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ones_row = d.nb16 * 16;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int s = s0 + ty * 16 + k, i = i0 + tx;
        int8_t v = (int8_t)PLM_PAD_STATE;
        if (s < d.Np && i < d.Lp32) v = rm[(size_t)s * d.Lp32 + i];       // rows N..Np-1 and columns L..Lp32-1 are pad
        if (i == ones_row) v = s < d.N ? (int8_t)0 : (int8_t)PLM_PAD_STATE;
        tile[ty * 16 + k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int i = i0 + ty * 16 + k, s = s0 + tx;
        if (i < cm_rows && s < d.Np) cm[(size_t)i * d.Np + s] = tile[tx][ty * 16 + k];
    }
}
hipError_t plm_launch_msa_columns(const PlmDims &d, const int8_t *msa_rm, int8_t *msa_cm, int cm_rows, hipStream_t st) {
    hipLaunchKernelGGL(k_msa_columns, dim3((d.Np + 63) / 64, (cm_rows + 63) / 64), dim3(256), 0, st, d, msa_rm, msa_cm, cm_rows);
    return hipGetLastError();
}

// =========================================================================================
// one-hot residual builder for the marginals:  R[s,(i,a)] = w_s [x_si = a]  in Rt layout
// =========================================================================================
__global__ __launch_bounds__(256) void k_onehot_rt(PlmDims d, const int8_t *__restrict__ msa_rm,
                                                  const float *__restrict__ w, char *__restrict__ Rt) {
    // one workgroup per (64-sequence group, site block): for every state a fragment of 64 lanes x 16 sequences
    const int s64 = blockIdx.x, b16l = blockIdx.y, b16 = d.b16_lo + b16l;
    for (int idx = threadIdx.x; idx < d.Q * 64; idx += 256) {
        const int a = idx >> 6, lane = idx & 63, kg = lane >> 4, ii = lane & 15;
        const int i = b16 * 16 + ii;
        u32 pw[PLM_BWD_MAXPLANES][4];
        const bool four = d.nplanes == 4;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            u32 x[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int s = s64 * 64 + kg * 16 + q4 * 4 + e;
                float v = 0.f;
                if (i < d.L && b16 < d.nb16 && msa_rm[(size_t)s * d.Lp32 + i] == a) v = w[s] * d.rscale;
                x[e] = digits_of(v, four);
            }
            u32 pl[PLM_BWD_MAXPLANES];
            planes_of4(x[0], x[1], x[2], x[3], pl);
#pragma unroll
            for (int p = 0; p < PLM_BWD_MAXPLANES; p++) pw[p][q4] = pl[p];
        }
#pragma unroll
        for (int p = 0; p < PLM_BWD_MAXPLANES; p++)
            if (p < d.nplanes)
                *(uint4 *)(Rt + rt_slot(d, p, s64, b16l * d.Q + a, lane)) = make_uint4(pw[p][0], pw[p][1], pw[p][2], pw[p][3]);
    }
}
hipError_t plm_launch_onehot_rt(const PlmDims &d, const int8_t *msa_rm, const float *w, void *Rt,
                                hipStream_t st) {
    if (d.b16_hi <= d.b16_lo) return hipSuccess;   // a trailing shard may own no site block
    hipLaunchKernelGGL(k_onehot_rt, dim3(2 * d.nst128, d.b16_hi - d.b16_lo), dim3(256), 0, st, d, msa_rm, w, (char *)Rt);
    return hipGetLastError();
}

// =========================================================================================
// scale of the coupling operand: jexp = 14 - exponent(max |J|)  (so |J| 2^jexp < 2^14)
// =========================================================================================
__global__ __launch_bounds__(256) void k_maxabs(const float *__restrict__ x, int64_t n, u32 *maxbits) {
    u32 m = 0;
    // x is 16-byte aligned and n a multiple of 4 in every caller (block-padded parameter vectors)
    const uint4 *x4 = (const uint4 *)x;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
        const uint4 v = x4[i];
        m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    }
    for (int64_t i = (n / 4) * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (u32)__shfl_down((int)m, o, 64));
    // ONE atomic per workgroup (round 6: one per wave of 1024 workgroups were 4096 atomics on one address -- 50 of the
    // kernel's 57 us at the headline, and as many on a shard with an eighth of the vector)
    __shared__ u32 wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (t) atomicMax(maxbits, t);
    }
}
// bias (PlmOptions::jexp_bias, measurement knob): added to the scale exponent of the coupling operand.  A different
// power-of-two pre-scale moves every hi/lo split point and every f32 rounding of the forward GEMM without
// changing the mathematics: the difference of two evaluations that differ only in it measures the rounding
// noise of the forward pass (tests/probes/noise_probe.py).  Negative values are safe; positive ones overflow f16.
__global__ void k_scale_from_max(const u32 *maxbits, int32_t *jexp, int bias) {
    const float mx = __uint_as_float(*maxbits);
    int e = 0;
    if (mx > 0.f && mx < INFINITY) frexpf(mx, &e);  // mx = m * 2^e, m in [0.5, 1)
    int s = PLM_R_EXP - e + bias;
    s = max(-100, min(100, s));
    *jexp = (mx > 0.f) ? s : 0;
}
hipError_t plm_launch_maxabs2(const float *a, int64_t na, const float *b, int64_t nb, u32 *maxbits, int32_t *jexp,
                              int jexp_bias, hipStream_t st) {
    hipError_t e = hipMemsetAsync(maxbits, 0, sizeof(u32), st);
    if (e != hipSuccess) return e;
    if (na > 0) hipLaunchKernelGGL(k_maxabs, dim3((unsigned)std::min<int64_t>(512, (na / 4 + 255) / 256 + 1)), dim3(256), 0, st, a, na, maxbits);
    if (nb > 0) hipLaunchKernelGGL(k_maxabs, dim3((unsigned)std::min<int64_t>(256, (nb / 4 + 255) / 256 + 1)), dim3(256), 0, st, b, nb, maxbits);
    hipLaunchKernelGGL(k_scale_from_max, dim3(1), dim3(1), 0, st, maxbits, jexp, jexp_bias);
    return hipGetLastError();
}
hipError_t plm_launch_maxabs(const PlmDims &d, const float *x, u32 *maxbits, int32_t *jexp, hipStream_t st) {
    hipError_t e = hipMemsetAsync(maxbits, 0, sizeof(u32), st);
    if (e != hipSuccess) return e;
    const int64_t n = d.n_local - d.nh_pad_l;
    hipLaunchKernelGGL(k_maxabs, dim3((unsigned)std::min<int64_t>(512, (n / 4 + 255) / 256 + 1)), dim3(256), 0, st, x + d.nh_pad_l, n, maxbits);
    hipLaunchKernelGGL(k_scale_from_max, dim3(1), dim3(1), 0, st, maxbits, jexp, d.jexp_bias);
    return hipGetLastError();
}

// =========================================================================================
// K_expand: native parameter vector -> forward B operand.
//   Bt[b16l][kstep=(u,b)][plane][a][lane=(kg,r)][e] = 2^jexp * J_{i,j}(a,b)
//   i = 16*b16 + r (state a),  j = 32u + 8kg + perm8(e) (state b);  0 when i == j / padding
// =========================================================================================
__device__ __forceinline__ float load_coupling(const PlmDims &d, const float *__restrict__ xj,
                                               const float *__restrict__ xhalo, int I, int ii, int a, int J, int jj,
                                               int b) {
    // coupling between (site 16I+ii, state a) and (site 16J+jj, state b); I is one of this shard's column blocks.
    // xj = coupling part of the LOCAL vector (own block pairs, plm_pair_local); a pair that belongs to the other shard
    // (sharded-state mode) lies in the halo received from it, same block layout.  A block (lo <= hi) holds
    // [a_lo][b_hi][ii_lo][jj_hi].
    const int QQ = d.Q * d.Q;
    if (I == J) {
        if (ii == jj) return 0.f;
        const float *blk = xj + (size_t)plm_pair_local(d, I, I, nullptr) * QQ * 256;
        return ii < jj ? blk[(a * d.Q + b) * 256 + ii * 16 + jj] : blk[(b * d.Q + a) * 256 + jj * 16 + ii];
    }
    int h;
    const int64_t k = plm_pair_local(d, min(I, J), max(I, J), &h);
    const float *blk = k >= 0 ? xj + (size_t)k * QQ * 256 : xhalo + (size_t)h * QQ * 256;
    return I < J ? blk[(a * d.Q + b) * 256 + ii * 16 + jj] : blk[(b * d.Q + a) * 256 + jj * 16 + ii];
}
// Sparse-MFMA layout (plm_internal.h).  One workgroup writes the two tiles (hi plane, lo plane) of an
// instruction slice ci of block u.  Tile = for every state a a dense B fragment of v_smfmac_f32_16x16x64_f16, 64
// lanes x 16 halves, stored as two 1 KB halves: slots 0-7 of every lane at fragment a, slots 8-15 at fragment Q + a
// (the two ds_read_b128 of k_fwd).  Which (site j, state b) a slot holds follows from the instruction's operand
// pairing (profiles/r02_smfmac_probe.txt): pair p of A lane (row, ga) multiplies B lane (n, gb = 2 (ga % 2) + p / 2),
// slots 8 (ga / 2) + 4 (p % 2) + e, and k_fwd gives pair p of slice ci the (site, state group)
//     gl = 4 ci + p,  site 32 u + 8 ga + gl % 8,  states 1 + 4 (gl / 8) + e  (as differences to state 0):
// a slice is four neighbouring sites (one dword of a lane's 8 alignment bytes) against ONE group of four states, so the
// 2-bit positions of a slice depend on the dword only and its values are four byte compares with one constant (k_fwd_w).
// k_fwd_ref: the constant the differences leave out, C[i][a] = sum_{j != i} J_ij(a, 0) (zero in gap mode, where state 0
// is not a model state), unscaled, behind the tiles.
__global__ __launch_bounds__(64) void k_fwd_ref(PlmDims d, const float *__restrict__ x, const float *__restrict__ xhalo,
                                               float *__restrict__ cref, double *__restrict__ cref64) {
    const int b16l = blockIdx.x, b16 = d.b16_lo + b16l, r = blockIdx.y, a = blockIdx.z, t = threadIdx.x;
    const int i = b16 * 16 + r;
    const float *__restrict__ xj = x + d.nh_pad_l;
    double s = 0;
    if (!d.gap_mode && i < d.L)
        for (int j = t; j < d.L; j += 64)
            if (j != i) s += (double)load_coupling(d, xj, xhalo, b16, r, a, j >> 4, j & 15, 0);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (t == 0) {
        cref[((size_t)b16l * 16 + r) * d.Q + a] = (float)s;
        cref64[((size_t)b16l * 16 + r) * d.Q + a] = s;
    }
}
__global__ __launch_bounds__(256) void k_expand(PlmDims d, const float *__restrict__ x,
                                               const float *__restrict__ xhalo,
                                               const int32_t *__restrict__ jexp, _Float16 *__restrict__ Bt) {
    const int NG = PLM_FWD_NG(d.Q), SPU = 4 * NG;
    const int u = blockIdx.x / (2 * NG), ci = blockIdx.x % (2 * NG), b16l = blockIdx.y, b16 = d.b16_lo + b16l;
    const float sc = ldexpf(1.f, *jexp);
    const float *__restrict__ xj = x + d.nh_pad_l;
    _Float16 *tile_hi = Bt + ((size_t)b16l * d.nu * SPU + (size_t)u * SPU + 2 * ci) * (size_t)(2 * d.Q * 512);
    _Float16 *tile_lo = tile_hi + (size_t)(2 * d.Q * 512);
    // blockIdx.z: a slice of the tile's (state, lane) positions -- one position per thread (round 6: a thread used to walk
    // 5 of them, 80 dependent gathers; a shard of three column blocks was 300 such workgroups and took as long as 19 blocks)
    for (int idx = blockIdx.z * 256 + threadIdx.x; idx < d.Q * 64; idx += gridDim.z * 256) {
        const int a = idx >> 6, lane = idx & 63, gb = lane >> 4, r = lane & 15;
        const int i = b16 * 16 + r;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            half8 hi, lo;
#pragma unroll
            for (int e8 = 0; e8 < 8; e8++) {
                const int ga = 2 * half + (gb >> 1), pp = 2 * (gb & 1) + (e8 >> 2), e = e8 & 3;
                const int gl = 4 * ci + pp, s8 = gl & 7, sg = gl >> 3;
                const int b = 4 * sg + e + 1, j = 32 * u + 8 * ga + s8;     // state 0 has no slot (reference state)
                // the difference to the reference state, scaled by a power of two and split hi = f16(v), lo = f16(v - hi):
                // 22 significant bits of the DIFFERENCE (the evaluations that need more run the exact kernel, k_fwd_x)
                float v = 0.f;
                if (b < d.Q && i < d.L && j < d.L && i != j) {
                    v = load_coupling(d, xj, xhalo, b16, r, a, j >> 4, j & 15, b);
                    if (!d.gap_mode) v -= load_coupling(d, xj, xhalo, b16, r, a, j >> 4, j & 15, 0);
                    v *= sc;
                }
                const _Float16 h = (_Float16)v;
                hi[e8] = h;
                lo[e8] = (_Float16)(v - (float)h);
            }
            *(half8 *)(tile_hi + (size_t)(half * d.Q + a) * 512 + lane * 8) = hi;
            *(half8 *)(tile_lo + (size_t)(half * d.Q + a) * 512 + lane * 8) = lo;
        }
    }
}

// The same parameters as the B operand of the EXACT forward GEMM (k_fwd_x): every coupling difference in 39-bit fixed
// point, D = rint((J_ij(a,b) - J_ij(a,0)) 2^(jexp + 23)) with the difference formed in f64 (|D| < 2^38), as FIVE signed
// base-256 digit planes.  (Four planes -- 31 bits of the LARGEST coupling -- were measured first: the alignments' few
// strong couplings are 100 x the typical one, and a quantum of 2^-31 max|J| left a coherent error of 2.7e-8 per
// potential, no better than the f16 planes; with five it is ~1e-10.)  Tile of K step (32-site block u, state group grp) and plane
// p: for every state a two 1 KB fragments of v_mfma_i32_16x16x64_i8 (h16 = 0, 1: the two 16-site halves of the block),
// lane (kg, r) = 16 K values of site column 16 b16 + r: byte 4 e + t = digit p of D for neighbour site
// j = 32 u + 8 kg + 4 h16 + t in state b = 4 grp + 1 + e.  Planes are the slowest index of a column block: the kernel runs
// one plane over the whole K range before the next.
#define PLM_FWDX_PLANES 5
#define PLM_FWDX_SHIFT 23
__global__ __launch_bounds__(256) void k_expand_x(PlmDims d, const float *__restrict__ x,
                                                 const float *__restrict__ xhalo,
                                                 const int32_t *__restrict__ jexp, char *__restrict__ Bt) {
    const int NG = PLM_FWD_NG(d.Q);
    const int u = blockIdx.x / NG, grp = blockIdx.x % NG, b16l = blockIdx.y, b16 = d.b16_lo + b16l;
    const double sc = ldexp(1.0, *jexp + PLM_FWDX_SHIFT);
    const float *__restrict__ xj = x + d.nh_pad_l;
    const size_t tile = (size_t)2 * d.Q * 1024, plane_stride = (size_t)d.nu * NG * tile;
    char *t0 = Bt + (size_t)b16l * PLM_FWDX_PLANES * plane_stride + ((size_t)u * NG + grp) * tile;
    for (int idx = threadIdx.x; idx < 2 * d.Q * 64; idx += 256) {
        const int h16 = idx / (d.Q * 64), rem = idx - h16 * d.Q * 64;
        const int a = rem >> 6, lane = rem & 63, kg = lane >> 4, r = lane & 15;
        const int i = b16 * 16 + r;
        u32 w[PLM_FWDX_PLANES][4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            u32 dig[4], top = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int b = 4 * grp + 1 + e, j = 32 * u + 8 * kg + 4 * h16 + t;
                double v = 0.0;
                if (b < d.Q && i < d.L && j < d.L && i != j) {
                    const float jb = load_coupling(d, xj, xhalo, b16, r, a, j >> 4, j & 15, b);
                    const float j0 = d.gap_mode ? 0.f : load_coupling(d, xj, xhalo, b16, r, a, j >> 4, j & 15, 0);
                    v = ((double)jb - (double)j0) * sc;
                }
                // signed base-256 digits: bytes of R + 0x8080808080 are the digits + 128 (|R| < 2^38 < 0x7f7f7f7f7f)
                const long long R = llrint(v);           // ties to even: unbiased
                const u64 dg = ((u64)R + 0x8080808080ull) ^ 0x8080808080ull;
                dig[t] = (u32)dg;
                top |= (u32)((dg >> 32) & 0xffu) << (8 * t);
            }
            u32 pl[PLM_BWD_MAXPLANES];
            planes_of4(dig[0], dig[1], dig[2], dig[3], pl);
#pragma unroll
            for (int p = 0; p < 4; p++) w[p][e] = pl[p];
            w[4][e] = top;
        }
#pragma unroll
        for (int p = 0; p < PLM_FWDX_PLANES; p++)
            *(uint4 *)(t0 + (size_t)p * plane_stride + (size_t)(h16 * d.Q + a) * 1024 + lane * 16) =
                make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
    }
}
// exact: the operand of k_fwd_x (accurate evaluation) instead of the f16 planes of k_fwd
hipError_t plm_launch_expand(const PlmDims &d, const float *x, const float *xhalo, const int32_t *jexp, void *Bt,
                             int exact, hipStream_t st) {
    if (d.b16_hi <= d.b16_lo) return hipSuccess;
    const int NG = PLM_FWD_NG(d.Q);
    if (exact)
        hipLaunchKernelGGL(k_expand_x, dim3(d.nu * NG, d.b16_hi - d.b16_lo), dim3(256), 0, st, d, x, xhalo, jexp, (char *)Bt);
    else
        hipLaunchKernelGGL(k_expand, dim3(d.nu * 2 * NG, d.b16_hi - d.b16_lo, (d.Q * 64 + 255) / 256), dim3(256), 0, st, d, x, xhalo,
                           jexp, (_Float16 *)Bt);
    hipLaunchKernelGGL(k_fwd_ref, dim3(d.b16_hi - d.b16_lo, 16, d.Q), dim3(64), 0, st, d, x, xhalo,
                       (float *)bt_cref32(d, Bt), (double *)bt_cref64(d, Bt));
    return hipGetLastError();
}

// =========================================================================================
// K_fwd: the coupling part of every conditional, HJ[s,(i,a)] = sum_{j != i} J_ij(a, x_sj)   (row a6, forward half)
//   One-hot(MSA) x J as a GEMM on the 2:4 sparse f16 MFMA.  Workgroup = 256 sequences x one 16-site block x one GROUP
//   of QG = Q / NSG states (NSG = 1: all states); 8 waves x 32 sequences.  K loop: one step = (32-site block u,
//   instruction slice ci, plane hi / lo); A = compressed one-hot fragments built in registers from the packed
//   alignment, B = the group's fragments of the Bt tile streamed global -> LDS by global_load_lds (double buffered, one
//   barrier per step).  Accumulator fragment `a` of a wave holds HJ[s, i, a0 + a] for 16 sites i (lane & 15) and 4
//   sequences per lane.  Epilogues (template parameter MODE): FWD_STORE writes HJ for k_hpass (the fit and plm_eval:
//   softmax, residuals and the field solver live there), FWD_ENERGY / FWD_POTENTIALS serve the statistical energies of
//   row N2.
//   Its accuracy is that of two f16 operand planes (22 bits of every coupling difference) and f32 accumulation over
//   the whole K range: an error of ~5e-7 per potential of which a sixth is the SAME for every sequence of a (site,
//   state) (round 4, tests/probes/potentials_probe.py) -- the gradient sums add that part up N-fold, |g_hip - g_f64| ~
//   3e-11 N L |x|.  Irrelevant far from the optimum; the last iterations of a fit and plm_eval run k_fwd_x below.
// =========================================================================================
struct FwdArgs {
    const int8_t *msa_rm;
    const char *Bt;
    const float *h;       // native vector (fields first)
    const int32_t *jexp;
    float *out;           // FWD_ENERGY: float2 [Np][blocks * NSG] energy partials; FWD_POTENTIALS: [N][L][Q]; FWD_STORE: HJ
};
// k_fwd MODE.  The GEMM yields HJ[s,i,a] = sum_{j != i} J_ij(a, x_sj), the coupling part of every conditional:
// 1 = statistical energies of sequences under a fitted model (reference twins
//     couplings/model.py:25-60 _hamiltonians and :63-109 _single_mutant_hamiltonians): per (sequence, site block, state
//     group) the pair (sum_i HJ[s,i,x_si], sum_i h_i(x_si)) over the sites whose state lies in the group;
// 2 = the potentials HJ[s,i,a] themselves;
// 3 = HJ stored in accumulator order (float4 per lane and state) for k_hpass below: the solver's forward pass.
enum { FWD_ENERGY = 1, FWD_POTENTIALS = 2, FWD_STORE = 3 };

// one K step of the forward GEMM for one wave: QG states x 2 row fragments, each instruction on both 8-slot halves of
// a state's dense fragment.  The B fragments of state A+2 are issued before state A computes (ring of 3 register
// pairs; without it hipcc waits lgkmcnt(0) before every group of MFMAs: LDS latency x QG).
template <int QG, int A>
__device__ __forceinline__ void fwd_state(f32x4 (&acc)[2][QG], const half8 &a0, const half8 &a1, int i0, int i1, u32 lb,
                                          half8 (&bh)[3], half8 (&bl)[3], const DmaPlan &dma) {
    constexpr int NP = (2 * QG + 7) / 8;
    if constexpr (A + 2 < QG) {
        bh[(A + 2) % 3] = lds_read_b128<(A + 2) * 1024>(lb);
        bl[(A + 2) % 3] = lds_read_b128<(QG + A + 2) * 1024>(lb);
    }
    constexpr int newer = (A + 2 < QG) ? 4 : (A + 1 < QG) ? 2 : 0;
    lds_wait<newer>(bh[A % 3], bl[A % 3]);
    // the two LDS reads of the fragment are the two halves of one 16-half dense B operand; a0 / a1 are compressed
    // one-hot fragments (8 halves + 2-bit positions i0 / i1): one instruction per row fragment covers 64 dense K slots
    const half16 bb = __builtin_shufflevector(bh[A % 3], bl[A % 3], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    acc[0][A] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a0, bb, acc[0][A], i0, 0, 0);
    dma_at<QG, A, NP, PLM_DMA_STAGGER_FWD>(dma);
    acc[1][A] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a1, bb, acc[1][A], i1, 0, 0);
}
template <int QG, int... A>
__device__ __forceinline__ void fwd_kstep(f32x4 (&acc)[2][QG], const half8 &a0, const half8 &a1, int i0, int i1, u32 lb,
                                          half8 (&bh)[3], half8 (&bl)[3], const DmaPlan &dma,
                                          std::integer_sequence<int, A...>) {
    bh[0] = lds_read_b128<0>(lb);
    bl[0] = lds_read_b128<QG * 1024>(lb);
    if constexpr (QG > 1) {
        bh[1] = lds_read_b128<1024>(lb);
        bl[1] = lds_read_b128<(QG + 1) * 1024>(lb);
    }
    (fwd_state<QG, A>(acc, a0, a1, i0, i1, lb, bh, bl, dma), ...);
}

template <int Q, int MODE, int NSG>
__global__ __launch_bounds__(512) void k_fwd(PlmDims d, FwdArgs A) {
    static_assert(Q % NSG == 0, "state groups must divide the alphabet");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the ONLY LDS object (guide 5/4a)
    constexpr int QG = Q / NSG;                                   // states of this workgroup
    constexpr int TILE_G = 2 * Q * 1024, TILE = 2 * QG * 1024, NP = (2 * QG + 7) / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // the same number, known to be wave-uniform
    // XCD-aware order: block b runs on XCD b % 8, each XCD has its own L2.  The (site block, state group, sequence
    // tile) work list is site-block major; XCD x takes the contiguous eighth [x, x+1) * nwork / 8 of it, so the ~32
    // blocks resident on an XCD stream the SAME Bt slab (9 MB at the headline; a third of it per state group) at about
    // the same time and a slab is fetched from HBM by at most two XCDs instead of all eight.
    const int nwork = d.nstiles * (d.b16_hi - d.b16_lo) * NSG, per_xcd = (nwork + 7) >> 3;
    const int wk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || wk >= nwork) return;
    const int stile = wk % d.nstiles, sg = (wk / d.nstiles) % NSG, b16l = wk / (d.nstiles * NSG);
    const int tile = b16l * d.nstiles + stile;            // (site block, sequence tile): the index HJ is laid out by
    const int a_lo = sg * QG;                             // first state of the group
    const int b16 = d.b16_lo + b16l;
    const int r = lane & 15, g = lane >> 4;
    const int s_wave = stile * PLM_SEQ_TILE + wave * 32;
    // a step's tile in Bt: [2 halves][Q states][1 KB]; the group's fragments are the runs [a_lo, a_lo + QG) of both halves
    // steps of a 32-site block: 2 NG instruction slices x 2 planes (hi, lo); gap mode needs no special case (k_expand
    // leaves the slots of state 0 zero)
    constexpr int NG = PLM_FWD_NG(Q), SPU = 4 * NG;
    const int nsteps = d.nu * SPU;
    const char *bt = A.Bt + (size_t)b16l * nsteps * TILE_G + (size_t)a_lo * 1024;
    // A operand: byte offsets of the two sequences' rows in msa_rm (< 2^31: Np * Lp32 bytes)
    const u32 arow0 = (u32)(s_wave + r) * (u32)d.Lp32 + 8 * g, arow1 = arow0 + 16 * (u32)d.Lp32;

    f32x4 acc[2][QG];
#pragma unroll
    for (int a = 0; a < QG; a++) {
        acc[0][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#if PLM_PROBE
    unsigned long long pr_wait = 0;
    const unsigned long long pr_t0 = PROBE_NOW();
#endif
    // prologue: tile of the first step
    if (nsteps > 0) {
        const DmaPlan first{bt, smem, wave_s, 2 * QG, (u32)lane * 16, false, nullptr, 0, nullptr, QG, (Q - QG) * 1024};
        dma_issue_all<NP>(first);
    }
    u64 na0 = *(const u64 *)(A.msa_rm + arow0), na1 = *(const u64 *)(A.msa_rm + arow1);   // bytes of the NEXT u
    half8 bh[3], bl[3];
    int t = 0, cur = 0;
    for (int u = 0; u < d.nu; ++u) {
        // hand over the 32 sites of this u (landed: a vmcnt(0) lies between the load and here),
        // then start fetching the next 32
        vm_landed(na0);
        vm_landed(na1);
        const u64 xa0 = na0, xa1 = na1;
        if (u + 1 < d.nu) {
            load_b64_inplace(na0, A.msa_rm, arow0 + 32 * (u + 1));
            load_b64_inplace(na1, A.msa_rm, arow1 + 32 * (u + 1));
        }
        for (int b = 0; b < SPU; ++b, ++t) {
#if PLM_PROBE
            const unsigned long long pa = PROBE_NOW();
#endif
            vm_wait<0>();       // my pieces of this step's tile have landed ...
            __syncthreads();    // ... and so have everybody's; nobody reads the other buffer any more
#if PLM_PROBE
            pr_wait += PROBE_NOW() - pa;
#endif
            const int nxt = cur ^ 1;
            const DmaPlan dma{bt + (size_t)(t + 1) * TILE_G, smem + nxt * TILE, wave_s, (t + 1 < nsteps) ? 2 * QG : 0,
                              (u32)lane * 16, wave_s >= 4, nullptr, 0, nullptr, QG, (Q - QG) * 1024};
            const u32 lb = lds_addr(smem + cur * TILE + lane * 16);
            // compressed one-hot fragments of instruction slice ci = b / 2 (the two planes of a slice share them):
            // pair p = (site s8, state group sg) of the lane's 8 sites; value 1 in the pair's first slot when the
            // site's state lies in the group, its 2-bit position = state % 4; the pair's second slot stays 0
            half8 a0, a1;
            int i0 = 0, i1 = 0;
            {
                const int ci = b >> 1;
#pragma unroll
                for (int pp = 0; pp < 4; pp++) {
                    const int gl = 4 * ci + pp, s8 = gl & 7, kg = gl >> 3;
                    const u32 x0 = (u32)(xa0 >> (8 * s8)) & 0xffu, x1 = (u32)(xa1 >> (8 * s8)) & 0xffu;
                    // state x > 0 sits in slot (x - 1); x = 0 (the reference state) wraps to a group that does not exist
                    ((u32 *)&a0)[pp] = (((x0 - 1u) >> 2) == (u32)kg) ? 0x3C00u : 0u;
                    ((u32 *)&a1)[pp] = (((x1 - 1u) >> 2) == (u32)kg) ? 0x3C00u : 0u;
                    i0 |= (int)(((x0 - 1u) & 3u) << (4 * pp));
                    i1 |= (int)(((x1 - 1u) & 3u) << (4 * pp));
                }
            }
            fwd_kstep<QG>(acc, a0, a1, i0, i1, lb, bh, bl, dma, std::make_integer_sequence<int, QG>{});
            cur = nxt;
        }
    }
#if PLM_PROBE
    if (lane == 0) {
        atomicAdd(&plm_probe_acc[0][0], pr_wait); atomicAdd(&plm_probe_acc[0][3], PROBE_NOW() - pr_t0);
        atomicAdd(&plm_probe_acc[0][5], 1ull);
    }
#endif
    // Potential of accumulator element (m, a, e): the descaled sum of the differences PLUS the reference-state constant
    // C_i(a) = sum_{j != i} J_ij(a, 0) (k_fwd_ref, f64), added in f64 and rounded ONCE.  (A C rounded to f32 would shift
    // the potentials of every sequence of a (site, state) by the same ~1e-7 |C| -- a coherent error the gradient sums
    // add up N-fold: round 3, tests/probes/fwd_bias_probe.py; the f64 constant has no such part.)
    const int je = *A.jexp;
    const double sc64 = ldexp(1.0, -je);
    const double *cref = (const double *)((const float *)(A.Bt + (size_t)d.blk_per_shard * d.nksteps * TILE_G) +
                                          (size_t)d.blk_per_shard * 16 * Q) + ((size_t)b16l * 16 + r) * Q + a_lo;
    auto value = [&](int m, int a, int e) -> float { return (float)__builtin_fma((double)acc[m][a][e], sc64, cref[a]); };
    if constexpr (MODE == FWD_STORE) {
        // The stored potentials: acc 2^-e + C with the f64 constant as a hi + lo pair of floats -- fma(acc, 2^-e, C_lo) + C_hi,
        // two f32 operations per value instead of convert / f64 fma / convert (round 5: the f64 epilogue was 0.26 of
        // k_fwd_w's 2.8 ms, with nothing to overlap it).  The scale is a power of two (exact); what made the f32
        // constant of round 2 harmful was its rounding, the same for every sequence -- the pair carries all 48 bits, and
        // the two roundings left depend on acc, i.e. differ from sequence to sequence.  k_fwd_w does the same operations.
        float4 *hj = (float4 *)A.out + ((size_t)tile * 8 + wave) * 2 * Q * 64 + lane;
        const float sc32 = (float)sc64;
#pragma unroll
        for (int a = 0; a < QG; a++) {
            const double c = cref[a];
            const float chi = (float)c, clo = (float)(c - (double)chi);
#pragma unroll
            for (int m = 0; m < 2; m++)
                store_hj(&hj[(size_t)(m * Q + a_lo + a) * 64],
                         __builtin_fmaf(acc[m][a][0], sc32, clo) + chi, __builtin_fmaf(acc[m][a][1], sc32, clo) + chi,
                         __builtin_fmaf(acc[m][a][2], sc32, clo) + chi, __builtin_fmaf(acc[m][a][3], sc32, clo) + chi);
        }
        return;
    }
    {
        // ---- statistical energies / potentials of the given sequences (no softmax) ------------
        const int i = b16 * 16 + r;
        const bool site_ok = i < d.L;
        if constexpr (MODE == FWD_ENERGY) {
            float hv[QG];
#pragma unroll
            for (int a = 0; a < QG; a++) hv[a] = site_ok ? A.h[(size_t)(i - d.h_site0) * Q + a_lo + a] : 0.f;
            const int nslot = (d.b16_hi - d.b16_lo) * NSG;
#pragma unroll
            for (int m = 0; m < 2; m++) {
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int s = s_wave + 16 * m + 4 * g + reg;
                    const int xi = A.msa_rm[(size_t)s * d.Lp32 + i] - a_lo;
                    float ej = 0.f, eh = 0.f;
#pragma unroll
                    for (int a = 0; a < QG; a++) {      // a state outside this workgroup's group (or padding) leaves 0
                        ej = (a == xi) ? value(m, a, reg) : ej;
                        eh = (a == xi) ? hv[a] : eh;
                    }
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {   // sum over the 16 sites of the block (lanes r)
                        ej += __shfl_xor(ej, o, 64);
                        eh += __shfl_xor(eh, o, 64);
                    }
                    if (r == 0) *(float2 *)(A.out + ((size_t)s * nslot + b16l * NSG + sg) * 2) = make_float2(ej, eh);
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 2; m++) {
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int s = s_wave + 16 * m + 4 * g + reg;
                    if (site_ok && s < d.N) {
                        float *o = A.out + ((size_t)s * d.L + i) * d.Qc + a_lo;   // the API's array: the problem's alphabet
#pragma unroll
                        for (int a = 0; a < QG; a++)
                            if (a_lo + a < d.Qc) o[a] = value(m, a, reg);
                    }
                }
            }
        }
        return;
    }
}


// =========================================================================================
// K_fwd_x: the same potentials EXACTLY (row a6, accurate evaluation: the last iterations of a fit, plm_eval).
//   Round 4 took the error of the evaluation apart (tests/probes/potentials_probe.py): what the
//   gradient sums cannot average out is the part of a potential's error that is the same for every sequence -- operand
//   bits lost (22 of the f32 differences' 24, and the f32 rounding of the differences themselves), and the rounding of
//   the matrix cores' f32 accumulation, which has a small negative mean.  f64 outer sums and a third f16 plane brought
//   it from 4.5e-4 |x| to 1.6e-4 (headline) / 3.9e-4 (N = 100 000), no further.  Integer arithmetic has none of it:
//     * operand = every coupling difference in 39-bit fixed point (difference formed in f64, one unbiased rounding at
//       2^-39 of the largest), five signed base-256 digit planes (k_expand_x);
//     * one-hot(MSA) x digit plane on v_mfma_i32_16x16x64_i8 with int32 accumulators: exact, whatever the order
//       (|sum| <= 128 * 128 * L);
//     * a workgroup runs ONE plane over the whole K range, then adds its accumulators, shifted by 8 p bits, to int64
//       sums and starts the next plane -- five flushes in all;
//     * potential = sum * 2^-(jexp + 30) + C_i(a) in f64, rounded once to f32.
//   The result is the correctly rounded potential of the f32 parameters up to 2^-39 max|J| per coupling; it does not
//   depend on tiling or order (bit-reproducible).  Same tiling as k_fwd with NSG state groups (QG = 7 states per
//   workgroup at Q = 21: 56 accumulator + 112 f64-sum registers), same tile geometry ([2][Q][1 KB] per K step: here the
//   two 16-site halves of a 32-site block, one MFMA each), same LDS-DMA double buffer.  K step = (plane, 32-site block
//   u, state group grp of 4 states); the one-hot A fragment of a (row fragment, half) is expanded from 4 alignment
//   bytes per lane with k_bwd's two-VALU-per-dword trick (value -128 for a match, folded into the scale).
//   Cost at the headline: 2.5 x the MFMA work of k_fwd (5 planes of int8 at the f16-sparse rate against 2).
// =========================================================================================
template <int QG, int F>
__device__ __forceinline__ void fwdx_frag(i32x4 (&acc)[2][QG], const i32x4 (&af)[2][2], u32 lb, i32x4 (&bf)[3],
                                          const DmaPlan &dma) {
    constexpr int NF = 2 * QG, A = F % QG, H = F / QG, NP = (NF + 7) / 8;
    if constexpr (F + 2 < NF) bf[(F + 2) % 3] = lds_read_b128_i<(F + 2) * 1024>(lb);
    constexpr int newer = (F + 2 < NF) ? 2 : (F + 1 < NF) ? 1 : 0;
    lds_wait_i<newer>(bf[F % 3]);
    acc[0][A] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0][H], bf[F % 3], acc[0][A], 0, 0, 0);
    dma_at<NF, F, NP, PLM_DMA_STAGGER_FWD>(dma);
    acc[1][A] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1][H], bf[F % 3], acc[1][A], 0, 0, 0);
}
template <int QG, int... F>
__device__ __forceinline__ void fwdx_kstep(i32x4 (&acc)[2][QG], const i32x4 (&af)[2][2], u32 lb, i32x4 (&bf)[3],
                                           const DmaPlan &dma, std::integer_sequence<int, F...>) {
    bf[0] = lds_read_b128_i<0>(lb);
    bf[1] = lds_read_b128_i<1024>(lb);
    (fwdx_frag<QG, F>(acc, af, lb, bf, dma), ...);
}

template <int Q, int MODE, int NSG>
__global__ __launch_bounds__(512) void k_fwd_x(PlmDims d, FwdArgs A) {
    static_assert(Q % NSG == 0, "state groups must divide the alphabet");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QG = Q / NSG, NG = PLM_FWD_NG(Q);
    constexpr int TILE_G = 2 * Q * 1024, TILE = 2 * QG * 1024, NP = (2 * QG + 7) / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    u32 k7f = 0x7f7f7f7fu;
    asm volatile("" : "+v"(k7f));                              // a VGPR constant (see onehot16)
    // work list and XCD-aware order as in k_fwd
    const int nwork = d.nstiles * (d.b16_hi - d.b16_lo) * NSG, per_xcd = (nwork + 7) >> 3;
    const int wk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || wk >= nwork) return;
    const int stile = wk % d.nstiles, sg = (wk / d.nstiles) % NSG, b16l = wk / (d.nstiles * NSG);
    const int tile = b16l * d.nstiles + stile;
    const int a_lo = sg * QG;
    const int b16 = d.b16_lo + b16l;
    const int r = lane & 15, g = lane >> 4;
    const int s_wave = stile * PLM_SEQ_TILE + wave * 32;
    const int per_plane = d.nu * NG, nsteps = PLM_FWDX_PLANES * per_plane;      // tiles of a column block, plane-major
    const char *bt = A.Bt + (size_t)b16l * nsteps * TILE_G + (size_t)a_lo * 1024;
    // A operand: the lane's 8 sites (8 g .. 8 g + 7 of the 32-site block) of its two sequences
    const u32 arow0 = (u32)(s_wave + r) * (u32)d.Lp32 + 8 * g, arow1 = arow0 + 16 * (u32)d.Lp32;

    i32x4 acc[2][QG];
    long long sum64[2][QG][4];
#pragma unroll
    for (int a = 0; a < QG; a++)
#pragma unroll
        for (int m = 0; m < 2; m++) {
            acc[m][a] = (i32x4){0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; e++) sum64[m][a][e] = 0;
        }
    if (nsteps > 0) {
        const DmaPlan first{bt, smem, wave_s, 2 * QG, (u32)lane * 16, false, nullptr, 0, nullptr, QG, (Q - QG) * 1024};
        dma_issue_all<NP>(first);
    }
    u64 na0 = *(const u64 *)(A.msa_rm + arow0), na1 = *(const u64 *)(A.msa_rm + arow1);   // bytes of the NEXT u
    u64 xa0 = 0, xa1 = 0;
    i32x4 bf[3];
    int cur = 0, u = 0, grp = 0;
    int shift = 0;                                             // 8 * plane
    for (int t = 0; t < nsteps; ++t) {
        vm_wait<0>();       // my pieces of this step's tile (and the alignment bytes in flight) have landed ...
        __syncthreads();    // ... and so have everybody's; nobody reads the other buffer any more
        if (grp == 0) {
            // hand over the 32 sites of this u, then start fetching the next block's (the plane loop wraps around)
            vm_landed(na0);
            vm_landed(na1);
            xa0 = na0;
            xa1 = na1;
            const int un = (u + 1 == d.nu) ? 0 : u + 1;
            load_b64_inplace(na0, A.msa_rm, arow0 + 32 * un);
            load_b64_inplace(na1, A.msa_rm, arow1 + 32 * un);
        }
        const int nxt = cur ^ 1;
        const DmaPlan dma{bt + (size_t)(t + 1) * TILE_G, smem + nxt * TILE, wave_s, (t + 1 < nsteps) ? 2 * QG : 0,
                          (u32)lane * 16, wave_s >= 4, nullptr, 0, nullptr, QG, (Q - QG) * 1024};
        const u32 lb = lds_addr(smem + cur * TILE + lane * 16);
        // one-hot fragments [row fragment][16-site half]: dword e = the lane's 4 sites of that half against state
        // 4 grp + 1 + e, -128 for a match (byte 4 e + t <-> neighbour site t, as k_expand_x lays the digits out)
        i32x4 af[2][2];
        {
            const u32 b0 = (u32)(4 * grp + 1) * 0x01010101u;
            const u32 k80 = 0x80808080u;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const u32 x0 = (u32)(xa0 >> (32 * h)), x1 = (u32)(xa1 >> (32 * h));
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const u32 bb = b0 + (u32)e * 0x01010101u;
                    u32 y0, y1, v0, v1;
                    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(y0) : "v"(x0), "s"(bb), "v"(k7f));
                    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(v0) : "v"(y0), "s"(k80));
                    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(y1) : "v"(x1), "s"(bb), "v"(k7f));
                    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(v1) : "v"(y1), "s"(k80));
                    af[0][h][e] = (int)v0;
                    af[1][h][e] = (int)v1;
                }
            }
        }
        fwdx_kstep<QG>(acc, af, lb, bf, dma, std::make_integer_sequence<int, 2 * QG>{});
        cur = nxt;
        if (++grp == NG) {
            grp = 0;
            if (++u == d.nu) {
                // end of a plane: its integer sums, weighted 256^p, go to the 64-bit sums (|total| < 2^23 2^32: exact)
                u = 0;
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int a = 0; a < QG; a++) {
#pragma unroll
                        for (int e = 0; e < 4; e++) sum64[m][a][e] += (long long)acc[m][a][e] * (1ll << shift);
                        acc[m][a] = (i32x4){0, 0, 0, 0};
                    }
                shift += 8;
            }
        }
    }
    // potential = sum / (-128 2^(jexp + 23)) + C_i(a), in f64, rounded once
    const double sc64 = -ldexp(1.0, -(*A.jexp + PLM_FWDX_SHIFT + 7));
    const double *cref = (const double *)((const float *)(A.Bt + (size_t)d.blk_per_shard * d.nksteps * TILE_G) +
                                          (size_t)d.blk_per_shard * 16 * Q) + ((size_t)b16l * 16 + r) * Q + a_lo;
    auto value = [&](int m, int a, int e) -> float { return (float)__builtin_fma((double)sum64[m][a][e], sc64, cref[a]); };
    if constexpr (MODE == FWD_STORE) {
        float4 *hj = (float4 *)A.out + ((size_t)tile * 8 + wave) * 2 * Q * 64 + lane;
#pragma unroll
        for (int a = 0; a < QG; a++) {
#pragma unroll
            for (int m = 0; m < 2; m++)
                store_hj(&hj[(size_t)(m * Q + a_lo + a) * 64], value(m, a, 0), value(m, a, 1), value(m, a, 2), value(m, a, 3));
        }
    } else {        // FWD_POTENTIALS
        const int i = b16 * 16 + r;
#pragma unroll
        for (int m = 0; m < 2; m++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int s = s_wave + 16 * m + 4 * g + reg;
                if (i < d.L && s < d.N) {
                    float *o = A.out + ((size_t)s * d.L + i) * d.Qc + a_lo;
#pragma unroll
                    for (int a = 0; a < QG; a++)
                        if (a_lo + a < d.Qc) o[a] = value(m, a, reg);
                }
            }
        }
    }
}

// dynamic LDS above 64 KB needs the attribute once per (kernel, device); std::call_once makes the latch safe when
// several host threads create contexts at the same time (dist.ThreadedShards does exactly that)
template <auto KERNEL> static hipError_t plm_allow_lds(size_t lds) {
    static std::once_flag once[PLM_MAX_DEVICES];
    static hipError_t result[PLM_MAX_DEVICES];
    const int dev = plm_current_device();
    std::call_once(once[dev], [&] {
        result[dev] = lds > 65536 ? hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                  : hipSuccess;
    });
    return result[dev];
}

// ---- k_fwd_w: the plain forward GEMM of the fit with 512 sequences per workgroup, K loop in assembly ----------------
// Workgroup = 4 waves (one per SIMD) x 128 sequences (8 row fragments) x one 16-site block x ONE group of 7 states; the
// wave's 8 x 7 accumulator fragments live in a[0:223], the slice blocks of plm_fwd_asm.inc (scripts/gen_fwd_asm.py:
// register map, schedule, synchronisation) do everything between the prologue and the store.  Same instruction, same
// operands and the same K order per accumulator as k_fwd: the stored potentials are bit-identical to k_fwd<21, STORE>'s,
// and HJ keeps its layout (a wave here = four (wave, row fragment pair) slots of k_fwd's two 256-sequence tiles).
#ifndef PLM_FWDW_INC
#define PLM_FWDW_INC "plm_fwd_asm.inc"
#endif
#include PLM_FWDW_INC
template <int IDX> __device__ __forceinline__ f32x4 fwdw_acc_read() {
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\t"
                 "v_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "n"(IDX), "n"(IDX + 1), "n"(IDX + 2), "n"(IDX + 3));
    return v;
}
template <int M, int A>
__device__ __forceinline__ void fwdw_store(float4 *hj, const float *chi, const float *clo, float sc32) {
    const f32x4 v = fwdw_acc_read<(M * PLM_FWDW_QG + A) * 4>();
    const float h = chi[A], l = clo[A];       // the f64 constant C_i(a) as hi + lo (see k_fwd's store epilogue)
    store_hj(&hj[(size_t)((M & 1) * 21 + A) * 64 + (size_t)(M >> 1) * 2 * 21 * 64],
             __builtin_fmaf(v[0], sc32, l) + h, __builtin_fmaf(v[1], sc32, l) + h,
             __builtin_fmaf(v[2], sc32, l) + h, __builtin_fmaf(v[3], sc32, l) + h);
}
template <int M, int... A>
__device__ __forceinline__ void fwdw_store_row(float4 *hj, const float *chi, const float *clo, float sc32, std::integer_sequence<int, A...>) {
    (fwdw_store<M, A>(hj, chi, clo, sc32), ...);
}
template <int... M>
__device__ __forceinline__ void fwdw_store_all(float4 *hj, const float *chi, const float *clo, float sc32, std::integer_sequence<int, M...>) {
    (fwdw_store_row<M>(hj, chi, clo, sc32, std::make_integer_sequence<int, PLM_FWDW_QG>{}), ...);
}

__global__ __launch_bounds__(256) void k_fwd_w(PlmDims d, FwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q = 21, QG = PLM_FWDW_QG, NSG = Q / QG, NM = PLM_FWDW_NM, SLOT = PLM_FWDW_SLOT;
    constexpr int TILE_G = 2 * Q * 1024, NG = PLM_FWD_NG(Q), SPU = 2 * NG;      // slices of a 32-site block
    static_assert(NG == 5 && 2 * 2 * QG * 1024 == SLOT, "plm_fwd_asm.inc is generated for 21 states in groups of 7");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work list as in k_fwd (site-block major, XCD x takes a contiguous eighth), over PAIRS of 256-sequence tiles
    const int npair = (d.nstiles + 1) >> 1;
    const int nwork = npair * (d.b16_hi - d.b16_lo) * NSG, per_xcd = (nwork + 7) >> 3;
    const int wk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || wk >= nwork) return;
    const int spair = wk % npair, sg = (wk / npair) % NSG, b16l = wk / (npair * NSG);
    const int a_lo = sg * QG;
    const int r = lane & 15, g = lane >> 4;
    const int stile = 2 * spair + (wv >> 1);                  // k_fwd's 256-sequence tile of this wave
    const bool tile_ok = stile < d.nstiles;                   // an odd tile count leaves the last pair's second half empty
    const int s_wave = tile_ok ? stile * PLM_SEQ_TILE + (wv & 1) * 128 : 0;
    const int nsteps = d.nu * SPU;                            // one step = one slice = both planes
    const char *bt = A.Bt + (size_t)b16l * nsteps * 2 * TILE_G + (size_t)a_lo * 1024 + wv * 1024;
    const u32 arow = (u32)(s_wave + r) * (u32)d.Lp32 + 8 * g, rstride = 16 * (u32)d.Lp32;

    // a step's tile in Bt is two planes x [2 halves][Q states][1 KB]; the group's fragments are four runs of QG KB.
    // LDS slot = those runs back to back ([plane][half][QG]); piece p of the slot = run p / QG, fragment p % QG; wave wv
    // copies pieces wv, wv + 4, ... (28 pieces: 7 each)
    const u32 l16 = (u32)lane * 16;
    u32 vo[PLM_FWDW_NVMEM];
#pragma unroll
    for (int k = 0; k < PLM_FWDW_NVMEM; k++) {
        const int p = wv + 4 * k, run = p / QG, f = p - run * QG;
        vo[k] = l16 + (u32)((run >> 1) * TILE_G + (run & 1) * Q * 1024 + f * 1024) - (u32)wv * 1024;
    }
    const u32 lw = lds_addr(smem + lane * 16);
    const u32 lw_base = __builtin_amdgcn_readfirstlane(lw - l16);
    const u32 m0t0 = lw_base + wv * 1024;
    const u32 cnt = lw_base + 4 * SLOT;
    if (tid == 0) *(u32 *)(smem + 4 * SLOT) = 0;
    const u32 one = 1, sel = 0x0c0c0200u;
    u32 st, sp;
    asm volatile(PLM_FWDW_ZERO_ASM ::: PLM_FWDW_CLOBBERS);
    for (int i = 0; i < 3; i++) {
        const int stepc = min(i, nsteps - 1);
        asm volatile(PLM_FWDW_ISSUE_ASM
                     :
                     : [tsrc] "s"(bt + (size_t)stepc * 2 * TILE_G), [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]),
                       [vo3] "v"(vo[3]), [vo4] "v"(vo[4]), [vo5] "v"(vo[5]), [vo6] "v"(vo[6]), [m0t] "s"(m0t0 + i * SLOT)
                     : "m0", "scc", "memory");
    }
    // alignment bytes of u = 0 -> the per-u operands; the loads of u = 1 go out
    asm volatile(PLM_FWDW_LOADS_ASM : : [arow] "v"(arow), [rstride] "s"(rstride), [xsrc] "s"(A.msa_rm) : PLM_FWDW_CLOBBERS);
    asm volatile(PLM_FWDW_UPREP_ASM
                 :
                 : [arow] "v"(arow), [rstride] "s"(rstride), [xsrc] "s"(A.msa_rm + 32 * min(1, d.nu - 1)), [sel] "s"(sel)
                 : PLM_FWDW_CLOBBERS);
    vm_wait<0>();
    __syncthreads();
    asm volatile(PLM_FWDW_PRIME_ASM : : [kg] "s"(0), [lbn] "v"(lw) : PLM_FWDW_CLOBBERS);

    int sc = 0;
    u32 tgt = 0;
    const char *tsrc = bt + (size_t)min(3, nsteps - 1) * 2 * TILE_G;
    int step = 0;
#define FWDW_SLICE(WHICH, KG)                                                                                          \
    {                                                                                                                  \
        const int sn = (sc + 1) & 3, snn = (sc + 3) & 3;                                                               \
        asm volatile(WHICH                                                                                             \
                     : [st] "=&s"(st), [sp] "=&s"(sp)                                                                  \
                     : [lb] "v"(lw + sc * SLOT), [lbn] "v"(lw + sn * SLOT), [kg] "s"(KG), [tsrc] "s"(tsrc),          \
                       [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [vo4] "v"(vo[4]),       \
                       [vo5] "v"(vo[5]), [vo6] "v"(vo[6]), [m0t] "s"(m0t0 + snn * SLOT), [cnt] "v"(cnt),               \
                       [one] "v"(one), [tgt] "s"(tgt)                                                                  \
                     : PLM_FWDW_CLOBBERS);                                                                             \
        sc = sn;                                                                                                       \
        tgt += 4;                                                                                                      \
        tsrc += (step + 4 < nsteps) ? (size_t)2 * TILE_G : 0;      /* past the last step it is copied again */         \
        ++step;                                                                                                        \
    }
    for (int u = 0; u < d.nu; ++u) {
#pragma unroll 1
        for (int k = 0; k < NG; ++k) {
            // slice 2k on set 0 builds slice 2k + 1 (second dword of the lane's sites, same state group) ...
            FWDW_SLICE(PLM_FWDW_EVEN_ASM, k)
            if (k == NG - 1)      // ... the per-u operands are free now: those of u + 1 (the loads of u + 2 go out)
                asm volatile(PLM_FWDW_UPREP_ASM
                             :
                             : [arow] "v"(arow), [rstride] "s"(rstride),
                               [xsrc] "s"(A.msa_rm + 32 * min(u + 2, d.nu - 1)), [sel] "s"(sel)
                             : PLM_FWDW_CLOBBERS);
            // ... slice 2k + 1 on set 1 builds slice 2k + 2: first dword, next state group (of the next u behind the last)
            FWDW_SLICE(PLM_FWDW_ODD_ASM, k + 1 < NG ? k + 1 : 0)
        }
    }
#undef FWDW_SLICE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    if (!tile_ok) return;
    const int je = *A.jexp;
    const double sc64 = ldexp(1.0, -je);
    const double *cref = (const double *)((const float *)(A.Bt + (size_t)d.blk_per_shard * d.nksteps * TILE_G) +
                                          (size_t)d.blk_per_shard * 16 * Q) + ((size_t)b16l * 16 + r) * Q + a_lo;
    // k_fwd's layout: [tile][wave of 32 sequences][row fragment pair][state][lane]; this wave's row fragment M is row
    // fragment M & 1 of wave 4 (wv & 1) + M / 2 there
    float4 *hj = (float4 *)A.out + ((size_t)(b16l * d.nstiles + stile) * 8 + (wv & 1) * 4) * 2 * Q * 64 + (size_t)a_lo * 64 + lane;
    float chi[QG], clo[QG];
#pragma unroll
    for (int a = 0; a < QG; a++) {
        const double c = cref[a];
        chi[a] = (float)c;
        clo[a] = (float)(c - (double)chi[a]);
    }
    fwdw_store_all(hj, chi, clo, (float)sc64, std::make_integer_sequence<int, NM>{});
}

// state groups per workgroup of the exact forward GEMM (56 + 112 registers of accumulators and f64 sums at 7 states)
int plm_fwd_groups(int q, int exact) {
    if (!exact) return q == 32 ? 2 : 1;
    return q == 32 ? 8 : q == 21 ? 3 : q == 20 ? 4 : 1;
}
template <int Q, int MODE, int NSG, bool EXACT>
static hipError_t fwd_launch(const PlmDims &d, const FwdArgs &A, hipStream_t st) {
    const int nwork = d.nstiles * (d.b16_hi - d.b16_lo) * NSG;
    const dim3 grid(8 * ((nwork + 7) / 8)), block(512);   // XCD-aware order, padded to 8
    const size_t lds = (size_t)2 * 2 * (Q / NSG) * 1024;  // double buffer of the group's fragments
    if constexpr (EXACT) {
        hipError_t e = plm_allow_lds<k_fwd_x<Q, MODE, NSG>>(lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_fwd_x<Q, MODE, NSG>), grid, block, lds, st, d, A);
    } else {
        hipError_t e = plm_allow_lds<k_fwd<Q, MODE, NSG>>(lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_fwd<Q, MODE, NSG>), grid, block, lds, st, d, A);
    }
    return hipGetLastError();
}
static hipError_t fwdw_launch(const PlmDims &d, const FwdArgs &A, hipStream_t st) {
    const int nwork = ((d.nstiles + 1) / 2) * (d.b16_hi - d.b16_lo) * (21 / PLM_FWDW_QG);
    const size_t lds = (size_t)4 * PLM_FWDW_SLOT + 16;   // ring of four tiles + the arrival counter
    hipError_t e = plm_allow_lds<k_fwd_w>(lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fwd_w, dim3(8 * ((nwork + 7) / 8)), dim3(256), lds, st, d, A);
    return hipGetLastError();
}
static hipError_t launch_forward_mode(const PlmDims &d, const FwdArgs &A, int mode, int exact, hipStream_t st) {
    if (d.b16_hi <= d.b16_lo) return hipSuccess;
#define FWD_CASE(QQ, NP, NX)      /* NP / NX: state groups per workgroup of the plain / the exact kernel (plm_fwd_groups) */ \
    case QQ:                                                                                           \
        if (mode == FWD_ENERGY) return fwd_launch<QQ, FWD_ENERGY, NP, false>(d, A, st);                \
        if (mode == FWD_POTENTIALS && exact) return fwd_launch<QQ, FWD_POTENTIALS, NX, true>(d, A, st); \
        if (mode == FWD_POTENTIALS) return fwd_launch<QQ, FWD_POTENTIALS, NP, false>(d, A, st);        \
        if (exact) return fwd_launch<QQ, FWD_STORE, NX, true>(d, A, st);                               \
        if (QQ == 21 && d.fwd_w) return fwdw_launch(d, A, st);                                         \
        return fwd_launch<QQ, FWD_STORE, NP, false>(d, A, st);
    switch (d.Q) {
        FWD_CASE(32, 2, 8)
        FWD_CASE(21, 1, 3)
        FWD_CASE(20, 1, 4)
        FWD_CASE(5, 1, 1)
        FWD_CASE(4, 1, 1)
    default:
        return hipErrorInvalidValue;
    }
#undef FWD_CASE
}
// statistical energies: mode 1 -> out = float2 [Np][blocks] partial sums, mode 2 -> out = potentials [N][L][Q]
// (exact, potentials only: from k_fwd_x -- Bt must hold its operand)
hipError_t plm_launch_forward_energy(const PlmDims &d, const int8_t *msa_rm, const void *Bt, const float *x,
                                     const int32_t *jexp, int potentials, int exact, float *out, hipStream_t st) {
    const FwdArgs A{msa_rm, (const char *)Bt, x, jexp, out};
    return launch_forward_mode(d, A, potentials ? FWD_POTENTIALS : FWD_ENERGY, potentials ? exact : 0, st);
}
// the forward GEMM alone; HJ goes to HBM in accumulator order.  exact: k_fwd_x (the last iterations of a fit, plm_eval)
hipError_t plm_launch_forward_store(const PlmDims &d, const int8_t *msa_rm, const void *Bt, const int32_t *jexp,
                                    float *hj, int exact, hipStream_t st) {
    const FwdArgs A{msa_rm, (const char *)Bt, nullptr, jexp, hj};
    return launch_forward_mode(d, A, FWD_STORE, exact, st);
}
size_t plm_hj_bytes(const PlmDims &d) { return (size_t)d.nstiles * (d.b16_hi - d.b16_lo) * 8 * 2 * d.Q * 1024; }

// =========================================================================================
// Field solver of the variable-projection fit.  For fixed couplings the objective is a sum
// of independent, strictly convex problems in the fields of one site:
//     phi_i(h) = sum_s w_s [ lse_a(HJ[s,i,a] + h_a) - (HJ[s,i,x_si] + h_{x_si}) ] + lambda_h |h|^2
// k_hpass streams HJ once (same tiling as k_fwd: 256 sequences x 16 sites per workgroup, the 8 (sequence, site)
// pairs of a lane belong to ONE site, so the softmax is in-lane register work) and produces per workgroup and site
//     g_a  = sum_s w_s (P_s(a) - [x_si = a])                      NS = Q values
//     M_ab = sum_s w_s P_s(a) P_s(b), b >= a                      Q (Q+1) / 2 values
// summed deterministically (lanes -> wave via shuffles, waves -> workgroup through LDS in fixed order, workgroups
// -> site in f64 by k_hsolve), and with WRITE_RT also the residuals (backward-pass B fragments) and -log P partials
// of the solver's forward epilogue.  k_hsolve takes one Newton step per site: H = diag(rowsum M) - M + 2 lambda_h I.
// =========================================================================================
#define PLM_HSTATS(Q) ((Q) + (Q) * ((Q) + 1) / 2)
// smallest alphabet that runs on the instantiation for Q states (plm_q_template: 2-4 -> 4, 5 -> 5, 6-20 -> 20, 21 -> 21, 22-32 -> 32)
__host__ __device__ constexpr int plm_qc_min(int Q) { return Q == 4 ? 2 : Q == 5 ? 5 : Q == 20 ? 6 : Q == 21 ? 21 : 22; }
#ifndef PLM_HESS_SAMPLE
#define PLM_HESS_SAMPLE 32    // measured round 6 (with the exact diagonal): 8 / 16 / 32 / 64 / 128 -> 95.7 / 96.9 / 100.5 / 99.5 / 99.2 it/s in the bench window
#endif
#ifndef PLM_NEWTON_CAP
#define PLM_NEWTON_CAP 3.0   // largest change of a field in one Newton step
#endif
// sum over the 4 lanes {l, l^16, l^32, l^48} (the 4 sequence groups of one site in an accumulator fragment),
// result in all of them: gfx950's row / half-wave swaps, two VALU ops per step instead of an LDS round trip
__device__ __forceinline__ float sum_over_g(float v) {
    unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    u = __float_as_uint(v);
    auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// exp_softmax_note.  The exponentials of the softmax: __expf(x) = v_exp_f32(x * 1.44269502f).  The float nearest to log2(e)
// is 1.33e-8 (relative) too small and the product is rounded, so the result carries a relative error of ~0.7 |x log2 e| ulp
// with a POSITIVE mean for the negative arguments of a softmax (scripts/ubench/softmax_bias.hip on the MI355X: mean +0.06
// ... +4.4 ulp for x in [-1, 0) ... [-24, -16); v_exp_f32 itself: -0.06 ulp): every softmax runs slightly too warm, the
// same way in every sequence.  Likewise the argument: potential + field rounded to f32.  Both are errors of ~1e-9 in P
// that do NOT average out over the sequences; a float32 emulation of this kernel on the host (round 4) shows the
// coherent part of the residuals' error vanish exactly when the exponentials' arguments are exact.  Round 3 corrected
// the constant alone, saw nothing (the forward GEMM's error was 5 x larger then) and dropped it.  The passes of an
// accurate evaluation (template parameter XACT of k_hpass) form the arguments in f64; the plain passes keep __expf.
// log(z) = log2(z) * ln 2 with the same care (the f32 ln 2 is 2.1e-9 too large; irrelevant for the gradient, kept exact
// for the objective's sake)
__device__ __forceinline__ float log_unbiased(float z) {
    const float l2 = __builtin_amdgcn_logf(z);
    return __builtin_fmaf(l2, -1.904654323148236e-9f, l2 * 0.693147182464599609375f);   // ln 2 = HI + LO
}
// (x == a) ? if_eq : if_ne as ONE compare into VCC and the select right behind it.  Written as C, hipcc hoists the 21
// compares of a softmax loop in front of their selects and keeps every mask in an SGPR pair -- with the selects of the
// observed state below that spilled 160 SGPRs in the residual-writing instantiations of k_hpass.
__device__ __forceinline__ float sel_eq(int x, int a, float if_ne, float if_eq) {
    float out;
    asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(out) : "s"(a), "v"(x), "v"(if_ne), "v"(if_eq) : "vcc");
    return out;
}
// the same compare serving two selects
__device__ __forceinline__ void sel_eq2(int x, int a, float ne0, float eq0, float ne1, float eq1, float &o0, float &o1) {
    asm("v_cmp_eq_u32 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %6, %7, vcc"
        : "=&v"(o0), "=&v"(o1) : "s"(a), "v"(x), "v"(ne0), "v"(eq0), "v"(ne1), "v"(eq1) : "vcc");
}
__global__ void k_sum_partials(const double *__restrict__ p, int n, double *out);
struct HpassArgs {
    const float4 *hj;
    const int8_t *msa_rm;
    const float *w;
    const double *h;      // fields of the local sites in f64 (the solver's copy: see k_hsolve); two buffers of hstride
    int hstride;          // doubles, the chain state says which one is current
    char *Rt;
    double *fx_part;
    float *hpart;         // [workgroup][16 sites][NH]  Hessian sums, NH = Q (Q + 1) / 2 (STATS == 2 only)
    double *gpart;        // [workgroup][16 sites][Q]   gradient sums, f64: their f32 accumulation was the noise floor
    float rscale;         //                            of the field solver (|g_h| ~ 1e-2 at N = 50 000)
    double *dpart;        // [workgroup][16 sites][Q]   diagonal second-order sums sum_s w P_a^2 (STATS == 2: the tiles without Hessian sums)
    const int *state;     // chain state of the field solver (PlmVpState, may be NULL) and which role this launch has in
    int cond;             // the chain (PLM_VP_*): a launch whose role is not wanted returns at once
    int sel;              // sequence tiles of this launch: 0 all, 1 the Hessian-sampled ones, 2 all the others
};
// does a launch with role `cond` run?  (wave-uniform: read through the scalar cache)
__device__ __forceinline__ bool vp_runs(const int *state, int cond) {
    if (!state || cond == PLM_VP_ALWAYS) return true;
    const PlmVpState *S = (const PlmVpState *)state;
    if (cond == PLM_VP_FINAL) return S->final_skip == 0;
    if (S->done) return false;
    return cond == PLM_VP_PASS_RT ? S->want_rt != 0 : S->want_rt == 0;
}
// STATS: 0 none, 1 gradient sums; a Hessian position of the chain is two launches: 2 = gradient + Hessian sums
// sum_s w P P^T on every PLM_HESS_SAMPLE-th sequence tile (118 KB of LDS statistics: one workgroup per CU), 3 = gradient +
// DIAGONAL second-order sums sum_s w P_a^2 on all the others (two workgroups per CU) -- k_hsolve then knows the diagonal
// exactly and rescales only the sampled off-diagonal part (round 6).  (One merged launch was measured: no gain, the sampled
// tiles are throughput, not a tail -- gpurun_out/r6f.)  XACT: exact softmax arguments
// The Hessian sums of a sampled tile are split over PLM_HESS_PARTS(Q) workgroups by ROWS of the upper triangle (each takes
// a third of the entries; all of them redo the tile's softmax, which is the cheap part): a sampled workgroup was 120 us of
// serial work on one CU -- the whole launch lasted as long as ONE of them, whatever the shard size (round 6: 0.12 ms of
// every Hessian position at 1 GPU and at 8).  Entry index of (a, b >= a): E(a) + b - a, E(a) = a Q - a (a - 1) / 2.
#define PLM_HESS_PARTS(Q) (((Q) == 20 || (Q) == 21) ? 3 : 1)
template <int Q> __host__ __device__ constexpr int hess_entry0(int a) { return a * Q - a * (a - 1) / 2; }
template <int Q> __host__ __device__ constexpr int hess_row_begin(int part) {      // first row of part `part` of PLM_HESS_PARTS(Q)
    if (part <= 0) return 0;
    if (part >= PLM_HESS_PARTS(Q)) return Q;
    int a = 0;
    while (a < Q && hess_entry0<Q>(a) * PLM_HESS_PARTS(Q) < part * (Q * (Q + 1) / 2)) a++;
    return a;
}
template <int Q> __host__ __device__ constexpr int hess_part_entries() {           // most entries any part holds
    int m = 0;
    for (int p = 0; p < PLM_HESS_PARTS(Q); p++) {
        const int n = hess_entry0<Q>(hess_row_begin<Q>(p + 1)) - hess_entry0<Q>(hess_row_begin<Q>(p));
        m = n > m ? n : m;
    }
    return m;
}
template <int Q, bool WRITE_RT, int STATS, bool XACT>
__global__ __launch_bounds__(512) void k_hpass(PlmDims d, HpassArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NH = (STATS == 2) ? Q * (Q + 1) / 2 : 0;
    constexpr int NSP = (STATS == 2) ? PLM_HESS_PARTS(Q) : 1;                 // row parts of a sampled tile's Hessian sums
    constexpr int NHP = (STATS == 2) ? hess_part_entries<Q>() : 0;            // LDS entries per wave and site
    if (!vp_runs(A.state, A.cond)) return;
    if (A.state) A.h += (size_t)((const PlmVpState *)A.state)->cur * A.hstride;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and known to be
    const int ns1 = (d.nstiles + PLM_HESS_SAMPLE - 1) / PLM_HESS_SAMPLE;
    const int sel = A.sel;
    const int part = (int)blockIdx.x % NSP, bid = (int)blockIdx.x / NSP;      // (NSP = 1 outside the sampled launch)
    int row_lo = 0, row_hi = Q;
    if constexpr (NSP > 1) {
        row_lo = part == 0 ? 0 : (part == 1 ? hess_row_begin<Q>(1) : hess_row_begin<Q>(2));
        row_hi = part == 0 ? hess_row_begin<Q>(1) : (part == 1 ? hess_row_begin<Q>(2) : Q);
    }
    const int ent_lo = row_lo * Q - row_lo * (row_lo - 1) / 2, ent_hi = row_hi * Q - row_hi * (row_hi - 1) / 2;
    const int nst = sel == 0 ? d.nstiles : (sel == 1 ? ns1 : d.nstiles - ns1);
    const int kt = bid % nst, b16l = bid / nst;
    // a statistics pass of the chain leaves quiet blocks alone (PlmVpState::quiet; wave-uniform, scalar cache)
    if (A.state && A.cond == PLM_VP_PASS && b16l < PLM_VP_MAXBLK && ((const PlmVpState *)A.state)->quiet[b16l]) return;
    const int stile = sel == 0 ? kt : (sel == 1 ? kt * PLM_HESS_SAMPLE : kt + kt / (PLM_HESS_SAMPLE - 1) + 1);
    // Alphabets above 21 states (the 32-state instantiation): the Hessian sums of eight waves do not fit the LDS -- two waves
    // of a sampled tile contribute them (k_hsolve scales accordingly), and the sampled tiles deliver their exact diagonal
    // sums like all the others
    constexpr int HW = Q > 21 ? 2 : 8;
    constexpr bool sampled = STATS == 2, DIAG = STATS == 3 || (STATS == 2 && Q > 21);
    const int blk = b16l * d.nstiles + stile;          // index of the (site block, sequence tile) pair everywhere
    const int b16 = d.b16_lo + b16l;
    const int r = lane & 15, g = lane >> 4;
    const int s_wave = stile * PLM_SEQ_TILE + wave * 32;
    const int gap = d.gap_mode;
    const int i = b16 * 16 + r;
    const bool site_ok = i < d.L;
    // The fields (+ the reference-state constants of the forward GEMM) as hi + lo f32 pairs of their f64 values.  A field
    // rounded to f32 shifts H of EVERY sequence by the same ~1e-7, a systematic error of the gradient sums that put a
    // floor of ~1e-2 under |g_h| at N = 50 000; with the low part added separately the rounding of (HJ + lo) + hi
    // differs from sequence to sequence and averages out.  The statistics-only instantiations read them once; the ones
    // that write residual fragments read them again for the second half (42 registers that would otherwise stay live
    // through the residual epilogue of the first half; the second read hits L2).
    float hv[XACT ? 1 : Q], hl[XACT ? 1 : Q];
    double hd[XACT ? Q : 1];                          // XACT: the fields as they are (f64)
    auto load_fields = [&](bool opaque) {
        u32 hoff = (u32)(site_ok ? i - d.h_site0 : 0) * Q;
        if (opaque) asm volatile("" : "+v"(hoff));   // per iteration: the loads must not be hoisted out of the loop
#pragma unroll
        for (int a = 0; a < Q; a++) {
            const double h64 = site_ok ? A.h[hoff + a] : 0.0;
            if constexpr (XACT) {
                hd[a] = h64;
            } else {
                hv[a] = (float)h64;
                hl[a] = (float)(h64 - (double)hv[a]);
            }
        }
    };
    // The statistics-only instantiations keep the hi / lo pairs in LDS instead ([site][state] float2, read per sequence):
    // 42 registers less puts them under 128 -- TWO workgroups per CU, whose load and compute phases overlap (round 6)
    constexpr bool LDSF = !WRITE_RT && !XACT && (STATS != 2 || Q > 21);     // (21 states: the sampled tiles are one workgroup per CU by LDS anyway)
    constexpr size_t STAT_BYTES = (size_t)16 * (8 * Q * sizeof(double) * (DIAG ? 2 : 1) + HW * NHP * sizeof(float));
    float2 *lf = (float2 *)(smem + STAT_BYTES);
    // FAST (round 6): the statistics passes of the plain evaluation are VALU-bound (~21 issue slots per (sequence, site,
    // state) = 0.2 of their 0.28-0.33 ms; persistent workgroups with prefetch were slower: NOTES_r06 2c e).  They need no
    // residual P - [x = a] per element: the sums take the unobserved states' exponentials with the weight w / Z folded into
    // one factor per sequence, and the OBSERVED state's term -w (1 - P_x) (resp. w P_x^2) goes into the LDS sums with one
    // add per sequence at a dynamic address -- the normalising multiply and the compare / select of every element are
    // gone, the exponent's argument is one fma, the padding mask is compiled in only where states can be dead.
    // Same precision argument as before: 1 - P_x is the sum of the other states' probabilities.
    constexpr bool FAST = !WRITE_RT && !XACT && (STATS == 1 || STATS == 3);
    if constexpr (LDSF) {
        for (int k = tid; k < 16 * Q; k += 512) {
            const int ii = b16 * 16 + k / Q;
            const double h64 = ii < d.L ? A.h[(size_t)(ii - d.h_site0) * Q + k % Q] : 0.0;
            const float hi = (float)h64;
            lf[k] = make_float2(hi, (float)(h64 - (double)hi));
        }
        __syncthreads();
    } else if constexpr (!WRITE_RT) load_fields(false);
    float fxl = 0.f;
    // statistics areas, one per wave: gradient sums [site][Q] in f64, then Hessian sums [site][NH] in f32.  Lanes add
    // with fire-and-forget LDS adds (the 4 lanes of a site collide on one address inside one instruction: resolved in
    // lane order; the two halves of the tile accumulate); waves never share an address and the areas are summed in
    // wave order at the end: bit-reproducible
    double *lg = (double *)smem + ((size_t)wave * 16 + r) * Q;
    double *ld = lg + (size_t)8 * 16 * Q;             // DIAG: a second f64 area behind the gradient sums
    float *lh0 = (float *)((double *)smem + (size_t)8 * 16 * Q * (DIAG ? 2 : 1));
    float *ls = lh0 + ((size_t)wave * 16 + r) * NHP;  // (waves below HW only; this part's entries, from ent_lo)
    if (STATS) {      // every wave clears what IT accumulates into: no barrier needed
        for (int k = lane; k < 16 * Q; k += 64) ((double *)smem)[(size_t)wave * 16 * Q + k] = 0.0;
        if constexpr (sampled)
            if (wave < HW)
                for (int k = lane; k < 16 * NHP; k += 64) lh0[(size_t)wave * 16 * NHP + k] = 0.f;
        if constexpr (DIAG)
            for (int k = lane; k < 16 * Q; k += 64) ((double *)smem)[(size_t)(8 + wave) * 16 * Q + k] = 0.0;
    }
    // wave-uniform base (SGPR pair) + one 32-bit per-lane byte offset: no 64-bit per-lane addresses
    const char *hj_u = (const char *)(A.hj + ((size_t)blk * 8 + wave) * 2 * Q * 64);
    const u32 lane16 = (u32)lane * 16;
    // the wave's 32 sequences are handled in two halves of 16 (4 per lane): half the registers of the whole tile
    int m_end = 2;
    asm volatile("" : "+s"(m_end));   // opaque trip count: the loop must not be unrolled (register pressure)
#pragma nounroll
    for (int m = 0; m < m_end; m++) {
        if constexpr (WRITE_RT) load_fields(true);
        f32x4 acc[Q];
#pragma unroll
        for (int a = 0; a < Q; a++) {
            const float4 v = *(const float4 *)(hj_u + (size_t)(m * Q + a) * 1024 + lane16);
            acc[a] = (f32x4){v.x, v.y, v.z, v.w};
        }
        float wk[4];     // weight of the lane's 4 (sequence, site) pairs (0: padding, gapped site in gap mode)
        float izk[4];    // FAST with diagonal sums: 1 / Z of the same pairs
        // After the softmax acc[a][reg] holds P(a) - [a = observed state]: the residual without its weight.  For the
        // observed state that is -(1 - P), formed as the SUM of the other states' probabilities: where a site is all but
        // certain (P = 1 - 1e-8: gap runs, conserved columns) the f32 P rounds to 1 - k 2^-24 and P - 1 loses every bit --
        // the same way for every such sequence, a coherent error the gradient sums add up (round 4: 0.8e-4 |x| at the
        // headline with everything else exact; -g, without the gap state's long runs: 0.2e-4).  The small probabilities
        // themselves are good to 1e-7 relative.
        int xk[4];       // observed state
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int s = s_wave + 16 * m + 4 * g + reg;
            const int xi = A.msa_rm[(size_t)s * d.Lp32 + i];
            const bool skip = gap && xi == 0;
            const float ws = (skip || !site_ok) ? 0.f : A.w[s];
            float mx = -INFINITY, Z = 0.f, hx = 0.f, zo = 0.f;     // zo: sum of the exponentials of the states NOT observed
            if constexpr (!XACT) {
                u32 foff = (u32)r * Q;
                if constexpr (LDSF) asm volatile("" : "+v"(foff));   // per sequence: the 21 pairs are read again, not kept
#pragma unroll
                for (int a = 0; a < Q; a++) {
                    float fh, fl;
                    if constexpr (LDSF) {
                        const float2 f = lf[foff + a];
                        fh = f.x;
                        fl = f.y;
                    } else {
                        fh = hv[a];
                        fl = hl[a];
                    }
                    // (states a >= Qc are padding of an alphabet that runs on the next instantiated size: impossible below the
                    // smallest alphabet of this instantiation -- at 21 states the test compiles away)
                    const float H = ((gap && a == 0) || (a >= plm_qc_min(Q) && a >= d.Qc)) ? -INFINITY : (acc[a][reg] + fl) + fh;
                    acc[a][reg] = H;
                    mx = fmaxf(mx, H);
                }
                if constexpr (FAST) {
                    const float mxl = -mx * 1.44269502f;            // exp(H - mx) = exp2(fma(H, log2 e, -mx log2 e))
#pragma unroll
                    for (int a = 0; a < Q; a++) {
                        const float ev = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[a][reg], 1.44269502f, mxl));
                        const float evo = sel_eq(xi, a, ev, 0.f);   // the exponentials of the states NOT observed
                        acc[a][reg] = evo;
                        Z += ev;
                        zo += evo;
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < Q; a++) {
                        const float H = acc[a][reg] - mx;
                        const float ev = __expf(H);
                        float evo;
                        sel_eq2(xi, a, hx, H, ev, 0.f, hx, evo);       // hx = H of the observed state; evo = ev of the others
                        acc[a][reg] = ev;
                        Z += ev;
                        zo += evo;
                    }
                }
            } else {
                // accurate evaluation: the argument of every exponential exactly.  H = potential + field in f64, the
                // reference mx any common value (the f32-rounded maximum), y = (H - mx) log2 e in f64 split into a float
                // and its remainder, exp = r + r y_lo ln 2 with r = v_exp_f32(y_hi): what is left is v_exp_f32's own
                // error, random with a mean of -0.06 ulp.  See exp_softmax_note above.
#pragma unroll
                for (int a = 0; a < Q; a++) {       // the reference: any value near the maximum serves
                    const float H = ((gap && a == 0) || a >= d.Qc) ? -INFINITY : acc[a][reg] + (float)hd[a];
                    mx = fmaxf(mx, H);
                }
                const double mxd = (double)mx;
#pragma unroll
                for (int a = 0; a < Q; a++) {
                    const double xs = ((double)acc[a][reg] + hd[a]) - mxd;
                    const double y = xs * 1.4426950408889634;
                    const float yh = (float)y, yl = (float)(y - (double)yh);
                    const float rr = __builtin_amdgcn_exp2f(yh);
                    const float ev = ((gap && a == 0) || a >= d.Qc) ? 0.f : __builtin_fmaf(rr, yl * 0.693147182464599609375f, rr);
                    float evo;
                    sel_eq2(xi, a, hx, (float)xs, ev, 0.f, hx, evo);
                    acc[a][reg] = ev;
                    Z += ev;
                    zo += evo;
                }
            }
            const float invZ = 1.f / Z;
            if (WRITE_RT && ws > 0.f) fxl -= ws * (hx - log_unbiased(Z));
            const float mpo = -(zo * invZ);                      // -(1 - P(observed))
            if constexpr (FAST) {
                // the observed state's terms, one LDS add each (a sequence that does not count -- padding, a gapped site in
                // gap mode -- has weight 0 and possibly no valid state: skipped)
                if (ws != 0.f) {
                    unsafeAtomicAdd(&lg[xi], (double)(ws * mpo));
                    if constexpr (STATS == 3) {
                        const float px = 1.f + mpo;
                        unsafeAtomicAdd(&ld[xi], (double)(ws * px * px));
                    }
                }
                wk[reg] = ws * invZ;                             // w / Z: acc holds the unnormalised exponentials of the other states
                izk[reg] = invZ;
                xk[reg] = xi;
            } else {
#pragma unroll
            for (int a = 0; a < Q; a++) acc[a][reg] = sel_eq(xi, a, acc[a][reg] * invZ, mpo);     // P - [a = observed]
            wk[reg] = ws;
            xk[reg] = xi;
            }
            asm volatile("" : "+v"(xk[reg]));   // no sharing of compare masks between the sections (SGPR spills)
            if constexpr (XACT) __builtin_amdgcn_sched_barrier(0);   // one sequence at a time: the f64 temporaries of four would spill
        }
        if constexpr (STATS != 0) {
            int idx = 0;
#pragma unroll
            for (int a = 0; a < Q; a++) {
                float t[4], ga = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    t[k] = wk[k] * acc[a][k];                             // w (P - [x = a])
                    ga += t[k];
                }
                // Q gradient sums: the 4 lanes of a site add straight into LDS (f64, one ds_add_f64, resolved in
                // lane order); the Hessian sums below are reduced in registers first
#ifndef PLM_EXP_NOGRAD
                if (NSP == 1 || part == 0) unsafeAtomicAdd(&lg[a], (double)ga);      // (one part of a sampled tile delivers them)
#else
                fxl += ga;
#endif
                if constexpr (DIAG) {
                    float da = 0.f;                                           // sum_k w P_a^2, P_a = acc + [x = a]
                    if constexpr (FAST) {
                        // t = (w / Z) e_a; P_a = e_a / Z = t / w ... the observed state's term went into the LDS sum above
#pragma unroll
                        for (int k = 0; k < 4; k++) da = fmaf(t[k] * acc[a][k], izk[k], da);
                    } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const bool obs = xk[k] == a;
                        da = fmaf(t[k] + (obs ? wk[k] : 0.f), acc[a][k] + (obs ? 1.f : 0.f), da);
                    }
                    }
#ifndef PLM_EXP_NOGRAD
                    unsafeAtomicAdd(&ld[a], (double)da);
#else
                    fxl += da;
#endif
                }
#ifndef PLM_EXP_NOHESS
                if (STATS == 2 && wave < HW && a >= row_lo && a < row_hi) {
                    // Hessian sums M_ab = sum_s w P_a P_b from every PLM_HESS_SAMPLE-th sequence tile only (scaled up
                    // by k_hsolve): the Newton iteration tolerates a few per cent of sampling error in H, the
                    // gradient sums above stay exact.  (Taking the diagonal M_aa from every tile buys nothing:
                    // H_aa = sum_b M_ab - M_aa + 2 lambda_h, it cancels.)
                    idx = a * Q - a * (a - 1) / 2 - ent_lo;
#pragma unroll
                    for (int k = 0; k < 4; k++) t[k] += (xk[k] == a) ? wk[k] : 0.f;     // w P(a)
#pragma unroll
                    for (int b = a; b < Q; b++) {
                        float v = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; k++) v = fmaf(t[k], acc[b][k] + ((xk[k] == b) ? 1.f : 0.f), v);
#ifndef PLM_EXP_NOSWAP
                        v = sum_over_g(v);
#endif
#ifndef PLM_EXP_NOLDS
                        if (g == 0) __builtin_amdgcn_ds_faddf(LDS_FPTR(&ls[idx]), v, 0, 0, false);
#else
                        fxl += v;
#endif
                        ++idx;
                    }
                }
#endif
            }
        }
        if constexpr (WRITE_RT) {
            // ---- residuals -> Rt: 24-bit fixed point, three signed digit planes, B fragments of the int8 backward
            // GEMM.  A fragment covers 64 sequences (two waves of this tile): this wave's half m is lane group
            // Gq = 2 (wave & 1) + m of it, and the lane's 4 sequences (4 g .. 4 g + 3 of the half) are dword g of the
            // 16-byte slot of lane (Gq, site r).
            __builtin_amdgcn_sched_barrier(0);   // nothing of the epilogue is to be hoisted into the statistics section
            const int s64 = stile * 4 + (wave >> 1);
            const int Gq = 2 * (wave & 1) + m;
            // wave-uniform base of (plane 0, this K half-step, state 0 of the site block) + the stride between planes;
            // per lane only a 32-bit offset: (state) * 2048 + slot * 16
            char *rt_u = A.Rt + rt_slot(d, 0, s64, b16l * Q, 0);
            const size_t rt_plane = (size_t)d.nst128 * d.nnfl * 2048;
            const u32 slot16 = (u32)(Gq * 16 + r) * 16;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                wk[reg] *= A.rscale;
                asm volatile("" : "+v"(xk[reg]));
            }
            // digit words of state a for the lane's 4 sequences
            const bool four = d.nplanes == 4;
            auto planes_of_state = [&](int a, u32 (&pl)[PLM_BWD_MAXPLANES]) {
                u32 x[4];
#pragma unroll
                for (int reg = 0; reg < 4; reg++)
                    x[reg] = digits_of(wk[reg] * acc[a][reg], four);
                planes_of4(x[0], x[1], x[2], x[3], pl);
            };
            // four states at a time: a 4 x 4 transpose over the lane groups g (2 + 2 row swaps per plane) leaves lane
            // group g with the whole 16-byte slot of state 4 q + g -> ONE 16-byte store per lane, plane and 4 states
#pragma unroll
            for (int q4 = 0; q4 + 4 <= Q; q4 += 4) {
                u32 pl[4][PLM_BWD_MAXPLANES];
#pragma unroll
                for (int k = 0; k < 4; k++) planes_of_state(q4 + k, pl[k]);
                const u32 off = (u32)(q4 + g) * 2048 + slot16;
#pragma unroll
                for (int p = 0; p < PLM_BWD_MAXPLANES; p++) {
                    if (p >= d.nplanes) break;
                    // X[k] = rows (k0 k1 k2 k3) over g.  permlane32_swap(X0, X2): upper half of X0 <-> lower half of X2
                    auto s02 = __builtin_amdgcn_permlane32_swap(pl[0][p], pl[2][p], false, false);   // (00 01 20 21) (02 03 22 23)
                    auto s13 = __builtin_amdgcn_permlane32_swap(pl[1][p], pl[3][p], false, false);   // (10 11 30 31) (12 13 32 33)
                    // permlane16_swap(A, B): odd rows of A <-> even rows of B
                    auto t0 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);        // (00 10 20 30) (01 11 21 31)
                    auto t1 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);        // (02 12 22 32) (03 13 23 33)
                    // row g now holds, for state q4 + g, the dwords of source rows 0, 1, 2, 3 = bytes 0-3, 4-7, 8-11, 12-15
                    *(uint4 *)(rt_u + p * rt_plane + off) = make_uint4(t0[0], t0[1], t1[0], t1[1]);
                }
                __builtin_amdgcn_sched_barrier(0);   // one group at a time (hipcc otherwise hoists every compare)
            }
            // the Q % 4 states left over: one dword per lane
#pragma unroll
            for (int a = Q - Q % 4; a < Q; a++) {
                u32 pl[PLM_BWD_MAXPLANES];
                planes_of_state(a, pl);
#pragma unroll
                for (int p = 0; p < PLM_BWD_MAXPLANES; p++)
                    if (p < d.nplanes) *(u32 *)(rt_u + p * rt_plane + (u32)a * 2048 + slot16 + 4 * g) = pl[p];
            }
        }
    }
    __syncthreads();
    if constexpr (STATS != 0) {
        if (NSP == 1 || part == 0) {
            double *gout = A.gpart + (size_t)blk * 16 * Q;
            for (int k = tid; k < 16 * Q; k += 512) {
                double v = 0;
#pragma unroll
                for (int wv = 0; wv < 8; wv++) v += ((const double *)smem)[(size_t)wv * 16 * Q + k];
                gout[k] = v;
            }
        }
        if constexpr (DIAG) {
            double *dout = A.dpart + (size_t)blk * 16 * Q;
            for (int k = tid; k < 16 * Q; k += 512) {
                double v = 0;
#pragma unroll
                for (int wv = 0; wv < 8; wv++) v += ((const double *)smem)[(size_t)(8 + wv) * 16 * Q + k];
                dout[k] = v;
            }
        }
        if constexpr (sampled) {
            float *out = A.hpart + (size_t)blk * 16 * NH;
            const int ne = ent_hi - ent_lo;                       // this part's entries of every site
            for (int k = tid; k < 16 * ne; k += 512) {
                const int site = k / ne, e = k - site * ne;
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < HW; wv++) v += lh0[((size_t)wv * 16 + site) * NHP + e];
                out[site * NH + ent_lo + e] = v;
            }
        }
    }
    if constexpr (WRITE_RT) {
        __syncthreads();
        const double tot = block_reduce_sum((double)fxl, (double *)smem);
        if (tid == 0) A.fx_part[blk] = tot;
    }
}
hipError_t plm_launch_hpass(const PlmDims &d, const float *hj, const int8_t *msa_rm, const float *w,
                            const double *h64, int write_rt, int stats, int exact, void *Rt, double *fx_part, float *hpart,
                            double *gpart, double *dpart, const int *state, int cond, hipStream_t st) {
    if (d.b16_hi <= d.b16_lo) return hipSuccess;
    const int nb = d.b16_hi - d.b16_lo, ns1 = (d.nstiles + PLM_HESS_SAMPLE - 1) / PLM_HESS_SAMPLE;
    const dim3 block(512);
    HpassArgs A{(const float4 *)hj, msa_rm, w, h64, (int)plm_h64_stride(d), (char *)Rt, fx_part,
                hpart, gpart, d.rscale, dpart, state, cond, 0};
#define HP_LAUNCH(QQ, WW, SS)                                                                          \
    {                                                                                                  \
        const dim3 grid((A.sel == 0 ? d.nstiles : (A.sel == 1 ? ns1 * ((SS) == 2 ? PLM_HESS_PARTS(QQ) : 1) : d.nstiles - ns1)) * nb); \
        const size_t lds = (size_t)16 * (8 * (QQ) * sizeof(double) * (((SS) == 3 || ((SS) == 2 && (QQ) > 21)) ? 2 : 1) + \
                                         ((SS) == 2 ? ((QQ) > 21 ? 2 : 8) * hess_part_entries<QQ>() : 0) * sizeof(float)) + \
                           (size_t)16 * (QQ) * sizeof(float2);      /* + the fields as hi / lo pairs */     \
        {                                                                                              \
            hipError_t e = exact ? plm_allow_lds<k_hpass<QQ, WW, SS, true>>(lds) : plm_allow_lds<k_hpass<QQ, WW, SS, false>>(lds); \
            if (e != hipSuccess) return e;                                                             \
        }                                                                                              \
        if (exact) hipLaunchKernelGGL((k_hpass<QQ, WW, SS, true>), grid, block, lds, st, d, A);        \
        else hipLaunchKernelGGL((k_hpass<QQ, WW, SS, false>), grid, block, lds, st, d, A);             \
    }
#define HP_CASE(QQ)                                                                                    \
    case QQ:                                                                                           \
        if (write_rt && stats == 2) return hipErrorInvalidValue;   /* never needed: residuals come last */ \
        else if (write_rt && stats == 0) HP_LAUNCH(QQ, true, 0)                                        \
        else if (write_rt) HP_LAUNCH(QQ, true, 1)                                                      \
        else if (stats == 2) {                                                                         \
            if (!dpart) return hipErrorInvalidValue;                                                   \
            /* the sampled tiles: one round of workgroups (a side stream for them, beside the pass over the others, was \
               measured in round 6: no gain -- gpurun_out/r6: fields 2.31 against 2.24 ms) */             \
            A.sel = 1;                                                                                 \
            HP_LAUNCH(QQ, false, 2)                                                                    \
            A.sel = 2;                                                                                 \
            if (d.nstiles > ns1) HP_LAUNCH(QQ, false, 3)                                               \
        } else HP_LAUNCH(QQ, false, 1)                                                                 \
        break;
    switch (d.Q) {
        HP_CASE(32)
        HP_CASE(21)
        HP_CASE(20)
        HP_CASE(5)
        HP_CASE(4)
    default:
        return hipErrorInvalidValue;
    }
#undef HP_CASE
#undef HP_LAUNCH
    return hipGetLastError();
}
size_t plm_hpart_bytes(const PlmDims &d) {
    return (size_t)d.nstiles * (d.b16_hi - d.b16_lo) * 16 * (d.Q * (d.Q + 1) / 2) * sizeof(float);
}
size_t plm_h64_stride(const PlmDims &d) {
    // per-site buffers are indexed by i - h_site0 over the LOCAL FIELD PART: the shard's own sites in sharded-state
    // mode, but all L sites in the replicated multi-shard mode (own_lo = 0, own_hi = nb16)
    const size_t nsites = (size_t)std::max(std::max(1, (d.b16_hi - d.b16_lo) * 16), std::min(d.L, d.own_hi * 16) - d.h_site0);
    return nsites * d.Q;
}
size_t plm_gpart_bytes(const PlmDims &d) { return (size_t)d.nstiles * (d.b16_hi - d.b16_lo) * 16 * d.Q * sizeof(double); }

// cnt[site][a] = sum_s w_s [x_si = a] over the sequences that count for the site (gap mode: the ungapped ones), f64, in a
// fixed summation order (bit-reproducible): the constant part of the exact first-order sums m_a = sum_s w_s P_s(a) =
// (gradient sum of the pass) + cnt_a that k_hsolve rescales the sampled Hessian with.  One workgroup per local site.
template <int Q>
__global__ __launch_bounds__(256) void k_site_counts(PlmDims d, const int8_t *__restrict__ msa_cm, const float *__restrict__ w,
                                                    double *__restrict__ cnt) {
    __shared__ double red[4];
    const int il = blockIdx.x, i = d.h_site0 + il;
    double acc[Q];
#pragma unroll
    for (int a = 0; a < Q; a++) acc[a] = 0.0;
    if (i < d.L)
        for (int s = threadIdx.x; s < d.N; s += 256) {
            const int x = msa_cm[(size_t)i * d.Np + s];
            const double ws = (d.gap_mode && x == 0) ? 0.0 : (double)w[s];
#pragma unroll
            for (int a = 0; a < Q; a++) acc[a] += (x == a) ? ws : 0.0;
        }
#pragma unroll
    for (int a = 0; a < Q; a++) {
        const double t = block_reduce_sum(acc[a], red);
        if (threadIdx.x == 0) cnt[(size_t)il * Q + a] = t;
        __syncthreads();
    }
}
hipError_t plm_launch_site_counts(const PlmDims &d, const int8_t *msa_cm, const float *w, double *cnt, hipStream_t st) {
    const int nsites = (d.b16_hi - d.b16_lo) * 16;
    if (nsites <= 0) return hipSuccess;
    switch (d.Q) {
    case 32: hipLaunchKernelGGL(k_site_counts<32>, dim3(nsites), dim3(256), 0, st, d, msa_cm, w, cnt); break;
    case 21: hipLaunchKernelGGL(k_site_counts<21>, dim3(nsites), dim3(256), 0, st, d, msa_cm, w, cnt); break;
    case 20: hipLaunchKernelGGL(k_site_counts<20>, dim3(nsites), dim3(256), 0, st, d, msa_cm, w, cnt); break;
    case 5: hipLaunchKernelGGL(k_site_counts<5>, dim3(nsites), dim3(256), 0, st, d, msa_cm, w, cnt); break;
    case 4: hipLaunchKernelGGL(k_site_counts<4>, dim3(nsites), dim3(256), 0, st, d, msa_cm, w, cnt); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Newton step on the fields of one site (one wave per site).  The workgroup partials of the last pass are summed
// in f64 (fixed order).  full = 1: the pass carried Hessian sums -- H = diag(rowsum M) - M + 2 lambda_h I is
// inverted (Gauss-Jordan in LDS, f64) and the inverse cached in hinv; otherwise the cached inverse is reused
// (simplified Newton: the Hessian moves slowly from one trial point to the next).  update = 0: only the squared
// gradient norm of the subproblem is recorded (verification of the point the residuals were computed at).
template <int Q>
__global__ __launch_bounds__(64) void k_hsolve(PlmDims d, const float *__restrict__ hpart,
                                              const double *__restrict__ gpart, int full,
                                              float *__restrict__ x, double *__restrict__ h64,
                                              double lambda_h, int update,
                                              double *__restrict__ hinv, double *__restrict__ g2_site,
                                              double tol_site2, const int *__restrict__ state, int chain, int hstride,
                                              const double *__restrict__ cnt, const double *__restrict__ dpart) {
    constexpr int NVF = PLM_HSTATS(Q);
    const PlmVpState *S = (const PlmVpState *)state;
    int cur = 0;
    if (S) {
        if (chain && S->done) return;
        // chain position with Hessian sums: they exist only if the pass ran in its statistics role (a pass that the
        // chain predicted to be the last wrote residual planes instead and steps with the cached inverses)
        if (full == 2) full = S->want_rt ? 0 : 1;
        cur = S->cur;
    } else if (full == 2) full = 1;
    __shared__ double st[NVF];
    __shared__ double Hm[Q][2 * Q + 1];             // [H | I] -> [I | H^-1] (odd row stride: no bank conflicts)
    __shared__ double Hi[Q][Q + 1];                 // the inverse the step is taken with
    __shared__ double gr[Q], Dv[Q], mv[Q], Md[Q];
    const int il = blockIdx.x, t = threadIdx.x;     // local site index
    const int i = d.h_site0 + il;
    if (i >= min(d.L, d.own_hi * 16)) {             // padding sites of the last block
        if (t == 0) g2_site[il] = 0.0;
        return;
    }
    const double *hc = h64 + (size_t)cur * hstride + (size_t)il * Q;                       // the fields the pass saw
    double *hn = h64 + (size_t)((S && chain) ? (cur ^ 1) : cur) * hstride + (size_t)il * Q; // where the step goes
    const int b16l = il >> 4, r = il & 15;
    // quiet block (PlmVpState::quiet): its sites no longer move.  A statistics pass skipped it -- the norm of its last
    // pass stands; a pass in the residual-writing role covered it -- fresh norm below, no step
    const bool quiet = S && chain && b16l < PLM_VP_MAXBLK && S->quiet[b16l];
    if (quiet) {
        if (update && t < Q) hn[t] = hc[t];
        if (!S->want_rt) return;
        update = 0;
        full = 0;
    }
    constexpr int NH = Q * (Q + 1) / 2;
    {
        // gradient sums: lane = sequence tile (mod 64), then the 64 lane sums per state are added in a fixed butterfly
        // -- 64 loads in flight per state instead of a chain of nstiles dependent ones
        // (round 6: two tiles per round, 2 Q loads in flight -- the loop was a chain of load latencies; same order of sums)
        double part[Q];
#pragma unroll
        for (int k = 0; k < Q; k++) part[k] = 0;
        for (int tt = t; tt < d.nstiles; tt += 128) {
            const bool two = tt + 64 < d.nstiles;
            const double *src0 = gpart + (((size_t)b16l * d.nstiles + tt) * 16 + r) * Q;
            const double *src1 = gpart + (((size_t)b16l * d.nstiles + (two ? tt + 64 : tt)) * 16 + r) * Q;
            double v0[Q], v1[Q];
#pragma unroll
            for (int k = 0; k < Q; k++) { v0[k] = src0[k]; v1[k] = src1[k]; }
#pragma unroll
            for (int k = 0; k < Q; k++) { part[k] += v0[k]; if (two) part[k] += v1[k]; }
        }
#pragma unroll
        for (int k = 0; k < Q; k++) {
            double v = part[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // same value in every lane
            if (t == 0) st[k] = v;
        }
    }
    if (full) {
        const int nsamp = (d.nstiles + PLM_HESS_SAMPLE - 1) / PLM_HESS_SAMPLE;
        for (int k = t; k < NH; k += 64) {      // Hessian sums exist for the sampled tiles only
            double v = 0;
            for (int j0 = 0; j0 < nsamp; j0 += 8) {      // eight loads in flight (round 6), summed in tile order as before
                float u[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int tt = min(j0 + j, nsamp - 1) * PLM_HESS_SAMPLE;
                    u[j] = hpart[(((size_t)b16l * d.nstiles + tt) * 16 + r) * NH + k];
                }
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (j0 + j < nsamp) v += (double)u[j];
            }
            st[Q + k] = v * ((double)d.nstiles / nsamp) * (Q > 21 ? 4.0 : 1.0);     // above 21 states 2 waves of 8 carry them
        }
        // exact diagonal second-order sums M_aa = sum_s w P_a^2 over ALL sequences (round 6): the tiles without Hessian
        // sums deliver them as f64 partials (k_hpass STATS = 3), the sampled tiles as the diagonal entries of their sums
        if (dpart) {      // lane = sequence tile (mod 64), as for the gradient sums above
            double part[Q];
#pragma unroll
            for (int k = 0; k < Q; k++) part[k] = 0;
            for (int tt0 = t; tt0 < d.nstiles; tt0 += 128) {      // two tiles per round, as for the gradient sums
                const bool two = tt0 + 64 < d.nstiles;
                double v[2][Q];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int tt = (h && two) ? tt0 + 64 : tt0;
                    const size_t blk = ((size_t)b16l * d.nstiles + tt) * 16 + r;
                    if (Q <= 21 && (tt % PLM_HESS_SAMPLE) == 0) {      // (above 21 states every tile delivers its diagonal sums)
#pragma unroll
                        for (int k = 0; k < Q; k++) v[h][k] = (double)hpart[blk * NH + (k * Q - k * (k - 1) / 2)];   // (k, k) of the upper triangle
                    } else {
#pragma unroll
                        for (int k = 0; k < Q; k++) v[h][k] = dpart[blk * Q + k];
                    }
                }
#pragma unroll
                for (int k = 0; k < Q; k++) { part[k] += v[0][k]; if (two) part[k] += v[1][k]; }
            }
#pragma unroll
            for (int k = 0; k < Q; k++) {
                double v = part[k];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (t == 0) Md[k] = v;
            }
        }
    }
    __syncthreads();
    const int a0 = d.gap_mode;                      // gap mode: state 0 is not a model state
    if (t < Q) gr[t] = (t < a0) ? 0.0 : st[t] + 2.0 * lambda_h * hc[t];
    __syncthreads();
    double *inv = hinv + (size_t)il * Q * Q;
    if (full) {
        for (int k = t; k < Q * Q; k += 64) {
            const int a = k / Q, b = k % Q, lo = min(a, b), hi = max(a, b);
            // index of (lo, hi) in the row-major upper triangle that follows the Q gradient sums
            Hm[a][b] = -st[Q + lo * Q - lo * (lo - 1) / 2 + (hi - lo)];
            Hm[a][Q + b] = (a == b) ? 1.0 : 0.0;
        }
        __syncthreads();
        // The second-order sums M~ come from every PLM_HESS_SAMPLE-th sequence tile (32nd since round 6); their row sums estimate the first-order sums
        // m_a = sum_s w P_s(a) -- which this pass knows EXACTLY (gradient sum + weighted count of the state).  A rare state
        // whose few sequences happen to sit in (or outside) the sampled tiles has its row of M~ off by up to 16x, and the
        // site then converges at 0.6 ... 0.94 per step (round 5 traces: tails of 5-9 passes; at config 3 ten times the
        // tolerance).  The sampled matrix is therefore rescaled symmetrically, X = D M~ D, until its row sums ARE m (three
        // Sinkhorn sweeps), and H = diag(rowsum X) - X + 2 lambda I: still a graph Laplacian (positive semidefinite, rows
        // summing to zero: the softmax gauge direction stays exact), with exact first-order content.  CPU lab
        // (tests/probes/field_solver_lab.py): contraction per step 0.03 -> 0.003.  (Scaling the rows to m without
        // restoring the zero row sums -- H = diag(m) - D M~ D -- breaks the cancellation between the two terms and is far
        // worse than no correction: measured, gpurun_out/r5c13.)
        // Round 6: with the DIAGONAL of sum_s w P P^T known exactly (Md) only the off-diagonal part of the sampled matrix
        // is rescaled, to the row sums m_a - M_aa it must have; the diagonal of H is then exact.  CPU lab
        // (tests/probes/field_solver_lab.py chain_dev_sink / chain_dev_xdiag / chain_dev_exact, N = 49 152, sampling 1/16):
        // 7.2 / 5.5 / 5.3 passes per evaluation -- nearly the exact Hessian's.
        const bool xdiag = cnt && dpart;
        if (t < Q) {
            mv[t] = cnt ? fmax(0.0, st[t] + cnt[(size_t)il * Q + t]) : 0.0;
            if (xdiag) {
                mv[t] = fmax(0.0, mv[t] - Md[t]);
                Hm[t][t] = 0.0;
            }
            Dv[t] = 1.0;
        }
        __syncthreads();
        if (cnt) {
            for (int sweep = 0; sweep < 3; sweep++) {
                double nd = 0.0;
                if (t < Q) {
                    double rs = 0;
                    for (int b = 0; b < Q; b++) rs -= Hm[t][b] * Dv[b];
                    rs *= Dv[t];
                    nd = rs > 0.0 ? Dv[t] * sqrt(mv[t] / rs) : 0.0;
                }
                __syncthreads();
                if (t < Q) Dv[t] = nd;
                __syncthreads();
            }
            for (int k = t; k < Q * Q; k += 64) Hm[k / Q][k % Q] *= Dv[k / Q] * Dv[k % Q];
            __syncthreads();
        }
        if (t < Q) {
            double rowsum = 0;
            for (int b = 0; b < Q; b++) rowsum -= Hm[t][b];   // sum_b X_ab (~ m_a)
            Hm[t][2 * Q] = rowsum;
        }
        __syncthreads();
        if (t < Q) Hm[t][t] += Hm[t][2 * Q] + 2.0 * lambda_h + 1e-12 * (1.0 + Hm[t][2 * Q]);
        __syncthreads();
        // Gauss-Jordan without pivoting (H is symmetric positive definite): lane -> one of the 2Q columns, held in
        // REGISTERS over all Q pivots (round 6; the LDS version read-modified-wrote its column once per row and pivot:
        // ~20 of the kernel's 34 us).  Row p of the own column is the lane's own register; the multipliers of a pivot
        // -- column p before it is eliminated -- are lane p's registers, broadcast with v_readlane.  Same operations
        // in the same order as before: the same inverse, bit for bit.
        {
            double col[Q];
            const int tc = min(t, 2 * Q - 1);
#pragma unroll
            for (int a = 0; a < Q; a++) col[a] = Hm[a][tc];
#pragma unroll
            for (int p = 0; p < Q; p++) {
                const double piv = 1.0 / lane_bcast(col[p], p);
                col[p] *= piv;
#pragma unroll
                for (int a = 0; a < Q; a++) {
                    if (a == p) continue;
                    const double f = lane_bcast(col[a], p);
                    col[a] -= f * col[p];
                }
            }
            if (t >= Q && t < 2 * Q) {
#pragma unroll
                for (int a = 0; a < Q; a++) Hm[a][t] = col[a];
            }
            __syncthreads();
        }
    }
    double g2 = 0;
    for (int a = 0; a < Q; a++) g2 += gr[a] * gr[a];       // every lane: the branches below must be uniform
    if (t == 0) g2_site[il] = g2;
    const bool move = update && g2 > tol_site2;
    if (!move && !full) {
        // nothing to do for this site; inside a chain its fields still have to exist in the other buffer
        if (S && chain && update && t < Q) hn[t] = hc[t];
        return;
    }
    // the inverse the step is taken with: fresh (sampled Hessian sums of this pass) or cached
    for (int k = t; k < Q * Q; k += 64) Hi[k / Q][k % Q] = full ? Hm[k / Q][Q + k % Q] : inv[k];
    __syncthreads();
    // (A per-site BFGS update of the inverse with the secant pair of the chain's previous step was built and measured
    // in round 5: near the f32 noise floor of the gradient sums y = g - g_prev is mostly noise, the inverses got
    // corrupted and the norm jumped between 3e-2 and 1e+1 from pass to pass -- gpurun_out/r5c4.  Removed.)
    const bool dirty = full != 0;
    if (dirty) for (int k = t; k < Q * Q; k += 64) inv[k] = Hi[k / Q][k % Q];
    if (!move) {
        if (S && chain && update && t < Q) hn[t] = hc[t];
        return;
    }
    // dh = H^-1 grad, one lane per state; cap far-away steps (a full Newton step can overshoot)
    double dh = 0;
    if (t < Q) {
        for (int b = 0; b < Q; b++) dh += Hi[t][b] * gr[b];
        if (t < a0) dh = 0;
    }
    // far from the optimum a full Newton step overshoots (saturated softmax: tiny Hessian entries): the step is
    // scaled so that no field moves by more than PLM_NEWTON_CAP.  Measured alternatives: a bound of 12 diverges (the
    // fit no longer converges); clamping every component on its own instead of scaling the step doubles the fit time;
    // an iterative-scaling step (h_a += log(count_a / model_a)) for the capped sites stalls config 2.
    double mxs = fabs(dh);
    for (int o = 32; o > 0; o >>= 1) mxs = fmax(mxs, __shfl_xor(mxs, o, 64));
    const double cap = (mxs > PLM_NEWTON_CAP) ? PLM_NEWTON_CAP / mxs : 1.0;
    if (t < Q) {
        const bool bad = mxs != mxs;
        // the solver iterates on an f64 copy of the fields; the parameter vector gets the rounded value
        const double hnew = bad ? hc[t] : hc[t] - cap * dh;
        hn[t] = hnew;
        if (!(S && chain)) x[(size_t)il * Q + t] = (float)hnew;
    }
}
__global__ __launch_bounds__(256) void k_fields_to_x(const double *__restrict__ h64, const int *__restrict__ state, int hstride,
                                                    float *__restrict__ x, int n) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int cur = state ? ((const PlmVpState *)state)->cur : 0;
    if (k < n) x[k] = (float)h64[(size_t)cur * hstride + k];
}
hipError_t plm_launch_fields_to_x(const PlmDims &d, const double *h64, const int *state, float *x, hipStream_t st) {
    const int n = (std::min(d.L, d.own_hi * 16) - d.h_site0) * d.Q;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fields_to_x, dim3((n + 255) / 256), dim3(256), 0, st, h64, state, (int)plm_h64_stride(d), x, n);
    return hipGetLastError();
}
// start of an evaluation: the solver's f64 fields <- the trial point's f32 fields
__global__ __launch_bounds__(256) void k_h64_init(const float *__restrict__ x, double *__restrict__ h64, int n) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) h64[k] = (double)x[k];
}
hipError_t plm_launch_h64_init(const PlmDims &d, const float *x, double *h64, hipStream_t st) {
    const int n = (std::min(d.L, d.own_hi * 16) - d.h_site0) * d.Q;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_h64_init, dim3((n + 255) / 256), dim3(256), 0, st, x, h64, n);
    return hipGetLastError();
}
// Bookkeeping of the field solver's chain after a pass + step (k_hpass, k_hsolve): g2 = sum of the per-site squared
// gradient norms of the pass just taken.  The chain is DONE when every site is within its share of the tolerance (then
// no site was moved by the k_hsolve in front of this kernel and the statistics -- and residual planes, if the pass wrote
// them -- of that pass are final), or when g2 is not a number (the line search deals with that); later launches of the
// chain return at once.  Otherwise it predicts from the contraction seen so far whether the NEXT pass will be the last
// one, i.e. should write the residual planes (1.0 GB of stores that only the last pass has to make): the host's rule of
// rounds 2-4 (`gh2 * rate <= tol2`) evaluated where the numbers are, one pass earlier than a host round trip could.
__global__ __launch_bounds__(256) void k_vp_check(const double *__restrict__ g2_site, int n, double *g2_out,
                                                 double tol_site2, double tol2, double floor2, int *state) {
    __shared__ double red[4];
    __shared__ int bad[4];
    PlmVpState *S = (PlmVpState *)state;
    if (S && S->done) {
        // (a host continuation of a sharded chain zeroes the verdict slots before it re-enqueues: a rank whose chain had
        // ended must publish them again, or the all-reduced verdict could never be "every rank done" -- ADVICE r5)
        if (threadIdx.x == 0) {
            g2_out[1] = (double)S->passes;
            g2_out[2] = 1.0;
        }
        return;
    }
    double s = 0;
    int open_sites = 0;
    __shared__ int loud_blocks;
    if (threadIdx.x == 0) loud_blocks = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = g2_site[i];
        s += v;
        open_sites += (v > tol_site2) ? 1 : 0;
    }
    // quiet blocks: within their share of a quarter of the tolerance -> they stop moving, later statistics passes
    // skip them (sticky for the rest of the chain; the flags only matter if the chain goes on)
    const int nblk = n / 16;
    if (S && nblk <= PLM_VP_MAXBLK && tol2 > 0.0)
        for (int b = threadIdx.x; b < nblk; b += 256) {
            double sb = 0;
            for (int k = 0; k < 16; k++) sb += g2_site[b * 16 + k];
            if (!S->quiet[b] && sb <= 0.25 * tol2 / nblk) S->quiet[b] = 1;
            if (!S->quiet[b]) atomicAdd(&loud_blocks, 1);
        }
    for (int o = 32; o > 0; o >>= 1) open_sites += __shfl_down(open_sites, o, 64);
    if ((threadIdx.x & 63) == 0) bad[threadIdx.x >> 6] = open_sites;
    const double t = block_reduce_sum(s, red);     // contains the barrier that publishes bad[]
    if (threadIdx.x == 0) {
        g2_out[0] = t;
        if (!S) return;
        // done: the solver's tolerance is on the NORM over all sites (what the stop rule of the fit adds to |g|^2); the
        // per-site shares only decide which sites still move.  (Waiting for every site to meet its share makes the whole
        // chain wait for the slowest one: measured, 9-14 passes per evaluation with the total long inside the tolerance.)
        const int open_total = bad[0] + bad[1] + bad[2] + bad[3];
        // ... or the solver has arrived at the noise floor of its f32 gradient sums, where an evaluation would only burn
        // passes: below the host's estimate of the floor (floor2) a single pass that no longer gains a factor 2 in norm
        // ends the chain; within a factor 10 of the tolerance two passes that together gain less than that do (the
        // estimate is only an estimate: at N = 100 000 the plain passes stall above it, and 95 of 271 chains of a config-3
        // fit ran out of positions until this rule existed; a site whose sampled Hessian is poor converges at 0.6 per
        // pass, 0.36 over two, and goes on).  Far from the tolerance the norm may stand still for a few capped steps:
        // no stall rule there.
        const bool near_tol = t <= 100.0 * tol2;
        const bool stall1 = S->passes >= 1 && t > 0.25 * S->g2_prev, stall2 = S->passes >= 2 && t > 0.25 * S->g2_prev2;
        const bool at_floor = (stall1 && t <= floor2) || (near_tol && stall2);
        const bool done = open_total == 0 || !(t > tol2) || at_floor || t != t;
        if (S->passes < PLM_VP_HIST) S->hist_loud[S->passes] = loud_blocks;
        if (S->passes < PLM_VP_HIST) S->hist[S->passes] = t + 1e-30 * 0 + (double)open_total * 1e9;   // debug trace: norm^2 (+ open sites * 1e9)
        S->passes += 1;
        g2_out[1] = (double)S->passes;    // passes this chain needed (the host resets the state per evaluation)
        g2_out[2] = done ? 1.0 : 0.0;     // the host reads the verdict with the scalars (sharded: summed over ranks)
        if (done) {
            S->done = 1;                            // the step k_hsolve just wrote to the other buffer is dropped
            if (S->want_rt) S->final_skip = 1;      // this pass wrote the residual planes: no separate last pass
        } else {
            S->cur ^= 1;                            // the chain goes on: the step becomes the current fields
            // contraction of the squared norm per pass: measured once two passes exist, before that the typical
            // simplified-Newton rate with sampled Hessians (~0.03 per step in norm)
            const double rate = (S->passes >= 2 && S->g2_prev > 0) ? fmin(1.0, t / S->g2_prev) : 1e-3;
            // ... or a first pass without gain next to the tolerance: the next one ends the chain either way
            S->want_rt = (t * fmax(1e-6, rate) <= tol2 || (near_tol && stall1)) ? 1 : 0;
        }
        S->g2_prev2 = S->g2_prev;
        S->g2_prev = t;
    }
}
__global__ void k_vp_reset(int *state, int want_rt, double *zero3) {
    PlmVpState *S = (PlmVpState *)state;
    if (zero3) zero3[0] = zero3[1] = zero3[2] = 0.0;      // the chain's scalar slots (norm, passes, verdict)
    S->done = 0; S->want_rt = want_rt; S->final_skip = 0; S->passes = 0; S->cur = 0; S->g2_prev = 0.0; S->g2_prev2 = 0.0;
    for (int k = 0; k < PLM_VP_HIST; k++) S->hist[k] = 0.0;
    for (int k = 0; k < PLM_VP_MAXBLK; k++) S->quiet[k] = 0;
}
hipError_t plm_launch_vp_reset(int *state, int want_rt, double *zero3, hipStream_t st) {
    hipLaunchKernelGGL(k_vp_reset, dim3(1), dim3(1), 0, st, state, want_rt, zero3);
    return hipGetLastError();
}
hipError_t plm_launch_hsolve(const PlmDims &d, const float *hpart, const double *gpart, int full, float *x, double *h64,
                             double lambda_h, int update, double *hinv, double *g2_site, double *g2_out, double tol2,
                             double floor2, int *state, int chain, const double *cnt, const double *dpart, hipStream_t st) {
    const int nsites = (d.b16_hi - d.b16_lo) * 16;
    int *cstate = chain ? state : nullptr;      // the chain's bookkeeping runs for chain positions only
    if (nsites <= 0) {      // a shard without sites: its chain is done at once
        hipError_t e = hipMemsetAsync(g2_out, 0, sizeof(double), st);
        if (e == hipSuccess && cstate) {
            hipLaunchKernelGGL(k_vp_check, dim3(1), dim3(256), 0, st, g2_site, 0, g2_out, 0.0, tol2, floor2, cstate);
            e = hipGetLastError();
        }
        return e;
    }
    // Every site steps at every position (tol_site2 = 0).  Rounds 2-4 left a site alone once it was within its share of
    // the tolerance; with the chain's end decided on the norm over all sites that put a floor of about the tolerance itself
    // under the norm (the frozen sites' shares) -- the step is tentative instead (k_vp_check commits it).
    const double tol_site2 = 0.0;
    const int hs = (int)plm_h64_stride(d);
    switch (d.Q) {
    case 32: hipLaunchKernelGGL(k_hsolve<32>, dim3(nsites), dim3(64), 0, st, d, hpart, gpart, full, x, h64, lambda_h, update, hinv, g2_site, tol_site2, state, chain, hs, cnt, dpart); break;
    case 21: hipLaunchKernelGGL(k_hsolve<21>, dim3(nsites), dim3(64), 0, st, d, hpart, gpart, full, x, h64, lambda_h, update, hinv, g2_site, tol_site2, state, chain, hs, cnt, dpart); break;
    case 20: hipLaunchKernelGGL(k_hsolve<20>, dim3(nsites), dim3(64), 0, st, d, hpart, gpart, full, x, h64, lambda_h, update, hinv, g2_site, tol_site2, state, chain, hs, cnt, dpart); break;
    case 5: hipLaunchKernelGGL(k_hsolve<5>, dim3(nsites), dim3(64), 0, st, d, hpart, gpart, full, x, h64, lambda_h, update, hinv, g2_site, tol_site2, state, chain, hs, cnt, dpart); break;
    case 4: hipLaunchKernelGGL(k_hsolve<4>, dim3(nsites), dim3(64), 0, st, d, hpart, gpart, full, x, h64, lambda_h, update, hinv, g2_site, tol_site2, state, chain, hs, cnt, dpart); break;
    default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_vp_check, dim3(1), dim3(256), 0, st, g2_site, nsites, g2_out, tol_site2, tol2, floor2, cstate);
    return hipGetLastError();
}

// out[s] = (E, E_J, E_h) in f64 from the per-block partials: E_J = 1/2 sum_i HJ[s,i,x_si] (every pair is seen
// from both of its sites), E_h = sum_i h_i(x_si)
__global__ __launch_bounds__(256) void k_energy_sum(const float2 *__restrict__ part, int n, int nblk,
                                                   double *__restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    double ej = 0, eh = 0;
    for (int b = 0; b < nblk; b++) {
        const float2 v = part[(size_t)s * nblk + b];
        ej += v.x;
        eh += v.y;
    }
    ej *= 0.5;
    out[(size_t)s * 3 + 0] = ej + eh;
    out[(size_t)s * 3 + 1] = ej;
    out[(size_t)s * 3 + 2] = eh;
}
hipError_t plm_launch_energy_sum(const PlmDims &d, const float *part, double *out, hipStream_t st) {
    hipLaunchKernelGGL(k_energy_sum, dim3((d.N + 255) / 256), dim3(256), 0, st, (const float2 *)part, d.N,
                       (d.b16_hi - d.b16_lo) * plm_fwd_groups(d.Q, 0), out);
    return hipGetLastError();
}

// =========================================================================================
// K_bwd: asymmetric gradient slab  G[(j,b),(i,a)] = sum_s [x_sj = b] * R_s(i,a)   (row a6 backward half)
//   on the int8 matrix cores, EXACT: the residuals arrive in 24-bit fixed point as three signed base-256 digit planes
//   (k_hpass / k_onehot_rt), a workgroup contracts ONE plane over its K range with v_mfma_i32_16x16x64_i8 into int32
//   accumulators, the consumers combine G0 + 256 G1 + 65536 G2.  GEMM over K = sequences; A = one-hot {0,1} bytes of
//   the column-major alignment expanded in registers (16 sequences per lane and instruction), B = digit fragments
//   (Rt) streamed through LDS.  Why this shape:
//     * the int8 instruction does twice the K of the f16 one in the same cycles; three digit planes replace the
//       f16 hi + lo planes of rounds 1-2: 3/4 of the matrix-core cycles, of the LDS reads and of the L2->LDS stream;
//     * one plane per workgroup keeps the 7 x 7 fragment register tile whole (three accumulator sets would not fit)
//       and takes the place of most of the split-K factor (3 x the workgroups of a launch for free);
//     * integer accumulation has no rounding: the sum is independent of tiling, K split and order (bit-reproducible
//       by construction) and the error of the gradient is the quantisation of the residuals alone (2^-24 of the
//       largest weight per term, unbiased) -- the f32 accumulation of rounds 1-2 left a systematic offset that grew
//       with N (4e-4 |x| at N = 50 000, 1e-3 |x| at N = 100 000).
//   A K step is 128 sequences = two 64-sequence halves (two fragments per column fragment in the tile, at the byte
//   offsets the f16 hi / lo planes used to have); workgroup = 8 waves as 4 (rows) x 2 (cols), wave tile FM x FN
//   accumulator fragments; |sum| <= 128 * 127 * K < 2^31 for K ranges below 132 000 sequences (make_dims splits K
//   accordingly).  XCD-aware block order: the
//   row tiles that share a (column tile, plane, K range) panel of Rt run on one XCD (block b runs on XCD b % 8).
// =========================================================================================
// one-hot of 16 packed states (4 dwords) against state b (replicated into every byte of bb): 16 int8 values, -128
// where the state matches and 0 elsewhere -- two VALU operations per dword (the expansion shares the issue port with
// the MFMAs: with {0, 1} values, four operations per dword, the kernel ran 5.0 ms).  States are < 128, so
// (x ^ b) + 0x7f sets bit 7 of a byte exactly when it DIFFERS from b, with no carries; v_bfi keeps the complement of
// that bit.  The factor -128 is part of gscale; |sum| <= 128 * 127 * K < 2^31 for K ranges below 132 000 sequences.
__device__ __forceinline__ i32x4 onehot16(const i32x4 &x, u32 bb, u32 k7f) {
    // hipcc rewrites the plain C of this as xor / sub / and (three operations); spelled out: v_xad_u32 = (x ^ b) + 0x7f..
    // (bb in an SGPR, the constant in a VGPR: one constant-bus operand per instruction), v_bfi_b32 = ~y & 0x80..
    i32x4 r;
    const u32 k80 = 0x80808080u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u32 y, v;
        asm("v_xad_u32 %0, %1, %2, %3" : "=v"(y) : "v"((u32)x[j]), "s"(bb), "v"(k7f));
        asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(v) : "v"(y), "s"(k80));
        r[j] = (int)v;
    }
    return r;
}
// fragment T of a K step (T = half * FN + column): the fragment T + 1 is in flight while the FM MFMAs of T run.
template <int FM, int FN, int T>
__device__ __forceinline__ void bwd_frag(i32x4 (&acc)[FM][FN], const i32x4 (&af)[FM], u32 lb, i32x4 (&bf)[2],
                                         const DmaPlan &dma) {
    constexpr int C = T % FN, NP = (4 * FN + 7) / 8;
    constexpr int me = T & 1, nx = (T + 1) & 1;
    if constexpr (T + 1 < 2 * FN) {
        constexpr int C1 = (T + 1) % FN, H1 = (T + 1) / FN;
        bf[nx] = lds_read_b128_i<C1 * 2048 + H1 * 1024>(lb);
        lds_wait_i<1>(bf[me]);
    } else {
        lds_wait_i<0>(bf[me]);
    }
#pragma unroll
    for (int f = 0; f < FM; f++) acc[f][C] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[f], bf[me], acc[f][C], 0, 0, 0);
    dma_at<2 * FN, T, NP, PLM_DMA_STAGGER_BWD>(dma);
}
template <int FM, int FN, int H, int... C>
__device__ __forceinline__ void bwd_half(i32x4 (&acc)[FM][FN], const i32x4 (&af)[FM], u32 lb, i32x4 (&bf)[2],
                                         const DmaPlan &dma, std::integer_sequence<int, C...>) {
    (bwd_frag<FM, FN, H * FN + C>(acc, af, lb, bf, dma), ...);
}

template <int Q, int FM, int FN>
__global__ __launch_bounds__(512) void k_bwd(PlmDims d, const int8_t *__restrict__ msa_cm,
                                            const char *__restrict__ Rt, int *__restrict__ G,
                                            const int *__restrict__ run) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // variable-projection fit: the launch is enqueued behind the field solver's chain and only does its work when
    // that chain has converged (device flag); otherwise the host finishes the fields and launches it again
    if (run && !*run) return;
    constexpr int TILE = 2 * FN * 2 * 1024;  // 2*FN column fragments x 2 halves of the K step
    constexpr int NP = (4 * FN + 7) / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // the same number, known to be wave-uniform
    const int wm = wave_s >> 1, wn = wave_s & 1;               // wave-uniform: the states of the row fragments go to SGPRs
    u32 k7f = 0x7f7f7f7fu;
    asm volatile("" : "+v"(k7f));                              // a VGPR constant (see onehot16)
    const int ngroups = d.ncol_tiles * d.nplanes * d.ksplit;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = (slot / d.nrow_tiles) * 8 + xcd;
    if (grp >= ngroups) return;
    const int row_tile = slot % d.nrow_tiles;
    const int col_tile = grp % d.ncol_tiles, pk = grp / d.ncol_tiles;
    const int plane = pk % d.nplanes, ks = pk / d.nplanes;
    const int per = (d.nst128 + d.ksplit - 1) / d.ksplit;
    const int k0 = ks * per, k1 = min(d.nst128, k0 + per);

    const int mf0 = (row_tile * 4 + wm) * FM;
    const bool row_ok = mf0 < d.nmf;
    const int j16 = row_ok ? mf0 / Q : 0, b0 = row_ok ? mf0 % Q : 0;
    const int nfl0 = col_tile * 2 * FN;
    const int r = lane & 15, g = lane >> 4;
    // A operand: byte offset of this lane's 16 sequences of its site row in msa_cm (< 2^31: (nb16 + 1) * 16 * Np bytes)
    const u32 acol = (u32)(j16 * 16 + r) * (u32)d.Np + 16 * g;

    i32x4 acc[FM][FN];
#pragma unroll
    for (int f = 0; f < FM; f++)
#pragma unroll
        for (int c = 0; c < FN; c++) acc[f][c] = (i32x4){0, 0, 0, 0};

#if PLM_PROBE
    unsigned long long pr_wait = 0;
    const unsigned long long pr_t0 = PROBE_NOW();
#endif
    // tile of step ss: 2*FN column fragments x 2 halves, contiguous in the plane; fragments past nnfl are not copied.
    // Behind it in each LDS buffer: the alignment bytes of the 4 row groups, [row group][half] x 1 KB (lane (g, r) =
    // 16 sequences of site row r): wave (wm, wn) brings in (wm, half wn) with one gathered piece, both waves of the row
    // group read both halves -- no per-lane prefetch registers, and the two column waves share one fetch.
    constexpr int ABYTES = 4 * 2 * 1024, BUF = TILE + ABYTES;
    const int np_valid = min(4 * FN, 2 * (d.nnfl - nfl0));
    const char *rt0 = Rt + ((size_t)plane * d.nst128 * d.nnfl + nfl0) * 2048;
    const size_t rt_step = (size_t)d.nnfl * 2048;
    const char *a_src = row_ok ? (const char *)msa_cm : nullptr;
    const u32 a_off0 = acol + 64 * (u32)wn;
    const int a_slot = TILE + ((wave_s >> 1) * 2 + (wave_s & 1)) * 1024;   // wave-uniform (it ends up in M0)
    if (k0 < k1) {
        const DmaPlan first{rt0 + (size_t)k0 * rt_step, smem, wave_s, np_valid, (u32)lane * 16, false,
                            a_src, a_off0 + PLM_BWD_KSTEP * (u32)k0, smem + a_slot};
        dma_issue_all<NP>(first);
    }
    i32x4 bf[2];
    const u32 lw = lds_addr(smem + (wn * FN) * 2048 + lane * 16);   // this wave's columns in buffer 0
    const u32 la = lds_addr(smem + TILE + (wm * 2) * 1024 + lane * 16);   // this row group's alignment bytes, half 0
    // the same register serves as the per-lane part of the LDS-DMA source address: lw = lane * 16 + lw_base
    const u32 lw_base = __builtin_amdgcn_readfirstlane(lw - (u32)lane * 16);
    int cur = 0;
    for (int ss = k0; ss < k1; ++ss) {
#if PLM_PROBE
        const unsigned long long pa = PROBE_NOW();
#endif
        vm_wait<0>();
        __syncthreads();
#if PLM_PROBE
        pr_wait += PROBE_NOW() - pa;
#endif
        const int nxt = cur ^ 1;
        const bool more = ss + 1 < k1;
        const DmaPlan dma{rt0 + (size_t)(ss + 1) * rt_step - lw_base, smem + nxt * BUF, wave_s, more ? np_valid : 0, lw,
                          wave_s >= 4, more ? a_src : nullptr, a_off0 + PLM_BWD_KSTEP * (u32)(ss + 1),
                          smem + nxt * BUF + a_slot};
        if (row_ok) {
            const u32 lb = lw + cur * BUF, lab = la + cur * BUF;
            i32x4 af[FM];
            i32x4 xa = lds_read_b128_i<0>(lab);
            bf[0] = lds_read_b128_i<0>(lb);
            lds_wait_i<1>(xa);
#pragma unroll
            for (int f = 0; f < FM; f++) af[f] = onehot16(xa, (u32)(b0 + f) * 0x01010101u, k7f);
            bwd_half<FM, FN, 0>(acc, af, lb, bf, dma, std::make_integer_sequence<int, FN>{});
            __builtin_amdgcn_sched_barrier(0);   // the second half's fragments take over the registers of the first's
            xa = lds_read_b128_i<1024>(lab);
            lds_wait_i<0>(xa);
#pragma unroll
            for (int f = 0; f < FM; f++) af[f] = onehot16(xa, (u32)(b0 + f) * 0x01010101u, k7f);
            bwd_half<FM, FN, 1>(acc, af, lb, bf, dma, std::make_integer_sequence<int, FN>{});
        } else {            // idle row waves still take part in the barrier and copy their share
            dma_issue_all<NP>(dma);
        }
        cur = nxt;
    }
#if PLM_PROBE
    const unsigned long long pr_t1 = PROBE_NOW();
#endif
    if (!row_ok) return;
    int *Gp = G + (size_t)(plane * d.ksplit + ks) * d.nmf * d.nnfl * 256;
#pragma unroll
    for (int f = 0; f < FM; f++)
#pragma unroll
        for (int c = 0; c < FN; c++) {
            const int nfl = nfl0 + wn * FN + c;
            if (nfl < d.nnfl) *(i32x4 *)(Gp + (((size_t)(mf0 + f) * d.nnfl + nfl) * 64 + lane) * 4) = acc[f][c];
        }
#if PLM_PROBE
    if (lane == 0) {
        const unsigned long long pr_t2 = PROBE_NOW();
        atomicAdd(&plm_probe_acc[1][0], pr_wait); atomicAdd(&plm_probe_acc[1][3], pr_t2 - pr_t0);
        atomicAdd(&plm_probe_acc[1][4], pr_t2 - pr_t1); atomicAdd(&plm_probe_acc[1][5], 1ull);
    }
#endif
}

// ---- k_bwd_w: the same GEMM with the accumulator tile in a[0:251] and the K step in assembly --------------------
// Workgroup = 4 waves (one per SIMD, 7 row fragments each), wave tile 7 x 9 accumulator fragments = 252 AccVGPRs;
// the whole K step -- 126 MFMAs, the LDS reads two fragments ahead, the one-hot expansion of the NEXT half step
// interleaved into the MFMA stream (2 VALU between MFMAs, hidden behind their 16 cycles), the LDS-DMA copies of the
// step three ahead, the arrival / check protocol that replaces the barrier -- is one asm block from plm_bwd_asm.inc
// (scripts/gen_bwd_asm.py has the register map, the schedule and the reasons).  A wave issues 63 MFMAs per 56 VALU
// operations of expansion (k_bwd: 49), and the expansion no longer sits in front of the MFMAs of its own half step.
// Digit tiles and alignment bytes live in rings of four LDS slots.  Results are bit-identical to k_bwd's (exact
// integer sums).
#ifndef PLM_BWDW_INC
#define PLM_BWDW_INC "plm_bwd_asm.inc"
#endif
#include PLM_BWDW_INC
template <int IDX> __device__ __forceinline__ i32x4 bwdw_acc_read() {
    i32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\t"
                 "v_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "n"(IDX), "n"(IDX + 1), "n"(IDX + 2), "n"(IDX + 3));
    return v;
}
template <int F, int C>
__device__ __forceinline__ void bwdw_store(int *Gp, const PlmDims &d, int mf0, int nfl0, int lane) {
    const int nfl = nfl0 + C;
    const i32x4 v = bwdw_acc_read<(F * PLM_BWDW_FN + C) * 4>();
    if (nfl < d.nnfl) *(i32x4 *)(Gp + (((size_t)(mf0 + F) * d.nnfl + nfl) * 64 + lane) * 4) = v;
}
template <int F, int... C>
__device__ __forceinline__ void bwdw_store_row(int *Gp, const PlmDims &d, int mf0, int nfl0, int lane,
                                               std::integer_sequence<int, C...>) {
    (bwdw_store<F, C>(Gp, d, mf0, nfl0, lane), ...);
}
template <int... F>
__device__ __forceinline__ void bwdw_store_all(int *Gp, const PlmDims &d, int mf0, int nfl0, int lane,
                                               std::integer_sequence<int, F...>) {
    (bwdw_store_row<F>(Gp, d, mf0, nfl0, lane, std::make_integer_sequence<int, PLM_BWDW_FN>{}), ...);
}

template <int Q>
__global__ __launch_bounds__(256) void k_bwd_w(PlmDims d, const int8_t *__restrict__ msa_cm,
                                              const char *__restrict__ Rt, int *__restrict__ G,
                                              const int *__restrict__ run) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (run && !*run) return;
    constexpr int FM = PLM_BWDW_FM, FNW = PLM_BWDW_FN;
    static_assert(Q % FM == 0, "the row fragments of a wave are states of one site block");
    static_assert(FNW == PLM_BWDW_COLS, "plm_internal.h and plm_bwd_asm.inc disagree");
    constexpr int TILE = FNW * 2 * 1024, ASLOT = 4 * 2 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32 k7f = 0x7f7f7f7fu;
    asm volatile("" : "+v"(k7f));
    const u32 k80 = 0x80808080u;
    const int ngroups = d.ncol_tiles * d.nplanes * d.ksplit;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = (slot / d.nrow_tiles) * 8 + xcd;
    if (grp >= ngroups) return;
    const int row_tile = slot % d.nrow_tiles;
    const int col_tile = grp % d.ncol_tiles, pk = grp / d.ncol_tiles;
    const int plane = pk % d.nplanes, ks = pk / d.nplanes;
    const int per = (d.nst128 + d.ksplit - 1) / d.ksplit;
    const int k0 = ks * per, k1 = min(d.nst128, k0 + per);
    if (k0 >= k1) return;

    // waves past the last row fragment run on row 0 (the K loop has no branches) and store nothing
    const int mf0 = (row_tile * 4 + wm) * FM;
    const bool row_ok = mf0 < d.nmf;
    const int j16 = row_ok ? mf0 / Q : 0, b0 = row_ok ? mf0 % Q : 0;
    const u32 b0x = (u32)b0 * 0x01010101u;
    const int nfl0 = col_tile * FNW;
    const int r = lane & 15, g = lane >> 4;
    const u32 acol = (u32)(j16 * 16 + r) * (u32)d.Np + 16 * g;

    // LDS: a ring of four digit tiles (FNW column fragments x 2 halves each), then a ring of four slots of alignment
    // bytes ([row group][half] x 1 KB; a wave copies and reads its own row group only).  Step s computes on slot
    // s % 4 and issues the copies of step s + 3 in its second half.  A wave ends a step with vmcnt(7) -- its copies
    // issued BEFORE this step have landed -- and an arrival on an LDS counter; in the middle of the next step it checks
    // that all four waves have arrived (scripts/gen_bwd_asm.py): then the tile of step s + 2 is complete (its first
    // fragments are read at the end of step s + 1: no LDS latency in front of a step) and the slot of step s may be
    // overwritten.  There is no s_barrier in the loop.
    // A wave copies pieces wm, wm + 4, ... of a tile: 5 copies, the last one repeating a piece for the waves that own
    // four; a tile past the last column fragment reads on into the next rows (Rt has that much slack behind it), a
    // step past the K range reads the last step again: every wave issues exactly PLM_BWDW_NVMEM copies per step.
    char *aring = smem + 4 * TILE;
    const char *rt0 = Rt + ((size_t)plane * d.nst128 * d.nnfl + nfl0) * 2048 + wm * 1024;
    const size_t rt_step = (size_t)d.nnfl * 2048;
    const char *a_src = (const char *)msa_cm;
    const u32 l16 = (u32)lane * 16;
    const u32 last = wm < (2 * FNW) % 4 || (2 * FNW) % 4 == 0 ? 4 : 3;       // the fifth copy: piece wm + 16, or wm + 12 again
    const u32 vo0 = l16, vo1 = l16 + 4096, vo2 = l16 + 8192, vo3 = l16 + 12288, vo4 = l16 + last * 4096, d4 = last * 4096;
    const u32 lw = lds_addr(smem + lane * 16);
    const u32 lw_base = __builtin_amdgcn_readfirstlane(lw - l16);
    const u32 la0 = lds_addr(aring + (wm * 2) * 1024 + lane * 16);
    const u32 m0t0 = lw_base + wm * 1024, m0a0 = lw_base + 4 * TILE + wm * 2048;
    // arrival counter of the K loop (see the step block): behind the rings
    const u32 cnt = lw_base + 4 * TILE + 4 * ASLOT;
    if (tid == 0) *(u32 *)(smem + 4 * TILE + 4 * ASLOT) = 0;
    const u32 one = 1;
    u32 st, sp;
    asm volatile(PLM_BWDW_ZERO_ASM ::: PLM_BWDW_CLOBBERS);
    for (int i = 0; i < 3; i++) {
        const int stepc = min(k0 + i, k1 - 1);
        asm volatile(PLM_BWDW_ISSUE_ASM
                     :
                     : [tsrc] "s"(rt0 + (size_t)stepc * rt_step), [vo0] "v"(vo0), [vo1] "v"(vo1), [vo2] "v"(vo2),
                       [vo3] "v"(vo3), [vo4] "v"(vo4), [m0t] "s"(m0t0 + i * TILE), [d4] "s"(d4),
                       [asrc] "s"(a_src + (size_t)PLM_BWD_KSTEP * stepc), [acol] "v"(acol), [acol1] "v"(acol + 64), [m0a] "s"(m0a0 + i * ASLOT)
                     : "m0", "scc", "memory");
    }
    vm_wait<0>();
    __syncthreads();
    asm volatile(PLM_BWDW_PRIME_ASM
                 : [st] "=&s"(st)
                 : [lan] "v"(la0), [lbn] "v"(lw), [b0x] "s"(b0x), [k7f] "v"(k7f), [k80] "s"(k80)
                 : PLM_BWDW_CLOBBERS);
    // loop state in scalar registers, advanced by additions: ring slot of step ss, sources of the copies of step ss + 3
    int sc = 0;
    const char *tsrc = rt0 + (size_t)min(k0 + 3, k1 - 1) * rt_step;
    const char *asrc = a_src + (size_t)PLM_BWD_KSTEP * min(k0 + 3, k1 - 1);
    u32 tgt = 0;                                 // arrivals of the steps before ss: one per wave and step
    for (int ss = k0; ss < k1; ++ss) {
        const int sn = (sc + 1) & 3, snn = (sc + 3) & 3;
        asm volatile(PLM_BWDW_STEP_ASM
                     : [st] "=&s"(st), [sp] "=&s"(sp)
                     : [lb] "v"(lw + sc * TILE), [lbn] "v"(lw + sn * TILE), [lan] "v"(la0 + sn * ASLOT), [b0x] "s"(b0x),
                       [k7f] "v"(k7f), [k80] "s"(k80), [tsrc] "s"(tsrc), [vo0] "v"(vo0), [vo1] "v"(vo1), [vo2] "v"(vo2),
                       [vo3] "v"(vo3), [vo4] "v"(vo4), [m0t] "s"(m0t0 + snn * TILE), [d4] "s"(d4), [asrc] "s"(asrc),
                       [acol] "v"(acol), [acol1] "v"(acol + 64), [m0a] "s"(m0a0 + snn * ASLOT), [cnt] "v"(cnt),
                       [one] "v"(one), [tgt] "s"(tgt)
                     : PLM_BWDW_CLOBBERS);
        sc = sn;
        tgt += 4;
        const bool more = ss + 4 < k1;           // past the K range the last step is copied again
        tsrc += more ? rt_step : 0;
        asrc += more ? PLM_BWD_KSTEP : 0;
    }
    // the copies still in flight land before the workgroup gives up its LDS; the last MFMAs have written a[..]
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    if (!row_ok) return;
    int *Gp = G + (size_t)(plane * d.ksplit + ks) * d.nmf * d.nnfl * 256;
    bwdw_store_all(Gp, d, mf0, nfl0, lane, std::make_integer_sequence<int, FM>{});
}

hipError_t plm_launch_backward(const PlmDims &d, const int8_t *msa_cm, const void *Rt, int32_t *G, const int *run,
                               hipStream_t st) {
    const int ngroups = d.ncol_tiles * d.nplanes * d.ksplit;
    if (d.bwd_w) {
        if (d.Q != 21) return hipErrorInvalidValue;
        const size_t lds = (size_t)4 * PLM_BWDW_FN * 2 * 1024 + 4 * 4 * 2 * 1024 + 16;
        hipError_t e = plm_allow_lds<k_bwd_w<21>>(lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_bwd_w<21>), dim3(8 * ((ngroups + 7) / 8) * d.nrow_tiles), dim3(256), lds, st, d, msa_cm,
                           (const char *)Rt, (int *)G, run);
        return hipGetLastError();
    }
    const dim3 grid(8 * ((ngroups + 7) / 8) * d.nrow_tiles), block(512);
#define BWD_CASE(QQ, M, N)                                                                             \
    case QQ: {                                                                                         \
        const size_t lds = (size_t)2 * (2 * N * 2 * 1024 + 4 * 2 * 1024);   /* two buffers: digit tile + alignment bytes */ \
        {                                                                                              \
            hipError_t e = plm_allow_lds<k_bwd<QQ, M, N>>(lds);                                        \
            if (e != hipSuccess) return e;                                                             \
        }                                                                                              \
        hipLaunchKernelGGL((k_bwd<QQ, M, N>), grid, block, lds, st, d, msa_cm, (const char *)Rt, (int *)G, run);   \
    } break;
    switch (d.Q) {
        BWD_CASE(32, 8, 5)
        BWD_CASE(21, 7, 7)
        BWD_CASE(20, 5, 5)
        BWD_CASE(5, 5, 5)
        BWD_CASE(4, 4, 4)
    default:
        return hipErrorInvalidValue;
    }
#undef BWD_CASE
    return hipGetLastError();
}

// exact value of one slab element from k_bwd's int32 partials: K ranges summed per digit plane in 64 bits, planes
// combined in f64 (< 2^53: exact), rounded ONCE to f32.  off = element offset inside one partial slab.
__device__ __forceinline__ float g_combine(const int *__restrict__ G, size_t off, int ks_count, size_t kstride,
                                           int nplanes) {
    const size_t pstride = (size_t)ks_count * kstride;
    double v = 0.0, wgt = 1.0;
    for (int p = 0; p < nplanes; p++) {
        long long sp = 0;     // a single K range stays below 2^31, their sum need not
        for (int k = 0; k < ks_count; k++) sp += G[off + p * pstride + k * kstride];
        v += wgt * (double)sp;
        wgt *= 256.0;
    }
    return (float)v;
}
// slab element either way: int32 partials (ks_count >= 1) or an already combined float slab (ks_count = 0: the gathered
// slabs of the replicated multi-shard mode)
__device__ __forceinline__ float g_value(const void *__restrict__ G, size_t off, int ks_count, size_t kstride,
                                         int nplanes) {
    return ks_count > 0 ? g_combine((const int *)G, off, ks_count, kstride, nplanes) : ((const float *)G)[off];
}

// combine the digit planes and K ranges into this shard's float slab of the exchange buffer
__global__ __launch_bounds__(256) void k_slab_reduce(const int *__restrict__ G, float *__restrict__ slab, int64_t n,
                                                    int ksplit, int nplanes) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        slab[i] = g_combine(G, (size_t)i, ksplit, (size_t)n, nplanes);
}
hipError_t plm_launch_slab_reduce(const PlmDims &d, const int32_t *G, float *slab, hipStream_t st) {
    const int64_t n = (int64_t)d.nmf * d.nnfl * 256;
    hipLaunchKernelGGL(k_slab_reduce, dim3(2048), dim3(256), 0, st, (const int *)G, slab, n, d.ksplit, d.nplanes);
    return hipGetLastError();
}

// =========================================================================================
// K_assemble: gradient in the native layout.
//   block pair (I<=J), states (a,b), tile element (ii,jj):
//     g = 2^-R * ( G[(J,b),(I,a)][jj][ii] + G[(I,a),(J,b)][ii][jj] ) + 2 lambda_J x
//   An accumulator fragment stores element (row, col) at float index ((row>>2)*16+col)*4+(row&3).
//   mode 1 (marginals): out = 2^-R * inv_neff * G[(J,b),(I,a)][jj][ii].
// =========================================================================================
__device__ __forceinline__ size_t g_frag(const PlmDims &d, int ks_count, size_t slab_stride, int block16,
                                         int state, int mf) {
    // fragment (row fragment mf, column = (block16, state)) -> float offset of partial 0
    const int sh = plm_shard_of(d, block16);
    const int nfl = (block16 - plm_shard_lo(d, sh)) * d.Q + state;
    return (size_t)sh * slab_stride + ((size_t)mf * d.nnfl + nfl) * 256;
}
// squared Frobenius norm of every coupling block J_ij of an own block pair: n2[pair][ii * 16 + jj] (group regulariser)
__global__ __launch_bounds__(256) void k_pair_norms(PlmDims d, const float *__restrict__ x, float *__restrict__ n2) {
    const size_t base = d.nh_pad_l + (size_t)blockIdx.x * d.Q * d.Q * 256 + threadIdx.x;
    double s = 0;
    for (int ab = 0; ab < d.Q * d.Q; ab++) {
        const float v = x[base + (size_t)ab * 256];
        s += (double)v * (double)v;
    }
    n2[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)s;
}
hipError_t plm_launch_pair_norms(const PlmDims &d, const float *x, float *n2, hipStream_t st) {
    if (d.np_own <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pair_norms, dim3((unsigned)d.np_own), dim3(256), 0, st, d, x, n2);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void k_assemble(PlmDims d, const void *__restrict__ G, int ks_count,
                                                 size_t slab_stride, const float *__restrict__ ghalo,
                                                 const float *__restrict__ x,
                                                 float *__restrict__ gout, float lambda_j,
                                                 double *__restrict__ reg_part, int mode, float scale,
                                                 const float *__restrict__ pair_n2, float lambda_g) {
    __shared__ double red[4];
    const int a = blockIdx.y;
    int I, J;
    plm_pair_of_local(d, blockIdx.x, &I, &J);
    const int ii = threadIdx.x >> 4, jj = threadIdx.x & 15;
    const int i = I * 16 + ii, j = J * 16 + jj;
    const bool valid = i < d.L && j < d.L && i < j;
    const int t1 = ((jj >> 2) * 16 + ii) * 4 + (jj & 3);   // element [row=jj][col=ii]
    const int t2 = ((ii >> 2) * 16 + jj) * 4 + (ii & 3);   // element [row=ii][col=jj]
    const size_t kstride = (size_t)d.nmf * d.nnfl * 256;
    const size_t xoff = d.nh_pad_l + ((size_t)blockIdx.x * d.Q + a) * d.Q * 256 + threadIdx.x;
    // sharded-state: of the two fragments of a pair in a rectangle shared with another shard, the one over the OTHER
    // shard's column block arrives, summed over split-K, in ghalo[rectangle pair][a][b][fragment]
    const bool remote1 = d.sharded && (I < d.own_lo || I >= d.own_hi);      // G[(J,b),(I,a)]: column block I is not own
    const bool remote2 = d.sharded && (J < d.own_lo || J >= d.own_hi);      // G[(I,a),(J,b)]: column block J is not own
    const size_t hoff = (remote1 || remote2) ? (((size_t)(blockIdx.x - d.ntri) * d.Q + a) * d.Q) * 256 + (remote1 ? t1 : t2) : 0;
    double reg = 0;
    // group regulariser lambda_g sum_{i<j} sqrt(|J_ij|^2 + delta^2) (plm_hip.h PLM_GROUP_DELTA): gradient lambda_g J / norm
    const float gnorm = (lambda_g > 0.f && valid) ? sqrtf(pair_n2[(size_t)blockIdx.x * 256 + threadIdx.x] +
                                                          (float)(PLM_GROUP_DELTA * PLM_GROUP_DELTA)) : 1.f;
    const float gcoef = (lambda_g > 0.f) ? lambda_g / gnorm : 0.f;
    for (int b = 0; b < d.Q; b++) {
        float v;
        if (remote1) {
            v = ghalo[hoff + (size_t)b * 256];
        } else {
            const size_t o1 = g_frag(d, ks_count, slab_stride, I, a, J * d.Q + b) + t1;
            v = g_value(G, o1, ks_count, kstride, d.nplanes);
        }
        float out;
        if (mode == 0) {
            float v2 = 0.f;
            if (remote2) {
                v2 = ghalo[hoff + (size_t)b * 256];
            } else {
                const size_t o2 = g_frag(d, ks_count, slab_stride, J, b, I * d.Q + a) + t2;
                v2 = g_value(G, o2, ks_count, kstride, d.nplanes);
            }
            const float xv = x[xoff + (size_t)b * 256];
            const bool live = valid && !(d.gap_mode && (a == 0 || b == 0)) && a < d.Qc && b < d.Qc;
            out = live ? fmaf(scale, v + v2, (2.f * lambda_j + gcoef) * xv) : 0.f;
            if (live) reg += (double)xv * (double)xv;
        } else {
            out = valid ? scale * v : 0.f;
        }
        gout[xoff + (size_t)b * 256] = out;
    }
    if (mode == 0) {
        // the group term of a site pair is counted once: by the block of state a = 0
        const double t = block_reduce_sum((double)lambda_j * reg + ((lambda_g > 0.f && a == 0 && valid) ? (double)lambda_g * gnorm : 0.0), red);
        if (threadIdx.x == 0) reg_part[(size_t)blockIdx.x * d.Q + a] = t;
    }
}
// k_assemble with 16-byte accesses (round 6; the int32 partials of this shard's own G only -- the gathered float slabs of
// the replicated multi-shard mode keep k_assemble).  k_assemble reads every accumulator fragment in 4-byte pieces, a
// wave touching all eight cache lines of a fragment for a quarter of their bytes: 0.21 ms for 0.49 GB at the headline,
// and as much on a shard of an eighth of the columns, whose K split multiplies the partial slabs.  Here a WAVE takes one
// state b (b = wave, wave + 4, ...) and reads whole fragments, one int4 / float4 per lane and plane:
//   * G[(J,b),(I,a)] holds element (row jj, col ii) at ((jj >> 2) * 16 + ii) * 4 + (jj & 3): lane l owns (ii = l & 15,
//     jj = 4 (l >> 4) + 0..3) -- four neighbours of the output tile row ii: one float4 of x and of the gradient;
//   * G[(I,a),(J,b)] (or the gradient halo, same layout) holds (row ii, col jj) at ((ii >> 2) * 16 + jj) * 4 + (ii & 3):
//     lane l reads (ii = 4 (l >> 4) + 0..3, jj = l & 15) and turns them round through 1 KB of LDS per wave.
// Element by element the arithmetic is k_assemble's (same sums in the same order): the gradient is bit-identical.
__device__ __forceinline__ void g_combine4(const int *__restrict__ G, size_t off, int ks_count, size_t kstride, int nplanes,
                                           float (&out)[4]) {
    const size_t pstride = (size_t)ks_count * kstride;
    double v[4] = {0.0, 0.0, 0.0, 0.0}, wgt = 1.0;
    for (int p = 0; p < nplanes; p++) {
        long long sp[4] = {0, 0, 0, 0};
        for (int k = 0; k < ks_count; k++) {
            const int4 q = *(const int4 *)(G + off + p * pstride + k * kstride);
            sp[0] += q.x; sp[1] += q.y; sp[2] += q.z; sp[3] += q.w;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += wgt * (double)sp[e];
        wgt *= 256.0;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) out[e] = (float)v[e];
}
__global__ __launch_bounds__(256) void k_assemble_v(PlmDims d, const int *__restrict__ G, int ks_count,
                                                   const float *__restrict__ ghalo, const float *__restrict__ x,
                                                   float *__restrict__ gout, float lambda_j, double *__restrict__ reg_part,
                                                   int mode, float scale, const float *__restrict__ pair_n2, float lambda_g) {
    __shared__ double red[4];
    __shared__ __attribute__((aligned(16))) float turn[4][256];
    const int a = blockIdx.y;
    int I, J;
    plm_pair_of_local(d, blockIdx.x, &I, &J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ii = lane & 15, jq = lane >> 4;                 // output: tile row ii, columns 4 jq .. 4 jq + 3
    const int i = I * 16 + ii;
    const size_t kstride = (size_t)d.nmf * d.nnfl * 256;
    const size_t xbase = d.nh_pad_l + ((size_t)blockIdx.x * d.Q + a) * d.Q * 256 + (size_t)ii * 16 + 4 * jq;
    // (see k_assemble: the fragment over the other shard's column block comes from the gradient halo, the sender's layout)
    const bool remote1 = d.sharded && (I < d.own_lo || I >= d.own_hi), remote2 = d.sharded && (J < d.own_lo || J >= d.own_hi);
    const size_t hbase = (remote1 || remote2) ? (((size_t)(blockIdx.x - d.ntri) * d.Q + a) * d.Q) * 256 + 4 * lane : 0;
    bool valid[4];
    float gcoef[4], gnorm[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int j = J * 16 + 4 * jq + e;
        valid[e] = i < d.L && j < d.L && i < j;
        gnorm[e] = (lambda_g > 0.f && valid[e]) ? sqrtf(pair_n2[(size_t)blockIdx.x * 256 + ii * 16 + 4 * jq + e] +
                                                        (float)(PLM_GROUP_DELTA * PLM_GROUP_DELTA)) : 1.f;
        gcoef[e] = (lambda_g > 0.f) ? lambda_g / gnorm[e] : 0.f;
    }
    double reg = 0;
    const int nb4 = (d.Q + 3) / 4;
    for (int bb = 0; bb < nb4; bb++) {                       // uniform trip count: the barriers below are workgroup-wide
        const int b = bb * 4 + wave;
        const bool on = b < d.Q;
        float v1[4] = {0.f, 0.f, 0.f, 0.f}, v2[4] = {0.f, 0.f, 0.f, 0.f};
        if (on) {
            if (remote1) {
                const float4 h = *(const float4 *)(ghalo + hbase + (size_t)b * 256);
                v1[0] = h.x; v1[1] = h.y; v1[2] = h.z; v1[3] = h.w;
            } else {
                g_combine4(G, g_frag(d, ks_count, 0, I, a, J * d.Q + b) + 4 * lane, ks_count, kstride, d.nplanes, v1);
            }
            if (mode == 0) {
                float w2[4];
                if (remote2) {
                    const float4 h = *(const float4 *)(ghalo + hbase + (size_t)b * 256);
                    w2[0] = h.x; w2[1] = h.y; w2[2] = h.z; w2[3] = h.w;
                } else {
                    g_combine4(G, g_frag(d, ks_count, 0, J, b, I * d.Q + a) + 4 * lane, ks_count, kstride, d.nplanes, w2);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) turn[wave][(4 * jq + e) * 16 + ii] = w2[e];     // (row 4 (l >> 4) + e, col l & 15)
            }
        }
        __syncthreads();
        if (on && mode == 0) {
            const float4 t = *(const float4 *)&turn[wave][ii * 16 + 4 * jq];
            v2[0] = t.x; v2[1] = t.y; v2[2] = t.z; v2[3] = t.w;
        }
        __syncthreads();
        if (on) {
            float o[4];
            if (mode == 0) {
                const float4 xv4 = *(const float4 *)(x + xbase + (size_t)b * 256);
                const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool live = valid[e] && !(d.gap_mode && (a == 0 || b == 0)) && a < d.Qc && b < d.Qc;
                    o[e] = live ? fmaf(scale, v1[e] + v2[e], (2.f * lambda_j + gcoef[e]) * xv[e]) : 0.f;
                    if (live) reg += (double)xv[e] * (double)xv[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = valid[e] ? scale * v1[e] : 0.f;
            }
            *(float4 *)(gout + xbase + (size_t)b * 256) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (mode == 0) {
        double grp = 0;      // the group term of a site pair is counted once: by the workgroup of state a = 0, wave 0
        if (lambda_g > 0.f && a == 0 && wave == 0)
#pragma unroll
            for (int e = 0; e < 4; e++) grp += valid[e] ? (double)lambda_g * gnorm[e] : 0.0;
        const double t = block_reduce_sum((double)lambda_j * reg + grp, red);
        if (threadIdx.x == 0) reg_part[(size_t)blockIdx.x * d.Q + a] = t;
    }
}
// field part: column sums of the residuals arrive through the "ones" row fragment
__global__ __launch_bounds__(256) void k_assemble_h(PlmDims d, const void *__restrict__ G, int ks_count,
                                                   size_t slab_stride, const float *__restrict__ x,
                                                   float *__restrict__ gout, float lambda_h,
                                                   double *__restrict__ reg_part, int mode, float scale) {
    __shared__ double red[4];
    double reg = 0;
    const size_t kstride = (size_t)d.nmf * d.nnfl * 256;
    const int site_end = min(d.L, d.own_hi * 16);      // local field part: sites [h_site0, site_end)
    {   // one element per thread, one workgroup per 256 elements (nh_pad_l is a multiple of 256)
        const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
        float out = 0.f;
        if (idx < (int64_t)(site_end - d.h_site0) * d.Q) {
            const int i = d.h_site0 + (int)(idx / d.Q), a = (int)(idx % d.Q);
            const size_t o = g_frag(d, ks_count, slab_stride, i >> 4, a, d.nb16 * d.Q) + (size_t)(i & 15) * 4;
            const float v = g_value(G, o, ks_count, kstride, d.nplanes);
            if (mode != 1) {
                const float xv = x[idx];
                if (!(d.gap_mode && a == 0) && a < d.Qc) {
                    // mode 2: reduced objective of the variable-projection fit -- the fields are at their optimum
                    // for the current couplings, their gradient entries are zero by construction
                    out = (mode == 2) ? 0.f : fmaf(scale, v, 2.f * lambda_h * xv);
                    reg += (double)xv * (double)xv;
                }
            } else {
                out = scale * v;
            }
        }
        gout[idx] = out;
    }
    if (mode != 1) {
        const double t = block_reduce_sum(reg, red);
        if (threadIdx.x == 0) reg_part[d.np_own * d.Q + blockIdx.x] = (double)lambda_h * t;
    }
}
hipError_t plm_launch_assemble(const PlmDims &d, const void *G, int ks_count, const float *ghalo, const float *x,
                               float *g, float lambda_h, float lambda_j, double *reg_part, int mode, float inv_neff,
                               const float *pair_n2, float lambda_g, hipStream_t st) {
    // replicated multi-shard mode reads the gathered float slabs (one per shard, ks_count = 0); otherwise G holds this
    // shard's own int32 plane / K-range partials
    const size_t slab_stride = (d.sharded || ks_count > 0) ? 0 : plm_slab_bytes(d) / 4;
    const float scale = d.gscale * (mode == 1 ? inv_neff : 1.f);
    if (d.np_own > 0 && ks_count > 0 && slab_stride == 0)      // this shard's own int32 partials: 16-byte accesses
        hipLaunchKernelGGL(k_assemble_v, dim3((unsigned)d.np_own, d.Q), dim3(256), 0, st, d, (const int *)G, ks_count, ghalo, x, g,
                           lambda_j, reg_part, mode == 2 ? 0 : mode, scale, pair_n2, mode == 1 ? 0.f : lambda_g);
    else if (d.np_own > 0)
        hipLaunchKernelGGL(k_assemble, dim3((unsigned)d.np_own, d.Q), dim3(256), 0, st, d, G, ks_count, slab_stride,
                           ghalo, x, g, lambda_j, reg_part, mode == 2 ? 0 : mode, scale, pair_n2, mode == 1 ? 0.f : lambda_g);
    hipLaunchKernelGGL(k_assemble_h, dim3((unsigned)(d.nh_pad_l / 256)), dim3(256), 0, st, d, G, ks_count, slab_stride, x, g, lambda_h,
                       reg_part, mode, scale);
    return hipGetLastError();
}

// fx = sum(nll partials) + sum(reg partials); nll either from local partials (single shard)
// or from the per-shard sums that travelled in the slab tails
__global__ __launch_bounds__(256) void k_sum_partials(const double *__restrict__ p, int n, double *out) {
    __shared__ double red[4];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    const double t = block_reduce_sum(s, red);
    if (threadIdx.x == 0) *out = t;
}
__global__ __launch_bounds__(256) void k_finish_fx(const double *__restrict__ fx_part, int nfx,
                                                  const double *__restrict__ shard_nll, size_t shard_stride,
                                                  int nshard, const double *__restrict__ reg_part, int nreg,
                                                  double *out2) {
    __shared__ double red[4];
    double s = 0;
    if (nshard > 0) {
        if (threadIdx.x == 0)
            for (int k = 0; k < nshard; k++) s += *(const double *)((const char *)shard_nll + k * shard_stride);
    } else {
        // sixteen loads in flight, summed in index order as before (the plain loop was a chain of load latencies: 14 us)
        for (int i0 = threadIdx.x; i0 < nfx; i0 += 16 * 256) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = fx_part[min(i0 + j * 256, nfx - 1)];
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (i0 + j * 256 < nfx) s += v[j];
        }
    }
    const double nll = block_reduce_sum(s, red);
    __syncthreads();
    double rsum = 0;
    for (int i0 = threadIdx.x; i0 < nreg; i0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = reg_part[min(i0 + j * 256, nreg - 1)];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (i0 + j * 256 < nreg) rsum += v[j];
    }
    const double reg = block_reduce_sum(rsum, red);
    if (threadIdx.x == 0) {
        out2[0] = nll + reg;
        out2[1] = nll;
    }
}
hipError_t plm_launch_partial_sum(const double *part, int n, double *out, hipStream_t st) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, part, n, out);
    return hipGetLastError();
}
hipError_t plm_launch_finish_fx(const PlmDims &d, const double *fx_part, int n_fx_part, const double *shard_nll,
                                int n_shard_nll, const double *reg_part, int n_reg_part, double *out2,
                                hipStream_t st) {
    hipLaunchKernelGGL(k_finish_fx, dim3(1), dim3(256), 0, st, fx_part, n_fx_part, shard_nll,
                       plm_slab_bytes(d), n_shard_nll, reg_part, n_reg_part, out2);
    return hipGetLastError();
}

// =========================================================================================
// L-BFGS streaming kernels (row a7): dot products in f64, linear combinations in f32
// =========================================================================================
struct DotArgs {
    const float *a[4];
    const float *b[4];
};
template <int NP>
__global__ __launch_bounds__(256) void k_dots(DotArgs A, int64_t n4, double *__restrict__ scratch) {
    __shared__ double red[4];
    double s[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) s[p] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const float4 u = ((const float4 *)A.a[p])[i], v = ((const float4 *)A.b[p])[i];
            s[p] += (double)u.x * v.x + (double)u.y * v.y + (double)u.z * v.z + (double)u.w * v.w;
        }
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const double t = block_reduce_sum(s[p], red);
        if (threadIdx.x == 0) scratch[(size_t)p * PLM_DOT_BLOCKS + blockIdx.x] = t;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_dots_final(const double *__restrict__ scratch, double *out) {
    __shared__ double red[4];
    double s = 0;
    for (int i = threadIdx.x; i < PLM_DOT_BLOCKS; i += 256) s += scratch[(size_t)blockIdx.x * PLM_DOT_BLOCKS + i];
    const double t = block_reduce_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}
hipError_t plm_launch_dots(int npairs, const float *const *a, const float *const *b, int64_t n, double *scratch,
                           double *out, hipStream_t st) {
    if (npairs < 1 || npairs > 4 || (n & 3)) return hipErrorInvalidValue;
    DotArgs A;
    for (int p = 0; p < 4; p++) {
        A.a[p] = a[p < npairs ? p : 0];
        A.b[p] = b[p < npairs ? p : 0];
    }
    const int64_t n4 = n / 4;
    switch (npairs) {
    case 1: hipLaunchKernelGGL(k_dots<1>, dim3(PLM_DOT_BLOCKS), dim3(256), 0, st, A, n4, scratch); break;
    case 2: hipLaunchKernelGGL(k_dots<2>, dim3(PLM_DOT_BLOCKS), dim3(256), 0, st, A, n4, scratch); break;
    case 3: hipLaunchKernelGGL(k_dots<3>, dim3(PLM_DOT_BLOCKS), dim3(256), 0, st, A, n4, scratch); break;
    default: hipLaunchKernelGGL(k_dots<4>, dim3(PLM_DOT_BLOCKS), dim3(256), 0, st, A, n4, scratch); break;
    }
    hipLaunchKernelGGL(k_dots_final, dim3(npairs), dim3(256), 0, st, scratch, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_lincomb(float4 *__restrict__ out, float ca, const float4 *__restrict__ a,
                                                float cb, const float4 *__restrict__ b, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 u = a[i];
        float4 r;
        if (b) {
            const float4 v = b[i];
            r.x = fmaf(cb, v.x, ca * u.x); r.y = fmaf(cb, v.y, ca * u.y);
            r.z = fmaf(cb, v.z, ca * u.z); r.w = fmaf(cb, v.w, ca * u.w);
        } else {
            r.x = ca * u.x; r.y = ca * u.y; r.z = ca * u.z; r.w = ca * u.w;
        }
        out[i] = r;
    }
}
hipError_t plm_launch_lincomb(float *out, float ca, const float *a, float cb, const float *b, int64_t n,
                              hipStream_t st) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lincomb, dim3(2048), dim3(256), 0, st, (float4 *)out, ca, (const float4 *)a, cb,
                       (const float4 *)b, n / 4);
    return hipGetLastError();
}

// =========================================================================================
// layout conversion canonical <-> native, Frobenius norms of zero-sum-gauge blocks
// =========================================================================================
__global__ __launch_bounds__(256) void k_canon_to_native(PlmDims d, const float *__restrict__ xc,
                                                        float *__restrict__ xn) {
    // canonical side: the problem's alphabet (Qc states per site); native side: the instantiated size Q
    const int a = blockIdx.y;
    int I, J;
    plm_pair_of_local(d, blockIdx.x, &I, &J);
    const int ii = threadIdx.x >> 4, jj = threadIdx.x & 15;
    const int i = I * 16 + ii, j = J * 16 + jj;
    const bool valid = i < d.L && j < d.L && i < j && a < d.Qc;
    const size_t QQc = (size_t)d.Qc * d.Qc;
    const size_t src = valid ? (size_t)d.L * d.Qc + (size_t)plm_pair_index(i, j, d.L) * QQc + (size_t)a * d.Qc : 0;
    const size_t dst = d.nh_pad_l + ((size_t)blockIdx.x * d.Q + a) * d.Q * 256 + threadIdx.x;
    for (int b = 0; b < d.Q; b++) xn[dst + (size_t)b * 256] = (valid && b < d.Qc) ? xc[src + b] : 0.f;
}
__global__ __launch_bounds__(256) void k_copy_h(PlmDims d, const float *__restrict__ src, float *__restrict__ dst,
                                               int to_native) {
    // canonical fields [L][Qc] <-> local field part [sites h_site0 .. min(L, 16*own_hi)][Q]
    const int nsites = min(d.L, d.own_hi * 16) - d.h_site0;
    const int64_t n = to_native ? d.nh_pad_l : (int64_t)nsites * d.Qc;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        if (to_native) {
            const int64_t il = k / d.Q;
            const int a = (int)(k % d.Q);
            dst[k] = (il < nsites && a < d.Qc) ? src[(size_t)(d.h_site0 + il) * d.Qc + a] : 0.f;
        } else {
            const int64_t il = k / d.Qc;
            const int a = (int)(k % d.Qc);
            dst[(size_t)(d.h_site0 + il) * d.Qc + a] = src[(size_t)il * d.Q + a];
        }
    }
}
__global__ __launch_bounds__(64) void k_native_to_canon(PlmDims d, const float *__restrict__ xn,
                                                       float *__restrict__ xc) {
    const int i = blockIdx.x, j = blockIdx.y;
    if (j <= i) return;
    const int I = i >> 4, J = j >> 4;
    const int64_t k = plm_pair_local(d, I, J, nullptr);
    if (k < 0) return;                                  // not an own pair: left untouched (sharded-state)
    const int QQc = d.Qc * d.Qc;
    const size_t src = d.nh_pad_l + (size_t)k * d.Q * d.Q * 256 + (i & 15) * 16 + (j & 15);
    const size_t dst = (size_t)d.L * d.Qc + (size_t)plm_pair_index(i, j, d.L) * QQc;
    for (int ab = threadIdx.x; ab < QQc; ab += 64)
        xc[dst + ab] = xn[src + (size_t)((ab / d.Qc) * d.Q + ab % d.Qc) * 256];
}
hipError_t plm_launch_canon_to_native(const PlmDims &d, const float *xc, float *xn, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_h, dim3(64), dim3(256), 0, st, d, xc, xn, 1);
    if (d.np_own > 0)
        hipLaunchKernelGGL(k_canon_to_native, dim3((unsigned)d.np_own, d.Q), dim3(256), 0, st, d, xc, xn);
    return hipGetLastError();
}
hipError_t plm_launch_native_to_canon(const PlmDims &d, const float *xn, float *xc, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_h, dim3(64), dim3(256), 0, st, d, xn, xc, 0);
    hipLaunchKernelGGL(k_native_to_canon, dim3(d.L, d.L), dim3(64), 0, st, d, xn, xc);
    return hipGetLastError();
}

// row a8: zero-sum gauge + Frobenius norm per pair (couplings/model.py:208-231, 792),
// one wave per pair; means and the squared norm are accumulated in f64.
//   g_lo: first state of the model -- 1 in gap mode (plmc -g: the gauge and the norm are those of the (Q-1)-state model,
//         row and column 0 of a block do not exist), else 0;
//   a_lo: first state that enters the norm (1: PLM_CONV_FN_NO_GAP, gauge over all states, gap state left out of the norm).
__global__ __launch_bounds__(64) void k_fn(int L, int Q, const float *__restrict__ jij, float *__restrict__ fn,
                                          int a_lo, int g_lo) {
    __shared__ float blk[32 * 32];
    __shared__ double rm[32], cm[32];
    __shared__ double red[1];
    const int i = blockIdx.x, j = blockIdx.y;
    if (j <= i) {
        if (threadIdx.x == 0 && j == i) fn[(size_t)i * L + i] = 0.f;
        return;
    }
    const int QQ = Q * Q, Qm = Q - g_lo;
    const float *src = jij + (size_t)plm_pair_index(i, j, L) * QQ;
    for (int k = threadIdx.x; k < QQ; k += 64) blk[k] = src[k];
    __syncthreads();
    if (threadIdx.x < Q) {
        double r = 0, c = 0;
        for (int k = g_lo; k < Q; k++) {
            r += blk[threadIdx.x * Q + k];
            c += blk[k * Q + threadIdx.x];
        }
        rm[threadIdx.x] = r / Qm;
        cm[threadIdx.x] = c / Qm;
    }
    __syncthreads();
    double m = 0;
    for (int k = g_lo; k < Q; k++) m += rm[k];
    m /= Qm;
    const int n_lo = a_lo > g_lo ? a_lo : g_lo;
    double ss = 0;
    for (int k = threadIdx.x; k < QQ; k += 64) {
        const double z = (double)blk[k] - rm[k / Q] - cm[k % Q] + m;
        if (k / Q >= n_lo && k % Q >= n_lo) ss += z * z;
    }
    const double t = block_reduce_sum(ss, red);
    if (threadIdx.x == 0) {
        const float v = (float)sqrt(t);
        fn[(size_t)i * L + j] = v;
        fn[(size_t)j * L + i] = v;
    }
}
hipError_t plm_launch_fn(const PlmDims &d, const float *jij_canon, float *fn, int a_lo, int g_lo, hipStream_t st) {
    hipLaunchKernelGGL(k_fn, dim3(d.L, d.L), dim3(64), 0, st, d.L, d.Q, jij_canon, fn, a_lo, g_lo);
    return hipGetLastError();
}

// gap mode (plmc -g): pair frequencies over the jointly ungapped sequences.  In: canonical blocks [pairs][Q][Q] of raw
// weighted counts; out: f(a, b) / total for a, b >= 1 (total = the block's sum over a, b >= 1, or `fixed_total` > 0:
// PLM_CONV_G_FREQ_TOTAL), row and column 0 zero.  The total is summed by one thread in block order in f64 and every
// entry is (float)(f / total) -- the arithmetic of the host loop this replaces, bit for bit.
__global__ __launch_bounds__(64) void k_gap_normalise_pairs(int Q, float *__restrict__ fij, double fixed_total) {
    __shared__ double tot_s;
    float *f = fij + (size_t)blockIdx.x * Q * Q;
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int a = 1; a < Q; a++)
            for (int b = 1; b < Q; b++) tot += f[a * Q + b];
        tot_s = fixed_total > 0 ? fixed_total : tot;
    }
    __syncthreads();
    const double tot = tot_s;
    for (int k = threadIdx.x; k < Q * Q; k += 64) {
        const int a = k / Q, b = k % Q;
        f[k] = (a && b && tot > 0) ? (float)((double)f[k] / tot) : 0.f;
    }
}
// PLM_FLAG_COMPACT_GAPS: canonical pair blocks [pairs][Q][Q] -> [pairs][Q-1][Q-1] without row and column 0
__global__ __launch_bounds__(256) void k_compact_gap_blocks(int Q, const float *__restrict__ in, float *__restrict__ out,
                                                           size_t npair) {
    const int Qn = Q - 1, QQn = Qn * Qn;
    for (size_t p = blockIdx.x; p < npair; p += gridDim.x) {
        const float *src = in + p * Q * Q;
        float *dst = out + p * QQn;
        for (int k = threadIdx.x; k < QQn; k += 256) dst[k] = src[(k / Qn + 1) * Q + (k % Qn + 1)];
    }
}
hipError_t plm_launch_compact_gap_blocks(const PlmDims &d, const float *blocks, float *out, hipStream_t st) {
    const size_t npair = (size_t)d.L * (d.L - 1) / 2;
    if (npair)
        hipLaunchKernelGGL(k_compact_gap_blocks, dim3((unsigned)std::min<size_t>(npair, 65535)), dim3(256), 0, st, d.Qc, blocks,
                           out, npair);
    return hipGetLastError();
}
hipError_t plm_launch_gap_normalise_pairs(const PlmDims &d, float *fij_canon, double fixed_total, hipStream_t st) {
    const size_t npair = (size_t)d.L * (d.L - 1) / 2;
    if (npair) hipLaunchKernelGGL(k_gap_normalise_pairs, dim3((unsigned)npair), dim3(64), 0, st, d.Qc, fij_canon, fixed_total);
    return hipGetLastError();
}

// =========================================================================================
// Diagonal of the Hessian at the start point (independent-site model), inverted: the H0 of the preconditioned
// L-BFGS (plm_host.cpp).  With p_i = the site's start distribution and v = p (1 - p):
//   d2f/dh_i(a)^2      = N_eff v_i(a) + 2 lambda_h
//   d2f/dJ_ij(a,b)^2   = N_eff ( f_j(b) v_i(a) + f_i(a) v_j(b) ) + 2 lambda_J
// fv = [f | v] as two L*Q arrays.  Structurally-zero entries (padding, i >= j inside a diagonal block, state 0 in
// gap mode) get 0, so a direction never leaves the parameter subspace.
// =========================================================================================
__global__ __launch_bounds__(256) void k_precond_j(PlmDims d, const float *__restrict__ fv, float neff, float lambda_j,
                                                  float *__restrict__ dinv) {
    const int a = blockIdx.y;
    int I, J;
    plm_pair_of_local(d, blockIdx.x, &I, &J);
    const int ii = threadIdx.x >> 4, jj = threadIdx.x & 15;
    const int i = I * 16 + ii, j = J * 16 + jj;
    const bool valid = i < d.L && j < d.L && i < j;
    const float *f = fv, *v = fv + (size_t)d.L * d.Q;
    const float fia = valid ? f[(size_t)i * d.Q + a] : 0.f, via = valid ? v[(size_t)i * d.Q + a] : 0.f;
    const size_t dst = d.nh_pad_l + ((size_t)blockIdx.x * d.Q + a) * d.Q * 256 + threadIdx.x;
    for (int b = 0; b < d.Q; b++) {
        float out = 0.f;
        if (valid && !(d.gap_mode && (a == 0 || b == 0)) && a < d.Qc && b < d.Qc)
            out = 1.f / (neff * (f[(size_t)j * d.Q + b] * via + fia * v[(size_t)j * d.Q + b]) + 2.f * lambda_j);
        dinv[dst + (size_t)b * 256] = out;
    }
}
__global__ __launch_bounds__(256) void k_precond_h(PlmDims d, const float *__restrict__ fv, float neff, float lambda_h,
                                                  float *__restrict__ dinv) {
    const int site_end = min(d.L, d.own_hi * 16);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= d.nh_pad_l) return;
    float out = 0.f;
    if (idx < (int64_t)(site_end - d.h_site0) * d.Q) {
        const int i = d.h_site0 + (int)(idx / d.Q), a = (int)(idx % d.Q);
        if (!(d.gap_mode && a == 0) && a < d.Qc) out = 1.f / (neff * fv[(size_t)d.L * d.Q + (size_t)i * d.Q + a] + 2.f * lambda_h);
    }
    dinv[idx] = out;
}
hipError_t plm_launch_precond(const PlmDims &d, const float *fv, float neff, float lambda_h, float lambda_j, float *dinv,
                              hipStream_t st) {
    if (d.np_own > 0)
        hipLaunchKernelGGL(k_precond_j, dim3((unsigned)d.np_own, d.Q), dim3(256), 0, st, d, fv, neff, lambda_j, dinv);
    hipLaunchKernelGGL(k_precond_h, dim3((unsigned)(d.nh_pad_l / 256)), dim3(256), 0, st, d, fv, neff, lambda_h, dinv);
    return hipGetLastError();
}

// =========================================================================================
// Alignment statistics of the align stage (row N3; twins: align/alignment.py:707-747 Alignment.count,
// :1157-1190 identities_to_seq).  HBM-bound byte work: the matrix is read once per kernel.
//   k_align_rows: one lane per sequence, 16 bytes of its row per load; gaps and identities to the query (the query
//                 row sits in LDS) counted with the packed-byte compare of the reweighting kernel
//   k_align_cols: one lane per column, lanes of a wave read consecutive bytes of a row (coalesced), the sequences are
//                 split over blockIdx.y and merged with integer atomics (order-independent)
// =========================================================================================
__device__ __forceinline__ int eq_bytes(u32 a, u32 b) {          // number of equal bytes (all bytes < 0x80)
    return 4 - __builtin_popcount(((a ^ b) + 0x7f7f7f7fu) & 0x80808080u);
}
__global__ __launch_bounds__(256) void k_align_rows(const int8_t *__restrict__ msa, int n, int L, int gap,
                                                   const int8_t *__restrict__ query, int32_t *__restrict__ seq_gaps,
                                                   int32_t *__restrict__ ident) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the query row, padded to 4 bytes with 0x7e
    const int Lw = (L + 3) / 4;
    for (int k = threadIdx.x; k < Lw * 4; k += 256) smem[k] = (query && k < L) ? query[k] : 0x7e;
    __syncthreads();
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int8_t *row = msa + (size_t)s * L;
    const u32 g4 = (u32)gap * 0x01010101u;
    int ng = 0, id = 0;
    const u32 *q32 = (const u32 *)smem;
    int k = 0;
    if ((((size_t)row) & 3) == 0) {           // aligned rows: dword loads
        for (; k + 4 <= L; k += 4) {
            const u32 v = *(const u32 *)(row + k);
            ng += eq_bytes(v, g4);
            id += eq_bytes(v, q32[k >> 2]);
        }
    }
    for (; k < L; k++) {
        ng += row[k] == gap;
        id += row[k] == smem[k];
    }
    if (seq_gaps) seq_gaps[s] = ng;
    if (ident) ident[s] = id;
}
__global__ __launch_bounds__(256) void k_align_cols(const int8_t *__restrict__ msa, int n, int L, int gap,
                                                   int32_t *__restrict__ col_gaps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int per = (n + gridDim.y - 1) / gridDim.y, s0 = blockIdx.y * per, s1 = min(n, s0 + per);
    if (c >= L) return;
    int cnt = 0;
    for (int s = s0; s < s1; s++) cnt += msa[(size_t)s * L + c] == gap;
    if (cnt) atomicAdd(&col_gaps[c], cnt);
}
hipError_t plm_launch_align_stats(const int8_t *msa, int n, int L, int gap_state, const int8_t *query, int32_t *seq_gaps,
                                  int32_t *col_gaps, int32_t *ident, hipStream_t st) {
    if (seq_gaps || ident)
        hipLaunchKernelGGL(k_align_rows, dim3((n + 255) / 256), dim3(256), (size_t)((L + 3) / 4) * 4, st, msa, n, L,
                           gap_state, query, seq_gaps, ident);
    if (col_gaps) {
        hipError_t e = hipMemsetAsync(col_gaps, 0, sizeof(int32_t) * L, st);
        if (e != hipSuccess) return e;
        const int ysplit = std::max(1, std::min(1024, n / 256));
        hipLaunchKernelGGL(k_align_cols, dim3((L + 255) / 256, ysplit), dim3(256), 0, st, msa, n, L, gap_state, col_gaps);
    }
    return hipGetLastError();
}

// =========================================================================================
// Sharded-state mode: the two exchanges of an evaluation.  Of the rectangle of block pairs shared by
// two shards each owns half of the rows (plm_pair_owner).
//   couplings: a shard's half of a rectangle lies contiguously in its local vector (behind the triangle, partner order):
//              it is sent from there, no packing; the partner finds it in xhalo[oth_base[owner] + pair].
//   gradient:  for every pair of the PARTNER's half a shard contributes the fragment over ITS column block, summed over
//              digit planes and K ranges (k_pack_g), fragment layout as in G: gsend[oth_base[owner] + pair][a][b][256];
//              the owner reads it as ghalo[pair - ntri] (k_assemble).
// =========================================================================================
__global__ __launch_bounds__(256) void k_pack_g(PlmDims d, const int *__restrict__ G, float *__restrict__ out) {
    // one workgroup per (pair of a partner's half, state a); a wave per state b, one int4 per lane,
    // plane and K range
    const int a = blockIdx.y;
    int64_t k = blockIdx.x;
    int p = 0;
    for (; p < d.nshards; p++) {
        if (p == d.shard) continue;
        const int64_t n = (int64_t)d.nblk_own * plm_shard_cnt(d, p) - plm_half_blocks(d, p);      // p's half
        if (k < n) break;
        k -= n;
    }
    if (p >= d.nshards) return;
    int I, J;             // the pair (I < J): this shard is the lower one of the two iff d.shard < p
    plm_half_pair(d, p, false, k, &I, &J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t dst = (((size_t)blockIdx.x * d.Q + a) * d.Q) * 256 + 4 * lane;
    const size_t kstride = (size_t)d.nmf * d.nnfl * 256;
    for (int b = wave; b < d.Q; b += 4) {
        // own column block I: G[(row J,b),(col I,a)]; own column block J: G[(row I,a),(col J,b)]
        const size_t o = (d.shard < p ? g_frag(d, d.ksplit, 0, I, a, J * d.Q + b) : g_frag(d, d.ksplit, 0, J, b, I * d.Q + a)) + 4 * lane;
        float v[4];
        g_combine4(G, o, d.ksplit, kstride, d.nplanes, v);
        *(float4 *)(out + dst + (size_t)b * 256) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
hipError_t plm_launch_pack_g(const PlmDims &d, const int32_t *G, float *sendbuf, hipStream_t st) {
    if (d.nblk_own <= 0 || d.nx_halo <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_g, dim3((unsigned)d.nx_halo, d.Q), dim3(256), 0, st, d, (const int *)G, sendbuf);
    return hipGetLastError();
}

// =========================================================================================
// vector-free L-BFGS (row a7): all dot products the two-loop recursion needs come from one
// pass over the history (queries x basis), the direction from one fused linear combination
// =========================================================================================
#ifndef MD_CHUNK
#define MD_CHUNK 14     // = 2 m + 2 at the default history m = 6: the queries are read once (8: 0.689, 16: 0.679, 14: 0.660 ms of vector kernels per iteration)
#endif
// Optional diagonal metric (preconditioned L-BFGS, H0 = gamma * diag(dinv)): the product <q, b> carries the weight
// dinv[i] when BOTH the query (bit q of wq) and the basis vector (bit k of wb) are flagged.
template <int NQ>
__global__ __launch_bounds__(256) void k_multidot(PlmVecList Q, PlmVecList B, int64_t n4, double *__restrict__ scratch,
                                                 const float4 *__restrict__ dinv, unsigned wq, unsigned long long wb) {
    // blockIdx.y selects a chunk of MD_CHUNK basis vectors (keeps the f64 accumulators in registers)
    __shared__ double red[4];
    const int nb = B.n, k0 = blockIdx.y * MD_CHUNK;
    const float4 *bp[MD_CHUNK];
#pragma unroll
    for (int k = 0; k < MD_CHUNK; k++) bp[k] = (const float4 *)B.v[min(k0 + k, nb - 1)];
    const unsigned wbc = (unsigned)(wb >> k0) & ((1u << MD_CHUNK) - 1u);
    const bool any_w = dinv != nullptr && wq != 0 && wbc != 0;
    double acc[NQ][MD_CHUNK];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int k = 0; k < MD_CHUNK; k++) acc[q][k] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 qv[NQ], qd[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) qd[q] = qv[q] = ((const float4 *)Q.v[q])[i];
        if (any_w) {
            const float4 dv = dinv[i];
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if ((wq >> q) & 1u) { qd[q].x *= dv.x; qd[q].y *= dv.y; qd[q].z *= dv.z; qd[q].w *= dv.w; }
        }
#pragma unroll
        for (int k = 0; k < MD_CHUNK; k++) {
            const float4 b = bp[k][i];
            const bool wk = (wbc >> k) & 1u;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const float4 a = wk ? qd[q] : qv[q];
                acc[q][k] += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int k = 0; k < MD_CHUNK; k++) {
            const double t = block_reduce_sum(acc[q][k], red);
            if (threadIdx.x == 0 && k0 + k < nb) scratch[((size_t)q * nb + k0 + k) * PLM_DOT_BLOCKS + blockIdx.x] = t;
            __syncthreads();
        }
}
hipError_t plm_launch_multidot(const PlmVecList &queries, const PlmVecList &basis, int64_t n, double *scratch,
                               double *out, const float *dinv, unsigned wq, unsigned long long wb, hipStream_t st) {
    if (queries.n < 1 || queries.n > 4 || basis.n < 1 || basis.n > PLM_MAX_BASIS || (n & 3)) return hipErrorInvalidValue;
    const int64_t n4 = n / 4;
    const dim3 grid(PLM_DOT_BLOCKS, (basis.n + MD_CHUNK - 1) / MD_CHUNK), block(256);
    const float4 *d4 = (const float4 *)dinv;
    switch (queries.n) {
    case 1: hipLaunchKernelGGL(k_multidot<1>, grid, block, 0, st, queries, basis, n4, scratch, d4, wq, wb); break;
    case 2: hipLaunchKernelGGL(k_multidot<2>, grid, block, 0, st, queries, basis, n4, scratch, d4, wq, wb); break;
    case 3: hipLaunchKernelGGL(k_multidot<3>, grid, block, 0, st, queries, basis, n4, scratch, d4, wq, wb); break;
    default: hipLaunchKernelGGL(k_multidot<4>, grid, block, 0, st, queries, basis, n4, scratch, d4, wq, wb); break;
    }
    hipLaunchKernelGGL(k_dots_final, dim3(queries.n * basis.n), dim3(256), 0, st, scratch, out);
    return hipGetLastError();
}

// out = sum_{k < kw} c_k b_k  +  dinv (.) sum_{k >= kw} c_k b_k     (dinv == nullptr: plain linear combination)
__global__ __launch_bounds__(256) void k_multiaxpy(float4 *__restrict__ out, PlmVecList B, PlmCoefList C, int64_t n4,
                                                  const float4 *__restrict__ dinv, int kw,
                                                  const float4 *__restrict__ xacc, float stp, float4 *__restrict__ trial) {
    const int nb = B.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 r = {0.f, 0.f, 0.f, 0.f}, t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < kw; k++) {
            const float4 b = ((const float4 *)B.v[k])[i];
            const float c = C.c[k];
            r.x = fmaf(c, b.x, r.x); r.y = fmaf(c, b.y, r.y); r.z = fmaf(c, b.z, r.z); r.w = fmaf(c, b.w, r.w);
        }
#pragma unroll 4
        for (int k = kw; k < nb; k++) {
            const float4 b = ((const float4 *)B.v[k])[i];
            const float c = C.c[k];
            t.x = fmaf(c, b.x, t.x); t.y = fmaf(c, b.y, t.y); t.z = fmaf(c, b.z, t.z); t.w = fmaf(c, b.w, t.w);
        }
        if (dinv) {
            const float4 dv = dinv[i];
            r.x = fmaf(dv.x, t.x, r.x); r.y = fmaf(dv.y, t.y, r.y); r.z = fmaf(dv.z, t.z, r.z); r.w = fmaf(dv.w, t.w, r.w);
        } else {
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        out[i] = r;
        if (trial) {      // the first trial point of the line search: k_lincomb's arithmetic on the rounded direction
            const float4 u = xacc[i];
            trial[i] = make_float4(fmaf(stp, r.x, 1.f * u.x), fmaf(stp, r.y, 1.f * u.y), fmaf(stp, r.z, 1.f * u.z),
                                   fmaf(stp, r.w, 1.f * u.w));
        }
    }
}
hipError_t plm_launch_multiaxpy(float *out, const PlmVecList &basis, const PlmCoefList &coef, int64_t n,
                                const float *dinv, int first_weighted, hipStream_t st) {
    if (basis.n < 1 || basis.n > PLM_MAX_BASIS || (n & 3)) return hipErrorInvalidValue;
    const int kw = dinv ? std::max(0, std::min(basis.n, first_weighted)) : basis.n;
    hipLaunchKernelGGL(k_multiaxpy, dim3(2048), dim3(256), 0, st, (float4 *)out, basis, coef, n / 4,
                       (const float4 *)dinv, kw, (const float4 *)nullptr, 0.f, (float4 *)nullptr);
    return hipGetLastError();
}
hipError_t plm_launch_multiaxpy_trial(float *out, const PlmVecList &basis, const PlmCoefList &coef, int64_t n,
                                      const float *dinv, int first_weighted, const float *xacc, float stp, float *trial,
                                      hipStream_t st) {
    if (basis.n < 1 || basis.n > PLM_MAX_BASIS || (n & 3) || !xacc || !trial) return hipErrorInvalidValue;
    const int kw = dinv ? std::max(0, std::min(basis.n, first_weighted)) : basis.n;
    hipLaunchKernelGGL(k_multiaxpy, dim3(2048), dim3(256), 0, st, (float4 *)out, basis, coef, n / 4,
                       (const float4 *)dinv, kw, (const float4 *)xacc, stp, (float4 *)trial);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_sy(float4 *__restrict__ s, float4 *__restrict__ y, const float4 *__restrict__ x,
                                           const float4 *__restrict__ xp, const float4 *__restrict__ g,
                                           const float4 *__restrict__ gp, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = x[i], b = xp[i], c = g[i], e = gp[i];
        s[i] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        y[i] = make_float4(c.x - e.x, c.y - e.y, c.z - e.z, c.w - e.w);
    }
}
hipError_t plm_launch_sy(float *s, float *y, const float *x, const float *xp, const float *g, const float *gp,
                         int64_t n, hipStream_t st) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sy, dim3(2048), dim3(256), 0, st, (float4 *)s, (float4 *)y, (const float4 *)x,
                       (const float4 *)xp, (const float4 *)g, (const float4 *)gp, n / 4);
    return hipGetLastError();
}

// ---- the pair, the Gram pass and the scalar products of a trial point in one pass (plm_launch_sy_multidot) -------------
#define SYD_OLD 10        // basis vectors read from memory per blockIdx.y: 2 (m - 1) at the default history m = 6
struct PlmSyDot {
    const float4 *x, *xp, *g, *gp, *dir, *dinv;
    float4 *s_new, *y_new;
    const float4 *old[PLM_MAX_BASIS];      // the basis vectors that are not s_new / y_new / g
    int pos[PLM_MAX_BASIS];                // ... and their positions in the caller's basis order
    int n_old, nb, pos_s, pos_y, pos_g0, pos_g1;      // (-1: the caller's basis does not hold that vector)
    unsigned wq;
    unsigned long long wb;
    int64_t n4, nh4;
};
__device__ __forceinline__ double dot4(const float4 &a, const float4 &b) {
    return (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
}
__global__ __launch_bounds__(256) void k_sy_multidot(PlmSyDot A, double *__restrict__ scratch) {
    __shared__ double red[4];
    const int k0 = blockIdx.y * SYD_OLD;
    const bool first = blockIdx.y == 0;
    const float4 *bp[SYD_OLD];
    unsigned wold = 0;
#pragma unroll
    for (int k = 0; k < SYD_OLD; k++) {
        const int kk = min(k0 + k, max(A.n_old - 1, 0));
        bp[k] = A.n_old > 0 ? A.old[kk] : A.g;
        if (A.n_old > 0 && ((A.wb >> A.pos[kk]) & 1ull)) wold |= 1u << k;
    }
    const int spos[4] = {A.pos_s, A.pos_y, A.pos_g0, A.pos_g1};
    bool wsp[4];
#pragma unroll
    for (int e = 0; e < 4; e++) wsp[e] = spos[e] >= 0 && ((A.wb >> spos[e]) & 1ull);
    const bool any_w = A.dinv != nullptr && A.wq != 0;
    double acc[3][SYD_OLD], sp[3][4], ex[3] = {0, 0, 0}, ssh = 0;      // ssh: s.s over the fields
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int k = 0; k < SYD_OLD; k++) acc[q][k] = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) sp[q][e] = 0;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n4; i += (int64_t)gridDim.x * 256) {
        const float4 xv = A.x[i], xpv = A.xp[i], gv = A.g[i], gpv = A.gp[i];
        float4 qv[3], qd[3];
        qv[0] = make_float4(xv.x - xpv.x, xv.y - xpv.y, xv.z - xpv.z, xv.w - xpv.w);
        qv[1] = make_float4(gv.x - gpv.x, gv.y - gpv.y, gv.z - gpv.z, gv.w - gpv.w);
        qv[2] = gv;
        if (first) {
            A.s_new[i] = qv[0];
            A.y_new[i] = qv[1];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) qd[q] = qv[q];
        if (any_w) {
            const float4 dv = A.dinv[i];
#pragma unroll
            for (int q = 0; q < 3; q++)
                if ((A.wq >> q) & 1u) { qd[q].x *= dv.x; qd[q].y *= dv.y; qd[q].z *= dv.z; qd[q].w *= dv.w; }
        }
        if (A.n_old > 0) {
#pragma unroll
            for (int k = 0; k < SYD_OLD; k++) {
                const float4 b = bp[k][i];
                const bool wk = (wold >> k) & 1u;
#pragma unroll
                for (int q = 0; q < 3; q++) acc[q][k] += dot4(wk ? qd[q] : qv[q], b);
            }
        }
        if (first) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float4 b = e == 0 ? qv[0] : e == 1 ? qv[1] : gv;
#pragma unroll
                for (int q = 0; q < 3; q++) sp[q][e] += dot4(wsp[e] ? qd[q] : qv[q], b);
            }
            ex[0] += dot4(gv, A.dir[i]);
            const double x2 = dot4(xv, xv);
            ex[1] += x2;
            if (i < A.nh4) {
                ex[2] += x2;
                ssh += dot4(qv[0], qv[0]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int k = 0; k < SYD_OLD; k++) {
            const double t = block_reduce_sum(acc[q][k], red);
            if (threadIdx.x == 0 && k0 + k < A.n_old)
                scratch[((size_t)q * A.nb + A.pos[k0 + k]) * PLM_DOT_BLOCKS + blockIdx.x] = t;
            __syncthreads();
        }
    if (!first) return;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double t = block_reduce_sum(sp[q][e], red);
            if (threadIdx.x == 0 && spos[e] >= 0) scratch[((size_t)q * A.nb + spos[e]) * PLM_DOT_BLOCKS + blockIdx.x] = t;
            __syncthreads();
        }
    {
        const double t = block_reduce_sum(ssh, red);
        if (threadIdx.x == 0) scratch[((size_t)3 * A.nb) * PLM_DOT_BLOCKS + blockIdx.x] = t;
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const double t = block_reduce_sum(ex[e], red);
        if (threadIdx.x == 0) scratch[((size_t)3 * A.nb + 1 + e) * PLM_DOT_BLOCKS + blockIdx.x] = t;
        __syncthreads();
    }
}
// rows [0, na) of the scratch go to out_a, the rows behind them to out_b
__global__ __launch_bounds__(256) void k_dots_final2(const double *__restrict__ scratch, double *out_a, int na, double *out_b) {
    __shared__ double red[4];
    double s = 0;
    for (int i = threadIdx.x; i < PLM_DOT_BLOCKS; i += 256) s += scratch[(size_t)blockIdx.x * PLM_DOT_BLOCKS + i];
    const double t = block_reduce_sum(s, red);
    if (threadIdx.x == 0) {
        if ((int)blockIdx.x < na) out_a[blockIdx.x] = t;
        else out_b[blockIdx.x - na] = t;
    }
}
hipError_t plm_launch_sy_multidot(float *s_new, float *y_new, const float *x, const float *xp, const float *g, const float *gp,
                                  const float *dir, const PlmVecList &basis, int64_t n, int64_t nh, double *scratch,
                                  double *out_md, double *out_ex, const float *dinv, unsigned wq, unsigned long long wb,
                                  hipStream_t st) {
    if (basis.n < 1 || basis.n > PLM_MAX_BASIS || (n & 3) || (nh & 3) || nh > n) return hipErrorInvalidValue;
    PlmSyDot A;
    memset(&A, 0, sizeof A);
    A.x = (const float4 *)x; A.xp = (const float4 *)xp; A.g = (const float4 *)g; A.gp = (const float4 *)gp;
    A.dir = (const float4 *)dir; A.dinv = (const float4 *)dinv;
    A.s_new = (float4 *)s_new; A.y_new = (float4 *)y_new;
    A.nb = basis.n; A.wq = wq; A.wb = wb; A.n4 = n / 4; A.nh4 = nh / 4;
    A.pos_s = A.pos_y = A.pos_g0 = A.pos_g1 = -1;
    for (int k = 0; k < basis.n; k++) {
        const float *v = basis.v[k];
        if (v == s_new && A.pos_s < 0) A.pos_s = k;
        else if (v == y_new && A.pos_y < 0) A.pos_y = k;
        else if (v == g && A.pos_g0 < 0) A.pos_g0 = k;
        else if (v == g && A.pos_g1 < 0) A.pos_g1 = k;
        else { A.old[A.n_old] = (const float4 *)v; A.pos[A.n_old++] = k; }
    }
    const dim3 grid(PLM_DOT_BLOCKS, std::max(1, (A.n_old + SYD_OLD - 1) / SYD_OLD)), block(256);
    hipLaunchKernelGGL(k_sy_multidot, grid, block, 0, st, A, scratch);
    hipLaunchKernelGGL(k_dots_final2, dim3(3 * basis.n + 4), dim3(256), 0, st, scratch, out_md, 3 * basis.n + 1, out_ex);
    return hipGetLastError();
}
