// plm_meanfield.hip -- mean-field direct coupling analysis on gfx950.
//
// Replaces the arithmetic of evcouplings/couplings/mean_field.py:163-222 (MeanFieldDCA.fit) and :842-893
// (direct_information):
//   weights (k_reweight), f_i / f_ij (the one-hot Gram GEMM of the PLM solver)    -- plm_host.cpp
//   pseudo-count regularisation + covariance matrix  C = rf_ij - rf_i rf_j       -- k_mf_cov      (:717-790, :897-940)
//   J = -C^-1                                                                     -- blocked Cholesky inverse (below)
//   couplings as dense / pair blocks, fields h_i                                  -- k_mf_extract, k_mf_fields (:943-1014)
//   direct information of every pair (fixed-point iteration of the two-site model)-- k_mf_di      (:792-893)
// Everything after the frequencies is float64, like the reference.  The inverse of the symmetric positive
// definite covariance matrix is a hand-written blocked Cholesky factorisation + triangular inverse + X^T X
// on one tiled f64 GEMM kernel: rocSOLVER does the same in one call, but loading its 0.9 GB library costs
// ~230 s on a fresh MI355X box (measured), against 0.1 s for the whole mean-field run at L = 300.
#include "plm_internal.h"
#include "../../include/plm_hip.h"
#include <math.h>
#include <string.h>
#include <vector>

int plm_fail(int code, const char *fmt, ...);   // plm_host.cpp: records the message for plm_last_error()

namespace {

__device__ __forceinline__ double mf_rfi(const float *__restrict__ fi, int q, int i, int a, double pc) {
    return (1.0 - pc) * (double)fi[(size_t)i * q + a] + pc / (double)q;
}
// regularised pair frequency of (i, a), (j, b); fij = i<j blocks [a][b] (canonical order)
__device__ __forceinline__ double mf_rfij(const float *__restrict__ fi, const float *__restrict__ fij, int L, int q,
                                          int i, int a, int j, int b, double pc) {
    if (i == j) return (1.0 - pc) * ((a == b) ? (double)fi[(size_t)i * q + a] : 0.0) + ((a == b) ? pc / (double)q : 0.0);
    double f;
    if (i < j) f = fij[(plm_pair_index(i, j, L) * q + a) * q + b];
    else f = fij[(plm_pair_index(j, i, L) * q + b) * q + a];
    return (1.0 - pc) * f + pc / ((double)q * q);
}

// C[(i,a),(j,b)] = rf_ij(a,b) - rf_i(a) rf_j(b),  a, b < q-1   (mean_field.py:897-940)
// stored n_pad x n_pad (n rounded up to the 64-wide blocks of the factorisation) with an identity tail
__global__ __launch_bounds__(256) void k_mf_cov(const float *__restrict__ fi, const float *__restrict__ fij, int L,
                                               int q, double pc, int np, double *__restrict__ C) {
    const int n = L * (q - 1);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)np * np) return;
    const int r = (int)(idx / np), c = (int)(idx % np);
    if (r >= n || c >= n) {
        C[idx] = (r == c) ? 1.0 : 0.0;
        return;
    }
    const int i = r / (q - 1), a = r % (q - 1), j = c / (q - 1), b = c % (q - 1);
    C[idx] = mf_rfij(fi, fij, L, q, i, a, j, b, pc) - mf_rfi(fi, q, i, a, pc) * mf_rfi(fi, q, j, b, pc);
}

// element (r, c) of the inverse (full symmetric storage, leading dimension = padded size, passed as n)
__device__ __forceinline__ double mf_inv(const double *__restrict__ A, int n, int r, int c) {
    return A[(size_t)r * n + c];
}

// J_ij(a,b) = -C^-1[(i,a),(j,b)] for a, b < q-1, 0 in the last row / column (mean_field.py:943-975):
// dense L x L x q x q doubles (diagonal blocks included, as the reference's reshape does) and / or the
// i<j blocks in float32 (the .model file's precision)
__global__ __launch_bounds__(256) void k_mf_extract(const double *__restrict__ Cinv, int n, int L, int q,
                                                   double *__restrict__ Jfull, float *__restrict__ Jpairs) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)L * L * q * q;
    if (idx >= tot) return;
    const int b = (int)(idx % q), a = (int)((idx / q) % q);
    const int j = (int)((idx / ((int64_t)q * q)) % L), i = (int)(idx / ((int64_t)q * q * L));
    double v = 0.0;
    if (a < q - 1 && b < q - 1) v = -mf_inv(Cinv, n, i * (q - 1) + a, j * (q - 1) + b);
    if (Jfull) Jfull[idx] = v;
    if (Jpairs && i < j) Jpairs[(plm_pair_index(i, j, L) * q + a) * q + b] = (float)v;
}

// h_i(a) = log(rf_i(a) / rf_i(q-1)) - sum_{j != i} sum_b J_ij(a,b) rf_j(b)      (mean_field.py:977-1014)
__global__ __launch_bounds__(256) void k_mf_fields(const double *__restrict__ Cinv, int n, const float *__restrict__ fi,
                                                  int L, int q, double pc, double *__restrict__ hi) {
    __shared__ double red[256];
    const int i = blockIdx.x / q, a = blockIdx.x % q;
    double s = 0.0;
    if (a < q - 1) {
        const int r = i * (q - 1) + a;
        for (int c = threadIdx.x; c < L * (q - 1); c += 256) {
            const int j = c / (q - 1), b = c % (q - 1);
            if (j != i) s += -mf_inv(Cinv, n, r, c) * mf_rfi(fi, q, j, b, pc);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) hi[(size_t)i * q + a] = log(mf_rfi(fi, q, i, a, pc) / mf_rfi(fi, q, i, q - 1, pc)) - red[0];
}

// Direct information of one pair per 64-lane wave (q <= 32): W = exp(J_ij); iterate
//   ht_i <- normalise(f_i / (W ht_j)),  ht_j <- normalise(f_j / (W^T ht_i))   from the uniform start until the
// largest change is <= 1e-4 (both updates use the OLD vectors, mean_field.py:792-840), then
//   P = W .* (ht_i ht_j^T) / sum,   DI = sum_ab P log((P + tiny) / (f_i(a) f_j(b) + tiny))   (:842-893)
// Source of the couplings: the inverse covariance matrix (Cinv, leading dimension n; frequencies regularised on the
// fly from the raw f32 fi) or, when Jdense is given, a dense L x L x q x q array with ready regularised f64
// frequencies rfi (the standalone plm_direct_information entry point).
__global__ __launch_bounds__(64) void k_mf_di(const double *__restrict__ Cinv, int n, const float *__restrict__ fi, int L,
                                             int q, double pc, const double *__restrict__ Jdense,
                                             const double *__restrict__ rfi, double *__restrict__ di) {
    __shared__ double W[32 * 33];
    __shared__ double hti[32], htj[32];
    const int lane = threadIdx.x;
    // pair number -> (i, j), i < j
    int64_t p = blockIdx.x;
    int i = 0;
    while (p >= L - 1 - i) { p -= L - 1 - i; i++; }
    const int j = i + 1 + (int)p;
    for (int k = lane; k < q * q; k += 64) {
        const int a = k / q, b = k % q;
        double v = 0.0;
        if (Jdense) v = Jdense[(((size_t)i * L + j) * q + a) * q + b];
        else if (a < q - 1 && b < q - 1) v = -mf_inv(Cinv, n, i * (q - 1) + a, j * (q - 1) + b);
        W[a * 33 + b] = exp(v);
    }
    const bool act = lane < q;
    const double fia = !act ? 0.0 : rfi ? rfi[(size_t)i * q + lane] : mf_rfi(fi, q, i, lane, pc);
    const double fja = !act ? 0.0 : rfi ? rfi[(size_t)j * q + lane] : mf_rfi(fi, q, j, lane, pc);
    if (act) hti[lane] = htj[lane] = 1.0 / (double)q;
    __syncthreads();
    auto wave_sum = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    auto wave_max = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        return v;
    };
    for (int it = 0; it < 100000; it++) {
        double t1 = 0.0, t2 = 0.0;   // lane a: (W ht_j)[a] and (ht_i W)[a]
        if (act)
            for (int b = 0; b < q; b++) {
                t1 += W[lane * 33 + b] * htj[b];
                t2 += hti[b] * W[b * 33 + lane];
            }
        double ui = act ? fia / t1 : 0.0, uj = act ? fja / t2 : 0.0;
        ui /= wave_sum(ui);
        uj /= wave_sum(uj);
        const double diff = wave_max(act ? fmax(fabs(ui - hti[lane]), fabs(uj - htj[lane])) : 0.0);
        __syncthreads();
        if (act) { hti[lane] = ui; htj[lane] = uj; }
        __syncthreads();
        if (!(diff > 1e-4)) break;
    }
    double psum = 0.0;
    if (act)
        for (int b = 0; b < q; b++) psum += W[lane * 33 + b] * hti[lane] * htj[b];
    psum = wave_sum(psum);
    const double tiny = 1.0e-100;
    double acc = 0.0;
    if (act)
        for (int b = 0; b < q; b++) {
            const double pab = W[lane * 33 + b] * hti[lane] * htj[b] / psum;
            const double fjb = rfi ? rfi[(size_t)j * q + b] : mf_rfi(fi, q, j, b, pc);
            acc += pab * log((pab + tiny) / (fia * fjb + tiny));
        }
    acc = wave_sum(acc);
    if (lane == 0) di[(size_t)i * L + j] = di[(size_t)j * L + i] = acc;
}

// ---- inverse of a symmetric positive definite matrix (n_pad a multiple of 64, row-major, full storage) ------
#define MF_NB 64
// C (+)= alpha * op(A) * op(B) on 64 x 64 output tiles; all dimensions multiples of 64 (K of 16).
// TA / TB: 0 = as stored, 1 = transposed.  256 threads, 4 x 4 outputs each, K tile 16 through LDS.
template <int TA, int TB>
__global__ __launch_bounds__(256) void k_dgemm(int K, double alpha, const double *__restrict__ A, int lda,
                                              const double *__restrict__ B, int ldb, double beta,
                                              double *__restrict__ C, int ldc) {
    __shared__ double As[16][MF_NB + 1], Bs[16][MF_NB + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * MF_NB, n0 = blockIdx.x * MF_NB;
    double acc[4][4] = {{0}};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = threadIdx.x; e < 16 * MF_NB; e += 256) {
            // As[kk][mm] = op(A)[m0 + mm][k0 + kk], Bs[kk][mm] = op(B)[k0 + kk][n0 + mm]; consecutive threads walk
            // the contiguous direction of the stored matrix
            if (TA) { const int kk = e / MF_NB, mm = e % MF_NB; As[kk][mm] = A[(size_t)(k0 + kk) * lda + m0 + mm]; }
            else    { const int kk = e % 16, mm = e / 16;       As[kk][mm] = A[(size_t)(m0 + mm) * lda + k0 + kk]; }
            if (TB) { const int kk = e % 16, mm = e / 16;       Bs[kk][mm] = B[(size_t)(n0 + mm) * ldb + k0 + kk]; }
            else    { const int kk = e / MF_NB, mm = e % MF_NB; Bs[kk][mm] = B[(size_t)(k0 + kk) * ldb + n0 + mm]; }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { a[r] = As[kk][ty * 4 + r]; b[r] = Bs[kk][tx * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[r][c] = fma(a[r], b[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double *o = C + (size_t)(m0 + ty * 4 + r) * ldc + n0 + tx * 4 + c;
            *o = alpha * acc[r][c] + ((beta == 0.0) ? 0.0 : beta * *o);
        }
}
template <int TA, int TB>
void dgemm(hipStream_t st, int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb,
           double beta, double *C, int ldc) {
    if (M <= 0 || N <= 0) return;
    hipLaunchKernelGGL((k_dgemm<TA, TB>), dim3(N / MF_NB, M / MF_NB), dim3(256), 0, st, K, alpha, A, lda, B, ldb, beta,
                       C, ldc);
}
// Cholesky factor of one 64 x 64 diagonal block in place (lower triangle; the upper is zeroed) and the inverse of
// that factor into Dinv (lower triangular).  One workgroup of 64 threads, thread t owns row t.
__global__ __launch_bounds__(64) void k_potrf_diag(double *__restrict__ A, int lda, double *__restrict__ Dinv,
                                                  int *__restrict__ info, int block) {
    __shared__ double Ls[MF_NB][MF_NB + 1], Xs[MF_NB][MF_NB + 1];
    const int t = threadIdx.x;
    for (int c = 0; c < MF_NB; c++) Ls[t][c] = A[(size_t)t * lda + c];
    __syncthreads();
    for (int j = 0; j < MF_NB; j++) {
        const double d = Ls[j][j];
        if (!(d > 0.0)) {                                  // not positive definite (or NaN)
            if (t == 0) atomicCAS(info, 0, block * MF_NB + j + 1);
            return;
        }
        const double sd = sqrt(d);
        __syncthreads();
        if (t == j) Ls[j][j] = sd;
        if (t > j) Ls[t][j] /= sd;
        __syncthreads();
        if (t > j)
            for (int c = j + 1; c <= t; c++) Ls[t][c] -= Ls[t][j] * Ls[c][j];
        __syncthreads();
    }
    // inverse of the lower-triangular factor: column t of X by forward substitution
    for (int r = 0; r < MF_NB; r++) {
        double v = (r == t) ? 1.0 : 0.0;
        if (r >= t) {
            for (int k = t; k < r; k++) v -= Ls[r][k] * Xs[k][t];
            v /= Ls[r][r];
        } else {
            v = 0.0;
        }
        Xs[r][t] = v;
    }
    __syncthreads();
    for (int c = 0; c < MF_NB; c++) {
        A[(size_t)t * lda + c] = (c <= t) ? Ls[t][c] : 0.0;
        Dinv[(size_t)t * MF_NB + c] = Xs[t][c];
    }
}
__global__ __launch_bounds__(256) void k_zero_upper_blocks(double *__restrict__ A, int np) {
    // after the factorisation the strictly upper block triangle still holds the input: clear it (L is lower)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)np * np) return;
    const int r = (int)(idx / np), c = (int)(idx % np);
    if (c / MF_NB > r / MF_NB) A[idx] = 0.0;
}

// A (np x np, SPD) -> A^-1 in place (full symmetric storage).  X, D: scratch np x np and nb x 64 x 64.
int spd_inverse(hipStream_t st, double *A, int np, double *X, double *D, int *info) {
    const int nb = np / MF_NB;
    (void)hipMemsetAsync(info, 0, sizeof(int), st);
    // 1. A = L L^T, right-looking by block columns
    for (int k = 0; k < nb; k++) {
        double *Akk = A + ((size_t)k * np + k) * MF_NB;
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(64), 0, st, Akk, np, D + (size_t)k * MF_NB * MF_NB, info, k);
        const int rest = np - (k + 1) * MF_NB;
        if (rest > 0) {
            double *Aik = A + ((size_t)(k + 1) * MF_NB) * np + (size_t)k * MF_NB;   // panel below the diagonal block
            // L_ik = A_ik L_kk^-T = A_ik (Dinv_k)^T : out of place through X's first columns, then copied back
            dgemm<0, 1>(st, rest, MF_NB, MF_NB, 1.0, Aik, np, D + (size_t)k * MF_NB * MF_NB, MF_NB, 0.0, X, MF_NB);
            (void)hipMemcpy2DAsync(Aik, sizeof(double) * np, X, sizeof(double) * MF_NB, sizeof(double) * MF_NB, rest,
                                   hipMemcpyDeviceToDevice, st);
            // trailing matrix -= L_panel L_panel^T (whole square: the upper part is never read)
            double *A22 = A + ((size_t)(k + 1) * MF_NB) * np + (size_t)(k + 1) * MF_NB;
            dgemm<0, 1>(st, rest, rest, MF_NB, -1.0, Aik, np, Aik, np, 1.0, A22, np);
        }
    }
    hipLaunchKernelGGL(k_zero_upper_blocks, dim3((unsigned)(((int64_t)np * np + 255) / 256)), dim3(256), 0, st, A, np);
    // 2. X = L^-1 by block rows: X_ii = Dinv_i, X_i,<i = -Dinv_i (L_i,<i X_<i,<i)
    (void)hipMemsetAsync(X, 0, sizeof(double) * (size_t)np * np, st);
    double *T = nullptr;
    if (hipMalloc((void **)&T, sizeof(double) * (size_t)MF_NB * np) != hipSuccess) return PLM_ENOMEM;
    for (int i = 0; i < nb; i++) {
        (void)hipMemcpy2DAsync(X + ((size_t)i * np + i) * MF_NB, sizeof(double) * np, D + (size_t)i * MF_NB * MF_NB,
                               sizeof(double) * MF_NB, sizeof(double) * MF_NB, MF_NB, hipMemcpyDeviceToDevice, st);
        if (i > 0) {
            const int w = i * MF_NB;
            dgemm<0, 0>(st, MF_NB, w, w, 1.0, A + (size_t)i * MF_NB * np, np, X, np, 0.0, T, np);
            dgemm<0, 0>(st, MF_NB, w, MF_NB, -1.0, D + (size_t)i * MF_NB * MF_NB, MF_NB, T, np, 0.0,
                        X + (size_t)i * MF_NB * np, np);
        }
    }
    // 3. A^-1 = X^T X
    dgemm<1, 0>(st, np, np, np, 1.0, X, np, X, np, 0.0, A, np);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(T);
    return e == hipSuccess ? PLM_OK : PLM_EDEVICE;
}

}  // namespace

// Device part of plm_meanfield: fi (L*q raw frequencies) and fij (raw i<j blocks) are device pointers.
// Outputs are device pointers too (any of them may be null): hi [L*q] f64, jfull [L*L*q*q] f64,
// jpairs [pairs*q*q] f32, di [L*L] f64.
int plm_meanfield_device(const float *fi, const float *fij, int L, int q, double pseudo_count, hipStream_t st,
                         double *hi, double *jfull, float *jpairs, double *di) {
    if (q < 2 || q > 32) return plm_fail(PLM_EUNSUPPORTED, "mean-field DCA supports 2..32 states");
    const int n = L * (q - 1), np = (n + MF_NB - 1) / MF_NB * MF_NB;
    double *C = nullptr, *X = nullptr, *D = nullptr;
    int *info = nullptr;
    auto done = [&](int code) {
        void *all[] = {C, X, D, info};
        for (void *b : all)
            if (b) (void)hipFree(b);
        return code;
    };
    if (hipMalloc((void **)&C, sizeof(double) * (size_t)np * np) != hipSuccess ||
        hipMalloc((void **)&X, sizeof(double) * (size_t)np * np) != hipSuccess ||
        hipMalloc((void **)&D, sizeof(double) * (size_t)np * MF_NB) != hipSuccess ||
        hipMalloc((void **)&info, sizeof(int)) != hipSuccess)
        return done(plm_fail(PLM_ENOMEM, "hipMalloc of the %d x %d covariance matrix failed", np, np));
    const int64_t nn = (int64_t)np * np;
    hipLaunchKernelGGL(k_mf_cov, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, fi, fij, L, q, pseudo_count, np, C);
    if (hipGetLastError() != hipSuccess) return done(plm_fail(PLM_EDEVICE, "k_mf_cov launch failed"));
    int rc = spd_inverse(st, C, np, X, D, info);
    if (rc) return done(plm_fail(rc, "the covariance inverse failed on the device"));
    int hinfo = 0;
    if (hipMemcpy(&hinfo, info, sizeof hinfo, hipMemcpyDeviceToHost) != hipSuccess)
        return done(plm_fail(PLM_EDEVICE, "reading the factorisation status failed"));
    if (hinfo)
        return done(plm_fail(PLM_EINVAL, "covariance matrix is not positive definite (pivot %d); raise the pseudo-count",
                             hinfo));
    if (jfull || jpairs) {
        const int64_t tot = (int64_t)L * L * q * q;
        hipLaunchKernelGGL(k_mf_extract, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, C, np, L, q, jfull, jpairs);
    }
    if (hi) hipLaunchKernelGGL(k_mf_fields, dim3(L * q), dim3(256), 0, st, C, np, fi, L, q, pseudo_count, hi);
    if (di) {
        (void)hipMemsetAsync(di, 0, sizeof(double) * (size_t)L * L, st);
        hipLaunchKernelGGL(k_mf_di, dim3((unsigned)((int64_t)L * (L - 1) / 2)), dim3(64), 0, st, C, np, fi, L, q,
                           pseudo_count, (const double *)nullptr, (const double *)nullptr, di);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return done(plm_fail(PLM_EDEVICE, "mean-field kernels failed"));
    return done(PLM_OK);
}

// direct information from given couplings and (regularised) frequencies -- device pointers
int plm_direct_information_device(const double *jdense, const double *rfi, int L, int q, hipStream_t st, double *di) {
    if (q < 2 || q > 32) return plm_fail(PLM_EUNSUPPORTED, "direct information supports 2..32 states");
    (void)hipMemsetAsync(di, 0, sizeof(double) * (size_t)L * L, st);
    hipLaunchKernelGGL(k_mf_di, dim3((unsigned)((int64_t)L * (L - 1) / 2)), dim3(64), 0, st, (const double *)nullptr, 0,
                       (const float *)nullptr, L, q, 0.0, jdense, rfi, di);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return plm_fail(PLM_EDEVICE, "k_mf_di failed");
    return PLM_OK;
}
