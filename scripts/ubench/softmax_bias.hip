// softmax_bias.hip -- is the f32 softmax of k_hpass biased?  A random error of P averages out over the sequences of a
// gradient sum, a systematic one adds up N-fold (|g_hip - g_f64| ~ N L observed at scale).
//   part 1: v_exp_f32 (2^y) against exp2 in f64: mean and rms relative error in units of 2^-24, by argument range
//   part 2: __expf(x) = v_exp_f32(x * log2e) against exp in f64
//   part 3: the softmax as k_hpass computes it (max subtraction, __expf, 1 / Z, P = e * invZ) on random 21-state
//           potentials: mean of (P_a - P_a^f64) / P_a^f64 for the top state and for the others, mean of sum_a P_a - 1
// build: hipcc -O3 --offload-arch=gfx950 softmax_bias.hip -o softmax_bias
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void k_exp2(const float *y, float *out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_exp2f(y[i]);
}
__global__ void k_expf(const float *x, float *out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __expf(x[i]);
}
// variant with the argument product carried exactly: y = x * L2E_HI (rounded), y_lo = its rounding error + x * L2E_LO,
// result = r + r * (y_lo * ln 2)
__global__ void k_expf2(const float *x, float *out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
    const float xv = x[i];
    const float y = xv * L2E_HI;
    const float ylo = __builtin_fmaf(xv, L2E_HI, -y) + xv * L2E_LO;
    const float r = __builtin_amdgcn_exp2f(y);
    out[i] = __builtin_fmaf(r, ylo * 0.693147182464599609375f, r);
}
template <int MODE> __global__ void k_softmax(const float *H, float *P, int n, int q) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *h = H + (size_t)i * q;
    float mx = -INFINITY;
    for (int a = 0; a < q; a++) mx = fmaxf(mx, h[a]);
    float Z = 0.f, e[32];
    for (int a = 0; a < q; a++) {
        const float xv = h[a] - mx;
        if (MODE == 0) e[a] = __expf(xv);
        else {
            const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
            const float y = xv * L2E_HI;
            const float ylo = __builtin_fmaf(xv, L2E_HI, -y) + xv * L2E_LO;
            const float r = __builtin_amdgcn_exp2f(y);
            e[a] = __builtin_fmaf(r, ylo * 0.693147182464599609375f, r);
        }
        Z += e[a];
    }
    const float invZ = 1.f / Z;
    for (int a = 0; a < q; a++) P[(size_t)i * q + a] = e[a] * invZ;
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    const int n = 1 << 20;
    std::vector<float> x(n), out(n);
    float *dx, *dout;
    hipMalloc(&dx, sizeof(float) * n * 32); hipMalloc(&dout, sizeof(float) * n * 32);
    const double U = ldexp(1.0, -24);
    const double ranges[][2] = {{-1, 0}, {-2, -1}, {-4, -2}, {-8, -4}, {-16, -8}, {-24, -16}};
    for (auto &rg : ranges) {
        for (int i = 0; i < n; i++) x[i] = (float)(rg[0] + (rg[1] - rg[0]) * urand());
        hipMemcpy(dx, x.data(), sizeof(float) * n, hipMemcpyHostToDevice);
        for (int which = 0; which < 3; which++) {
            if (which == 0) hipLaunchKernelGGL(k_exp2, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
            if (which == 1) hipLaunchKernelGGL(k_expf, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
            if (which == 2) hipLaunchKernelGGL(k_expf2, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
            hipMemcpy(out.data(), dout, sizeof(float) * n, hipMemcpyDeviceToHost);
            double m = 0, s2 = 0, mxe = 0;
            for (int i = 0; i < n; i++) {
                const double ref = which == 0 ? exp2((double)x[i]) : exp((double)x[i]);
                const double e = ((double)out[i] - ref) / ref / U;
                m += e; s2 += e * e; mxe = fmax(mxe, fabs(e));
            }
            printf("%-28s arg in [%4g,%4g): rel err mean %+8.4f  rms %7.4f  max %7.3f   (units of 2^-24)\n",
                   which == 0 ? "v_exp_f32 vs exp2" : which == 1 ? "__expf vs exp" : "__expf, exact argument", rg[0], rg[1], m / n,
                   sqrt(s2 / n), mxe);
        }
    }
    // softmax on random potentials: one dominant state + spread
    const int q = 21, ns = 1 << 18;
    std::vector<float> H((size_t)ns * q), P((size_t)ns * q);
    for (double sigma : {1.0, 3.0}) {
        for (size_t k = 0; k < H.size(); k++) H[k] = (float)(sigma * nrand());
        hipMemcpy(dx, H.data(), sizeof(float) * H.size(), hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 0) hipLaunchKernelGGL(k_softmax<0>, dim3(ns / 256), dim3(256), 0, 0, dx, dout, ns, q);
            else hipLaunchKernelGGL(k_softmax<1>, dim3(ns / 256), dim3(256), 0, 0, dx, dout, ns, q);
            hipMemcpy(P.data(), dout, sizeof(float) * P.size(), hipMemcpyDeviceToHost);
            double btop = 0, boff = 0, bsum = 0, woff = 0, rtop = 0, roff = 0;
            for (int i = 0; i < ns; i++) {
                double mx = -1e300, Z = 0, p[32];
                int top = 0;
                for (int a = 0; a < q; a++) if (H[(size_t)i * q + a] > mx) { mx = H[(size_t)i * q + a]; top = a; }
                for (int a = 0; a < q; a++) { p[a] = exp((double)H[(size_t)i * q + a] - mx); Z += p[a]; }
                double sum = 0;
                for (int a = 0; a < q; a++) {
                    p[a] /= Z;
                    const double e = (double)P[(size_t)i * q + a] - p[a];     // absolute error of P_a
                    sum += P[(size_t)i * q + a];
                    if (a == top) { btop += e; rtop += e * e; } else { boff += e; roff += e * e; woff += p[a]; }
                }
                bsum += sum - 1.0;
            }
            printf("softmax sigma %.0f %-22s: mean abs err of P_top %+.3e (rms %.2e), of the others (summed) %+.3e (rms %.2e), mean(sum P - 1) %+.3e\n",
                   sigma, mode == 0 ? "as k_hpass (__expf)" : "exact exp argument", btop / ns, sqrt(rtop / ns), boff / ns, sqrt(roff / ns / (q - 1)), bsum / ns);
        }
    }
    return 0;
}
