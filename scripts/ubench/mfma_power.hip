// mfma_power.hip -- sustained MFMA throughput on MI355X for the instruction shapes the one-hot GEMM
// kernels could use, with the operand statistics they would see (A = sparse 0/1, B = dense values).
// Registers only (no LDS / HBM): 8 waves per CU, every CU busy for a few ms, so the result is the
// power-management ceiling of each shape rather than its issue rate.   Build: hipcc -O3
// --offload-arch=gfx950 mfma_power.hip -o mfma_power ;  run: ./mfma_power [ms]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct Out { unsigned long long cyc, wall; float sink; };

template <int MODE> struct Tr;
// MODE 0: f16 16x16x32   1: f16 32x32x16   2: i8 16x16x64   3: i8 32x32x32
template <> struct Tr<0> { typedef f32x4 C; typedef half8 AB; static constexpr int NACC = 48; static constexpr double OPS = 2.0 * 16 * 16 * 32;
    static __device__ C mma(AB a, AB b, C c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };
template <> struct Tr<1> { typedef f32x16 C; typedef half8 AB; static constexpr int NACC = 12; static constexpr double OPS = 2.0 * 32 * 32 * 16;
    static __device__ C mma(AB a, AB b, C c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); } };
template <> struct Tr<2> { typedef i32x4 C; typedef i32x4 AB; static constexpr int NACC = 48; static constexpr double OPS = 2.0 * 16 * 16 * 64;
    static __device__ C mma(AB a, AB b, C c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); } };
template <> struct Tr<3> { typedef i32x16 C; typedef i32x4 AB; static constexpr int NACC = 12; static constexpr double OPS = 2.0 * 32 * 32 * 32;
    static __device__ C mma(AB a, AB b, C c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); } };

// DEP = 0: every MFMA of an iteration has its own accumulator.  DEP = d > 0: the accumulators are visited
// in groups of d, each group twice in a row (a0 a1 .. a0 a1 ..): the dependent MFMA follows d instructions
// later, the pattern of the hi/lo planes in k_fwd (d = 2) and k_bwd (d = 7).
template <int MODE, int DEP>
__global__ __launch_bounds__(512) void k(const uint4 *__restrict__ adata, const uint4 *__restrict__ bdata, int iters, Out *out) {
    typedef Tr<MODE> T;
    typename T::C acc[T::NACC];
#pragma unroll
    for (int i = 0; i < T::NACC; i++)
        for (int r = 0; r < (int)(sizeof(typename T::C) / 4); r++) acc[i][r] = 0;
    typename T::AB a[3], b[4];
    const int tid = blockIdx.x * 512 + threadIdx.x;
    for (int i = 0; i < 3; i++) { uint4 v = adata[(tid * 3 + i) & 0xffff]; a[i] = *(typename T::AB *)&v; }
    for (int i = 0; i < 4; i++) { uint4 v = bdata[(tid * 4 + i) & 0xffff]; b[i] = *(typename T::AB *)&v; }
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (DEP == 0) {
#pragma unroll
            for (int i = 0; i < T::NACC; i++) acc[i] = T::mma(a[i % 3], b[i & 3], acc[i]);
        } else {
#pragma unroll
            for (int g0 = 0; g0 + DEP <= T::NACC / 2; g0 += DEP) {
#pragma unroll
                for (int i = 0; i < DEP; i++) acc[g0 + i] = T::mma(a[i % 3], b[0], acc[g0 + i]);
#pragma unroll
                for (int i = 0; i < DEP; i++) acc[g0 + i] = T::mma(a[i % 3], b[1], acc[g0 + i]);
            }
        }
        // rotate the B registers so the operands change like streamed fragments do
        typename T::AB t = b[0]; b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < T::NACC; i++)
        for (int r = 0; r < (int)(sizeof(typename T::C) / 4); r++) s += (float)acc[i][r];
    if (tid == 0) { out->cyc = c1 - c0; out->wall = w1 - w0; }
    if (s == 12345.678f) out->sink = s;
}

template <int MODE, int DEP = 0> void run(const char *name, const uint4 *a, const uint4 *b, Out *dout, double ms_target, const char *what) {
    typedef Tr<MODE> T;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    float ms = 0;
    for (int pass = 0; pass < 3; pass++) {
        hipEventRecord(e0);
        for (int r = 0; r < (pass == 2 ? 5 : 1); r++) hipLaunchKernelGGL((k<MODE, DEP>), dim3(256), dim3(512), 0, 0, a, b, iters, dout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * ms_target / ms);
    }
    ms /= 5;
    Out o; hipMemcpy(&o, dout, sizeof o, hipMemcpyDeviceToHost);
    const double per_iter = DEP == 0 ? T::NACC : 2 * DEP * ((T::NACC / 2) / DEP);
    const double nm = 256.0 * 8 * (double)iters * per_iter;
    printf("%-14s %-10s %7.3f ms  %8.1f Tops/s   %5.1f cyc/MFMA/SIMD  clock %.2f GHz (s_memtime/wall_clock)\n", name, what, ms,
           nm * T::OPS / (ms * 1e-3) / 1e12, (double)o.cyc / ((double)iters * per_iter * 2), (double)o.cyc / ((double)o.wall * 10.0) );
}

int main(int argc, char **argv) {
    const double ms = argc > 1 ? atof(argv[1]) : 6.0;
    const int n = 65536;
    std::vector<uint4> az(n), ad(n), bz(n), bf(n), bi(n), asp16(n), asp8(n);
    srand(1);
    for (int i = 0; i < n; i++) {
        az[i] = bz[i] = make_uint4(0, 0, 0, 0);
        unsigned w[4], v[4], s16[4], s8[4];
        for (int k2 = 0; k2 < 4; k2++) {
            // dense f16: two random values in [-2, 2) (sign + exponent 0x3c00..0x4000 + random mantissa)
            unsigned h0 = ((rand() & 1) << 15) | (0x3800 + (rand() % 0x0c00)), h1 = ((rand() & 1) << 15) | (0x3800 + (rand() % 0x0c00));
            w[k2] = h0 | (h1 << 16);
            v[k2] = (unsigned)rand() ^ ((unsigned)rand() << 16);       // dense int8
            unsigned o16 = 0, o8 = 0;
            for (int e = 0; e < 2; e++) if (rand() % 21 == 0) o16 |= 0x3c00u << (16 * e);   // one-hot f16 1.0, density 1/21
            for (int e = 0; e < 4; e++) if (rand() % 21 == 0) o8 |= 0x01u << (8 * e);      // one-hot int8 1
            s16[k2] = o16; s8[k2] = o8;
        }
        bf[i] = make_uint4(w[0], w[1], w[2], w[3]); bi[i] = make_uint4(v[0], v[1], v[2], v[3]);
        asp16[i] = make_uint4(s16[0], s16[1], s16[2], s16[3]); asp8[i] = make_uint4(s8[0], s8[1], s8[2], s8[3]);
    }
    uint4 *d[6]; Out *dout;
    const std::vector<uint4> *src[6] = {&az, &bf, &bi, &asp16, &asp8, &bz};
    for (int i = 0; i < 6; i++) { hipMalloc(&d[i], n * 16); hipMemcpy(d[i], src[i]->data(), n * 16, hipMemcpyHostToDevice); }
    hipMalloc(&dout, sizeof(Out));
    if (argc > 2) {   // dependency-distance study on the one-hot case
        run<0, 0>("f16 16x16x32", d[3], d[1], dout, ms, "indep");
        run<0, 1>("f16 16x16x32", d[3], d[1], dout, ms, "dep d=1");
        run<0, 2>("f16 16x16x32", d[3], d[1], dout, ms, "dep d=2");
        run<0, 4>("f16 16x16x32", d[3], d[1], dout, ms, "dep d=4");
        run<0, 7>("f16 16x16x32", d[3], d[1], dout, ms, "dep d=7");
        run<0, 12>("f16 16x16x32", d[3], d[1], dout, ms, "dep d=12");
        return 0;
    }
    for (int rep = 0; rep < 2; rep++) {
        run<0>("f16 16x16x32", d[0], d[5], dout, ms, "zeros");
        run<0>("f16 16x16x32", d[3], d[1], dout, ms, "onehot*dense");
        run<0>("f16 16x16x32", d[1], d[1], dout, ms, "dense*dense");
        run<1>("f16 32x32x16", d[0], d[5], dout, ms, "zeros");
        run<1>("f16 32x32x16", d[3], d[1], dout, ms, "onehot*dense");
        run<1>("f16 32x32x16", d[1], d[1], dout, ms, "dense*dense");
        run<2>("i8 16x16x64", d[0], d[5], dout, ms, "zeros");
        run<2>("i8 16x16x64", d[4], d[2], dout, ms, "onehot*dense");
        run<2>("i8 16x16x64", d[2], d[2], dout, ms, "dense*dense");
        run<3>("i8 32x32x32", d[0], d[5], dout, ms, "zeros");
        run<3>("i8 32x32x32", d[4], d[2], dout, ms, "onehot*dense");
        run<3>("i8 32x32x32", d[2], d[2], dout, ms, "dense*dense");
    }
    return 0;
}
