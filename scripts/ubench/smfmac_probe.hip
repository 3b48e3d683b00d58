// smfmac_probe.hip -- groundwork for a sparse-MFMA forward GEMM: gfx950's
// v_smfmac_f32_16x16x64_f16 takes a 2:4-compressed A operand (8 halves + 2-bit positions per lane) and a dense B
// (16 halves per lane).  The one-hot alignment operand satisfies 2:4 by construction when 4 consecutive K slots are 4
// states of one site.  This probe (a) measures the sustained rate of the instruction with one-hot-like operands next
// to the dense v_mfma_f32_16x16x32_f16, registers only, every CU busy (same harness as mfma_power.hip), and (b)
// DECODES the operand layout empirically: one non-zero in A at (lane, slot) with one 2-bit field of the index
// register set, B filled with the code of its own (lane, slot) -- the result matrix then names the B element every
// compressed A element multiplies, and the row it lands in.
// Build: hipcc -O3 --offload-arch=gfx950 smfmac_probe.hip -o smfmac_probe ; run: ./smfmac_probe [ms]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- (b) layout decoding, one wave
struct Combo { int lane, slot; unsigned idx; };
__global__ __launch_bounds__(64) void k_decode(const Combo *combos, int n, float *out) {
    const int lane = threadIdx.x;
    half16 b;
#pragma unroll
    for (int s = 0; s < 16; s++) b[s] = (_Float16)(float)(lane * 16 + s + 1);     // 1 .. 1024: exact in f16
    for (int c = 0; c < n; c++) {
        const Combo cb = combos[c];
        half8 a;
#pragma unroll
        for (int s = 0; s < 8; s++) a[s] = (_Float16)((lane == cb.lane && s == cb.slot) ? 1.f : 0.f);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, acc, (int)cb.idx, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) out[((size_t)c * 64 + lane) * 4 + r] = acc[r];
    }
}

// ---------------------------------------------------------------- (a) sustained rate
struct Out { unsigned long long cyc, wall; float sink; };
template <int SPARSE>
__global__ __launch_bounds__(512) void k_rate(const uint4 *__restrict__ adata, const uint4 *__restrict__ bdata, int iters, Out *out) {
    constexpr int NACC = 48;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tid = blockIdx.x * 512 + threadIdx.x;
    half8 a[3];
    uint4 bq[8];
    for (int i = 0; i < 3; i++) { uint4 v = adata[(tid * 3 + i) & 0xffff]; a[i] = *(half8 *)&v; }
    for (int i = 0; i < 8; i++) bq[i] = bdata[(tid * 8 + i) & 0xffff];
    const int idx = 0x4444 ^ ((tid * 2654435761u) & 0x1111);    // pairs (0,1) or (1,1): legal-looking positions
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if constexpr (SPARSE) {
                half16 b;
                *(uint4 *)&b = bq[(2 * i) & 7];
                *((uint4 *)&b + 1) = bq[(2 * i + 1) & 7];
                acc[i] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a[i % 3], b, acc[i], idx, 0, 0);
            } else {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i % 3], *(half8 *)&bq[i & 7], acc[i], 0, 0, 0);
            }
        }
        uint4 t = bq[0];
#pragma unroll
        for (int i = 0; i < 7; i++) bq[i] = bq[i + 1];
        bq[7] = t;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (tid == 0) { out->cyc = c1 - c0; out->wall = w1 - w0; }
    if (s == 12345.678f) out->sink = s;
}
template <int SPARSE> void run_rate(const char *name, const uint4 *a, const uint4 *b, Out *dout, double ms_target) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    float ms = 0;
    for (int pass = 0; pass < 3; pass++) {
        hipEventRecord(e0);
        for (int r = 0; r < (pass == 2 ? 5 : 1); r++) hipLaunchKernelGGL((k_rate<SPARSE>), dim3(256), dim3(512), 0, 0, a, b, iters, dout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * ms_target / ms);
    }
    ms /= 5;
    Out o; hipMemcpy(&o, dout, sizeof o, hipMemcpyDeviceToHost);
    const double nm = 256.0 * 8 * (double)iters * 48;
    const double ops = 2.0 * 16 * 16 * (SPARSE ? 64 : 32);
    printf("%-34s %7.3f ms  %8.1f Tops/s (dense-equivalent)  %5.1f cyc/instr/SIMD  clock %.2f GHz\n", name, ms,
           nm * ops / (ms * 1e-3) / 1e12, (double)o.cyc / ((double)iters * 48 * 2), (double)o.cyc / ((double)o.wall * 10.0));
}

int main(int argc, char **argv) {
    const double ms = argc > 1 ? atof(argv[1]) : 6.0;
    // ---- (a)
    const int n = 65536;
    std::vector<uint4> aoh(n), bd(n);
    srand(1);
    for (int i = 0; i < n; i++) {
        unsigned w[4], s16[4];
        for (int k2 = 0; k2 < 4; k2++) {
            unsigned h0 = ((rand() & 1) << 15) | (0x3800 + (rand() % 0x0c00)), h1 = ((rand() & 1) << 15) | (0x3800 + (rand() % 0x0c00));
            w[k2] = h0 | (h1 << 16);
            s16[k2] = (rand() % 6 == 0) ? 0x3c00u : 0u;      // a 1.0 in the first value of a pair: 4 of 24 states
        }
        bd[i] = make_uint4(w[0], w[1], w[2], w[3]);
        aoh[i] = make_uint4(s16[0], s16[1], s16[2], s16[3]);
    }
    uint4 *da, *db; Out *dout;
    hipMalloc(&da, n * 16); hipMalloc(&db, n * 16); hipMalloc(&dout, sizeof(Out));
    hipMemcpy(da, aoh.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(db, bd.data(), n * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        run_rate<0>("v_mfma_f32_16x16x32_f16  onehot*dense", da, db, dout, ms);
        run_rate<1>("v_smfmac_f32_16x16x64_f16 onehot*dense", da, db, dout, ms);
    }
    // ---- (b)
    std::vector<Combo> combos;
    const int lanes[5] = {0, 16, 32, 48, 5};
    for (int li = 0; li < 5; li++)
        for (int t = 0; t < 8; t++) {
            combos.push_back({lanes[li], t, 0u});
            for (int p = 0; p < 32; p += 2)
                for (unsigned v = 1; v < 4; v++) combos.push_back({lanes[li], t, v << p});
        }
    Combo *dc; float *dres;
    hipMalloc(&dc, combos.size() * sizeof(Combo)); hipMalloc(&dres, combos.size() * 256 * sizeof(float));
    hipMemcpy(dc, combos.data(), combos.size() * sizeof(Combo), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_decode, dim3(1), dim3(64), 0, 0, dc, (int)combos.size(), dres);
    std::vector<float> res(combos.size() * 256);
    hipMemcpy(res.data(), dres, res.size() * sizeof(float), hipMemcpyDeviceToHost);
    // C layout assumed as for the dense 16x16 shapes: lane (n = lane % 16, g = lane / 16) holds C[4 g + r][n]
    printf("# decode: A lane (row m = lane %% 16, k-group ga = lane / 16), compressed slot t, index register value ->\n");
    printf("#         row of C that is non-zero, and for column n the B element (lane gb = ./16, slot) it multiplied\n");
    for (size_t c = 0; c < combos.size(); c++) {
        const Combo &cb = combos[c];
        int row = -1, gb = -1, sb = -1, bad = 0, nz = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 4; r++) {
                const float v = res[(c * 64 + lane) * 4 + r];
                if (v == 0.f) continue;
                nz++;
                const int m = 4 * (lane / 16) + r, nn = lane % 16, code = (int)v - 1, lb = code / 16, s = code % 16;
                if (lb % 16 != nn) bad++;
                if (row < 0) { row = m; gb = lb / 16; sb = s; }
                else if (row != m || gb != lb / 16 || sb != s) bad++;
            }
        // print the baseline (idx 0) of every (lane, slot) and only those index values that change the outcome
        static int base_gb, base_sb;
        if (cb.idx == 0) { base_gb = gb; base_sb = sb; }
        if (cb.idx == 0 || gb != base_gb || sb != base_sb)
            printf("A lane %2d (m %2d ga %d) slot %d idx 0x%08x -> C row %2d, B (gb %d, slot %2d)  nonzeros %d%s\n", cb.lane,
                   cb.lane % 16, cb.lane / 16, cb.slot, cb.idx, row, gb, sb, nz, bad ? "  INCONSISTENT" : "");
    }
    return 0;
}
