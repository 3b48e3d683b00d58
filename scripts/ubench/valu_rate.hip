// valu_rate.hip -- issue rate of the integer VALU instructions k_reweight_reg is made of: how many
// cycles a wave64 instruction occupies its SIMD for.  The kernel's "VALU floor" (3 instructions per 4 alignment sites) was
// priced at 2 cycles per instruction in rounds 1-4; this measures it, per instruction and for the kernel's triple, at 1, 2
// and 4 waves per SIMD, with every operand in registers (no memory traffic at all).
//   build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32;
#define REP16(X) X X X X X X X X X X X X X X X X

// MODE 0: v_xad_u32 (VGPR, SGPR, VGPR) + v_and_b32 + v_bcnt_u32_b32 -- the triple of k_reweight_reg, 16 independent words
// MODE 1: v_and_b32 only        MODE 2: v_bcnt_u32_b32 only      MODE 3: v_xad_u32 with the SGPR operand only
// MODE 4: v_xad_u32 with VGPR operands only                      MODE 5: v_and_or_b32 only
// MODE 6: the merged form: 3 x (v_xad + v_and_or) + 1 v_bcnt per 3 words (7 instructions per 12 sites)
// MODE 7: v_add_u32 only (reference: the plainest integer instruction)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(u32 *out, int iters, u32 seed) {
    u32 m[16], acc = 0, c7f = 0x7f7f7f7fu, k80 = 0x80808080u;
    const u32 sg = __builtin_amdgcn_readfirstlane(seed);
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = threadIdx.x * 2654435761u + k * 40503u + seed;
    asm volatile("" : "+v"(c7f), "+v"(k80));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u32 y;
            if (MODE == 0) {
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(y) : "v"(m[k]), "s"(sg), "v"(c7f));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(y) : "v"(y), "v"(k80));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(y));
            } else if (MODE == 1) {
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(y) : "v"(m[k]), "v"(k80));
                m[k] = y;
            } else if (MODE == 2) {
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(m[k]) : "v"(k80));
            } else if (MODE == 3) {
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(y) : "v"(m[k]), "s"(sg), "v"(c7f));
                m[k] = y;
            } else if (MODE == 4) {
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(y) : "v"(m[k]), "v"(k80), "v"(c7f));
                m[k] = y;
            } else if (MODE == 5) {
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(y) : "v"(m[k]), "v"(k80), "v"(c7f));
                m[k] = y;
            } else if (MODE == 7) {
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(y) : "v"(m[k]), "v"(k80));
                m[k] = y;
            }
        }
        if (MODE == 6) {
#pragma unroll
            for (int k = 0; k + 2 < 16; k += 3) {   // 15 of the 16 words: 5 groups of 3
                u32 y0, y1, y2, z;
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(y0) : "v"(m[k]), "s"(sg), "v"(c7f));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(z) : "v"(y0), "v"(k80));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(y1) : "v"(m[k + 1]), "s"(sg), "v"(c7f));
                asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(z) : "v"(y1), "v"(k80));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(y2) : "v"(m[k + 2]), "s"(sg), "v"(c7f));
                asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(z) : "v"(y2), "v"(k80));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(z));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) acc += m[k];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> static void run(const char *name, int instr_per_iter, u32 *out, int ncu) {
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD: blocks of 4 waves, wps blocks per CU
        const int blocks = ncu * wps;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1u);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        // each SIMD runs wps waves; instructions it issues = wps * iters * instr_per_iter
        const double per_simd = (double)wps * iters * instr_per_iter;
        printf("%-44s waves/SIMD %d: %8.3f ms  %6.3f ns per wave-instruction per SIMD (2.4 GHz: %5.2f cycles)\n", name, wps, ms,
               ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.gcnArchName, ncu, p.clockRate / 1000);
    u32 *out;
    hipMalloc(&out, (size_t)ncu * 8 * 256 * 4);
    run<7>("v_add_u32 (VGPR, VGPR)", 16, out, ncu);
    run<1>("v_and_b32 (VGPR, VGPR)", 16, out, ncu);
    run<2>("v_bcnt_u32_b32 (accumulating)", 16, out, ncu);
    run<3>("v_xad_u32 (VGPR, SGPR, VGPR)", 16, out, ncu);
    run<4>("v_xad_u32 (VGPR, VGPR, VGPR)", 16, out, ncu);
    run<5>("v_and_or_b32", 16, out, ncu);
    run<0>("triple xad+and+bcnt (k_reweight_reg)", 48, out, ncu);
    run<6>("merged 3x(xad+and_or)+bcnt, 15 words", 35, out, ncu);
    return 0;
}
