// Microbenchmark: how much integer VALU work overlaps with i8 MFMAs on one SIMD (gfx950)?
// Variants: MFMA only, VALU only, both interleaved; 1 or 2 waves per SIMD (grid = 256 CUs x (4|8) waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int NMFMA, int NVALU>   // per inner iteration
__global__ __launch_bounds__(256) void k(int iters, int *out, int seed) {
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, 5, 7, 9};
    v16i acc0 = {0}, acc1 = {0};
    v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    unsigned x0 = threadIdx.x + seed, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) {
            if (SHAPE == 0) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
            } else {
                if (m & 1) c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < NVALU; ++v) {   // independent integer ops (8 chains)
                switch (v & 7) {
                    case 0: x0 = __builtin_amdgcn_alignbyte(x0, x1, 1); break;
                    case 1: x1 = (x1 << 8) + x2; break;
                    case 2: x2 = __mul24((int)x2, (int)x3); break;
                    case 3: x3 = (unsigned)((int)x3 >> 3) + x4; break;
                    case 4: x4 = __umul24(x4, x5) + x6; break;
                    case 5: x5 = x5 + x6 + x7; break;
                    case 6: x6 = (unsigned)(float)x6; break;
                    case 7: x7 = (x7 > x0) ? x7 + 1 : x7; break;
                }
            }
        }
    }
    int r = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    for (int e = 0; e < 16; ++e) r ^= acc0[e] ^ acc1[e];
    for (int e = 0; e < 4; ++e) r ^= c0[e] ^ c1[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int SHAPE, int NM, int NV>
void run(const char *name, int wgs_per_cu) {
    int *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 2000;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<SHAPE, NM, NV><<<256 * wgs_per_cu, 256>>>(10, out, 1);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(s);
        k<SHAPE, NM, NV><<<256 * wgs_per_cu, 256>>>(iters, out, rep);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    // per SIMD: wgs_per_cu waves, each iters * (NM mfma + NM*NV valu)
    double ns_per_iter_per_simd = best * 1e6 / iters;       // wall ns for one iteration of every resident wave
    printf("%-34s waves/SIMD=%d  NM=%2d NV/mfma=%2d : %.3f ms  -> %.1f ns per iter (%.1f ns per MFMA-slot per wave-pair)\n", name,
           wgs_per_cu, NM, NV, best, ns_per_iter_per_simd, ns_per_iter_per_simd / (NM ? NM : 1));
    hipFree(out);
}

int main() {
    // MFMA only
    run<0, 8, 0>("32x32x32 mfma only", 1);
    run<0, 8, 0>("32x32x32 mfma only", 2);
    run<1, 8, 0>("16x16x64 mfma only", 1);
    run<1, 8, 0>("16x16x64 mfma only", 2);
    // interleaved with VALU
    run<0, 8, 2>("32x32x32 + 2 valu/mfma", 1);
    run<0, 8, 4>("32x32x32 + 4 valu/mfma", 1);
    run<0, 8, 6>("32x32x32 + 6 valu/mfma", 1);
    run<0, 8, 8>("32x32x32 + 8 valu/mfma", 1);
    run<0, 8, 12>("32x32x32 + 12 valu/mfma", 1);
    run<0, 8, 4>("32x32x32 + 4 valu/mfma", 2);
    run<0, 8, 8>("32x32x32 + 8 valu/mfma", 2);
    run<0, 8, 12>("32x32x32 + 12 valu/mfma", 2);
    run<1, 8, 2>("16x16x64 + 2 valu/mfma", 1);
    run<1, 8, 4>("16x16x64 + 4 valu/mfma", 1);
    run<1, 8, 6>("16x16x64 + 6 valu/mfma", 1);
    run<1, 8, 4>("16x16x64 + 4 valu/mfma", 2);
    run<1, 8, 6>("16x16x64 + 6 valu/mfma", 2);
    return 0;
}
