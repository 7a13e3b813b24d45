#include <hip/hip_runtime.h>
#include <stdio.h>
// VALU-only throughput for the integer op mix of the Gabor epilogue, 1..4 waves per SIMD.
__global__ __launch_bounds__(256) void k(int iters, int *out, int seed) {
    unsigned x0 = threadIdx.x + seed, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int v = 0; v < 64; ++v) {
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
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
int main() {
    int *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int w = 1; w <= 4; ++w) {
        const int iters = 4000;
        k<<<256 * w, 256>>>(10, out, 1); hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(s); k<<<256 * w, 256>>>(iters, out, rep); hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
        }
        printf("valu only waves/SIMD=%d: %.3f ms -> %.2f ns per VALU wave-instr per SIMD\n", w, best, best * 1e6 / (iters * 64.0 * w));
    }
    return 0;
}
