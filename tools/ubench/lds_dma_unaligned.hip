// Does LDS-DMA (global_load_lds_dwordx4) take a global source address at any byte alignment, and what does an
// 8-byte-aligned ds_read2_b64 / 4-byte-aligned read of the result cost?  Prints OK/BAD per byte shift and ns per DMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dma_copy(const unsigned char *__restrict__ src, int shift, unsigned char *__restrict__ out,
                                                int reps) {
    __shared__ __attribute__((aligned(16))) unsigned char s[8][4096];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned char *g = src + (size_t)blockIdx.x * 4096 + shift + k + 16 * tid;
            unsigned char *l = &s[k][0] + 16 * 64 * wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)l, 16, 0, 0);
        }
        __syncthreads();
    }
    for (int k = 0; k < 8; ++k)
        reinterpret_cast<v4i *>(out + ((size_t)blockIdx.x * 8 + k) * 4096)[tid] = reinterpret_cast<v4i *>(&s[k][0])[tid];
}

int main() {
    const int nblk = 2048;
    const size_t n = (size_t)nblk * 4096 + 64;
    unsigned char *h = (unsigned char *)malloc(n), *ho = (unsigned char *)malloc((size_t)nblk * 8 * 4096);
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    unsigned char *d, *o;
    hipMalloc(&d, n);
    hipMalloc(&o, (size_t)nblk * 8 * 4096);
    hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(dma_copy, dim3(nblk), dim3(256), 0, 0, d, shift, o, 1);
        hipMemcpy(ho, o, (size_t)nblk * 8 * 4096, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int b = 0; b < nblk; ++b)
            for (int k = 0; k < 8; ++k)
                for (int i = 0; i < 4096; ++i)
                    bad += ho[((size_t)b * 8 + k) * 4096 + i] != h[(size_t)b * 4096 + shift + k + i];
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(dma_copy, dim3(nblk), dim3(256), 0, 0, d, shift, o, 64);
        hipEventRecord(e0);
        hipLaunchKernelGGL(dma_copy, dim3(nblk), dim3(256), 0, 0, d, shift, o, 64);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("base shift %d (copies at +0..+7): %s (%zu bad bytes), %.3f ms for %d x 64 x 8 DMA rounds = %.1f GB/s into LDS\n", shift,
               bad ? "BAD" : "OK", bad, ms, nblk, (double)nblk * 64 * 8 * 4096 / ms / 1e6);
    }
    return 0;
}
