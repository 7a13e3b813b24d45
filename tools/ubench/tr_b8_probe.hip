// What does ds_read_b64_tr_b8 deliver? (round 6; the guide documents the 16-bit form only: cdna_hip_programming.md T10.)
// Every lane supplies an address of its own (lane * PITCH, 8-byte aligned); LDS byte a holds a & 255 in pass 0 and a >> 8 in pass 1,
// so the two passes together give the SOURCE ADDRESS of every byte a lane receives: printed as (supplying lane, byte offset).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v2i __attribute__((ext_vector_type(2)));
constexpr int PITCH = 264;
template <int WHICH>   // 0: tr_b8, 1: tr_b16 (the documented one, as a check of the decoding)
__global__ void probe(unsigned *out, int pass) {
    __shared__ __attribute__((aligned(16))) unsigned char s[64 * PITCH + 64];
    for (int i = threadIdx.x; i < 64 * PITCH + 64; i += 64) s[i] = (unsigned char)(pass ? i >> 8 : i);
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s + threadIdx.x * PITCH;
    v2i r;
    if (WHICH == 0) asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    else asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[2 * threadIdx.x] = r[0];
    out[2 * threadIdx.x + 1] = r[1];
}
int main() {
    unsigned *d, h[2][128];
    hipMalloc(&d, sizeof h[0]);
    for (int which = 0; which < 2; ++which) {
        for (int pass = 0; pass < 2; ++pass) {
            if (which == 0) probe<0><<<1, 64>>>(d, pass); else probe<1><<<1, 64>>>(d, pass);
            hipMemcpy(h[pass], d, sizeof h[0], hipMemcpyDeviceToHost);
        }
        printf("%s: lane -> its 8 bytes as (supplying lane : byte offset inside that lane's 8 bytes)\n", which ? "ds_read_b64_tr_b16" : "ds_read_b64_tr_b8");
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) {
                const unsigned lo = (h[0][2 * l + b / 4] >> (8 * (b % 4))) & 255, hi = (h[1][2 * l + b / 4] >> (8 * (b % 4))) & 255;
                const unsigned a = hi << 8 | lo;
                printf(" %2u:%u", a / PITCH, a % PITCH);
            }
            printf("\n");
        }
    }
    return 0;
}
