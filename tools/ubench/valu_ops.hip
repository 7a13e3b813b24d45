// Per-opcode VALU throughput on gfx950: ns per wave64 instruction per SIMD at 1..4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OPS(X) \
    X(0, "v_add_u32",        a = a + b;) \
    X(1, "v_lshl_add_u32",   a = (a << 3) + b;) \
    X(2, "v_add3_u32",       a = a + b + c;) \
    X(3, "v_ashrrev_i32",    a = (unsigned)((int)a >> 3) ^ b;) \
    X(4, "v_mul_i32_i24",    a = (unsigned)__mul24((int)a, (int)b);) \
    X(5, "v_mad_i32_i24",    a = (unsigned)(__mul24((int)a, (int)b) + (int)c);) \
    X(6, "v_mul_u32_u24",    a = __umul24(a, b);) \
    X(7, "v_alignbyte_b32",  a = __builtin_amdgcn_alignbyte(a, b, 1);) \
    X(8, "v_perm_b32",       a = __builtin_amdgcn_perm(a, b, 0x06040200u);) \
    X(9, "v_cvt_f32_u32",    a = __float_as_uint((float)a) ^ b;) \
    X(10, "v_cvt_u32_f32",   a = (unsigned)__uint_as_float(a | 0x3f800000u) + b;) \
    X(11, "v_sqrt_f32",      a = __float_as_uint(__builtin_amdgcn_sqrtf(__uint_as_float(a & 0x7fffffffu)));) \
    X(12, "v_mul_f32",       a = __float_as_uint(__uint_as_float(a) * 0.999f);) \
    X(13, "v_fma_f32",       a = __float_as_uint(fmaf(__uint_as_float(a), 1.0001f, __uint_as_float(b)));) \
    X(14, "v_cmp+v_addc",    a = a + (b < a ? 1u : 0u);) \
    X(15, "v_xor_b32",       a = a ^ b;) \
    X(16, "v_lshl_or_b32",   a = (a << 16) | b;) \
    X(17, "v_mul_lo_u32",    a = a * b;) \
    X(18, "v_and_or",        a = (a & 0xffffu) | c;) \
    X(19, "v_cndmask",       a = (c & 1u) ? a : b;)

template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out, unsigned seed) {
    unsigned x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * (2 * i + 3) + seed;
    unsigned b = seed * 7 + threadIdx.x, c = seed + 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {          // 8 independent chains x 8 = 64 instructions per iteration
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned a = x[i];
#define X(ID, NAME, BODY) if (OP == ID) { BODY }
                OPS(X)
#undef X
                x[i] = a;
            }
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, unsigned *out) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    printf("%-18s", name);
    for (int w = 1; w <= 4; w *= 2) {
        const int iters = 2000;
        k<OP><<<256 * w, 256>>>(10, out, 1); hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(s); k<OP><<<256 * w, 256>>>(iters, out, rep + 1); hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
        }
        printf("  w=%d: %5.2f ns", w, best * 1e6 / (iters * 64.0 * w));
    }
    printf("\n");
}

int main() {
    unsigned *out; hipMalloc(&out, 256 * 4 * 256 * 4);
#define X(ID, NAME, BODY) run<ID>(NAME, out);
    OPS(X)
#undef X
    return 0;
}
