// Per-opcode VALU throughput, second set; operands rotate between 8 live registers so nothing folds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OPS(X) \
    X(0, "v_add_u32",          a = a + b;) \
    X(1, "v_xor_b32",          a = a ^ b;) \
    X(2, "v_alignbit_b32",     a = __builtin_amdgcn_alignbit(a, b, 8);) \
    X(3, "v_alignbyte_b32",    a = __builtin_amdgcn_alignbyte(a, b, 1);) \
    X(4, "v_mul_u32_u24",      a = __umul24(a, b);) \
    X(5, "v_mad_u32_u24",      a = __umul24(a, b) + a;) \
    X(6, "mul24 + add",        a = __umul24(a, b) + c;) \
    X(7, "v_lshl_add_u32",     a = (a << 8) + b;) \
    X(8, "v_add3_u32",         a = a + b + c;) \
    X(9, "v_ashrrev_i32",      a = (unsigned)((int)a >> 5) + b;) \
    X(10, "v_bfe_i32",         a = (unsigned)(((int)(a << 5)) >> 16) + b;) \
    X(11, "sub ashr31 add",    { unsigned d = a - b; a = b + (unsigned)((int)d >> 31); }) \
    X(12, "cmp + addc",        a = b + (a <= b ? 1u : 0u);) \
    X(13, "v_min_u32",         a = min(a, b) + c;) \
    X(14, "v_cvt_f32_u32",     a = __float_as_uint((float)(a & 0x7fffffffu)) + b;) \
    X(15, "magic u23->f32",    a = __float_as_uint(__uint_as_float((a & 0x7fffffu) | 0x4B000000u) - 8388608.0f) + b;) \
    X(16, "v_cvt_u32_f32",     a = (unsigned)__uint_as_float((a & 0x0fffffffu) | 0x40000000u) + b;) \
    X(17, "magic f32->u (fma)", a = (__float_as_uint(fmaf(__uint_as_float((a & 0x0fffffffu) | 0x40000000u), 0.99999905f, 8388607.5f)) - 0x4B000000u) + b;) \
    X(18, "v_sqrt_f32",        a = __float_as_uint(__builtin_amdgcn_sqrtf(__uint_as_float((a & 0x3fffffffu) | 0x10000000u))) + b;) \
    X(19, "v_rsq_f32",         a = __float_as_uint(__builtin_amdgcn_rsqf(__uint_as_float((a & 0x3fffffffu) | 0x10000000u))) + b;) \
    X(20, "v_fma_f32",         a = __float_as_uint(fmaf(__uint_as_float(a), 1.0001f, __uint_as_float(b)));) \
    X(21, "v_perm_b32",        a = __builtin_amdgcn_perm(a, b, 0x06040200u);) \
    X(22, "v_cndmask",         a = (b & 1u) ? a : c;) \
    X(23, "i64 lshl_add",      { unsigned long long t = ((unsigned long long)a << 8) + ((unsigned long long)b << 16) + c; a = (unsigned)t ^ (unsigned)(t >> 32); }) \
    X(24, "i64 cmp+sel",       { long long p = ((long long)(int)a << 20) + b; long long q = ((long long)(int)c << 20) + a; a = (p < q) ? b : c; }) \
    X(25, "v_cvt_f32_ubyte0",  a = __float_as_uint((float)(a & 255u)) + b;) \
    X(26, "v_lshlrev_b32",     a = (a << 3) ^ b;) \
    X(27, "v_and_or_b32",      a = (a & 0xffffu) | b;) \
    X(28, "v_lshl_or_b32",     a = (a << 16) | (b & 0xffffu);)

template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out, unsigned seed) {
    unsigned x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * (2 * i + 3) + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned a = x[i];
                const unsigned b = x[(i + 3) & 7], c = x[(i + 5) & 7];
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
    printf("%-20s", name);
    for (int w = 1; w <= 4; w *= 2) {
        const int iters = 1000;
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
