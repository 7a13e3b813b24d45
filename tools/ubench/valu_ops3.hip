// Per-opcode VALU throughput, third set (round 2): candidates for the Gabor epilogue. Same harness as valu_ops2.hip:
// operands rotate between 8 live registers so nothing folds; prints ns per wave-instruction-group per SIMD at 1, 2, 4
// waves per SIMD. "epi old" / "epi new" are whole per-output epilogues (SPEC.md §3: recombine, shift, |.|^2, isqrt).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned isqrt_old(unsigned n) {
    const unsigned bits = __float_as_uint(__builtin_amdgcn_sqrtf((float)n) + 8388608.0f);
    const int d = (int)(n - __umul24(bits, bits));
    return bits - 0x4B000000u + (unsigned)(d >> 31);
}
__device__ __forceinline__ unsigned isqrt_new(unsigned n) {
    const unsigned bits = __float_as_uint(__builtin_amdgcn_sqrtf((float)n) + 8388608.0f);
    const unsigned sq = __umul24(bits, bits);
    unsigned q;
    asm("v_cmp_gt_u32 vcc, %1, %2\n\tv_subbrev_co_u32 %0, vcc, %4, %3, vcc" : "=v"(q) : "v"(sq), "v"(n), "v"(bits), "v"(0x4B000000u) : "vcc");
    return q;
}
__device__ __forceinline__ unsigned isqrt_tail(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_cmp_gt_u32 vcc, %1, %2\n\tv_subbrev_co_u32 %0, vcc, %3, %1, vcc" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}
__device__ __forceinline__ unsigned dot2_vop3p(unsigned a) {
    unsigned d;
    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(d) : "v"(a));
    return d;
}
#define OPS(X) \
    X(0, "v_add_u32",          a = a + b;) \
    X(1, "v_mad_i32_i24",      a = (unsigned)(__mul24((int)a, (int)b) + (int)c);) \
    X(2, "mul24(x,256)+y",     a = (unsigned)(__mul24((int)a, 256) + (int)b);) \
    X(3, "mul24(x,256)+y+z",   a = (unsigned)(__mul24((int)a, 256) + (int)b + (int)c);) \
    X(4, "sdot2(a,a,0)",       a = (unsigned)__builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, a), 0, false) ^ b;) \
    X(5, "sdot2(a,a,b)",       a = (unsigned)__builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, a), (int)b, false);) \
    X(6, "v_dot2 vop3p asm",   a = dot2_vop3p(a) ^ b;) \
    X(7, "mul24 a*a + b*b",    a = (unsigned)(__mul24((int)(a & 0xffff), (int)(a & 0xffff))) + (unsigned)__mul24((int)(b & 0xffff), (int)(b & 0xffff));) \
    X(8, "cmp+subbrev asm",    { unsigned d; d = isqrt_tail(a, b, c); a = d; }) \
    X(9, "sub ashr31 add3",    { const int d = (int)(a - b); a = c - 0x4B000000u + (unsigned)(d >> 31); }) \
    X(10, "v_perm (sgpr sel)", a = __builtin_amdgcn_perm(a, b, 0x06050201u);) \
    X(11, "ashr x2",           a = (unsigned)((int)a >> 8) ^ (unsigned)((int)b >> 8);) \
    X(12, "isqrt old (7)",     a = isqrt_old(a & 0x7fffffffu) + b;) \
    X(13, "isqrt new (6)",     a = isqrt_new(a & 0x7fffffffu) + b;) \
    X(14, "epi old",           { const int vr = (__mul24((int)a, 256) + (int)b + (int)c) >> 8; const int vi = (__mul24((int)c, 256) + (int)a) >> 8; \
                                 a = isqrt_old(((unsigned)__mul24(vr & 0x7fff, vr & 0x7fff) + (unsigned)__mul24(vi & 0x7fff, vi & 0x7fff))); }) \
    X(15, "epi new",           { const int vr = __mul24((int)a, 256) + (int)b + (int)c; const int vi = __mul24((int)c, 256) + (int)a; \
                                 const unsigned pk = __builtin_amdgcn_perm((unsigned)vr, (unsigned)vi, 0x06050201u) & 0x7fff7fffu; \
                                 a = isqrt_new((unsigned)__builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, pk), __builtin_bit_cast(v2s, pk), 0, false)); }) \
    X(16, "epi perm+mul24+old", { const int vr = __mul24((int)a, 256) + (int)b + (int)c; const int vi = __mul24((int)c, 256) + (int)a; \
                                 const int ar = (vr >> 8) & 0x7fff; const int ai = (vi >> 8) & 0x7fff; \
                                 a = isqrt_old((unsigned)__mul24(ar, ar) + (unsigned)__mul24(ai, ai)); }) \
    X(17, "v_sqrt_f32",        a = __float_as_uint(__builtin_amdgcn_sqrtf(__uint_as_float((a & 0x3fffffffu) | 0x10000000u))) + b;) \
    X(18, "v_cvt_f32_u32",     a = __float_as_uint((float)(a & 0x7fffffffu)) + b;) \
    X(19, "v_lshlrev_b32",     a = (a << 8) ^ b;) \
    X(20, "v_alignbit 8",      a = __builtin_amdgcn_alignbit(a, b, 8);) \
    X(21, "v_alignbit 16",     a = __builtin_amdgcn_alignbit(a, b, 16);)

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
    printf("%-22s", name);
    for (int w = 1; w <= 4; w *= 2) {
        const int iters = 1000;
        k<OP><<<256 * w, 256>>>(10, out, 1); hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(s); k<OP><<<256 * w, 256>>>(iters, out, rep + 1); hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
        }
        printf("  w=%d: %6.2f ns", w, best * 1e6 / (iters * 64.0 * w));
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
