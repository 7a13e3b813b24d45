// Issue cost of single VALU opcodes on gfx950, measured with inline asm so that hipcc cannot fold or fuse anything:
// 8 independent register chains, 64 instructions per loop iteration, 1 / 2 / 4 waves per SIMD on all 256 CUs.
// Prints ns per wave-instruction per SIMD (wall time / instructions issued by ONE wave... times waves sharing the SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(I) I(0, 1, 2) I(1, 2, 3) I(2, 3, 4) I(3, 4, 5) I(4, 5, 6) I(5, 6, 7) I(6, 7, 0) I(7, 0, 1)
#define BODY(TXT)                                                                                                     \
    for (int i = 0; i < iters; ++i) {                                                                                 \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                               \
            asm volatile(TXT : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(sh), "v"(kv) : "vcc"); \
        }                                                                                                             \
    }
// one asm block = 8 instructions, dst = chain register a, sources = a and the two neighbours
#define I3(OP) "" OP " %0, %0, %1\n\t" OP " %1, %1, %2\n\t" OP " %2, %2, %3\n\t" OP " %3, %3, %4\n\t" OP " %4, %4, %5\n\t" OP " %5, %5, %6\n\t" OP " %6, %6, %7\n\t" OP " %7, %7, %0"
#define I3S(OP) "" OP " %0, %8, %0\n\t" OP " %1, %8, %1\n\t" OP " %2, %8, %2\n\t" OP " %3, %8, %3\n\t" OP " %4, %8, %4\n\t" OP " %5, %8, %5\n\t" OP " %6, %8, %6\n\t" OP " %7, %8, %7"
#define I4(OP) "" OP " %0, %0, %1, %2\n\t" OP " %1, %1, %2, %3\n\t" OP " %2, %2, %3, %4\n\t" OP " %3, %3, %4, %5\n\t" OP " %4, %4, %5, %6\n\t" OP " %5, %5, %6, %7\n\t" OP " %6, %6, %7, %0\n\t" OP " %7, %7, %0, %1"
#define I4S(OP) "" OP " %0, %0, %8, %1\n\t" OP " %1, %1, %8, %2\n\t" OP " %2, %2, %8, %3\n\t" OP " %3, %3, %8, %4\n\t" OP " %4, %4, %8, %5\n\t" OP " %5, %5, %8, %6\n\t" OP " %6, %6, %8, %7\n\t" OP " %7, %7, %8, %0"
#define I2(OP) "" OP " %0, %0\n\t" OP " %1, %1\n\t" OP " %2, %2\n\t" OP " %3, %3\n\t" OP " %4, %4\n\t" OP " %5, %5\n\t" OP " %6, %6\n\t" OP " %7, %7"
#define ICMP "v_cmp_gt_u32 vcc, %0, %1\n\tv_subbrev_co_u32 %0, vcc, %9, %0, vcc\n\tv_cmp_gt_u32 vcc, %2, %3\n\tv_subbrev_co_u32 %2, vcc, %9, %2, vcc\n\t" \
             "v_cmp_gt_u32 vcc, %4, %5\n\tv_subbrev_co_u32 %4, vcc, %9, %4, vcc\n\tv_cmp_gt_u32 vcc, %6, %7\n\tv_subbrev_co_u32 %6, vcc, %9, %6, vcc"

#define OPS(X)                                    \
    X(0, "v_add_u32", I3("v_add_u32"))            \
    X(1, "v_sub_u32", I3("v_sub_u32"))            \
    X(2, "v_xor_b32", I3("v_xor_b32"))            \
    X(3, "v_lshlrev_b32 (sgpr)", I3S("v_lshlrev_b32")) \
    X(4, "v_ashrrev_i32 (sgpr)", I3S("v_ashrrev_i32")) \
    X(5, "v_lshl_add_u32 (sgpr sh)", I4S("v_lshl_add_u32")) \
    X(6, "v_add3_u32", I4("v_add3_u32"))          \
    X(7, "v_mad_i32_i24", I4("v_mad_i32_i24"))    \
    X(8, "v_mad_u32_u24", I4("v_mad_u32_u24"))    \
    X(9, "v_mul_i32_i24", I3("v_mul_i32_i24"))    \
    X(10, "v_mul_u32_u24", I3("v_mul_u32_u24"))   \
    X(11, "v_mul_lo_u32", I3("v_mul_lo_u32"))     \
    X(12, "v_perm_b32", I4("v_perm_b32"))         \
    X(13, "v_alignbit_b32", I4S("v_alignbit_b32")) \
    X(14, "v_bfe_i32 (v,8,24)", "v_bfe_i32 %0, %0, 8, 24\n\tv_bfe_i32 %1, %1, 8, 24\n\tv_bfe_i32 %2, %2, 8, 24\n\tv_bfe_i32 %3, %3, 8, 24\n\tv_bfe_i32 %4, %4, 8, 24\n\tv_bfe_i32 %5, %5, 8, 24\n\tv_bfe_i32 %6, %6, 8, 24\n\tv_bfe_i32 %7, %7, 8, 24") \
    X(15, "v_cvt_f32_u32", I2("v_cvt_f32_u32"))   \
    X(16, "v_cvt_f32_i32", I2("v_cvt_f32_i32"))   \
    X(17, "v_cvt_u32_f32", I2("v_cvt_u32_f32"))   \
    X(18, "v_sqrt_f32", I2("v_sqrt_f32"))         \
    X(19, "v_rsq_f32", I2("v_rsq_f32"))           \
    X(20, "v_add_f32", I3("v_add_f32"))           \
    X(21, "v_mul_f32", I3("v_mul_f32"))           \
    X(22, "v_fma_f32", I4("v_fma_f32"))           \
    X(23, "v_cmp+v_subbrev (pair)", ICMP)         \
    X(24, "v_and_or_b32", I4("v_and_or_b32"))     \
    X(25, "v_lshl_or_b32 (sgpr sh)", I4S("v_lshl_or_b32")) \
    X(26, "v_mov_b32", I2("v_mov_b32"))           \
    X(27, "v_cvt_f16_f32", I2("v_cvt_f16_f32"))   \
    X(28, "v_sqrt_f16", I2("v_sqrt_f16"))         \
    X(29, "v_mad_u32_u16", I4("v_mad_u32_u16"))   \
    X(30, "v_mad_i32_i16", I4("v_mad_i32_i16"))   \
    X(31, "v_add_lshl_u32 (sgpr sh)", "v_add_lshl_u32 %0, %0, %1, %8\n\tv_add_lshl_u32 %1, %1, %2, %8\n\tv_add_lshl_u32 %2, %2, %3, %8\n\tv_add_lshl_u32 %3, %3, %4, %8\n\tv_add_lshl_u32 %4, %4, %5, %8\n\tv_add_lshl_u32 %5, %5, %6, %8\n\tv_add_lshl_u32 %6, %6, %7, %8\n\tv_add_lshl_u32 %7, %7, %0, %8") \
    X(32, "v_sad_u32", I4("v_sad_u32"))           \
    X(33, "v_mul_hi_i32_i24", I3("v_mul_hi_i32_i24"))

template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out, unsigned seed, int sh) {
    unsigned x0 = threadIdx.x + seed, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3, x4 = x0 * 11 + 4, x5 = x0 * 13 + 5, x6 = x0 * 17 + 6, x7 = x0 * 19 + 7;
    unsigned kv = 0x4B000000u;
#define X(N, NAME, TXT) if (OP == N) { BODY(TXT) }
    OPS(X)
#undef X
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <int OP>
void run(const char *name, int per_block) {
    unsigned *out;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 2000;
    printf("%-28s", name);
    for (int w = 1; w <= 4; w *= 2) {
        hipEvent_t s, e;
        hipEventCreate(&s); hipEventCreate(&e);
        k<OP><<<256 * w, 256>>>(10, out, 1, 8);
        hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(s);
            k<OP><<<256 * w, 256>>>(iters, out, rep, 8);
            hipEventRecord(e);
            hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            if (ms < best) best = ms;
        }
        // one SIMD runs w waves, each iters * 8 blocks * per_block instructions
        printf("  w=%d: %6.2f ns", w, best * 1e6 / ((double)iters * 8 * per_block * w));
    }
    printf("   (per wave-instruction, per SIMD)\n");
    hipFree(out);
}

int main() {
#define X(N, NAME, TXT) run<N>(NAME, 8);
    OPS(X)
#undef X
    return 0;
}
