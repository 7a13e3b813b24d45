// What does ONE VALU opcode cost beside a running i8 MFMA stream on gfx950?  (round 3)
// Every instruction of the loop is inline asm (asm volatile keeps program order), so the stream is exactly:
//     7 x { v_mfma_i32_32x32x32_i8 ; G x OP }      per iteration,
// with the MFMAs either one dependent chain (NACC = 1: every MFMA accumulates into the previous one's tuple, the shape of
// gabor_mfma_kernel's tile-major chains) or two independent chains taking turns (NACC = 2, round 2's kernel).
// 2 waves per SIMD on every CU (grid 512 x 256) unless WAVES says otherwise. Prints ns per MFMA slot (= wall time /
// (iterations x 7) per SIMD, both waves' MFMAs counted) and the implied cost per VALU instruction beyond the bare stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MFMA(ACC) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define MFMA16(ACC) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))

// one group = 4 instructions on 8 rotating registers (two groups cover all eight chains)
#define G3(OP, A, B, C, D) OP " %" #A ", %" #A ", %" #B "\n\t" OP " %" #B ", %" #B ", %" #C "\n\t" OP " %" #C ", %" #C ", %" #D "\n\t" OP " %" #D ", %" #D ", %" #A
#define G4(OP, A, B, C, D) OP " %" #A ", %" #A ", %" #B ", %" #C "\n\t" OP " %" #B ", %" #B ", %" #C ", %" #D "\n\t" OP " %" #C ", %" #C ", %" #D ", %" #A "\n\t" OP " %" #D ", %" #D ", %" #A ", %" #B
#define G2(OP, A, B, C, D) OP " %" #A ", %" #A "\n\t" OP " %" #B ", %" #B "\n\t" OP " %" #C ", %" #C "\n\t" OP " %" #D ", %" #D
#define GS(OP, A, B, C, D) OP " %" #A ", %8, %" #A "\n\t" OP " %" #B ", %8, %" #B "\n\t" OP " %" #C ", %8, %" #C "\n\t" OP " %" #D ", %8, %" #D
#define GDOT(A, B, C, D) "v_dot2_i32_i16 %" #A ", %" #A ", %" #A ", 0\n\tv_dot2_i32_i16 %" #B ", %" #B ", %" #B ", 0\n\tv_dot2_i32_i16 %" #C ", %" #C ", %" #C ", 0\n\tv_dot2_i32_i16 %" #D ", %" #D ", %" #D ", 0"
#define GDOTC(A, B, C, D) "v_dot2c_i32_i16 %" #A ", %" #B ", %" #B "\n\tv_dot2c_i32_i16 %" #B ", %" #C ", %" #C "\n\tv_dot2c_i32_i16 %" #C ", %" #D ", %" #D "\n\tv_dot2c_i32_i16 %" #D ", %" #A ", %" #A
#define GCMP(A, B, C, D) "v_cmp_gt_u32_e64 s[20:21], %" #A ", %" #B "\n\tv_subb_co_u32_e64 %" #A ", s[20:21], %" #A ", 0, s[20:21]\n\tv_cmp_gt_u32_e64 s[22:23], %" #C ", %" #D "\n\tv_subb_co_u32_e64 %" #C ", s[22:23], %" #C ", 0, s[22:23]"
#define GCMPV(A, B, C, D) "v_cmp_gt_u32 vcc, %" #A ", %" #B "\n\tv_subbrev_co_u32 %" #A ", vcc, 0, %" #A ", vcc\n\tv_cmp_gt_u32 vcc, %" #C ", %" #D "\n\tv_subbrev_co_u32 %" #C ", vcc, 0, %" #C ", vcc"
#define GSAR(A, B, C, D) "v_sub_u32 %" #A ", %" #A ", %" #B "\n\tv_ashrrev_i32 %" #A ", 31, %" #A "\n\tv_add3_u32 %" #C ", %" #C ", %" #A ", %8\n\tv_sub_u32 %" #D ", %" #D ", %" #B

// dependency distance 1 / 2: every instruction reads the result of the previous one / of the one before that
#define D1_3(OP, A) OP " %" #A ", %" #A ", %" #A "\n\t" OP " %" #A ", %" #A ", %" #A "\n\t" OP " %" #A ", %" #A ", %" #A "\n\t" OP " %" #A ", %" #A ", %" #A
#define D2_3(OP, A, B) OP " %" #A ", %" #A ", %" #A "\n\t" OP " %" #B ", %" #B ", %" #B "\n\t" OP " %" #A ", %" #A ", %" #A "\n\t" OP " %" #B ", %" #B ", %" #B
#define D1_4(OP, A) OP " %" #A ", %" #A ", %" #A ", %" #A "\n\t" OP " %" #A ", %" #A ", %" #A ", %" #A "\n\t" OP " %" #A ", %" #A ", %" #A ", %" #A "\n\t" OP " %" #A ", %" #A ", %" #A ", %" #A
#define D2_4(OP, A, B) OP " %" #A ", %" #A ", %" #A ", %" #A "\n\t" OP " %" #B ", %" #B ", %" #B ", %" #B "\n\t" OP " %" #A ", %" #A ", %" #A ", %" #A "\n\t" OP " %" #B ", %" #B ", %" #B ", %" #B
#define D1_2(OP, A) OP " %" #A ", %" #A "\n\t" OP " %" #A ", %" #A "\n\t" OP " %" #A ", %" #A "\n\t" OP " %" #A ", %" #A
#define D2_2(OP, A, B) OP " %" #A ", %" #A "\n\t" OP " %" #B ", %" #B "\n\t" OP " %" #A ", %" #A "\n\t" OP " %" #B ", %" #B

#define OPS(X)                                              \
    X(0, "(no VALU)", "s_nop 0", "s_nop 0")                               \
    X(1, "v_add_u32", G3("v_add_u32", 0, 1, 2, 3), G3("v_add_u32", 4, 5, 6, 7)) \
    X(2, "v_xor_b32", G3("v_xor_b32", 0, 1, 2, 3), G3("v_xor_b32", 4, 5, 6, 7)) \
    X(3, "v_mad_i32_i24", G4("v_mad_i32_i24", 0, 1, 2, 3), G4("v_mad_i32_i24", 4, 5, 6, 7)) \
    X(4, "v_mul_i32_i24", G3("v_mul_i32_i24", 0, 1, 2, 3), G3("v_mul_i32_i24", 4, 5, 6, 7)) \
    X(5, "v_mul_u32_u24", G3("v_mul_u32_u24", 0, 1, 2, 3), G3("v_mul_u32_u24", 4, 5, 6, 7)) \
    X(6, "v_ashrrev_i32 (sgpr)", GS("v_ashrrev_i32", 0, 1, 2, 3), GS("v_ashrrev_i32", 4, 5, 6, 7)) \
    X(7, "v_perm_b32", G4("v_perm_b32", 0, 1, 2, 3), G4("v_perm_b32", 4, 5, 6, 7)) \
    X(8, "v_pk_add_u16", G3("v_pk_add_u16", 0, 1, 2, 3), G3("v_pk_add_u16", 4, 5, 6, 7)) \
    X(9, "v_dot2_i32_i16 (vop3p)", GDOT(0, 1, 2, 3), GDOT(4, 5, 6, 7)) \
    X(10, "v_dot2c_i32_i16 (vop2)", GDOTC(0, 1, 2, 3), GDOTC(4, 5, 6, 7)) \
    X(11, "v_mad_i32_i16", G4("v_mad_i32_i16", 0, 1, 2, 3), G4("v_mad_i32_i16", 4, 5, 6, 7)) \
    X(12, "v_cvt_f32_u32", G2("v_cvt_f32_u32", 0, 1, 2, 3), G2("v_cvt_f32_u32", 4, 5, 6, 7)) \
    X(13, "v_sqrt_f32", G2("v_sqrt_f32", 0, 1, 2, 3), G2("v_sqrt_f32", 4, 5, 6, 7)) \
    X(14, "v_add_f32", G3("v_add_f32", 0, 1, 2, 3), G3("v_add_f32", 4, 5, 6, 7)) \
    X(15, "v_add3_u32", G4("v_add3_u32", 0, 1, 2, 3), G4("v_add3_u32", 4, 5, 6, 7)) \
    X(16, "v_cmp_e64 + v_subb_e64 (sgpr pair)", GCMP(0, 1, 2, 3), GCMP(4, 5, 6, 7)) \
    X(17, "v_cmp + v_subbrev (vcc)", GCMPV(0, 1, 2, 3), GCMPV(4, 5, 6, 7)) \
    X(18, "v_sub + v_ashr + v_add3 (+ v_sub)", GSAR(0, 1, 2, 3), GSAR(4, 5, 6, 7)) \
    X(19, "v_lshl_or_b32 (sgpr sh)", "v_lshl_or_b32 %0, %0, %8, %1\n\tv_lshl_or_b32 %1, %1, %8, %2\n\tv_lshl_or_b32 %2, %2, %8, %3\n\tv_lshl_or_b32 %3, %3, %8, %0", \
                                     "v_lshl_or_b32 %4, %4, %8, %5\n\tv_lshl_or_b32 %5, %5, %8, %6\n\tv_lshl_or_b32 %6, %6, %8, %7\n\tv_lshl_or_b32 %7, %7, %8, %4") \
    X(20, "v_mov_b32", G2("v_mov_b32", 0, 1, 2, 3), G2("v_mov_b32", 4, 5, 6, 7)) \
    X(22, "v_add_u32, distance 1", D1_3("v_add_u32", 0), D1_3("v_add_u32", 4)) \
    X(23, "v_add_u32, distance 2", D2_3("v_add_u32", 0, 1), D2_3("v_add_u32", 4, 5)) \
    X(24, "v_mad_i32_i24, distance 1", D1_4("v_mad_i32_i24", 0), D1_4("v_mad_i32_i24", 4)) \
    X(25, "v_mad_i32_i24, distance 2", D2_4("v_mad_i32_i24", 0, 1), D2_4("v_mad_i32_i24", 4, 5)) \
    X(26, "v_perm_b32, distance 1", D1_4("v_perm_b32", 0), D1_4("v_perm_b32", 4)) \
    X(27, "v_perm_b32, distance 2", D2_4("v_perm_b32", 0, 1), D2_4("v_perm_b32", 4, 5)) \
    X(28, "v_sqrt_f32, distance 1", D1_2("v_sqrt_f32", 0), D1_2("v_sqrt_f32", 4)) \
    X(29, "v_sqrt_f32, distance 2", D2_2("v_sqrt_f32", 0, 1), D2_2("v_sqrt_f32", 4, 5)) \
    X(30, "v_cvt_f32_u32, distance 1", D1_2("v_cvt_f32_u32", 0), D1_2("v_cvt_f32_u32", 4)) \
    X(31, "v_mul_u32_u24, distance 1", D1_3("v_mul_u32_u24", 0), D1_3("v_mul_u32_u24", 4)) \
    X(21, "v_alignbit_b32 (sgpr)", "v_alignbit_b32 %0, %0, %1, %8\n\tv_alignbit_b32 %1, %1, %2, %8\n\tv_alignbit_b32 %2, %2, %3, %8\n\tv_alignbit_b32 %3, %3, %0, %8", \
                                   "v_alignbit_b32 %4, %4, %5, %8\n\tv_alignbit_b32 %5, %5, %6, %8\n\tv_alignbit_b32 %6, %6, %7, %8\n\tv_alignbit_b32 %7, %7, %4, %8")

// one asm statement per MFMA slot (separate statements make hipcc put an s_nop between them)
#define SLOT(MNEM, ACC, TXT)                                                                                                     \
    asm volatile(MNEM " %9, %10, %11, %9\n\t" TXT                                                                               \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                                \
                 : "s"(sh), "v"(ACC), "v"(a), "v"(b)                                                                             \
                 : "vcc", "s20", "s21", "s22", "s23")

// SHAPE 0: 32x32x32, 1: 16x16x64.  NACC accumulators taking turns.  G = VALU instructions after each MFMA: 0, 4 or 8.
// (the accumulators are asm INPUTS that the MFMA overwrites in place: nothing reads them afterwards but the next MFMA)
template <int OP, int SHAPE, int NACC, int G>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out, unsigned seed, int sh) {
    unsigned x0 = threadIdx.x + seed, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3, x4 = x0 * 11 + 4, x5 = x0 * 13 + 5, x6 = x0 * 17 + 6, x7 = x0 * 19 + 7;
    v4i a = {(int)x0, (int)x1, (int)x2, (int)x3}, b = {(int)x4, (int)x5, (int)x6, (int)x7};
    v16i acc0 = {0}, acc1 = {0};
    v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(c0), "+v"(c1));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const bool second = NACC == 2 && (m & 1);
#define X(N, NAME, TA, TB)                                                                                       \
    if (OP == N) {                                                                                               \
        if (SHAPE == 0) {                                                                                        \
            if (G >= 8) { if (second) SLOT("v_mfma_i32_32x32x32_i8", acc1, TA "\n\t" TB); else SLOT("v_mfma_i32_32x32x32_i8", acc0, TA "\n\t" TB); } \
            else if (G >= 4) { if (second) SLOT("v_mfma_i32_32x32x32_i8", acc1, TA); else SLOT("v_mfma_i32_32x32x32_i8", acc0, TA); } \
            else { if (second) SLOT("v_mfma_i32_32x32x32_i8", acc1, ""); else SLOT("v_mfma_i32_32x32x32_i8", acc0, ""); } \
        } else {                                                                                                 \
            if (G >= 8) { if (second) SLOT("v_mfma_i32_16x16x64_i8", c1, TA "\n\t" TB); else SLOT("v_mfma_i32_16x16x64_i8", c0, TA "\n\t" TB); } \
            else if (G >= 4) { if (second) SLOT("v_mfma_i32_16x16x64_i8", c1, TA); else SLOT("v_mfma_i32_16x16x64_i8", c0, TA); } \
            else { if (second) SLOT("v_mfma_i32_16x16x64_i8", c1, ""); else SLOT("v_mfma_i32_16x16x64_i8", c0, ""); } \
        }                                                                                                        \
    }
            OPS(X)
#undef X
        }
    }
    asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(c0), "+v"(c1));
    unsigned r = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    for (int e = 0; e < 16; ++e) r ^= acc0[e] ^ acc1[e];
    for (int e = 0; e < 4; ++e) r ^= c0[e] ^ c1[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP, int SHAPE, int NACC, int G>
double run(int waves) {
    static unsigned *out = nullptr;
    if (!out) hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 1500;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<OP, SHAPE, NACC, G><<<256 * waves, 256>>>(10, out, 1, 8);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(s);
        k<OP, SHAPE, NACC, G><<<256 * waves, 256>>>(iters, out, rep, 8);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    return best * 1e6 / ((double)iters * 7 * waves);   // ns per MFMA slot per SIMD
}

template <int OP>
void row(const char *name) {
    const double d0 = run<0, 0, 1, 0>(2), i0 = run<0, 0, 2, 0>(2);
    const double d4 = run<OP, 0, 1, 4>(2), d8 = run<OP, 0, 1, 8>(2), i4 = run<OP, 0, 2, 4>(2), i8 = run<OP, 0, 2, 8>(2);
    printf("%-36s 1 chain/wave: %5.1f %5.1f %5.1f ns per MFMA slot (G = 0, 4, 8) -> %4.2f ns/VALU   |  2 chains/wave: %5.1f %5.1f %5.1f -> %4.2f ns/VALU\n",
           name, d0, d4, d8, (d8 - d4) / 4, i0, i4, i8, (i8 - i4) / 4);
}

int main() {
    printf("bare MFMA streams, ns per MFMA per SIMD:\n");
    printf("  32x32x32 i8   1 wave/SIMD: dependent chain %.2f, two chains %.2f   2 waves/SIMD: dependent %.2f, two chains %.2f\n",
           run<0, 0, 1, 0>(1), run<0, 0, 2, 0>(1), run<0, 0, 1, 0>(2), run<0, 0, 2, 0>(2));
    printf("  16x16x64 i8   1 wave/SIMD: dependent chain %.2f, two chains %.2f   2 waves/SIMD: dependent %.2f, two chains %.2f\n",
           run<0, 1, 1, 0>(1), run<0, 1, 2, 0>(1), run<0, 1, 1, 0>(2), run<0, 1, 2, 0>(2));
    printf("16x16x64 i8 + v_add_u32 x G (2 waves/SIMD, two chains): G=4 %.2f  G=8 %.2f ; + v_mad_i32_i24: G=4 %.2f  G=8 %.2f ns per MFMA slot\n",
           run<1, 1, 2, 4>(2), run<1, 1, 2, 8>(2), run<3, 1, 2, 4>(2), run<3, 1, 2, 8>(2));
#define X(N, NAME, TA, TB) if (N != 0) row<N>(NAME);
    OPS(X)
#undef X
    return 0;
}
