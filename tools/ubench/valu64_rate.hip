// Issue cost of the 64-bit vector opcodes the Lloyd pass's key arithmetic could use (round 6): v_mad_i64_i32, v_cmp_lt_i64 +
// 2 v_cndmask (an int64 minimum), v_lshl_add_u64, and their double-precision stand-ins v_cvt_f64_i32, v_fma_f64, v_min_f64
// (integers below 2^53 are exact in f64), beside v_add_u32 as the yardstick. Four independent chains, inline asm, 1 / 2 / 3 waves
// per SIMD on all CUs; prints ns per wave-instruction and SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef long long i64;
#define LOOP(TXT) for (int i = 0; i < iters; ++i) { _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile(TXT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1) : "s"(sc) : "vcc"); }
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, i64 *out, int seed, int sc) {
    i64 a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    int x0 = seed ^ threadIdx.x, x1 = x0 * 9 + 1;
    if (OP == 0) LOOP("v_add_u32 %4, %4, %5\n\tv_add_u32 %5, %5, %4\n\tv_add_u32 %4, %4, %5\n\tv_add_u32 %5, %5, %4")
    if (OP == 1) LOOP("v_mad_i64_i32 %0, vcc, %4, %6, %0\n\tv_mad_i64_i32 %1, vcc, %5, %6, %1\n\tv_mad_i64_i32 %2, vcc, %4, %6, %2\n\tv_mad_i64_i32 %3, vcc, %5, %6, %3")
    if (OP == 2) LOOP("v_cmp_lt_i64 vcc, %0, %1\n\tv_cmp_lt_i64 vcc, %1, %2\n\tv_cmp_lt_i64 vcc, %2, %3\n\tv_cmp_lt_i64 vcc, %3, %0")
    if (OP == 3) LOOP("v_lshl_add_u64 %0, %0, 1, %1\n\tv_lshl_add_u64 %1, %1, 1, %2\n\tv_lshl_add_u64 %2, %2, 1, %3\n\tv_lshl_add_u64 %3, %3, 1, %0")
    if (OP == 4) LOOP("v_cvt_f64_i32 %0, %4\n\tv_cvt_f64_i32 %1, %5\n\tv_cvt_f64_i32 %2, %4\n\tv_cvt_f64_i32 %3, %5")
    if (OP == 5) LOOP("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %2, %2, %3, %0\n\tv_fma_f64 %3, %3, %0, %1")
    if (OP == 6) LOOP("v_min_f64 %0, %0, %1\n\tv_min_f64 %1, %1, %2\n\tv_min_f64 %2, %2, %3\n\tv_min_f64 %3, %3, %0")
    if (OP == 7) LOOP("v_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc")
    if (OP == 8) LOOP("v_add_f64 %0, %0, %1\n\tv_add_f64 %1, %1, %2\n\tv_add_f64 %2, %2, %3\n\tv_add_f64 %3, %3, %0")
    if (OP == 9) LOOP("v_mul_lo_u32 %4, %4, %5\n\tv_mul_lo_u32 %5, %5, %4\n\tv_mul_lo_u32 %4, %4, %5\n\tv_mul_lo_u32 %5, %5, %4")
    if (a0 + a1 + a2 + a3 + x0 + x1 == 0x123456789LL) out[0] = 1;
}
template <int OP>
void run(const char *name) {
    i64 *out; hipMalloc(&out, 8);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 2000;
    printf("%-34s", name);
    for (int wps : {1, 2, 3}) {                      // waves per SIMD: 256 CUs x wps workgroups of 4 waves
        k<OP><<<256 * wps, 256>>>(10, out, 1, 3); hipDeviceSynchronize();
        hipEventRecord(s); k<OP><<<256 * wps, 256>>>(iters, out, 1, 3); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double n = (double)iters * 16 * 4 * wps;                      // instructions one SIMD issues
        printf("  %d w/SIMD %6.2f ns", wps, ms * 1e6 / n);
    }
    printf("\n");
}
int main() {
    run<0>("v_add_u32"); run<9>("v_mul_lo_u32"); run<7>("v_cndmask_b32"); run<1>("v_mad_i64_i32"); run<2>("v_cmp_lt_i64");
    run<3>("v_lshl_add_u64"); run<4>("v_cvt_f64_i32"); run<5>("v_fma_f64"); run<8>("v_add_f64"); run<6>("v_min_f64");
    return 0;
}
