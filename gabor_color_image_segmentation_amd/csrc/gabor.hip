// gabor.hip — SPEC.md §3 on gfx950: pyramid levels, reflect-padded planes, the filter bank as an im2col GEMM on
// v_mfma_i32_32x32x32_i8 with a fused magnitude epilogue, and the slab -> canonical tensor unpacker.
// The reference ships no code for this path; slot: /root/reference/BSD_metrics/script.py:30.
//
// Kernels
//   gabor_plane_kernel      level 0: interleaved RGB -> planar (pixel - 128) with the reflect border materialised.
//   gabor_down_kernel       level L >= 1: 2x2 block mean of level L-1 (round half up, edge replication), written as the
//                           padded plane of level L and, when a further level follows, as a compact image.
//   gabor_strip_kernel      the packed edge strips of the slab (csrc/common.h): right / bottom edges of one or two pixels of
//                           every level of a bank of at most two levels, outside the MFMA kernel's tile grid.
//   gabor_mfma_kernel       one pyramid level: A = packed 2-digit int8 taps of the level's filters (rows = filter x
//                           {re_lo,re_hi,im_lo,im_hi} x two pixel shifts, resident in registers), B = (pixel-128) windows
//                           read as aligned 16-byte pieces of an LDS tile kept twice (second copy two bytes to the right;
//                           LDS-DMA double buffer), exact int32 accumulate, fused epilogue (digit recombine, >>shift,
//                           |.|^2, exact isqrt) -> the level's part of the slab.
// Nothing here allocates or frees device memory or blocks the host; every entry point is ordered on the caller's stream
// (gcs_gabor_features forks level 1 of a large two-level batch onto a side stream and joins it back, see there).
#include "common.h"
#include <mutex>

constexpr int G_TW = 64;            // output tile width  (8 lanes-in-x * 8 shifts)
constexpr int G_TH = 32;            // output tile height (4 waves * 8 rows)
constexpr int G_HALO = 7;
constexpr int G_LROWS = G_TH + 15;  // 47 rows: halo 14 + the zero-tap row 15
constexpr int G_LPITCH = 96;        // bytes per LDS tile row (>= 64 + 16 + 12)

// scipy.ndimage mode='reflect' (d c b a | a b c d | d c b a) at any distance (SPEC.md §3)
__device__ __forceinline__ int reflect(int i, int n) {
    // the padded planes reach at most 7 samples before and 46 samples past a level (halo + tile padding): one fold is
    // enough when the level has at least 47 samples, and it costs three VALU instructions instead of an integer division
    // (the pre-pass kernels evaluate this several times per thread)
    if (n >= 47 && i >= -n && i < 2 * n) return i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// floor(sqrt(n)) for n < 2^31 (SPEC.md §3: n <= 2 * 32767^2, kept by gcs_bank_pack's bound on sum |tapq|), exact, in 7 VALU ops:
//   r    = v_sqrt_f32(float(n))        |r - s| <= 1.5e-7 * s <= 0.007 < 0.5   (s = true root)
//   bits = r + 2^23 (as uint)          the sum has ulp 1: bits = 0x4B000000 + RNE(r), RNE(r) in {floor(s), floor(s)+1}
//   qr^2 = v_mul_u32_u24(bits, bits)   the multiplier only sees the low 24 bits, i.e. qr = RNE(r) (< 2^16)
//   q    = qr - (qr^2 > n)             sign arithmetic, one v_add3: bits - 0x4B000000 + ((int)(n - qr^2) >> 31);
//                                      qr <= 46341 so qr^2 < 2^31 + 2^17 and the signed difference cannot overflow.
// Plain C on purpose (no inline asm): an asm statement's VGPR writes are invisible to hipcc's hazard recognizer, and once
// the epilogue is interleaved with a running MFMA chain an asm result can be allocated to a dead accumulator lane that
// the matrix pipe is still about to write (write-after-write: wrong pixels; found the hard way in round 2).
// Exhaustively checked over [0, 2^31) by tests/test_gpu_parity.py::test_isqrt31_exhaustive.
__device__ __forceinline__ unsigned isqrt31(unsigned n) {
    const unsigned bits = __float_as_uint(__builtin_amdgcn_sqrtf((float)n) + 8388608.0f);
    const int d = (int)(n - __umul24(bits, bits));
    return bits - 0x4B000000u + (unsigned)(d >> 31);
}

// The same root for the shift-8 epilogue of gabor_mfma_kernel, one instruction shorter: returns 0x4B000000 + floor(sqrt(n))
// (the caller only keeps the low 16 bits) with the "- (qr^2 > n)" done by v_cmp_gt_u32 into an SGPR pair of the register
// allocator's choice and v_subb_co_u32 (hipcc's own selection of this pattern goes through VCC, which chains the epilogues
// of a scheduling region one behind the other). The two asm statements read and write only ordinary values (never an
// accumulator register), and the kernel uses them only where every lane of every accumulator tuple is read, so the register
// allocator has no dead accumulator lane to hand to an asm result while the matrix pipe still owes it a write (the round-2
// bug described above). Checked over the whole domain by gcs_selftest_isqrt.
__device__ __forceinline__ unsigned isqrt31_biased(unsigned n) {
    const unsigned bits = __float_as_uint(__builtin_amdgcn_sqrtf((float)n) + 8388608.0f);
    const unsigned sq = __umul24(bits, bits);
    unsigned long long gt, carry_out;
    unsigned q;
    asm("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(gt) : "v"(sq), "v"(n));
    asm("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(q), "=s"(carry_out) : "v"(bits), "s"(gt));
    return q;
}
// a_re^2 + a_im^2 of the packed pair (a_re | a_im << 16), both int16, by two v_mad_i32_i16 (low halves, then high halves
// through op_sel): two ordinary-rate instructions where v_dot2_i32_i16 costs about two and a half beside a running MFMA
// chain (tools/ubench/mfma_beside; profiles/r3_notes.md)
__device__ __forceinline__ unsigned norm2_i16x2(unsigned p) {
    unsigned lo, n;
    asm("v_mad_i32_i16 %0, %1, %1, 0" : "=v"(lo) : "v"(p));
    asm("v_mad_i32_i16 %0, %1, %1, %2 op_sel:[1,1,0,0]" : "=v"(n) : "v"(p), "v"(lo));
    return n;
}

// out[i] = 1 if isqrt31 is wrong anywhere in [i * chunk, (i+1) * chunk) ∩ [0, n_max] (test hook; SPEC.md §3 domain)
__global__ void isqrt31_check_kernel(unsigned n_max, unsigned chunk, unsigned *__restrict__ bad) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long lo = (unsigned long long)i * chunk;
    unsigned wrong = 0;
    for (unsigned long long n = lo; n < lo + chunk && n <= n_max; ++n) {
        const unsigned long long q = isqrt31((unsigned)n);
        wrong |= (q * q > n) | ((q + 1) * (q + 1) <= n);
        wrong |= isqrt31_biased((unsigned)n) != (unsigned)q + 0x4B000000u;      // the epilogue's form of the same root
    }
    if (wrong) atomicAdd(bad, 1u);
}

extern "C" int gcs_selftest_isqrt(unsigned n_max, unsigned *bad_dev, gcs_stream_t stream) {
    if (!bad_dev) return gcs_fail(GCS_EINVAL, "gcs_selftest_isqrt: NULL pointer");
    const unsigned chunk = 4096;
    const unsigned threads = n_max / chunk + 1;
    hipError_t e = hipMemsetAsync(bad_dev, 0, sizeof(unsigned), stream);
    if (e != hipSuccess) return gcs_hip_fail(e, "gcs_selftest_isqrt(memset)");
    hipLaunchKernelGGL(isqrt31_check_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, n_max, chunk, bad_dev);
    GCS_CHECK_LAUNCH("gcs_selftest_isqrt");
    return GCS_OK;
}

// Level-0 pre-pass: plane[b][c][r][u] = img[b][refl(r-7)][refl(u-7)][c] - 128 (interleaved uint8 RGB -> planar int8 with
// the reflect border and the tile over-read materialised, so the MFMA kernel stages tiles as aligned 16-byte copies with no
// index arithmetic). A work item = 16 consecutive bytes of one plane row for the three channels. Away from the left / right
// borders the 16 pixels are 48 contiguous source bytes: three 16-byte loads at any byte alignment (gfx950 global memory
// takes them), de-interleaved with v_perm_b32 (9 per 4 pixels), three aligned 16-byte stores. Items that touch a border or
// the padding go pixel by pixel through reflect(); they are walked in a loop of their own, so that no wave mixes the two
// paths. Round 3's form (4 bytes x 4 rows per thread) took 36-45 us per 64 images in front of the level-0 MFMA launch -
// on the stage's critical path (profiles/r4_notes.md); two items per thread are in flight here.
typedef unsigned __attribute__((aligned(1))) unaligned_u32;
typedef int __attribute__((ext_vector_type(4), aligned(1))) v4i_a1;
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void plane_deinterleave4(unsigned d0, unsigned d1, unsigned d2, unsigned &r, unsigned &g, unsigned &b) {
    // bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 -> one dword per channel, then pixel - 128
    const unsigned r01 = __builtin_amdgcn_perm(d1, d0, 0x07060300u);   // R0 R1 . .   (bytes 0, 3 of d0; 6, 7 unused)
    const unsigned g01 = __builtin_amdgcn_perm(d1, d0, 0x07060401u);   // G0 G1 . .   (byte 1 of d0, byte 0 of d1)
    const unsigned b01 = __builtin_amdgcn_perm(d1, d0, 0x07060502u);   // B0 B1 . .   (byte 2 of d0, byte 1 of d1)
    const unsigned r23 = __builtin_amdgcn_perm(d2, d1, 0x05020000u);   // . . R2 R3   (byte 2 of d1, byte 1 of d2)
    const unsigned g23 = __builtin_amdgcn_perm(d2, d1, 0x06030000u);   // . . G2 G3   (byte 3 of d1, byte 2 of d2)
    const unsigned b23 = __builtin_amdgcn_perm(d2, d2, 0x03000000u);   // . . B2 B3   (bytes 0, 3 of d2)
    r = __builtin_amdgcn_perm(r23, r01, 0x07060100u) ^ 0x80808080u;
    g = __builtin_amdgcn_perm(g23, g01, 0x07060100u) ^ 0x80808080u;
    b = __builtin_amdgcn_perm(b23, b01, 0x07060100u) ^ 0x80808080u;
}

// Split slab (csrc/common.h): the pre-pass of level L clears that level's byte of every tile's flag word in front of the level's
// bank launches, which set it where a top nibble is non-zero (zmask: bit L = clear byte L; level 0 also clears the bytes of the
// levels the bank does not have). base == NULL: wide slab, nothing to do.
struct GaborFlagZero {
    unsigned char *base;      // the slab
    long long img_bytes;
    unsigned flag_off;
    int ntiles, zmask;
};
__device__ __forceinline__ void gabor_zero_flags(const GaborFlagZero &z, int b, int first, int stride) {
    if (!z.base) return;
    unsigned char *f = z.base + (size_t)b * z.img_bytes + z.flag_off;
    for (int t = first; t < z.ntiles; t += stride)
#pragma unroll
        for (int L = 0; L < GCS_LEVELS_MAX; ++L)
            if (z.zmask >> L & 1) f[4 * t + L] = 0;
}

// (the body: workgroup bx of gdx of image b, so that a small call can run it beside the level-1 pre-pass in ONE launch)
__device__ __forceinline__ void gabor_plane_body(int bx, int gdx, int b, const uint8_t *__restrict__ src, int Hs, int Ws, int HL,
                                                 int WL, int Hp, int Wp, int8_t *__restrict__ planes) {
    const int ng = Wp / 16;                                          // 16-byte groups per plane row
    // interior groups g: level columns 16 g - 7 .. 16 g + 8 all inside [0, WL)
    const int g_lo = 1, g_hi = WL >= 25 ? (WL - 9) / 16 : 0;         // interior: g_lo <= g <= g_hi (none for narrow levels)
    const int n_in = g_hi >= g_lo ? g_hi - g_lo + 1 : 0;
    const int n_bd = ng - n_in;
    const int tid = bx * 256 + threadIdx.x, nthr = gdx * 256;
    int8_t *pb = planes + (size_t)b * 3 * Hp * Wp;
    const size_t cstride = (size_t)Hp * Wp;
    // ---- interior items, two per thread in flight
    const int items_in = Hp * n_in;
    for (int it = tid; it < items_in; it += 2 * nthr) {
        const int it1 = it + nthr < items_in ? it + nthr : it;       // (the last round repeats an item: same bytes twice)
        v4i s[2][3];
        int rr[2], gg[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = j ? it1 : it;
            rr[j] = i / n_in;
            gg[j] = g_lo + (i - rr[j] * n_in);
            const int ly = reflect(rr[j] - G_HALO, HL);
            const uint8_t *p = src + (((size_t)b * Hs + ly) * Ws + (16 * gg[j] - G_HALO)) * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q) s[j][q] = *reinterpret_cast<const v4i_a1 *>(p + 16 * q);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned d[12] = {(unsigned)s[j][0][0], (unsigned)s[j][0][1], (unsigned)s[j][0][2], (unsigned)s[j][0][3],
                                    (unsigned)s[j][1][0], (unsigned)s[j][1][1], (unsigned)s[j][1][2], (unsigned)s[j][1][3],
                                    (unsigned)s[j][2][0], (unsigned)s[j][2][1], (unsigned)s[j][2][2], (unsigned)s[j][2][3]};
            unsigned o[3][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) plane_deinterleave4(d[3 * q], d[3 * q + 1], d[3 * q + 2], o[0][q], o[1][q], o[2][q]);
            int8_t *dst = pb + (size_t)rr[j] * Wp + 16 * gg[j];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                *reinterpret_cast<v4i *>(dst + c * cstride) = v4i{(int)o[c][0], (int)o[c][1], (int)o[c][2], (int)o[c][3]};
        }
    }
    // ---- border / padding items: pixel by pixel through reflect()
    const int items_bd = Hp * n_bd;
    for (int it = tid; it < items_bd; it += nthr) {
        const int r = it / n_bd, k = it - r * n_bd;
        const int g = k < g_lo ? k : k + n_in;                       // groups 0 .. g_lo-1, then g_hi+1 .. ng-1
        const int ly = reflect(r - G_HALO, HL);
        unsigned o[3][4] = {};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lx = reflect(16 * g + e - G_HALO, WL);
            const uint8_t *p = src + (((size_t)b * Hs + ly) * Ws + lx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c][e >> 2] |= ((unsigned)p[c] ^ 0x80u) << (8 * (e & 3));
        }
        int8_t *dst = pb + (size_t)r * Wp + 16 * g;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<v4i *>(dst + c * cstride) = v4i{(int)o[c][0], (int)o[c][1], (int)o[c][2], (int)o[c][3]};
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void gabor_plane_kernel(const uint8_t *__restrict__ src, int Hs, int Ws, int HL, int WL,
                                                          int Hp, int Wp, int8_t *__restrict__ planes,
                                                          uint8_t *__restrict__ img_out, GaborFlagZero fz) {
    static_assert(MODE == 0, "levels >= 1 use gabor_down_kernel");
    gabor_zero_flags(fz, (int)blockIdx.y, (int)blockIdx.x * 256 + (int)threadIdx.x, (int)gridDim.x * 256);
    gabor_plane_body((int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, src, Hs, Ws, HL, WL, Hp, Wp, planes);
}

// Levels >= 1, staged through LDS (same-run rocprofv3: 33 us per 64 images against 47 us for gabor_plane_kernel<1>'s
// unaligned-dword form and 41 us for one byte load per sample): one workgroup per (padded plane row r, image b); the two
// source rows are copied to LDS with coalesced dword loads (from the enclosing aligned dwords: a row starts at an
// arbitrary byte), then every thread assembles output dwords from LDS bytes. SRC_RGB: level 1 from the interleaved
// input; otherwise level L > 1 from the compact planar level L-1 image [B][3][Hs][Ws].
template <bool SRC_RGB>
__device__ __forceinline__ void gabor_down_body(int r, int b, unsigned char *s_rows, const uint8_t *__restrict__ src,
                                                size_t src_bytes, int Hs, int Ws, int HL, int WL, int Hp, int Wp,
                                                int8_t *__restrict__ planes, uint8_t *__restrict__ img_out) {
    const int ly = reflect(r - G_HALO, HL);
    constexpr int NSEG = SRC_RGB ? 1 : 3;                            // planar source: one segment per channel
    const int seg_bytes = SRC_RGB ? Ws * 3 : Ws;
    const int seg_pitch = (seg_bytes + 3 + 3) & ~3;                  // LDS bytes per staged segment (room for the misalignment)
    const int ys[2] = {2 * ly, min(2 * ly + 1, Hs - 1)};
    int lead[2 * NSEG];
#pragma unroll
    for (int i = 0; i < 2 * NSEG; ++i) {
        const int row = i / NSEG, c = i % NSEG;
        const size_t off = SRC_RGB ? ((size_t)b * Hs + ys[row]) * Ws * 3 : (((size_t)b * 3 + c) * Hs + ys[row]) * Ws;
        const size_t a0 = off & ~(size_t)3;
        lead[i] = (int)(off - a0);
        const int ndw = (lead[i] + seg_bytes + 3) >> 2;
        unsigned *dst = reinterpret_cast<unsigned *>(s_rows + (size_t)i * seg_pitch);
        for (int d = threadIdx.x; d < ndw; d += 256) {
            const size_t g = a0 + 4 * (size_t)d;
            unsigned v;
            if (g + 4 <= src_bytes) {
                v = *reinterpret_cast<const unsigned *>(src + g);
            } else {                                                 // the last dword of the buffer: bytewise
                v = 0;
                for (int e = 0; e < 4; ++e)
                    if (g + e < src_bytes) v |= (unsigned)src[g + e] << (8 * e);
            }
            dst[d] = v;
        }
    }
    __syncthreads();
    auto px = [&](int row, int c, int x) -> unsigned {               // source pixel (ys[row], x), channel c
        const int i = SRC_RGB ? row : row * 3 + c;
        return s_rows[(size_t)i * seg_pitch + lead[i] + (SRC_RGB ? 3 * x + c : x)];
    };
    const bool row_in = r >= G_HALO && r - G_HALO < HL;
    for (int u4 = threadIdx.x; u4 < Wp / 4; u4 += 256) {
        unsigned o[3] = {0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 4 * u4 + e;
            const int lx = reflect(u - G_HALO, WL);
            const int x0 = 2 * lx, x1 = min(2 * lx + 1, Ws - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned m = (px(0, c, x0) + px(0, c, x1) + px(1, c, x0) + px(1, c, x1) + 2u) >> 2;
                if (img_out && row_in && u >= G_HALO && u - G_HALO < WL)
                    img_out[(((size_t)b * 3 + c) * HL + (r - G_HALO)) * WL + (u - G_HALO)] = (uint8_t)m;
                o[c] |= (m ^ 0x80u) << (8 * e);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<unsigned *>(planes + (((size_t)b * 3 + c) * Hp + r) * Wp + 4 * u4) = o[c];
    }
}
template <bool SRC_RGB>
__global__ __launch_bounds__(256) void gabor_down_kernel(const uint8_t *__restrict__ src, size_t src_bytes, int Hs, int Ws,
                                                         int HL, int WL, int Hp, int Wp, int8_t *__restrict__ planes,
                                                         uint8_t *__restrict__ img_out, GaborFlagZero fz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_rows[];
    gabor_zero_flags(fz, (int)blockIdx.y, (int)blockIdx.x * 256 + (int)threadIdx.x, (int)gridDim.x * 256);
    gabor_down_body<SRC_RGB>((int)blockIdx.x, (int)blockIdx.y, s_rows, src, src_bytes, Hs, Ws, HL, WL, Hp, Wp, planes, img_out);
}
// The level-0 and level-1 pre-passes of a two-level bank in ONE launch (small calls: each is a 5 - 7 us launch of its own in
// front of the bank, script.py:22-30 calls the slot once per image): workgroups [0, n0) run the level-0 body, the rest one
// padded level-1 row each.
__global__ __launch_bounds__(256) void gabor_pre01_kernel(const uint8_t *__restrict__ src, size_t src_bytes, int H, int W, int n0,
                                                          int H0, int W0, int Hp0, int Wp0, int8_t *__restrict__ planes0,
                                                          int H1, int W1, int Hp1, int Wp1, int8_t *__restrict__ planes1,
                                                          GaborFlagZero fz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_rows[];
    gabor_zero_flags(fz, (int)blockIdx.y, (int)blockIdx.x * 256 + (int)threadIdx.x, (int)gridDim.x * 256);
    if ((int)blockIdx.x < n0) gabor_plane_body((int)blockIdx.x, n0, (int)blockIdx.y, src, H, W, H0, W0, Hp0, Wp0, planes0);
    else gabor_down_body<true>((int)blockIdx.x - n0, (int)blockIdx.y, s_rows, src, src_bytes, H, W, H1, W1, Hp1, Wp1, planes1, nullptr);
}

#ifndef GCS_GABOR_WAVES
#define GCS_GABOR_WAVES 2
#endif
#ifndef GCS_GABOR_MTMAX_
#define GCS_GABOR_MTMAX_ 3
#endif
constexpr int GCS_GABOR_MTMAX = GCS_GABOR_MTMAX_;   // row tiles (of four filters) per launch; two for the 15-row frame

// One pyramid level of a launch. Consecutive levels with the same filter count share ONE launch (tiles of level L, then
// L+1, ... in one list): a level boundary then costs one reload of the A operand per workgroup instead of a kernel
// boundary (drain + ramp of 512 persistent workgroups, and a launch for a one-image call).
struct GaborLevel {
    const int8_t *planes;    // padded planes [B][3][Hp][Wp]
    const int8_t *apack;     // packed taps of this launch's row tiles
    const int32_t *bias;
    // The tile list covers the level's MAIN region, rows [0, HLm) x columns [0, pitchLm): the whole level, or - when the
    // slab packs edge strips (csrc/common.h) - the level without them (gabor_strip_kernel computes those). The region is cut
    // into HALF tiles of 32 x 32 pixels, htx per row, hcount per image; a workgroup's 64 x 32 tile is any two consecutive
    // half tiles of one image (tiles_per_image = ceil(hcount / 2)), so a region 15 half tiles wide (480 pixels) costs 7.5
    // tiles per tile row, not 8.
    int HLm, Hp, Wp, pitchLm, htx, hcount, tiles_per_image;
    int tile_end;            // end of this level's tiles in the launch's tile list
    int L, offL;
};
struct GaborLevels {
    GaborLevel lv[GCS_LEVELS_MAX];
};
// Where the MFMA kernel's features go: the slab's main blocks (csrc/common.h).
struct GaborSlab {
    int bx_n;                // main blocks per block row
    int ntiles, tile_bytes;
    long long img_bytes;     // bytes of one image's slab
    // split slab (csrc/common.h; GaborLevel::offL is then the level's first SLOT of a tile)
    int S;
    unsigned mid_off, top_off, flag_off;
};

// The bank of one level as an im2col GEMM on v_mfma_i32_32x32x32_i8 (round 3 layout).
//   A rows  one 32-row tile = FOUR filters x {re_lo, re_hi, im_lo, im_hi} x TWO pixel shifts: row 8i + 4h + j is digit j of
//           filter 2(i >> 1) + h of the tile, evaluated for the pixel (i & 1) to the right of the B column's pixel: the taps
//           of that row sit one K-slot further right in the 16-slot frame row (gcs_bank_pack). A level's 12 filters are
//           exactly MT = 3 tiles (round 2 padded them to 16), and the one-pixel shifts cost no instruction.
//   K       K-step kk, half h, slot j = tap (dy = 2kk + h, dx = j) of the 16 x 16 frame, as before; KS = 7 K-steps when
//           ksize <= 13 (frame rows 1..13), 8 for the 15-row frame.
//   B cols  32 pixels (8 x 4): column (li, lyy) is the pixel PAIR x0 + 8li + 2pp (+0, +1), row row0 + lyy. Its fragment
//           for K-step kk is the 16 window bytes starting at byte 8li + 2pp of tile row trow + 2kk + h. LDS holds the tile
//           TWICE, the second copy shifted by two bytes (both written by LDS-DMA, the second from source address + 2), so
//           every fragment is an aligned read of copy (pp & 1) at byte 8li + 4(pp >> 1): no v_alignbit / v_perm at all
//           (round 2: 21 per pixel-shift step = a fifth of the kernel's VALU instructions).
//   D       lane (n, h) holds rows 8i + 4h + j: both pixels of the pair for the two filters 2fp + h of the tile, i.e. one
//           packed dword of the slab per (filter, pair) as before.
// MT row tiles per launch (A operand = MT x KS lane-linear 16-byte fragments in registers), GQ = filter pairs that exist in
// the last tile (the epilogue of absent filters is not compiled). LVL = the pyramid level of a single-level launch (0 or 1:
// the store path of that level alone is compiled, every level field is a loop-invariant kernel argument), or -1 = FUSED: the
// tile list may hold several levels and the level is a run-time value. FAST = the 11-instruction epilogue for shift == 8
// (every Q15 bank) when all four accumulator quads of every tile are in use (GQ == 2); see `epilogue` below.
// Per 4-row block a wave runs 4 MT chains of KS MFMAs (pair-major, tile-minor); chain t and the epilogue of chain t-1
// share one scheduling region over two accumulator tuples, and the next pair's fragments are read from LDS while the
// current pair's chains run.
constexpr int G_COPY = 3 * G_LROWS * G_LPITCH;      // bytes of one copy of a tile (three channels)


// SPLIT = the split slab (csrc/common.h: low bytes, MID nibbles and TOP nibbles in three planar arrays + a flag per tile and level);
// banks of at most two levels: store paths for levels 0 and 1 only, whatever LVL.
template <int MT, int GQ, int KS, int LVL, bool FAST, bool SPLIT = false>
__global__ __launch_bounds__(256, GCS_GABOR_WAVES) void gabor_mfma_kernel(
    GaborLevels G, int FLv, int fbase0, int shift, unsigned char *__restrict__ feats, int total_tiles, GaborSlab S) {
    // blockIdx.y = row-tile GROUP of the launch: group q works on filters fbase0 + 4 MT q .. of every tile of the list (a small call -
    // one to four BSD images - has fewer tiles than the chip has slots, so its three row tiles run side by side as three groups
    // of MT = 1 workgroups instead of one after the other in every workgroup; every other launch has one group)
    const int grp = blockIdx.y, fbase = fbase0 + 4 * MT * grp;
    // Persistent workgroups: the A operand and the biases are loaded ONCE, then the workgroup walks
    // tiles blockIdx.x, +gridDim.x, ... ; the next tile streams into the other LDS buffer by LDS-DMA
    // (global_load_lds: no VGPRs, lands while this tile computes). The tile image is a flat run of
    // 2 x 846 16-byte chunks, i.e. exactly the lane-linear destination LDS-DMA wants.
    constexpr bool FUSED = LVL < 0;
    __shared__ __attribute__((aligned(16))) int8_t s_tile[2][2][3][G_LROWS][G_LPITCH];
    constexpr int NCHUNK1 = G_COPY / 16, NCHUNK = 2 * NCHUNK1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and the compiler should know it (SGPR address math)

    // level of a tile of the list (uniform; a workgroup's tiles only move up the levels). Selected field by field: a
    // dynamically indexed by-value kernel argument would be copied to scratch.
    auto level_of = [&](int tile, int from) {
        if constexpr (!FUSED) return 0;
        if constexpr (LVL == -2) return tile >= G.lv[0].tile_end ? 1 : 0;
        int l = from;
        while (l + 1 < GCS_LEVELS_MAX && tile >= (l == 0 ? G.lv[0].tile_end : l == 1 ? G.lv[1].tile_end : G.lv[2].tile_end)) ++l;
        return l;
    };
    auto pick = [&](int l) -> GaborLevel {
        if constexpr (LVL == -2) return l == 0 ? G.lv[0] : G.lv[1];
        return l == 0 ? G.lv[0] : l == 1 ? G.lv[1] : l == 2 ? G.lv[2] : G.lv[3];
    };
    static_assert(GCS_LEVELS_MAX == 4, "level_of / pick enumerate four levels");

    // the two half tiles (rows y, columns x of their first pixel) of tile `rem` of an image; an absent second half (odd
    // half-tile count) is placed below the region: it is staged from the first half's window and stores nothing
    auto halves_of = [&](int rem, int htx, int hcount, int &ya, int &xa, int &yb, int &xb) {
        const int q0 = 2 * rem, hy = q0 / htx, hx = q0 - hy * htx;
        ya = hy * G_TH;
        xa = hx * (G_TW / 2);
        const bool wrap = hx + 1 == htx;
        yb = q0 + 1 < hcount ? (wrap ? ya + G_TH : ya) : (1 << 28);
        xb = wrap ? 0 : xa + G_TW / 2;
    };
    auto stage_tile = [&](int tile, int lvl, int buf) {
        const GaborLevel v = pick(lvl);
        const int t_ = tile - (lvl ? pick(lvl - 1).tile_end : 0);
        const int b_ = t_ / v.tiles_per_image, rem = t_ % v.tiles_per_image;
        int ya, xa, yb, xb;
        halves_of(rem, v.htx, v.hcount, ya, xa, yb, xb);
        if (yb >= (1 << 28)) { yb = ya; xb = xa; }
        const int8_t *srcA = v.planes + ((size_t)b_ * 3 * v.Hp + ya) * v.Wp + xa;
        const int8_t *srcB = v.planes + ((size_t)b_ * 3 * v.Hp + yb) * v.Wp + xb;
#pragma unroll
        for (int k = 0; k < (NCHUNK + 255) / 256; ++k) {
            const int i = tid + 256 * k;
            if (i < NCHUNK) {
                const int copy = i >= NCHUNK1, i1 = i - copy * NCHUNK1;
                const int ch16 = i1 % (G_LPITCH / 16), rc = i1 / (G_LPITCH / 16);
                const int row = rc % G_LROWS, c = rc / G_LROWS;
                // an LDS row = the 48-byte windows of the two half tiles side by side (32 pixels + 14 of halo + the second
                // copy's 2); the second copy is the same window read two bytes further right (LDS-DMA takes any source alignment)
                const bool right = ch16 >= G_LPITCH / 32;
                const int8_t *g = (right ? srcB : srcA) + ((size_t)c * v.Hp + row) * v.Wp + 16 * (ch16 - (right ? G_LPITCH / 32 : 0)) + 2 * copy;
                // LDS destination: wave-uniform base (this wave's first chunk) in M0 + lane * 16. Issued as asm: hipcc models
                // the builtin as a store to "some LDS" and then drains vmcnt - the DMA AND every feature store still in
                // flight - in front of the next LDS read of the loop, i.e. at the top of every 4-row block (round 2 and the
                // first round-3 build: waves parked in s_waitcnt a quarter of their cycles). The DMA only ever targets the
                // buffer nobody reads before the barrier at the end of the tile, where vmcnt is drained explicitly.
                const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) int8_t *)(&s_tile[buf][0][0][0][0]) +
                                     16u * (unsigned)(256 * k + 64 * wave);
                unsigned m0_saved;             // M0 is reserved by hipcc: hand it back as it was
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_saved) : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(g));
            }
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int li = r & 7, lyy = r >> 3;
    const int lcol = 8 * li + (li >= 4 ? G_LPITCH / 2 - G_TW / 2 : 0);   // first window byte of the lane's pixels in its half's 48-byte LDS window

    int k256 = 256, k65536 = 65536;
    asm volatile("" : "+s"(k256), "+s"(k65536));      // opaque multipliers: keep v_mad_i32_i24 / v_mad_u32_u24, not shifts

    // ---- per level: the whole A operand lives in registers (MT x KS lane-linear 16-byte fragments) with the biases
    v4i afr[MT][KS];
    int bias_v[MT][2];
    int HL = 0, pitchL = 0, htx = 1, hcount = 1, tiles_per_image = 1, tile0 = 0, L = 0, offL = 0;   // (HL, pitchL: the main region's)
    int side_sh = 3, npl = KP_TP;          // slab geometry of the level (csrc/common.h): sub-block side 8 >> L, pixels per plane of a tile
    auto enter_level = [&](int lvl) {
        const GaborLevel v = pick(lvl);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                afr[mt][kk] = reinterpret_cast<const v4i *>(v.apack)[((size_t)(MT * grp + mt) * 8 + kk) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int fp = 0; fp < 2; ++fp) bias_v[mt][fp] = v.bias[4 * (MT * grp + mt) + 2 * fp + h];
        HL = v.HLm; pitchL = v.pitchLm; htx = v.htx; hcount = v.hcount; tiles_per_image = v.tiles_per_image;
        tile0 = lvl ? pick(lvl - 1).tile_end : 0;
        L = LVL >= 0 ? LVL : v.L;
        offL = v.offL;
        side_sh = 3 - L;
        npl = KP_TP >> (2 * L);
    };

    int tile = blockIdx.x;
    int lvl = FUSED ? -1 : 0, lvl_next = tile < total_tiles ? level_of(tile, 0) : 0;
    if constexpr (!FUSED) enter_level(0);
    if (tile < total_tiles) stage_tile(tile, lvl_next, 0);
    // hipcc does not know about the DMA (asm): drain vmcnt by hand before every barrier that publishes a tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; tile < total_tiles; tile += gridDim.x, ++it) {
      const int buf = it & 1;
      if constexpr (FUSED) {
          if (lvl_next != lvl) {
              lvl = lvl_next;
              enter_level(lvl);
          }
      }
      if (tile + (int)gridDim.x < total_tiles) {
          lvl_next = level_of(tile + gridDim.x, lvl);
          stage_tile(tile + gridDim.x, lvl_next, buf ^ 1);
      }
      const int tl = tile - tile0;
      const int b = tl / tiles_per_image, trem = tl % tiles_per_image;
      int y0a, x0a, y0b, x0b;                                       // the tile's two half tiles: lanes li < 4 / li >= 4
      halves_of(trem, htx, hcount, y0a, x0a, y0b, x0b);
      const int y0 = li < 4 ? y0a : y0b, x0 = (li < 4 ? x0a : x0b) - (li < 4 ? 0 : G_TW / 2);   // per lane; x0 + 8 li = the lane's first pixel
      // The wave's work in this tile: 4-row blocks (channel c, row block rb), a software pipeline over ALL of them: every
      // block is 4 MT chains of KS MFMAs (pixel pair pp, tile mt); chain t shares its scheduling region with the epilogue of
      // chain t-1 - for the first chain of a block that is the LAST chain of the previous block, whose stores follow it - and
      // the fragments of the next pixel pair (for the last pair: of the next block) are read from LDS one pair ahead. Round 2
      // and the first round-3 build drained the pipeline at every block: a region of MFMAs alone, a region of VALU alone and
      // an exposed LDS read per block.
      const int ytop = (y0a < y0b ? y0a : y0b) + wave * 8;          // this wave's first row in the upper of the two half tiles
      const bool rows1 = ytop + 4 < HL;                             // both row blocks of this wave hold region rows (in some half)
      const int nblk = ytop < HL ? (rows1 ? 6 : 3) : 0;             // waves wholly below the region skip the work, not the barrier
      if (nblk) {
        unsigned outp[MT][2][4];           // [tile][filter pair][pixel pair]: two uint16 magnitudes each
        v16i acc[2];
        v4i win[KS];                       // the fragments of ONE pixel pair, refilled behind the last chain that reads them
        int c = 0, trow = wave * 8 + lyy;                            // current block
        int st_c = 0, st_trow = 0;                                   // the block whose results are in outp
        // fragment kk of pixel pair pp = 2 qq + copy of block (wc, wtrow): 16 bytes of copy `copy` at byte 8 li + 4 qq of
        // tap row 2 kk + h (this half-wave's parity)
        auto load_fragment = [&](int kk, int wc, int wtrow, int copy, int qq) -> v4i {
            const int8_t *rp = &s_tile[buf][copy][wc][wtrow + 2 * kk + h][lcol];
            if (qq == 0) {
                const v2i lo = *reinterpret_cast<const v2i *>(rp), hi = *reinterpret_cast<const v2i *>(rp + 8);
                return v4i{lo[0], lo[1], hi[0], hi[1]};
            }
            const int *dp = reinterpret_cast<const int *>(rp + 4);    // 4-byte aligned: dword reads straight into the registers
            return v4i{dp[0], dp[1], dp[2], dp[3]};
        };
        // One output of the epilogue: accumulator quad i = 2 fp + sx of a finished chain = {re_lo, re_hi, im_lo, im_hi} of filter
        // 2 fp + h of the tile at pixel 2 pp + sx -> its magnitude (FAST: plus 0x4B000000, dropped when the pair is packed).
        auto epi_out = [&](const v16i &ac, int mt, int fp, int sx) -> unsigned {
            const int i = 2 * fp + sx;
            if constexpr (FAST) {
                // a = (256 H + L + bias) >> 8 = H + ((L + bias) >> 8), |a| < 2^15 (gcs_bank_pack bounds sum |tapq|), so
                // the low 16 bits of H and bytes 1-2 of L + bias add up to a in 16-bit arithmetic: one v_perm packs
                // (H_re, H_im), one packs and shifts (L_re + bias, L_im), v_pk_add_u16 gives (a_re | a_im << 16), two
                // v_mad_i32_i16 its squared norm, then the root: 13 VALU instructions per output with packing (round 2: 15
                // plus 3.5 for the window). Variants measured in profiles/r3_notes.md (v_dot2 for the norm, the general front
                // end with only the root tail changed).
                const unsigned ph = __builtin_amdgcn_perm((unsigned)ac[4 * i + 3], (unsigned)ac[4 * i + 1], 0x05040100u);
                const unsigned pl = __builtin_amdgcn_perm((unsigned)ac[4 * i + 2], (unsigned)(ac[4 * i + 0] + bias_v[mt][fp]), 0x06050201u);
                const v2s pa = __builtin_bit_cast(v2s, ph) + __builtin_bit_cast(v2s, pl);
                return isqrt31_biased(norm2_i16x2(__builtin_bit_cast(unsigned, pa)));
            } else {
                // v = 256*hi + lo (+ bias): v_mad_i32_i24 (|hi| < 2^22); the factor sits in an SGPR the
                // compiler cannot see through, or it turns the multiply into a left shift (2.3x slower here)
                const int v_re = __mul24(ac[4 * i + 1], k256) + ac[4 * i + 0] + bias_v[mt][fp];
                const int v_im = __mul24(ac[4 * i + 3], k256) + ac[4 * i + 2];
                const int a_re = v_re >> shift, a_im = v_im >> shift;
                return isqrt31((unsigned)__mul24(a_re, a_re) + (unsigned)__mul24(a_im, a_im));
            }
        };
        // the two pixels of a pair -> one dword of the slab: offset-binary (x ^ 0x8080: both bytes are then signed MFMA digits
        // for the k-means pass, which stages them untouched), the odd pixel in the high half
        // (split slab: the pair as (lo_a, lo_b, hi_a, hi_b), no offset yet - store_block separates the bytes of four pairs)
        auto epi_pack = [&](unsigned q0, unsigned q1, int mt, int fp, int pp) {
            unsigned &o = outp[mt][fp][pp];
            if constexpr (SPLIT) {
                o = __builtin_amdgcn_perm(q1, q0, 0x05010400u);
            } else {
                if constexpr (FAST) o = __builtin_amdgcn_perm(q1, q0, 0x05040100u);   // low halves: the 0x4B000000 bias drops out
                else o = __umul24(q1, k65536) + q0;
                o ^= 0x80808080u;
            }
            asm volatile("" : "+v"(o));     // materialise now: hipcc otherwise sinks the packing into the store branches
        };
        // Slice `slot` (= the K-step the chain in flight is at) of the epilogue of finished chain (mt, pp): slots 1..4 one
        // output each, slot 5 the packing; slot 0 leaves the last MFMA of the finished chain time to land.
        unsigned qv[4];
        auto epi_slice = [&](const v16i &ac, int mt, int pp, int slot) {
            const bool two = !(mt == MT - 1 && GQ == 1);          // both filter pairs of this tile exist
            if (slot >= 1 && slot <= 4) {
                const int fp = (slot - 1) >> 1, sx = (slot - 1) & 1;
                if (fp == 0 || two) qv[slot - 1] = epi_out(ac, mt, fp, sx);
            } else if (slot == 5) {
                epi_pack(qv[0], qv[1], mt, 0, pp);
                if (two) epi_pack(qv[2], qv[3], mt, 1, pp);
            }
        };
        auto store_block = [&]() {
            // 8 consecutive level pixels x = x0 + 8*li .. +7 of row oy of the MAIN region (packed edge strips are not in the
            // tile list: gabor_strip_kernel). Level 0: one row of one 8x8 block = one 16-byte store; level L: 2^L pieces of
            // 8 >> L pixels, one per block (a block holds (8 >> L)^2 level-L pixels).
            // compile-time geometry for single-level launches
            const int Lc = LVL >= 0 ? LVL : L, ssh = LVL >= 0 ? 3 - LVL : side_sh, nplc = LVL >= 0 ? (KP_TP >> (2 * (LVL >= 0 ? LVL : 0))) : npl;
            const int oy = y0 + st_trow, ox = x0 + 8 * li;
            if constexpr (SPLIT) {
              if (oy < HL && ox < pitchL) {
                // Split slab: the lane's 8 pixels are 8 consecutive SLOTS of one block row (level 0) or 4 + 4 slots of one parent
                // row of two neighbouring blocks (level 1: consecutive too when both blocks lie in one tile). Per filter: the low bytes (XOR 0x80) go to LO at the slot index, the
                // nibbles 8..11 / 12..15 of slots i and i + g / 2 of the group share byte i of MID / TOP at half the slot index.
                const int by = oy >> ssh, iy = oy & ((1 << ssh) - 1);
                const int bx0 = ox >> ssh;
                unsigned char *img = feats + (size_t)b * S.img_bytes;
                const unsigned pl0 = (unsigned)(offL + (st_c * FLv + fbase) * nplc);          // first slot of the launch's first plane of channel st_c
                unsigned char *lo_base = img + pl0, *mid_base = img + S.mid_off + (pl0 >> 1), *top_base = img + S.top_off + (pl0 >> 1);
                // slot order inside a plane: (slot row, block in tile, slot column): the wave's 4 rows x 4 blocks are one run
                const unsigned row_slot = (unsigned)((iy << (ssh + 2)) + h * nplc);
                auto lane_slot = [&](int p) -> unsigned {
                    const int blk = by * S.bx_n + bx0 + p;
                    return (unsigned)(blk >> 2) * (unsigned)S.S + (unsigned)((blk & 3) << ssh) + row_slot;
                };
                auto planes = [&](auto &&put) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int fp = 0; fp < 2; ++fp) {
                            if (mt == MT - 1 && fp >= GQ) continue;
                            const bool last = mt == MT - 1 && fp == GQ - 1;
                            if (last && fbase + 4 * mt + 2 * fp + h >= FLv) continue;
                            put((unsigned)((4 * mt + 2 * fp) * nplc), outp[mt][fp][0], outp[mt][fp][1], outp[mt][fp][2], outp[mt][fp][3]);
                        }
                };
                if (Lc == 0) {
                    const unsigned s0 = lane_slot(0), n0 = s0 >> 1;
                    unsigned any = 0;
                    planes([&](unsigned po, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
                        const unsigned l0 = __builtin_amdgcn_perm(w1, w0, 0x05040100u) ^ 0x80808080u;
                        const unsigned l1 = __builtin_amdgcn_perm(w3, w2, 0x05040100u) ^ 0x80808080u;
                        const unsigned h0 = __builtin_amdgcn_perm(w1, w0, 0x07060302u);          // high bytes of pixels 0..3
                        const unsigned h1 = __builtin_amdgcn_perm(w3, w2, 0x07060302u);          // ... of pixels 4..7
                        const unsigned mid = (h0 & 0x0f0f0f0fu) | ((h1 << 4) & 0xf0f0f0f0u);
                        const unsigned top = ((h0 >> 4) & 0x0f0f0f0fu) | (h1 & 0xf0f0f0f0u);
                        __builtin_nontemporal_store(v2i{(int)l0, (int)l1}, reinterpret_cast<v2i *>(lo_base + po + s0));
                        __builtin_nontemporal_store(mid, reinterpret_cast<unsigned *>(mid_base + (po >> 1) + n0));
                        __builtin_nontemporal_store(top, reinterpret_cast<unsigned *>(top_base + (po >> 1) + n0));
                        any |= top;
                    });
                    if (any) img[S.flag_off + 4u * (unsigned)((by * S.bx_n + bx0) >> 2)] = 1;
                } else {
                    const unsigned s0 = lane_slot(0), s1 = lane_slot(1);
                    const bool has1 = bx0 + 1 < S.bx_n;
                    unsigned any0 = 0, any1 = 0;
                    auto half = [&](unsigned po, unsigned wa, unsigned wb, unsigned sl, unsigned &any) {
                        const unsigned l = __builtin_amdgcn_perm(wb, wa, 0x05040100u) ^ 0x80808080u;
                        const unsigned hh = __builtin_amdgcn_perm(wb, wa, 0x07060302u);          // high bytes of the four pixels
                        const unsigned mid = (hh & 0x0f0fu) | ((hh >> 12) & 0xf0f0u);
                        const unsigned top = ((hh >> 4) & 0x0f0fu) | ((hh >> 16) & 0xf0f0u);
                        __builtin_nontemporal_store(l, reinterpret_cast<unsigned *>(lo_base + po + sl));
                        *reinterpret_cast<uint16_t *>(mid_base + (po >> 1) + (sl >> 1)) = (uint16_t)mid;
                        *reinterpret_cast<uint16_t *>(top_base + (po >> 1) + (sl >> 1)) = (uint16_t)top;
                        any |= top;
                    };
                    planes([&](unsigned po, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
                        half(po, w0, w1, s0, any0);
                        if (has1) half(po, w2, w3, s1, any1);
                    });
                    if (any0) img[S.flag_off + 4u * (unsigned)((by * S.bx_n + bx0) >> 2) + 1u] = 1;
                    if (any1) img[S.flag_off + 4u * (unsigned)((by * S.bx_n + bx0 + 1) >> 2) + 1u] = 1;
                }
              }
            } else
            if (oy < HL && ox < pitchL) {
                const int by = oy >> ssh, iy = oy & ((1 << ssh) - 1);
                const int bx0 = ox >> ssh;
                // Address = uniform part (image, level, channel, filter pair: SGPRs, scalar ALU) + one 32-bit lane offset per
                // target block (tile of the block, block inside the tile, row inside the block, filter parity h), so that a
                // store costs no vector address arithmetic (global_store with an SGPR base).
                unsigned char *ubase = feats + (size_t)b * S.img_bytes + offL + (size_t)(st_c * FLv + fbase) * nplc * 2;
                const unsigned row_off = (unsigned)(((iy << ssh) + h * nplc) * 2);
                auto lane_off = [&](int p) -> unsigned {           // block bx0 + p, this lane's row inside it
                    const int blk = by * S.bx_n + bx0 + p;
                    return (unsigned)(blk >> 2) * (unsigned)S.tile_bytes + (unsigned)((blk & 3) << (2 * ssh)) * 2u + row_off;
                };
                auto planes = [&](auto &&put) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int fp = 0; fp < 2; ++fp) {
                            if (mt == MT - 1 && fp >= GQ) continue;
                            // an odd filter count leaves the h = 1 half of the LAST pair without a filter
                            const bool last = mt == MT - 1 && fp == GQ - 1;
                            if (last && fbase + 4 * mt + 2 * fp + h >= FLv) continue;
                            put(ubase + (size_t)(4 * mt + 2 * fp) * nplc * 2, outp[mt][fp][0], outp[mt][fp][1], outp[mt][fp][2],
                                outp[mt][fp][3]);
                        }
                };
                if (Lc == 0) {
                    // nontemporal: the slab (0.9 GB per 64 images) is read back only by the Lloyd passes; plain stores
                    // leave ~0.3 GB of it dirty in L2 / Infinity Cache and the first pass then shares HBM with their
                    // write-back (same-box A/B: first pass 0.219 -> 0.186 ms, step -2 %)
                    const unsigned o0 = lane_off(0);
                    planes([&](unsigned char *up, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
                        __builtin_nontemporal_store(v4i{(int)w0, (int)w1, (int)w2, (int)w3}, reinterpret_cast<v4i *>(up + o0));
                    });
                } else if (Lc == 1 || LVL == -2) {
                    const unsigned o0 = lane_off(0), o1 = lane_off(1);
                    const bool has1 = bx0 + 1 < S.bx_n;
                    planes([&](unsigned char *up, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
                        __builtin_nontemporal_store(v2i{(int)w0, (int)w1}, reinterpret_cast<v2i *>(up + o0));
                        if (has1) __builtin_nontemporal_store(v2i{(int)w2, (int)w3}, reinterpret_cast<v2i *>(up + o1));
                    });
                } else if (Lc == 2) {
                    unsigned o[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) o[p] = lane_off(p);
                    planes([&](unsigned char *up, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
                        const unsigned w[4] = {w0, w1, w2, w3};
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            if (bx0 + p < S.bx_n) *reinterpret_cast<unsigned *>(up + o[p]) = w[p];
                    });
                } else {
                    unsigned o[8];
#pragma unroll
                    for (int p = 0; p < 8; ++p) o[p] = lane_off(p);
                    planes([&](unsigned char *up, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
                        const unsigned w[4] = {w0, w1, w2, w3};
#pragma unroll
                        for (int p = 0; p < 8; ++p)
                            if (bx0 + p < S.bx_n)
                                *reinterpret_cast<uint16_t *>(up + o[p]) = (uint16_t)(w[p >> 1] >> (16 * (p & 1)));
                    });
                }
            }
        };
        constexpr int NT = 4 * MT;
        static_assert(NT % 2 == 0, "the accumulator ring must close over a block");
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) win[kk] = load_fragment(kk, 0, trow, 0, 0);
#pragma unroll 1
        for (int j = 0; j < nblk; ++j) {
            // next block (for the fragments of its first pixel pair; past the end: any valid address)
            const int jn = j + 1 < nblk ? j + 1 : 0;
            const int cn = rows1 ? jn >> 1 : jn, trown = wave * 8 + (rows1 ? (jn & 1) * 4 : 0) + lyy;
            // chain t = (window step ws, tile mt); window steps in the order (copy 0, qq 0), (copy 0, qq 1), (copy 1, qq 0),
            // (copy 1, qq 1): pixel pair pp = 2 qq + copy. The last chain of a window step refills every fragment register
            // right behind the MFMA that read it, with the fragment of the next step (of the next block after the last
            // step): one chain of lead for an LDS read, and one set of fragment registers instead of two.
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ws = t / MT, mt = t % MT;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t & 1][e] = 0;   // inline-constant C of the first MFMA
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    // slot kk of the region: one MFMA of chain t, the refill of the fragment it has just read (last chain of a
                    // window step), one slice of the epilogue of chain t-1 - for the first chain of a block that is the LAST
                    // chain of the previous block, whose stores follow it. The scheduling barrier after every slot keeps the
                    // MFMAs evenly spaced and the LDS reads behind the MFMA that frees their registers (hoisted, they need a
                    // second set of fragment registers: 22 spilled VGPRs at MT = 3).
                    acc[t & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[mt][kk], win[kk], acc[t & 1], 0, 0, 0);
                    if (mt == MT - 1)
                        win[kk] = ws < 3 ? load_fragment(kk, c, trow, (ws + 1) >> 1, (ws + 1) & 1) : load_fragment(kk, cn, trown, 0, 0);
                    if (t > 0) {
                        const int wp = (t - 1) / MT;
                        epi_slice(acc[(t - 1) & 1], (t - 1) % MT, 2 * (wp & 1) + (wp >> 1), kk);
                    } else if (j > 0) {
                        epi_slice(acc[(NT - 1) & 1], MT - 1, 3, kk);
                    }
                    if (kk == KS - 1) __builtin_amdgcn_sched_barrier(0);
                }
                if (t == 0 && j > 0) store_block();
            }
            st_c = c;
            st_trow = trow;
            c = cn;
            trow = trown;
        }
#pragma unroll
        for (int slot = 1; slot <= 5; ++slot) epi_slice(acc[(NT - 1) & 1], MT - 1, 3, slot);
        store_block();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the next tile has landed in LDS
      __syncthreads();   // next tile landed (vmcnt drained) and this buffer is free to refill
    }
}

// ---------------------------------------------------------------------------- edge strips
// The packed edge strips of the slab (csrc/common.h: right edge of one or two pixel columns, bottom edge of one or two rows,
// banks of at most two levels - every BSD500 image, both levels): 802 + 402 of a 481 x 321 image's pixels. Inside the tile
// grid of gabor_mfma_kernel they cost a whole extra tile column and tile row per level (84 + 22 instead of 75 + 20 tile
// times per image); here they are 87 small tasks per image. One wave = one task = 32 pixel PAIRS of one channel: for the
// column strip the pairs (WmL, WmL + 1) of 32 consecutive rows, for a row strip 64 consecutive pixels of the row. The window
// fragments come straight from the padded planes in global memory (any byte alignment; the planes were written a moment
// ago and sit in L2 / Infinity Cache), the A fragments from the packed bank, same K binding and accumulator layout as
// gabor_mfma_kernel; general (any shift) epilogue; every magnitude is stored as one uint16 at gcs_slab_offset.
struct StripLevel {
    const int8_t *planes;    // padded planes [B][3][Hp][Wp] of the level
    const int8_t *apack;     // the level's packed taps (all its row tiles)
    const int32_t *bias;
    int Hp, Wp, HL, WL;
    int col_x, nseg_col;     // column strip: level column col_x (and col_x + 1), rows [0, HL) in nseg_col segments of 32 rows
    int row_y, row_n;        // row strips: level rows [row_y, row_y + row_n) ...
    int row_w, nseg_row;     // ... columns [0, row_w), nseg_row segments of 64 pixels per row
    int task_end;            // end of this level's tasks in an image's task list (task = (segment, channel))
    int FL, MT, row0;        // filters of the level, its row tiles, its first physical plane
    int L;                   // the pyramid level
};
struct StripArgs {
    StripLevel lv[2];
    int n_levels, tasks_per_image;
};
template <int KS>
__global__ __launch_bounds__(256) void gabor_strip_kernel(StripArgs A, GcsLayout lo, int shift, unsigned char *__restrict__ feats,
                                                          int total_tasks) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);     // wave-uniform
    if (task >= total_tasks) return;
    const int b = task / A.tasks_per_image;
    int t = task - b * A.tasks_per_image;
    const int lvl = A.n_levels > 1 && t >= A.lv[0].task_end ? 1 : 0;
    const StripLevel v = lvl ? A.lv[1] : A.lv[0];
    if (lvl) t -= A.lv[0].task_end;
    const int seg = t / 3, c = t - 3 * seg;
    const int n = lane & 31, h = lane >> 5;
    int y, x, xlim;                                           // this lane's pixel pair (y, x), (y, x + 1); columns below xlim exist
    bool ok;
    if (seg < v.nseg_col) {
        y = 32 * seg + n;
        x = v.col_x;
        xlim = v.WL;
        ok = y < v.HL;
    } else {
        const int s2 = seg - v.nseg_col, rr = s2 / v.nseg_row, sg = s2 - rr * v.nseg_row;
        y = v.row_y + rr;
        x = 64 * sg + 2 * n;
        xlim = v.row_w;
        ok = x < v.row_w;
    }
    // window rows y + 2 kk + h of the padded plane (plane row r = level row r - 7), 16 columns from plane column x; lanes
    // without a pixel read a valid window and store nothing
    const int8_t *wp = v.planes + ((size_t)(b * 3 + c) * v.Hp + (ok ? y : 0) + h) * v.Wp + (ok ? x : 0);
    v4i win[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) win[kk] = *reinterpret_cast<const v4i_a1 *>(wp + (size_t)2 * kk * v.Wp);
    // Where the pair's magnitudes go: the slot of pixel (y, x) in the channel's first plane of this level; filter fl sits
    // fl planes further (same tile, same slot), the pair's second pixel in the next slot (x is even: slots ix, ix + 1 of a
    // main or strip block; level-1 parents q, q + 1 of one row of four): one 4-byte store per filter
    const size_t base = lo.split ? 0 : gcs_slab_offset(lo, b, v.row0 + c * v.FL, (ok ? y : 0) << v.L, (ok ? x : 0) << v.L);
    const int plane_bytes = (KP_TP >> (2 * v.L)) * 2;
    const bool both = x + 1 < xlim;
    // split slab: the slot of the pair's first pixel in the channel's first plane of this level (filter fl: fl planes further),
    // its nibble byte, whether the lane owns TWO adjacent slots (everything but the one-column right strip of level 1, whose
    // neighbouring slot is the next row's), how far the lane with the other half of the nibble group is, and who stores the bytes
    unsigned slot0 = 0, nbyte0 = 0;
    bool pairs = true, writer = false;
    int pdist = 1;
    if (lo.split) {
        slot0 = gcs_split_slot(lo, v.row0 + c * v.FL, (ok ? y : 0) << v.L, (ok ? x : 0) << v.L);
        int sh;
        gcs_split_nibble(v.L, slot0, nbyte0, sh);
        writer = sh == 0;
        const bool column = seg < v.nseg_col;
        pairs = !(column && v.L == 1);
        // lanes per half group: column strip, level 0: a block row holds 4 pixel rows (8 slots = 4 parents x 2 columns): 4 / 2 = 2
        // row pairs = 4 lanes; level 1: 4 parents = 4 lanes, half = 2. Row strip, level 0: 8 slots = 4 lanes, half = 2; level 1: 4
        // slots = 2 lanes, half = 1.
        pdist = column ? (v.L == 0 ? 4 : 2) : (v.L == 0 ? 2 : 1);
    }
    const v4i *ap = reinterpret_cast<const v4i *>(v.apack) + lane;
    v4i a_cur[KS], a_nxt[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) a_cur[kk] = ap[(size_t)kk * 64];
    for (int mt = 0; mt < v.MT; ++mt) {
        const int mn = mt + 1 < v.MT ? mt + 1 : mt;               // the next row tile's fragments travel during this one's work
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a_nxt[kk] = ap[((size_t)mn * 8 + kk) * 64];
        v16i acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_cur[kk], win[kk], acc, 0, 0, 0);
        // accumulator quad i = 2 fp + sx = {re_lo, re_hi, im_lo, im_hi} of filter 4 mt + 2 fp + h at pixel x + sx
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {
            const int fl = 4 * mt + 2 * fp + h;
            const int bias = v.bias[fl < v.FL ? fl : 0];
            unsigned mag[2];
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const int i = 2 * fp + sx;
                const int v_re = acc[4 * i + 1] * 256 + acc[4 * i + 0] + bias, v_im = acc[4 * i + 3] * 256 + acc[4 * i + 2];
                const int a_re = v_re >> shift, a_im = v_im >> shift;
                mag[sx] = isqrt31((unsigned)__mul24(a_re, a_re) + (unsigned)__mul24(a_im, a_im));
            }
            if (lo.split) {
                // Split slab: a slot's MID / TOP nibble shares its byte with the slot half a group further on (csrc/common.h), which
                // another lane of this wave computes: the lane `pdist` further on in the 32-lane half. Every lane hands its pair to
                // the shuffle (absent pixels as 0); the lanes whose slots are the FIRST half of their group (`writer`) store whole bytes.
                const unsigned pk = (ok && fl < v.FL) ? (mag[0] | (both ? mag[1] << 16 : 0u)) : 0u;
                const unsigned pt = (unsigned)__shfl_down((int)pk, pdist, 32);          // (every lane takes part)
                const unsigned pr = n + pdist < 32 ? pt : 0u;
                if (ok && fl < v.FL) {
                    unsigned char *img = feats + (size_t)b * lo.img_bytes;
                    const unsigned sl = slot0 + (unsigned)fl * (unsigned)(KP_TP >> (2 * v.L));
                    if (pairs) *reinterpret_cast<uint16_t *>(img + sl) = (uint16_t)(((pk & 0xffu) | ((pk >> 8) & 0xff00u)) ^ 0x8080u);
                    else img[sl] = (unsigned char)((pk & 0xffu) ^ 0x80u);
                    if (writer) {
                        // own high bytes (a, b) and the partner's (c, d): byte 0 = a | c << 4, byte 1 = b | d << 4 (nibble-wise)
                        const unsigned own = ((pk >> 8) & 0xffu) | ((pk >> 16) & 0xff00u), oth = ((pr >> 8) & 0xffu) | ((pr >> 16) & 0xff00u);
                        const unsigned mid = (own & 0x0f0fu) | ((oth << 4) & 0xf0f0u), top = ((own >> 4) & 0x0f0fu) | (oth & 0xf0f0u);
                        const unsigned nb = nbyte0 + (unsigned)fl * (unsigned)(KP_TP >> (2 * v.L + 1));
                        if (pairs) {
                            *reinterpret_cast<uint16_t *>(img + lo.mid_off + nb) = (uint16_t)mid;
                            *reinterpret_cast<uint16_t *>(img + lo.top_off + nb) = (uint16_t)top;
                        } else {
                            img[lo.mid_off + nb] = (unsigned char)mid;
                            img[lo.top_off + nb] = (unsigned char)top;
                        }
                        if (top & (pairs ? 0xffffu : 0xffu)) img[lo.flag_off + 4u * (slot0 / (unsigned)lo.S) + (unsigned)v.L] = 1;
                    }
                }
            } else
            if (ok && fl < v.FL) {
                unsigned char *dst = feats + base + (size_t)fl * plane_bytes;
                if (both) *reinterpret_cast<unsigned *>(dst) = (mag[0] | (mag[1] << 16)) ^ 0x80808080u;
                else *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(mag[0] ^ 0x8080u);
            }
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a_cur[kk] = a_nxt[kk];
    }
}

// ------------------------------------------------------------------------------ workspace
static inline int gabor_hp(int H) { return (H + G_TH - 1) / G_TH * G_TH + 15; }
static inline int gabor_wp(int W) { return (W + G_TW - 1) / G_TW * G_TW + 32; }

struct GaborWs {
    size_t plane_off[GCS_LEVELS_MAX];   // padded planes [B][3][Hp][Wp] int8 of level L
    size_t img_off[GCS_LEVELS_MAX];     // compact planar image [B][3][HL][WL] uint8 of level L (1 <= L <= n_levels-2)
    int Hp[GCS_LEVELS_MAX], Wp[GCS_LEVELS_MAX], HL[GCS_LEVELS_MAX], WL[GCS_LEVELS_MAX];
    size_t total;
};
static GaborWs gabor_ws(int B, int H, int W, int n_levels) {
    GaborWs ws{};
    size_t off = 0;
    int h = H, w = W;
    for (int L = 0; L < n_levels; ++L) {
        ws.HL[L] = h;
        ws.WL[L] = w;
        ws.Hp[L] = gabor_hp(h);
        ws.Wp[L] = gabor_wp(w);
        ws.plane_off[L] = off;
        off += ((size_t)B * 3 * ws.Hp[L] * ws.Wp[L] + 255) / 256 * 256;
        if (L >= 1 && L + 1 < n_levels) {
            ws.img_off[L] = off;
            off += ((size_t)B * 3 * h * w + 255) / 256 * 256;
        }
        h = (h + 1) / 2;
        w = (w + 1) / 2;
    }
    ws.total = off + 256;   // the second LDS copy of the last tile row reads two bytes past its plane row
    return ws;
}

// Side stream of gcs_gabor_features, one per device, created on first use and kept for the life of the process.
struct GaborSide {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool tried = false;
};
constexpr int GCS_MAX_DEVICES = 64;
constexpr long long GCS_GABOR_FORK_MIN_PIXELS = 1 << 21;
static std::mutex g_side_mu;
static GaborSide g_side[GCS_MAX_DEVICES];
static GaborSide *gabor_side() {                     // call with g_side_mu held; NULL: run on the caller's stream only
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GCS_MAX_DEVICES) return nullptr;
    GaborSide &g = g_side[dev];
    if (!g.tried) {
        g.tried = true;
        if (hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&g.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            g.s = nullptr;
        }
    }
    return g.s ? &g : nullptr;
}

// One launch of the bank kernel, for either slab format.
template <int MT, int GQ, int KS, int LV, bool FA>
static void gabor_launch(bool split, dim3 grid, dim3 block, hipStream_t st, const GaborLevels &G, int FLg, int f0, int shift,
                         unsigned char *feats, int total_tiles, const GaborSlab &slab) {
    if (split) {
        hipLaunchKernelGGL((gabor_mfma_kernel<MT, GQ, KS, LV, FA, true>), grid, block, 0, st, G, FLg, f0, shift, feats, total_tiles, slab);
        return;
    }
    hipLaunchKernelGGL((gabor_mfma_kernel<MT, GQ, KS, LV, FA, false>), grid, block, 0, st, G, FLg, f0, shift, feats, total_tiles, slab);
}

extern "C" size_t gcs_gabor_workspace_bytes(int B, int H, int W, int n_scales) {
    if (B <= 0 || H <= 0 || W <= 0 || n_scales < 1 || n_scales > GCS_SCALES_MAX) return 0;
    return gabor_ws(B, H, W, (n_scales + 1) / 2).total;
}

extern "C" int gcs_gabor_features(const uint8_t *img, int B, int H, int W, const int8_t *packed,
                                  const int32_t *bias, int n_scales, int n_orient, int ksize, int shift, void *workspace,
                                  uint16_t *feats, gcs_stream_t stream) {
    if (!img || !packed || !bias || !feats || !workspace)
        return gcs_fail(GCS_EINVAL, "gcs_gabor_features: NULL pointer");
    if (B <= 0) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: B must be > 0");
    if (H < 8 || W < 8) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: H and W must be >= 8");
    if (shift < 0 || shift > 23) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: shift out of range");
    if (ksize < 1 || ksize > GCS_KSIZE_MAX || (ksize & 1) == 0)
        return gcs_fail(GCS_EINVAL, "gcs_gabor_features: ksize must be odd and <= 15");
    if (B > 65535) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: B too large for one launch");
    GcsLayout lo;
    if (!gcs_make_layout(H, W, n_scales, n_orient, &lo))
        return gcs_fail(GCS_EINVAL, "gcs_gabor_features: need 1 <= n_scales <= 8, n_orient >= 1");
    if ((long long)B * lo.img_bytes > 0x7fffffffffffLL) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: slab too large");
    if ((long long)lo.ntiles * lo.tile_bytes > 0xffffffffLL)   // the kernel addresses one image's slab with 32-bit lane offsets
        return gcs_fail(GCS_EINVAL, "gcs_gabor_features: one image's feature slab must stay below 4 GiB");
    const GaborWs ws = gabor_ws(B, H, W, lo.n_levels);
    if (ws.Hp[0] / 4 + 1 > 65535) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: H too large for one launch");
    if ((size_t)W * 6 + 16 > 60 * 1024) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: W too large for the pyramid row buffer");
    unsigned char *wsb = static_cast<unsigned char *>(workspace);
    const dim3 block(256);
    // Main region of level L (what gabor_mfma_kernel's tile list covers) and whether gabor_strip_kernel has work
    const bool pack_r = lo.Wm != GCS_NO_STRIP, pack_b = lo.Hm != GCS_NO_STRIP, strips = pack_r || pack_b;
    auto region_h = [&](int L) { return pack_b ? lo.Hm >> L : ws.HL[L]; };
    auto region_w = [&](int L) { return pack_r ? lo.Wm >> L : ws.WL[L]; };
    auto half_tiles = [&](int L) { return (long long)((region_w(L) + G_TW / 2 - 1) / (G_TW / 2)) * ((region_h(L) + G_TH - 1) / G_TH); };
    // Two-level bank (the default): level 1 only needs the input images, so its pre-pass and its MFMA launch run on a side
    // stream forked from the caller's stream here and joined back before this call returns to it: beside the level-0
    // pre-pass and in the tail of the level-0 MFMA launch (same-box A/B, three runs: stage 0.552 -> 0.530, 0.565 -> 0.542,
    // 0.547 -> 0.532 ms per 64 images). Deeper banks share one MFMA launch that needs every level's planes first; forking
    // only their pre-passes costs more in cross-queue waits than it hides (8x8 bank 0.80 -> 0.87 ms): one stream. Small
    // batches stay on one stream too.
    GaborSide *sd = nullptr;
    std::unique_lock<std::mutex> side_lock;
    // Not while the caller's stream is being captured into a graph: the process-wide side stream would join that capture and
    // stay in capture mode until the caller ends it, long after the mutex below is released (a concurrent call from another
    // thread would then enqueue into a foreign capture). A captured step runs its levels on the one stream.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (lo.n_levels == 2 && (long long)B * H * W >= GCS_GABOR_FORK_MIN_PIXELS && cap == hipStreamCaptureStatusNone) {
        side_lock = std::unique_lock<std::mutex>(g_side_mu);     // the fork / join events are reused: one enqueue at a time
        sd = gabor_side();
        if (!sd) side_lock.unlock();
    }
    bool forked = false;
    if (sd) {
        if (hipEventRecord(sd->fork, stream) == hipSuccess && hipStreamWaitEvent(sd->s, sd->fork, 0) == hipSuccess) forked = true;
        else { (void)hipGetLastError(); sd = nullptr; side_lock.unlock(); }
    }
    auto join = [&]() -> int {                               // the caller's stream waits for everything on the side stream
        if (!forked) return GCS_OK;
        forked = false;
        hipError_t e = hipEventRecord(sd->join, sd->s);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, sd->join, 0);
        side_lock.unlock();
        return e == hipSuccess ? GCS_OK : gcs_hip_fail(e, "gcs_gabor_features(join)");
    };
#define GCS_STREAM_OF(L) ((forked && (L) >= 1) ? sd->s : stream)
#define GCS_GABOR_CHECK(what)                                    \
    do {                                                         \
        hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) {                                  \
            (void)join();                                        \
            return gcs_hip_fail(e_, what);                       \
        }                                                        \
    } while (0)
    // A call so small that the tiles of ALL its levels fit the resident slots at once (one to four BSD images) also takes
    // the fused list: level 1's tiles then run beside level 0's instead of in a launch of their own behind them (one image:
    // the two launches take 19 us each, one after the other; the slot is called once per image, script.py:22-30).
    long long tiles_all = 0;
    for (int L = 0; L < lo.n_levels; ++L) tiles_all += (long long)B * ((half_tiles(L) + 1) / 2);
    // (not for a split-slab bank whose level is three row tiles that cannot run as groups - 9 to 11 filters per level -: the fused
    //  three-tile kernel with both split store paths live spills 50 VGPRs; such a call takes the two single-level launches)
    const bool fuse_small = lo.n_levels == 2 && tiles_all <= 2LL * gcs_cu_count() &&
                            !(lo.split && mtiles(lo.FL[0]) == 3 && lo.FL[0] % 4 != 0);
    // split slab: which flag bytes the pre-pass of level L clears (csrc/common.h; level 0 also those of absent levels)
    auto flag_zero = [&](int zmask) {
        GaborFlagZero z{};
        if (lo.split) z = GaborFlagZero{reinterpret_cast<unsigned char *>(feats), lo.img_bytes, lo.flag_off, lo.ntiles, zmask};
        return z;
    };
    const int zmask0 = 1 | (0xf & ~((1 << lo.n_levels) - 1));
    // ---- pre-passes: the padded planes of every level (level L >= 2 reads level L-1's compact image)
    for (int L = 0; L < lo.n_levels; ++L) {
        if (fuse_small && !forked) {                           // a small call: both pre-passes of the two-level bank in one launch
            if (L == 1) continue;
            const int n0 = (ws.Hp[0] * (ws.Wp[0] / 16) + 511) / 512;
            const size_t lds = 2 * (size_t)((W * 3 + 6) & ~3);
            hipLaunchKernelGGL(gabor_pre01_kernel, dim3(n0 + ws.Hp[1], B), block, lds, stream, img, (size_t)B * H * W * 3, H, W, n0,
                               ws.HL[0], ws.WL[0], ws.Hp[0], ws.Wp[0], reinterpret_cast<int8_t *>(wsb + ws.plane_off[0]),
                               ws.HL[1], ws.WL[1], ws.Hp[1], ws.Wp[1], reinterpret_cast<int8_t *>(wsb + ws.plane_off[1]),
                               flag_zero(zmask0 | 2));
            GCS_GABOR_CHECK("gcs_gabor_features(pad)");
            continue;
        }
        int8_t *planes = reinterpret_cast<int8_t *>(wsb + ws.plane_off[L]);
        const int HL = ws.HL[L], WL = ws.WL[L], Hp = ws.Hp[L], Wp = ws.Wp[L];
        // level-0 pre-pass: about two interior items (16 bytes x 3 channels) per thread and loop round
        const int pitems = Hp * (Wp / 16);
        const dim3 pgrid((pitems + 511) / 512, B);
        uint8_t *img_out = (L >= 1 && L + 1 < lo.n_levels) ? wsb + ws.img_off[L] : nullptr;
        if (L == 0)
            hipLaunchKernelGGL((gabor_plane_kernel<0>), pgrid, block, 0, GCS_STREAM_OF(L), img, H, W, HL, WL, Hp, Wp, planes,
                               (uint8_t *)nullptr, flag_zero(zmask0));
        else if (L == 1) {
            const size_t lds = 2 * (size_t)((W * 3 + 6) & ~3);
            hipLaunchKernelGGL((gabor_down_kernel<true>), dim3(Hp, B), block, lds, GCS_STREAM_OF(L), img, (size_t)B * H * W * 3, H, W,
                               HL, WL, Hp, Wp, planes, img_out, flag_zero(2));
        } else {
            const int Hs = ws.HL[L - 1], Ws = ws.WL[L - 1];
            const size_t lds = 6 * (size_t)((Ws + 6) & ~3);
            hipLaunchKernelGGL((gabor_down_kernel<false>), dim3(Hp, B), block, lds, GCS_STREAM_OF(L),
                               (const uint8_t *)(wsb + ws.img_off[L - 1]), (size_t)B * 3 * Hs * Ws, Hs, Ws, HL, WL, Hp, Wp,
                               planes, img_out, flag_zero(0));
        }
        GCS_GABOR_CHECK("gcs_gabor_features(pad)");
    }
    // ---- the bank: one launch per run of levels with the same filter count (every level of an even-scale bank)
    int mt_base[GCS_LEVELS_MAX];
    for (int L = 0, m = 0; L < lo.n_levels; ++L) {
        mt_base[L] = m;
        m += mtiles(lo.FL[L]);
    }
    const int mtmax = ksize <= 13 ? GCS_GABOR_MTMAX : 2;     // A operand: MT x KS x 4 VGPRs (84 for 3 x 7, 64 for 2 x 8)
    auto launch_group = [&](int L0, int &L1) -> int {
        L1 = L0 + 1;
        // fused lists pay ~2 % for level fields that are no longer launch constants and win the small levels' ramp and
        // tail back: a gain from three levels on (8x8 bank: 0.90 -> 0.80 ms), a small loss for two (0.552 -> 0.557 ms)
        if (lo.n_levels > 2 || fuse_small)
            while (L1 < lo.n_levels && lo.FL[L1] == lo.FL[L0]) ++L1;
        const int FLg = lo.FL[L0], MT = mtiles(FLg);
        if (L0 == 0 && L1 > 1)                       // this launch reads planes the side stream is still writing
            if (int rc = join()) return rc;
        // three row tiles only for single launches of level 0 / 1 (compile-time level): with the level a run-time value the
        // store path of every level is live and a third tile's 28 A registers spill
        // (a fused list of exactly levels 0 and 1 - LVL = -2: two store paths - keeps the third tile too)
        const bool two_fused = lo.n_levels == 2 && L0 == 0 && L1 == 2;
        const int mtmax_here = ((L1 - L0 == 1 && L0 <= 1) || two_fused) ? mtmax : 2;
        // a small call (fuse_small) whose filters fill whole row tiles: ONE launch of MT groups of one-tile workgroups (blockIdx.y)
        const bool grouped = fuse_small && FLg % 4 == 0 && MT > 1;
        const int passes = grouped ? 1 : (MT + mtmax_here - 1) / mtmax_here;
        const int per_pass = grouped ? MT : (MT + passes - 1) / passes;   // 4 tiles -> 2 + 2, not 3 + 1
        for (int mt0 = 0; mt0 < MT; mt0 += per_pass) {
            const int n = grouped ? 1 : MT - mt0 >= per_pass ? per_pass : MT - mt0;
            // filters of this launch: [4*mt0, min(FL, 4*(mt0+n))) of each level (planes c*FL + f)
            const int fl_here = FLg - 4 * mt0 < 4 * n ? FLg - 4 * mt0 : 4 * n;
            const int gq = (fl_here - 4 * (n - 1) + 1) / 2;
            GaborLevels G{};
            long long total_ll = 0;
            for (int L = L0; L < L1; ++L) {
                GaborLevel &v = G.lv[L - L0];
                v.planes = reinterpret_cast<const int8_t *>(wsb + ws.plane_off[L]);
                v.apack = packed + (size_t)(mt_base[L] + mt0) * 8 * 64 * 16;
                v.bias = bias + (size_t)(mt_base[L] + mt0) * 4;
                v.HLm = region_h(L);
                v.Hp = ws.Hp[L];
                v.Wp = ws.Wp[L];
                v.pitchLm = pack_r ? region_w(L) : round_up(ws.WL[L], 8);
                v.htx = (region_w(L) + G_TW / 2 - 1) / (G_TW / 2);
                v.hcount = (int)half_tiles(L);
                v.tiles_per_image = (v.hcount + 1) / 2;
                total_ll += (long long)v.tiles_per_image * B;
                if (total_ll > 0x3fffffffLL) return gcs_fail(GCS_EINVAL, "gcs_gabor_features: too many tiles");
                v.tile_end = (int)total_ll;
                v.L = L;
                v.offL = lo.split ? lo.sl0[L] : lo.off[L];
            }
            for (int i = L1 - L0; i < GCS_LEVELS_MAX; ++i) {      // unused entries: never selected (tile < total_tiles)
                G.lv[i] = G.lv[L1 - L0 - 1];
                G.lv[i].tile_end = 0x7fffffff;
            }
            const int total_tiles = (int)total_ll;
            const GaborSlab slab{lo.bx_n, lo.ntiles, lo.tile_bytes, lo.img_bytes, lo.S, lo.mid_off, lo.top_off, lo.flag_off};
            // persistent grid: one workgroup per resident slot (two 54 KB workgroups per CU)
            const int slots = gcs_cu_count() * 2;
            const dim3 grid(total_tiles < slots ? total_tiles : slots, grouped ? MT : 1);
#define GCS_GABOR_LAUNCH4(MT_, GQ_, KS_, LV_, FA_)                                                                     \
    gabor_launch<MT_, GQ_, KS_, LV_, FA_>(lo.split != 0, grid, block, GCS_STREAM_OF(L0), G, FLg, 4 * mt0, shift,          \
                                          reinterpret_cast<unsigned char *>(feats), total_tiles, slab)
            // single launches of level 0 / level 1 (every bank of at most two levels) compile that level's store path alone
#define GCS_GABOR_LAUNCH3(MT_, GQ_, KS_, FA_)                                       \
    do {                                                                            \
        if (L1 - L0 == 1 && L0 == 0) GCS_GABOR_LAUNCH4(MT_, GQ_, KS_, 0, FA_);      \
        else if (L1 - L0 == 1 && L0 == 1) GCS_GABOR_LAUNCH4(MT_, GQ_, KS_, 1, FA_); \
        else GCS_GABOR_LAUNCH4(MT_, GQ_, KS_, -1, FA_);                             \
    } while (0)
            // the short epilogue needs shift == 8 (bytes 1-2 of the low accumulator) and no unused accumulator quad
#define GCS_GABOR_LAUNCH2(MT_, KS_)                                       \
    do {                                                                  \
        if (gq == 1) GCS_GABOR_LAUNCH3(MT_, 1, KS_, false);               \
        else if (shift == 8) GCS_GABOR_LAUNCH3(MT_, 2, KS_, true);        \
        else GCS_GABOR_LAUNCH3(MT_, 2, KS_, false);                       \
    } while (0)
            // 7 K-steps need the kernel inside rows 1..13 of the 15-row frame (ksize <= 13)
            if (ksize <= 13) {
                if (n == 3) {                    // single launch of level 0 or 1, or the fused list of both (mtmax_here)
#define GCS_GABOR_LAUNCH3T(GQ_, FA_)                                        \
    do {                                                                    \
        if (two_fused) GCS_GABOR_LAUNCH4(3, GQ_, 7, -2, FA_);               \
        else if (L0 == 0) GCS_GABOR_LAUNCH4(3, GQ_, 7, 0, FA_);             \
        else GCS_GABOR_LAUNCH4(3, GQ_, 7, 1, FA_);                          \
    } while (0)
                    if (gq == 1) GCS_GABOR_LAUNCH3T(1, false);
                    else if (shift == 8) GCS_GABOR_LAUNCH3T(2, true);
                    else GCS_GABOR_LAUNCH3T(2, false);
#undef GCS_GABOR_LAUNCH3T
                }
                else if (n == 2) GCS_GABOR_LAUNCH2(2, 7);
                else GCS_GABOR_LAUNCH2(1, 7);
            } else {
                if (n == 2) GCS_GABOR_LAUNCH2(2, 8);
                else GCS_GABOR_LAUNCH2(1, 8);
            }
#undef GCS_GABOR_LAUNCH4
#undef GCS_GABOR_LAUNCH3
#undef GCS_GABOR_LAUNCH2
            GCS_GABOR_CHECK("gcs_gabor_features");
        }
        return GCS_OK;
    };
    // The packed edge strips (57 + 30 tasks of a few microseconds per BSD image), one small launch per level on the level's
    // stream IN FRONT of its MFMA launch: level 0's right behind the plane pre-pass, while the chip is still empty (a launch
    // behind the MFMA kernels cost 27 us at the end of the stage; reserving four CUs for it beside level 0 held level 1 back:
    // profiles/r4_notes.md), level 1's on the side stream in the tail of level 0 like level 1 itself.
    auto launch_strips = [&](int L0s, int L1s) -> int {
        StripArgs A{};
        A.n_levels = L1s - L0s;
        int tasks = 0;
        for (int L = L0s; L < L1s; ++L) {
            StripLevel &v = A.lv[L - L0s];
            v.planes = reinterpret_cast<const int8_t *>(wsb + ws.plane_off[L]);
            v.apack = packed + (size_t)mt_base[L] * 8 * 64 * 16;
            v.bias = bias + (size_t)mt_base[L] * 4;
            v.Hp = ws.Hp[L]; v.Wp = ws.Wp[L]; v.HL = ws.HL[L]; v.WL = ws.WL[L];
            v.col_x = pack_r ? lo.Wm >> L : 0;
            v.nseg_col = pack_r ? (ws.HL[L] + 31) / 32 : 0;
            v.row_y = pack_b ? lo.Hm >> L : 0;
            v.row_n = pack_b ? ws.HL[L] - (lo.Hm >> L) : 0;
            v.row_w = pack_r ? lo.Wm >> L : ws.WL[L];
            v.nseg_row = (v.row_w + 63) / 64;
            tasks += 3 * (v.nseg_col + v.row_n * v.nseg_row);
            v.task_end = tasks;
            v.FL = lo.FL[L]; v.MT = mtiles(lo.FL[L]); v.row0 = lo.row0[L]; v.L = L;
        }
        if (A.n_levels == 1) A.lv[1] = A.lv[0];
        A.tasks_per_image = tasks;
        const long long total = (long long)tasks * B;
        if (total > 0x3fffffffLL) { (void)join(); return gcs_fail(GCS_EINVAL, "gcs_gabor_features: too many strip tasks"); }
        hipStream_t ss = GCS_STREAM_OF(L0s);
        const dim3 sgrid((unsigned)((total + 3) / 4));
        if (ksize <= 13)
            hipLaunchKernelGGL((gabor_strip_kernel<7>), sgrid, block, 0, ss, A, lo, shift, reinterpret_cast<unsigned char *>(feats), (int)total);
        else
            hipLaunchKernelGGL((gabor_strip_kernel<8>), sgrid, block, 0, ss, A, lo, shift, reinterpret_cast<unsigned char *>(feats), (int)total);
        GCS_GABOR_CHECK("gcs_gabor_features(strips)");
        return GCS_OK;
    };
    for (int L0 = 0, L1 = 0; L0 < lo.n_levels; L0 = L1) {
        // (a fused list - L1 > L0 + 1 - joins the side stream first: every level's planes are then ready on `stream`)
        int Lend = L0 + 1;
        if (lo.n_levels > 2 || fuse_small)
            while (Lend < lo.n_levels && lo.FL[Lend] == lo.FL[L0]) ++Lend;
        if (strips) {
            if (L0 == 0 && Lend > 1)
                if (int rc = join()) return rc;
            if (int rc = launch_strips(L0, Lend < 2 ? Lend : 2)) return rc;
        }
        if (int rc = launch_group(L0, L1)) return rc;
    }
    if (int rc = join()) return rc;
#undef GCS_GABOR_CHECK
#undef GCS_STREAM_OF
    return GCS_OK;
}

// ------------------------------------------------------------------------------- unpack
// Slab -> canonical [B][D][H][W] uint16 of SPEC.md §3 (level-L responses replicated over 2^L blocks).
__global__ void unpack_kernel(const unsigned char *__restrict__ feats, GcsLayout lo, size_t planes,
                              uint16_t *__restrict__ out) {
    const size_t hw = (size_t)lo.H * lo.W;
    const size_t n = planes * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pl = i / hw;
        const int rem = (int)(i % hw);
        const int b = (int)(pl / lo.D), d = (int)(pl % lo.D);
        out[i] = (uint16_t)gcs_slab_value(feats, lo, b, gcs_plane_of_logical(lo, d), rem / lo.W, rem % lo.W);
    }
}

extern "C" int gcs_features_unpack(const uint16_t *feats, int B, int H, int W, int n_scales, int n_orient,
                                   uint16_t *out, gcs_stream_t stream) {
    if (!feats || !out) return gcs_fail(GCS_EINVAL, "gcs_features_unpack: NULL pointer");
    GcsLayout lo;
    if (B <= 0 || !gcs_make_layout(H, W, n_scales, n_orient, &lo))
        return gcs_fail(GCS_EINVAL, "gcs_features_unpack: bad shape");
    hipLaunchKernelGGL(unpack_kernel, dim3(2048), dim3(256), 0, stream, reinterpret_cast<const unsigned char *>(feats),
                       lo, (size_t)B * lo.D, out);
    GCS_CHECK_LAUNCH("gcs_features_unpack");
    return GCS_OK;
}
