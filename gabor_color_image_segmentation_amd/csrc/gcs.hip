// libgcs — gfx950 (MI355X) kernels for the Gabor-bank + k-means segmenter path and their
// C ABI (include/gcs.h). Arithmetic: SPEC.md (exact integers). The reference ships no code
// for this path; the slot filled is /root/reference/BSD_metrics/script.py:30.
//
// Kernels
//   gabor_pad_kernel         interleaved RGB -> planar (pixel - 128) with the reflect border materialised.
//   gabor_mfma_kernel        im2col GEMM on v_mfma_i32_32x32x32_i8: A = packed 2-digit int8 taps (rows = filter x
//                            {re_lo,re_hi,im_lo,im_hi}, resident in registers), B = (pixel-128) windows built from
//                            an LDS tile (LDS-DMA double buffer) by dword reads + v_alignbit, exact int32 accumulate,
//                            fused epilogue (digit recombine, >>shift, |.|^2, exact isqrt) -> tile-major u16 slab.
//   kmeans_pass_mfma_kernel  one Lloyd pass (assign + update) on the matrix cores for D <= 207; HBM-bound stream.
//   kmeans_assign_kernel     generic pass for D >= 208: exact integer argmin via fp32 byte-digit FMAs (all partial
//                            sums < 2^24, hence exact), LDS-replicated u32 accumulators.
//   kmeans_reduce_kernel     element-major partial sums -> int64 sums (+ the centroid update when single-rank).
//   kmeans_finalize / init / features_gather / unpack / widen: small helpers.
//   boundary_* , cc_*        boundary recall/precision counts (metrics.py:58-96) and connected regions (SPEC.md 7).
// Nothing here allocates, frees or synchronises; every entry point enqueues on the caller's stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gcs.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------- errors
static thread_local char g_err[256] = "";
static int fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
static int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return GCS_EHIP;
}
#define GCS_CHECK_LAUNCH(what)                          \
    do {                                                \
        hipError_t e_ = hipGetLastError();              \
        if (e_ != hipSuccess) return hip_fail(e_, what); \
    } while (0)

extern "C" int gcs_abi_version(void) { return GCS_ABI_VERSION; }
extern "C" const char *gcs_last_error(void) { return g_err; }

#ifndef GCS_KP_TP
#define GCS_KP_TP 256
#endif
// ------------------------------------------------------------------------- geometry (host)
static inline int round_up(int a, int m) { return (a + m - 1) / m * m; }
static inline int mtiles(int F) { return (F + 7) / 8; }

extern "C" size_t gcs_bank_packed_bytes(int F) { return F > 0 ? (size_t)mtiles(F) * 8 * 64 * 16 : 0; }
extern "C" size_t gcs_bank_bias_count(int F) { return F > 0 ? (size_t)mtiles(F) * 8 : 0; }
extern "C" size_t gcs_feature_pitch(int W) { return W > 0 ? (size_t)round_up(W, 8) : 0; }
// Pixels per feature plane / label plane: H*pitch rounded up to the k-means tile (256 px), so
// a tile never crosses a plane boundary and staging needs no bounds checks.
extern "C" size_t gcs_feature_plane_stride(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    return ((size_t)H * gcs_feature_pitch(W) + 255) / 256 * 256;
}
extern "C" size_t gcs_feature_slab_bytes(int B, int H, int W, int D) {
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0) return 0;
    return (size_t)B * D * gcs_feature_plane_stride(H, W) * sizeof(uint16_t);
}
extern "C" size_t gcs_label_slab_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * gcs_feature_plane_stride(H, W);
}
// Workgroups (= partial-sum rows) per image of one Lloyd pass. Sized to the machine: the pass
// kernel runs 3 workgroups per CU, so B*parts aims at one full wave of 256*3 workgroups (a
// second, partly filled wave of workgroups would idle most of the chip). Lower bound: one
// workgroup's int32 accumulators must not overflow (<= 65536 pixels); upper bound: at least
// 8 tiles per workgroup to amortise its prologue.
#ifndef GCS_KP_SLOTS
#define GCS_KP_SLOTS (768 * 256 / GCS_KP_TP)
#endif
extern "C" size_t gcs_kmeans_parts_per_image(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t px = gcs_feature_plane_stride(H, W);
    const size_t need = (px + 65535) / 65536;
    size_t most = px / GCS_KP_TP / 8;
    if (most < 1) most = 1;
    size_t want = (GCS_KP_SLOTS + (size_t)B - 1) / (size_t)B;
    if (want > most) want = most;
    return need > want ? need : want;
}
extern "C" size_t gcs_kmeans_partial_bytes(int B, int H, int W, int D, int k) {
    if (B <= 0 || D <= 0 || k <= 0) return 0;
    return (size_t)B * gcs_kmeans_parts_per_image(B, H, W) * k * (D + 1) * sizeof(uint64_t);
}

// ------------------------------------------------------------------------ bank pack (host)
// A-fragment of v_mfma_i32_32x32x32_i8: lane l = (r = l&31, h = l>>5) supplies row r of the
// 32-row tile, 16 consecutive k. We bind k-slot (kk, h, j) to tap (dy = 2*kk + h, dx = j) of
// a 16x16 frame whose row 15 / column 15 are zero; both operands use the same binding, so
// only "A row = lane&31, B col = lane&31" and the C/D map (cdna guide §3) are relied on.
// Row r of tile mt = filter 8*mt + r/4, part r%4 in {re_lo, re_hi, im_lo, im_hi}: the four
// parts of one filter land in one lane's accumulator quad (rows 4g..4g+3).
extern "C" int gcs_bank_pack(const int16_t *tapq, int F, int ks, int8_t *packed, int32_t *bias) {
    if (!tapq || !packed || !bias) return fail(GCS_EINVAL, "gcs_bank_pack: NULL pointer");
    if (F <= 0) return fail(GCS_EINVAL, "gcs_bank_pack: n_filters must be > 0");
    if (ks < 1 || ks > GCS_KSIZE_MAX || (ks & 1) == 0)
        return fail(GCS_EINVAL, "gcs_bank_pack: ksize must be odd and <= 15");
    const int MT = mtiles(F);
    const int off = (GCS_KSIZE_MAX - ks) / 2; // centre smaller kernels in the 15x15 frame
    memset(packed, 0, gcs_bank_packed_bytes(F));
    memset(bias, 0, gcs_bank_bias_count(F) * sizeof(int32_t));
    for (int f = 0; f < F; ++f) {
        long s_re = 0, s_im = 0;
        for (int t = 0; t < ks * ks; ++t) {
            s_re += tapq[((size_t)f * 2 + 0) * ks * ks + t];
            s_im += tapq[((size_t)f * 2 + 1) * ks * ks + t];
        }
        if (s_im != 0) return fail(GCS_EINVAL, "gcs_bank_pack: imaginary taps must sum to zero");
        bias[f] = (int32_t)(128 * s_re);
    }
    for (int mt = 0; mt < MT; ++mt)
        for (int kk = 0; kk < 8; ++kk)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, h = lane >> 5;
                const int f = 8 * mt + r / 4, part = r & 3;
                int8_t *dst = packed + (((size_t)mt * 8 + kk) * 64 + lane) * 16;
                if (f >= F) continue;
                const int dy = 2 * kk + h - off;
                if (dy < 0 || dy >= ks) continue;
                for (int j = 0; j < 16; ++j) {
                    const int dx = j - off;
                    if (dx < 0 || dx >= ks) continue;
                    const int q = tapq[(((size_t)f * 2 + (part >> 1)) * ks + dy) * ks + dx];
                    if (q > 32639 || q < -32639)
                        return fail(GCS_EINVAL, "gcs_bank_pack: tap outside two-digit range");
                    const int lo = ((q + 128) & 255) - 128;
                    const int hi = (q - lo) >> 8;
                    dst[j] = (int8_t)((part & 1) ? hi : lo);
                }
            }
    return GCS_OK;
}

// Feature slab addressing (tile-major): [B][tile][D][KP_TP px] uint16, tile = flat pixel index
// pp = y*pitch + x divided by KP_TP (the k-means tile). One k-means tile (all D planes of KP_TP pixels)
// is a single contiguous run; 8-pixel (16-byte) groups never straddle a tile.
#ifndef GCS_KP_TP
#define GCS_KP_TP 256
#endif
constexpr int KP_TP = GCS_KP_TP;          // pixels per k-means tile = threads per k-means workgroup (64 px per wave)
__device__ __forceinline__ size_t slab_index(size_t image_tile0, int D, int d, int pp) {
    return ((image_tile0 + (size_t)(pp / KP_TP)) * D + d) * KP_TP + (pp % KP_TP);
}

// Partial sums, element-major: [set][element i of k*(D+1)][rows of that set] uint64, rows = the workgroups that
// contribute to the set (per-image codebooks: the image's `parts`; one codebook: all gridDim.y * parts). The 768
// values an output element is summed from are then one contiguous 6 KB run for kmeans_reduce_kernel.
__device__ __forceinline__ size_t partial_index(int per_image, int b, int part, int parts, int i, int row_len) {
    return per_image ? ((size_t)b * row_len + i) * parts + part
                     : (size_t)i * ((size_t)gridDim.y * parts) + (size_t)b * parts + part;
}

// ================================================================================ Gabor
constexpr int G_TW = 64;            // output tile width  (4 lanes-in-x * 16 shifts)
constexpr int G_TH = 32;            // output tile height (4 waves * 8 rows)
constexpr int G_HALO = 7;
constexpr int G_LROWS = G_TH + 15;  // 47 rows: halo 14 + the zero-tap row 15
constexpr int G_LPITCH = 96;        // bytes per LDS tile row (>= 64 + 16 + 12)


__device__ __forceinline__ int reflect_clamp(int i, int n) {
    if (i < 0) i = -1 - i;
    if (i >= n) i = 2 * n - 1 - i;
    return min(max(i, 0), n - 1); // only reached for pixels whose outputs are not stored
}

// floor(sqrt(n)) for n <= 2 * 32642^2 < 2^31 (SPEC.md §3 bound), exact, in 7 VALU ops
// (measured on gfx950, tools/ubench/valu_ops2: v_cvt_u32_f32 ~3 ns and v_cmp+v_addc ~4.3 ns per
// wave-instruction, against ~1.2 ns for an add):
//   r    = v_sqrt_f32(float(n))        |r - s| <= 1.5e-7 * s <= 0.007 < 0.5   (s = true root)
//   bits = r + 2^23 (as uint)          the sum has ulp 1: bits = 0x4B000000 + RNE(r), RNE(r) in {floor(s), floor(s)+1}
//   qr^2 = v_mul_u32_u24(bits, bits)   the multiplier only sees the low 24 bits, i.e. qr = RNE(r) (< 2^16)
//   q    = qr - (qr^2 > n)             sign arithmetic, one v_add3: bits - 0x4B000000 + ((int)(n - qr^2) >> 31);
//                                      qr <= 46164 so qr^2 < 2^31 and the signed difference cannot overflow.
// 7 VALU instructions; in this kernel every VALU instruction costs ~4.2 cycles whatever its kind (PMC),
// so the count is what matters.
__device__ __forceinline__ unsigned isqrt31(unsigned n) {
    const unsigned bits = __float_as_uint(__builtin_amdgcn_sqrtf((float)n) + 8388608.0f);
    const int d = (int)(n - __umul24(bits, bits));
    return bits - 0x4B000000u + (unsigned)(d >> 31);
}

// Pre-pass: interleaved uint8 RGB -> planar (pixel - 128) int8 with the reflect border and
// the tile over-read already materialised: plane[b][c][r][u] = img[b][refl(r-7)][refl(u-7)][c] - 128
// for r < tiles_y*32 + 15, u < tiles_x*64 + 32. The main kernel then stages tiles with aligned
// 16-byte copies and no index arithmetic. ~26 B of extra HBM traffic per pixel-channel row: noise.
__global__ __launch_bounds__(256) void gabor_pad_kernel(const uint8_t *__restrict__ img, int H, int W, int Hp,
                                                        int Wp, int8_t *__restrict__ planes) {
    const int b = blockIdx.z, r = blockIdx.y;
    const int gy = reflect_clamp(r - G_HALO, H);
    const uint8_t *row = img + ((size_t)b * H + gy) * W * 3;
    for (int u4 = blockIdx.x * blockDim.x + threadIdx.x; u4 < Wp / 4; u4 += gridDim.x * blockDim.x) {
        unsigned o[3] = {0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gx = reflect_clamp(4 * u4 + e - G_HALO, W);
            const uint8_t *p = row + (size_t)gx * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] |= (unsigned)(p[c] ^ 0x80) << (8 * e);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<unsigned *>(planes + (((size_t)b * 3 + c) * Hp + r) * Wp + 4 * u4) = o[c];
    }
}

#ifndef GCS_GABOR_WAVES
#define GCS_GABOR_WAVES 2
#endif
constexpr int GCS_GABOR_MTMAX = 2;   // row tiles per launch: 3 needs ~250 VGPRs at 2 waves/SIMD and spills (measured slower)
template <int MT, bool FULLF>   // FULLF: n_filters is a multiple of 8 -> no per-filter store guard
__global__ __launch_bounds__(256, GCS_GABOR_WAVES) void gabor_mfma_kernel(
    const int8_t *__restrict__ planes, int H, int Hp, int Wp, const int8_t *__restrict__ apack,
    const int32_t *__restrict__ bias, int mt0, int F, int shift, uint16_t *__restrict__ feats, int pitch,
    size_t pstride, int tiles_x, int tiles_per_image, int total_tiles) {
    // Persistent workgroups: the A operand and the biases are loaded ONCE, then the workgroup walks
    // tiles blockIdx.x, +gridDim.x, ... ; the next tile streams into the other LDS buffer by LDS-DMA
    // (global_load_lds: no VGPRs, lands while this tile computes). The tile image is a flat run of
    // 846 16-byte chunks, i.e. exactly the lane-linear destination LDS-DMA wants.
    __shared__ __attribute__((aligned(16))) int8_t s_tile[2][3][G_LROWS][G_LPITCH];
    constexpr int NCHUNK = 3 * G_LROWS * (G_LPITCH / 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int D = 3 * F;

    auto stage_tile = [&](int tile, int buf) {
        const int b_ = tile / tiles_per_image, rem = tile % tiles_per_image;
        const int y0_ = (rem / tiles_x) * G_TH, x0_ = (rem % tiles_x) * G_TW;
        const int8_t *src0 = planes + ((size_t)b_ * 3 * Hp + y0_) * Wp + x0_;
#pragma unroll
        for (int k = 0; k < (NCHUNK + 255) / 256; ++k) {
            const int i = tid + 256 * k;
            if (i < NCHUNK) {
                const int ch16 = i % (G_LPITCH / 16), rc = i / (G_LPITCH / 16);
                const int row = rc % G_LROWS, c = rc / G_LROWS;
                const int8_t *g = src0 + ((size_t)c * Hp + row) * Wp + 16 * ch16;
                // LDS destination: wave-uniform base (this wave's first chunk) + lane * 16
                int8_t *l = &s_tile[buf][0][0][0] + 16 * (256 * k + 64 * wave);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)g,
                    (__attribute__((address_space(3))) void *)l, 16, 0, 0);
            }
        }
    };

    // ---- the whole A operand lives in registers: MT x 8 lane-linear 16-byte fragments
    v4i afr[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            afr[mt][kk] = reinterpret_cast<const v4i *>(apack)[((size_t)(mt0 + mt) * 8 + kk) * 64 + lane];

    // Pixel columns of one MFMA N-tile: x = x0 + 8*li + s (li = 0..7), y = row0 + lyy (lyy = 0..3),
    // for a pixel shift s = 4*qq + t in 0..7. A lane therefore ends up owning 8 consecutive
    // pixels (16 bytes) per filter and a store instruction writes whole 128-byte lines.
    const int r = lane & 31, h = lane >> 5;
    const int li = r & 7, lyy = r >> 3;

    int bias_v[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_v[mt][g] = bias[8 * (mt0 + mt) + 2 * g + h];

    int tile = blockIdx.x;
    if (tile < total_tiles) stage_tile(tile, 0);
    __syncthreads();                       // drains the LDS-DMA (vmcnt) and orders it for every wave
    for (int it = 0; tile < total_tiles; tile += gridDim.x, ++it) {
      const int buf = it & 1;
      if (tile + (int)gridDim.x < total_tiles) stage_tile(tile + gridDim.x, buf ^ 1);
      const int b = tile / tiles_per_image, trem = tile % tiles_per_image;
      const int y0 = (trem / tiles_x) * G_TH, x0 = (trem % tiles_x) * G_TW;
      if (y0 + wave * 8 < H) {            // waves wholly below the image skip the work, not the barrier
    for (int c = 0; c < 3; ++c) {
#pragma unroll 1
        for (int rb = 0; rb < 2; ++rb) {       // two 4-row blocks per wave
            if (y0 + wave * 8 + rb * 4 >= H) break;   // block wholly below the image (H = 321: 1 row in the last tile row)
            const int trow = wave * 8 + rb * 4 + lyy;
            unsigned outp[MT][4][4];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                // window: 8 tap rows (this half-wave's parity) x 20 bytes starting at 8*li + 4*qq
                int win[8][5];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int *rp = reinterpret_cast<const int *>(&s_tile[buf][c][trow + 2 * kk + h][8 * li + 4 * qq]);
#pragma unroll
                    for (int j = 0; j < 5; ++j) win[kk][j] = rp[j];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v16i acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[mt][e] = 0;   // inline-constant C of the first MFMA
                    // the wave inside its MFMA chain outranks the one in its (pure VALU) epilogue: -2.5 % (A/B)
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        // B fragment of pixel shift s = 4qq + t: bytes [t, t+16) of the 20-byte window
                        v4i bf;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            bf[j] = (t == 0) ? win[kk][j]       // v_alignbit_b32: same result as v_alignbyte, 2.4x the rate
                                             : (int)__builtin_amdgcn_alignbit((unsigned)win[kk][j + 1],
                                                                              (unsigned)win[kk][j], 8 * t);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[mt][kk], bf, acc[mt], 0, 0, 0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    // epilogue: rows 4g..4g+3 of this lane = {re_lo, re_hi, im_lo, im_hi} of one filter.
                    // All 4*MT magnitudes are computed as independent chains (ILP), then pinned.
                    unsigned mag[MT][4];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // v = 256*hi + lo (+ bias): v_mad_i32_i24 (|hi| < 2^22), not a shift (slow here)
                            const int a_re = (__mul24(acc[mt][4 * g + 1], 256) + acc[mt][4 * g + 0] + bias_v[mt][g]) >> shift;
                            const int a_im = (__mul24(acc[mt][4 * g + 3], 256) + acc[mt][4 * g + 2]) >> shift;
                            const unsigned n = (unsigned)__mul24(a_re, a_re) + (unsigned)__mul24(a_im, a_im);
                            mag[mt][g] = isqrt31(n);
                        }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            unsigned &o = outp[mt][g][2 * qq + (t >> 1)];
                            if ((t & 1) == 0)
                                o = mag[mt][g];
                            else
                                o = __umul24(mag[mt][g], 65536u) + o;      // pack the odd pixel into the high half
                            // materialise now: otherwise hipcc sinks the whole epilogue into the
                            // store branches and keeps every accumulator live until then
                            asm volatile("" : "+v"(o));
                        }
                    // keep hipcc from building all four shifts' fragments up front
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // 8 consecutive pixels x = x0 + 8*li .. +7 of row oy: one 16-byte store per filter;
            // the 8 li-lanes of a row cover 128 contiguous bytes
            const int oy = y0 + trow, ox = x0 + 8 * li;
            if (oy < H && ox < pitch) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int f = 8 * (mt0 + mt) + 2 * g + h;
                        if (FULLF || f < F) {
                            // the slab holds offset-binary features (x ^ 0x8080): both bytes are then
                            // signed MFMA digits for the k-means pass, which stages them untouched
                            uint16_t *dst = feats + slab_index((size_t)b * (pstride / KP_TP), D, c * F + f, oy * pitch + ox);
                            *reinterpret_cast<uint4 *>(dst) =
                                make_uint4(outp[mt][g][0] ^ 0x80808080u, outp[mt][g][1] ^ 0x80808080u,
                                           outp[mt][g][2] ^ 0x80808080u, outp[mt][g][3] ^ 0x80808080u);
                        }
                    }
            }
        }
    }
      }
      __syncthreads();   // next tile landed (vmcnt drained) and this buffer is free to refill
    }
}

static inline int gabor_hp(int H) { return (H + G_TH - 1) / G_TH * G_TH + 15; }
static inline int gabor_wp(int W) { return (W + G_TW - 1) / G_TW * G_TW + 32; }

extern "C" size_t gcs_gabor_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * 3 * gabor_hp(H) * gabor_wp(W);
}

extern "C" int gcs_gabor_features(const uint8_t *img, int B, int H, int W, const int8_t *packed,
                                  const int32_t *bias, int F, int shift, void *workspace, uint16_t *feats,
                                  gcs_stream_t stream) {
    if (!img || !packed || !bias || !feats || !workspace)
        return fail(GCS_EINVAL, "gcs_gabor_features: NULL pointer");
    if (B <= 0 || F <= 0) return fail(GCS_EINVAL, "gcs_gabor_features: B and n_filters must be > 0");
    if (H < 8 || W < 8) return fail(GCS_EINVAL, "gcs_gabor_features: H and W must be >= 8");
    if (shift < 0 || shift > 23) return fail(GCS_EINVAL, "gcs_gabor_features: shift out of range");
    if (B > 65535) return fail(GCS_EINVAL, "gcs_gabor_features: B too large for one launch");
    const int pitch = (int)gcs_feature_pitch(W);
    const size_t pstride = gcs_feature_plane_stride(H, W);
    const int tiles_x = (W + G_TW - 1) / G_TW, tiles_y = (H + G_TH - 1) / G_TH;
    const int Hp = gabor_hp(H), Wp = gabor_wp(W);
    if (Hp > 65535) return fail(GCS_EINVAL, "gcs_gabor_features: H too large for one launch");
    int8_t *planes = static_cast<int8_t *>(workspace);
    hipLaunchKernelGGL(gabor_pad_kernel, dim3((Wp / 4 + 255) / 256, Hp, B), dim3(256), 0, stream, img, H, W, Hp, Wp,
                       planes);
    GCS_CHECK_LAUNCH("gcs_gabor_features(pad)");
    const int tiles_per_image = tiles_x * tiles_y;
    const long long total_ll = (long long)tiles_per_image * B;
    if (total_ll > 0x7fffffffLL) return fail(GCS_EINVAL, "gcs_gabor_features: too many tiles");
    const int total_tiles = (int)total_ll;
    const dim3 block(256);
    const int MT = mtiles(F);
    for (int mt0 = 0; mt0 < MT; mt0 += GCS_GABOR_MTMAX) {
        const int n = MT - mt0 >= GCS_GABOR_MTMAX ? GCS_GABOR_MTMAX : MT - mt0;
        // persistent grid: one workgroup per resident slot (2 per CU at MT >= 2, 3 at MT == 1)
        const int slots = 256 * (n == 1 ? 3 : 2);
        const dim3 grid(total_tiles < slots ? total_tiles : slots);
#define GCS_GABOR_LAUNCH(MT_, FF_)                                                                              \
    hipLaunchKernelGGL((gabor_mfma_kernel<MT_, FF_>), grid, block, 0, stream, planes, H, Hp, Wp, packed, bias, mt0, \
                       F, shift, feats, pitch, pstride, tiles_x, tiles_per_image, total_tiles)
        const bool fullf = (F % 8) == 0;
        if (n == 2) { if (fullf) GCS_GABOR_LAUNCH(2, true); else GCS_GABOR_LAUNCH(2, false); }
        else { if (fullf) GCS_GABOR_LAUNCH(1, true); else GCS_GABOR_LAUNCH(1, false); }
#undef GCS_GABOR_LAUNCH
        GCS_CHECK_LAUNCH("gcs_gabor_features");
    }
    return GCS_OK;
}

// ------------------------------------------------------------------------------- unpack
__global__ void unpack_kernel(const uint16_t *__restrict__ feats, int H, int W, int pitch, size_t pstride,
                              size_t planes, size_t planes_per_image, uint16_t *__restrict__ out) {
    const size_t n = planes * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pl = i / ((size_t)H * W);
        const int rem = (int)(i % ((size_t)H * W));
        const int D_ = (int)(planes_per_image);
        out[i] = feats[slab_index((pl / D_) * (pstride / KP_TP), D_, (int)(pl % D_), (rem / W) * pitch + rem % W)] ^ 0x8080u;
    }
}

extern "C" int gcs_features_unpack(const uint16_t *feats, int B, int H, int W, int D, uint16_t *out,
                                   gcs_stream_t stream) {
    if (!feats || !out) return fail(GCS_EINVAL, "gcs_features_unpack: NULL pointer");
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0) return fail(GCS_EINVAL, "gcs_features_unpack: bad shape");
    hipLaunchKernelGGL(unpack_kernel, dim3(2048), dim3(256), 0, stream, feats, H, W,
                       (int)gcs_feature_pitch(W), gcs_feature_plane_stride(H, W), (size_t)B * D, (size_t)D, out);
    GCS_CHECK_LAUNCH("gcs_features_unpack");
    return GCS_OK;
}

// =============================================================================== k-means
__global__ void kmeans_init_kernel(const uint16_t *__restrict__ feats, int H, int W, int pitch, size_t pstride,
                                   int D, int k, uint16_t *__restrict__ cent) {
    const int set = blockIdx.x; // image index == set index (n_sets == 1 -> image 0)
    const long P = (long)H * W;
    for (int i = threadIdx.x; i < k * D; i += blockDim.x) {
        const int j = i / D, d = i % D;
        const long p = ((2L * j + 1) * P) / (2L * k);
        const int y = (int)(p / W), x = (int)(p % W);
        cent[((size_t)set * k + j) * D + d] = feats[slab_index((size_t)set * (pstride / KP_TP), D, d, y * pitch + x)] ^ 0x8080u;
    }
}

extern "C" int gcs_kmeans_init(const uint16_t *feats, int B, int H, int W, int D, int k, int n_sets,
                               uint16_t *cent, gcs_stream_t stream) {
    if (!feats || !cent) return fail(GCS_EINVAL, "gcs_kmeans_init: NULL pointer");
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0) return fail(GCS_EINVAL, "gcs_kmeans_init: bad shape");
    if (k < 1 || k > GCS_K_MAX) return fail(GCS_EINVAL, "gcs_kmeans_init: k must be in 1..16");
    if (n_sets != 1 && n_sets != B) return fail(GCS_EINVAL, "gcs_kmeans_init: n_sets must be 1 or B");
    hipLaunchKernelGGL(kmeans_init_kernel, dim3(n_sets), dim3(256), 0, stream, feats, H, W,
                       (int)gcs_feature_pitch(W), gcs_feature_plane_stride(H, W), D, k, cent);
    GCS_CHECK_LAUNCH("gcs_kmeans_init");
    return GCS_OK;
}

// out[i][d] = feature d of pixel (b, y, x) = byx[i]; b < 0 gives a zero row. Lets a rank publish the
// SPEC.md §4 init centroids it owns when an image is sharded by rows (BASELINE config 5).
__global__ void features_gather_kernel(const uint16_t *__restrict__ feats, int pitch, size_t pstride, int D, int n,
                                       const int32_t *__restrict__ byx, uint16_t *__restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * D; i += gridDim.x * blockDim.x) {
        const int r = i / D, d = i % D;
        const int b = byx[3 * r], y = byx[3 * r + 1], x = byx[3 * r + 2];
        out[i] = b < 0 ? (uint16_t)0
                       : (uint16_t)(feats[slab_index((size_t)b * (pstride / KP_TP), D, d, y * pitch + x)] ^ 0x8080u);
    }
}

extern "C" int gcs_features_gather(const uint16_t *feats, int B, int H, int W, int D, int n, const int32_t *byx,
                                   uint16_t *out, gcs_stream_t stream) {
    if (!feats || !byx || !out) return fail(GCS_EINVAL, "gcs_features_gather: NULL pointer");
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0 || n <= 0) return fail(GCS_EINVAL, "gcs_features_gather: bad shape");
    hipLaunchKernelGGL(features_gather_kernel, dim3((n * D + 255) / 256), dim3(256), 0, stream, feats,
                       (int)gcs_feature_pitch(W), gcs_feature_plane_stride(H, W), D, n, byx, out);
    GCS_CHECK_LAUNCH("gcs_features_gather");
    return GCS_OK;
}

// Exact integer argmin with fp32 digit arithmetic: x = 256*xh + xl, c = 256*ch + cl (bytes);
//   sum_d x*c = 65536*sum xh*ch + 256*sum (xh*cl + xl*ch) + sum xl*cl,
// every partial sum stays below 2^24 over a chunk of <= 128 planes, so fp32 FMA is exact.
// score_j = |c_j|^2 - 2 sum_d x_d c_jd (the |x|^2 term is common to all j).
constexpr int KM_CHUNK = 128;

template <int K>
__global__ __launch_bounds__(256) void kmeans_assign_kernel(
    const uint16_t *__restrict__ feats, const uint16_t *__restrict__ cent, int H, int W, int pitch, size_t plane,
    int D, int per_image, int parts, int R, int row_lo, int row_hi, uint8_t *__restrict__ labels,
    uint64_t *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve: cdig float [D][K][2] | cnorm int64 [K] | acc u32 [K][D+1][R]
    float *cdig = reinterpret_cast<float *>(smem);
    long long *cnorm = reinterpret_cast<long long *>(smem + (((size_t)D * K * 2 * 4 + 15) & ~(size_t)15));
    unsigned *acc = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(cnorm) + ((K * 8 + 15) & ~15));

    const int tid = threadIdx.x;
    const int b = blockIdx.y, part = blockIdx.x;
    const uint16_t *cset = cent + (size_t)(per_image ? b : 0) * K * D;
    const int D1 = D + 1;

    for (int i = tid; i < K * D; i += 256) {
        const int j = i / D, d = i % D;
        const unsigned cv = cset[i];
        cdig[(d * K + j) * 2 + 0] = (float)(cv & 255u);
        cdig[(d * K + j) * 2 + 1] = (float)(cv >> 8);
    }
    if (tid < K) {
        long long s = 0;
        for (int d = 0; d < D; ++d) {
            const long long cv = cset[tid * D + d];
            s += cv * cv;
        }
        cnorm[tid] = s;
    }
    for (int i = tid; i < K * D1 * R; i += 256) acc[i] = 0u;
    __syncthreads();

    const int ppr = pitch >> 1; // pixel pairs per row
    const long npairs = (long)H * ppr;
    const size_t tile0 = (size_t)b * (plane / KP_TP);          // slab holds x ^ 0x8080, tile-major
    const int rep = tid & (R - 1);

    for (long q = (long)part * 256 + tid; q < npairs; q += (long)parts * 256) {
        const int y = (int)(q / ppr), x = 2 * (int)(q % ppr);
        const int off = y * pitch + x;
        long long S[2][K];
#pragma unroll
        for (int j = 0; j < K; ++j) S[0][j] = S[1][j] = 0;
        for (int d0 = 0; d0 < D; d0 += KM_CHUNK) {
            const int d1 = min(D, d0 + KM_CHUNK);
            float a0[2][K], a1[2][K], a2[2][K];
#pragma unroll
            for (int j = 0; j < K; ++j) a0[0][j] = a0[1][j] = a1[0][j] = a1[1][j] = a2[0][j] = a2[1][j] = 0.f;
            for (int d = d0; d < d1; ++d) {
                const unsigned u = *reinterpret_cast<const unsigned *>(feats + slab_index(tile0, D, d, off)) ^ 0x80808080u;
                const float xl0 = (float)(u & 255u), xh0 = (float)((u >> 8) & 255u);
                const float xl1 = (float)((u >> 16) & 255u), xh1 = (float)(u >> 24);
                const float *cd = cdig + d * K * 2;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float cl = cd[2 * j], ch = cd[2 * j + 1];
                    a0[0][j] = fmaf(xl0, cl, a0[0][j]);
                    a1[0][j] = fmaf(xh0, cl, fmaf(xl0, ch, a1[0][j]));
                    a2[0][j] = fmaf(xh0, ch, a2[0][j]);
                    a0[1][j] = fmaf(xl1, cl, a0[1][j]);
                    a1[1][j] = fmaf(xh1, cl, fmaf(xl1, ch, a1[1][j]));
                    a2[1][j] = fmaf(xh1, ch, a2[1][j]);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < K; ++j)
                    S[p][j] += ((long long)(unsigned)a2[p][j] << 16) + ((long long)(unsigned)a1[p][j] << 8) +
                               (long long)(unsigned)a0[p][j];
        }
        int lab[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            long long best = cnorm[0] - 2 * S[p][0];
            int bj = 0;
#pragma unroll
            for (int j = 1; j < K; ++j) {
                const long long sc = cnorm[j] - 2 * S[p][j];
                if (sc < best) {
                    best = sc;
                    bj = j;
                }
            }
            lab[p] = bj;
        }
        *reinterpret_cast<uint16_t *>(labels + (size_t)b * plane + (size_t)y * pitch + x) =
            (uint16_t)(lab[0] | (lab[1] << 8));
        // accumulate (second pass over this thread's planes; L2-resident)
        const bool rows_ok = y >= row_lo && y < row_hi;
        const bool v0 = rows_ok && x < W, v1 = rows_ok && x + 1 < W;
        if (v0) {
            unsigned *a_0 = acc + (size_t)lab[0] * D1 * R + rep;
            unsigned *a_1 = acc + (size_t)lab[1] * D1 * R + rep;
            if (v1 && lab[0] == lab[1]) {
                for (int d = 0; d < D; ++d) {
                    const unsigned u = *reinterpret_cast<const unsigned *>(feats + slab_index(tile0, D, d, off)) ^ 0x80808080u;
                    atomicAdd(a_0 + d * R, (u & 0xffffu) + (u >> 16));
                }
                atomicAdd(a_0 + D * R, 2u);
            } else {
                for (int d = 0; d < D; ++d) {
                    const unsigned u = *reinterpret_cast<const unsigned *>(feats + slab_index(tile0, D, d, off)) ^ 0x80808080u;
                    atomicAdd(a_0 + d * R, u & 0xffffu);
                    if (v1) atomicAdd(a_1 + d * R, u >> 16);
                }
                atomicAdd(a_0 + D * R, 1u);
                if (v1) atomicAdd(a_1 + D * R, 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < K * D1; i += 256) {
        uint64_t s = 0;
        for (int rr = 0; rr < R; ++rr) s += acc[(size_t)i * R + rr];
        partials[partial_index(per_image, b, part, parts, i, K * D1)] = s;
    }
}

// ---------------------------------------------------------------------------------------
// One Lloyd pass on the matrix cores (D <= 207: 80-row LDS tile for D <= 79, 208-row tile above).
// Per 256-pixel tile, staged ONCE in LDS as
// u16 planes (each byte offset by -128 so it is a signed MFMA digit):
//   assign:  scores[(j,pat)][px] = A_pat[(j,pat)][k] * X[k][px] on v_mfma_i32_32x32x32_i8, k =
//            (plane, byte). Patterns per cluster j: LL = cl*xl, M = ch*xl + cl*xh, HH = ch*xh, so
//            sum_d x_d c_jd = LL + 256 M + 65536 HH exactly (int32 partials, int64 combine);
//            argmin_j |c_j|^2 - 2 sum_d x_d c_jd, ties -> lowest j (SPEC.md §4).
//   update:  sums[j][byte-plane] = onehot[j][px] * X[px][byte-plane] on v_mfma_i32_16x16x64_i8;
//            one spare byte-plane is all ones and yields the counts. Accumulators live in
//            registers for the whole workgroup; nothing but the tile load touches HBM.
// The one-hot digit is 0x80 (= -128) to save a shift; it is divided out exactly at the end.
// D = a * b + c with a 64-bit accumulator in ONE instruction. hipcc strength-reduces the C expression
// into sign extensions, 64-bit shifts and borrow chains (~10 instructions); the count is what costs here.
__device__ __forceinline__ long long mad_i64_i32(int a, int b, long long c) {
    long long d;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c) : "vcc");
    return d;
}

constexpr int KP_PITCH = KP_TP * 2 + 64;  // bytes per plane row: +64 B = 16 banks per row, so the 4 rows x 64 B of a
                                          // tr_b16 half-wave and the 8 rows of a ds_read_b128 lane group hit distinct banks
constexpr int KP_DSTEPS_NARROW = 5;       // D <= 79  (every 4x6 bank): 80 plane rows, 46 KB LDS, 3 workgroups / CU
constexpr int KP_DSTEPS_WIDE = 13;        // D <= 207 (the 8x8 bank, D = 192): 208 plane rows, 120 KB LDS, 1 workgroup / CU

#ifndef GCS_KP_WAVES
#define GCS_KP_WAVES 3
#endif
// DSTEPS = assign K-steps (16 planes = 32 byte-features each); LDS holds ROWS = 16*DSTEPS plane rows (>= D + 1:
// the spare row D is the count row); the update has NT = 2*DSTEPS N-tiles (8 planes = 16 byte-planes each).
// NST = staging chunks per thread >= ceil(D/8); EXACT: D == 8*NST.
template <int KT, int NST, bool EXACT, int DSTEPS>
__global__ __launch_bounds__(KP_TP, (DSTEPS == KP_DSTEPS_NARROW && KT == 1 && (NST <= 6 || (EXACT && NST <= 9)) ? GCS_KP_WAVES
                                     : DSTEPS == KP_DSTEPS_NARROW ? 2 : 1)) void kmeans_pass_mfma_kernel(
    const uint16_t *__restrict__ feats, const uint16_t *__restrict__ cent, int H, int W, int pitch, int pstride,
    int D, int K, int per_image, int parts, int reverse, int row_lo, int row_hi,
    uint8_t *__restrict__ labels, uint64_t *__restrict__ partials) {
    constexpr int KP_ROWS = 16 * DSTEPS, KP_DSTEPS = DSTEPS, KP_NT = 2 * DSTEPS;
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[KP_ROWS * KP_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char s_lab[KP_TP];
    __shared__ long long s_const[16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, part = blockIdx.x;
    const uint16_t *cset = cent + (size_t)(per_image ? b : 0) * K * D;
    const int p_lo = row_lo * pitch, p_hi = row_hi * pitch;   // pixels that vote: rows [row_lo, row_hi)
    const int ntiles = pstride / KP_TP;              // plane stride is a whole number of tiles
    const uint16_t *fb = feats + (size_t)b * ntiles * D * KP_TP;   // this image's tiles, each D*256 contiguous

    // ---- centroids -> LDS scratch (borrowed from the tile buffer): [8*KT clusters][KP_ROWS planes] u16, stored
    //      offset-binary (c ^ 0x8080: low byte = digit cl, high byte = digit ch), zero outside K x D. One coalesced
    //      read of the K*D centroid block instead of 16 scattered 2-byte global loads per fragment dword.
    uint16_t *cs = reinterpret_cast<uint16_t *>(s_tile);
    for (int i = tid; i < 8 * KT * KP_ROWS; i += KP_TP) {
        const int j = i / KP_ROWS, d = i % KP_ROWS;
        cs[i] = (j < K && d < D) ? (uint16_t)(cset[j * D + d] ^ 0x8080u) : (uint16_t)0;
    }
    __syncthreads();
    // ---- per-cluster key base (exact int64): 16 * (|c|^2 - 2*(offset terms of the -128 digits)) + j.
    //      key_j = base_j - 32 R0 - 8192 R1 - 2^21 R2 = 16 * score_j + j, so ONE 64-bit minimum yields the
    //      best score and the lowest index on ties. 16 lanes per cluster, folded with lane shuffles.
    {
      for (int j = tid >> 4; j < 16; j += KP_TP / 16) {
        const int sub = tid & 15;
        long long nrm = 0, scl = 0, sch = 0;
        if (j < K)
            for (int d = sub; d < D; d += 16) {
                const long long c = cs[j * KP_ROWS + d] ^ 0x8080u;
                nrm += c * c;
                scl += c & 255;
                sch += c >> 8;
            }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            nrm += __shfl_xor(nrm, m);
            scl += __shfl_xor(scl, m);
            sch += __shfl_xor(sch, m);
        }
        if (sub == 0) {
            const long long q = 16384LL * D;
            const long long g = (128 * scl - q) + 256 * (128 * (sch + scl) - 2 * q) + 65536 * (128 * sch - q);
            s_const[j] = j < K ? 16 * (nrm - 2 * g) + j : (1LL << 62) + j;
        }
      }
    }
    // ---- assign A fragments: row r = 4*jj + pat of tile mt (cluster j = 8*mt + jj);
    //      k-slot (h, t) of K-step kk = (plane 16*kk + 8*h + t/2, byte t&1): the 8 planes of a fragment are one
    //      16-byte scratch read. Per plane (u16 w = digits cl | ch << 8) the pattern bytes (byte 0, byte 1) are
    //      LL = (cl, 0) = w & 0x00ff, M = (ch, cl) = bytes swapped, HH = (0, ch) = w & 0xff00, row 3 = 0.
    v4i apat[KT][KP_DSTEPS];
    {
        const int r = lane & 31, h = lane >> 5;
        const int jj = r >> 2, pat = r & 3;
        const unsigned msk = pat == 0 ? 0x00ff00ffu : pat == 1 ? 0xffffffffu : pat == 2 ? 0xff00ff00u : 0u;
        const unsigned sel = pat == 1 ? 0x02030001u : 0x03020100u;
#pragma unroll
        for (int mt = 0; mt < KT; ++mt)
#pragma unroll
            for (int kk = 0; kk < KP_DSTEPS; ++kk) {
                const v4i w = *reinterpret_cast<const v4i *>(&cs[(8 * mt + jj) * KP_ROWS + 16 * kk + 8 * h]);
                v4i f;
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = (int)(__builtin_amdgcn_perm(0u, (unsigned)w[e], sel) & msk);
                apat[mt][kk] = f;
            }
    }
    __syncthreads();                                   // scratch reads done: the tile buffer is free again
    // the count row (plane D): byte-planes 2D, 2D+1 read as +1 for every pixel of every tile
    if (tid < KP_TP / 2) reinterpret_cast<unsigned *>(&s_tile[D * KP_PITCH])[tid] = 0x01010101u;
    v4i accu[KP_NT];
#pragma unroll
    for (int nt = 0; nt < KP_NT; ++nt) accu[nt] = v4i{0, 0, 0, 0};

    // ---- staging: the tile is ONE contiguous D*512-byte run of the slab (tile-major layout, already
    //      offset-binary): chunk ci = tid + 256*i is 16 bytes at byte 16*ci -> plane ci>>5, pixels 8*(ci&31)..
    //      Loads and LDS writes are UNCONDITIONAL: a per-chunk guard makes hipcc branch around every
    //      load / write with exec masking and drain vmcnt(0) before each write. Chunk rows beyond the
    //      last plane (D not a multiple of 8, or a coarser NST bucket) are clamped to plane D-1: they
    //      re-read and re-write row D-1 with its own data.
    const int sd0 = tid / (KP_TP / 8), spo = 8 * (tid % (KP_TP / 8));
    v4i st[NST];
    int srow[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) srow[i] = EXACT ? sd0 + 8 * i : min(sd0 + 8 * i, D - 1);
    auto stage_load = [&](int tile) {
        const v4i *src = reinterpret_cast<const v4i *>(fb + (size_t)tile * D * KP_TP) + (tid % (KP_TP / 8));
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = src[srow[i] * (KP_TP / 8)];
    };
    auto stage_write = [&]() {
        unsigned char *dst = &s_tile[spo * 2];
#pragma unroll
        for (int i = 0; i < NST; ++i) *reinterpret_cast<v4i *>(dst + srow[i] * KP_PITCH) = st[i];
    };

    const int un = lane & 15, ug = lane >> 4;             // update operand coordinates
    const unsigned usel = (un & 1) ? 0x07050301u : 0x06040200u;
    const unsigned eqr = (unsigned)un * 0x01010101u;
    const int cnt_bp = 2 * D;

    // Sweep order: workgroup `part` takes logical tiles part, part+parts, ...; on odd passes the physical
    // order is reversed (boustrophedon), so a pass starts on the tiles the previous pass read last, i.e. on
    // what is still in the 256 MiB Infinity Cache.
    auto phys = [&](int lt) { return reverse ? ntiles - 1 - lt : lt; };
    int ltile = part;
    if (ltile < ntiles) stage_load(phys(ltile));
    for (; ltile < ntiles; ltile += parts) {
        const int tile = phys(ltile);
        stage_write();
        __syncthreads();
        // the next tile's loads go out first: a wave issuing them outranks the waves of the other workgroups that
        // are in their compute phase (18 interleaved A/B runs: 0.252 -> 0.244 ms per pass)
        __builtin_amdgcn_s_setprio(3);
        if (ltile + parts < ntiles) stage_load(phys(ltile + parts));   // in flight during the MFMAs
        __builtin_amdgcn_s_setprio(0);

        const int pp0 = tile * KP_TP;
        const int x0 = pp0 % pitch;               // column of the tile's first pixel (one modulo per tile)
        // -------- assign: two 32-pixel sub-tiles per wave
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int n = lane & 31, h = lane >> 5;
            const int pl = wave * 64 + sub * 32 + n;
            v16i acc[KT];
#pragma unroll
            for (int mt = 0; mt < KT; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][e] = 0;
            // B fragments by hardware transpose: per 16-lane group ds_read_b64_tr_b16 reads a block of
            // 4 rows (planes) x 16 columns (pixels) of 16-bit elements and gives lane i column i, i.e.
            // the four planes of ITS pixel (cdna guide T10). Lane 4q+p of the group supplies the address
            // of row q, columns 4p..4p+3. Two reads = 8 planes = the 16-byte fragment of one K-step.
            // (Replaces 8 ds_read_u16 + 4 pack ops per K-step.) One asm statement: loads + their wait.
            v4i bfr[KP_DSTEPS];
            {
                const int i16 = lane & 15, pxblk = (lane >> 4) & 1;
                const unsigned addr = (unsigned)(size_t)&s_tile[(8 * h + (i16 >> 2)) * KP_PITCH +
                                                                (wave * 64 + sub * 32 + 16 * pxblk + 4 * (i16 & 3)) * 2];
                typedef int v2i __attribute__((ext_vector_type(2)));
                v2i fa[KP_DSTEPS], fb[KP_DSTEPS];
#pragma unroll
                for (int kk = 0; kk < KP_DSTEPS; ++kk)       // the DS offset field holds 16 bits: K-step base in the VGPR
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\t"
                                 "ds_read_b64_tr_b16 %1, %2 offset:%c3"
                                 : "=&v"(fa[kk]), "=&v"(fb[kk])
                                 : "v"(addr + kk * 16 * KP_PITCH), "i"(4 * KP_PITCH)
                                 : "memory");
                // hipcc does not count asm loads: one explicit wait, then tie every destination register to it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < KP_DSTEPS; ++kk) {
                    asm volatile("" : "+v"(fa[kk]), "+v"(fb[kk]));
                    bfr[kk] = v4i{fa[kk][0], fa[kk][1], fb[kk][0], fb[kk][1]};
                }
            }
#pragma unroll
            for (int kk = 0; kk < KP_DSTEPS; ++kk)
#pragma unroll
                for (int mt = 0; mt < KT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(apat[mt][kk], bfr[kk], acc[mt], 0, 0, 0);
            // key = 16*score + j via v_mad_i64_i32 (3 instructions per cluster instead of ~25 of sign
            // extension / 64-bit shift / borrow arithmetic): U = R0 + 256 R1 fits int32 (|U| < 2^30).
            long long best = 0x7fffffffffffffffLL;
#pragma unroll
            for (int mt = 0; mt < KT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int u = __mul24(acc[mt][4 * g + 1], 256) + acc[mt][4 * g];
                    long long key = mad_i64_i32(u, -32, s_const[8 * mt + 2 * g + h]);   // base: LDS broadcast read
                    key = mad_i64_i32(acc[mt][4 * g + 2], -2097152, key);
                    best = key < best ? key : best;
                }
            // partner half's key by v_permlane32_swap (VALU; no LDS round trip like ds_bpermute)
            const unsigned blo = (unsigned)best, bhi = (unsigned)((unsigned long long)best >> 32);
            const auto s0 = __builtin_amdgcn_permlane32_swap(blo, blo, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(bhi, bhi, false, false);
            // after swap(x, x): element 1 holds the upper half's x in lanes 0-31, element 0 the lower half's x in lanes 32-63
            const unsigned plo = h ? s0[0] : s0[1], phi = h ? s1[0] : s1[1];
            const long long pb = (long long)(((unsigned long long)phi << 32) | plo);
            const int bj = (int)((pb < best ? pb : best) & 15);
            if (h == 0) {
                const int pp = pp0 + pl;
                int x = x0 + pl;                     // pixel column: at most ceil(256/pitch)+1 wraps
                if (pitch >= KP_TP) {
                    if (x >= pitch) x -= pitch;
                } else {
                    x %= pitch;
                }
                const bool valid = pp >= p_lo && pp < p_hi && x < W;
                s_lab[pl] = valid ? (unsigned char)bj : (unsigned char)0xFF;
                labels[(size_t)b * pstride + pp] = (uint8_t)bj;
            }
        }
        // -------- update: one-hot MFMA over this wave's 64 pixels
        {
            const v4i lw = *reinterpret_cast<const v4i *>(&s_lab[wave * 64 + 16 * ug]);
            v4i oh;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned x = (unsigned)lw[i] ^ eqr;                    // byte == 0 <=> label == un
                const unsigned y = (x | 0x80808080u) - 0x01010101u;        // top bit clear <=> byte == 0
                oh[i] = (int)(~y & 0x80808080u);                            // digit -128 where label == un
            }
#pragma unroll
            for (int nt = 0; nt < KP_NT; ++nt) {
                const int d = 8 * nt + (un >> 1);
                const v4i *src = reinterpret_cast<const v4i *>(&s_tile[d * KP_PITCH + (wave * 64 + 16 * ug) * 2]);
                const v4i w0 = src[0], w1 = src[1];
                v4i bx;
                bx[0] = (int)__builtin_amdgcn_perm((unsigned)w0[1], (unsigned)w0[0], usel);
                bx[1] = (int)__builtin_amdgcn_perm((unsigned)w0[3], (unsigned)w0[2], usel);
                bx[2] = (int)__builtin_amdgcn_perm((unsigned)w1[1], (unsigned)w1[0], usel);
                bx[3] = (int)__builtin_amdgcn_perm((unsigned)w1[3], (unsigned)w1[2], usel);
                accu[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(oh, bx, accu[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- fold the four waves' accumulators (rows = clusters, cols = byte-planes) and emit the row: every wave
    //      parks its registers in its own slice of the tile buffer (no zero-fill, no atomics), one barrier.
    constexpr int RW = KP_NT * 16;                            // byte-planes per cluster row
    int *red = reinterpret_cast<int *>(s_tile);               // [KP_TP / 64 waves][16][RW]
    static_assert((KP_TP / 64) * 16 * RW * 4 <= KP_ROWS * KP_PITCH, "fold buffer exceeds the tile buffer");
#pragma unroll
    for (int nt = 0; nt < KP_NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * 16 + 4 * ug + e) * RW + 16 * nt + un] = accu[nt][e];
    __syncthreads();
    const int D1 = D + 1;
    auto folded = [&](int j, int bp) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < KP_TP / 64; ++w) s += red[(w * 16 + j) * RW + bp];
        return -(long long)s / 128;                           // the one-hot digit is -128
    };
    for (int i = tid; i < K * D1; i += KP_TP) {
        const int j = i / D1, e = i % D1;
        const long long nj = folded(j, cnt_bp);
        long long out = nj;
        if (e < D) out = (folded(j, 2 * e) + 128 * nj) + 256 * (folded(j, 2 * e + 1) + 128 * nj);
        partials[partial_index(per_image, b, part, parts, i, K * D1)] = (uint64_t)out;
    }
}

static size_t assign_lds_bytes(int D, int k, int R) {
    size_t a = ((size_t)D * k * 2 * 4 + 15) & ~(size_t)15;
    size_t c = ((size_t)k * 8 + 15) & ~(size_t)15;
    return a + c + (size_t)k * (D + 1) * R * 4;
}

template <int K>
static int launch_assign(const uint16_t *feats, const uint16_t *cent, int B, int H, int W, int D, int n_sets,
                         int row_lo, int row_hi, uint8_t *labels, uint64_t *partials, hipStream_t stream) {
    const int parts = (int)gcs_kmeans_parts_per_image(B, H, W);
    int R = 32;
    while (R > 1 && assign_lds_bytes(D, K, R) > 120 * 1024) R >>= 1;
    const size_t lds = assign_lds_bytes(D, K, R);
    if (lds > 160 * 1024) return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: k*D too large for LDS");
    static size_t lds_granted = 0; // raise the dynamic-LDS cap once per instantiation
    if (lds > lds_granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kmeans_assign_kernel<K>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(assign)");
        lds_granted = lds;
    }
    hipLaunchKernelGGL(kmeans_assign_kernel<K>, dim3(parts, B), dim3(256), lds, stream, feats, cent, H, W,
                       (int)gcs_feature_pitch(W), gcs_feature_plane_stride(H, W), D, n_sets == B ? 1 : 0, parts, R,
                       row_lo, row_hi, labels, partials);
    GCS_CHECK_LAUNCH("gcs_kmeans_assign_accumulate");
    return GCS_OK;
}

extern "C" int gcs_kmeans_assign_accumulate(const uint16_t *feats, const uint16_t *cent, int B, int H, int W,
                                            int D, int k, int n_sets, int row_lo, int row_hi, int reverse,
                                            uint8_t *labels, uint64_t *partials, gcs_stream_t stream) {
    if (!feats || !cent || !labels || !partials)
        return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: NULL pointer");
    if (row_lo < 0 || row_hi > H || row_lo >= row_hi)
        return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: need 0 <= row_lo < row_hi <= H");
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0 || B > 65535)
        return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: bad shape");
    if (k < 1 || k > GCS_K_MAX) return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: k must be in 1..16");
    if (n_sets != 1 && n_sets != B)
        return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: n_sets must be 1 or B");
    if (D < 16 * KP_DSTEPS_WIDE) { // matrix-core pass (every BASELINE bank: 4x6 -> D = 72, 8x8 -> D = 192)
        const int parts = (int)gcs_kmeans_parts_per_image(B, H, W);
        const int pitch = (int)gcs_feature_pitch(W);
        if ((long long)H * pitch > 0x7fffffffLL / 2) return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: image too large");
        const int pstride = (int)gcs_feature_plane_stride(H, W);
#define GCS_KP_LAUNCH(KT_, NST_, DS_)                                                                         \
    if (D == 8 * NST_)                                                                                        \
        GCS_KP_LAUNCH2(KT_, NST_, true, DS_);                                                                 \
    else                                                                                                      \
        GCS_KP_LAUNCH2(KT_, NST_, false, DS_)
#define GCS_KP_LAUNCH2(KT_, NST_, EX_, DS_)                                                                   \
    hipLaunchKernelGGL((kmeans_pass_mfma_kernel<KT_, NST_, EX_, DS_>), dim3(parts, B), dim3(KP_TP), 0, stream, feats, cent, \
                       H, W, pitch, pstride, D, k, n_sets == B ? 1 : 0, parts, reverse ? 1 : 0, row_lo, row_hi,  \
                       labels, partials)
        const int nst = (D + 7) / 8;
        if (D < 16 * KP_DSTEPS_NARROW) {
            if (k <= 8) {
                if (nst <= 3) { GCS_KP_LAUNCH(1, 3, KP_DSTEPS_NARROW); }
                else if (nst <= 6) { GCS_KP_LAUNCH(1, 6, KP_DSTEPS_NARROW); }
                else if (nst <= 9) { GCS_KP_LAUNCH(1, 9, KP_DSTEPS_NARROW); }
                else { GCS_KP_LAUNCH(1, 10, KP_DSTEPS_NARROW); }
            } else {
                if (nst <= 9) { GCS_KP_LAUNCH(2, 9, KP_DSTEPS_NARROW); }
                else { GCS_KP_LAUNCH(2, 10, KP_DSTEPS_NARROW); }
            }
        } else if (k <= 8) {
            if (nst <= 16) { GCS_KP_LAUNCH(1, 16, KP_DSTEPS_WIDE); }
            else if (nst <= 24) { GCS_KP_LAUNCH(1, 24, KP_DSTEPS_WIDE); }
            else { GCS_KP_LAUNCH(1, 26, KP_DSTEPS_WIDE); }
        } else {
            if (nst <= 16) { GCS_KP_LAUNCH(2, 16, KP_DSTEPS_WIDE); }
            else if (nst <= 24) { GCS_KP_LAUNCH(2, 24, KP_DSTEPS_WIDE); }
            else { GCS_KP_LAUNCH(2, 26, KP_DSTEPS_WIDE); }
        }
#undef GCS_KP_LAUNCH
#undef GCS_KP_LAUNCH2
        GCS_CHECK_LAUNCH("gcs_kmeans_assign_accumulate");
        return GCS_OK;
    }
    switch (k) { // generic VALU pass for wider feature vectors
#define GCS_CASE(KK) \
    case KK:         \
        return launch_assign<KK>(feats, cent, B, H, W, D, n_sets, row_lo, row_hi, labels, partials, stream);
        GCS_CASE(1) GCS_CASE(2) GCS_CASE(3) GCS_CASE(4) GCS_CASE(5) GCS_CASE(6) GCS_CASE(7) GCS_CASE(8)
        GCS_CASE(9) GCS_CASE(10) GCS_CASE(11) GCS_CASE(12) GCS_CASE(13) GCS_CASE(14) GCS_CASE(15) GCS_CASE(16)
#undef GCS_CASE
    }
    return fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: unreachable");
}

// sums[set][e] = sum of the set's partial values of element e (one contiguous run, see partial_index). Integer
// sums: any order gives the same bits. One wave per element: coalesced reads, shuffle fold.
// FIN: the SPEC.md §4 update is applied in the same launch (single-rank case, no all-reduce in between): the wave
// also folds the count element of its cluster, so no second kernel and no cross-block dependency is needed.
template <bool FIN>
__global__ __launch_bounds__(256) void kmeans_reduce_kernel(const uint64_t *__restrict__ partials,
                                                            int rows_per_set, int row_len, int D1,
                                                            long long *__restrict__ sums,
                                                            uint16_t *__restrict__ cent) {
    const int set = blockIdx.y, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= row_len) return;                                  // whole waves leave; no barrier below
    const int j = e / D1, d = e - j * D1;
    const uint64_t *p = partials + ((size_t)set * row_len + e) * rows_per_set;
    const uint64_t *pc = partials + ((size_t)set * row_len + j * D1 + (D1 - 1)) * rows_per_set;
    uint64_t s = 0, c = 0;
    for (int r = lane; r < rows_per_set; r += 64) {
        s += p[r];
        if (FIN) c += pc[r];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m);
        if (FIN) c += __shfl_xor(c, m);
    }
    if (lane == 0) {
        if (sums) sums[(size_t)set * row_len + e] = (long long)s;
        if (FIN && d < D1 - 1 && c > 0)
            cent[((size_t)set * (row_len / D1) + j) * (D1 - 1) + d] = (uint16_t)((2 * s + c) / (2 * c));
    }
}

static int reduce_args_ok(const void *partials, int B, int H, int W, int D, int k, int n_sets, const char *who) {
    if (!partials) return fail(GCS_EINVAL, "gcs_kmeans_reduce: NULL pointer");
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0 || k < 1 || k > GCS_K_MAX) return fail(GCS_EINVAL, who);
    if (n_sets != 1 && n_sets != B) return fail(GCS_EINVAL, "gcs_kmeans_reduce: n_sets must be 1 or B");
    return GCS_OK;
}

extern "C" int gcs_kmeans_reduce(const uint64_t *partials, int B, int H, int W, int D, int k, int n_sets,
                                 int64_t *sums, gcs_stream_t stream) {
    if (!sums) return fail(GCS_EINVAL, "gcs_kmeans_reduce: NULL pointer");
    if (int rc = reduce_args_ok(partials, B, H, W, D, k, n_sets, "gcs_kmeans_reduce: bad shape")) return rc;
    const int parts = (int)gcs_kmeans_parts_per_image(B, H, W);
    const int row_len = k * (D + 1);
    const int rows_per_set = n_sets == B ? parts : B * parts;
    hipLaunchKernelGGL(kmeans_reduce_kernel<false>, dim3((row_len + 3) / 4, n_sets), dim3(256), 0, stream, partials,
                       rows_per_set, row_len, D + 1, reinterpret_cast<long long *>(sums), (uint16_t *)nullptr);
    GCS_CHECK_LAUNCH("gcs_kmeans_reduce");
    return GCS_OK;
}

extern "C" int gcs_kmeans_reduce_finalize(const uint64_t *partials, int B, int H, int W, int D, int k, int n_sets,
                                          int64_t *sums, uint16_t *cent, gcs_stream_t stream) {
    if (!cent) return fail(GCS_EINVAL, "gcs_kmeans_reduce_finalize: NULL pointer");
    if (int rc = reduce_args_ok(partials, B, H, W, D, k, n_sets, "gcs_kmeans_reduce_finalize: bad shape")) return rc;
    const int parts = (int)gcs_kmeans_parts_per_image(B, H, W);
    const int row_len = k * (D + 1);
    const int rows_per_set = n_sets == B ? parts : B * parts;
    hipLaunchKernelGGL(kmeans_reduce_kernel<true>, dim3((row_len + 3) / 4, n_sets), dim3(256), 0, stream, partials,
                       rows_per_set, row_len, D + 1, reinterpret_cast<long long *>(sums), cent);
    GCS_CHECK_LAUNCH("gcs_kmeans_reduce_finalize");
    return GCS_OK;
}

__global__ void kmeans_finalize_kernel(const long long *__restrict__ sums, int n, int k, int D,
                                       uint16_t *__restrict__ cent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = i % D, j = (i / D) % k, set = i / (D * k);
    const long long *row = sums + ((size_t)set * k + j) * (D + 1);
    const long long cnt = row[D];
    if (cnt > 0) cent[i] = (uint16_t)((2 * row[d] + cnt) / (2 * cnt));
}

extern "C" int gcs_kmeans_finalize(const int64_t *sums, int n_sets, int k, int D, uint16_t *cent,
                                   gcs_stream_t stream) {
    if (!sums || !cent) return fail(GCS_EINVAL, "gcs_kmeans_finalize: NULL pointer");
    if (n_sets <= 0 || D <= 0 || k < 1 || k > GCS_K_MAX) return fail(GCS_EINVAL, "gcs_kmeans_finalize: bad shape");
    const int n = n_sets * k * D;
    hipLaunchKernelGGL(kmeans_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const long long *>(sums), n, k, D, cent);
    GCS_CHECK_LAUNCH("gcs_kmeans_finalize");
    return GCS_OK;
}

// Block = 64 x 4 threads: four image rows per block, a thread widens 4 labels (one aligned dword of the slab:
// pitch % 8 == 0) per step of 64 dwords.
__global__ __launch_bounds__(256) void widen_kernel_rows(const uint8_t *__restrict__ labels, int H, int W, int pitch,
                                                         size_t pstride, int rows, int32_t *__restrict__ out) {
    const int by = blockIdx.x * 4 + threadIdx.y;     // b*H + y
    if (by >= rows) return;
    const int b = by / H, y = by - b * H;
    const unsigned *src = reinterpret_cast<const unsigned *>(labels + (size_t)b * pstride + (size_t)y * pitch);
    int32_t *dst = out + (size_t)by * W;
    for (int x4 = threadIdx.x; 4 * x4 < W; x4 += 64) {
        const unsigned v = src[x4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * x4 + e < W) dst[4 * x4 + e] = (int32_t)((v >> (8 * e)) & 255u);
    }
}

extern "C" int gcs_labels_widen(const uint8_t *labels, int B, int H, int W, int32_t *out, gcs_stream_t stream) {
    if (!labels || !out) return fail(GCS_EINVAL, "gcs_labels_widen: NULL pointer");
    if (B <= 0 || H <= 0 || W <= 0 || (long long)B * H > 0x7fffffffLL)
        return fail(GCS_EINVAL, "gcs_labels_widen: bad shape");
    const int rows = B * H;
    hipLaunchKernelGGL(widen_kernel_rows, dim3((rows + 3) / 4), dim3(64, 4), 0, stream, labels, H, W,
                       (int)gcs_feature_pitch(W), gcs_feature_plane_stride(H, W), rows, out);
    GCS_CHECK_LAUNCH("gcs_labels_widen");
    return GCS_OK;
}

// ======================================================================= boundary scoring (§8f-1)
// Integer restatement of /root/reference/BSD_metrics/metrics.py:25-51,58-96 for ONE image:
//   bd(M)   = thick boundaries of an integer map M: max != min over the 3x3 cross (find_boundaries
//             defaults; reflect border == clamped indices for max/min filters)
//   dil5(b) = 5x5 binary dilation (dilation(., rectangle(5,5)); same border argument)
// counts[0] = sum bd(L);  per annotator a: counts[1+3a] = sum dil5(bd(L)) & bd(T_a)   (recall numerator)
//                                           counts[2+3a] = sum bd(T_a)                 (recall denominator)
//                                           counts[3+3a] = sum bd(L) & dil5(bd(T_a))   (precision numerator)
// The float divisions and the per-annotator mean stay on the host, in the reference's order.
template <typename T>
__device__ __forceinline__ bool thick_boundary(const T *m, int H, int W, int y, int x) {
    const T c = m[(size_t)y * W + x];
    const T u = m[(size_t)max(y - 1, 0) * W + x], d = m[(size_t)min(y + 1, H - 1) * W + x];
    const T l = m[(size_t)y * W + max(x - 1, 0)], r = m[(size_t)y * W + min(x + 1, W - 1)];
    return u != c || d != c || l != c || r != c;   // max != min over {c,u,d,l,r}
}

// maps: plane 0 = boundaries of the label map, planes 1..A = boundaries of the annotator maps
__global__ void boundary_maps_kernel(const int32_t *__restrict__ labels, const uint16_t *__restrict__ truth, int A,
                                     int H, int W, uint8_t *__restrict__ maps) {
    const int n = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (A + 1) * n; i += gridDim.x * blockDim.x) {
        const int a = i / n, p = i % n, y = p / W, x = p % W;
        maps[i] = a == 0 ? thick_boundary(labels, H, W, y, x)
                         : thick_boundary(truth + (size_t)(a - 1) * n, H, W, y, x);
    }
}

__device__ __forceinline__ bool dilated5(const uint8_t *b, int H, int W, int y, int x) {
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            if (xx >= 0 && xx < W && b[(size_t)yy * W + xx]) return true;
        }
    }
    return false;
}

__global__ void boundary_counts_kernel(const uint8_t *__restrict__ maps, int A, int H, int W,
                                       unsigned long long *__restrict__ counts) {
    const int n = H * W;
    const int a = blockIdx.y;                         // 0: label-only count, 1..A: annotator a-1
    unsigned c0 = 0, c1 = 0, c2 = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int y = p / W, x = p % W;
        const bool bl = maps[p];
        if (a == 0) {
            c0 += bl;
        } else {
            const uint8_t *tb = maps + (size_t)a * n;
            const bool bt = tb[p];
            c0 += bt && dilated5(maps, H, W, y, x);  // recall numerator
            c1 += bt;                                  // recall denominator
            c2 += bl && dilated5(tb, H, W, y, x);    // precision numerator
        }
    }
    // wave reduction, one atomic per wave (integers: order-independent)
    for (int m = 32; m >= 1; m >>= 1) {
        c0 += __shfl_xor(c0, m);
        c1 += __shfl_xor(c1, m);
        c2 += __shfl_xor(c2, m);
    }
    if ((threadIdx.x & 63) == 0) {
        if (a == 0) {
            atomicAdd(&counts[0], (unsigned long long)c0);
        } else {
            atomicAdd(&counts[1 + 3 * (a - 1)], (unsigned long long)c0);
            atomicAdd(&counts[2 + 3 * (a - 1)], (unsigned long long)c1);
            atomicAdd(&counts[3 + 3 * (a - 1)], (unsigned long long)c2);
        }
    }
}

extern "C" size_t gcs_boundary_scratch_bytes(int A, int H, int W) {
    if (A <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)(A + 1) * H * W;
}

extern "C" int gcs_boundary_counts(const int32_t *labels, const uint16_t *truth, int A, int H, int W, void *scratch,
                                   uint64_t *counts, gcs_stream_t stream) {
    if (!labels || !truth || !scratch || !counts) return fail(GCS_EINVAL, "gcs_boundary_counts: NULL pointer");
    if (A <= 0 || A > 65535 || H <= 0 || W <= 0 || (long long)H * W * (A + 1) > 0x7fffffffLL)
        return fail(GCS_EINVAL, "gcs_boundary_counts: bad shape");
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)(1 + 3 * A) * sizeof(uint64_t), stream);
    if (e != hipSuccess) return hip_fail(e, "gcs_boundary_counts(memset)");
    uint8_t *maps = static_cast<uint8_t *>(scratch);
    const int n = H * W;
    hipLaunchKernelGGL(boundary_maps_kernel, dim3(min(2048, ((A + 1) * n + 255) / 256)), dim3(256), 0, stream, labels,
                       truth, A, H, W, maps);
    GCS_CHECK_LAUNCH("gcs_boundary_counts(maps)");
    hipLaunchKernelGGL(boundary_counts_kernel, dim3(min(256, (n + 255) / 256), A + 1), dim3(256), 0, stream, maps, A, H,
                       W, reinterpret_cast<unsigned long long *>(counts));
    GCS_CHECK_LAUNCH("gcs_boundary_counts");
    return GCS_OK;
}

// ================================================================== connected regions (§8f-4)
// SPEC.md §7: 4-connected components of equal labels, renumbered 0,1,2,... in raster order of each
// component's first pixel (so "Regions" = max+1 at /root/reference/BSD_metrics/metrics.py:51 counts
// connected regions, as it does for the SLIC output the slot holds today). Lock-free union-find:
// parents only ever decrease (atomicMin), a root is the smallest pixel index of its component, and a
// failed link (someone re-parented the node meanwhile) retries from the displaced parent, so no
// equivalence is lost even when a find reads a stale pointer.
__device__ __forceinline__ int cc_find(const int *parent, int x) {
    for (;;) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
}

__device__ __forceinline__ void cc_unite(int *parent, int a, int b) {
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // link the larger root under the smaller
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                          // a was re-parented meanwhile: merge that chain too
    }
}

__global__ void cc_union_kernel(const int32_t *__restrict__ labels, int H, int W, int *__restrict__ parent) {
    const int P = H * W;
    const int32_t *lab = labels + (size_t)blockIdx.y * P;
    int *par = parent + (size_t)blockIdx.y * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int y = p / W, x = p % W;
        const int32_t l = lab[p];
        if (x + 1 < W && lab[p + 1] == l) cc_unite(par, p, p + 1);
        if (y + 1 < H && lab[p + W] == l) cc_unite(par, p, p + W);
    }
}

__global__ void cc_local_init_kernel(int H, int W, int *__restrict__ parent) {
    const int P = H * W;
    int *par = parent + (size_t)blockIdx.y * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) par[p] = p;
}

// one workgroup per image: flatten, count roots per contiguous chunk, scan, hand out ids in raster order
__global__ __launch_bounds__(1024) void cc_rank_kernel(int H, int W, int *__restrict__ parent, int *__restrict__ rootid) {
    __shared__ int s_cnt[1024];
    const int P = H * W;
    int *par = parent + (size_t)blockIdx.x * P;
    int *rid = rootid + (size_t)blockIdx.x * P;
    const int tid = threadIdx.x;
    const int chunk = (P + 1023) / 1024;
    const int lo = min(P, tid * chunk), hi = min(P, lo + chunk);
    int cnt = 0;
    for (int p = lo; p < hi; ++p) cnt += par[p] == p;
    s_cnt[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
        const int v = tid >= off ? s_cnt[tid - off] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    int id = s_cnt[tid] - cnt;                           // exclusive prefix = first id of this chunk
    for (int p = lo; p < hi; ++p)
        if (par[p] == p) rid[p] = id++;
}

__global__ void cc_relabel_kernel(int H, int W, const int *__restrict__ parent, const int *__restrict__ rootid,
                                  int32_t *__restrict__ out) {
    const int P = H * W;
    const int *par = parent + (size_t)blockIdx.y * P;
    const int *rid = rootid + (size_t)blockIdx.y * P;
    int32_t *o = out + (size_t)blockIdx.y * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        int r = par[p];
        while (par[r] != r) r = par[r];                  // the union kernel has finished: plain loads are current
        o[p] = rid[r];
    }
}

extern "C" size_t gcs_connected_scratch_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)2 * B * H * W * sizeof(int32_t);
}

extern "C" int gcs_connected_regions(const int32_t *labels, int B, int H, int W, void *scratch, int32_t *out,
                                     gcs_stream_t stream) {
    if (!labels || !scratch || !out) return fail(GCS_EINVAL, "gcs_connected_regions: NULL pointer");
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL)
        return fail(GCS_EINVAL, "gcs_connected_regions: bad shape");
    const int P = H * W;
    int *parent = static_cast<int *>(scratch);
    int *rootid = parent + (size_t)B * P;
    const dim3 grid(min(1024, (P + 255) / 256), B), block(256);
    hipLaunchKernelGGL(cc_local_init_kernel, grid, block, 0, stream, H, W, parent);
    GCS_CHECK_LAUNCH("gcs_connected_regions(init)");
    hipLaunchKernelGGL(cc_union_kernel, grid, block, 0, stream, labels, H, W, parent);
    GCS_CHECK_LAUNCH("gcs_connected_regions(union)");
    hipLaunchKernelGGL(cc_rank_kernel, dim3(B), dim3(1024), 0, stream, H, W, parent, rootid);
    GCS_CHECK_LAUNCH("gcs_connected_regions(rank)");
    hipLaunchKernelGGL(cc_relabel_kernel, grid, block, 0, stream, H, W, parent, rootid, out);
    GCS_CHECK_LAUNCH("gcs_connected_regions");
    return GCS_OK;
}

// ======================================================================= region tables (§8f-2)
// Integer part of /root/reference/BSD_metrics/metrics.py:102-146 (label x annotator contingency table and region
// areas) and :160-181 (4-neighbour perimeter: image-border pixels, or pixels with a different 4-neighbour). One
// thread per pixel; workgroup-private tables in LDS when they fit (k-means label maps: a few clusters, every
// atomic on a handful of addresses), global atomics otherwise (connected regions: thousands of sparse rows).
__global__ __launch_bounds__(256) void region_counts_kernel(const int32_t *__restrict__ labels,
                                                            const uint16_t *__restrict__ truth, int A, int H, int W,
                                                            int n_seg, int stride, int use_lds,
                                                            unsigned *__restrict__ hist, unsigned *__restrict__ area,
                                                            unsigned *__restrict__ perim) {
    extern __shared__ unsigned s_tab[];                        // [A][n_seg][stride] hist | [n_seg] area | [n_seg] perim
    const int n_hist = A * n_seg * stride, n_tab = n_hist + 2 * n_seg;
    if (use_lds) {
        for (int i = threadIdx.x; i < n_tab; i += blockDim.x) s_tab[i] = 0u;
        __syncthreads();
    }
    unsigned *t_hist = use_lds ? s_tab : hist, *t_area = use_lds ? s_tab + n_hist : area,
             *t_perim = use_lds ? s_tab + n_hist + n_seg : perim;
    const int P = H * W;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int l = labels[p];
        if ((unsigned)l >= (unsigned)n_seg) continue;          // caller passes n_seg = max + 1; never index outside
        const int y = p / W, x = p - y * W;
        bool edge = y == 0 || y == H - 1 || x == 0 || x == W - 1;
        if (!edge) edge = labels[p - W] != l || labels[p + W] != l || labels[p - 1] != l || labels[p + 1] != l;
        atomicAdd(&t_area[l], 1u);
        if (edge) atomicAdd(&t_perim[l], 1u);
        for (int a = 0; a < A; ++a) {
            const int t = truth[(size_t)a * P + p];
            if (t < stride) atomicAdd(&t_hist[((size_t)a * n_seg + l) * stride + t], 1u);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_tab; i += blockDim.x) {
            const unsigned v = s_tab[i];
            if (v) atomicAdd(i < n_hist ? &hist[i] : i < n_hist + n_seg ? &area[i - n_hist] : &perim[i - n_hist - n_seg], v);
        }
    }
}

extern "C" int gcs_region_counts(const int32_t *labels, const uint16_t *truth, int A, int H, int W, int n_segments,
                                 int n_truth_labels, uint32_t *hist, uint32_t *area, uint32_t *perim,
                                 gcs_stream_t stream) {
    if (!labels || !truth || !hist || !area || !perim) return fail(GCS_EINVAL, "gcs_region_counts: NULL pointer");
    if (A <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL || n_segments <= 0 || n_truth_labels <= 0 ||
        (long long)A * n_segments * n_truth_labels > 0x3fffffffLL)
        return fail(GCS_EINVAL, "gcs_region_counts: bad shape");
    const size_t n_hist = (size_t)A * n_segments * n_truth_labels;
    hipError_t e = hipMemsetAsync(hist, 0, n_hist * sizeof(uint32_t), stream);
    if (e == hipSuccess) e = hipMemsetAsync(area, 0, (size_t)n_segments * sizeof(uint32_t), stream);
    if (e == hipSuccess) e = hipMemsetAsync(perim, 0, (size_t)n_segments * sizeof(uint32_t), stream);
    if (e != hipSuccess) return hip_fail(e, "gcs_region_counts(memset)");
    const size_t lds = (n_hist + 2 * (size_t)n_segments) * sizeof(unsigned);
    const int use_lds = lds <= 48 * 1024;
    const int P = H * W;
    const int blocks = use_lds ? min(256, (P + 1023) / 1024) : min(2048, (P + 255) / 256);
    hipLaunchKernelGGL(region_counts_kernel, dim3(blocks), dim3(256), use_lds ? lds : 0, stream, labels, truth, A, H, W,
                       n_segments, n_truth_labels, use_lds, hist, area, perim);
    GCS_CHECK_LAUNCH("gcs_region_counts");
    return GCS_OK;
}
