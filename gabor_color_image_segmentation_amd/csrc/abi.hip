// abi.hip — host-only part of the C ABI (include/gcs.h): error plumbing, slab geometry, bank packing.
// The reference ships no code for this path; the slot filled is
// /root/reference/BSD_metrics/script.py:30. Arithmetic: SPEC.md.
#include "common.h"

static thread_local char g_err[256] = "";
int gcs_fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
int gcs_hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return GCS_EHIP;
}

int gcs_cu_count() {
    static int cached[64];                                   // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cached[dev] = prop.multiProcessorCount;
        else { (void)hipGetLastError(); cached[dev] = 256; }
    }
    return cached[dev];
}
extern "C" int gcs_device_cu_count(void) { return gcs_cu_count(); }

// Device -> pinned host memory through the copy engines. hipMemcpyAsync / hipMemcpyDtoHAsync in this direction are served by a
// blit kernel (__amd_rocclr_copyBuffer in the kernel trace) that fills the chip: for as long as PCIe takes, the kernels of
// every other stream crawl. The pitched form goes down the runtime's rectangle path, which hands the copy to SDMA
// (tools/d2h_engine_probe.py: 55 GB/s, no kernel in the trace; profiles/r3_notes.md).
extern "C" int gcs_download(const void *src_dev, void *dst_host, size_t bytes, gcs_stream_t stream) {
    if (!bytes) return GCS_OK;
    if (!src_dev || !dst_host) return gcs_fail(GCS_EINVAL, "gcs_download: NULL pointer");
    const size_t row = 65536, rows = bytes / row, rest = bytes - rows * row;
    hipError_t e = hipSuccess;
    if (rows) e = hipMemcpy2DAsync(dst_host, row, src_dev, row, row, rows, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && rest)
        e = hipMemcpy2DAsync(static_cast<char *>(dst_host) + rows * row, rest, static_cast<const char *>(src_dev) + rows * row,
                             rest, rest, 1, hipMemcpyDeviceToHost, stream);
    return e == hipSuccess ? GCS_OK : gcs_hip_fail(e, "gcs_download(hipMemcpy2DAsync)");
}

extern "C" int gcs_abi_version(void) { return GCS_ABI_VERSION; }
extern "C" const char *gcs_last_error(void) { return g_err; }

// ------------------------------------------------------------------------- geometry (host)
static bool bank_levels(int n_scales, int n_orient, int FL[GCS_LEVELS_MAX], int *n_levels) {
    if (n_scales < 1 || n_scales > GCS_SCALES_MAX || n_orient < 1 || n_orient > 2730) return false;
    *n_levels = (n_scales + 1) / 2;
    for (int L = 0; L < *n_levels; ++L) FL[L] = (n_scales - 2 * L >= 2 ? 2 : 1) * n_orient;
    return true;
}

extern "C" size_t gcs_bank_packed_bytes(int n_scales, int n_orient) {
    int FL[GCS_LEVELS_MAX], nl;
    if (!bank_levels(n_scales, n_orient, FL, &nl)) return 0;
    size_t mt = 0;
    for (int L = 0; L < nl; ++L) mt += mtiles(FL[L]);
    return mt * 8 * 64 * 16;
}
extern "C" size_t gcs_bank_bias_count(int n_scales, int n_orient) {
    int FL[GCS_LEVELS_MAX], nl;
    if (!bank_levels(n_scales, n_orient, FL, &nl)) return 0;
    size_t mt = 0;
    for (int L = 0; L < nl; ++L) mt += mtiles(FL[L]);
    return mt * 4;
}
extern "C" size_t gcs_feature_slab_bytes(int B, int H, int W, int n_scales, int n_orient) {
    GcsLayout lo;
    if (B <= 0 || !gcs_make_layout(H, W, n_scales, n_orient, &lo)) return 0;
    return (size_t)B * (size_t)lo.img_bytes;      // per image: ntiles * tile_bytes feature bytes (+ the flag words of a split slab)
}
extern "C" size_t gcs_feature_pass_bytes(int B, int H, int W, int n_scales, int n_orient) {
    GcsLayout lo;
    if (B <= 0 || !gcs_make_layout(H, W, n_scales, n_orient, &lo)) return 0;
    return (size_t)B * lo.ntiles * (lo.split ? (size_t)lo.S + lo.S / 2 : (size_t)lo.tile_bytes);
}
extern "C" size_t gcs_label_slab_bytes(int B, int H, int W) {            // uint8 raster map [B][H][W], padded to 16 bytes
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return ((size_t)B * H * W + 15) / 16 * 16;
}
// Workgroups (= partial-sum rows) per image of one Lloyd pass. Sized to the machine: the pass
// kernel runs 3 workgroups per CU, so B*parts aims at one full wave of 256*3 workgroups (a
// second, partly filled wave of workgroups would idle most of the chip). Lower bound: one
// workgroup's int32 accumulators must not overflow (<= 65536 pixels); upper bound: at least
// 2 tiles per workgroup (its prologue - centroids, key bases, A fragments - costs about as much as one tile).
#ifndef GCS_KP_SLOTS
#define GCS_KP_SLOTS 768
#endif
extern "C" size_t gcs_kmeans_parts_per_image(int B, int H, int W) {
    // from the shape alone (tiles without packed edge strips: an upper bound for every bank), so that the partial-sum
    // buffer and the reduce calls need not know the bank
    if (B <= 0 || H <= 0 || W <= 0 || gcs_tiles_upper(H, W) > 0x0fffffffLL) return 0;
    const size_t px = (size_t)gcs_tiles_upper(H, W) * KP_TP;
    const size_t need = (px + 65535) / 65536;
    // small batches (B * parts would leave most of the 768 slots empty): down to 2 tiles per workgroup - one image then
    // runs on 313 workgroups instead of 78 and a pass takes a third of the time (it is latency-bound: 14 MB per image).
    // One tile per workgroup (626 rows): the pass takes the same 15.6 us - its cost is the launch, the centroid prologue
    // and the fold, not the tiles - and the reduce of twice the rows makes the step slower (0.223 vs 0.208 ms); three
    // tiles: 0.217 ms.
#ifndef GCS_KP_SMALL_TILES
#define GCS_KP_SMALL_TILES 2
#endif
    size_t most = px / KP_TP / GCS_KP_SMALL_TILES;
    if (most < 1) most = 1;
    size_t want = (GCS_KP_SLOTS + (size_t)B - 1) / (size_t)B;
    if (want > most) want = most;
    return need > want ? need : want;
}
extern "C" size_t gcs_kmeans_partial_bytes(int B, int H, int W, int D, int k) {
    if (B <= 0 || D <= 0 || k <= 0) return 0;
    return (size_t)B * gcs_kmeans_parts_per_image(B, H, W) * partial_chunks(k * (D + 1)) * KP_PCH * sizeof(uint64_t);
}

// ------------------------------------------------------------------------ bank pack (host)
// A-fragment of v_mfma_i32_32x32x32_i8: lane l = (r = l&31, h = l>>5) supplies row r of the
// 32-row tile, 16 consecutive k. We bind k-slot (kk, h, j) to tap (dy = 2*kk + h, dx = j) of
// a 16x16 frame whose row 15 / column 15 are zero; both operands use the same binding, so
// only "A row = lane&31, B col = lane&31" and the C/D map (cdna guide §3) are relied on.
// Row r = 8i + 4hh + part of tile mt = filter 4*mt + 2*(i >> 1) + hh OF ITS LEVEL, part in {re_lo, re_hi, im_lo, im_hi},
// evaluated for the pixel s = i & 1 to the right of the B column's pixel: its taps sit s slots further right in the frame
// row (slot j holds dx = j - s; the frame is 16 wide, a 15-tap row shifted by one still fits). A lane's accumulator quads
// (rows 8i + 4h .. +3, i = 0..3) are then the two pixels of a pair for the two filters 2fp + h of the tile. Levels are
// packed one after the other, each starting on a fresh tile: [level][mt][kk][lane][16 bytes]; bias [level][mt*4 + f%4].
extern "C" int gcs_bank_pack(const int16_t *tapq, int n_scales, int n_orient, int ks, int8_t *packed, int32_t *bias) {
    if (!tapq || !packed || !bias) return gcs_fail(GCS_EINVAL, "gcs_bank_pack: NULL pointer");
    int FL[GCS_LEVELS_MAX], nl;
    if (!bank_levels(n_scales, n_orient, FL, &nl))
        return gcs_fail(GCS_EINVAL, "gcs_bank_pack: need 1 <= n_scales <= 8 and n_orient >= 1");
    if (ks < 1 || ks > GCS_KSIZE_MAX || (ks & 1) == 0)
        return gcs_fail(GCS_EINVAL, "gcs_bank_pack: ksize must be odd and <= 15");
    const int off = (GCS_KSIZE_MAX - ks) / 2; // centre smaller kernels in the 15x15 frame
    memset(packed, 0, gcs_bank_packed_bytes(n_scales, n_orient));
    memset(bias, 0, gcs_bank_bias_count(n_scales, n_orient) * sizeof(int32_t));
    int mt_base = 0;
    for (int L = 0; L < nl; ++L) {
        const int f0 = 2 * L * n_orient;
        for (int fl = 0; fl < FL[L]; ++fl) {
            const int f = f0 + fl;
            long s_re = 0, s_im = 0, a_re = 0, a_im = 0;
            for (int t = 0; t < ks * ks; ++t) {
                const int q_re = tapq[((size_t)f * 2 + 0) * ks * ks + t], q_im = tapq[((size_t)f * 2 + 1) * ks * ks + t];
                s_re += q_re;
                s_im += q_im;
                a_re += q_re < 0 ? -q_re : q_re;
                a_im += q_im < 0 ? -q_im : q_im;
            }
            if (s_im != 0) return gcs_fail(GCS_EINVAL, "gcs_bank_pack: imaginary taps must sum to zero");
            // |response| <= 255 * sum|tapq| must stay below 2^23, so that the Q7 value (response >> 8 at the Q15 shift) is a
            // 16-bit integer and re^2 + im^2 < 2^31: the domain of the kernel's exact square root (SPEC.md §3; an envelope of
            // unit DC gain gives sum|tapq| <= 2^15 + ks^2 / 2)
            if (a_re > GCS_TAP_ABS_SUM_MAX || a_im > GCS_TAP_ABS_SUM_MAX)
                return gcs_fail(GCS_EINVAL, "gcs_bank_pack: sum of |tapq| of a filter exceeds 32896 (response would leave 24 bits)");
            bias[mt_base * 4 + fl] = (int32_t)(128 * s_re);
        }
        for (int mt = 0; mt < mtiles(FL[L]); ++mt)
            for (int kk = 0; kk < 8; ++kk)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, h = lane >> 5;
                    const int i = r >> 3, hh = (r >> 2) & 1, part = r & 3;
                    const int fl = 4 * mt + 2 * (i >> 1) + hh, s = i & 1;
                    int8_t *dst = packed + (((size_t)(mt_base + mt) * 8 + kk) * 64 + lane) * 16;
                    if (fl >= FL[L]) continue;
                    const int f = f0 + fl;
                    const int dy = 2 * kk + h - off;
                    if (dy < 0 || dy >= ks) continue;
                    for (int j = 0; j < 16; ++j) {
                        const int dx = j - s - off;
                        if (dx < 0 || dx >= ks) continue;
                        const int q = tapq[(((size_t)f * 2 + (part >> 1)) * ks + dy) * ks + dx];
                        if (q > 32639 || q < -32639)
                            return gcs_fail(GCS_EINVAL, "gcs_bank_pack: tap outside two-digit range");
                        const int lo = ((q + 128) & 255) - 128;
                        const int hi = (q - lo) >> 8;
                        dst[j] = (int8_t)((part & 1) ? hi : lo);
                    }
                }
        mt_base += mtiles(FL[L]);
    }
    return GCS_OK;
}
