/* gcs.h — C ABI of libgcs.so: the MI355X (gfx950) Gabor-bank + k-means segmenter path.
 *
 * Drop-in boundary: the reference has NO plugin/operator/FFI registry for this path; its
 * whole interface is one positional Python call,
 *     labels = <callable>(img)            /root/reference/BSD_metrics/script.py:30
 * with img (H,W,3) uint8 (script.py:25) and labels (H,W) integer consumed by
 * metrics.__init__ (/root/reference/BSD_metrics/metrics.py:43-51). The functions below are
 * what a binding for that slot calls underneath; INTEGRATION.md shows the
 * ctypes stub. Arithmetic is defined by SPEC.md (exact integers).
 *
 * Conventions: every pointer named *_dev is device memory owned by the caller; no device memory is
 * allocated or freed and the host is never blocked here; work is ordered on `stream` (a hipStream_t; NULL = default
 * stream): when a call returns, everything it enqueued precedes whatever the caller enqueues on `stream` next.
 * gcs_gabor_features may run part of a large batch on a library-owned side stream (one per device, created on first
 * use) between an event fork from and an event join back into `stream` — the pattern stream capture records as a
 * graph. Return 0 on success, GCS_E* otherwise, with a thread-local message in gcs_last_error(). Not thread-safe per
 * buffer.
 */
#ifndef GCS_H
#define GCS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t *gcs_stream_t; /* == hipStream_t */

enum {
    GCS_OK = 0,
    GCS_EINVAL = 1, /* bad argument (shape, range, NULL) */
    GCS_EHIP = 2    /* HIP runtime error at launch */
};

#define GCS_ABI_VERSION 18
#define GCS_KSIZE_MAX 15  /* tap frame: 15 rows x 16 columns (SPEC.md §2) */
#define GCS_K_MAX 16      /* clusters */
#define GCS_TAP_ABS_SUM_MAX 32896 /* per filter and part: 255 * sum|tapq| < 2^23 (gcs_bank_pack rejects larger banks) */
#define GCS_SCALES_MAX 8  /* octave pyramid of at most 4 levels: scales 2L, 2L+1 run on level L (SPEC.md §2) */

int gcs_abi_version(void);
/* Compute units of the current HIP device (hipDeviceProp_t::multiProcessorCount): the persistent Gabor grid is sized from it;
 * the Lloyd pass geometry (gcs_kmeans_parts_per_image: a pure host function, so that buffers can be sized without a device)
 * is tuned for the 256 CUs of an MI355X in SPX mode and merely less efficient elsewhere. */
int gcs_device_cu_count(void);
const char *gcs_last_error(void);

/* The bank is described by (n_scales, n_orient): F = n_scales * n_orient filters, f = s * n_orient + o,
 * D = 3F features per pixel, d = c * F + f (SPEC.md §2-§3). */

/* ---- host-only helpers (no GPU needed) ------------------------------------------------ */

/* Filters are packed level by level, each level padded to a multiple of 4 filters (one 32-row MFMA tile =
 * 4 filters x {re,im} x {lo,hi} digits x 2 pixel shifts). Bytes of the packed A-operand image / number of int32 bias words. */
size_t gcs_bank_packed_bytes(int n_scales, int n_orient);
size_t gcs_bank_bias_count(int n_scales, int n_orient);

/* Pack quantised taps tapq[F][2][ks][ks] (int16, SPEC.md §2) into the lane-linear int8 A-fragments of
 * v_mfma_i32_32x32x32_i8 and the per-filter bias 128*sum(tapq_re) (pixels are fed as img-128).
 * Replaces nothing in the reference (bank absent). */
int gcs_bank_pack(const int16_t *tapq, int n_scales, int n_orient, int ksize, int8_t *packed, int32_t *bias);

/* Feature slab: every pyramid level at its own resolution, tile-major (a tile = four 8x8-pixel blocks with
 * their level-1..3 parents, one contiguous run; edge strips of one or two pixel columns / rows - both sides of a BSD500
 * image are 8k + 1 pixels - are packed into virtual blocks for banks of at most two levels; see csrc/common.h), uint16
 * stored offset-binary (x ^ 0x8080: both bytes are signed MFMA digits). Opaque to callers; gcs_features_unpack gives the
 * canonical [B][D][H][W] uint16 tensor of SPEC.md §3. "Label slab": a uint8 label map in RASTER order, [B][H][W]
 * (gcs_label_slab_bytes = B*H*W rounded up to 16). */
size_t gcs_feature_slab_bytes(int B, int H, int W, int n_scales, int n_orient);
/* ABI 18: banks of at most two pyramid levels with D <= 79 (every 4x6-style bank) keep the slab SPLIT: per image three planar
 * arrays in the same tile / plane / slot order - the low bytes, bits 8..11 and bits 12..15 of every value - and one flag word
 * per tile that says whether any of the tile's bits 12..15 is set (values of 4096 and more are 6e-5 of what BSD500 produces:
 * profiles/r6_notes.md). A Lloyd pass streams the first two arrays and, for flagged tiles only, the third: exact for any data.
 * gcs_feature_slab_bytes(B, ...) == B * gcs_feature_slab_bytes(1, ...) for every bank: image b's slab starts at b times that
 * (callers may hand a sub-batch's part of a slab to any entry point). gcs_feature_pass_bytes: the feature bytes ONE Lloyd
 * pass reads when no tile is flagged (3/4 of the feature bytes of a split slab, all of them otherwise; padding included). */
size_t gcs_feature_pass_bytes(int B, int H, int W, int n_scales, int n_orient);
size_t gcs_label_slab_bytes(int B, int H, int W);
/* uint64 partial sums written by one assign pass: one row of k * (D+1) values per k-means workgroup, stored in chunks
 * of 16 elements, padded (opaque: only gcs_kmeans_reduce / gcs_kmeans_reduce_finalize read them). */
size_t gcs_kmeans_parts_per_image(int B, int H, int W);
size_t gcs_kmeans_partial_bytes(int B, int H, int W, int D, int k);

/* Device memory -> pinned host memory on `stream` through the copy engines (SDMA). The pipelined host path
 * (Segmenter.segment_stream) uses it instead of hipMemcpyAsync, which in this direction is served by a chip-filling blit
 * kernel that stalls the kernels of every other stream for as long as PCIe takes. */
int gcs_download(const void *src_dev, void *dst_host, size_t bytes, gcs_stream_t stream);

/* ---- device entry points ---------------------------------------------------------------- */

/* Scratch for gcs_gabor_features: reflect-padded planar (pixel-128) pyramid levels. */
size_t gcs_gabor_workspace_bytes(int B, int H, int W, int n_scales);

/* SPEC.md §3: img_dev [B][H][W][3] uint8 -> feats_dev slab (pyramid + filter bank + magnitude).
 * Fills the slot's first stage (script.py:30). Requires H, W >= 8; ksize and shift are those of the packed bank
 * (ksize <= 13 and shift == 8, i.e. every default-style Q15 bank, take the shorter kernels). workspace_dev:
 * gcs_gabor_workspace_bytes() bytes of device scratch, contents undefined before and after. */
int gcs_gabor_features(const uint8_t *img_dev, int B, int H, int W, const int8_t *packed_dev,
                       const int32_t *bias_dev, int n_scales, int n_orient, int ksize, int shift, void *workspace_dev,
                       uint16_t *feats_dev, gcs_stream_t stream);

/* Slab -> canonical [B][D][H][W] uint16 (level-L responses replicated over 2^L blocks; tests / debugging). */
int gcs_features_unpack(const uint16_t *feats_dev, int B, int H, int W, int n_scales, int n_orient,
                        uint16_t *out_dev, gcs_stream_t stream);

/* SPEC.md §4 init. n_sets == B: per-image codebooks (set s from image s); n_sets == 1:
 * global codebook from image 0. centroids_dev: uint16 [n_sets][k][D]. */
int gcs_kmeans_init(const uint16_t *feats_dev, int B, int H, int W, int n_scales, int n_orient, int k,
                    int n_sets, uint16_t *centroids_dev, gcs_stream_t stream);

/* out_dev uint16 [n][D] = feature vectors of the n pixels byx_dev[i] = (image, row, col) (int32 triples in
 * device memory); b < 0 yields a zero row. Used to publish init centroids when one image is
 * sharded by rows over several ranks (BASELINE config 5). */
int gcs_features_gather(const uint16_t *feats_dev, int B, int H, int W, int n_scales, int n_orient, int n,
                        const int32_t *byx_dev, uint16_t *out_dev, gcs_stream_t stream);

/* SPEC.md §4 assign + per-workgroup partial sums (one streaming pass over the slab).
 * Only rows [row_lo, row_hi) of each image vote in the sums (whole image: 0, H); halo rows of
 * a row-sharded image are labelled but do not vote. `reverse` != 0 sweeps the slab back to front:
 * alternate it from pass to pass so that each pass starts on what the previous one left in the
 * Infinity Cache (results do not depend on it). labels_dev: uint8 [B][H][W] label map (every pixel of every image is
 * labelled, rows outside the voting window included); partials_dev:
 * gcs_kmeans_partial_bytes() bytes, fully overwritten (no zeroing needed). D <= 207 (every BASELINE bank) runs on
 * the matrix cores, wider feature vectors on a generic VALU pass; k <= GCS_K_MAX. The same n_sets must be passed
 * to the reduce call that follows (it selects the partial layout).
 * Either output may be NULL (not both): labels_dev == NULL skips the label store (the passes whose assignment nobody
 * reads: all but the last), partials_dev == NULL skips the sums (the last pass: assignment only). */
int gcs_kmeans_assign_accumulate(const uint16_t *feats_dev, const uint16_t *centroids_dev, int B,
                                 int H, int W, int n_scales, int n_orient, int k, int n_sets, int row_lo,
                                 int row_hi, int reverse, uint8_t *labels_dev, uint64_t *partials_dev,
                                 gcs_stream_t stream);

/* The LAST Lloyd pass in one launch: assignment only (SPEC.md §4; the schedule's last pass has no update) with the label map
 * written straight in raster order, out_dev [B][H][W] int32 (out_u8 == 0: what metrics.py:43-51 consumes) or uint8
 * (out_u8 != 0). Same result as gcs_kmeans_assign_accumulate(labels, NULL) (followed by gcs_labels_widen for int32).
 * scratch_labels_dev: a uint8 label map (gcs_label_slab_bytes) that is needed, and then also filled, only for int32 output of
 * feature vectors of 208 or more planes (the generic pass); may be NULL otherwise - for D <= 207 it is never written, even when
 * passed. Whole images only (no row window). */
int gcs_kmeans_assign_raster(const uint16_t *feats_dev, const uint16_t *centroids_dev, int B, int H, int W, int n_scales,
                             int n_orient, int k, int n_sets, int reverse, void *out_dev, int out_u8,
                             uint8_t *scratch_labels_dev, gcs_stream_t stream);

/* partials -> sums_dev int64 [n_sets][k][D+1] ([..][D] = count). Deterministic slab
 * reduction (no float, no atomics). In global mode the caller all-reduces sums_dev across
 * ranks (RCCL, int64 sum) between this call and gcs_kmeans_finalize. */
int gcs_kmeans_reduce(const uint64_t *partials_dev, int B, int H, int W, int D, int k, int n_sets,
                      int64_t *sums_dev, gcs_stream_t stream);

/* SPEC.md §4 update: c = floor((2S + n) / (2n)), empty cluster keeps its centroid. */
int gcs_kmeans_finalize(const int64_t *sums_dev, int n_sets, int k, int D,
                        uint16_t *centroids_dev, gcs_stream_t stream);

/* Single-rank update: gcs_kmeans_reduce + gcs_kmeans_finalize in ONE launch (no all-reduce needed
 * in between). sums_dev may be NULL (then only the centroids are written). */
int gcs_kmeans_reduce_finalize(const uint64_t *partials_dev, int B, int H, int W, int D, int k,
                               int n_sets, int64_t *sums_dev, uint16_t *centroids_dev,
                               gcs_stream_t stream);

/* uint8 label map [B][H][W] -> int32 [B][H][W] (the dtype handed to metrics.py:43). Both pointers 4-byte aligned. */
int gcs_labels_widen(const uint8_t *labels_dev, int B, int H, int W, int32_t *out_dev,
                     gcs_stream_t stream);

/* Test hook: counts in *bad_dev (uint32, device) the 4096-value chunks of [0, n_max] on which the kernels' 7-instruction
 * exact integer square root and its biased form in the epilogue (SPEC.md §3: n <= 2 * 32767^2 < 2^31, guaranteed by the
 * tap-sum bound of gcs_bank_pack) is wrong. Expected 0. */
int gcs_selftest_isqrt(unsigned n_max, unsigned *bad_dev, gcs_stream_t stream);
/* Test hook (host only): working workgroups per image of the deep-bank Lloyd pass for a batch shape, computed from the
 * shape's tile count WITHOUT packed edge strips (an upper bound of every bank's tile count; the launcher uses the bank's own
 * count, which can only lower the pixels per workgroup); B * H * W / that many pixels per workgroup must stay below the
 * int32 accumulator bound (262 144 pixels). 0 for a bad shape. */
int gcs_selftest_native_parts(int B, int H, int W);

/* ---- boundary scoring of one image -------------------------------------- */

/* Integer part of /root/reference/BSD_metrics/metrics.py:25-51 (thick find_boundaries of the label
 * map and of each annotator map), :58-74 (recall) and :77-96 (precision): labels_dev int32 [H][W],
 * truth_dev uint16 [A][H][W] (the `Segmentation` arrays groundtruth.py:22-26 returns). Writes
 * counts_dev uint64 [1 + 3A]: [0] = #boundary pixels of the label map (metrics.py:90); for
 * annotator a: [1+3a] = sum(dilate5(bd(labels)) & bd(T_a)) and [2+3a] = sum(bd(T_a)) (metrics.py:69-72),
 * [3+3a] = sum(bd(labels) & dilate5(bd(T_a))) (metrics.py:93-94). The caller divides and averages in
 * the reference's order. scratch_dev: gcs_boundary_scratch_bytes() bytes. */
size_t gcs_boundary_scratch_bytes(int A, int H, int W);
int gcs_boundary_counts(const int32_t *labels_dev, const uint16_t *truth_dev, int A, int H, int W,
                        void *scratch_dev, uint64_t *counts_dev, gcs_stream_t stream);

/* Batched form: labels_dev int32 [B][H][W], truth_dev uint16 [T][H][W] = the annotator maps of image 0, then of image 1, ...
 * (T = total annotators; ragged: BSD500 has 4-9 per image, groundtruth.py:33-50), img_of_dev int32 [T] = image of each
 * annotator map. ONE pair of launches for the whole batch. counts_dev uint64 [B + 3T]: [b] = #boundary pixels of label map
 * b; for annotator t: [B+3t], [B+3t+1], [B+3t+2] = the three sums above. scratch_dev: gcs_boundary_batch_scratch_bytes(). */
size_t gcs_boundary_batch_scratch_bytes(int B, int T, int H, int W);
int gcs_boundary_counts_batch(const int32_t *labels_dev, const uint16_t *truth_dev, const int32_t *img_of_dev, int B, int T,
                              int H, int W, void *scratch_dev, uint64_t *counts_dev, gcs_stream_t stream);

/* ---- boundary scoring on RESIDENT ground truth (ABI 17) -------------------------------------- */

/* The annotator maps are constants of the data set: /root/reference/BSD_metrics/metrics.py:48-49 re-derives
 * find_boundaries(truth) for every image it scores and groundtruth.py:44-48 rescans the directories per id. gcs_truth_prepare
 * does that work ONCE per annotator map: truth_dev uint16 [T][H][W] -> planes_dev, gcs_bit_planes_bytes(T, H, W) bytes of BIT
 * planes (rows of 64-bit words, bit i of word w = pixel 64 w + i; all thick-boundary planes bd(T_t), then all 5x5-dilated planes),
 * bd_counts_dev uint64 [T] = sum bd(T_t) (the recall denominators, metrics.py:72), and - when truth8_dev is not NULL - the maps
 * narrowed to uint8 [T][H][W] for gcs_region_counts_batch_u8. The CALLER guarantees that every annotator label is below 256 when
 * it passes truth8_dev (BSD500: at most 208; the narrowing keeps the low byte and does not check: evaluate_gpu.DeviceTruth does).
 * The caller keeps all of it on the device for as long as it scores images of these ids. */
size_t gcs_bit_planes_bytes(int M, int H, int W);
int gcs_truth_prepare(const uint16_t *truth_dev, int T, int H, int W, void *planes_dev, uint64_t *bd_counts_dev,
                      uint8_t *truth8_dev, gcs_stream_t stream);

/* gcs_boundary_counts_batch on prepared annotator planes: the same counts_dev uint64 [B + 3T] (every element written, no
 * atomics), from the bit planes of the B label maps (scratch_dev: gcs_bit_planes_bytes(B, H, W)) ANDed with the resident planes.
 * seg_max_dev int32 [B] (may be NULL): the largest label of each map (metrics.py:51 wants max + 1). */
int gcs_boundary_counts_resident(const int32_t *labels_dev, const void *truth_planes_dev, const uint64_t *truth_bd_counts_dev,
                                 const int32_t *img_of_dev, int B, int T, int H, int W, void *scratch_dev, uint64_t *counts_dev,
                                 int32_t *seg_max_dev, gcs_stream_t stream);

/* ---- region tables of one image ------------------------------------------ */

/* Integer part of /root/reference/BSD_metrics/metrics.py:102-146 (undersegmentation: the label x annotator
 * contingency table `hist` and the region areas) and :160-201 (compactness: the 4-neighbour `perimeters`).
 * labels_dev int32 [H][W] with values in [0, n_segments) (metrics.py:51: n_segments = max + 1); truth_dev
 * uint16 [A][H][W] with values < n_truth_labels (the largest `max(truth) + 1` over the annotators,
 * metrics.py:116). Writes hist_dev uint32 [A][n_segments][n_truth_labels], area_dev and perim_dev uint32
 * [n_segments] (zeroed here). The caller does the reference's float arithmetic in the reference's order. */
int gcs_region_counts(const int32_t *labels_dev, const uint16_t *truth_dev, int A, int H, int W,
                      int n_segments, int n_truth_labels, uint32_t *hist_dev, uint32_t *area_dev,
                      uint32_t *perim_dev, gcs_stream_t stream);

/* Batched form (same ragged truth stack as gcs_boundary_counts_batch): first_dev int32 [B+1] = index of each image's first
 * annotator map (first[B] = T), max_annotators = the largest per-image count (it sizes the
 * workgroup-private tables; an image that brings more annotators than stated falls back to global atomics, still exact). hist_dev uint32 [T][n_segments][n_truth_labels],
 * area_dev / perim_dev uint32 [B][n_segments]. One launch for the whole batch. */
int gcs_region_counts_batch(const int32_t *labels_dev, const uint16_t *truth_dev, const int32_t *first_dev, int B, int T,
                            int max_annotators, int H, int W, int n_segments, int n_truth_labels, uint32_t *hist_dev,
                            uint32_t *area_dev, uint32_t *perim_dev, gcs_stream_t stream);

/* What metrics.py:128-140 takes from the tables of gcs_region_counts_batch, per annotator map t of image img_of[t]:
 * under_dev[t] = sum_seg (area[seg] - max_col hist[t][seg][col]) (metrics.py:129-130) and under_np_dev[t] = sum_seg sum_col
 * min(hist, rowsum - hist) (metrics.py:137-139), uint64 [T] each, integers. The host divides by H * W and averages over the
 * annotators in the reference's order without ever fetching the tables. */
int gcs_region_reduce(const uint32_t *hist_dev, const uint32_t *area_dev, const int32_t *img_of_dev, int T, int n_segments,
                      int n_truth_labels, uint64_t *under_dev, uint64_t *under_np_dev, gcs_stream_t stream);

/* gcs_region_counts_batch on annotator maps narrowed to uint8 by gcs_truth_prepare (n_truth_labels <= 256): half the bytes. */
int gcs_region_counts_batch_u8(const int32_t *labels_dev, const uint8_t *truth8_dev, const int32_t *first_dev, int B, int T,
                               int max_annotators, int H, int W, int n_segments, int n_truth_labels, uint32_t *hist_dev,
                               uint32_t *area_dev, uint32_t *perim_dev, gcs_stream_t stream);

/* Everything metrics.get_metrics() (metrics.py:246-255) needs of a batch, on resident ground truth, in one call (six launches:
 * gcs_boundary_counts_resident + gcs_region_counts_batch[_u8] + gcs_region_reduce with their zeroing folded into one launch). truth_maps_dev: the uint8 (truth_is_u8 != 0) or uint16 annotator maps; scratch_dev:
 * gcs_bit_planes_bytes(B, H, W); hist_dev [T][n_segments][n_truth_labels] is scratch the caller may keep on the device; the other
 * outputs as documented at the three calls. */
int gcs_score_batch_resident(const int32_t *labels_dev, const void *truth_planes_dev, const uint64_t *truth_bd_counts_dev,
                             const void *truth_maps_dev, int truth_is_u8, const int32_t *first_dev, const int32_t *img_of_dev, int B,
                             int T, int max_annotators, int H, int W, int n_segments, int n_truth_labels, void *scratch_dev,
                             uint32_t *hist_dev, uint64_t *counts_dev, int32_t *seg_max_dev, uint32_t *area_dev, uint32_t *perim_dev,
                             uint64_t *under_dev, uint64_t *under_np_dev, gcs_stream_t stream);

/* ---- connected regions (SPEC.md §7) -------------------------------------- */

/* labels_dev int32 [B][H][W] -> out_dev int32 [B][H][W]: 4-connected components of equal labels,
 * renumbered 0,1,2,... in raster order of each component's first pixel, per image. Makes
 * `Regions = max + 1` (/root/reference/BSD_metrics/metrics.py:51) count connected regions.
 * scratch_dev: gcs_connected_scratch_bytes() bytes. out_dev may not alias labels_dev. */
size_t gcs_connected_scratch_bytes(int B, int H, int W);
int gcs_connected_regions(const int32_t *labels_dev, int B, int H, int W, void *scratch_dev,
                          int32_t *out_dev, gcs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GCS_H */
