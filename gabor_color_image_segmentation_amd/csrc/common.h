// common.h — shared by the translation units of libgcs.so (gfx950 only): error plumbing and the
// geometry of the feature slab. Arithmetic: SPEC.md. Not part of the ABI (include/gcs.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gcs.h"

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------- errors
int gcs_fail(int code, const char *msg);                  // records the thread-local message, returns code
int gcs_hip_fail(hipError_t e, const char *what);
#define GCS_CHECK_LAUNCH(what)                               \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return gcs_hip_fail(e_, what); \
    } while (0)

// Compute units of the current device (hipDeviceProp_t::multiProcessorCount, cached per device): sizes the persistent grids.
// 256 on an MI355X in SPX mode, the machine the launch geometry was tuned on; a partitioned device reports fewer.
int gcs_cu_count();

static inline int round_up(int a, int m) { return (a + m - 1) / m * m; }
static inline int mtiles(int F) { return (F + 3) / 4; }   // 32-row MFMA tiles of a level: four filters each (csrc/abi.hip)

// ------------------------------------------------------------------------ slab geometry
// The feature slab keeps every pyramid level at its own resolution (SPEC.md §3). The image is cut into
// 8x8-pixel BLOCKS; a block owns 8x8 level-0 pixel SLOTS (iy, ix) and the slots of their parents: 4x4 level-1, 2x2 level-2
// and 1 level-3 (the parent of slot (iy, ix) at level L is slot (iy >> L, ix >> L)). Four consecutive blocks form a TILE =
// the 256 pixels one k-means workgroup step handles (one block per wave). In the WIDE slab (deep banks; every bank before
// round 6) a tile is ONE contiguous run of bytes:
//     [level 0: D_0 planes x 256 u16][level 1: D_1 x 64][level 2: D_2 x 16][level 3: D_3 x 4]
// (each level block padded to 16 bytes), slot order inside a plane = (block in tile, slot row, slot column) of that
// level's sub-block. Planes are in PHYSICAL order: level-major, then channel, then filter-in-level; logical
// feature d = c*F + f (SPEC.md §3) with f = 2L*n_orient + fl. Values are stored offset-binary (x ^ 0x8080) so
// that both bytes are signed MFMA digits.
//
// Which pixel sits in which slot (round 4):
//   MAIN blocks   bx_n x by_n blocks in raster order, block (by, bx) slot (iy, ix) = pixel (8 by + iy, 8 bx + ix).
//   EDGE STRIPS   both sides of every BSD500 image are 8k + 1 pixels (481 x 321): with main blocks alone the last block
//                 column and row would hold ONE pixel column / row each and 3.7 % of the bytes every Lloyd pass streams
//                 would be padding. For banks of at most two pyramid levels, a right edge of 1 or 2 columns
//                 (W mod 8 in {1, 2}) and a bottom edge of 1 or 2 rows are therefore packed into VIRTUAL blocks behind the
//                 main ones: a strip is one level-1 parent wide, a virtual block holds 16 consecutive parents of it
//                 (parent q = 4 (iy >> 1) + (ix >> 1) of the block's 4x4 parent slots) with their up to four pixels:
//                   right strip, block v:  pixel (2 (16 v + q) + (iy & 1), Wm + (ix & 1)),  rows 0 .. H-1 (corner included)
//                   bottom strip, block v: pixel (Hm + (iy & 1), 2 (16 v + q) + (ix & 1)),  columns 0 .. Wb-1
//                 so the level-1 replication the kernels do (slot (iy, ix) -> parent slot (iy >> 1, ix >> 1)) holds
//                 unchanged, and 481 x 321 takes 2 426 blocks (2 400 + 11 + 15) instead of 2 501. Deeper banks keep
//                 main blocks only (a strip block's 8 level-2 parents do not fit a block's 2x2 level-2 slots).
//
// SPLIT slab (round 6: banks of at most two levels with D <= 79, i.e. every 4x6-style bank). SPEC.md §3 allows g <= 46 163, but
// values of 4096 and more are 5.6e-5 of what BSD500 produces and 2e-7 of the synthetic bench batch (tools/design/
// narrow_slab_study.py, profiles/r6_notes.md), while every Lloyd pass streamed 16 bits for each. A value is therefore stored
// as three pieces in three planar arrays per image, tile-major and plane-major as above (a tile = S slots, level L starting at
// slot sl0[L]), but with the slots of a plane in (slot row, block in tile, slot column) order - a tile's level-0 plane is the
// raster of its 8 x 32 pixels -: the Gabor kernel's wave then stores 4 rows x 4 blocks x 8 pixels of a plane as ONE run of 128
// low bytes / 64 nibble bytes (block-major slots made those stores 32- and 16-byte pieces: the stage took 0.71 instead of 0.42 ms):
//     LO   [ntiles][S]      the low byte, XOR 0x80 (the signed low MFMA digit)
//     MID  [ntiles][S / 2]  bits 8..11, two slots per byte
//     TOP  [ntiles][S / 2]  bits 12..15, two slots per byte
//     FLAG [ntiles][4]      byte L != 0: some TOP nibble of the tile's level-L planes is non-zero
// A pass reads LO and MID of every tile (12 bits per value: 67.5 instead of 90 B/px for the default bank) and the TOP run of the
// tiles whose flag word is non-zero - exact for any data, and never more bytes than the wide slab. Nibble pairing inside a group of
// g = max(2, 8 >> L) consecutive slots (a block row on level 0, a parent row on level 1): byte i of the group (i < g / 2) holds
// slot i in bits 0..3 and slot i + g / 2 in bits 4..7, so that `x & 0x0f0f0f0f` and `(x >> 4) & 0x0f0f0f0f` are the high bytes of
// the group's first and second half in slot order (the passes' unpack) and a lane of the Gabor kernel owns whole bytes.
constexpr int GCS_LEVELS_MAX = 4;
constexpr int KP_TP = 256;            // pixels per tile = threads per k-means workgroup
constexpr int GCS_NO_STRIP = 1 << 29; // Wm / Hm of a layout without that strip: no coordinate reaches it

struct GcsLayout {
    int H, W;                         // full-resolution image
    int bx_n, by_n, nmain;            // MAIN 8x8 blocks per row / column / image
    int nR, nB;                       // virtual blocks of the right / bottom edge strip (0: not packed)
    int Wm, Hm;                       // first column of the right strip / first row of the bottom strip (GCS_NO_STRIP: none)
    int Wb;                           // columns of the bottom strip: [0, Wb)
    int nblk, ntiles;                 // blocks / tiles per image
    int n_levels, n_orient, F, D;     // pyramid levels, orientations, filters, features (3F)
    int FL[GCS_LEVELS_MAX];           // filters on level L
    int DL[GCS_LEVELS_MAX];           // planes on level L (3 * FL)
    int row0[GCS_LEVELS_MAX];         // first physical plane of level L
    int off[GCS_LEVELS_MAX];          // byte offset of level L inside a tile
    int HL[GCS_LEVELS_MAX], WL[GCS_LEVELS_MAX];   // level image size
    int tile_bytes;                   // feature bytes of one tile (wide: one contiguous run; split: 2 S, in three arrays)
    // split slab (see above); wide slabs: split == 0, img_bytes == ntiles * tile_bytes
    int split;
    int S;                            // slots per tile
    int sl0[GCS_LEVELS_MAX];          // first slot of level L inside a tile
    unsigned mid_off, top_off, flag_off;   // byte offsets of the MID / TOP / FLAG arrays inside one image's slab
    long long img_bytes;              // bytes of one image's slab (gcs_feature_slab_bytes(B) = B * img_bytes)
};

// Which banks take the split slab: those whose Lloyd pass is kmeans_pass_mfma_kernel's narrow bucket and whose Gabor stores are
// the level-0 / level-1 paths.
// (-DGCS_NO_SPLIT: every bank on the wide slab, for same-box A/B runs against the rounds 2-5 format)
#ifdef GCS_NO_SPLIT
static inline bool gcs_split_bank(int, int) { return false; }
#else
static inline bool gcs_split_bank(int n_levels, int D) { return n_levels <= 2 && D < 80; }
#endif

// Tiles per image when no edge strip is packed: an upper bound of every bank's tile count for the shape (sizes that
// must not depend on the bank: partial-sum rows per image).
static inline long long gcs_tiles_upper(int H, int W) {
    return (((long long)(W + 7) / 8) * ((H + 7) / 8) + 3) / 4;
}

// Returns false when the shape is not representable.
static inline bool gcs_make_layout(int H, int W, int n_scales, int n_orient, GcsLayout *lo) {
    if (H <= 0 || W <= 0 || n_scales < 1 || n_scales > 2 * GCS_LEVELS_MAX || n_orient < 1) return false;
    memset(lo, 0, sizeof *lo);
    lo->H = H;
    lo->W = W;
    lo->n_levels = (n_scales + 1) / 2;
    const bool may_pack = lo->n_levels <= 2 && H >= 8 && W >= 8;
    const bool pack_r = may_pack && ((W & 7) == 1 || (W & 7) == 2);
    const bool pack_b = may_pack && ((H & 7) == 1 || (H & 7) == 2);
    lo->bx_n = pack_r ? W / 8 : (W + 7) / 8;
    lo->by_n = pack_b ? H / 8 : (H + 7) / 8;
    lo->Wm = pack_r ? 8 * lo->bx_n : GCS_NO_STRIP;
    lo->Hm = pack_b ? 8 * lo->by_n : GCS_NO_STRIP;
    lo->Wb = pack_r ? lo->Wm : W;
    lo->nR = pack_r ? ((H + 1) / 2 + 15) / 16 : 0;
    lo->nB = pack_b ? ((lo->Wb + 1) / 2 + 15) / 16 : 0;
    const long long nmain = (long long)lo->bx_n * lo->by_n, nblk = nmain + lo->nR + lo->nB;
    if (nblk > 0x3fffffffLL) return false;
    lo->nmain = (int)nmain;
    lo->nblk = (int)nblk;
    lo->ntiles = (lo->nblk + 3) / 4;
    lo->n_orient = n_orient;
    const long long F = (long long)n_scales * n_orient;
    if (F > 21845) return false;      // D = 3F must fit the uint16 plane indices used on the host side
    lo->F = (int)F;
    lo->D = 3 * lo->F;
    int row = 0, off = 0, h = H, w = W;
    for (int L = 0; L < lo->n_levels; ++L) {
        const int scales = n_scales - 2 * L >= 2 ? 2 : 1;
        lo->FL[L] = scales * n_orient;
        lo->DL[L] = 3 * lo->FL[L];
        lo->row0[L] = row;
        lo->off[L] = off;
        lo->HL[L] = h;
        lo->WL[L] = w;
        row += lo->DL[L];
        off += round_up(lo->DL[L] * (KP_TP >> (2 * L)) * 2, 16);
        h = (h + 1) / 2;
        w = (w + 1) / 2;
    }
    lo->tile_bytes = off;
    lo->img_bytes = (long long)lo->ntiles * lo->tile_bytes;
    if (gcs_split_bank(lo->n_levels, lo->D)) {
        lo->split = 1;
        int sl = 0;
        for (int L = 0; L < lo->n_levels; ++L) {
            lo->sl0[L] = sl;
            sl += round_up(lo->DL[L] * (KP_TP >> (2 * L)), 32);
        }
        lo->S = sl;
        lo->tile_bytes = 2 * sl;
        const long long lo_bytes = (long long)lo->ntiles * sl;
        if (2 * lo_bytes + 4LL * lo->ntiles + 256 > 0xffffffffLL) return false;
        lo->mid_off = (unsigned)lo_bytes;
        lo->top_off = (unsigned)(lo_bytes + lo_bytes / 2);
        lo->flag_off = (unsigned)(2 * lo_bytes);
        lo->img_bytes = 2 * lo_bytes + round_up(4 * lo->ntiles, 256);
    }
    return true;
}

// Pixel (y, x) -> its block and slot. The corner of two packed strips belongs to the right one.
__host__ __device__ __forceinline__ void gcs_locate(const GcsLayout &lo, int y, int x, int &blk, int &iy, int &ix) {
    if (x >= lo.Wm) {
        const int R = y >> 1, q = R & 15;
        blk = lo.nmain + (R >> 4);
        iy = 2 * (q >> 2) + (y & 1);
        ix = 2 * (q & 3) + (x & 1);
    } else if (y >= lo.Hm) {
        const int C = x >> 1, q = C & 15;
        blk = lo.nmain + lo.nR + (C >> 4);
        iy = 2 * (q >> 2) + (y & 1);
        ix = 2 * (q & 3) + (x & 1);
    } else {
        blk = (y >> 3) * lo.bx_n + (x >> 3);
        iy = y & 7;
        ix = x & 7;
    }
}
// Slot (iy, ix) of STRIP block blk (blk >= nmain) -> pixel; `xlim` = the first column that is not this strip's.
__device__ __forceinline__ void gcs_strip_pixel(const GcsLayout &lo, int blk, int iy, int ix, int &y, int &x, int &xlim) {
    const bool right = blk < lo.nmain + lo.nR;
    const int v = blk - lo.nmain - (right ? 0 : lo.nR);
    const int s = 32 * v + 8 * (iy >> 1) + 2 * (ix >> 1);
    // (the fields as VALUES first: hipcc may otherwise fold a select of two kernel-argument loads into ONE vector load from a
    //  selected address - a global_load_dword + s_waitcnt vmcnt(0) inside the tile loop of a Lloyd pass)
    const int Hm = __builtin_amdgcn_readfirstlane(lo.Hm), Wm = __builtin_amdgcn_readfirstlane(lo.Wm);
    const int Wf = __builtin_amdgcn_readfirstlane(lo.W), Wb = __builtin_amdgcn_readfirstlane(lo.Wb);
    y = right ? s + (iy & 1) : Hm + (iy & 1);
    x = right ? Wm + (ix & 1) : s + (ix & 1);
    xlim = right ? Wf : Wb;
}

// physical plane -> (level, plane in level); logical feature <-> physical plane
__host__ __device__ __forceinline__ int gcs_level_of_plane(const GcsLayout &lo, int r) {
    int L = 0;
#pragma unroll
    for (int i = 1; i < GCS_LEVELS_MAX; ++i)
        if (i < lo.n_levels && r >= lo.row0[i]) L = i;
    return L;
}
__host__ __device__ __forceinline__ int gcs_logical_of_plane(const GcsLayout &lo, int r) {
    const int L = gcs_level_of_plane(lo, r);
    const int ri = r - lo.row0[L];
    const int c = ri / lo.FL[L], fl = ri - c * lo.FL[L];
    return c * lo.F + 2 * L * lo.n_orient + fl;
}
__host__ __device__ __forceinline__ int gcs_plane_of_logical(const GcsLayout &lo, int d) {
    const int c = d / lo.F, f = d - c * lo.F;
    const int L = (f / lo.n_orient) >> 1;
    return lo.row0[L] + c * lo.FL[L] + (f - 2 * L * lo.n_orient);
}

// WIDE slab: byte offset (from the slab base) of the value of physical plane r at FULL-resolution pixel (y, x) of image b:
// the level-L parent slot (iy >> L, ix >> L) inside the pixel's block. (Either format: gcs_slab_value.)
__device__ __forceinline__ size_t gcs_slab_offset(const GcsLayout &lo, int b, int r, int y, int x) {
    const int L = gcs_level_of_plane(lo, r);
    int blk, iy, ix;
    gcs_locate(lo, y, x, blk, iy, ix);
    const int side = 8 >> L;                                          // sub-block side at level L
    const int npl = KP_TP >> (2 * L);                                 // pixels per plane of a tile at level L
    return (size_t)b * lo.img_bytes + (size_t)(blk >> 2) * lo.tile_bytes + lo.off[L] +
           ((size_t)(r - lo.row0[L]) * npl + (blk & 3) * side * side + (iy >> L) * side + (ix >> L)) * 2;
}

// ---- split slab: slot index (inside the image: tile * S + slot in tile) of physical plane r at full-resolution pixel (y, x)
__host__ __device__ __forceinline__ unsigned gcs_split_slot(const GcsLayout &lo, int r, int y, int x) {
    const int L = gcs_level_of_plane(lo, r);
    int blk, iy, ix;
    gcs_locate(lo, y, x, blk, iy, ix);
    const int side = 8 >> L, npl = KP_TP >> (2 * L);
    return (unsigned)(blk >> 2) * (unsigned)lo.S + (unsigned)(lo.sl0[L] + (r - lo.row0[L]) * npl + (iy >> L) * 4 * side +
                                                              (blk & 3) * side + (ix >> L));
}
// byte (relative to the MID / TOP array) and bit shift of a slot's nibble: group of g = max(2, 8 >> L) slots, see above
__host__ __device__ __forceinline__ void gcs_split_nibble(int L, unsigned slot, unsigned &byte, int &shift) {
    const unsigned g = (8u >> L) > 2u ? (8u >> L) : 2u, h = g >> 1;
    const unsigned gi = slot % g, second = gi >= h ? 1u : 0u;
    byte = (slot / g) * h + gi - second * h;
    shift = 4 * (int)second;
}
// The value (SPEC.md §3: plain uint16, no offset) of physical plane r at full-resolution pixel (y, x) of image b, either format.
__device__ __forceinline__ unsigned gcs_slab_value(const unsigned char *feats, const GcsLayout &lo, int b, int r, int y, int x) {
    if (!lo.split) return *reinterpret_cast<const uint16_t *>(feats + gcs_slab_offset(lo, b, r, y, x)) ^ 0x8080u;
    const unsigned char *img = feats + (size_t)b * lo.img_bytes;
    const unsigned slot = gcs_split_slot(lo, r, y, x);
    unsigned nb;
    int sh;
    gcs_split_nibble(gcs_level_of_plane(lo, r), slot, nb, sh);
    const unsigned lo8 = img[slot] ^ 0x80u, mid = (img[lo.mid_off + nb] >> sh) & 15u, top = (img[lo.top_off + nb] >> sh) & 15u;
    return lo8 | (mid << 8) | (top << 12);
}

// Partial sums of one Lloyd pass: [set][chunk of 16 elements][row][16] uint64, one row per k-means workgroup (sets = images
// for per-image codebooks with `parts` rows each, one set of nb * parts rows otherwise). A workgroup writes whole 128-byte
// lines (16 consecutive elements of ITS row: no line is shared between workgroups - with the element-major layout of
// round 1 every 8-byte store was a partial-line write and the stores of a pass cost 4 us); the reduce kernel reads a
// chunk's rows as one contiguous block. The padding elements of the last chunk are never written and never used.
constexpr int KP_PCH = 16;
__host__ __device__ __forceinline__ int partial_chunks(int row_len) { return (row_len + KP_PCH - 1) / KP_PCH; }
__device__ __forceinline__ size_t partial_index(int per_image, int b, int part, int parts, int nb, int i, int row_len) {
    const size_t rows = per_image ? (size_t)parts : (size_t)nb * parts;
    const size_t row = per_image ? (size_t)part : (size_t)b * parts + part;
    const size_t set = per_image ? (size_t)b : 0;
    return ((set * partial_chunks(row_len) + (size_t)(i / KP_PCH)) * rows + row) * KP_PCH + (size_t)(i % KP_PCH);
}
