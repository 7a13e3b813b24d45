// kmeans.hip — SPEC.md §4 on gfx950 over the pyramid feature slab (csrc/common.h).
// The reference ships no code for this path; slot: /root/reference/BSD_metrics/script.py:30.
//
// Kernels
//   kmeans_pass_mfma_kernel  one Lloyd pass (assign + update) on the matrix cores for D <= 207; a stream of the
//                            slab, which keeps pyramid level L at 1/4^L of the pixels: a tile's coarse planes are
//                            replicated over their 2^L x 2^L blocks while they are staged into LDS. 4x6-style banks
//                            (at most two levels, D <= 79) read the SPLIT slab: 12 of the 16 bits of every value, the
//                            last 4 for flagged tiles only (template flag SPLIT); with k <= 8 their level 1 stays COMPACT
//                            in LDS (CL1: block rows held as even | odd pixels, no replication).
//   kmeans_pass_native_kernel  the pass for deep banks (BASELINE config 4): every level at its own resolution.
//   kmeans_assign_kernel     generic pass for D >= 208: exact integer argmin via fp32 byte-digit FMAs (all partial
//                            sums < 2^24, hence exact), LDS-replicated u32 accumulators.
//   kmeans_reduce_kernel     element-major partial sums -> int64 sums (+ the centroid update when single-rank).
//   kmeans_finalize / init / features_gather / labels_widen: small helpers.
// Label maps leave every pass in RASTER order ([B][H][W] uint8 or int32): which slot of which block holds a pixel
// (csrc/common.h: main blocks and packed edge strips) is the passes' own business.
// Nothing here allocates, frees or synchronises; every entry point enqueues on the caller's stream.
#include "common.h"
#include <type_traits>

#define LAYOUT_OR_FAIL(lo, who)                                        \
    GcsLayout lo;                                                      \
    if (B <= 0 || !gcs_make_layout(H, W, n_scales, n_orient, &lo))     \
        return gcs_fail(GCS_EINVAL, who ": bad shape or bank")

// ------------------------------------------------------------------------ init / gather
__global__ void kmeans_init_kernel(const unsigned char *__restrict__ feats, GcsLayout lo, int k,
                                   uint16_t *__restrict__ cent) {
    const int set = blockIdx.x; // image index == set index (n_sets == 1 -> image 0)
    const long P = (long)lo.H * lo.W;
    for (int i = threadIdx.x; i < k * lo.D; i += blockDim.x) {
        const int j = i / lo.D, d = i % lo.D;
        const long p = ((2L * j + 1) * P) / (2L * k);
        const int y = (int)(p / lo.W), x = (int)(p % lo.W);
        cent[((size_t)set * k + j) * lo.D + d] = (uint16_t)gcs_slab_value(feats, lo, set, gcs_plane_of_logical(lo, d), y, x);
    }
}

extern "C" int gcs_kmeans_init(const uint16_t *feats, int B, int H, int W, int n_scales, int n_orient, int k,
                               int n_sets, uint16_t *cent, gcs_stream_t stream) {
    if (!feats || !cent) return gcs_fail(GCS_EINVAL, "gcs_kmeans_init: NULL pointer");
    LAYOUT_OR_FAIL(lo, "gcs_kmeans_init");
    if (k < 1 || k > GCS_K_MAX) return gcs_fail(GCS_EINVAL, "gcs_kmeans_init: k must be in 1..16");
    if (n_sets != 1 && n_sets != B) return gcs_fail(GCS_EINVAL, "gcs_kmeans_init: n_sets must be 1 or B");
    hipLaunchKernelGGL(kmeans_init_kernel, dim3(n_sets), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned char *>(feats), lo, k, cent);
    GCS_CHECK_LAUNCH("gcs_kmeans_init");
    return GCS_OK;
}

// out[i][d] = feature d of pixel (b, y, x) = byx[i]; b < 0 gives a zero row. Lets a rank publish the
// SPEC.md §4 init centroids it owns when an image is sharded by rows (BASELINE config 5).
__global__ void features_gather_kernel(const unsigned char *__restrict__ feats, GcsLayout lo, int n,
                                       const int32_t *__restrict__ byx, uint16_t *__restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * lo.D; i += gridDim.x * blockDim.x) {
        const int r = i / lo.D, d = i % lo.D;
        const int b = byx[3 * r], y = byx[3 * r + 1], x = byx[3 * r + 2];
        out[i] = b < 0 ? (uint16_t)0 : (uint16_t)gcs_slab_value(feats, lo, b, gcs_plane_of_logical(lo, d), y, x);
    }
}

extern "C" int gcs_features_gather(const uint16_t *feats, int B, int H, int W, int n_scales, int n_orient, int n,
                                   const int32_t *byx, uint16_t *out, gcs_stream_t stream) {
    if (!feats || !byx || !out) return gcs_fail(GCS_EINVAL, "gcs_features_gather: NULL pointer");
    LAYOUT_OR_FAIL(lo, "gcs_features_gather");
    if (n <= 0) return gcs_fail(GCS_EINVAL, "gcs_features_gather: n must be > 0");
    hipLaunchKernelGGL(features_gather_kernel, dim3((n * lo.D + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned char *>(feats), lo, n, byx, out);
    GCS_CHECK_LAUNCH("gcs_features_gather");
    return GCS_OK;
}

// ---------------------------------------------------------------------------------------
// Generic pass (D >= 208). Exact integer argmin with fp32 digit arithmetic: x = 256*xh + xl, c = 256*ch + cl (bytes);
//   sum_d x*c = 65536*sum xh*ch + 256*sum (xh*cl + xl*ch) + sum xl*cl,
// every partial sum stays below 2^24 over a chunk of <= 128 planes, so fp32 FMA is exact.
// score_j = |c_j|^2 - 2 sum_d x_d c_jd (the |x|^2 term is common to all j). Planes are walked in PHYSICAL order
// (level-major); the centroid digits are permuted to match when they are loaded.
constexpr int KM_CHUNK = 128;

// two horizontally adjacent pixels (x even) of physical plane r: one aligned dword on level 0, the same parent twice above
// (wide slab: feature vectors of 208 or more planes never take the split one)
__device__ __forceinline__ unsigned feature_pair(const unsigned char *feats, const GcsLayout &lo, int b, int r, int y, int x) {
    const unsigned char *p = feats + gcs_slab_offset(lo, b, r, y, x);
    if (r < lo.DL[0]) return *reinterpret_cast<const unsigned *>(p) ^ 0x80808080u;
    const unsigned v = *reinterpret_cast<const uint16_t *>(p) ^ 0x8080u;
    return v | (v << 16);
}

template <int K>
__global__ __launch_bounds__(256) void kmeans_assign_kernel(
    const unsigned char *__restrict__ feats, const uint16_t *__restrict__ cent, GcsLayout lo, int per_image, int parts,
    int R, int row_lo, int row_hi, uint8_t *__restrict__ labels, uint64_t *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve: cdig float [D][K][2] | cnorm int64 [K] | acc u32 [K][D+1][R]
    const int D = lo.D, H = lo.H, W = lo.W;
    float *cdig = reinterpret_cast<float *>(smem);
    long long *cnorm = reinterpret_cast<long long *>(smem + (((size_t)D * K * 2 * 4 + 15) & ~(size_t)15));
    unsigned *acc = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(cnorm) + ((K * 8 + 15) & ~15));

    const int tid = threadIdx.x;
    const int b = blockIdx.y, part = blockIdx.x;
    const uint16_t *cset = cent + (size_t)(per_image ? b : 0) * K * D;
    const int D1 = D + 1;

    for (int i = tid; i < K * D; i += 256) {
        const int j = i / D, r = i % D;                      // r = physical plane
        const unsigned cv = cset[j * D + gcs_logical_of_plane(lo, r)];
        cdig[(r * K + j) * 2 + 0] = (float)(cv & 255u);
        cdig[(r * K + j) * 2 + 1] = (float)(cv >> 8);
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

    const int ppr = (W + 1) >> 1; // pixel pairs per row
    const long npairs = (long)H * ppr;
    const int rep = tid & (R - 1);

    for (long q = (long)part * 256 + tid; q < npairs; q += (long)parts * 256) {
        const int y = (int)(q / ppr), x = 2 * (int)(q % ppr);
        long long S[2][K];
#pragma unroll
        for (int j = 0; j < K; ++j) S[0][j] = S[1][j] = 0;
        for (int d0 = 0; d0 < D; d0 += KM_CHUNK) {
            const int d1 = min(D, d0 + KM_CHUNK);
            float a0[2][K], a1[2][K], a2[2][K];
#pragma unroll
            for (int j = 0; j < K; ++j) a0[0][j] = a0[1][j] = a1[0][j] = a1[1][j] = a2[0][j] = a2[1][j] = 0.f;
            for (int d = d0; d < d1; ++d) {
                const unsigned u = feature_pair(feats, lo, b, d, y, x);
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
        if (labels) {   // uint8 raster map [B][H][W]
            uint8_t *lp = labels + ((size_t)b * H + y) * W + x;
            lp[0] = (uint8_t)lab[0];
            if (x + 1 < W) lp[1] = (uint8_t)lab[1];
        }
        // accumulate (second pass over this thread's planes; L2-resident)
        const bool rows_ok = partials && y >= row_lo && y < row_hi;
        const bool v0 = rows_ok && x < W, v1 = rows_ok && x + 1 < W;
        if (v0) {
            unsigned *a_0 = acc + (size_t)lab[0] * D1 * R + rep;
            unsigned *a_1 = acc + (size_t)lab[1] * D1 * R + rep;
            if (v1 && lab[0] == lab[1]) {
                for (int d = 0; d < D; ++d) {
                    const unsigned u = feature_pair(feats, lo, b, d, y, x);
                    atomicAdd(a_0 + d * R, (u & 0xffffu) + (u >> 16));
                }
                atomicAdd(a_0 + D * R, 2u);
            } else {
                for (int d = 0; d < D; ++d) {
                    const unsigned u = feature_pair(feats, lo, b, d, y, x);
                    atomicAdd(a_0 + d * R, u & 0xffffu);
                    if (v1) atomicAdd(a_1 + d * R, u >> 16);
                }
                atomicAdd(a_0 + D * R, 1u);
                if (v1) atomicAdd(a_1 + D * R, 1u);
            }
        }
    }
    __syncthreads();
    if (!partials) return;
    for (int i = tid; i < K * D1; i += 256) {
        const int j = i / D1, e = i % D1;                   // e = logical feature (or D = count)
        const int pe = e < D ? gcs_plane_of_logical(lo, e) : D;
        uint64_t s = 0;
        for (int rr = 0; rr < R; ++rr) s += acc[((size_t)j * D1 + pe) * R + rr];
        partials[partial_index(per_image, b, part, parts, (int)gridDim.y, i, K * D1)] = s;
    }
}

// ---------------------------------------------------------------------------------------
// One Lloyd pass on the matrix cores (D <= 207: 80-row LDS tile for D <= 79, 208-row tile above).
// Per 256-pixel tile (four 8x8 blocks, one per wave), staged ONCE in LDS as D rows of 256 u16 (each byte offset by
// -128 so it is a signed MFMA digit; rows in PHYSICAL plane order, coarse levels replicated to full resolution):
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

// Tile loads of the passes and the Infinity Cache (256 MiB). Passes sweep the slab in alternating directions, so a pass STARTS on the
// bytes its predecessor read last: those are worth keeping in the cache; everything before them is evicted before anyone returns
// and allocating it only costs. The split-slab pass and the deep-bank pass therefore load the list positions below `nt_limit` with
// the nontemporal hint (buffer_load ... nt) and the last KP_MALL_KEEP_DEFAULT bytes of every sweep plain. Measured, round 6 (same box,
// two interleaved rounds each; tools/dbg/mall_keep_sweep.sh - then an environment hook, now variant builds -, profiles/r6_mall_keep_sweep.txt), whole steps of 64 images:
//   deep-bank pass (8x8 bank, 1 265 MB per pass, HBM-bound): every load plain 0.237 - 0.243 ms per pass, step 3 075 - 3 117 Mpix/s;
//     every load nt 0.220 - 0.223 ms, 3 172; the last 128 / 192 / 256 / 320 / 384 / 512 MB plain: 0.215 - 0.218 / 0.216 - 0.217 /
//     0.214 - 0.219 / 0.214 - 0.217 / 0.215 - 0.216 / 0.219 - 0.220 ms, step 3 193 - 3 202 / 3 216 - 3 228 / 3 213 - 3 252 /
//     3 216 - 3 236 / 3 209 - 3 213 / 3 199 - 3 215 Mpix/s: + 4 % with 192 - 320 MB (5.9 TB/s of algorithmic bytes);
//   split-slab pass (4x6 bank, 671 MB per pass, bound by its dependency chains): every load plain 5 277 - 5 296 Mpix/s, every load
//     nt 5 144 - 5 201 (the next pass no longer finds the end of the sweep in the cache), the last 128 / 192 / 256 / 320 / 384 /
//     512 MB plain: 5 279 - 5 289 / 5 272 - 5 316 / 5 308 - 5 325 / 5 316 - 5 320 / 5 304 - 5 335 / 5 202 - 5 316: + 0.7 % at 256 MB.
//     (Isolated passes of a sequence that interleaves other work, tools/ab.py: every load nt 0.141 - 0.144 against 0.151 - 0.161 ms.)
// History: round 4 introduced the hint as `nt ? __builtin_nontemporal_load(p) : *p` - hipcc merges the two arms into ONE plain
// load (and two branches around global loads likewise): from that commit until round 6 no pass kernel contained an `nt` load
// (ISA), whatever the flag said. The cache policy of a raw buffer load is an immediate operand: two instructions that stay two.
// The wide-slab kernels of kmeans_pass_mfma_kernel (banks outside the BASELINE configurations, -DGCS_NO_SPLIT) load plain.
#ifndef GCS_KP_MALL_KEEP_MB       // (variant builds for same-box sweeps: 0 = every load nt, a huge value = every load plain)
#define GCS_KP_MALL_KEEP_MB 256
#endif
constexpr long long KP_MALL_KEEP_DEFAULT = (long long)GCS_KP_MALL_KEEP_MB << 20;
// list positions (per sweep list: the whole batch, or one image with per-image codebooks) below the result are loaded `nt`
static int kp_nt_limit(const GcsLayout &lo, int B, int n_sets, long long tile_stream_bytes) {
    const long long lists = n_sets == B ? B : 1, nlist = (long long)lo.ntiles * (n_sets == B ? 1 : B);
    const long long keep_tiles = KP_MALL_KEEP_DEFAULT / tile_stream_bytes / lists;
    return (int)(nlist > keep_tiles ? nlist - keep_tiles : 0);
}
// Logical feature of plane `pl` of level LL, and its inverse, with the level a COMPILE-TIME constant: `lo` is a by-value kernel
// argument, and indexing one of its arrays with a per-lane level (gcs_logical_of_plane / gcs_plane_of_logical on a run-time plane)
// makes hipcc fetch the element from the kernarg segment with a VECTOR load and wait for it - four dependent loads and
// s_waitcnt vmcnt(0) in front of every centroid gather and every partial-row store of the round-4 kernels (ISA; stamps:
// profiles/r5_notes.md). The callers unroll over the levels and keep the result of the lane's own level.
template <int LL>
__device__ __forceinline__ int kp_logical_of(const GcsLayout &lo, int pl) {
    const int c = pl / lo.FL[LL];
    return c * lo.F + 2 * LL * lo.n_orient + (pl - c * lo.FL[LL]);
}
// physical plane of logical feature e, or -1 when e is not on level LL (also gives the level: the caller's LL)
template <int LL>
__device__ __forceinline__ int kp_plane_on_level(const GcsLayout &lo, int c, int f) {
    const int fl = f - 2 * LL * lo.n_orient;                  // filter index inside level LL
    return (LL < lo.n_levels && fl >= 0 && fl < lo.FL[LL]) ? lo.row0[LL] + c * lo.FL[LL] + fl : -1;
}
constexpr int KP_PITCH = KP_TP * 2 + 64;  // bytes per plane row: +64 B = 16 banks per row, so the 4 rows x 64 B of a
                                          // tr_b16 half-wave and the 8 rows of a ds_read_b128 lane group hit distinct banks
constexpr int KP_P1 = 128 + 48;           // bytes per COMPACT plane row (kmeans_pass_mfma_kernel, CL1: 64 parents + 48 B: the 4 rows x 2 lane groups
                                          // of a tr_b16 read and the 16 rows of an update read hit distinct 8-byte bank slots; 16-byte aligned)
constexpr int KP_DSTEPS_NARROW = 5;       // D <= 79  (every 4x6 bank): 80 plane rows, 46 KB LDS, 3 workgroups / CU
constexpr int KP_DSTEPS_WIDE = 13;        // D <= 207 (the 8x8 bank, D = 192): 208 plane rows, 120 KB LDS, 1 workgroup / CU

#ifndef GCS_KP_WAVES
#define GCS_KP_WAVES 3
#endif
// Ablation builds of kmeans_pass_mfma_kernel for same-box A/B runs (tools/build_variant.sh x -DGCS_ABL=n, tools/ab.py; results are
// WRONG by construction): bit 0 = no assign phase, bit 1 = no update phase, bit 2 = the split slab's items go to LDS as loaded (no unpack).
#ifndef GCS_ABL
#define GCS_ABL 0
#endif
// DSTEPS = assign K-steps (16 planes = 32 byte-features each); LDS holds ROWS = 16*DSTEPS plane rows (>= D + 1:
// the spare row D is the count row); the update has NT = 2*DSTEPS N-tiles (8 planes = 16 byte-planes each).
// NST = 16-byte staging chunks per thread >= ceil(tile_bytes / 4096); surplus chunks re-copy the tile's last chunk.
// WAVES = 4 (narrow pass: wave w owns block w of the tile) or 8 (wide pass: 131 KB of LDS allow one workgroup per CU, so
// it brings 8 waves: wave w works on block w & 3; in the assign phase it takes the block's 32-pixel half w >> 2, in the
// update phase all 64 pixels for half of the plane tiles -> half the accumulators, twice the waves to hide latency).
// -DGCS_KP_PHASES (debugging aid, tools/dbg/pass_phases.py): every wave of kmeans_pass_mfma_kernel adds up, over its tile loop,
// the shader-clock cycles it spends in each phase of a tile (s_memtime around: staging writes | first barrier | next tile's loads
// | assign | update | second barrier) and stores the six sums behind the loop. Costs ~10 % of the wave's cycles; never in the product.
#ifdef GCS_KP_PHASES
__device__ unsigned long long g_kp_phases[1024 * 4 * 8];
extern "C" int gcs_debug_kp_phases(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kp_phases), sizeof(unsigned long long) * 1024 * 4 * 8);
}
#define KP_PHASE_DECL unsigned long long kp_ph[6] = {0, 0, 0, 0, 0, 0}, kp_t0 = __builtin_amdgcn_s_memtime(), kp_tiles = 0
#define KP_PHASE(k)                                                   \
    do {                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        kp_ph[k] += t_ - kp_t0;                                       \
        kp_t0 = t_;                                                   \
    } while (0)
#define KP_PHASE_STORE                                                                                           \
    do {                                                                                                         \
        const int wg_ = (int)(blockIdx.y * gridDim.x + blockIdx.x);                                              \
        if (lane == 0 && wg_ < 1024) {                                                                           \
            for (int k_ = 0; k_ < 6; ++k_) g_kp_phases[(wg_ * 4 + (wid & 3)) * 8 + k_] = kp_ph[k_];             \
            g_kp_phases[(wg_ * 4 + (wid & 3)) * 8 + 6] = kp_tiles;                                               \
        }                                                                                                        \
    } while (0)
#else
#define KP_PHASE_DECL
#define KP_PHASE(k)
#define KP_PHASE_STORE
#endif

// L0T (CL1 kernel only): plane tiles (16 rows) known to lie wholly inside level 0 - the launcher passes 2 for banks whose level 0 has 32
// planes or more (the 4x6 bank: 36), else 0: see the update's operand reads.
// SPLIT (round 6): the split slab of csrc/common.h (narrow pass only). NST then counts staging ROUNDS: an ITEM = 16 consecutive slots of a
// tile = 16 low bytes + 8 bytes of MID nibbles (+ 8 bytes of TOP nibbles when the tile's flag word says that one of them is set),
// unpacked into the same LDS image as the wide slab's: 16 pixels of a level-0 plane row, or the 4 x 4 parents of one block of a
// level-1 plane, replicated over the block's 64 pixels.
template <int KT, int NST, int DSTEPS, int WAVES, bool SPLIT = false, int L0T = 0>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 8 ? 2
                                          : DSTEPS == KP_DSTEPS_NARROW && KT == 1 && NST <= (SPLIT ? 3 : 6) ? GCS_KP_WAVES
                                          : DSTEPS == KP_DSTEPS_NARROW ? 2 : 1)) void kmeans_pass_mfma_kernel(
    const unsigned char *__restrict__ feats, const uint16_t *__restrict__ cent, GcsLayout lo, int K, int per_image,
    int parts, int reverse, int row_lo, int row_hi, uint64_t *__restrict__ partials, void *__restrict__ raster,
    int raster_u8, int nt_flag) {                      // (nt_flag: the split slab's nt_limit, see lloyd_pass; unused by the wide kernels)
    constexpr int KP_ROWS = 16 * DSTEPS, KP_DSTEPS = DSTEPS, KP_NT = 2 * DSTEPS;
    constexpr int NTHR = 64 * WAVES;                         // threads per workgroup
    constexpr int NT_OWN = WAVES == 8 ? (KP_NT + 1) / 2 : KP_NT;   // update plane tiles a wave accumulates
    // compact copy of pyramid levels >= 2 of one tile (level 1 is replicated straight from the staging registers):
    // at most (D / 2) * 32 bytes plus 16-byte padding per level; sized for the worst case of the bucket
    constexpr int KP_COARSE = DSTEPS == KP_DSTEPS_NARROW ? 40 * 32 + 64 : 104 * 32 + 64;
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[KP_ROWS * KP_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char s_coarse[KP_COARSE];
    __shared__ __attribute__((aligned(16))) unsigned char s_lab[KP_TP];
    __shared__ long long s_const[16];
    constexpr int KP_FLAGS = 320;                            // tiles of one workgroup whose flag is kept (more: read as set, always exact)
    __shared__ unsigned char s_flag[SPLIT ? KP_FLAGS : 4];
    static_assert(!SPLIT || (DSTEPS == KP_DSTEPS_NARROW && WAVES == 4), "the split slab is the narrow pass's");

    // either output may be absent (host contract): raster == NULL on the passes whose assignment nobody reads (every
    // pass but the last), partials == NULL on the last pass, whose sums nobody reads (no update phase, no fold)
    const bool do_acc = partials != nullptr;                 // (raster: see the assign phase)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = tid >> 6;                                // 0 .. WAVES-1
    const int wave = wid & 3;                                // the block of the tile this wave works on
    const int half = wid >> 2;                               // WAVES == 8: which half of the assign / update work
    const int b = blockIdx.y, part = blockIdx.x;
    const int D = lo.D;
    const uint16_t *cset = cent + (size_t)(per_image ? b : 0) * K * D;
    const int ntiles = lo.ntiles;
    // The tile list this workgroup strides through. Per-image codebooks: the tiles of image b, stride `parts`. One
    // global codebook: the tiles of the WHOLE batch as one list, stride B * parts: at any moment the resident workgroups
    // then read one contiguous window of the slab (B * parts tiles, 17 MB) instead of B separate ones, which is what
    // HBM's channel / bank interleave is built for (same box: pass 0.190 -> 0.175 ms; profiles/r2_notes.md).
    const int nimg = per_image ? 1 : (int)gridDim.y;
    const int G = parts * nimg, g = per_image ? part : b * parts + part;
    const int nlist = ntiles * nimg;
    const unsigned char *fb = feats + (size_t)(per_image ? b : 0) * lo.img_bytes;   // first image of the list (wide: each tile one contiguous run)

    // ---- centroids -> LDS scratch (borrowed from the tile buffer): [8*KT clusters][KP_ROWS planes] u16 in PHYSICAL
    //      plane order, stored offset-binary (c ^ 0x8080: low byte = digit cl, high byte = digit ch), zero outside K x D.
    uint16_t *cs = reinterpret_cast<uint16_t *>(s_tile);
    for (int i = tid; i < 8 * KT * KP_ROWS; i += NTHR) {
        const int j = i / KP_ROWS, r = i % KP_ROWS;
        int e = 0;                                               // logical feature of physical plane r (kp_logical_of: levels unrolled)
        if (r >= lo.row0[0] && r < lo.row0[0] + lo.DL[0]) e = kp_logical_of<0>(lo, r - lo.row0[0]);
        if (lo.n_levels > 1 && r >= lo.row0[1] && r < lo.row0[1] + lo.DL[1]) e = kp_logical_of<1>(lo, r - lo.row0[1]);
        if (lo.n_levels > 2 && r >= lo.row0[2] && r < lo.row0[2] + lo.DL[2]) e = kp_logical_of<2>(lo, r - lo.row0[2]);
        if (lo.n_levels > 3 && r >= lo.row0[3] && r < lo.row0[3] + lo.DL[3]) e = kp_logical_of<3>(lo, r - lo.row0[3]);
        cs[i] = (j < K && r < D) ? (uint16_t)(cset[j * D + e] ^ 0x8080u) : (uint16_t)0;
    }
    if constexpr (SPLIT) {
        // the flag words of this workgroup's tiles (csrc/common.h): is any TOP nibble of the tile non-zero?
        for (int it = tid; it < KP_FLAGS; it += NTHR) {
            const long long lt = (long long)g + (long long)it * G;
            unsigned char f = 0;
            if (lt < nlist) {
                const int T = reverse ? nlist - 1 - (int)lt : (int)lt;
                const int bi = T / ntiles, tn = T - bi * ntiles;
                f = *reinterpret_cast<const unsigned *>(fb + (size_t)bi * lo.img_bytes + lo.flag_off + 4 * (size_t)tn) != 0u;
            }
            s_flag[it] = f;
        }
    }
    __syncthreads();
    // ---- per-cluster key base (exact int64): 16 * (|c|^2 - 2*(offset terms of the -128 digits)) + j.
    //      key_j = base_j - 32 R0 - 8192 R1 - 2^21 R2 = 16 * score_j + j, so ONE 64-bit minimum yields the
    //      best score and the lowest index on ties. 16 lanes per cluster, folded with lane shuffles.
    {
      for (int j = tid >> 4; j < 16; j += NTHR / 16) {
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
    // CL1 (round 6, the split narrow pass with k <= 8): level 1 stays COMPACT in LDS - a plane row of level 1 (and every row behind it:
    // the count row, the padding) is 64 parents in (parent row, block, parent column) order at pitch KP_P1, NOT the 2 x 2 replication
    // to 256 pixels: a level-1 item then costs what a level-0 item costs (14 instead of 30 VALU instructions, 2 instead of 8
    // ds_write_b128: the replication was half of a tile's LDS write traffic and made three of the four waves' staging 45 % longer than
    // the fourth's). What makes it possible: the pixels of a block row lie in LDS as (x0 x2 x4 x6 | x1 x3 x5 x7), which is also the
    // column order of the assign MFMA and the K order of the update MFMA. The four consecutive elements a lane's address hands to
    // ds_read_b64_tr_b16 are then four pixels with four DIFFERENT consecutive parents - the parent row itself, read by the lanes of the
    // even pixels, of the odd pixels and of both fine rows alike -, and the 16 bytes = 8 pixels of an update operand are the parent
    // row twice (the same 8 bytes read into both halves of the operand).
    constexpr bool CL1 = SPLIT && KT == 1;
    const int DL0 = lo.DL[0];
    auto row_addr = [&](int r) -> int {                     // byte offset of plane row r in s_tile (CL1: rows >= DL0 are compact)
        return CL1 && r >= DL0 ? DL0 * KP_PITCH + (r - DL0) * KP_P1 : r * KP_PITCH;
    };
    // CL1: the 16-byte chunks of a full row r are swizzled by (r >> 3 & 1) * 32 bytes (chunk bit 1): the update's operand reads take
    // 8 bytes per lane from 16 consecutive rows, and rows r, r + 4, r + 8, r + 12 start on the same banks (the pitch is 16 dwords
    // modulo 64: what the transposed reads want) - four addresses per bank without the swizzle, two with it (tools/design/
    // lds_bank_model.py rules; SQ_LDS_BANK_CONFLICT). The transposed reads (4 consecutive rows of one aligned group of 8) and the staging
    // writes (8 lanes = one row) see a uniform shift.
    auto row_swz = [&](int r) -> int { return CL1 && r < DL0 ? ((r >> 3) & 1) * 32 : 0; };
    if (tid < (CL1 ? 32 : KP_TP / 2)) reinterpret_cast<unsigned *>(&s_tile[row_addr(D)])[tid] = 0x01010101u;   // (a compact row: 64 parents)
    // UPD2 (round 6, the split narrow pass with k <= 8): the update's MFMA rows are (cluster j, byte b), its K slots (pixel, byte) and
    // its columns 16 PLANES - sums[(j, b)][plane] = sel[(j, b)][(px, t)] * X[(px, t)][plane] with sel = the one-hot digit where t == b -,
    // so that the B operand is a plane row AS IT LIES in LDS (8 pixels x (lo, hi) = one 16-byte read, no byte de-interleave: 40 v_perm
    // per tile and wave less) and 80 plane rows are 5 accumulator tiles instead of 10 (the deep-bank pass's form, kmeans_pass_native_kernel).
    constexpr bool UPD2 = SPLIT && KT == 1;
    constexpr int NACC = UPD2 ? DSTEPS : NT_OWN;
    v4i accu[NACC];
#pragma unroll
    for (int nt = 0; nt < NACC; ++nt) accu[nt] = v4i{0, 0, 0, 0};

    // ---- staging (wide slab; the split slab's items: stage_load_split and the SPLIT branch of stage_write below): the tile is ONE
    //      contiguous run of tile_bytes (csrc/common.h), already offset-binary. Chunk
    //      ci = tid + 256*i is 16 bytes at byte 16*ci:
    //        level-0 chunks (the first 32*D_0) are 8 pixels of plane row ci>>5: copied as they are;
    //        level-1 chunks (the next 8*D_1) are 2 rows x 4 pixels of one block's 4x4 parents: each row is replicated
    //          into two fine rows of 8 pixels, i.e. 64 contiguous bytes of the plane row, straight from the registers
    //          (SPEC.md §3: feat[y][x] = g_L[y >> L][x >> L]);
    //        the rest (levels >= 2: deep banks only) goes to s_coarse untouched and is replicated by expand_deep().
    //      Loads and LDS writes are UNCONDITIONAL per wave: a per-chunk guard makes hipcc branch around every
    //      load / write with exec masking and drain vmcnt(0) before each write. Chunks beyond the tile
    //      are clamped to its last chunk: they re-read and re-write it with its own data.
    const int n0 = SPLIT ? 16 * lo.DL[0] : 32 * lo.DL[0];   // level-0 chunks (split slab: level-0 items)
    const int n1 = lo.n_levels > 1 ? 8 * lo.DL[1] : 0; // level-1 chunks
    const int nchunk = SPLIT ? lo.S >> 4 : lo.tile_bytes >> 4;   // (split slab: items per tile)
    v4i st[NST];
    v2i sm[NST], stt[NST];                             // split slab: MID and TOP nibbles of the item
#pragma unroll
    for (int i = 0; i < NST; ++i) stt[i] = v2i{0, 0};
    int sdst[NST], ssrc[NST];
    int scls[NST];                                     // wave-uniform: 0 = every lane copies, 1 = every lane replicates, 2 = mixed
    bool sl1[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int ci = min(tid + NTHR * i, nchunk - 1);
        ssrc[i] = ci;
        const int c1 = ci - n0;
        const bool l1 = SPLIT ? c1 >= 0 : c1 >= 0 && c1 < n1;
        sl1[i] = l1;
        if constexpr (SPLIT)
            // slots of a plane come in (row, block, column) order (csrc/common.h). Level-0 item: row (ci & 15) >> 1 of blocks
            // 2 (ci & 1), 2 (ci & 1) + 1 of plane ci >> 4: two 16-byte pieces 128 bytes apart. Level-1 item: parent row c1 & 3 of
            // the four blocks of plane c1 >> 2: per block 4 parents = fine rows 2 p, 2 p + 1 = 32 bytes at q * 128 + 32 p.
            // (CL1: the item is parent row c1 & 3 of the plane's compact row - 16 parents = 32 bytes, blocks 0, 1 | blocks 2, 3)
            sdst[i] = l1 ? (CL1 ? (int)(size_t)&s_tile[row_addr(lo.row0[1] + (c1 >> 2)) + (c1 & 3) * 32]
                                : (int)(size_t)&s_tile[(lo.row0[1] + (c1 >> 2)) * KP_PITCH + (c1 & 3) * 32])
                         : (int)(size_t)&s_tile[(ci >> 4) * KP_PITCH + (((ci & 1) * 256 + ((ci & 15) >> 1) * 16) ^ row_swz(ci >> 4))];
        else
            sdst[i] = ci < n0 ? (int)(size_t)&s_tile[(ci >> 5) * KP_PITCH + (ci & 31) * 16]
                      : l1    ? (int)(size_t)&s_tile[(lo.row0[1] + (c1 >> 3)) * KP_PITCH + (c1 & 7) * 64]
                              : (int)(size_t)&s_coarse[(c1 - n1) * 16];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(l1);
        scls[i] = m == 0ull ? 0 : m == ~0ull ? 1 : 2;
    }
    auto stage_load = [&](int tile) {
        const v4i *src = reinterpret_cast<const v4i *>(fb + (size_t)tile * lo.tile_bytes);
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = src[ssrc[i]];
    };
    // split slab: the tile whose LO run starts at p_lo and whose MID run at p_mid (uniform pointers; 32-bit lane offsets: the loads
    // take an SGPR base and need no vector address arithmetic); the TOP run only when the tile's flag word is set
    // Raw buffer loads (one descriptor over the tile's LO run; the MID and TOP runs at scalar offsets from it): the cache policy is an
    // IMMEDIATE of the intrinsic, so the two forms - plain, and `nt` for the part of the sweep that no later pass finds in the Infinity
    // Cache (see lloyd_pass) - are different instructions. Written as `nt ? __builtin_nontemporal_load(p) : *p`, or as two branches
    // around global loads, hipcc merges them into ONE plain load: until round 6 not a single `nt` load was left in the pass kernels
    // (ISA). The address is SGPR descriptor + 32-bit lane offset + SGPR offset: no vector address arithmetic (hipcc built 64-bit lane
    // addresses for the global loads: a v_lshl_add_u64 per load and 18 VGPRs of offsets).
    auto stage_load_split = [&](const unsigned char *p_lo, unsigned mid_rel, bool top, bool nt) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(p_lo), 0, -1, 0x00020000);
        const unsigned top_rel = mid_rel + (unsigned)(lo.top_off - lo.mid_off);
        auto go = [&](auto aux_c) {
            constexpr int AUX = decltype(aux_c)::value;          // gfx940+: bit 1 = nt
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const unsigned o = (unsigned)ssrc[i] * 16u;
                st[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, AUX));
                sm[i] = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(o >> 1), (int)mid_rel, AUX));
            }
            if (top) {
#pragma unroll
                for (int i = 0; i < NST; ++i)
                    stt[i] = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((unsigned)ssrc[i] * 8u), (int)top_rel, AUX));
            } else {
                // no TOP run: stage_write's unpack does not read stt then. The registers are given a fresh (undefined) value so that
                // no old value lives through this path: with one, hipcc merged the paths through copies of the registers just
                // loaded - behind s_waitcnt vmcnt(0), i.e. every tile waited for its successor's loads at once (ISA, round 6)
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    v2i u;
                    asm("" : "=v"(u));
                    stt[i] = u;
                }
            }
        };
        if (nt) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 0>{});
    };
    typedef __attribute__((address_space(3))) v4i *lds_v4i_ptr;
    // (wide slab) a level-1 chunk of the LDS image: coarse row 0 = pixels (v0.lo, v0.hi, v1.lo, v1.hi), row 1 = (v2.., v3..): each
    // pixel twice, each row into two fine rows = four 16-byte pieces of a 64-byte region. Lanes are 64 bytes apart, so piece k
    // of lanes l and l+2 would share banks (4-way conflicts: SQ_LDS_BANK_CONFLICT 0.4 M -> 17.7 M cycles per launch when every
    // lane wrote its pieces in the same order). Lanes therefore start on different pieces: bit 2 of the lane picks which coarse
    // row goes first, bit 1 which of its two pieces; the eight lanes of a ds_write_b128 group then cover 32 distinct banks.
    // split slab: the high bytes (XOR 0x80) of the first and the second half of a nibble group from its MID and TOP dwords
    // (`top_regs`: the tile in the staging registers brought its TOP nibbles - uniform; without them three instructions do)
    bool top_regs = false;
    // (TOPP: a compile-time copy of top_regs - ONE branch per tile around two forms of stage_write; tested inside split_hi it
    //  became four scalar branches per nibble group)
    auto stage_write_as = [&](auto topp) {
    constexpr bool TOPP = decltype(topp)::value;
    auto split_hi = [&](unsigned mid, unsigned top, unsigned &e, unsigned &o) {
        if constexpr (TOPP) {
            e = ((mid & 0x0f0f0f0fu) | ((top << 4) & 0xf0f0f0f0u)) ^ 0x80808080u;
            o = (((mid >> 4) & 0x0f0f0f0fu) | (top & 0xf0f0f0f0u)) ^ 0x80808080u;
        } else {
            e = (mid & 0x0f0f0f0fu) | 0x80808080u;
            o = ((mid >> 4) & 0x0f0f0f0fu) | 0x80808080u;
        }
    };
    {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const v4i v = st[i];
            if constexpr (SPLIT) {
                const v2i m = sm[i], t = stt[i];
                auto level0 = [&]() {
                    // two groups of 8 pixels (the row of two neighbouring blocks): low bytes (a, b), high bytes e (pixels 0..3)
                    // and o (pixels 4..7) -> u16 pairs
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        unsigned e, o;
                        split_hi((unsigned)m[g], (unsigned)t[g], e, o);
                        const unsigned a = (unsigned)v[2 * g], bb = (unsigned)v[2 * g + 1];
                        v4i w;
                        if constexpr (CL1) {                                    // (x0 x2 | x4 x6 | x1 x3 | x5 x7)
                            w[0] = (int)__builtin_amdgcn_perm(e, a, 0x06020400u);
                            w[1] = (int)__builtin_amdgcn_perm(o, bb, 0x06020400u);
                            w[2] = (int)__builtin_amdgcn_perm(e, a, 0x07030501u);
                            w[3] = (int)__builtin_amdgcn_perm(o, bb, 0x07030501u);
                        } else {
                            w[0] = (int)__builtin_amdgcn_perm(e, a, 0x05010400u);
                            w[1] = (int)__builtin_amdgcn_perm(e, a, 0x07030602u);
                            w[2] = (int)__builtin_amdgcn_perm(o, bb, 0x05010400u);
                            w[3] = (int)__builtin_amdgcn_perm(o, bb, 0x07030602u);
                        }
                        *reinterpret_cast<lds_v4i_ptr>(sdst[i] + 128 * g) = w;
                    }
                };
                auto level1 = [&]() {
                    // one parent row of the tile's four blocks: block q's four parents are v[q] (low bytes) and two bytes of m / t
                    // (a nibble group = one parent row of one block); every parent twice, into the fine rows 2 p and 2 p + 1.
                    // Eight consecutive lanes are the four parent rows of two planes (576 bytes apart = 64 modulo 128): the
                    // odd plane's lanes write their two identical pieces in the other order, so that a ds_write_b128 group
                    // covers eight distinct 16-byte columns.
                    const int e16 = ((lane >> 2) & 1) * 16;
                    if constexpr (CL1) {
                        // compact: the four parents of a block as they come - two blocks = one 16-byte store, no replication
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            unsigned e, o;                               // e = (A p0, A p1, B p0, B p1), o = (A p2, A p3, B p2, B p3)
                            split_hi((unsigned)m[j], (unsigned)t[j], e, o);
                            const unsigned a0 = (unsigned)v[2 * j], a1 = (unsigned)v[2 * j + 1];
                            v4i w;
                            w[0] = (int)__builtin_amdgcn_perm(e, a0, 0x05010400u);   // A: parents 0, 1
                            w[1] = (int)__builtin_amdgcn_perm(o, a0, 0x05030402u);   //    parents 2, 3
                            w[2] = (int)__builtin_amdgcn_perm(e, a1, 0x07010600u);   // B
                            w[3] = (int)__builtin_amdgcn_perm(o, a1, 0x07030602u);
                            *reinterpret_cast<lds_v4i_ptr>(sdst[i] + 16 * j) = w;
                        }
                        return;
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        unsigned e, o;                                   // e = (A p0, A p1, B p0, B p1), o = (A p2, A p3, B p2, B p3)
                        split_hi((unsigned)m[j], (unsigned)t[j], e, o);
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            const unsigned a = (unsigned)v[2 * j + qq];
                            const unsigned w0 = __builtin_amdgcn_perm(e, a, qq ? 0x07010600u : 0x05010400u);   // parents 0, 1
                            const unsigned w1 = __builtin_amdgcn_perm(o, a, qq ? 0x07030602u : 0x05030402u);   // parents 2, 3
                            v4i w;
                            w[0] = (int)__builtin_amdgcn_perm(0u, w0, 0x01000100u);
                            w[1] = (int)__builtin_amdgcn_perm(0u, w0, 0x03020302u);
                            w[2] = (int)__builtin_amdgcn_perm(0u, w1, 0x01000100u);
                            w[3] = (int)__builtin_amdgcn_perm(0u, w1, 0x03020302u);
                            const int d = sdst[i] + 128 * (2 * j + qq);
                            *reinterpret_cast<lds_v4i_ptr>(d + e16) = w;
                            *reinterpret_cast<lds_v4i_ptr>(d + 16 - e16) = w;
                        }
                    }
                };
                if (GCS_ABL & 4) {
                    *reinterpret_cast<lds_v4i_ptr>(sdst[i]) = v;
                    *reinterpret_cast<lds_v4i_ptr>(sdst[i] + 128) = v4i{m[0], m[1], t[0], t[1]};
                } else
                if (scls[i] == 0) level0();
                else if (scls[i] == 1) level1();
                else if (sl1[i]) level1();
                else level0();
            } else if (scls[i] == 0) {
                *reinterpret_cast<lds_v4i_ptr>(sdst[i]) = v;
            } else {
                // (the wide slab's level-1 chunk, in the code shape the wide kernels' register allocation was tuned with)
                const bool f = (lane >> 2) & 1;
                const unsigned x0 = f ? (unsigned)v[2] : (unsigned)v[0], x1 = f ? (unsigned)v[3] : (unsigned)v[1];
                const unsigned y0 = f ? (unsigned)v[0] : (unsigned)v[2], y1 = f ? (unsigned)v[1] : (unsigned)v[3];
                v4i ra, rb;
                ra[0] = (int)__builtin_amdgcn_perm(0u, x0, 0x01000100u);
                ra[1] = (int)__builtin_amdgcn_perm(0u, x0, 0x03020302u);
                ra[2] = (int)__builtin_amdgcn_perm(0u, x1, 0x01000100u);
                ra[3] = (int)__builtin_amdgcn_perm(0u, x1, 0x03020302u);
                rb[0] = (int)__builtin_amdgcn_perm(0u, y0, 0x01000100u);
                rb[1] = (int)__builtin_amdgcn_perm(0u, y0, 0x03020302u);
                rb[2] = (int)__builtin_amdgcn_perm(0u, y1, 0x01000100u);
                rb[3] = (int)__builtin_amdgcn_perm(0u, y1, 0x03020302u);
                if (scls[i] == 1 || sl1[i]) {
                    const int e16 = ((lane >> 1) & 1) * 16, f32 = f ? 32 : 0;
                    *reinterpret_cast<lds_v4i_ptr>(sdst[i] + f32 + e16) = ra;
                    *reinterpret_cast<lds_v4i_ptr>(sdst[i] + f32 + 16 - e16) = ra;
                    *reinterpret_cast<lds_v4i_ptr>(sdst[i] + 32 - f32 + e16) = rb;
                    *reinterpret_cast<lds_v4i_ptr>(sdst[i] + 32 - f32 + 16 - e16) = rb;
                } else {                                  // a lane of a mixed wave whose chunk is not level 1
                    *reinterpret_cast<lds_v4i_ptr>(sdst[i]) = v;
                }
            }
        }
    }
    };
    auto stage_write = [&]() {
        if (SPLIT && top_regs) stage_write_as(std::true_type{});
        else stage_write_as(std::false_type{});
    };
    // Levels >= 2 (deep banks): every group of 8 consecutive pixels of a plane row (one row of one 8x8 block) is the
    // replication of 8 >> L level-L pixels of the compact copy in s_coarse.
    auto expand_deep = [&]() {
        for (int L = 2; L < lo.n_levels; ++L) {
            const int side = 8 >> L;                              // level-L pixels per block side
            const unsigned char *srcL = s_coarse + (lo.off[L] - lo.off[2]);
            const int items = lo.DL[L] * 32;                      // (plane, block in tile, fine row)
            for (int it = tid; it < items; it += NTHR) {
                const int rr = it >> 5, grp = it & 31;
                const int blkq = grp >> 3, iy = grp & 7;
                const unsigned char *s = srcL + (((rr * 4 + blkq) * side + (iy >> L)) * side) * 2;
                v4i o;
                if (L == 2) {
                    const unsigned v = *reinterpret_cast<const unsigned *>(s);  // 2 pixels
                    o[0] = o[1] = (int)__builtin_amdgcn_perm(0u, v, 0x01000100u);
                    o[2] = o[3] = (int)__builtin_amdgcn_perm(0u, v, 0x03020302u);
                } else {
                    const unsigned v = *reinterpret_cast<const uint16_t *>(s);  // 1 pixel
                    o[0] = o[1] = o[2] = o[3] = (int)(v | (v << 16));
                }
                *reinterpret_cast<v4i *>(&s_tile[(lo.row0[L] + rr) * KP_PITCH + (blkq * 64 + iy * 8) * 2]) = o;
            }
        }
    };

    const int un = lane & 15, ug = lane >> 4;             // update operand coordinates
    // CL1: LDS addresses of the assign's transposed reads (K-step kk, read rd: plane row 16 kk + 8 h + (i16 >> 2) + 4 rd, first
    // sub-tile) and of the update's operand reads (plane tile pt: row 16 pt + un; first half): a full row holds the lane's pixels at
    // their place in the tile, a compact row the parent row of the lane's two fine rows (of its fine row: update)
    unsigned a_tr[KP_DSTEPS][2], a_up[KP_DSTEPS][2];
    if constexpr (CL1) {
        const int i16 = lane & 15, pxblk = (lane >> 4) & 1, hh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < KP_DSTEPS; ++kk)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int r = 16 * kk + 8 * hh + (i16 >> 2) + 4 * rd;
                const int off = r < DL0 ? ((wave * 64 + 16 * pxblk + 4 * (i16 & 3)) * 2) ^ row_swz(r) : pxblk * 32 + wave * 8;
                a_tr[kk][rd] = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)&s_tile[row_addr(r) + off];
            }
#pragma unroll
        for (int pt = 0; pt < KP_DSTEPS; ++pt) {
            const int r = 16 * pt + un;
            const bool full = r < DL0;
            const int off = full ? ((wave * 64 + 8 * ug) * 2) ^ row_swz(r) : (ug >> 1) * 32 + wave * 8;
            a_up[pt][0] = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)&s_tile[row_addr(r) + off];
            a_up[pt][1] = a_up[pt][0] + (full ? 8u : 0u);     // second half of the operand: the next 4 pixels, or the parent row again
        }
    }
    const unsigned usel = (un & 1) ? 0x07050301u : 0x06040200u;
    const unsigned eqr = (unsigned)un * 0x01010101u;
    const int cnt_bp = 2 * D;

    // Sweep order: workgroup g takes list positions g, g+G, ...; on odd passes the physical order is reversed
    // (boustrophedon), so a pass starts on the tiles the previous pass read last, i.e. on what is still in the 256 MiB
    // Infinity Cache.
    auto phys = [&](int lt) { return reverse ? nlist - 1 - lt : lt; };
    int ltile = g;
    if constexpr (!SPLIT) {
        if (ltile < nlist) stage_load(phys(ltile));
    }
    // this wave's block (one 8x8 block per wave) as (block row, block column) inside ITS image, advanced without a
    // division: which pixels exist and vote is decided from it. A step moves the tile index inside the image by
    // s1 = G mod ntiles, or by s1 - ntiles when that runs past the image's last tile (global list only).
    const int s1 = __builtin_amdgcn_readfirstlane(G % ntiles);
    const int q1 = __builtin_amdgcn_readfirstlane(4 * s1 / lo.bx_n), r1 = 4 * s1 - q1 * lo.bx_n;
    const int q2 = __builtin_amdgcn_readfirstlane(4 * (ntiles - s1) / lo.bx_n), r2 = 4 * (ntiles - s1) - q2 * lo.bx_n;
    int tin = __builtin_amdgcn_readfirstlane(phys(g < nlist ? g : 0) % ntiles);   // tile index inside its image
    int by, bx;
    {
        const int blk0 = 4 * tin + wave;
        by = blk0 / lo.bx_n;
        bx = blk0 - by * lo.bx_n;
    }
    if constexpr (SPLIT) {                 // wave-uniform: in SGPRs, advanced on the scalar unit (the wide kernels' register
        by = __builtin_amdgcn_readfirstlane(by);   // allocation was tuned with them in VGPRs: left alone)
        bx = __builtin_amdgcn_readfirstlane(bx);
    }
    // split slab: the tile that is LOADED next - one step ahead of `tin` - as two running pointers (its LO and MID runs) and its
    // tile index inside its image, advanced like `tin`: s1 tiles on / back modulo the image, qG (+ 1 on a wrap) images on / back -
    // one of two precomputed 64-bit strides per pointer -, and the iteration whose flag that load needs
    const int qG = __builtin_amdgcn_readfirstlane(G / ntiles);
    int tin_l = tin, it_l = 0;
    const unsigned char *p_lo_l = nullptr;
    unsigned mid_rel_l = 0;                                  // the tile's MID run, in bytes from its LO run (< 2^32: host check)
    long long d_lo[2] = {0, 0};                              // [wrap]
    int d_rel[2] = {0, 0};
    if constexpr (SPLIT) {
        const int bi0 = __builtin_amdgcn_readfirstlane(phys(g < nlist ? g : 0) / ntiles);
        const unsigned char *img = fb + (size_t)bi0 * lo.img_bytes;
        p_lo_l = img + (size_t)tin * lo.S;
        mid_rel_l = (unsigned)(lo.mid_off - (size_t)tin * (lo.S >> 1));      // mid_off + tin S / 2 - tin S
        const long long sg = reverse ? -1 : 1;
        d_lo[0] = sg * ((long long)qG * lo.img_bytes + (long long)s1 * lo.S);
        d_lo[1] = sg * ((long long)(qG + 1) * lo.img_bytes + (long long)(s1 - ntiles) * lo.S);
        d_rel[0] = (int)(sg * -(long long)s1 * (lo.S >> 1));
        d_rel[1] = (int)(sg * -(long long)(s1 - ntiles) * (lo.S >> 1));
    }
    auto tile_has_top = [&](int it) -> bool {
        return __builtin_amdgcn_readfirstlane(it < KP_FLAGS ? (int)s_flag[it < KP_FLAGS ? it : 0] : 1) != 0;
    };
    const int nt_limit = __builtin_amdgcn_readfirstlane(nt_flag);   // split slab: list positions below it are loaded `nt`
    auto load_next_split = [&](bool top, int pos) {
        top_regs = top;
        stage_load_split(p_lo_l, mid_rel_l, top_regs, pos < nt_limit);
        const int tn = reverse ? tin_l - s1 : tin_l + s1;
        const bool wrap = reverse ? tn < 0 : tn >= ntiles;
        tin_l = wrap ? (reverse ? tn + ntiles : tn - ntiles) : tn;
        p_lo_l += wrap ? d_lo[1] : d_lo[0];
        mid_rel_l += (unsigned)(wrap ? d_rel[1] : d_rel[0]);
        ++it_l;
    };
    if constexpr (SPLIT) {
        if (ltile < nlist) load_next_split(tile_has_top(0), ltile);
    }
    // CL1: the lane's four key bases live in registers (the kernel has them to spare since the compact level 1; the wide kernels, at
    // their 168, read them from LDS in every sub-tile)
    long long kbase[KT][4];
    if constexpr (CL1) {
#pragma unroll
        for (int mt = 0; mt < KT; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) kbase[mt][g4] = s_const[8 * mt + 2 * g4 + (lane >> 5)];
    }
    KP_PHASE_DECL;
    for (; ltile < nlist; ltile += G) {
        const int tile = phys(ltile);
        // (split slab) the flag of the tile loaded in this iteration: read from LDS here, used behind the barrier
        int flag_next = 1;
        if constexpr (SPLIT) flag_next = it_l < KP_FLAGS ? (int)s_flag[it_l < KP_FLAGS ? it_l : 0] : 1;
        stage_write();
        KP_PHASE(0);
        __syncthreads();
        KP_PHASE(1);
        // the next tile's loads go out first: a wave issuing them outranks the waves of the other workgroups that
        // are in their compute phase (18 interleaved A/B runs: 0.252 -> 0.244 ms per pass)
        __builtin_amdgcn_s_setprio(3);
        if (ltile + G < nlist) {                              // in flight during the MFMAs
            if constexpr (SPLIT) load_next_split(__builtin_amdgcn_readfirstlane(flag_next) != 0, ltile + G);
            else stage_load(phys(ltile + G));
        }
        __builtin_amdgcn_s_setprio(0);
        KP_PHASE(2);
        if (!SPLIT && lo.n_levels > 2) {
            expand_deep();
            __syncthreads();
        }

        const int blk = 4 * tin + wave;                          // block index inside the image
        // (CL1; by, bx are scalars) a main block that lies wholly inside the image and the voting rows, on a pass that writes no
        // label map: the assign epilogue then skips the per-pixel existence tests (12 of its 44 vector instructions per sub-tile)
        const bool full_blk = CL1 && !raster && blk < lo.nmain && 8 * bx + 8 <= lo.W && 8 * by >= row_lo &&
                              8 * by + 8 <= (row_hi < lo.H ? row_hi : lo.H);
        // -------- assign: two 32-pixel sub-tiles per wave (rows 4*sub .. 4*sub+3 of the block)
#pragma unroll
        for (int sub_i = 0; sub_i < ((GCS_ABL & 1) ? 0 : WAVES == 8 ? 1 : 2); ++sub_i) {
            const int sub = WAVES == 8 ? half : sub_i;
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
                v2i fa[KP_DSTEPS], fbv[KP_DSTEPS];
                if constexpr (CL1) {
                    // rows of either kind (a_tr: one address per K-step and read, set up before the tile loop); the second 32-pixel
                    // sub-tile is 64 bytes further in BOTH: 32 pixels of a full row, two parent rows of a compact one
#pragma unroll
                    for (int kk = 0; kk < KP_DSTEPS; ++kk)
                        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c4\n\t"
                                     "ds_read_b64_tr_b16 %1, %3 offset:%c4"
                                     : "=&v"(fa[kk]), "=&v"(fbv[kk])
                                     : "v"(a_tr[kk][0]), "v"(a_tr[kk][1]), "i"(sub * 64)
                                     : "memory");
                } else
#pragma unroll
                for (int kk = 0; kk < KP_DSTEPS; ++kk)       // the DS offset field holds 16 bits: K-step base in the VGPR
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\t"
                                 "ds_read_b64_tr_b16 %1, %2 offset:%c3"
                                 : "=&v"(fa[kk]), "=&v"(fbv[kk])
                                 : "v"(addr + kk * 16 * KP_PITCH), "i"(4 * KP_PITCH)
                                 : "memory");
                // hipcc does not count asm loads: one explicit wait, then tie every destination register to it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < KP_DSTEPS; ++kk) {
                    asm volatile("" : "+v"(fa[kk]), "+v"(fbv[kk]));
                    bfr[kk] = v4i{fa[kk][0], fa[kk][1], fbv[kk][0], fbv[kk][1]};
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
                    long long key = mad_i64_i32(u, -32, CL1 ? kbase[mt][g] : s_const[8 * mt + 2 * g + h]);   // base: registers, or an LDS broadcast read
                    key = mad_i64_i32(acc[mt][4 * g + 2], -2097152, key);
                    best = key < best ? key : best;
                }
            // partner half's key by v_permlane32_swap (VALU; no LDS round trip like ds_bpermute)
            const unsigned blo = (unsigned)best, bhi = (unsigned)((unsigned long long)best >> 32);
            const auto s0 = __builtin_amdgcn_permlane32_swap(blo, blo, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(bhi, bhi, false, false);
            // after swap(x, x): element 0 holds the LOWER half's x in both halves, element 1 the UPPER half's: the minimum of the two
            // pairs is the pixel's best key in every lane, with no select by half (and only its low word is needed: the label)
            const long long ka = (long long)(((unsigned long long)s1[0] << 32) | s0[0]);
            const long long kb = (long long)(((unsigned long long)s1[1] << 32) | s0[1]);
            const int bj = (int)((kb < ka ? s0[1] : s0[0]) & 15);
            if (h == 0 && CL1 && full_blk) {
                // (wave-uniform) every pixel of the block exists and votes, and no label map is asked for: nothing to decide
                s_lab[pl] = (unsigned char)bj;
            } else
            if (h == 0) {
                // which pixel this slot holds (csrc/common.h): a main block's, or - rarely - an edge strip's
                // (CL1: MFMA column n is the pixel at place n of the block row order (x0 x2 x4 x6 | x1 x3 x5 x7))
                const auto col_of = [&](int nn) { return CL1 ? 2 * (nn & 3) + ((nn >> 2) & 1) : nn & 7; };
                int y = 8 * by + 4 * sub + (n >> 3), x = 8 * bx + col_of(n), xlim = lo.W;
                if (blk >= lo.nmain) {           // 26 of the 2 426 blocks of a BSD image
                    // the slot coordinates are re-derived from an opaque copy of the lane number: derived from `n` they are
                    // loop invariants, hipcc keeps them in VGPRs across the tile loop and the pass (168 VGPRs for three
                    // workgroups per CU) spills
                    int no = n;
                    asm volatile("" : "+v"(no));
                    gcs_strip_pixel(lo, blk, 4 * sub + (no >> 3), col_of(no), y, x, xlim);
                }
                const bool inimg = blk < lo.nblk && y < lo.H && x < xlim;
                const bool valid = inimg && y >= row_lo && y < row_hi;      // votes in the sums (halo rows do not)
                s_lab[pl] = valid ? (unsigned char)bj : (unsigned char)0xFF;
                // the label map itself, raster order [B][H][W] (the last pass): in a main block eight lanes cover one row
                // of the block, 32 (int32) or 8 (uint8) contiguous bytes
                if (raster && inimg) {
                    const size_t o = ((size_t)(per_image ? b : tile / ntiles) * lo.H + y) * lo.W + x;
                    if (raster_u8) static_cast<uint8_t *>(raster)[o] = (uint8_t)bj;
                    else static_cast<int32_t *>(raster)[o] = bj;
                }
            }
        }
        KP_PHASE(3);
        if (WAVES == 8 && do_acc) __syncthreads();             // the block's labels come from two waves
        // -------- update: one-hot MFMA over the block's 64 pixels
        if constexpr (UPD2) {
          if (do_acc && !(GCS_ABL & 2)) {
            // lane (row r = un = (j, b), K group ug): pixels 8 ug .. 8 ug + 7 of the block's 32-pixel half hf
            const unsigned eqj = (unsigned)(un >> 1) * 0x01010101u;
            const unsigned sel01 = (un & 1) ? 0x010c000cu : 0x0c010c00u, sel23 = (un & 1) ? 0x030c020cu : 0x0c030c02u;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const v2i lw = *reinterpret_cast<const v2i *>(&s_lab[wave * 64 + hf * 32 + 8 * ug]);
                v4i oh;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned x = (unsigned)lw[i] ^ eqj;                    // byte == 0 <=> label == j
                    const unsigned y = (x | 0x80808080u) - 0x01010101u;        // top bit clear <=> byte == 0
                    const unsigned d = ~y & 0x80808080u;                        // digit -128 where label == j
                    oh[2 * i] = (int)__builtin_amdgcn_perm(0u, d, sel01);       // (px, t): the digit where t == b, 0 elsewhere
                    oh[2 * i + 1] = (int)__builtin_amdgcn_perm(0u, d, sel23);
                }
                // The five plane tiles' operand reads go out TOGETHER, each MFMA waits for its own (counted lgkmcnt: LDS returns in
                // order; whatever else is in flight only makes a wait longer). Left to itself hipcc reads, waits and multiplies
                // tile by tile - five exposed LDS latencies per half - whatever the source order and however many registers are
                // free (profiles/r6_notes.md); asm loads are invisible to its wait counting, hence the explicit waits.
                static_assert(DSTEPS == 5, "the update's read batch is written out for five plane tiles");
                // (CL1: two 8-byte reads per plane tile - the halves of a full row's 16 bytes, or a compact row's parent row twice. The
                //  first L0T plane tiles are known to hold full rows only: ONE 16-byte read each, conflict-free with the rows' swizzle
                //  where the 8-byte reads of 16 consecutive rows cannot do better than two addresses per bank)
                v4i bq[DSTEPS];
                v2i bl[DSTEPS], bh[DSTEPS];
#pragma unroll
                for (int pt = 0; pt < DSTEPS; ++pt) {
                    if (pt < L0T)
                        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(bq[pt]) : "v"(a_up[pt][0]), "i"(hf * 64) : "memory");
                    else
                        asm volatile("ds_read_b64 %0, %2 offset:%c4\n\t"
                                     "ds_read_b64 %1, %3 offset:%c4"
                                     : "=&v"(bl[pt]), "=&v"(bh[pt])
                                     : "v"(a_up[pt][0]), "v"(a_up[pt][1]), "i"(hf * 64)
                                     : "memory");
                }
                // reads issued behind plane tile pt's: one per later tile below L0T, two per later tile from L0T on
#define KP_UPD_YOUNGER(pt_) (((pt_) + 1 < L0T ? L0T - 1 - (pt_) : 0) + 2 * (DSTEPS - ((pt_) + 1 < L0T ? L0T : (pt_) + 1)))
#define KP_UPD_STEP(pt_)                                                                                  \
    do {                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KP_UPD_YOUNGER(pt_)) : "memory");                       \
        if ((pt_) < L0T) {                                                                                \
            asm volatile("" : "+v"(bq[pt_]));                                                             \
        } else {                                                                                          \
            asm volatile("" : "+v"(bl[pt_]), "+v"(bh[pt_]));                                              \
            bq[pt_] = v4i{bl[pt_][0], bl[pt_][1], bh[pt_][0], bh[pt_][1]};                                \
        }                                                                                                 \
        accu[pt_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(oh, bq[pt_], accu[pt_], 0, 0, 0);               \
    } while (0)
                KP_UPD_STEP(0);
                KP_UPD_STEP(1);
                KP_UPD_STEP(2);
                KP_UPD_STEP(3);
                KP_UPD_STEP(4);
#undef KP_UPD_STEP
#undef KP_UPD_YOUNGER
            }
          }
        } else
        if (do_acc && !(GCS_ABL & 2)) {
            const v4i lw = *reinterpret_cast<const v4i *>(&s_lab[wave * 64 + 16 * ug]);
            v4i oh;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned x = (unsigned)lw[i] ^ eqr;                    // byte == 0 <=> label == un
                const unsigned y = (x | 0x80808080u) - 0x01010101u;        // top bit clear <=> byte == 0
                oh[i] = (int)(~y & 0x80808080u);                            // digit -128 where label == un
            }
#pragma unroll
            for (int nti = 0; nti < NT_OWN; ++nti) {
                const int nt = WAVES == 8 ? min(half * NT_OWN + nti, KP_NT - 1) : nti;    // (a clamped duplicate is dropped below)
                const int d = 8 * nt + (un >> 1);
                const v4i *src = reinterpret_cast<const v4i *>(&s_tile[d * KP_PITCH + (wave * 64 + 16 * ug) * 2]);
                const v4i w0 = src[0], w1 = src[1];
                v4i bx_;
                bx_[0] = (int)__builtin_amdgcn_perm((unsigned)w0[1], (unsigned)w0[0], usel);
                bx_[1] = (int)__builtin_amdgcn_perm((unsigned)w0[3], (unsigned)w0[2], usel);
                bx_[2] = (int)__builtin_amdgcn_perm((unsigned)w1[1], (unsigned)w1[0], usel);
                bx_[3] = (int)__builtin_amdgcn_perm((unsigned)w1[3], (unsigned)w1[2], usel);
                accu[nti] = __builtin_amdgcn_mfma_i32_16x16x64_i8(oh, bx_, accu[nti], 0, 0, 0);
            }
        }
        {                                                        // next tile of this workgroup: s1 tiles on / back, modulo the image
            const int tn = reverse ? tin - s1 : tin + s1;
            const bool wrap = reverse ? tn < 0 : tn >= ntiles;
            const bool up = reverse == wrap;                     // block index grows
            const int dq = wrap ? q2 : q1, dr = wrap ? r2 : r1;
            tin = wrap ? (reverse ? tn + ntiles : tn - ntiles) : tn;
            if (up) {
                bx += dr;
                by += dq;
                if (bx >= lo.bx_n) { bx -= lo.bx_n; ++by; }
            } else {
                bx -= dr;
                by -= dq;
                if (bx < 0) { bx += lo.bx_n; --by; }
            }
        }
        KP_PHASE(4);
        __syncthreads();
        KP_PHASE(5);
#ifdef GCS_KP_PHASES
        ++kp_tiles;
#endif
    }
    KP_PHASE_STORE;

    if (!do_acc) return;
    if constexpr (UPD2) {
        // ---- fold (UPD2): rows = (cluster, byte), columns = planes
        constexpr int RW2 = 16 * DSTEPS;                      // planes per row
        int *red = reinterpret_cast<int *>(s_tile);           // [4 blocks of the tile][16 rows][RW2]
        static_assert(4 * 16 * RW2 * 4 <= KP_ROWS * KP_PITCH, "fold buffer exceeds the tile buffer");
#pragma unroll
        for (int pt = 0; pt < DSTEPS; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(wave * 16 + 4 * ug + e) * RW2 + 16 * pt + un] = accu[pt][e];
        __syncthreads();
        const int D1 = D + 1;
        auto folded = [&](int j, int bb, int plane) {
            int sm_ = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) sm_ += red[(w * 16 + 2 * j + bb) * RW2 + plane];
            return -(long long)sm_ / 128;                     // the one-hot digit is -128
        };
        for (int i = tid; i < K * D1; i += NTHR) {
            const int j = i / D1, e = i % D1;                 // e = LOGICAL feature (or D = the count)
            const long long nj = folded(j, 0, D);             // the count row reads +1 in both bytes
            long long out = nj;
            if (e < D) {
                const int c = e / lo.F, f = e - c * lo.F;       // physical plane of logical feature e (levels unrolled)
                int pe = kp_plane_on_level<0>(lo, c, f);
                { const int q = kp_plane_on_level<1>(lo, c, f); pe = q >= 0 ? q : pe; }
                { const int q = kp_plane_on_level<2>(lo, c, f); pe = q >= 0 ? q : pe; }
                { const int q = kp_plane_on_level<3>(lo, c, f); pe = q >= 0 ? q : pe; }
                out = (folded(j, 0, pe) + 128 * nj) + 256 * (folded(j, 1, pe) + 128 * nj);
            }
            partials[partial_index(per_image, b, part, parts, (int)gridDim.y, i, K * D1)] = (uint64_t)out;
        }
    }
    if constexpr (!UPD2) {
    // ---- fold the four waves' accumulators (rows = clusters, cols = byte-planes) and emit the row: every wave
    //      parks its registers in its own slice of the tile buffer (no zero-fill, no atomics), one barrier.
    constexpr int RW = KP_NT * 16;                            // byte-planes per cluster row
    int *red = reinterpret_cast<int *>(s_tile);               // [4 blocks of the tile][16][RW]
    static_assert(4 * 16 * RW * 4 <= KP_ROWS * KP_PITCH, "fold buffer exceeds the tile buffer");
#pragma unroll
    for (int nti = 0; nti < NT_OWN; ++nti) {
        const int nt = WAVES == 8 ? half * NT_OWN + nti : nti;
        if (nt < KP_NT)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(wave * 16 + 4 * ug + e) * RW + 16 * nt + un] = accu[nti][e];
    }
    __syncthreads();
    const int D1 = D + 1;
    auto folded = [&](int j, int bp) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * 16 + j) * RW + bp];
        return -(long long)s / 128;                           // the one-hot digit is -128
    };
    for (int i = tid; i < K * D1; i += NTHR) {
        const int j = i / D1, e = i % D1;                     // e = LOGICAL feature (or D = the count)
        const long long nj = folded(j, cnt_bp);
        long long out = nj;
        if (e < D) {
            const int c = e / lo.F, f = e - c * lo.F;           // physical plane of logical feature e (levels unrolled)
            int pe = kp_plane_on_level<0>(lo, c, f);
            { const int q = kp_plane_on_level<1>(lo, c, f); pe = q >= 0 ? q : pe; }
            { const int q = kp_plane_on_level<2>(lo, c, f); pe = q >= 0 ? q : pe; }
            { const int q = kp_plane_on_level<3>(lo, c, f); pe = q >= 0 ? q : pe; }
            out = (folded(j, 2 * pe) + 128 * nj) + 256 * (folded(j, 2 * pe + 1) + 128 * nj);
        }
        partials[partial_index(per_image, b, part, parts, (int)gridDim.y, i, K * D1)] = (uint64_t)out;
    }
    }
}

// ---------------------------------------------------------------------------------------
// One Lloyd pass for DEEP banks (80 <= D <= 207 with at most 48 planes on every pyramid level, k <= 8: the 8x8 bank of
// BASELINE config 4), every level consumed at its OWN resolution. The tile goes to LDS as it sits in HBM (32.6 KB for
// the 8x8 bank) instead of being replicated to 208 full-resolution plane rows (106 KB):
//   assign:  per level L the partial scores S_L[(j,pat)][parent] = A_pat^L * X^L on v_mfma_i32_32x32x32_i8 (3 K-steps of
//            16 planes per level; N = the block's 64 pixels, 16 / 4 / 1 parents). The key of SPEC.md §4 is linear in the
//            planes: key_j(px) = base_j - 32 U - 2^21 R2 with U = R0 + 256 R1 and (U, R2) summed over the levels at the
//            pixel's parents; the coarse (U, R2) pairs travel through a wave-private LDS table.
//   update:  (round 5) rows of the MFMA = (cluster j, byte b) - k <= 8 fills the 16 rows -, K = (pixel or parent, byte),
//            columns = 16 PLANES: sums[(j, b)][plane] = sel[(j, b)][(px, t)] * X[(px, t)][plane] with sel = the one-hot digit
//            (level 0: -128) or the count of voting pixels of label j under the parent (coarse levels) where t == b, 0
//            elsewhere. The B operand is then the plane row AS IT LIES in LDS (16 bytes = 8 pixels x (lo, hi): no byte
//            de-interleave), and a level's 48 planes are 3 accumulator tiles instead of 6: 48 accumulator VGPRs for the four
//            levels instead of 96 - with the compact tables below what lets THREE workgroups share a CU (168 VGPRs,
//            52.5 KB of LDS) instead of two. Level 0 on v_mfma_i32_16x16x64_i8 (two 32-pixel halves), the coarse levels on
//            v_mfma_i32_16x16x32_i8 (8 / 4 / 2 of the 8 K-slots of a lane group); n_j by v_bcnt.
// Same tile list, sweep order, validity rules, outputs and partial layout as kmeans_pass_mfma_kernel.
constexpr int NV_DL = 48, NV_KS = 3, NV_UT = 3;              // planes (LDS rows), assign K-steps and update plane tiles per level
constexpr int NV_NST = 8;                                     // 16-byte staging chunks per thread (tile_bytes <= 32 768)
constexpr int NV_P0 = KP_TP * 2, NV_P1 = 128 + 16, NV_P2 = 32 + 8, NV_P3 = 8;   // LDS bytes per plane row of level L
// Every LDS image below is laid out against the lane groups the LDS really serves (MI355X_MICROARCH.md, LDS: ds_read_b128 in FOUR
// NON-CONTIGUOUS groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -, ds_read_b64 / _tr_b16 in two
// groups of 32, ds_read_b32 in two groups of 32 on 32 banks), checked access by access with tools/design/lds_bank_model.py. The first
// round-5 build assumed contiguous groups of 16 and measured SQ_LDS_BANK_CONFLICT 31.5 M cycles per launch = 810 per tile.
//  * Level 0 (512-byte rows, no padding): chunk c of plane row r sits at chunk c ^ nv_swz(r). Transposed reads (4 consecutive rows x
//    64 bytes per half wave) want the HIGH two bits of the swizzle to differ over 4 consecutive rows; the update's operand read (16
//    rows, one chunk each; a lane group holds rows {0-3, 12-15} of K-group g and rows {4-11} of K-group g ^ 1, whose chunk differs by
//    XOR 2) wants the low two bits of rows 4-11 closed under XOR 2: the Gray code of r >> 2.
//  * Level 1 (128-byte rows + 16): 16 consecutive rows start in 16 distinct 16-byte columns (9 r mod 16) for the update's read, 4
//    consecutive rows' 32-byte windows are disjoint for the transposed read.
//  * Level 2 (32-byte rows + 8): 16 consecutive rows start in 16 distinct banks of the 32 a ds_read_b32 sees (10 r mod 32).
__host__ __device__ constexpr int nv_swz(int r) { return ((r & 3) << 2) | (((r >> 2) & 3) ^ ((r >> 3) & 1)); }
constexpr int NV_OFF1 = NV_DL * NV_P0, NV_OFF2 = NV_OFF1 + NV_DL * NV_P1, NV_OFF3 = NV_OFF2 + NV_DL * NV_P2;
constexpr int NV_END = NV_OFF3 + NV_DL * NV_P3;
constexpr int NV_PART_W = 8 * (16 + 4 + 1);                  // (U, R2) pairs per wave: [cluster][16 | 4 | 1 parents of level 1 | 2 | 3]
// A fragments per (level, K-step): slot 32 h + 16 g + 4 q + pat for K-half h, pattern pat (3 = all zero) of cluster jj, g = parity of
// jj's bit count, q = jj >> 1: the 16 lanes of a ds_read_b128 lane group hold four clusters of ONE parity class ({0, 3, 5, 6} or
// {1, 2, 4, 7}), so their 16 slots are 16 consecutive 16-byte columns (the round-5 first build, 49 slots with one shared zero slot,
// put a group's lanes on 8 columns: 384 of the 810 conflict cycles per tile)
constexpr int NV_APAT_SLOTS = 64;

// B fragments by hardware transpose (see kmeans_pass_mfma_kernel): issue only; nv_wait() then waits once for everything
template <int PITCH, int OFS = 0>
__device__ __forceinline__ void nv_issue(unsigned addr, v2i (&fa)[NV_KS], v2i (&fb)[NV_KS]) {
    static_assert(OFS + (NV_KS - 1) * 16 * PITCH + 4 * PITCH < 65536, "K-step offsets must fit the 16-bit DS offset field");
#pragma unroll
    for (int kk = 0; kk < NV_KS; ++kk)                       // ONE address register per chain: the K-steps are immediates
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\t"
                     "ds_read_b64_tr_b16 %1, %2 offset:%c4"
                     : "=&v"(fa[kk]), "=&v"(fb[kk])
                     : "v"(addr), "i"(OFS + kk * 16 * PITCH), "i"(OFS + kk * 16 * PITCH + 4 * PITCH)
                     : "memory");
}
// level 0: the two reads of a K-step have bases of their own (swizzled rows)
__device__ __forceinline__ void nv_issue0(unsigned addr_a, unsigned addr_b, v2i (&fa)[NV_KS], v2i (&fb)[NV_KS]) {
#pragma unroll
    for (int kk = 0; kk < NV_KS; ++kk)
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c4\n\t"
                     "ds_read_b64_tr_b16 %1, %3 offset:%c4"
                     : "=&v"(fa[kk]), "=&v"(fb[kk])
                     : "v"(addr_a), "v"(addr_b), "i"(kk * 16 * NV_P0)
                     : "memory");
}
__device__ __forceinline__ void nv_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void nv_take(v2i (&fa)[NV_KS], v2i (&fb)[NV_KS], v4i (&bfr)[NV_KS]) {
#pragma unroll
    for (int kk = 0; kk < NV_KS; ++kk) {
        asm volatile("" : "+v"(fa[kk]), "+v"(fb[kk]));
        bfr[kk] = v4i{fa[kk][0], fa[kk][1], fb[kk][0], fb[kk][1]};
    }
}
__device__ __forceinline__ long long nv_pack64(unsigned lo, unsigned hi) { return (long long)(((unsigned long long)hi << 32) | lo); }

// MINB = workgroups per CU the register budget is set for (3: 168 VGPRs); N0 = staging rounds wholly inside level 0 (see staging)
template <int NL, int MINB, int N0>
__global__ __launch_bounds__(256, MINB) void kmeans_pass_native_kernel(
    const unsigned char *__restrict__ feats, const uint16_t *__restrict__ cent, GcsLayout lo, int K, int per_image,
    int parts, int parts_eff, int reverse, int row_lo, int row_hi, uint64_t *__restrict__ partials,
    void *__restrict__ raster, int raster_u8, int nt_flag) {
    // LDS, one carve-up: [tile as in HBM, rows padded | (U, R2) tables | assign A fragments | labels | key bases | n_j].
    // The transposed reads of level 3 run up to 64 bytes past the tile (unused columns): they land in the tables.
    constexpr int TILE_B = NL == 2 ? NV_OFF2 : NL == 3 ? NV_OFF3 : NV_END;
    constexpr int PART_O = TILE_B, APAT_O = PART_O + 4 * NV_PART_W * 8, LAB_O = APAT_O + NL * NV_KS * NV_APAT_SLOTS * 16;
    constexpr int CONST_O = LAB_O + KP_TP, NJ_O = CONST_O + 16 * 8, LDS_B = NJ_O + 16 * 8;
    static_assert(MINB * ((LDS_B + 1279) / 1280) <= 128, "LDS: gfx950 allocates 160 KB in 1280-byte granules");
    __shared__ __attribute__((aligned(64))) unsigned char s_mem[LDS_B];   // (stage_write XORs bits 4-5 of full level-0 addresses: the base must be a multiple of 64)
    unsigned char *const s_tile = s_mem;
    v4i *const s_apat = reinterpret_cast<v4i *>(s_mem + APAT_O);             // [level][K-step][slot]
    unsigned char *const s_lab = s_mem + LAB_O;
    long long *const s_const = reinterpret_cast<long long *>(s_mem + CONST_O);
    long long *const s_nj = reinterpret_cast<long long *>(s_mem + NJ_O);

    typedef __attribute__((address_space(3))) v4i *lds_v4i_ptr;
    typedef __attribute__((address_space(3))) unsigned char *lds_uchar_ptr;
    const bool do_acc = partials != nullptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grid (B, parts): workgroups are dispatched part-major, so the parts_eff * B working ones are the first to start
    const int b = blockIdx.x, part = blockIdx.y, nb = (int)gridDim.x;
    const int D = lo.D, D1 = D + 1;
    const uint16_t *cset = cent + (size_t)(per_image ? b : 0) * K * D;
    const int ntiles = lo.ntiles;
    const bool working = part < parts_eff;                   // MINB workgroups per CU are resident: the others only emit zeros
    const int nimg = per_image ? 1 : nb;
    const int G = parts_eff * nimg, g = per_image ? part : part * nb + b;
    const int nlist = ntiles * nimg;
    const size_t img0 = per_image ? (size_t)b * ntiles : 0;
    const unsigned char *fb = feats + img0 * lo.tile_bytes;
    auto prow = [&](int i) -> size_t { return partial_index(per_image, b, part, parts, nb, i, K * D1); };
    if (!working) {                                              // a zero partial row, nothing else
        if (do_acc)
            for (int i = tid; i < K * D1; i += 256) partials[prow(i)] = 0;
        return;
    }

    // ---- staging: chunk ci (16 bytes at byte 16*ci of the tile) keeps its place inside its level (level 0 swizzled, level-1
    //      plane rows at NV_P1, level-2 rows at NV_P2). The FIRST tile's loads go out inside the centroid prologue, right behind its
    //      gathers (below): the tile's HBM latency (3 - 4 us under load) passes under the key bases and the A fragments. (In
    //      kmeans_pass_mfma_kernel, whose prologue is 3 us, the same move measured nothing: 0.1566 against 0.1566 ms; not done there.)
    const int nchunk = lo.tile_bytes >> 4;
    const int c1s = NL > 1 ? lo.off[1] >> 4 : nchunk, c2s = NL > 2 ? lo.off[2] >> 4 : nchunk,
              c3s = NL > 3 ? lo.off[3] >> 4 : nchunk;
    // Rounds i < N0 lie wholly inside level 0 (N0 = 6 for the 48-plane level 0 of every bank with 8 orientations, else 0): chunk
    // tid + 256 i is 16 bytes at offset 16 tid + 4096 i of the tile, plane row (tid >> 5) + 8 i, whose swizzle is that of row tid >> 5
    // ^ 2 for odd i - ONE address register for all of them, a scalar add on the tile base per round. The other rounds keep a table.
    v4i st[NV_NST];
    unsigned sadr[NV_NST - N0];                             // per chunk: LDS byte address << 16 | byte offset inside the tile (both < 65 536)
    const unsigned s0adr = (unsigned)(size_t)(lds_uchar_ptr)s_mem + (tid >> 5) * NV_P0 + (((tid & 31) ^ nv_swz(tid >> 5)) << 4);
    static_assert((nv_swz(0) ^ nv_swz(8)) == 3 && (nv_swz(7) ^ nv_swz(15)) == 3 && nv_swz(5) == nv_swz(21), "staging: rows r and r + 8");
    int split = 0;                                          // rounds that hold level-2 chunks (40-byte rows: two 8-byte stores)
#pragma unroll
    for (int i = N0; i < NV_NST; ++i) {
        const int ci = min(tid + 256 * i, nchunk - 1);
        int d;
        if (ci < c1s) d = (ci >> 5) * NV_P0 + ((ci & 31) ^ nv_swz(ci >> 5)) * 16;
        else if (ci < c2s) d = NV_OFF1 + ((ci - c1s) >> 3) * NV_P1 + ((ci - c1s) & 7) * 16;
        else if (ci < c3s) d = NV_OFF2 + ((ci - c2s) >> 1) * NV_P2 + ((ci - c2s) & 1) * 16;
        else d = NV_OFF3 + (ci - c3s) * 16;
        sadr[i - N0] = ((unsigned)(size_t)(lds_uchar_ptr)s_mem + (unsigned)d) << 16 | (unsigned)(ci * 16);
        if (NL > 2 && 256 * i < c3s && 256 * i + 255 >= c2s) split |= 1 << i;
    }
    // Raw buffer loads: a descriptor over the tile (uniform base) + 32-bit lane offset; the cache policy is an immediate of the
    // intrinsic, so the plain and the `nt` form both survive (see stage_load_split of kmeans_pass_mfma_kernel and lloyd_pass)
    auto stage_load = [&](int tile, bool nt) {
        const unsigned char *tb = fb + (size_t)tile * lo.tile_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(tb), 0, -1, 0x00020000);
        unsigned o0 = (unsigned)tid * 16u;
        asm volatile("" : "+v"(o0));                        // (opaque: see below)
        auto go = [&](auto aux_c) {
            constexpr int AUX = decltype(aux_c)::value;     // gfx940+: bit 1 = nt
#pragma unroll
            for (int i = 0; i < N0; ++i) st[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o0, i * 4096, AUX));
#pragma unroll
            for (int i = N0; i < NV_NST; ++i) {
                unsigned o = sadr[i - N0] & 0xffffu;        // (opaque: hoisted out of the tile loop these spilled - and a reload
                asm volatile("" : "+v"(o));                 //  inside the loop waits for vmcnt(0))
                st[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, AUX));
            }
        };
        if (nt) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 0>{});
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int i = 0; i < N0; ++i) *reinterpret_cast<lds_v4i_ptr>((s0adr ^ ((i & 1) * 48u)) + i * 8 * NV_P0) = st[i];
#pragma unroll
        for (int i = N0; i < NV_NST; ++i) {
            const unsigned a = sadr[i - N0] >> 16;
            if (NL > 2 && (split >> i & 1)) {                // (wave-uniform) level-2 rows are 8-byte aligned only
                typedef __attribute__((address_space(3))) v2i *lds_v2i_ptr;
                *reinterpret_cast<lds_v2i_ptr>(a) = v2i{st[i][0], st[i][1]};
                *reinterpret_cast<lds_v2i_ptr>(a + 8) = v2i{st[i][2], st[i][3]};
            } else {
                *reinterpret_cast<lds_v4i_ptr>(a) = st[i];
            }
        }
    };
    auto phys = [&](int lt) { return reverse ? nlist - 1 - lt : lt; };
    // ---- centroids -> scratch [8 clusters][4 levels][48 planes] u16, offset-binary, zero where nothing exists. The six gathers of a
    //      thread go out FIRST, the first tile's eight loads behind them, and only then are the gathers consumed: loads return in
    //      order, so a gather issued behind the tile (the first round-5 build) waited for the tile's 32 KB as well (stamps: the
    //      gather loop took 5.6 - 6.7 us of a 10 - 11.6 us prologue).
    uint16_t *cs = reinterpret_cast<uint16_t *>(s_tile);
    static_assert(8 * 4 * NV_DL * 2 <= TILE_B && (8 * 4 * NV_DL) % 256 == 0, "centroid scratch: tile buffer, whole rounds");
    constexpr int NCS = 8 * 4 * NV_DL / 256;
    unsigned cv[NCS];
#pragma unroll
    for (int r = 0; r < NCS; ++r) {
        const int i = tid + 256 * r;
        const int j = i / (4 * NV_DL), L = (i / NV_DL) & 3, pl = i % NV_DL;
        bool ok = false;                                     // (levels unrolled: kp_logical_of)
        int e = 0;
        if (L == 0 && pl < lo.DL[0]) { ok = true; e = kp_logical_of<0>(lo, pl); }
        if (NL > 1 && L == 1 && pl < lo.DL[1]) { ok = true; e = kp_logical_of<1>(lo, pl); }
        if (NL > 2 && L == 2 && pl < lo.DL[2]) { ok = true; e = kp_logical_of<2>(lo, pl); }
        if (NL > 3 && L == 3 && pl < lo.DL[3]) { ok = true; e = kp_logical_of<3>(lo, pl); }
        ok = ok && j < K;
        const int src = ok ? j * D + e : 0;
        cv[r] = (unsigned)cset[src] | (ok ? 0u : 0x10000u);   // (bit 16: nothing exists there)
    }
    int ltile = g;
    const int nt_limit = __builtin_amdgcn_readfirstlane(nt_flag);   // list positions below it are loaded `nt` (lloyd_pass)
    if (ltile < nlist) stage_load(phys(ltile), ltile < nt_limit);
#pragma unroll
    for (int r = 0; r < NCS; ++r) cs[tid + 256 * r] = (cv[r] & 0x10000u) ? (uint16_t)0 : (uint16_t)(cv[r] ^ 0x8080u);
    __syncthreads();
    for (int j = tid >> 4; j < 16; j += 16) {                // key base, as in kmeans_pass_mfma_kernel
        const int sub = tid & 15;
        long long nrm = 0, scl = 0, sch = 0;
        if (j < K && j < 8) {
#pragma unroll
            for (int L = 0; L < NL; ++L)                       // (levels unrolled: every lo.DL[L] a plain kernel argument)
                for (int pl = sub; pl < lo.DL[L]; pl += 16) {
                    const unsigned c = cs[(j * 4 + L) * NV_DL + pl] ^ 0x8080u;
                    nrm += (long long)((unsigned long long)c * c);
                    scl += c & 255;
                    sch += c >> 8;
                }
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            nrm += __shfl_xor(nrm, m);
            scl += __shfl_xor(scl, m);
            sch += __shfl_xor(sch, m);
        }
        if (sub == 0) {
            const long long q = 16384LL * D;
            const long long gg = (128 * scl - q) + 256 * (128 * (sch + scl) - 2 * q) + 65536 * (128 * sch - q);
            s_const[j] = j < K ? 16 * (nrm - 2 * gg) + j : (1LL << 62) + j;
        }
    }
    // ---- assign A fragments per level: row r = 4*jj + pat (cluster jj), k-slot (h, t) of K-step kk = (plane 16*kk + 8*h + t/2,
    //      byte t&1) of the level; patterns LL / M / HH as in kmeans_pass_mfma_kernel. Row pattern 3 is all zero: the 16 lanes
    //      that hold it read a zero slot of their own (NV_APAT_SLOTS).
    const int a_jj = (lane & 31) >> 2;
    const int a_slot = 32 * (lane >> 5) + 16 * (__builtin_popcount(a_jj) & 1) + 4 * (a_jj >> 1) + (lane & 3);
    if (wave < NL) {                                         // one level per wave
        const int r = lane & 31, h = lane >> 5;
        const int jj = r >> 2, pat = r & 3;
        const unsigned msk = pat == 0 ? 0x00ff00ffu : pat == 1 ? 0xffffffffu : pat == 2 ? 0xff00ff00u : 0u;
        const unsigned sel = pat == 1 ? 0x02030001u : 0x03020100u;
        {
            const int L = wave;
#pragma unroll
            for (int kk = 0; kk < NV_KS; ++kk) {
                const v4i w = *reinterpret_cast<const v4i *>(&cs[(jj * 4 + L) * NV_DL + 16 * kk + 8 * h]);
                v4i f;
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = (int)(__builtin_amdgcn_perm(0u, (unsigned)w[e], sel) & msk);
                s_apat[(L * NV_KS + kk) * NV_APAT_SLOTS + a_slot] = f;   // (pattern 3: msk == 0, f == 0)
            }
        }
    }
    __syncthreads();                                   // scratch reads done: the tile buffer is free
    v4i accu[NL][NV_UT];
#pragma unroll
    for (int L = 0; L < NL; ++L)
#pragma unroll
        for (int nt = 0; nt < NV_UT; ++nt) accu[L][nt] = v4i{0, 0, 0, 0};
    int cntacc = 0;

    // update operand coordinates: row um = 2 * cluster + byte, K-group ukg = pixel rows 2 ukg, 2 ukg + 1 of the block
    const int um = lane & 15, ukg = lane >> 4;
    // one-hot bytes (b0 b1 b2 b3) of four pixels -> K-slots (px, t): (b0 0 b1 0 | b2 0 b3 0) for the low-byte rows, shifted up one
    // byte for the high-byte rows (v_perm: selectors 4 .. 7 = bytes of the zero operand). ONE register holds the row's personality:
    // the selector for (b0, b1); the one for (b2, b3) is it ^ 0x02020202, the byte shift of the count operands (it & 4) << 1,
    // and the cluster's compare pattern comes from the lane number (three invariants fewer than the tile loop can keep).
    const unsigned uselA0 = (um & 1) ? 0x01040004u : 0x04010400u;

    const int s1 = __builtin_amdgcn_readfirstlane(G % ntiles);
    const int q1 = __builtin_amdgcn_readfirstlane(4 * s1 / lo.bx_n), r1 = 4 * s1 - q1 * lo.bx_n;
    const int q2 = __builtin_amdgcn_readfirstlane(4 * (ntiles - s1) / lo.bx_n), r2 = 4 * (ntiles - s1) - q2 * lo.bx_n;
    int tin = __builtin_amdgcn_readfirstlane(phys(g < nlist ? g : 0) % ntiles);
    int by, bx;
    {
        const int blk0 = 4 * tin + wave;
        by = blk0 / lo.bx_n;
        bx = blk0 - by * lo.bx_n;
    }
    // ---- LDS addresses as plain integers: every lane keeps ONE base per access pattern and everything else is an
    //      immediate of the DS instruction (left to itself hipcc hoists one register per (level, cluster pair, K-step) out of the
    //      tile loop: 30 more invariants than three workgroups per CU leave room for)
    typedef __attribute__((address_space(3))) const v4i *lds_cv4i;
    typedef __attribute__((address_space(3))) const v2i *lds_cv2i;
    typedef __attribute__((address_space(3))) v2i *lds_v2i;
    typedef __attribute__((address_space(3))) const long long *lds_ci64;
    typedef __attribute__((address_space(3))) const unsigned *lds_cu32;
    typedef __attribute__((address_space(3))) const uint16_t *lds_cu16;
    typedef __attribute__((address_space(3))) unsigned char *lds_u8;
    const unsigned L0 = (unsigned)(size_t)(lds_uchar_ptr)s_mem;            // LDS address of the carve-up
    unsigned a_tr[4], a_tr0b, a_apat, a_pw[3], a_pr[2], a_labw, a_labr, a_ub[4];
    {
        const int n = lane & 31, h = lane >> 5, i16 = lane & 15, pxblk = (lane >> 4) & 1;
        const int rowq = 8 * h + (i16 >> 2), colq = 4 * (i16 & 3);
        // transposed reads. level 0 (swizzled rows, see nv_swz): the first read of a K-step takes rows rowq + 16 kk, the second rows
        // + 4 - their chunk columns differ (nv_swz(r + 4) != nv_swz(r)), hence two bases; sub-tile 1 = both ^ 64.
        // level 1: the block's 16 parents are columns 16*wave .. +15; level 2: its 4 parents are columns
        // 4*wave .. +3 of the plane's 16; level 3: its parent is column `wave` of the plane's 4 (the transpose read wants
        // 8-byte-aligned column starts, so these two read the whole plane row). The second 16-lane block of a half wave
        // (output columns 16 .. 31: kept by no coarse level) reads the addresses of the first: a broadcast, no bank of its own.
        const int c0 = wave * 8 + 2 * pxblk + ((i16 & 3) >> 1);
        a_tr[0] = rowq * NV_P0 + ((c0 ^ nv_swz(rowq)) << 4) + (i16 & 1) * 8;
        a_tr0b = (rowq + 4) * NV_P0 + ((c0 ^ nv_swz(rowq + 4)) << 4) + (i16 & 1) * 8;
        a_tr[1] = L0 + NV_OFF1 + rowq * NV_P1 + (16 * wave + colq) * 2;
        a_tr[2] = L0 + NV_OFF2 + rowq * NV_P2 + colq * 2;
        a_tr[3] = L0 + NV_OFF3 + rowq * NV_P3 + (16 * pxblk + colq) * 2;
        a_apat = L0 + APAT_O + a_slot * 16;
        // the wave's (U, R2) table: [cluster][16] level 1 | 128 + [cluster][4] level 2 | 160 + [cluster] level 3; cluster 2 gq + h
        const unsigned pw = L0 + PART_O + wave * NV_PART_W * 8;
        a_pw[0] = pw + (h * 16 + (n & 15)) * 8;
        a_pw[1] = pw + (128 + h * 4 + (n & 3)) * 8;
        a_pw[2] = pw + (160 + h) * 8;
        a_pr[0] = pw + (h * 16 + (n >> 4) * 4 + ((n & 7) >> 1)) * 8;       // sub-tile 0; sub-tile 1: 8 parents on
        a_pr[1] = pw + (128 + h * 4 + ((n & 7) >> 2)) * 8;                 // sub-tile 1: 2 parents on
        a_labw = L0 + LAB_O + wave * 64 + n;
        a_labr = L0 + LAB_O + wave * 64 + 16 * ukg;
        a_ub[0] = um * NV_P0 + (((wave * 8 + 2 * ukg) ^ nv_swz(um)) << 4);          // pixel row 2 ukg of the block; row 2 ukg + 1: ^ 16
        a_ub[1] = L0 + NV_OFF1 + um * NV_P1 + (wave * 16 + 4 * ukg) * 2;
        a_ub[2] = L0 + NV_OFF2 + um * NV_P2 + (wave * 4 + (ukg >> 1) * 2) * 2;
        a_ub[3] = L0 + NV_OFF3 + um * NV_P3 + wave * 2;
    }
    for (; ltile < nlist; ltile += G) {
        const int tile = phys(ltile);
        stage_write();
        __syncthreads();
        __builtin_amdgcn_s_setprio(3);
        if (ltile + G < nlist) stage_load(phys(ltile + G), ltile + G < nt_limit);   // in flight during the MFMAs
        __builtin_amdgcn_s_setprio(0);

        const int blk = 4 * tin + wave;
        const int n = lane & 31, h = lane >> 5;
        // -------- assign
        // Software pipeline over the five MFMA chains of a tile (levels 1 .. NL-1, then the two 32-pixel sub-tiles of level 0):
        // the transposed reads of chain c + 1 go out before the MFMAs of chain c, one chain's fragments in flight at a time
        // (all of them at once: 36 more VGPRs than three workgroups per CU leave).
        v2i fa[2][NV_KS], fbv[2][NV_KS];
        auto issue_chain = [&](int c, v2i (&xa)[NV_KS], v2i (&xb)[NV_KS]) {     // c = 0 .. NL-2: level c + 1; NL-1, NL: sub-tiles
            if (c == 0 && NL > 1) nv_issue<NV_P1>(a_tr[1], xa, xb);
            else if (c == 1 && NL > 2) nv_issue<NV_P2>(a_tr[2], xa, xb);
            else if (c == 2 && NL > 3) nv_issue<NV_P3>(a_tr[3], xa, xb);
            else nv_issue0(L0 + (a_tr[0] ^ (c == NL - 1 ? 0u : 64u)), L0 + (a_tr0b ^ (c == NL - 1 ? 0u : 64u)), xa, xb);
        };
        auto apat = [&](int L, int kk) { return *reinterpret_cast<lds_cv4i>(a_apat + (L * NV_KS + kk) * NV_APAT_SLOTS * 16); };
        issue_chain(0, fa[0], fbv[0]);
        // coarse levels: (U, R2) per cluster and parent of this wave's block -> the wave's table
#pragma unroll
        for (int L = 1; L < NL; ++L) {
            v4i bfr[NV_KS];
            nv_wait();
            nv_take(fa[(L - 1) & 1], fbv[(L - 1) & 1], bfr);
            issue_chain(L, fa[L & 1], fbv[L & 1]);
            v16i acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0;
#pragma unroll
            for (int kk = 0; kk < NV_KS; ++kk) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(apat(L, kk), bfr[kk], acc, 0, 0, 0);
            const bool keep = L == 1 ? n < 16 : L == 2 ? (n >> 2) == wave : n == wave;
            const int cstride = L == 1 ? 32 : L == 2 ? 8 : 2;          // two clusters on
            if (keep)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<lds_v2i>(a_pw[L - 1] + gq * cstride * 8) =
                        v2i{__mul24(acc[4 * gq + 1], 256) + acc[4 * gq], acc[4 * gq + 2]};
        }
        // level 0: two 32-pixel sub-tiles (rows 4*sub .. 4*sub+3 of the block), one after the other
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            v16i acc0;
            {
                const int c = NL - 1 + sub;
                v4i bfr[NV_KS];
                nv_wait();
                nv_take(fa[c & 1], fbv[c & 1], bfr);                          // chain c travels in buffer c & 1
                if (sub == 0) issue_chain(c + 1, fa[(c + 1) & 1], fbv[(c + 1) & 1]);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc0[e] = 0;
#pragma unroll
                for (int kk = 0; kk < NV_KS; ++kk) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(apat(0, kk), bfr[kk], acc0, 0, 0, 0);
            }
            long long best = 0x7fffffffffffffffLL;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                int u = __mul24(acc0[4 * gq + 1], 256) + acc0[4 * gq], r2v = acc0[4 * gq + 2];
                if (NL > 1) { const v2i c = *reinterpret_cast<lds_cv2i>(a_pr[0] + gq * 256 + sub * 64); u += c[0]; r2v += c[1]; }
                if (NL > 2) { const v2i c = *reinterpret_cast<lds_cv2i>(a_pr[1] + gq * 64 + sub * 16); u += c[0]; r2v += c[1]; }
                if (NL > 3) { const v2i c = *reinterpret_cast<lds_cv2i>(a_pw[2] + gq * 16); u += c[0]; r2v += c[1]; }
                // key base of cluster 2 gq + h: a broadcast read (uniform address per half wave)
                const long long kb = *reinterpret_cast<lds_ci64>(L0 + CONST_O + gq * 16 + h * 8);
                long long key = mad_i64_i32(u, -32, kb);
                key = mad_i64_i32(r2v, -2097152, key);
                best = key < best ? key : best;
            }
            const unsigned blo = (unsigned)best, bhi = (unsigned)((unsigned long long)best >> 32);
            const auto s0 = __builtin_amdgcn_permlane32_swap(blo, blo, false, false);
            const auto s1v = __builtin_amdgcn_permlane32_swap(bhi, bhi, false, false);
            // (element 0 = the lower half's best in both halves, element 1 = the upper half's: see kmeans_pass_mfma_kernel)
            const long long ka = (long long)(((unsigned long long)s1v[0] << 32) | s0[0]);
            const long long kb2 = (long long)(((unsigned long long)s1v[1] << 32) | s0[1]);
            const int bj = (int)((kb2 < ka ? s0[1] : s0[0]) & 15);
            if (h == 0) {
                const int yi = 4 * sub + (n >> 3), xi = n & 7;               // pixel inside the block
                int y = 8 * by + yi, x = 8 * bx + xi, xlim = lo.W;          // see kmeans_pass_mfma_kernel
                if (NL <= 2 && blk >= lo.nmain) {               // (deeper banks have main blocks only)
                    int no = n;
                    asm volatile("" : "+v"(no));
                    gcs_strip_pixel(lo, blk, 4 * sub + (no >> 3), no & 7, y, x, xlim);
                }
                const bool inimg = blk < lo.nblk && y < lo.H && x < xlim;
                const bool valid = inimg && y >= row_lo && y < row_hi;
                *reinterpret_cast<lds_u8>(a_labw + sub * 32) = valid ? (unsigned char)bj : (unsigned char)0xFF;
                if (raster && inimg) {                        // raster label map (see kmeans_pass_mfma_kernel)
                    const size_t o = ((size_t)(per_image ? b : tile / ntiles) * lo.H + y) * lo.W + x;
                    if (raster_u8) static_cast<uint8_t *>(raster)[o] = (uint8_t)bj;
                    else static_cast<int32_t *>(raster)[o] = bj;
                }
            }
        }
        // -------- update (the block's labels were written by this wave: no barrier)
        if (do_acc) {
            const v4i lw = *reinterpret_cast<lds_cv4i>(a_labr);   // labels of pixel rows 2 ukg (bytes 0-7), 2 ukg + 1
            unsigned uselA = uselA0, lno = (unsigned)lane;
            asm volatile("" : "+v"(uselA), "+v"(lno));            // (opaque: what follows is recomputed per tile, not hoisted)
            const unsigned uselB = uselA ^ 0x02020202u, ush = (uselA & 4u) << 1;
            const unsigned eqr = ((lno >> 1) & 7u) * 0x01010101u;
            v4i oh;                                              // byte 0x80 where label == this row's cluster
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned x = (unsigned)lw[i] ^ eqr;
                const unsigned y = (x | 0x80808080u) - 0x01010101u;
                oh[i] = (int)(~y & 0x80808080u);
            }
            cntacc += __builtin_popcount((unsigned)oh[0]) + __builtin_popcount((unsigned)oh[1]) +
                      __builtin_popcount((unsigned)oh[2]) + __builtin_popcount((unsigned)oh[3]);
            // level 0: half hf = pixel row 2 ukg + hf of the block for this K-group; K-slot 2 q + t = (pixel q of the row, byte t)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                v4i a0;
                a0[0] = (int)__builtin_amdgcn_perm(0u, (unsigned)oh[2 * hf], uselA);
                a0[1] = (int)__builtin_amdgcn_perm(0u, (unsigned)oh[2 * hf], uselB);
                a0[2] = (int)__builtin_amdgcn_perm(0u, (unsigned)oh[2 * hf + 1], uselA);
                a0[3] = (int)__builtin_amdgcn_perm(0u, (unsigned)oh[2 * hf + 1], uselB);
#pragma unroll
                for (int nt = 0; nt < NV_UT; ++nt) {
                    const v4i bq = *reinterpret_cast<lds_cv4i>(L0 + (a_ub[0] ^ (hf * 16u)) + nt * 16 * NV_P0);
                    accu[0][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bq, accu[0][nt], 0, 0, 0);
                }
            }
            if constexpr (NL > 1) {
                // counts of this row's label under the 4 level-1 parents of pixel rows 2 ukg, 2 ukg + 1 (parent row ukg, columns 0..3)
                const unsigned e0 = (unsigned)oh[0] >> 7, e1 = (unsigned)oh[1] >> 7, e2 = (unsigned)oh[2] >> 7, e3 = (unsigned)oh[3] >> 7;
                const unsigned sa = e0 + e2, sb = e1 + e3;       // bytes: columns 0..3 / 4..7, both rows
                const unsigned ta = (sa & 0x00ff00ffu) + ((sa >> 8) & 0x00ff00ffu);   // bytes (c0 0 c1 0): K-slots (parent 0, lo) (0, hi) (1, lo) (1, hi)
                const unsigned tb = (sb & 0x00ff00ffu) + ((sb >> 8) & 0x00ff00ffu);   // parents 2, 3
                const long long a1 = nv_pack64(ta << ush, tb << ush);
#pragma unroll
                for (int nt = 0; nt < NV_UT; ++nt) {
                    const v2i w = *reinterpret_cast<lds_cv2i>(a_ub[1] + nt * 16 * NV_P1);
                    accu[1][nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, nv_pack64((unsigned)w[0], (unsigned)w[1]), accu[1][nt], 0, 0, 0);
                }
                if constexpr (NL > 2) {
                    // level 2: this K-group's partial counts of the parents (ukg >> 1, 0 / 1): K-slots (parent column, byte)
                    const unsigned cl = (ta & 0xffu) + (ta >> 16), cr = (tb & 0xffu) + (tb >> 16);
                    const long long a2 = nv_pack64((cl | (cr << 16)) << ush, 0u);
#pragma unroll
                    for (int nt = 0; nt < NV_UT; ++nt) {
                        const unsigned w = *reinterpret_cast<lds_cu32>(a_ub[2] + nt * 16 * NV_P2);
                        accu[2][nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a2, nv_pack64(w, 0u), accu[2][nt], 0, 0, 0);
                    }
                    if constexpr (NL > 3) {
                        const long long a3 = nv_pack64((cl + cr) << ush, 0u);   // level 3: the block's one parent
#pragma unroll
                        for (int nt = 0; nt < NV_UT; ++nt) {
                            const unsigned w = *reinterpret_cast<lds_cu16>(a_ub[3] + nt * 16 * NV_P3);
                            accu[3][nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a3, nv_pack64(w, 0u), accu[3][nt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        {                                                        // next tile of this workgroup (see kmeans_pass_mfma_kernel)
            const int tn = reverse ? tin - s1 : tin + s1;
            const bool wrap = reverse ? tn < 0 : tn >= ntiles;
            const bool up = reverse == wrap;
            const int dq = wrap ? q2 : q1, dr = wrap ? r2 : r1;
            tin = wrap ? (reverse ? tn + ntiles : tn - ntiles) : tn;
            if (up) {
                bx += dr;
                by += dq;
                if (bx >= lo.bx_n) { bx -= lo.bx_n; ++by; }
            } else {
                bx -= dr;
                by -= dq;
                if (bx < 0) { bx += lo.bx_n; --by; }
            }
        }
        __syncthreads();
    }
    if (!do_acc) return;

    // ---- fold, every level at once: the whole LDS image is free now, so each wave parks all its accumulators ([wave][level][16 rows]
    //      [48 planes] ints; row 2 j + t = byte t of cluster j) and its voting pixel counts, two barriers, and the row is written in
    //      LOGICAL feature order (consecutive threads = consecutive 8-byte elements of the partial row). The first round-5 build folded
    //      level by level through the tile buffer: eight barriers and stores in physical plane order.
    constexpr int RW = NV_UT * 16;
    int *red = reinterpret_cast<int *>(s_mem);
    int *s_cnt = red + 4 * NL * 16 * RW;                          // [wave][K-group][row]
    static_assert((4 * NL * 16 * RW + 4 * 4 * 16) * 4 <= NJ_O, "fold scratch exceeds the LDS image in front of s_nj");
#pragma unroll
    for (int L = 0; L < NL; ++L)
#pragma unroll
        for (int nt = 0; nt < NV_UT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[((wave * NL + L) * 16 + 4 * ukg + e) * RW + 16 * nt + um] = accu[L][nt][e];
    s_cnt[(wave * 4 + ukg) * 16 + um] = cntacc;
    __syncthreads();
    if (tid < 8) {
        long long c = 0;
        for (int w = 0; w < 16; ++w) c += s_cnt[w * 16 + 2 * tid];
        s_nj[tid] = c;
    }
    __syncthreads();
    for (int i = tid; i < K * D1; i += 256) {
        const int j = i / D1, e = i - j * D1;                     // e = LOGICAL feature (or D = the count)
        const long long nj = s_nj[j];
        long long out = nj;
        if (e < D) {
            const int c = e / lo.F, f = e - c * lo.F;           // level and plane in its level of logical feature e (levels unrolled)
            int L = 0, pl = 0;
            { const int q = kp_plane_on_level<0>(lo, c, f); if (q >= 0) { L = 0; pl = q - lo.row0[0]; } }
            if (NL > 1) { const int q = kp_plane_on_level<1>(lo, c, f); if (q >= 0) { L = 1; pl = q - lo.row0[1]; } }
            if (NL > 2) { const int q = kp_plane_on_level<2>(lo, c, f); if (q >= 0) { L = 2; pl = q - lo.row0[2]; } }
            if (NL > 3) { const int q = kp_plane_on_level<3>(lo, c, f); if (q >= 0) { L = 3; pl = q - lo.row0[3]; } }
            long long flo = 0, fhi = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                flo += red[((w * NL + L) * 16 + 2 * j) * RW + pl];
                fhi += red[((w * NL + L) * 16 + 2 * j + 1) * RW + pl];
            }
            if (L == 0) {                                         // the one-hot digit is -128
                flo = -flo / 128;
                fhi = -fhi / 128;
            }
            out = (flo + 128 * nj) + 256 * (fhi + 128 * nj);
        }
        partials[prow(i)] = (uint64_t)out;
    }
}

static size_t assign_lds_bytes(int D, int k, int R) {
    size_t a = ((size_t)D * k * 2 * 4 + 15) & ~(size_t)15;
    size_t c = ((size_t)k * 8 + 15) & ~(size_t)15;
    return a + c + (size_t)k * (D + 1) * R * 4;
}

template <int K>
static int launch_assign(const uint16_t *feats, const uint16_t *cent, int B, const GcsLayout &lo, int n_sets,
                         int row_lo, int row_hi, uint8_t *labels, uint64_t *partials, hipStream_t stream) {
    const int D = lo.D;
    const int parts = (int)gcs_kmeans_parts_per_image(B, lo.H, lo.W);
    int R = 32;
    while (R > 1 && assign_lds_bytes(D, K, R) > 120 * 1024) R >>= 1;
    const size_t lds = assign_lds_bytes(D, K, R);
    if (lds > 160 * 1024) return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: k*D too large for LDS");
    // raise the dynamic-LDS cap: per device and cheap, so set on every launch (no process-wide cache to race on)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kmeans_assign_kernel<K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return gcs_hip_fail(e, "hipFuncSetAttribute(assign)");
    hipLaunchKernelGGL(kmeans_assign_kernel<K>, dim3(parts, B), dim3(256), lds, stream,
                       reinterpret_cast<const unsigned char *>(feats), cent, lo, n_sets == B ? 1 : 0, parts, R, row_lo,
                       row_hi, labels, partials);
    GCS_CHECK_LAUNCH("gcs_kmeans_assign_accumulate");
    return GCS_OK;
}

// Working workgroups per image of the native pass: GCS_NV_MINB (three) 4-wave workgroups per CU are resident, so at most
// 256 * GCS_NV_MINB / B of the `parts` workgroups of an image work (the others only write zero rows) - but never so few that a
// wave's int32 MFMA accumulators can overflow: they are flushed only at the end of the pass, a voting pixel adds up to 128 * 128
// to one of them and a wave sees a quarter of its workgroup's pixels, so a workgroup may own at most 2^31 / 2^14 * 4 = 524 288
// pixels; the bound used is half of that. (`parts` itself keeps a workgroup below 65 536 pixels: gcs_kmeans_parts_per_image.)
#ifndef GCS_NV_MINB
#define GCS_NV_MINB 3
#endif
constexpr long long NV_MAX_PX_PER_WORKGROUP = 262144;
static int native_parts_eff(int B, int parts, long long px_image, int minb = GCS_NV_MINB) {
    const int slots = 256 * minb;
    int eff = slots / B > 0 ? slots / B : 1;
    const long long need = (px_image + NV_MAX_PX_PER_WORKGROUP - 1) / NV_MAX_PX_PER_WORKGROUP;
    if (eff < need) eff = (int)need;
    return eff < parts ? eff : parts;
}
// Test hook (host only): the working workgroups per image the native pass would use, 0 for a bad shape.
extern "C" int gcs_selftest_native_parts(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || gcs_tiles_upper(H, W) > 0x3fffffffLL) return 0;
    return native_parts_eff(B, (int)gcs_kmeans_parts_per_image(B, H, W), gcs_tiles_upper(H, W) * KP_TP);
}

extern "C" int gcs_labels_widen(const uint8_t *labels, int B, int H, int W, int32_t *out, gcs_stream_t stream);

// One Lloyd pass, whatever it emits: uint8 label map, partial sums, raster label map (any subset, not none).
static int lloyd_pass(const uint16_t *feats, const uint16_t *cent, int B, int H, int W, int n_scales, int n_orient, int k,
                      int n_sets, int row_lo, int row_hi, int reverse, uint8_t *labels, uint64_t *partials, void *raster,
                      int raster_u8, gcs_stream_t stream) {
    if (!feats || !cent || (!labels && !partials && !raster))
        return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: NULL pointer (labels and partials may not both be NULL)");
    LAYOUT_OR_FAIL(lo, "gcs_kmeans_assign_accumulate");
    if (row_lo < 0 || row_hi > H || row_lo >= row_hi)
        return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: need 0 <= row_lo < row_hi <= H");
    if (B > 65535) return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: B too large for one launch");
    if ((long long)B * lo.ntiles > 0x1fffffffLL)   // 4 * (tile index in the batch list) is kept in an int
        return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: batch too large for one launch");
    if (k < 1 || k > GCS_K_MAX) return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: k must be in 1..16");
    if (n_sets != 1 && n_sets != B)
        return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: n_sets must be 1 or B");
    const int D = lo.D;
    if (D < 16 * KP_DSTEPS_WIDE) { // matrix-core pass (every BASELINE bank: 4x6 -> D = 72, 8x8 -> D = 192)
        const int parts = (int)gcs_kmeans_parts_per_image(B, H, W);
        // the matrix-core passes have ONE label output, in raster order: the caller's raster map, or its uint8 label map
        // (assign_accumulate with labels: every pixel of the image is labelled, halo rows of a row window included)
        void *lab_out = raster ? raster : static_cast<void *>(labels);
        const int lab_u8 = raster ? raster_u8 : 1;
        // which tile loads carry the nontemporal hint (see kp_nt_limit)
        const int nt_flag = 0;                                        // the wide-slab kernels load plain
#define GCS_KP_LAUNCHW(KT_, NST_, DS_, WV_)                                                                              \
    hipLaunchKernelGGL((kmeans_pass_mfma_kernel<KT_, NST_, DS_, WV_>), dim3(parts, B), dim3(64 * WV_), 0, stream,         \
                       reinterpret_cast<const unsigned char *>(feats), cent, lo, k, n_sets == B ? 1 : 0, parts,          \
                       reverse ? 1 : 0, row_lo, row_hi, partials, lab_out, lab_u8, nt_flag)
#define GCS_KP_LAUNCH(KT_, NST_, DS_) GCS_KP_LAUNCHW(KT_, NST_, DS_, 4)
        const int nchunk = lo.tile_bytes / 16;
        const int nst = (nchunk + 255) / 256;                         // staging chunks per thread (4-wave workgroups)
        if (lo.split) {                                               // (D < 80, at most two levels: csrc/common.h)
            const int rounds = ((lo.S >> 4) + 255) / 256;             // staging rounds: items of 16 slots per thread
            const int nt_limit = kp_nt_limit(lo, B, n_sets, lo.tile_bytes / 4 * 3);   // (a pass streams 3/4 of a tile's bytes)
            if ((unsigned long long)lo.img_bytes >= (1ull << 32))    // MID / TOP runs are addressed by 32-bit offsets from the LO run
                return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: image too large for the split slab's pass");
#define GCS_KP_LAUNCHS(KT_, NR_, ...)                                                                                          \
    hipLaunchKernelGGL((kmeans_pass_mfma_kernel<KT_, NR_, KP_DSTEPS_NARROW, 4, true __VA_OPT__(,) __VA_ARGS__>), dim3(parts, B), dim3(256), 0, stream, \
                       reinterpret_cast<const unsigned char *>(feats), cent, lo, k, n_sets == B ? 1 : 0, parts,             \
                       reverse ? 1 : 0, row_lo, row_hi, partials, lab_out, lab_u8, nt_limit)
            // (measured and dropped, profiles/r6_notes.md: eight waves per workgroup at two workgroups per CU - 0.28 against 0.15 ms
            //  per pass -, the assign A fragments in LDS, the second sub-tile's transposed reads under the first one's epilogue)
            if (k <= 8 && rounds <= 3 && lo.DL[0] >= 32) { GCS_KP_LAUNCHS(1, 3, 2); }
            else if (k <= 8 && rounds <= 3) { GCS_KP_LAUNCHS(1, 3); }
            else if (k <= 8) { GCS_KP_LAUNCHS(1, 5); }
            else { GCS_KP_LAUNCHS(2, 5); }
#undef GCS_KP_LAUNCHS
        } else if (D < 16 * KP_DSTEPS_NARROW) {
            if (k <= 8) {
                if (nst <= 3) { GCS_KP_LAUNCH(1, 3, KP_DSTEPS_NARROW); }
                else if (nst <= 6) { GCS_KP_LAUNCH(1, 6, KP_DSTEPS_NARROW); }
                else if (nst <= 9) { GCS_KP_LAUNCH(1, 9, KP_DSTEPS_NARROW); }
                else { GCS_KP_LAUNCH(1, 10, KP_DSTEPS_NARROW); }
            } else {
                if (nst <= 6) { GCS_KP_LAUNCH(2, 6, KP_DSTEPS_NARROW); }
                else { GCS_KP_LAUNCH(2, 10, KP_DSTEPS_NARROW); }
            }
        } else if (k <= 8) {
            bool native = nchunk <= 256 * NV_NST && lo.n_levels >= 2;          // every level at most 48 planes: levels at own resolution
            for (int L = 0; L < lo.n_levels; ++L) native = native && lo.DL[L] <= NV_DL;
#ifdef GCS_KP_NO_NATIVE
            native = false;
#endif
            if (native) {
                // GCS_NV_MINB 4-wave workgroups per CU are resident: that many work, the others write zero partial rows
                // (four levels whose level 0 has fewer than 48 planes keep all eight staging addresses in a table: that variant
                // does not fit 168 VGPRs and runs with two workgroups per CU)
                const int minb = lo.n_levels == 4 && lo.DL[0] != NV_DL ? 2 : GCS_NV_MINB;
                const int parts_eff = native_parts_eff(B, parts, (long long)lo.ntiles * KP_TP, minb);
                const int nt_limit = kp_nt_limit(lo, B, n_sets, lo.tile_bytes);
#define GCS_NV_LAUNCH_(NL_, MINB_, N0_)                                                                                       \
    hipLaunchKernelGGL((kmeans_pass_native_kernel<NL_, MINB_, N0_>), dim3(B, parts), dim3(256), 0, stream,                   \
                       reinterpret_cast<const unsigned char *>(feats), cent, lo, k, n_sets == B ? 1 : 0, parts, parts_eff,  \
                       reverse ? 1 : 0, row_lo, row_hi, partials, lab_out, lab_u8, nt_limit)
#define GCS_NV_LAUNCH(NL_, MINB0_)                                          \
    do {                                                                    \
        if (lo.DL[0] == NV_DL) GCS_NV_LAUNCH_(NL_, GCS_NV_MINB, 6);         \
        else GCS_NV_LAUNCH_(NL_, MINB0_, 0);                                \
    } while (0)
                if (lo.n_levels == 2) GCS_NV_LAUNCH(2, GCS_NV_MINB);
                else if (lo.n_levels == 3) GCS_NV_LAUNCH(3, GCS_NV_MINB);
                else GCS_NV_LAUNCH(4, 2);
#undef GCS_NV_LAUNCH
#undef GCS_NV_LAUNCH_
            }
            // pyramid banks (config 4: 2 040 chunks per tile) fit 5 chunks per thread of an 8-wave workgroup without spills
            else if ((nchunk + 511) / 512 <= 5) { GCS_KP_LAUNCHW(1, 5, KP_DSTEPS_WIDE, 8); }
            else if (nst <= 18) { GCS_KP_LAUNCH(1, 18, KP_DSTEPS_WIDE); }
            else { GCS_KP_LAUNCH(1, 26, KP_DSTEPS_WIDE); }
        } else {
            if (nst <= 10) { GCS_KP_LAUNCH(2, 10, KP_DSTEPS_WIDE); }
            else if (nst <= 18) { GCS_KP_LAUNCH(2, 18, KP_DSTEPS_WIDE); }
            else { GCS_KP_LAUNCH(2, 26, KP_DSTEPS_WIDE); }
        }
#undef GCS_KP_LAUNCHW
#undef GCS_KP_LAUNCH
        GCS_CHECK_LAUNCH("gcs_kmeans_assign_accumulate");
        return GCS_OK;
    }
    // generic pass: uint8 raster labels only. A uint8 raster map is written directly, an int32 one through the scratch map.
    if (raster && raster_u8) labels = static_cast<uint8_t *>(raster);
    else if (raster && !labels)
        return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_raster: feature vectors of 208 or more planes need the scratch label map for int32 output");
    int rc = GCS_EINVAL;
    switch (k) { // generic VALU pass for wider feature vectors
#define GCS_CASE(KK) \
    case KK:         \
        rc = launch_assign<KK>(feats, cent, B, lo, n_sets, row_lo, row_hi, labels, partials, stream); \
        break;
        GCS_CASE(1) GCS_CASE(2) GCS_CASE(3) GCS_CASE(4) GCS_CASE(5) GCS_CASE(6) GCS_CASE(7) GCS_CASE(8)
        GCS_CASE(9) GCS_CASE(10) GCS_CASE(11) GCS_CASE(12) GCS_CASE(13) GCS_CASE(14) GCS_CASE(15) GCS_CASE(16)
#undef GCS_CASE
    }
    if (rc != GCS_OK || !raster || raster_u8) return rc;
    return gcs_labels_widen(labels, B, H, W, static_cast<int32_t *>(raster), stream);   // uint8 -> int32: one more launch
}

extern "C" int gcs_kmeans_assign_accumulate(const uint16_t *feats, const uint16_t *cent, int B, int H, int W,
                                            int n_scales, int n_orient, int k, int n_sets, int row_lo, int row_hi,
                                            int reverse, uint8_t *labels, uint64_t *partials, gcs_stream_t stream) {
    if (!labels && !partials)
        return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_accumulate: NULL pointer (labels and partials may not both be NULL)");
    return lloyd_pass(feats, cent, B, H, W, n_scales, n_orient, k, n_sets, row_lo, row_hi, reverse, labels, partials, nullptr, 0,
                      stream);
}

extern "C" int gcs_kmeans_assign_raster(const uint16_t *feats, const uint16_t *cent, int B, int H, int W, int n_scales,
                                        int n_orient, int k, int n_sets, int reverse, void *out, int out_u8,
                                        uint8_t *scratch_labels, gcs_stream_t stream) {
    if (!out) return gcs_fail(GCS_EINVAL, "gcs_kmeans_assign_raster: NULL pointer");
    return lloyd_pass(feats, cent, B, H, W, n_scales, n_orient, k, n_sets, 0, H, reverse, scratch_labels, nullptr, out,
                      out_u8 ? 1 : 0, stream);
}

// sums[set][e] = sum over the set's rows of element e (layout: partial_index in common.h). Integer sums: any order gives
// the same bits. One 1024-thread workgroup per chunk of 16 elements: thread (row group rg = t >> 4, element t & 15) adds
// rows rg, rg + 64, ...; a wave reads four whole rows = 512 contiguous bytes per load.
// FIN: the SPEC.md §4 update is applied in the same launch (single-rank case, no all-reduce in between): every thread
// also folds the count element of its element's cluster, so no second kernel and no cross-block dependency is needed.
// RG = row groups per workgroup (threads = 16 * RG): 64 for the long row lists of one global codebook, 16 for the `parts`
// rows of a per-image codebook.
template <bool FIN, int RG>
__global__ __launch_bounds__(16 * RG) void kmeans_reduce_kernel(const uint64_t *__restrict__ partials,
                                                             int rows_per_set, int row_len, int D1,
                                                             long long *__restrict__ sums,
                                                             uint16_t *__restrict__ cent) {
    __shared__ unsigned long long sm_s[RG][KP_PCH], sm_c[RG][KP_PCH];
    const int set = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    const int e16 = t & (KP_PCH - 1), rg = t >> 4;
    const int nch = partial_chunks(row_len);
    const int e = chunk * KP_PCH + e16;
    const int ee = e < row_len ? e : row_len - 1;              // padding columns of the last chunk: read a valid one
    const int j = ee / D1, d = ee - j * D1;
    const int ec = j * D1 + (D1 - 1);                          // the count element of this element's cluster
    const uint64_t *p = partials + (((size_t)set * nch + chunk) * rows_per_set) * KP_PCH + (ee - chunk * KP_PCH);
    const uint64_t *pc = partials + (((size_t)set * nch + ec / KP_PCH) * rows_per_set) * KP_PCH + ec % KP_PCH;
    uint64_t s = 0, c = 0;
    int r = rg;
    // the loop is latency-bound: 12 (then 4) rows in flight per thread
    auto burst = [&](auto n_c) {
        constexpr int N = decltype(n_c)::value;
        for (; r + RG * (N - 1) < rows_per_set; r += RG * N) {
            uint64_t a[N], q[N];
#pragma unroll
            for (int u = 0; u < N; ++u) {
                a[u] = p[(size_t)(r + RG * u) * KP_PCH];
                q[u] = FIN ? pc[(size_t)(r + RG * u) * KP_PCH] : 0;
            }
#pragma unroll
            for (int u = 0; u < N; ++u) {
                s += a[u];
                c += q[u];
            }
        }
    };
    burst(std::integral_constant<int, 12>{});
    burst(std::integral_constant<int, 4>{});
    burst(std::integral_constant<int, 1>{});
    sm_s[rg][e16] = s;
    if (FIN) sm_c[rg][e16] = c;
    __syncthreads();
    if (RG == 64) {                                            // 64 -> 16 row groups
        if (t < 256) {
            s = sm_s[4 * rg][e16] + sm_s[4 * rg + 1][e16] + sm_s[4 * rg + 2][e16] + sm_s[4 * rg + 3][e16];
            if (FIN) c = sm_c[4 * rg][e16] + sm_c[4 * rg + 1][e16] + sm_c[4 * rg + 2][e16] + sm_c[4 * rg + 3][e16];
        }
        __syncthreads();
        if (t < 256) {
            sm_s[rg][e16] = s;
            if (FIN) sm_c[rg][e16] = c;
        }
        __syncthreads();
    }
    if (t < KP_PCH && e < row_len) {
        s = 0;
        c = 0;
        for (int q = 0; q < 16; ++q) {
            s += sm_s[q][e16];
            if (FIN) c += sm_c[q][e16];
        }
        if (sums) sums[(size_t)set * row_len + e] = (long long)s;
        if (FIN && d < D1 - 1 && c > 0)
            cent[((size_t)set * (row_len / D1) + j) * (D1 - 1) + d] = (uint16_t)((2 * s + c) / (2 * c));
    }
}

static int reduce_args_ok(const void *partials, int B, int H, int W, int D, int k, int n_sets, const char *who) {
    if (!partials) return gcs_fail(GCS_EINVAL, "gcs_kmeans_reduce: NULL pointer");
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0 || k < 1 || k > GCS_K_MAX) return gcs_fail(GCS_EINVAL, who);
    if (n_sets != 1 && n_sets != B) return gcs_fail(GCS_EINVAL, "gcs_kmeans_reduce: n_sets must be 1 or B");
    return GCS_OK;
}

extern "C" int gcs_kmeans_reduce(const uint64_t *partials, int B, int H, int W, int D, int k, int n_sets,
                                 int64_t *sums, gcs_stream_t stream) {
    if (!sums) return gcs_fail(GCS_EINVAL, "gcs_kmeans_reduce: NULL pointer");
    if (int rc = reduce_args_ok(partials, B, H, W, D, k, n_sets, "gcs_kmeans_reduce: bad shape")) return rc;
    const int parts = (int)gcs_kmeans_parts_per_image(B, H, W);
    const int row_len = k * (D + 1);
    const int rows_per_set = n_sets == B ? parts : B * parts;
    if (rows_per_set > 64)
        hipLaunchKernelGGL((kmeans_reduce_kernel<false, 64>), dim3(partial_chunks(row_len), n_sets), dim3(1024), 0, stream,
                           partials, rows_per_set, row_len, D + 1, reinterpret_cast<long long *>(sums), (uint16_t *)nullptr);
    else
        hipLaunchKernelGGL((kmeans_reduce_kernel<false, 16>), dim3(partial_chunks(row_len), n_sets), dim3(256), 0, stream,
                           partials, rows_per_set, row_len, D + 1, reinterpret_cast<long long *>(sums), (uint16_t *)nullptr);
    GCS_CHECK_LAUNCH("gcs_kmeans_reduce");
    return GCS_OK;
}

extern "C" int gcs_kmeans_reduce_finalize(const uint64_t *partials, int B, int H, int W, int D, int k, int n_sets,
                                          int64_t *sums, uint16_t *cent, gcs_stream_t stream) {
    if (!cent) return gcs_fail(GCS_EINVAL, "gcs_kmeans_reduce_finalize: NULL pointer");
    if (int rc = reduce_args_ok(partials, B, H, W, D, k, n_sets, "gcs_kmeans_reduce_finalize: bad shape")) return rc;
    const int parts = (int)gcs_kmeans_parts_per_image(B, H, W);
    const int row_len = k * (D + 1);
    const int rows_per_set = n_sets == B ? parts : B * parts;
    if (rows_per_set > 64)
        hipLaunchKernelGGL((kmeans_reduce_kernel<true, 64>), dim3(partial_chunks(row_len), n_sets), dim3(1024), 0, stream,
                           partials, rows_per_set, row_len, D + 1, reinterpret_cast<long long *>(sums), cent);
    else
        hipLaunchKernelGGL((kmeans_reduce_kernel<true, 16>), dim3(partial_chunks(row_len), n_sets), dim3(256), 0, stream,
                           partials, rows_per_set, row_len, D + 1, reinterpret_cast<long long *>(sums), cent);
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
    if (!sums || !cent) return gcs_fail(GCS_EINVAL, "gcs_kmeans_finalize: NULL pointer");
    if (n_sets <= 0 || D <= 0 || k < 1 || k > GCS_K_MAX) return gcs_fail(GCS_EINVAL, "gcs_kmeans_finalize: bad shape");
    const int n = n_sets * k * D;
    hipLaunchKernelGGL(kmeans_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const long long *>(sums), n, k, D, cent);
    GCS_CHECK_LAUNCH("gcs_kmeans_finalize");
    return GCS_OK;
}

// ----------------------------------------------------------------------------- uint8 label map -> int32
__global__ __launch_bounds__(256) void labels_widen_kernel(const uint8_t *__restrict__ labels, size_t n, int32_t *__restrict__ out) {
    // four labels per thread where the dword is whole (the map starts on an allocation boundary: aligned)
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        const unsigned v = *reinterpret_cast<const unsigned *>(labels + i4);
        typedef int __attribute__((ext_vector_type(4), aligned(4))) v4i_a4;
        *reinterpret_cast<v4i_a4 *>(out + i4) = v4i_a4{(int)(v & 255u), (int)((v >> 8) & 255u), (int)((v >> 16) & 255u), (int)(v >> 24)};
    } else {
        for (size_t i = i4; i < n; ++i) out[i] = labels[i];
    }
}

extern "C" int gcs_labels_widen(const uint8_t *labels, int B, int H, int W, int32_t *out, gcs_stream_t stream) {
    if (!labels || !out || B <= 0 || H <= 0 || W <= 0) return gcs_fail(GCS_EINVAL, "gcs_labels_widen: bad argument");
    if ((reinterpret_cast<uintptr_t>(labels) & 3) || (reinterpret_cast<uintptr_t>(out) & 3))
        return gcs_fail(GCS_EINVAL, "gcs_labels_widen: pointers must be 4-byte aligned");
    const size_t n = (size_t)B * H * W;
    const size_t blocks = (n / 4 + 1 + 255) / 256;
    if (blocks > 0x7fffffffull) return gcs_fail(GCS_EINVAL, "gcs_labels_widen: map too large for one launch");
    hipLaunchKernelGGL(labels_widen_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, labels, n, out);
    GCS_CHECK_LAUNCH("gcs_labels_widen");
    return GCS_OK;
}
