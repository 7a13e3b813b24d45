// scoring.hip — evaluation-side kernels of libgcs.so (gfx950): boundary recall / precision counts
// (/root/reference/BSD_metrics/metrics.py:58-96), region tables (metrics.py:102-201) and connected regions (SPEC.md §7).
// Nothing here allocates, frees or synchronises; every entry point enqueues on the caller's stream.
#include "common.h"

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

// Batched form: B label maps [B][H][W] and T annotator maps [T][H][W] (all annotators of all images, image after image);
// img_of[t] = image of annotator t (NULL: every annotator belongs to image 0, the single-image call).
// maps: planes 0..B-1 = boundaries of the label maps, planes B..B+T-1 = boundaries of the annotator maps
__global__ void boundary_maps_kernel(const int32_t *__restrict__ labels, const uint16_t *__restrict__ truth, int B, int T,
                                     int H, int W, uint8_t *__restrict__ maps) {
    const size_t n = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(B + T) * n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i / n), p = (int)(i % n), y = p / W, x = p % W;
        maps[i] = q < B ? thick_boundary(labels + (size_t)q * n, H, W, y, x)
                        : thick_boundary(truth + (size_t)(q - B) * n, H, W, y, x);
    }
}

// counts: [b] = sum bd(L_b) for b < B; then per annotator t: [B + 3t] = sum dil5(bd(L_b)) & bd(T_t)   (recall numerator),
// [B + 3t + 1] = sum bd(T_t) (recall denominator), [B + 3t + 2] = sum bd(L_b) & dil5(bd(T_t)) (precision numerator),
// b = img_of[t]. With B = 1 this is the single-image layout [1 + 3A].
// One workgroup per (16 x 64 pixel tile, plane q): both boundary planes of the tile go to LDS with their two-pixel halo
// (pixels outside the image count as "no boundary", as dilation(., rectangle(5,5)) treats its border), the 5 x 5 dilation
// is evaluated separably (5 horizontal ORs into LDS, 5 vertical ORs per pixel) and the three sums leave the workgroup as
// one atomic each. Round 3 walked the 25 taps in global memory with a division per pixel: 336 us for 16 BSD images and
// their 87 annotator maps, 77 % of a segment + score loop (profiles/r4_notes.md).
constexpr int BC_TH = 16, BC_TW = 64, BC_HALO = 2;
__global__ __launch_bounds__(256) void boundary_counts_kernel(const uint8_t *__restrict__ maps, const int32_t *__restrict__ img_of,
                                                              int B, int T, int H, int W, unsigned long long *__restrict__ counts) {
    __shared__ uint8_t s_raw[2][BC_TH + 2 * BC_HALO][BC_TW + 2 * BC_HALO + 4];   // [label | annotator] boundary bits with halo
    __shared__ uint8_t s_hor[2][BC_TH + 2 * BC_HALO][BC_TW];                     // OR over the 5 horizontal neighbours
    __shared__ unsigned s_sum[3];
    const size_t n = (size_t)H * W;
    const int q = blockIdx.z;                         // < B: label-only count of image q; else annotator q - B
    const int b = q < B ? q : (img_of ? img_of[q - B] : 0);
    const uint8_t *lb = maps + (size_t)b * n;
    const uint8_t *tb = maps + (size_t)q * n;         // annotator plane (q >= B)
    const int x0 = blockIdx.x * BC_TW, y0 = blockIdx.y * BC_TH, tid = threadIdx.x;
    unsigned c0 = 0, c1 = 0, c2 = 0;
    if (tid < 3) s_sum[tid] = 0;
    if (q < B) {                                      // sum bd(L_q): no stencil
        for (int i = tid; i < BC_TH * BC_TW; i += 256) {
            const int y = y0 + i / BC_TW, x = x0 + i % BC_TW;
            if (y < H && x < W) c0 += lb[(size_t)y * W + x];
        }
    } else {
        constexpr int RW = BC_TW + 2 * BC_HALO, RH = BC_TH + 2 * BC_HALO;
        for (int i = tid; i < RH * RW; i += 256) {
            const int r = i / RW, c = i % RW, y = y0 + r - BC_HALO, x = x0 + c - BC_HALO;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            s_raw[0][r][c] = in ? lb[(size_t)y * W + x] : (uint8_t)0;
            s_raw[1][r][c] = in ? tb[(size_t)y * W + x] : (uint8_t)0;
        }
        __syncthreads();
        for (int i = tid; i < RH * BC_TW; i += 256) {
            const int r = i / BC_TW, c = i % BC_TW;
#pragma unroll
            for (int m = 0; m < 2; ++m)
                s_hor[m][r][c] = s_raw[m][r][c] | s_raw[m][r][c + 1] | s_raw[m][r][c + 2] | s_raw[m][r][c + 3] | s_raw[m][r][c + 4];
        }
        __syncthreads();
        for (int i = tid; i < BC_TH * BC_TW; i += 256) {
            const int r = i / BC_TW, c = i % BC_TW;
            if (y0 + r < H && x0 + c < W) {
                const unsigned bl = s_raw[0][r + BC_HALO][c + BC_HALO], bt = s_raw[1][r + BC_HALO][c + BC_HALO];
                const unsigned dl = s_hor[0][r][c] | s_hor[0][r + 1][c] | s_hor[0][r + 2][c] | s_hor[0][r + 3][c] | s_hor[0][r + 4][c];
                const unsigned dt = s_hor[1][r][c] | s_hor[1][r + 1][c] | s_hor[1][r + 2][c] | s_hor[1][r + 3][c] | s_hor[1][r + 4][c];
                c0 += bt & dl;                        // recall numerator
                c1 += bt;                             // recall denominator
                c2 += bl & dt;                        // precision numerator
            }
        }
    }
    // wave reduction, then one atomic per sum and workgroup (integers: order-independent)
    for (int m = 32; m >= 1; m >>= 1) {
        c0 += __shfl_xor(c0, m);
        c1 += __shfl_xor(c1, m);
        c2 += __shfl_xor(c2, m);
    }
    __syncthreads();
    if ((tid & 63) == 0) {
        atomicAdd(&s_sum[0], c0);
        atomicAdd(&s_sum[1], c1);
        atomicAdd(&s_sum[2], c2);
    }
    __syncthreads();
    if (tid == 0) {
        if (q < B) {
            if (s_sum[0]) atomicAdd(&counts[q], (unsigned long long)s_sum[0]);
        } else {
            if (s_sum[0]) atomicAdd(&counts[B + 3 * (q - B)], (unsigned long long)s_sum[0]);
            if (s_sum[1]) atomicAdd(&counts[B + 3 * (q - B) + 1], (unsigned long long)s_sum[1]);
            if (s_sum[2]) atomicAdd(&counts[B + 3 * (q - B) + 2], (unsigned long long)s_sum[2]);
        }
    }
}

extern "C" size_t gcs_boundary_scratch_bytes(int A, int H, int W) {
    if (A <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)(A + 1) * H * W;
}
extern "C" size_t gcs_boundary_batch_scratch_bytes(int B, int T, int H, int W) {
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)(B + T) * H * W;
}

static int boundary_counts_launch(const int32_t *labels, const uint16_t *truth, const int32_t *img_of, int B, int T, int H,
                                  int W, void *scratch, uint64_t *counts, hipStream_t stream, const char *who) {
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)(B + 3 * T) * sizeof(uint64_t), stream);
    if (e != hipSuccess) return gcs_hip_fail(e, who);
    uint8_t *maps = static_cast<uint8_t *>(scratch);
    const int n = H * W;
    const size_t total = (size_t)(B + T) * n;
    hipLaunchKernelGGL(boundary_maps_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)),
                       dim3(256), 0, stream, labels, truth, B, T, H, W, maps);
    GCS_CHECK_LAUNCH(who);
    hipLaunchKernelGGL(boundary_counts_kernel, dim3((W + BC_TW - 1) / BC_TW, (H + BC_TH - 1) / BC_TH, B + T), dim3(256), 0, stream,
                       maps, img_of, B, T, H, W, reinterpret_cast<unsigned long long *>(counts));
    GCS_CHECK_LAUNCH(who);
    return GCS_OK;
}

extern "C" int gcs_boundary_counts(const int32_t *labels, const uint16_t *truth, int A, int H, int W, void *scratch,
                                   uint64_t *counts, gcs_stream_t stream) {
    if (!labels || !truth || !scratch || !counts) return gcs_fail(GCS_EINVAL, "gcs_boundary_counts: NULL pointer");
    if (A <= 0 || A > 65534 || H <= 0 || W <= 0 || (long long)H * W * (A + 1) > 0x7fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_boundary_counts: bad shape");
    return boundary_counts_launch(labels, truth, nullptr, 1, A, H, W, scratch, counts, stream, "gcs_boundary_counts");
}

extern "C" int gcs_boundary_counts_batch(const int32_t *labels, const uint16_t *truth, const int32_t *img_of, int B, int T,
                                         int H, int W, void *scratch, uint64_t *counts, gcs_stream_t stream) {
    if (!labels || !truth || !img_of || !scratch || !counts)
        return gcs_fail(GCS_EINVAL, "gcs_boundary_counts_batch: NULL pointer");
    if (B <= 0 || T <= 0 || B + T > 65535 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_boundary_counts_batch: bad shape");
    return boundary_counts_launch(labels, truth, img_of, B, T, H, W, scratch, counts, stream, "gcs_boundary_counts_batch");
}

// ============================================================ boundary scoring on resident ground truth (round 5)
// The annotator maps are constants of the data set (/root/reference/BSD_metrics/metrics.py:48-49 recomputes
// find_boundaries(truth) for every image it scores, groundtruth.py:44-48 rescans the directories per id): their thick boundaries
// bd(T_t), the 5x5 dilation dil5(bd(T_t)) and sum bd(T_t) are computed ONCE per annotator map (gcs_truth_prepare) and kept on the
// device as BIT planes, rows of 64-bit words (bit i of word w of a row = pixel 64 w + i), all bd planes first, then all dil5
// planes. A scoring call derives the two bit planes of each label map and the three sums are AND + popcount over 2 568 words per
// plane (321 x 481): kilobytes, where round 4 uploaded 40 MB of annotator maps per 24 images and re-derived 129 byte planes.
__host__ __device__ static inline int bits_wp(int W) { return (W + 63) / 64; }

// bd bit planes of M maps: one wave per 64 consecutive pixels of a row (coalesced), the plane word by ballot. `smax`: per-map
// maximum value (metrics.py:51 for the label maps; zeroed by the caller), or NULL.
template <typename T>
__global__ __launch_bounds__(256) void bits_boundary_kernel(const T *__restrict__ maps, int M, int H, int W,
                                                            unsigned long long *__restrict__ bd, int *__restrict__ smax) {
    const int wp = bits_wp(W);
    const long long nw = (long long)M * H * wp;
    const int lane = threadIdx.x & 63;
    for (long long wi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); wi < nw; wi += (long long)gridDim.x * 4) {
        const int wx = (int)(wi % wp), y = (int)((wi / wp) % H), m = (int)(wi / ((long long)wp * H));
        const int x = wx * 64 + lane;
        const T *mp = maps + (size_t)m * H * W;
        bool b = false;
        int v = 0;
        if (x < W) {
            b = thick_boundary(mp, H, W, y, x);
            v = (int)mp[(size_t)y * W + x];
        }
        const unsigned long long word = __ballot(b);
        if (lane == 0) bd[wi] = word;
        if (smax) {
            // one atomic per wave only while it can still raise the maximum: unconditional, 41 000 waves hammered 16 addresses and
            // this kernel took 305 us for 16 BSD label maps (6 us without)
            for (int s = 32; s >= 1; s >>= 1) v = max(v, __shfl_xor(v, s));
            if (lane == 0 && v > __hip_atomic_load(&smax[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&smax[m], v);
        }
    }
}

// dil5 bit planes: OR over the 5 x 5 neighbourhood; pixels outside the image count as 0 (dilation(., rectangle(5,5)) through
// scipy's max filter with its reflect border sees in-image values only)
__global__ __launch_bounds__(256) void bits_dilate_kernel(const unsigned long long *__restrict__ bd, int M, int H, int W,
                                                          unsigned long long *__restrict__ dil) {
    const int wp = bits_wp(W);
    const long long nw = (long long)M * H * wp;
    const unsigned long long last_mask = (W & 63) ? ((1ull << (W & 63)) - 1) : ~0ull;
    for (long long wi = (long long)blockIdx.x * blockDim.x + threadIdx.x; wi < nw; wi += (long long)gridDim.x * blockDim.x) {
        const int wx = (int)(wi % wp), y = (int)((wi / wp) % H);
        const unsigned long long *row0 = bd + (wi - wx) - (long long)y * wp;    // row 0 of this map
        unsigned long long acc = 0;
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            const unsigned long long *r = row0 + (long long)yy * wp;
            const unsigned long long w = r[wx], pv = wx > 0 ? r[wx - 1] : 0ull, nx = wx + 1 < wp ? r[wx + 1] : 0ull;
            acc |= w | (w << 1) | (w << 2) | (w >> 1) | (w >> 2) | (pv >> 63) | (pv >> 62) | (nx << 63) | (nx << 62);
        }
        dil[wi] = wx == wp - 1 ? acc & last_mask : acc;
    }
}

__global__ __launch_bounds__(256) void bits_popcount_kernel(const unsigned long long *__restrict__ planes, int words,
                                                            unsigned long long *__restrict__ out) {
    __shared__ unsigned s_sum[4];
    const unsigned long long *p = planes + (size_t)blockIdx.x * words;
    unsigned c = 0;
    for (int i = threadIdx.x; i < words; i += 256) c += __popcll(p[i]);
    for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// 5 x 5 dilation of one word of a bd plane (rows of `wp` words, H rows), see bits_dilate_kernel
__device__ __forceinline__ unsigned long long bits_dilate_word(const unsigned long long *__restrict__ plane, int H, int wp, int y,
                                                               int wx, unsigned long long last_mask) {
    unsigned long long acc = 0;
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const unsigned long long *r = plane + (long long)yy * wp;
        const unsigned long long w = r[wx], pv = wx > 0 ? r[wx - 1] : 0ull, nx = wx + 1 < wp ? r[wx + 1] : 0ull;
        acc |= w | (w << 1) | (w << 2) | (w >> 1) | (w >> 2) | (pv >> 63) | (pv >> 62) | (nx << 63) | (nx << 62);
    }
    return wx == wp - 1 ? acc & last_mask : acc;
}

// One workgroup per output plane q (no atomics, no zeroing): q < B: counts[q] = sum bd(L_q); else annotator t = q - B of image
// b = img_of[t]: counts[B + 3t] = sum dil5(bd(L_b)) & bd(T_t), [B + 3t + 1] = sum bd(T_t) (prepared), [B + 3t + 2] = sum bd(L_b) &
// dil5(bd(T_t)). Annotator planes [2][T][words] (bd, then dil5). Label planes: [2][B][words], or (INLINE) the bd planes alone,
// dilated word by word right here (a label plane is dilated once per annotator of its image, 15 loads instead of one from a few
// KB that sit in L2: cheaper than the launch it saves).
template <bool INLINE>   // INLINE: dilate the label planes word by word here (one launch less; measured 28 us against 5 + 8: not used)
__global__ __launch_bounds__(256) void bits_counts_kernel(const unsigned long long *__restrict__ lab,
                                                          const unsigned long long *__restrict__ tru,
                                                          const unsigned long long *__restrict__ tru_bd_counts,
                                                          const int32_t *__restrict__ img_of, int B, int T, int H, int W,
                                                          unsigned long long *__restrict__ counts) {
    __shared__ unsigned s_sum[2][4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int wp = bits_wp(W), words = H * wp;
    const unsigned long long last_mask = (W & 63) ? ((1ull << (W & 63)) - 1) : ~0ull;
    unsigned c0 = 0, c1 = 0;
    if (q < B) {
        const unsigned long long *l = lab + (size_t)q * words;
        for (int i = tid; i < words; i += 256) c0 += __popcll(l[i]);
    } else {
        const int t = q - B, b = img_of[t];
        const unsigned long long *lbd = lab + (size_t)b * words, *ldil = lab + ((size_t)B + b) * words;
        const unsigned long long *tbd = tru + (size_t)t * words, *tdil = tru + ((size_t)T + t) * words;
        for (int i = tid; i < words; i += 256) {
            const unsigned long long ld = INLINE ? bits_dilate_word(lbd, H, wp, i / wp, i % wp, last_mask) : ldil[i];
            c0 += __popcll(ld & tbd[i]);
            c1 += __popcll(lbd[i] & tdil[i]);
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        c0 += __shfl_xor(c0, s);
        c1 += __shfl_xor(c1, s);
    }
    if ((tid & 63) == 0) {
        s_sum[0][tid >> 6] = c0;
        s_sum[1][tid >> 6] = c1;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long r0 = (unsigned long long)s_sum[0][0] + s_sum[0][1] + s_sum[0][2] + s_sum[0][3];
        const unsigned long long r1 = (unsigned long long)s_sum[1][0] + s_sum[1][1] + s_sum[1][2] + s_sum[1][3];
        if (q < B) counts[q] = r0;
        else {
            counts[B + 3 * (q - B)] = r0;
            counts[B + 3 * (q - B) + 1] = tru_bd_counts[q - B];
            counts[B + 3 * (q - B) + 2] = r1;
        }
    }
}

// several small buffers zeroed in ONE launch (a hipMemsetAsync each is a launch each)
struct ZeroList { unsigned *p[4]; unsigned n[4]; };
__global__ void zero_kernel(ZeroList z) {
    for (int k = 0; k < 4; ++k)
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < z.n[k]; i += gridDim.x * blockDim.x) z.p[k][i] = 0u;
}

__global__ void narrow_u16_u8_kernel(const uint16_t *__restrict__ in, size_t n, uint8_t *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (uint8_t)in[i];
}

static inline unsigned grid_for(long long items, int per_block, unsigned cap) {
    const long long g = (items + per_block - 1) / per_block;
    return (unsigned)(g < 1 ? 1 : g > cap ? cap : g);
}

extern "C" size_t gcs_bit_planes_bytes(int M, int H, int W) {
    if (M <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)M * 2 * H * bits_wp(W) * sizeof(unsigned long long);
}

extern "C" int gcs_truth_prepare(const uint16_t *truth, int T, int H, int W, void *planes, uint64_t *bd_counts, uint8_t *truth8,
                                 gcs_stream_t stream) {
    if (!truth || !planes || !bd_counts) return gcs_fail(GCS_EINVAL, "gcs_truth_prepare: NULL pointer");
    if (T <= 0 || T > 1000000 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_truth_prepare: bad shape");
    const int words = H * bits_wp(W);
    const long long nw = (long long)T * words;
    unsigned long long *bd = static_cast<unsigned long long *>(planes), *dil = bd + nw;
    hipLaunchKernelGGL(bits_boundary_kernel<uint16_t>, dim3(grid_for(nw, 4, 16384)), dim3(256), 0, stream, truth, T, H, W, bd,
                       (int *)nullptr);
    GCS_CHECK_LAUNCH("gcs_truth_prepare(boundaries)");
    hipLaunchKernelGGL(bits_dilate_kernel, dim3(grid_for(nw, 256, 16384)), dim3(256), 0, stream, bd, T, H, W, dil);
    GCS_CHECK_LAUNCH("gcs_truth_prepare(dilation)");
    hipLaunchKernelGGL(bits_popcount_kernel, dim3(T), dim3(256), 0, stream, bd, words, reinterpret_cast<unsigned long long *>(bd_counts));
    GCS_CHECK_LAUNCH("gcs_truth_prepare(counts)");
    if (truth8) {
        const size_t n = (size_t)T * H * W;
        hipLaunchKernelGGL(narrow_u16_u8_kernel, dim3(grid_for((long long)n, 256, 8192)), dim3(256), 0, stream, truth, n, truth8);
        GCS_CHECK_LAUNCH("gcs_truth_prepare(narrow)");
    }
    return GCS_OK;
}

extern "C" int gcs_boundary_counts_resident(const int32_t *labels, const void *truth_planes, const uint64_t *truth_bd_counts,
                                            const int32_t *img_of, int B, int T, int H, int W, void *scratch, uint64_t *counts,
                                            int32_t *seg_max, gcs_stream_t stream) {
    if (!labels || !truth_planes || !truth_bd_counts || !img_of || !scratch || !counts)
        return gcs_fail(GCS_EINVAL, "gcs_boundary_counts_resident: NULL pointer");
    if (B <= 0 || T <= 0 || B + T > 1000000 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_boundary_counts_resident: bad shape");
    const int words = H * bits_wp(W);
    const long long nw = (long long)B * words;
    unsigned long long *bd = static_cast<unsigned long long *>(scratch), *dil = bd + nw;
    if (seg_max) {
        hipError_t e = hipMemsetAsync(seg_max, 0, (size_t)B * sizeof(int32_t), stream);
        if (e != hipSuccess) return gcs_hip_fail(e, "gcs_boundary_counts_resident(memset)");
    }
    hipLaunchKernelGGL(bits_boundary_kernel<int32_t>, dim3(grid_for(nw, 4, 16384)), dim3(256), 0, stream, labels, B, H, W, bd, seg_max);
    GCS_CHECK_LAUNCH("gcs_boundary_counts_resident(boundaries)");
    hipLaunchKernelGGL(bits_dilate_kernel, dim3(grid_for(nw, 256, 16384)), dim3(256), 0, stream, bd, B, H, W, dil);
    GCS_CHECK_LAUNCH("gcs_boundary_counts_resident(dilation)");
    hipLaunchKernelGGL(bits_counts_kernel<false>, dim3(B + T), dim3(256), 0, stream, bd, static_cast<const unsigned long long *>(truth_planes),
                       reinterpret_cast<const unsigned long long *>(truth_bd_counts), img_of, B, T, H, W,
                       reinterpret_cast<unsigned long long *>(counts));
    GCS_CHECK_LAUNCH("gcs_boundary_counts_resident");
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
    if (!labels || !scratch || !out) return gcs_fail(GCS_EINVAL, "gcs_connected_regions: NULL pointer");
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_connected_regions: bad shape");
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
// Batched: blockIdx.y = image b with annotators first[b] .. first[b+1]-1 of the concatenated truth stack (first == NULL:
// one image with annotators 0 .. A-1); hist [T][n_seg][stride], area / perim [B][n_seg].
template <typename TT>
__global__ __launch_bounds__(256) void region_counts_kernel(const int32_t *__restrict__ labels,
                                                            const TT *__restrict__ truth,
                                                            const int32_t *__restrict__ first, int A, int H, int W,
                                                            int n_seg, int stride, int use_lds,
                                                            unsigned *__restrict__ hist, unsigned *__restrict__ area,
                                                            unsigned *__restrict__ perim) {
    extern __shared__ unsigned s_tab[];                        // [A_b][n_seg][stride] hist | [n_seg] area | [n_seg] perim
    const int b = blockIdx.y;
    const int t0 = first ? first[b] : 0, a_n = first ? first[b + 1] - first[b] : A;
    const int P = H * W;
    labels += (size_t)b * P;
    truth += (size_t)t0 * P;
    hist += (size_t)t0 * n_seg * stride;
    area += (size_t)b * n_seg;
    perim += (size_t)b * n_seg;
    const int n_hist = a_n * n_seg * stride, n_tab = n_hist + 2 * n_seg;
    // the dynamic LDS was sized by the host for A annotators per image: an image that brings more (a caller that understated
    // max_annotators) takes the global-atomics path instead of writing past its allocation
    use_lds = use_lds && a_n <= A;
    if (use_lds) {
        for (int i = threadIdx.x; i < n_tab; i += blockDim.x) s_tab[i] = 0u;
        __syncthreads();
    }
    unsigned *t_hist = use_lds ? s_tab : hist, *t_area = use_lds ? s_tab + n_hist : area,
             *t_perim = use_lds ? s_tab + n_hist + n_seg : perim;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int l = labels[p];
        if ((unsigned)l >= (unsigned)n_seg) continue;          // caller passes n_seg = max + 1; never index outside
        const int y = p / W, x = p - y * W;
        bool edge = y == 0 || y == H - 1 || x == 0 || x == W - 1;
        if (!edge) edge = labels[p - W] != l || labels[p + W] != l || labels[p - 1] != l || labels[p + 1] != l;
        atomicAdd(&t_area[l], 1u);
        if (edge) atomicAdd(&t_perim[l], 1u);
        for (int a = 0; a < a_n; ++a) {
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

template <typename TT>
static int region_counts_launch(const int32_t *labels, const TT *truth, const int32_t *first, int B, int T, int Amax,
                                int H, int W, int n_segments, int n_truth_labels, uint32_t *hist, uint32_t *area,
                                uint32_t *perim, hipStream_t stream, const char *who, bool zeroed = false) {
    const size_t n_hist = (size_t)T * n_segments * n_truth_labels;
    if (!zeroed) {
        hipError_t e = hipMemsetAsync(hist, 0, n_hist * sizeof(uint32_t), stream);
        if (e == hipSuccess) e = hipMemsetAsync(area, 0, (size_t)B * n_segments * sizeof(uint32_t), stream);
        if (e == hipSuccess) e = hipMemsetAsync(perim, 0, (size_t)B * n_segments * sizeof(uint32_t), stream);
        if (e != hipSuccess) return gcs_hip_fail(e, who);
    }
    const size_t lds = ((size_t)Amax * n_segments * n_truth_labels + 2 * (size_t)n_segments) * sizeof(unsigned);
    const int use_lds = lds <= 48 * 1024;
    const int P = H * W;
    int blocks = use_lds ? min(256, (P + 1023) / 1024) : min(2048, (P + 255) / 256);
    if (B > 16) blocks = min(blocks, 32);
    hipLaunchKernelGGL(region_counts_kernel<TT>, dim3(blocks, B), dim3(256), use_lds ? lds : 0, stream, labels, truth, first,
                       Amax, H, W, n_segments, n_truth_labels, use_lds, hist, area, perim);
    GCS_CHECK_LAUNCH(who);
    return GCS_OK;
}

extern "C" int gcs_region_counts(const int32_t *labels, const uint16_t *truth, int A, int H, int W, int n_segments,
                                 int n_truth_labels, uint32_t *hist, uint32_t *area, uint32_t *perim,
                                 gcs_stream_t stream) {
    if (!labels || !truth || !hist || !area || !perim) return gcs_fail(GCS_EINVAL, "gcs_region_counts: NULL pointer");
    if (A <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL || n_segments <= 0 || n_truth_labels <= 0 ||
        (long long)A * n_segments * n_truth_labels > 0x3fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_region_counts: bad shape");
    return region_counts_launch(labels, truth, nullptr, 1, A, A, H, W, n_segments, n_truth_labels, hist, area, perim, stream,
                                "gcs_region_counts");
}

extern "C" int gcs_region_counts_batch(const int32_t *labels, const uint16_t *truth, const int32_t *first, int B, int T,
                                       int max_annotators, int H, int W, int n_segments, int n_truth_labels, uint32_t *hist,
                                       uint32_t *area, uint32_t *perim, gcs_stream_t stream) {
    if (!labels || !truth || !first || !hist || !area || !perim)
        return gcs_fail(GCS_EINVAL, "gcs_region_counts_batch: NULL pointer");
    if (B <= 0 || B > 65535 || T <= 0 || max_annotators <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL ||
        n_segments <= 0 || n_truth_labels <= 0 || (long long)T * n_segments * n_truth_labels > 0x3fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_region_counts_batch: bad shape");
    return region_counts_launch(labels, truth, first, B, T, max_annotators, H, W, n_segments, n_truth_labels, hist, area,
                                perim, stream, "gcs_region_counts_batch");
}

// The two integer sums metrics.py:128-140 takes from an annotator's contingency table, one workgroup per annotator map t of image
// b = img_of[t]:  under[t] = sum_seg (area[b][seg] - max_col hist[t][seg][col])                       (metrics.py:129-130)
//                 under_np[t] = sum_seg sum_col min(hist[t][seg][col], rowsum[t][seg] - hist[t][seg][col])   (metrics.py:137-139)
// Integers: exact in any order. The host then needs 16 bytes per annotator instead of the table (109 x 8 x 114 counters for 16 BSD
// images) and its array passes over it.
__global__ __launch_bounds__(256) void region_reduce_kernel(const unsigned *__restrict__ hist, const unsigned *__restrict__ area,
                                                            const int32_t *__restrict__ img_of, int n_seg, int stride,
                                                            unsigned long long *__restrict__ under,
                                                            unsigned long long *__restrict__ under_np) {
    __shared__ unsigned long long s_u[4], s_n[4];
    const int t = blockIdx.x, b = img_of[t], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned *h = hist + (size_t)t * n_seg * stride;
    unsigned long long u = 0, n = 0;                        // (lane 0 of each wave carries the wave's share)
    for (int seg = wave; seg < n_seg; seg += 4) {           // one wave per table row, lanes across its columns
        const unsigned *row = h + (size_t)seg * stride;
        unsigned mx = 0;
        unsigned long long sum = 0;
        for (int c = lane; c < stride; c += 64) {
            const unsigned v = row[c];
            mx = v > mx ? v : mx;
            sum += v;
        }
        for (int s = 32; s >= 1; s >>= 1) {
            mx = max(mx, (unsigned)__shfl_xor((int)mx, s));
            sum += __shfl_xor(sum, s);
        }
        unsigned long long mn = 0;
        for (int c = lane; c < stride; c += 64) {
            const unsigned long long v = row[c], o = sum - v;
            mn += v < o ? v : o;
        }
        for (int s = 32; s >= 1; s >>= 1) mn += __shfl_xor(mn, s);
        u += (unsigned long long)area[(size_t)b * n_seg + seg] - mx;    // area >= row sum >= max: never negative
        n += mn;
    }
    if (lane == 0) {
        s_u[wave] = u;
        s_n[wave] = n;
    }
    __syncthreads();
    if (tid == 0) {
        under[t] = s_u[0] + s_u[1] + s_u[2] + s_u[3];
        under_np[t] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
    }
}

extern "C" int gcs_region_reduce(const uint32_t *hist, const uint32_t *area, const int32_t *img_of, int T, int n_segments,
                                 int n_truth_labels, uint64_t *under, uint64_t *under_np, gcs_stream_t stream) {
    if (!hist || !area || !img_of || !under || !under_np) return gcs_fail(GCS_EINVAL, "gcs_region_reduce: NULL pointer");
    if (T <= 0 || T > 1000000 || n_segments <= 0 || n_truth_labels <= 0) return gcs_fail(GCS_EINVAL, "gcs_region_reduce: bad shape");
    hipLaunchKernelGGL(region_reduce_kernel, dim3(T), dim3(256), 0, stream, hist, area, img_of, n_segments, n_truth_labels,
                       reinterpret_cast<unsigned long long *>(under), reinterpret_cast<unsigned long long *>(under_np));
    GCS_CHECK_LAUNCH("gcs_region_reduce");
    return GCS_OK;
}

// the same tables from annotator maps narrowed to uint8 (gcs_truth_prepare: BSD500's largest annotator label is 208), half the bytes
extern "C" int gcs_region_counts_batch_u8(const int32_t *labels, const uint8_t *truth8, const int32_t *first, int B, int T,
                                          int max_annotators, int H, int W, int n_segments, int n_truth_labels, uint32_t *hist,
                                          uint32_t *area, uint32_t *perim, gcs_stream_t stream) {
    if (!labels || !truth8 || !first || !hist || !area || !perim)
        return gcs_fail(GCS_EINVAL, "gcs_region_counts_batch_u8: NULL pointer");
    if (B <= 0 || B > 65535 || T <= 0 || max_annotators <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL ||
        n_segments <= 0 || n_truth_labels <= 0 || n_truth_labels > 256 || (long long)T * n_segments * n_truth_labels > 0x3fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_region_counts_batch_u8: bad shape");
    return region_counts_launch(labels, truth8, first, B, T, max_annotators, H, W, n_segments, n_truth_labels, hist, area,
                                perim, stream, "gcs_region_counts_batch_u8");
}

// Everything evaluate.metrics.get_metrics() needs of a batch, on resident ground truth, in SIX launches: zero | bd(L) bit planes +
// label maxima | their dilation | boundary counts | region tables | their reduction.
extern "C" int gcs_score_batch_resident(const int32_t *labels, const void *truth_planes, const uint64_t *truth_bd_counts,
                                        const void *truth_maps, int truth_is_u8, const int32_t *first, const int32_t *img_of, int B,
                                        int T, int max_annotators, int H, int W, int n_segments, int n_truth_labels, void *scratch,
                                        uint32_t *hist, uint64_t *counts, int32_t *seg_max, uint32_t *area, uint32_t *perim,
                                        uint64_t *under, uint64_t *under_np, gcs_stream_t stream) {
    if (!labels || !truth_planes || !truth_bd_counts || !truth_maps || !first || !img_of || !scratch || !hist || !counts || !seg_max ||
        !area || !perim || !under || !under_np)
        return gcs_fail(GCS_EINVAL, "gcs_score_batch_resident: NULL pointer");
    if (B <= 0 || B > 65535 || T <= 0 || B + T > 1000000 || max_annotators <= 0 || H <= 0 || W <= 0 ||
        (long long)H * W > 0x7fffffffLL || n_segments <= 0 || n_truth_labels <= 0 || (truth_is_u8 && n_truth_labels > 256) ||
        (long long)T * n_segments * n_truth_labels > 0x3fffffffLL || (long long)B * n_segments > 0x3fffffffLL)
        return gcs_fail(GCS_EINVAL, "gcs_score_batch_resident: bad shape");
    const int words = H * bits_wp(W);
    const long long nw = (long long)B * words;
    ZeroList z;
    z.p[0] = hist; z.n[0] = (unsigned)((size_t)T * n_segments * n_truth_labels);
    z.p[1] = area; z.n[1] = (unsigned)((size_t)B * n_segments);       // (B * n_segments < 2^30: checked above)
    z.p[2] = perim; z.n[2] = (unsigned)((size_t)B * n_segments);
    z.p[3] = reinterpret_cast<unsigned *>(seg_max); z.n[3] = (unsigned)B;
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for((long long)z.n[0], 256, 1024)), dim3(256), 0, stream, z);
    GCS_CHECK_LAUNCH("gcs_score_batch_resident(zero)");
    unsigned long long *bd = static_cast<unsigned long long *>(scratch);
    hipLaunchKernelGGL(bits_boundary_kernel<int32_t>, dim3(grid_for(nw, 4, 16384)), dim3(256), 0, stream, labels, B, H, W, bd, seg_max);
    GCS_CHECK_LAUNCH("gcs_score_batch_resident(boundaries)");
    hipLaunchKernelGGL(bits_dilate_kernel, dim3(grid_for(nw, 256, 16384)), dim3(256), 0, stream, bd, B, H, W, bd + nw);
    GCS_CHECK_LAUNCH("gcs_score_batch_resident(dilation)");
    hipLaunchKernelGGL(bits_counts_kernel<false>, dim3(B + T), dim3(256), 0, stream, bd, static_cast<const unsigned long long *>(truth_planes),
                       reinterpret_cast<const unsigned long long *>(truth_bd_counts), img_of, B, T, H, W,
                       reinterpret_cast<unsigned long long *>(counts));
    GCS_CHECK_LAUNCH("gcs_score_batch_resident(counts)");
    int rc = truth_is_u8 ? region_counts_launch(labels, static_cast<const uint8_t *>(truth_maps), first, B, T, max_annotators, H, W,
                                                n_segments, n_truth_labels, hist, area, perim, stream, "gcs_score_batch_resident(regions)", true)
                         : region_counts_launch(labels, static_cast<const uint16_t *>(truth_maps), first, B, T, max_annotators, H, W,
                                                n_segments, n_truth_labels, hist, area, perim, stream, "gcs_score_batch_resident(regions)", true);
    if (rc != GCS_OK) return rc;
    hipLaunchKernelGGL(region_reduce_kernel, dim3(T), dim3(256), 0, stream, hist, area, img_of, n_segments, n_truth_labels,
                       reinterpret_cast<unsigned long long *>(under), reinterpret_cast<unsigned long long *>(under_np));
    GCS_CHECK_LAUNCH("gcs_score_batch_resident");
    return GCS_OK;
}
