/* CPU ORACLE IN C — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product.
 *
 * A second, independent restatement of SPEC.md §3-§4 (explicit pyramid / reflect indexing and plain
 * loops; oracle/spec_oracle.py uses scipy.ndimage instead). PARITY UNPINNED against the
 * reference: /root/reference holds no Gabor or k-means code to restate (the
 * slot is /root/reference/BSD_metrics/script.py:30). Pinned only against spec_oracle.py
 * (tests/test_oracle.py) and the committed fixtures in tests/golden/.
 *
 * Build: make -C oracle   (gcc -O2 -shared -> oracle/_build/liboracle.so)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* The loops below are parallelised with OpenMP where iterations are independent (rows of a response, pixels of an
 * assign pass with per-thread integer sums): integer arithmetic, so the result does not depend on the thread count.
 * oracle_set_threads(1) gives the single-thread baseline bench.py reports. */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline int refl(int i, int n) {
    /* scipy.ndimage mode='reflect' (d c b a | a b c d | d c b a), any distance */
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

static uint32_t isqrt64(uint64_t n) {
    uint64_t lo = 0, hi = 0xffffffffull; /* largest r with r*r <= n */
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (mid * mid <= n) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}

/* SPEC.md §3 pyramid step: src [Hs][Ws][3] u8 -> dst [ceil(Hs/2)][ceil(Ws/2)][3] u8, the 2x2 block mean
 * (round half up) of src edge-replicated to even size. */
static void pyramid_down(const uint8_t *src, int Hs, int Ws, uint8_t *dst) {
    const int Hd = (Hs + 1) / 2, Wd = (Ws + 1) / 2;
    for (int y = 0; y < Hd; ++y)
        for (int x = 0; x < Wd; ++x)
            for (int c = 0; c < 3; ++c) {
                int acc = 2;
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j) {
                        const int yy = 2 * y + i < Hs ? 2 * y + i : Hs - 1;
                        const int xx = 2 * x + j < Ws ? 2 * x + j : Ws - 1;
                        acc += src[((size_t)yy * Ws + xx) * 3 + c];
                    }
                dst[((size_t)y * Wd + x) * 3 + c] = (uint8_t)(acc >> 2);
            }
}

/* One filter on one level image: lev [Hl][Wl][3] u8, channel c -> g [Hl][Wl] u16 (SPEC.md §3 response). */
static void level_response(const uint8_t *lev, int Hl, int Wl, int c, const int16_t *tre, const int16_t *tim, int ks,
                           int shift, const int *ry, const int *rx, uint16_t *g) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Hl; ++y)
        for (int x = 0; x < Wl; ++x) {
            int64_t vre = 0, vim = 0;
            for (int dy = 0; dy < ks; ++dy) {
                const uint8_t *row = lev + (size_t)ry[y + dy] * Wl * 3 + c;
                for (int dx = 0; dx < ks; ++dx) {
                    const int64_t p = row[(size_t)rx[x + dx] * 3];
                    vre += p * tre[dy * ks + dx];
                    vim += p * tim[dy * ks + dx];
                }
            }
            /* arithmetic shift == floor division by 2^shift */
            const int64_t are = vre >= 0 ? vre >> shift : -((-vre + ((int64_t)1 << shift) - 1) >> shift);
            const int64_t aim = vim >= 0 ? vim >> shift : -((-vim + ((int64_t)1 << shift) - 1) >> shift);
            g[(size_t)y * Wl + x] = (uint16_t)isqrt64((uint64_t)(are * are + aim * aim));
        }
}

/* SPEC.md §3: img [H][W][3] u8, tapq [F][2][ks][ks] i16 -> out [3F][H][W] u16, d = c*F + f. Filter f runs on
 * pyramid level (f / n_orient) / 2 and its response is replicated over 2^L x 2^L blocks. */
int oracle_gabor_features(const uint8_t *img, int H, int W, const int16_t *tapq, int F, int ks, int shift,
                          int n_orient, uint16_t *out) {
    const int R = (ks - 1) / 2;
    const int n_levels = ((F - 1) / n_orient) / 2 + 1;
    uint8_t *lev = (uint8_t *)malloc((size_t)H * W * 3), *nxt = (uint8_t *)malloc((size_t)H * W * 3);
    uint16_t *g = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)H * W);
    int *ry = (int *)malloc(sizeof(int) * (size_t)(H + 2 * R));
    int *rx = (int *)malloc(sizeof(int) * (size_t)(W + 2 * R));
    if (!lev || !nxt || !g || !ry || !rx) return 1;
    memcpy(lev, img, (size_t)H * W * 3);
    int Hl = H, Wl = W;
    for (int L = 0; L < n_levels; ++L) {
        for (int i = 0; i < Hl + 2 * R; ++i) ry[i] = refl(i - R, Hl);
        for (int i = 0; i < Wl + 2 * R; ++i) rx[i] = refl(i - R, Wl);
        for (int f = 0; f < F; ++f) {
            if ((f / n_orient) / 2 != L) continue;
            const int16_t *tre = tapq + ((size_t)f * 2 + 0) * ks * ks;
            const int16_t *tim = tapq + ((size_t)f * 2 + 1) * ks * ks;
            for (int c = 0; c < 3; ++c) {
                level_response(lev, Hl, Wl, c, tre, tim, ks, shift, ry, rx, g);
                uint16_t *o = out + ((size_t)c * F + f) * H * W;
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) o[(size_t)y * W + x] = g[(size_t)(y >> L) * Wl + (x >> L)];
            }
        }
        if (L + 1 < n_levels) {
            pyramid_down(lev, Hl, Wl, nxt);
            uint8_t *t = lev; lev = nxt; nxt = t;
            Hl = (Hl + 1) / 2;
            Wl = (Wl + 1) / 2;
        }
    }
    free(lev); free(nxt); free(g); free(ry); free(rx);
    return 0;
}

/* SPEC.md §4 on nimg images of planar features feats [nimg][D][P] u16, one codebook
 * (nimg == 1: the per-image mode). labels [nimg][P] i32, cent [k][D] u16 (returned). */
int oracle_kmeans(const uint16_t *feats, int nimg, int D, long P, int k, int n_iter, int32_t *labels,
                  uint16_t *cent) {
    int64_t *c = (int64_t *)malloc(sizeof(int64_t) * (size_t)k * D);
    int64_t *sum = (int64_t *)malloc(sizeof(int64_t) * (size_t)k * D);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    if (!c || !sum || !cnt) return 1;
    for (int j = 0; j < k; ++j) {
        const long p = ((2L * j + 1) * P) / (2L * k);
        for (int d = 0; d < D; ++d) c[(size_t)j * D + d] = feats[(size_t)d * P + p]; /* image 0 */
    }
    for (int t = 0; t < n_iter; ++t) {
        memset(sum, 0, sizeof(int64_t) * (size_t)k * D);
        memset(cnt, 0, sizeof(int64_t) * (size_t)k);
        for (int b = 0; b < nimg; ++b) {
            const uint16_t *fb = feats + (size_t)b * D * P;
#pragma omp parallel
            {
                int64_t *lsum = (int64_t *)calloc((size_t)k * D + k, sizeof(int64_t));   /* per-thread sums | counts */
                int64_t *lcnt = lsum + (size_t)k * D;
#pragma omp for schedule(static)
                for (long p = 0; p < P; ++p) {
                    int64_t best = 0;
                    int bj = 0;
                    for (int j = 0; j < k; ++j) {
                        int64_t dist = 0;
                        for (int d = 0; d < D; ++d) {
                            const int64_t df = (int64_t)fb[(size_t)d * P + p] - c[(size_t)j * D + d];
                            dist += df * df;
                        }
                        if (j == 0 || dist < best) { best = dist; bj = j; }
                    }
                    labels[(size_t)b * P + p] = bj;
                    lcnt[bj]++;
                    for (int d = 0; d < D; ++d) lsum[(size_t)bj * D + d] += fb[(size_t)d * P + p];
                }
#pragma omp critical
                {
                    for (int i = 0; i < k * D; ++i) sum[i] += lsum[i];
                    for (int j = 0; j < k; ++j) cnt[j] += lcnt[j];
                }
                free(lsum);
            }
        }
        if (t < n_iter - 1)
            for (int j = 0; j < k; ++j)
                if (cnt[j] > 0)
                    for (int d = 0; d < D; ++d)
                        c[(size_t)j * D + d] = (2 * sum[(size_t)j * D + d] + cnt[j]) / (2 * cnt[j]);
    }
    for (int i = 0; i < k * D; ++i) cent[i] = (uint16_t)c[i];
    free(c);
    free(sum);
    free(cnt);
    return 0;
}
