/* CPU ORACLE IN C — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product.
 *
 * A second, independent restatement of SPEC.md §3-§4 (explicit reflect indexing and plain
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

/* SPEC.md §3: img [H][W][3] u8, tapq [F][2][ks][ks] i16 -> out [3F][H][W] u16, d = c*F + f */
int oracle_gabor_features(const uint8_t *img, int H, int W, const int16_t *tapq, int F, int ks, int shift,
                          uint16_t *out) {
    const int R = (ks - 1) / 2;
    int *ry = (int *)malloc(sizeof(int) * (size_t)(H + 2 * R));
    int *rx = (int *)malloc(sizeof(int) * (size_t)(W + 2 * R));
    if (!ry || !rx) return 1;
    for (int i = 0; i < H + 2 * R; ++i) ry[i] = refl(i - R, H);
    for (int i = 0; i < W + 2 * R; ++i) rx[i] = refl(i - R, W);
    for (int c = 0; c < 3; ++c)
        for (int f = 0; f < F; ++f) {
            const int16_t *tre = tapq + ((size_t)f * 2 + 0) * ks * ks;
            const int16_t *tim = tapq + ((size_t)f * 2 + 1) * ks * ks;
            uint16_t *o = out + ((size_t)c * F + f) * H * W;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    int64_t vre = 0, vim = 0;
                    for (int dy = 0; dy < ks; ++dy) {
                        const uint8_t *row = img + (size_t)ry[y + dy] * W * 3 + c;
                        for (int dx = 0; dx < ks; ++dx) {
                            const int64_t p = row[(size_t)rx[x + dx] * 3];
                            vre += p * tre[dy * ks + dx];
                            vim += p * tim[dy * ks + dx];
                        }
                    }
                    /* arithmetic shift == floor division by 2^shift */
                    const int64_t are = vre >= 0 ? vre >> shift : -((-vre + ((int64_t)1 << shift) - 1) >> shift);
                    const int64_t aim = vim >= 0 ? vim >> shift : -((-vim + ((int64_t)1 << shift) - 1) >> shift);
                    o[(size_t)y * W + x] = (uint16_t)isqrt64((uint64_t)(are * are + aim * aim));
                }
        }
    free(ry);
    free(rx);
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
                cnt[bj]++;
                for (int d = 0; d < D; ++d) sum[(size_t)bj * D + d] += fb[(size_t)d * P + p];
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
