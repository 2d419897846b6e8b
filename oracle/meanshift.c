/*
 * meanshift.c -- CPU oracle for the segment workload as BASELINE.json labels it ("opencv2fx/segment mean-shift"):
 * cv::pyrMeanShiftFiltering(src, dst, sp, sr, maxLevel, TermCriteria(ITER+EPS, 5, 1)) on 8-bit 3-channel images.
 *
 * TEST INFRASTRUCTURE ONLY -- see ofxcv_oracle.h.  PARITY UNPINNED.  The reference's segment plugin calls
 * cvPyrSegmentation (opencv2fx/segment/segment.cpp:296-302; OpenCV <= 2.4 legacy module, pyramid linking), whose
 * source is not in the reference tree and cannot be restated bit-exactly without it (SURVEY.md section 0.6 / 8(a)
 * row S); the benchmark configuration asks for mean-shift semantics instead.  This file restates the published
 * algorithm of OpenCV's modules/imgproc/src/segmentation.cpp (pyrMeanShiftFiltering) and
 * modules/imgproc/src/pyramids.cpp (pyrDown_/pyrUp_ for 8-bit: 5x5 [1 4 6 4 1] fixed point).  Everything is
 * integer arithmetic except the window means (cvRound of sum * (1./count) in double).
 */
#include "ofxcv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* pyrDown_, 8UC3, BORDER_REFLECT_101: dst(x,y) = (sum_{i,j} k[i]k[j] src(r101(2y+i-2), r101(2x+j-2)) + 128) >> 8 */
static void pyr_down_8u3(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
            for (int c = 0; c < 3; c++) {
                int s = 0;
                for (int i = 0; i < 5; i++) {
                    int sy = orc_border_reflect101(2 * y + i - 2, sh);
                    for (int j = 0; j < 5; j++) {
                        int sx = orc_border_reflect101(2 * x + j - 2, sw);
                        s += k[i] * k[j] * src[((size_t)sy * sw + sx) * 3 + c];
                    }
                }
                dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)((s + 128) >> 8);
            }
}

/* pyrUp_, 8UC3.  Horizontal pass into 32-bit rows (even column: s[x-1] + 6 s[x] + s[x+1], odd: 4 (s[x] + s[x+1]);
 * left edge even: 6 s[0] + 2 s[1]; right edge: even s[W-2] + 7 s[W-1], odd 8 s[W-1]; a 1-pixel-wide source gives 8 s),
 * vertical pass even row: r[y-1] + 6 r[y] + r[y+1], odd row: 4 (r[y] + r[y+1]) with source rows r101(2 sy, 2 H)/2;
 * result (v + 32) >> 6.  dst is 2*ssize or 2*ssize - 1 in each dimension. */
static void pyr_up_row(const uint8_t *s, int sw, int *row /* 2*sw*3 */)
{
    for (int c = 0; c < 3; c++) {
        if (sw == 1) {
            row[c] = row[3 + c] = s[c] * 8;
            continue;
        }
        row[c] = s[c] * 6 + s[3 + c] * 2;
        row[3 + c] = (s[c] + s[3 + c]) * 4;
        int sx = (sw - 1) * 3 + c;
        row[(sw - 1) * 6 + c] = s[sx - 3] + s[sx] * 7;
        row[(sw - 1) * 6 + 3 + c] = s[sx] * 8;
        for (int x = 1; x < sw - 1; x++) {
            row[x * 6 + c] = s[(x - 1) * 3 + c] + s[x * 3 + c] * 6 + s[(x + 1) * 3 + c];
            row[x * 6 + 3 + c] = (s[x * 3 + c] + s[(x + 1) * 3 + c]) * 4;
        }
    }
}
static void pyr_up_8u3(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    int *r0 = (int *)malloc(sizeof(int) * (size_t)sw * 6), *r1 = (int *)malloc(sizeof(int) * (size_t)sw * 6),
        *r2 = (int *)malloc(sizeof(int) * (size_t)sw * 6);
    for (int y = 0; y < sh; y++) {
        int ym = orc_border_reflect101((y - 1) * 2, sh * 2) / 2, yp = orc_border_reflect101((y + 1) * 2, sh * 2) / 2;
        pyr_up_row(src + (size_t)ym * sw * 3, sw, r0);
        pyr_up_row(src + (size_t)y * sw * 3, sw, r1);
        pyr_up_row(src + (size_t)yp * sw * 3, sw, r2);
        uint8_t *d0 = dst + (size_t)(2 * y) * dw * 3;
        uint8_t *d1 = dst + (size_t)(2 * y + 1 < dh - 1 ? 2 * y + 1 : dh - 1) * dw * 3;
        for (int x = 0; x < dw * 3; x++) {
            uint8_t t1 = (uint8_t)(((r1[x] + r2[x]) * 4 + 32) >> 6);
            uint8_t t0 = (uint8_t)((r0[x] + r1[x] * 6 + r2[x] + 32) >> 6);
            d1[x] = t1; /* when dh is odd the last source row writes row dh-1 twice: t1 first, then t0 */
            d0[x] = t0;
        }
    }
    free(r0); free(r1); free(r2);
}

static inline int sq(int v) { return v * v; }

int orc_pyr_mean_shift(const uint8_t *src0, int w, int h, double sp0, double sr, int max_level, int max_iter, double eps,
                       uint8_t *dst0)
{
    if (max_level < 0 || max_level > 8 || w <= 0 || h <= 0) return -1;
    if (max_iter < 1) max_iter = 1;
    if (max_iter > 100) max_iter = 100;
    if (eps < 0) eps = 0;
    const int isr2 = orc_cv_round(sr * sr), isr22 = isr2 > 16 ? isr2 : 16;
    uint8_t *srcp[9], *dstp[9];
    int lw[9], lh[9];
    srcp[0] = (uint8_t *)src0;
    dstp[0] = dst0;
    lw[0] = w;
    lh[0] = h;
    for (int l = 1; l <= max_level; l++) {
        lw[l] = (lw[l - 1] + 1) / 2;
        lh[l] = (lh[l - 1] + 1) / 2;
        srcp[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l] * 3);
        dstp[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l] * 3);
        pyr_down_8u3(srcp[l - 1], lw[l - 1], lh[l - 1], srcp[l], lw[l], lh[l]);
    }
    uint8_t *mask0 = (uint8_t *)malloc((size_t)w * h), *mtmp = (uint8_t *)malloc((size_t)w * h);

    for (int level = max_level; level >= 0; level--) {
        const uint8_t *src = srcp[level];
        uint8_t *dst = dstp[level];
        const int W = lw[level], H = lh[level];
        uint8_t *mask = NULL;
        float sp = (float)(sp0 / (1 << level));
        if (sp < 1) sp = 1;

        if (level < max_level) {
            const int W1 = lw[level + 1], H1 = lh[level + 1];
            const uint8_t *d1 = dstp[level + 1];
            pyr_up_8u3(d1, W1, H1, dst, W, H);
            memset(mask0, 0, (size_t)W * H);
            for (int i = 1; i < H1 - 1; i++)
                for (int j = 1; j < W1 - 1; j++) {
                    const uint8_t *p = d1 + ((size_t)i * W1 + j) * 3;
                    int c0 = p[0], c1 = p[1], c2 = p[2], hit = 0;
                    for (int di = -1; di <= 1 && !hit; di++)
                        for (int dj = -1; dj <= 1 && !hit; dj++) {
                            if (!di && !dj) continue;
                            const uint8_t *q = d1 + ((size_t)(i + di) * W1 + (j + dj)) * 3;
                            hit = sq(c0 - q[0]) + sq(c1 - q[1]) + sq(c2 - q[2]) >= isr22;
                        }
                    /* mask row pointer starts at row 1 and advances two rows per coarse row: fine pixel (2i-1, 2j-1) */
                    if (2 * i - 1 < H && 2 * j - 1 < W) mask0[(size_t)(2 * i - 1) * W + (2 * j - 1)] = (uint8_t)hit;
                }
            /* cv::dilate(m, m, Mat()): 3x3 rect, outside never contributes */
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    uint8_t m = 0;
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            int yy = y + dy, xx = x + dx;
                            if (yy < 0 || xx < 0 || yy >= H || xx >= W) continue;
                            if (mask0[(size_t)yy * W + xx] > m) m = mask0[(size_t)yy * W + xx];
                        }
                    mtmp[(size_t)y * W + x] = m;
                }
            memcpy(mask0, mtmp, (size_t)W * H);
            mask = mask0;
        }

        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) {
                if (mask && !mask[(size_t)i * W + j]) continue;
                int x0 = j, y0 = i, x1, y1;
                const uint8_t *sp_ = src + ((size_t)i * W + j) * 3;
                int c0 = sp_[0], c1 = sp_[1], c2 = sp_[2];
                for (int iter = 0; iter < max_iter; iter++) {
                    int count = 0, s0 = 0, s1 = 0, s2 = 0, sx = 0, sy = 0;
                    int minx = orc_cv_round(x0 - sp), miny = orc_cv_round(y0 - sp);
                    int maxx = orc_cv_round(x0 + sp), maxy = orc_cv_round(y0 + sp);
                    if (minx < 0) minx = 0;
                    if (miny < 0) miny = 0;
                    if (maxx > W - 1) maxx = W - 1;
                    if (maxy > H - 1) maxy = H - 1;
                    for (int y = miny; y <= maxy; y++) {
                        int row_count = 0;
                        const uint8_t *ptr = src + ((size_t)y * W + minx) * 3;
                        for (int x = minx; x <= maxx; x++, ptr += 3) {
                            int t0 = ptr[0], t1 = ptr[1], t2 = ptr[2];
                            if (sq(t0 - c0) + sq(t1 - c1) + sq(t2 - c2) <= isr2) {
                                s0 += t0; s1 += t1; s2 += t2;
                                sx += x;
                                row_count++;
                            }
                        }
                        count += row_count;
                        sy += y * row_count;
                    }
                    if (count == 0) break;
                    double icount = 1. / count;
                    x1 = orc_cv_round(sx * icount);
                    y1 = orc_cv_round(sy * icount);
                    s0 = orc_cv_round(s0 * icount);
                    s1 = orc_cv_round(s1 * icount);
                    s2 = orc_cv_round(s2 * icount);
                    int stop = (x0 == x1 && y0 == y1) ||
                               abs(x1 - x0) + abs(y1 - y0) + sq(s0 - c0) + sq(s1 - c1) + sq(s2 - c2) <= eps;
                    x0 = x1; y0 = y1;
                    c0 = s0; c1 = s1; c2 = s2;
                    if (stop) break;
                }
                uint8_t *d = dst + ((size_t)i * W + j) * 3;
                d[0] = (uint8_t)c0; d[1] = (uint8_t)c1; d[2] = (uint8_t)c2;
            }
    }
    for (int l = 1; l <= max_level; l++) {
        free(srcp[l]);
        free(dstp[l]);
    }
    free(mask0);
    free(mtmp);
    return 0;
}
