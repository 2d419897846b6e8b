/*
 * inpaint.c -- CPU oracle for the opencv2fx/inpaint render() body (opencv2fx/inpaint/inpaint.cpp:286-358):
 *   cvCvtColor(RGBA2RGB) x2, cvCvtColor(RGBA2GRAY)      :303-305
 *   cvThreshold(mask, mask, 0, 255, CV_THRESH_BINARY_INV) :307
 *   cvDilate(mask, mask, NULL, (int)t2)                   :309
 *   cvInpaint(image0, mask, image1, t1, CV_INPAINT_TELEA) :311-318
 *   RGB -> RGBA(a = 255) write-back (noise == 0 path)     :320-358
 *
 * TEST INFRASTRUCTURE ONLY -- see ofxcv_oracle.h.  PARITY UNPINNED: the arithmetic lives in OpenCV 2.4
 * (modules/imgproc/src/color.cpp, thresh.cpp, morph.cpp and modules/photo/src/inpaint.cpp), which the
 * reference neither vendors nor pins and which is absent from this image.  The Telea part restates the
 * published algorithm of photo/src/inpaint.cpp: icvCalcFMM (outward march, negated distances),
 * icvTeleaInpaintFMM, FastMarching_solve and the FIFO-stable CvPriorityQueueFloat, including its quirks:
 *   - VectorLength() returns the SQUARED length, so the distance weight is 1/(|r|^2 * |r|);
 *   - central image gradients are multiplied by 2.0f;
 *   - rows/columns of the padded map with index <= 1 are never marched (first image row / column);
 *   - the sample row/column for k == 1 / l == 1 is clamped to image row / column 1 (km, lm).
 */
#include "ofxcv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };

/* ---- FIFO-stable priority queue: pop order == CvPriorityQueueFloat (a sorted list where a new element is
 *      inserted after every element with T <= its own) -- realised as a binary heap on (T, push sequence). */
typedef struct {
    float T;
    unsigned seq;
    int i, j;
} HeapElem;
typedef struct {
    HeapElem *e;
    int n, cap;
    unsigned seq;
} Heap;

static int heap_less(const HeapElem *a, const HeapElem *b) { return a->T < b->T || (a->T == b->T && a->seq < b->seq); }
static void heap_init(Heap *hp, int cap)
{
    hp->e = (HeapElem *)malloc(sizeof(HeapElem) * (size_t)(cap > 0 ? cap : 1));
    hp->n = 0;
    hp->cap = cap;
    hp->seq = 0;
}
static void heap_push(Heap *hp, int i, int j, float T)
{
    if (hp->n >= hp->cap) { /* cannot happen: every pixel is pushed at most once (see header of cvInpaint) */
        hp->cap = hp->cap * 2 + 16;
        hp->e = (HeapElem *)realloc(hp->e, sizeof(HeapElem) * (size_t)hp->cap);
    }
    HeapElem el = {T, hp->seq++, i, j};
    int k = hp->n++;
    while (k > 0) {
        int p = (k - 1) / 2;
        if (!heap_less(&el, &hp->e[p])) break;
        hp->e[k] = hp->e[p];
        k = p;
    }
    hp->e[k] = el;
}
static int heap_pop(Heap *hp, int *i, int *j)
{
    if (hp->n == 0) return 0;
    *i = hp->e[0].i;
    *j = hp->e[0].j;
    HeapElem el = hp->e[--hp->n];
    int k = 0;
    for (;;) {
        int c = 2 * k + 1;
        if (c >= hp->n) break;
        if (c + 1 < hp->n && heap_less(&hp->e[c + 1], &hp->e[c])) c++;
        if (!heap_less(&hp->e[c], &el)) break;
        hp->e[k] = hp->e[c];
        k = c;
    }
    if (hp->n > 0) hp->e[k] = el;
    return 1;
}

static inline float min4(float a, float b, float c, float d)
{
    a = a < b ? a : b;
    c = c < d ? c : d;
    return a < c ? a : c;
}

/* photo/src/inpaint.cpp FastMarching_solve */
static float fmm_solve(int i1, int j1, int i2, int j2, const uint8_t *f, const float *t, int ecols)
{
    double sol, a11, a22, m12;
    a11 = t[i1 * ecols + j1];
    a22 = t[i2 * ecols + j2];
    m12 = a11 < a22 ? a11 : a22;
    if (f[i1 * ecols + j1] != INSIDE) {
        if (f[i2 * ecols + j2] != INSIDE) {
            if (fabs(a11 - a22) >= 1.0) sol = 1 + m12;
            else sol = (a11 + a22 + sqrt((double)(2 - (a11 - a22) * (a11 - a22)))) * 0.5;
        } else
            sol = 1 + a11;
    } else if (f[i2 * ecols + j2] != INSIDE)
        sol = 1 + a22;
    else
        sol = 1 + m12;
    return (float)sol;
}

/* 3x3 cross / (2r+1)^2 rect dilation of a {0,v} map; pixels outside the map never contribute */
static void dilate_map(const uint8_t *src, uint8_t *dst, int rows, int cols, int r, int cross)
{
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) {
            uint8_t m = 0;
            for (int di = -r; di <= r; di++)
                for (int dj = -r; dj <= r; dj++) {
                    if (cross && di != 0 && dj != 0) continue;
                    int ii = i + di, jj = j + dj;
                    if (ii < 0 || jj < 0 || ii >= rows || jj >= cols) continue;
                    uint8_t v = src[ii * cols + jj];
                    if (v > m) m = v;
                }
            dst[i * cols + j] = m;
        }
}
static void set_border0(uint8_t *m, int rows, int cols)
{
    for (int j = 0; j < cols; j++) m[j] = m[(rows - 1) * cols + j] = 0;
    for (int i = 0; i < rows; i++) m[i * cols] = m[i * cols + cols - 1] = 0;
}

/* photo/src/inpaint.cpp icvCalcFMM(out, t, Out, negate = true) */
static void calc_fmm_negate(uint8_t *f, float *t, Heap *hp, int erows, int ecols)
{
    int ii, jj;
    while (heap_pop(hp, &ii, &jj)) {
        f[ii * ecols + jj] = CHANGE;
        for (int q = 0; q < 4; q++) {
            int i = ii, j = jj;
            if (q == 0) i = ii - 1;
            else if (q == 1) j = jj - 1;
            else if (q == 2) i = ii + 1;
            else j = jj + 1;
            if (i <= 0 || j <= 0 || i > erows || j > ecols) continue;
            if (f[i * ecols + j] == INSIDE) {
                float dist = min4(fmm_solve(i - 1, j, i, j - 1, f, t, ecols), fmm_solve(i + 1, j, i, j - 1, f, t, ecols),
                                  fmm_solve(i - 1, j, i, j + 1, f, t, ecols), fmm_solve(i + 1, j, i, j + 1, f, t, ecols));
                t[i * ecols + j] = dist;
                f[i * ecols + j] = BAND;
                heap_push(hp, i, j, dist);
            }
        }
    }
    for (int i = 0; i < erows * ecols; i++)
        if (f[i] == CHANGE) {
            f[i] = KNOWN;
            t[i] = -t[i];
        }
}

#define IMG(r, c, ch) ((float)out[((size_t)(r) * w + (c)) * 3 + (ch)])

/* photo/src/inpaint.cpp icvTeleaInpaintFMM (method 1) / icvNSInpaintFMM (method 0), 3-channel branches: the same
 * front march, a different colour rule for the pixel that has just received its T */
static void inpaint_fmm(uint8_t *f, float *t, uint8_t *out, int w, int h, int range, Heap *hp, int32_t *order, int method)
{
    const int erows = h + 2, ecols = w + 2;
    int ii, jj, filled = 0;
    while (heap_pop(hp, &ii, &jj)) {
        f[ii * ecols + jj] = KNOWN;
        for (int q = 0; q < 4; q++) {
            int i = ii, j = jj;
            if (q == 0) i = ii - 1;
            else if (q == 1) j = jj - 1;
            else if (q == 2) i = ii + 1;
            else j = jj + 1;
            if (i <= 1 || j <= 1 || i > erows - 1 || j > ecols - 1) continue;
            if (f[i * ecols + j] != INSIDE) continue;
            float dist = min4(fmm_solve(i - 1, j, i, j - 1, f, t, ecols), fmm_solve(i + 1, j, i, j - 1, f, t, ecols),
                              fmm_solve(i - 1, j, i, j + 1, f, t, ecols), fmm_solve(i + 1, j, i, j + 1, f, t, ecols));
            t[i * ecols + j] = dist;
#define F(a, b) f[(a) * ecols + (b)]
#define T(a, b) t[(a) * ecols + (b)]
            for (int color = 0; method == ORC_INPAINT_NS && color <= 2; color++) {
                /* icvNSInpaintFMM: weights from the isophote direction (gradient rotated by 90 degrees, taken from
                 * absolute byte differences), no level-set term; VectorLength() is the squared length as in Telea */
                float Ia = 0, s = 1.0e-20f, wgt, dst, dir, gx, gy, rx, ry;
#define IMGI(r, c, ch) ((int)out[((size_t)(r) * w + (c)) * 3 + (ch)])
                for (int k = i - range; k <= i + range; k++) {
                    int km = k - 1 + (k == 1), kp = k - 1 - (k == erows - 2);
                    for (int l = j - range; l <= j + range; l++) {
                        int lm = l - 1 + (l == 1), lp = l - 1 - (l == ecols - 2);
                        if (!(k > 0 && l > 0 && k < erows - 1 && l < ecols - 1)) continue;
                        if (F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
                        ry = (float)(i - k);
                        rx = (float)(j - l);
                        float vl = rx * rx + ry * ry;
                        dst = 1 / (vl * vl + 1);
                        if (F(k + 1, l) != INSIDE) {
                            if (F(k - 1, l) != INSIDE)
                                gx = (float)(abs(IMGI(kp + 1, lm, color) - IMGI(kp, lm, color)) + abs(IMGI(kp, lm, color) - IMGI(km - 1, lm, color)));
                            else gx = (float)(abs(IMGI(kp + 1, lm, color) - IMGI(kp, lm, color))) * 2.0f;
                        } else {
                            if (F(k - 1, l) != INSIDE) gx = (float)(abs(IMGI(kp, lm, color) - IMGI(km - 1, lm, color))) * 2.0f;
                            else gx = 0;
                        }
                        if (F(k, l + 1) != INSIDE) {
                            if (F(k, l - 1) != INSIDE)
                                gy = (float)(abs(IMGI(km, lp + 1, color) - IMGI(km, lm, color)) + abs(IMGI(km, lm, color) - IMGI(km, lm - 1, color)));
                            else gy = (float)(abs(IMGI(km, lp + 1, color) - IMGI(km, lm, color))) * 2.0f;
                        } else {
                            if (F(k, l - 1) != INSIDE) gy = (float)(abs(IMGI(km, lm, color) - IMGI(km, lm - 1, color))) * 2.0f;
                            else gy = 0;
                        }
                        gx = -gx;
                        dir = rx * gx + ry * gy;
                        if (fabs(dir) <= 0.01) dir = 0.000001f;
                        else dir = (float)fabs((rx * gx + ry * gy) / sqrt((double)(vl * (gx * gx + gy * gy))));
                        wgt = dst * dir;
                        Ia += (float)wgt * (float)(IMGI(km, lm, color));
                        s += wgt;
                    }
                }
#undef IMGI
                int iv = (int)lrint((double)Ia / s); /* cv::saturate_cast<uchar>(double) */
                out[((size_t)(i - 1) * w + (j - 1)) * 3 + color] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
            }
            for (int color = 0; method == ORC_INPAINT_TELEA && color <= 2; color++) {
                float gradIx, gradIy, gradTx, gradTy, rx, ry;
                float Ia = 0, Jx = 0, Jy = 0, s = 1.0e-20f, wgt, dst, lev, dir, sat;
                if (F(i, j + 1) != INSIDE) {
                    if (F(i, j - 1) != INSIDE) gradTx = (float)((T(i, j + 1) - T(i, j - 1))) * 0.5f;
                    else gradTx = (float)((T(i, j + 1) - T(i, j)));
                } else {
                    if (F(i, j - 1) != INSIDE) gradTx = (float)((T(i, j) - T(i, j - 1)));
                    else gradTx = 0;
                }
                if (F(i + 1, j) != INSIDE) {
                    if (F(i - 1, j) != INSIDE) gradTy = (float)((T(i + 1, j) - T(i - 1, j))) * 0.5f;
                    else gradTy = (float)((T(i + 1, j) - T(i, j)));
                } else {
                    if (F(i - 1, j) != INSIDE) gradTy = (float)((T(i, j) - T(i - 1, j)));
                    else gradTy = 0;
                }
                for (int k = i - range; k <= i + range; k++) {
                    int km = k - 1 + (k == 1), kp = k - 1 - (k == erows - 2);
                    for (int l = j - range; l <= j + range; l++) {
                        int lm = l - 1 + (l == 1), lp = l - 1 - (l == ecols - 2);
                        if (!(k > 0 && l > 0 && k < erows - 1 && l < ecols - 1)) continue;
                        if (F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
                        ry = (float)(i - k);
                        rx = (float)(j - l);
                        float vl = rx * rx + ry * ry; /* VectorLength(): squared length */
                        dst = (float)(1. / (vl * sqrt((double)vl)));
                        lev = (float)(1. / (1 + fabsf(T(k, l) - T(i, j))));
                        dir = rx * gradTx + ry * gradTy;
                        if (fabs(dir) <= 0.01) dir = 0.000001f;
                        wgt = (float)fabs(dst * lev * dir);
                        if (F(k, l + 1) != INSIDE) {
                            if (F(k, l - 1) != INSIDE) gradIx = (float)((IMG(km, lp + 1, color) - IMG(km, lm - 1, color))) * 2.0f;
                            else gradIx = (float)((IMG(km, lp + 1, color) - IMG(km, lm, color)));
                        } else {
                            if (F(k, l - 1) != INSIDE) gradIx = (float)((IMG(km, lp, color) - IMG(km, lm - 1, color)));
                            else gradIx = 0;
                        }
                        if (F(k + 1, l) != INSIDE) {
                            if (F(k - 1, l) != INSIDE) gradIy = (float)((IMG(kp + 1, lm, color) - IMG(km - 1, lm, color))) * 2.0f;
                            else gradIy = (float)((IMG(kp + 1, lm, color) - IMG(km, lm, color)));
                        } else {
                            if (F(k - 1, l) != INSIDE) gradIy = (float)((IMG(kp, lm, color) - IMG(km - 1, lm, color)));
                            else gradIy = 0;
                        }
                        Ia += (float)wgt * (float)(IMG(km, lm, color));
                        Jx -= (float)wgt * (float)(gradIx * rx);
                        Jy -= (float)wgt * (float)(gradIy * ry);
                        s += wgt;
                    }
                }
                sat = (float)((Ia / s + (Jx + Jy) / (sqrtf(Jx * Jx + Jy * Jy) + 1.0e-20f) + 0.5f));
                /* cv::saturate_cast<uchar>(float): round to nearest (cvRound) then clamp */
                int iv = (int)lrintf(sat);
                out[((size_t)(i - 1) * w + (j - 1)) * 3 + color] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
            }
#undef F
#undef T
            f[i * ecols + j] = BAND;
            heap_push(hp, i, j, dist);
            if (order) order[(size_t)(i - 1) * w + (j - 1)] = ++filled;
        }
    }
}

int orc_inpaint(const uint8_t *rgb, const uint8_t *mask_in, int w, int h, double radius, int method,
                uint8_t *out, float *t_map, uint8_t *f_map, int32_t *order)
{
    if (method != ORC_INPAINT_NS && method != ORC_INPAINT_TELEA) return -1;
    int range = orc_cv_round(radius);
    if (range < 1) range = 1;
    if (range > 100) range = 100;
    const int ecols = w + 2, erows = h + 2;
    const size_t en = (size_t)ecols * erows;
    uint8_t *mask = (uint8_t *)calloc(en, 1), *band = (uint8_t *)malloc(en), *ring = (uint8_t *)malloc(en);
    uint8_t *f = (uint8_t *)calloc(en, 1);
    float *t = (float *)malloc(sizeof(float) * en);
    memcpy(out, rgb, (size_t)w * h * 3);
    if (order) memset(order, 0, sizeof(int32_t) * (size_t)w * h);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            if (mask_in[(size_t)i * w + j] != 0) mask[(i + 1) * ecols + j + 1] = INSIDE;
    for (size_t i = 0; i < en; i++) t[i] = 1.0e6f;
    dilate_map(mask, band, erows, ecols, 1, 1);
    int nheap = 0;
    for (size_t i = 0; i < en; i++) nheap += band[i] != 0;
    int rc = 0;
    if (nheap > 0) {
        Heap hp, outq;
        heap_init(&hp, nheap);
        for (size_t i = 0; i < en; i++) band[i] = (uint8_t)(band[i] > mask[i] ? band[i] - mask[i] : 0);
        set_border0(band, erows, ecols);
        for (int i = 0; i < erows; i++)
            for (int j = 0; j < ecols; j++)
                if (band[i * ecols + j] != 0) heap_push(&hp, i, j, 0);
        for (size_t i = 0; i < en; i++) {
            if (band[i]) { f[i] = BAND; t[i] = 0; }
            if (mask[i]) f[i] = INSIDE;
        }
        if (method == ORC_INPAINT_TELEA) {
            /* CV_INPAINT_TELEA: distances outside the hole (negative), within `range` of it */
            dilate_map(mask, ring, erows, ecols, range, 0);
            int nring = 0;
            for (size_t i = 0; i < en; i++) {
                ring[i] = (uint8_t)(ring[i] > mask[i] ? ring[i] - mask[i] : 0);
                nring += ring[i] != 0;
            }
            if (nring > 0) {
                heap_init(&outq, nring);
                for (int i = 0; i < erows; i++)
                    for (int j = 0; j < ecols; j++)
                        if (band[i * ecols + j] != 0) heap_push(&outq, i, j, 0);
                for (size_t i = 0; i < en; i++) ring[i] = (uint8_t)(ring[i] > band[i] ? ring[i] - band[i] : 0);
                set_border0(ring, erows, ecols);
                calc_fmm_negate(ring, t, &outq, erows, ecols);
                /* the reference passes `mask` (INSIDE where hole, KNOWN elsewhere) as the flag map */
                inpaint_fmm(mask, t, out, w, h, range, &hp, order, method);
                free(outq.e);
            }
        } else {
            /* CV_INPAINT_NS: no outside distances; T stays 1e6 off the band */
            inpaint_fmm(mask, t, out, w, h, range, &hp, order, method);
        }
        free(hp.e);
    }
    if (t_map) memcpy(t_map, t, sizeof(float) * en);
    if (f_map) memcpy(f_map, mask, en);
    free(mask); free(band); free(ring); free(f); free(t);
    return rc;
}

int orc_inpaint_telea(const uint8_t *rgb, const uint8_t *mask_in, int w, int h, double radius,
                      uint8_t *out, float *t_map, uint8_t *f_map, int32_t *order)
{
    return orc_inpaint(rgb, mask_in, w, h, radius, ORC_INPAINT_TELEA, out, t_map, f_map, order);
}

/* cvCvtColor(RGBA2GRAY) 8-bit: (R*4899 + G*9617 + B*1868 + 8192) >> 14 ; threshold BINARY_INV at 0 ; 3x3 rect dilate x iters */
void orc_inpaint_mask(const uint8_t *rgba, ptrdiff_t row_bytes, int w, int h, int dilate_iters, uint8_t *mask)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = rgba + y * row_bytes;
        for (int x = 0; x < w; x++) {
            int g = (s[x * 4] * 4899 + s[x * 4 + 1] * 9617 + s[x * 4 + 2] * 1868 + 8192) >> 14;
            mask[(size_t)y * w + x] = g > 0 ? 0 : 255;
        }
    }
    if (dilate_iters > 0) {
        uint8_t *tmp = (uint8_t *)malloc((size_t)w * h);
        for (int it = 0; it < dilate_iters; it++) {
            dilate_map(mask, tmp, h, w, 1, 0);
            memcpy(mask, tmp, (size_t)w * h);
        }
        free(tmp);
    }
}

int orc_inpaint_render(const uint8_t *src, ptrdiff_t src_row_bytes, int w, int h, double radius, double dilation,
                       uint8_t *dst, ptrdiff_t dst_row_bytes)
{
    uint8_t *rgb = (uint8_t *)malloc((size_t)w * h * 3), *res = (uint8_t *)malloc((size_t)w * h * 3);
    uint8_t *mask = (uint8_t *)malloc((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) rgb[((size_t)y * w + x) * 3 + c] = src[y * src_row_bytes + x * 4 + c];
    orc_inpaint_mask(src, src_row_bytes, w, h, dilation > 0 ? (int)dilation : 0, mask);
    int rc = orc_inpaint_telea(rgb, mask, w, h, radius, res, NULL, NULL, NULL);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t *d = dst + y * dst_row_bytes + x * 4;
            for (int c = 0; c < 3; c++) d[c] = res[((size_t)y * w + x) * 3 + c];
            d[3] = 255;
        }
    free(rgb); free(res); free(mask);
    return rc;
}
