/*
 * farneback.c -- CPU oracle: restatement of cv::calcOpticalFlowFarneback as the reference
 * calls it (VectorGenerator/VectorGenerator.cpp:403: pyr_scale 0.5, winsize 3, flags 0,
 * levels/iterations/poly_n/poly_sigma from the plugin parameters :390-399).
 *
 * TEST INFRASTRUCTURE ONLY -- see ofxcv_oracle.h.  PARITY UNPINNED: the algorithm lives in
 * OpenCV (modules/video/src/optflowgf.cpp, imgproc smooth.cpp/filter.cpp/imgwarp.cpp), which
 * is neither vendored nor version-pinned by the reference and is absent from this image.
 * Each function below names the upstream function whose published behaviour it restates.
 *
 * GENERATION: the restatement follows the OpenCV 2.4 / 3.x sources (the reference's CI installs 2.4, .travis.yml:22-30;
 * VectorGenerator.cpp:80-88 compiles against 2.x-4.x).  The one place where 4.x is known to differ in the last bit is
 * getGaussianKernel (4.x normalises the taps in double before the cast to float, getGaussianKernelBitExact): see
 * orc_set_gaussian_kernel_generation(), a switch that only tests/test_cv2_crosscheck.py flips when it finds a 4.x cv2.
 *
 * Build with -ffp-contract=off: OpenCV's scalar code is evaluated op by op in the type of
 * each expression (float*float in float, running sums in double), and the oracle follows that.
 */
#include "ofxcv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cvRound: nearest, ties to even (SSE cvtsd2si semantics) */
int orc_cv_round(double v) { return (int)lrint(v); }
int orc_cv_floor(double v) { return (int)floor(v); }

/* borderInterpolate(p, len, BORDER_REFLECT_101) */
int orc_border_reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* which getGaussianKernel is restated: 3 (default) = 2.4 / 3.x, 4 = 4.x */
static int g_gauss_generation = 3;
void orc_set_gaussian_kernel_generation(int generation) { g_gauss_generation = generation >= 4 ? 4 : 3; }
int orc_get_gaussian_kernel_generation(void) { return g_gauss_generation; }

/* smooth.cpp getGaussianKernel(n, sigma, CV_32F).
 * 2.4 / 3.x: the taps are cast to float first, the sum runs over the float values (in double), then float(tap * 1/sum).
 * 4.x (getGaussianKernelBitExact): taps and their sum stay double (softdouble there, libm exp here: the exponential may
 * differ in the last bit of the double, which the cast to float hides except at a rounding boundary), one cast at the end. */
void orc_gaussian_kernel_f32(int n, double sigma, float *k)
{
    if (g_gauss_generation >= 4) {
        static const double small_tab4[4][7] = {
            {1.},
            {0.25, 0.5, 0.25},
            {0.0625, 0.25, 0.375, 0.25, 0.0625},
            {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125}};
        if (n % 2 == 1 && n <= 7 && sigma <= 0) {
            for (int i = 0; i < n; i++) k[i] = (float)small_tab4[n >> 1][i];
            return;
        }
        double sigmaX4 = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
        double scale4 = -0.5 / (sigmaX4 * sigmaX4), sum4 = 0, t4[256];
        for (int i = 0; i < n && i < 256; i++) {
            double x = i - (n - 1) * 0.5;
            t4[i] = exp(scale4 * x * x);
            sum4 += t4[i];
        }
        sum4 = 1. / sum4;
        for (int i = 0; i < n && i < 256; i++) k[i] = (float)(t4[i] * sum4);
        return;
    }
    static const float small_tab[4][7] = {
        {1.f},
        {0.25f, 0.5f, 0.25f},
        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float *fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : NULL;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

/*
 * smooth.cpp GaussianBlur -> sepFilter2D, CV_32F source and kernel, BORDER_REFLECT_101.
 * filter.cpp evaluation order for a float image:
 *   rows:    ksize<=5 symmetric  -> S[0]*k0 + (S[-1]+S[1])*k1 (+ (S[-2]+S[2])*k2)   (SymmRowSmallFilter)
 *            otherwise           -> left-to-right sum over all taps                  (RowFilter)
 *   columns: ksize==3            -> (S[-1]+S[1])*k1 + S[0]*k0                        (SymmColumnSmallFilter)
 *            otherwise           -> k0*S[0] + sum_k k[k]*(S[k]+S[-k])                (SymmColumnFilter)
 * rows first into an f32 buffer, then columns.
 */
/* Filter contraction (orc_set_filter_contraction): OpenCV 4.x dispatches the separable filters and the vertical pass of resize to
 * universal-intrinsics code (AVX2 / NEON: SymmRowSmallVec_32f, RowVec_32f, SymmColumnSmallVec_32f, SymmColumnVec_32f, VResizeLinearVec_32f)
 * whose taps are v_muladd / v_fma: one rounding where the scalar loops of 2.4 / 3.x (and this file, built with -ffp-contract=off) have two.
 *   0 (default)  scalar order, every product and sum rounded
 *   1            the vector paths' form: fma(S[-k] + S[k], k_k, acc) for the symmetric small row / every column filter, fma(S[j], k_j, acc) for
 *                the general row filter, fma(S0, b0, S1 * b1) for the vertical lerp of resize; the horizontal lerp (HResizeLinear: no float
 *                vector path) stays as it is.  Restated from memory of the 4.x sources: "parity unpinned" like the rest; the switch
 *                exists so that the day a cv2 is at hand tests/test_cv2_crosscheck.py can say which variant it is.  optflowgf.cpp itself is plain
 *                C++ built for the baseline ISA: nothing in it is contracted. */
static int g_filter_contraction = 0;
void orc_set_filter_contraction(int on) { g_filter_contraction = on ? 1 : 0; }
int orc_get_filter_contraction(void) { return g_filter_contraction; }
static inline float madd(float a, float b, float c) { return g_filter_contraction ? fmaf(a, b, c) : a * b + c; }

void orc_gaussian_blur_f32(const float *src, int w, int h, float *dst, int ksize, double sigma)
{
    float kbuf[64];
    float *kern = ksize <= 64 ? kbuf : (float *)malloc(sizeof(float) * ksize);
    int r = ksize / 2;
    orc_gaussian_kernel_f32(ksize, sigma, kern);
    const float *kc = kern + r;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * h);
    int *xi = (int *)malloc(sizeof(int) * (w + 2 * r));
    for (int x = -r; x < w + r; x++) xi[x + r] = orc_border_reflect101(x, w);

    for (int y = 0; y < h; y++) {
        const float *S = src + (size_t)y * w;
        float *D = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            const int *ix = xi + x + r; /* ix[j] = source column of tap j */
            float s;
            if (ksize == 3) {
                s = madd(S[ix[-1]] + S[ix[1]], kc[1], S[ix[0]] * kc[0]);
            } else if (ksize == 5) {
                s = madd(S[ix[-2]] + S[ix[2]], kc[2], madd(S[ix[-1]] + S[ix[1]], kc[1], S[ix[0]] * kc[0]));
            } else {
                s = kern[0] * S[ix[-r]];
                for (int j = 1; j < ksize; j++) s = madd(S[ix[j - r]], kern[j], s);
            }
            D[x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        float *D = dst + (size_t)y * w;
        if (ksize == 3) {
            const float *S0 = tmp + (size_t)orc_border_reflect101(y - 1, h) * w;
            const float *S1 = tmp + (size_t)y * w;
            const float *S2 = tmp + (size_t)orc_border_reflect101(y + 1, h) * w;
            for (int x = 0; x < w; x++) D[x] = madd(S0[x] + S2[x], kc[1], S1[x] * kc[0]);
        } else {
            const float *S1 = tmp + (size_t)y * w;
            for (int x = 0; x < w; x++) D[x] = kc[0] * S1[x];
            for (int k = 1; k <= r; k++) {
                const float *Sa = tmp + (size_t)orc_border_reflect101(y + k, h) * w;
                const float *Sb = tmp + (size_t)orc_border_reflect101(y - k, h) * w;
                float f = kc[k];
                for (int x = 0; x < w; x++) D[x] = madd(Sa[x] + Sb[x], f, D[x]);
            }
        }
    }
    free(xi);
    free(tmp);
    if (kern != kbuf) free(kern);
}

/*
 * imgwarp.cpp resize(INTER_LINEAR) for float: resizeGeneric_<HResizeLinear<float,float,float,1>,
 * VResizeLinear<float,float,float>>.  Horizontal lerp first, then vertical; coefficients are
 * float; fx = (float)((dx+0.5)*scale_x - 0.5); source index clamped with the fraction zeroed.
 */
static void resize_coeffs(int ssize, int dsize, int *ofs, float *a0, float *a1)
{
    double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = orc_cv_floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        a0[d] = 1.f - f;
        a1[d] = f;
    }
}

/* imgwarp.cpp / resize.cpp cv::resize: "in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to
 * INTER_LINEAR" -- an INTER_LINEAR request whose scale is exactly 2 in both directions is served by resizeAreaFast_: the sum of
 * the 2x2 block times 0.25f.  Same value as the bilinear form up to the association of the three float additions:
 *   0 (default)  bilinear: (a*.5 + b*.5)*.5 + (c*.5 + d*.5)*.5 = ((a+b) + (c+d))/4 -- also what the 4.x universal-intrinsics
 *                path of ResizeAreaFastVec_SIMD_32f computes (row pairs first)
 *   1            ((a + b) + c) + d   the scalar loop of resizeAreaFast_Invoker (2.4.x, builds without SIMD)
 *   2            (a + c) + (b + d)   ResizeAreaFastVec_SIMD_32f of 3.x (SSE2: the two rows are added first)
 * (a b / c d = the block, row-major).  Restated from memory of the upstream sources: "parity unpinned" like the rest. */
static int g_resize_generation = 0;
void orc_set_resize_generation(int generation) { g_resize_generation = generation < 0 || generation > 2 ? 0 : generation; }
int orc_get_resize_generation(void) { return g_resize_generation; }

void orc_resize_linear_f32(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    if (sw == dw && sh == dh) { /* resize(): dsize == ssize -> copyTo */
        memcpy(dst, src, sizeof(float) * (size_t)sw * sh * cn);
        return;
    }
    if (g_resize_generation && sw == 2 * dw && sh == 2 * dh) {
        for (int dy = 0; dy < dh; dy++) {
            const float *S0 = src + (size_t)(2 * dy) * sw * cn, *S1 = S0 + (size_t)sw * cn;
            float *D = dst + (size_t)dy * dw * cn;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    float a = S0[2 * dx * cn + c], b = S0[(2 * dx + 1) * cn + c], cc = S1[2 * dx * cn + c], d = S1[(2 * dx + 1) * cn + c];
                    float sum = g_resize_generation == 1 ? ((a + b) + cc) + d : (a + cc) + (b + d);
                    D[dx * cn + c] = sum * 0.25f;
                }
        }
        return;
    }
    int *xo = (int *)malloc(sizeof(int) * dw), *yo = (int *)malloc(sizeof(int) * dh);
    float *xa0 = (float *)malloc(sizeof(float) * dw), *xa1 = (float *)malloc(sizeof(float) * dw);
    float *ya0 = (float *)malloc(sizeof(float) * dh), *ya1 = (float *)malloc(sizeof(float) * dh);
    float *r0 = (float *)malloc(sizeof(float) * (size_t)dw * cn), *r1 = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    resize_coeffs(sw, dw, xo, xa0, xa1);
    resize_coeffs(sh, dh, yo, ya0, ya1);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yo[dy], sy1 = sy0 + 1 < sh ? sy0 + 1 : sh - 1; /* VResize clips the 2nd row index */
        const float *S0 = src + (size_t)sy0 * sw * cn, *S1 = src + (size_t)sy1 * sw * cn;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xo[dx];
            for (int c = 0; c < cn; c++) {
                if (sx + 1 < sw) {
                    r0[dx * cn + c] = S0[sx * cn + c] * xa0[dx] + S0[(sx + 1) * cn + c] * xa1[dx];
                    r1[dx * cn + c] = S1[sx * cn + c] * xa0[dx] + S1[(sx + 1) * cn + c] * xa1[dx];
                } else { /* dx >= xmax: D = S[sx]*ONE */
                    r0[dx * cn + c] = S0[sx * cn + c] * 1.f;
                    r1[dx * cn + c] = S1[sx * cn + c] * 1.f;
                }
            }
        }
        float b0 = ya0[dy], b1 = ya1[dy];
        float *D = dst + (size_t)dy * dw * cn;
        for (int i = 0; i < dw * cn; i++) D[i] = madd(r0[i], b0, r1[i] * b1);
    }
    free(xo); free(yo); free(xa0); free(xa1); free(ya0); free(ya1); free(r0); free(r1);
}

/* 6x6 SPD inverse via Cholesky (G.inv(DECOMP_CHOLESKY)); returns inverse in place */
static void chol_inverse6(double A[6][6])
{
    double L[6][6] = {{0}}, Li[6][6] = {{0}};
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            L[i][j] = i == j ? sqrt(s) : s / L[j][j];
        }
    for (int c = 0; c < 6; c++) /* Li = L^-1, forward substitution per column */
        for (int i = 0; i < 6; i++) {
            double s = i == c ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= L[i][k] * Li[k][c];
            Li[i][c] = s / L[i][i];
        }
    for (int i = 0; i < 6; i++) /* A^-1 = Li^T Li */
        for (int j = 0; j < 6; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += Li[k][i] * Li[k][j];
            A[i][j] = s;
        }
}

/* optflowgf.cpp FarnebackPrepareGaussian */
void orc_polyexp_prepare(int n, double sigma, float *g0, float *xg0, float *xxg0, double ig[4])
{
    float *g = g0 + n, *xg = xg0 + n, *xxg = xxg0 + n;
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    chol_inverse6(G);
    ig[0] = G[1][1]; /* ig11 */
    ig[1] = G[0][3]; /* ig03 */
    ig[2] = G[3][3]; /* ig33 */
    ig[3] = G[5][5]; /* ig55 */
}

/* optflowgf.cpp FarnebackPolyExp */
void orc_polyexp(const float *src, int width, int height, float *dst, int n, double sigma)
{
    float *kbuf = (float *)malloc(sizeof(float) * (n * 6 + 3));
    float *_row = (float *)malloc(sizeof(float) * (size_t)(width + n * 2) * 3);
    float *g = kbuf + n, *xg = g + n * 2 + 1, *xxg = xg + n * 2 + 1;
    float *row = _row + n * 3;
    double ig[4];
    orc_polyexp_prepare(n, sigma, kbuf, kbuf + (2 * n + 1), kbuf + 2 * (2 * n + 1), ig);
    double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];

    for (int y = 0; y < height; y++) {
        float g0 = g[0], g1, g2;
        const float *srow0 = src + (size_t)y * width, *srow1 = 0;
        float *drow = dst + (size_t)y * width * 5;

        /* vertical part of convolution (float) */
        for (int x = 0; x < width; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * width;
            srow1 = src + (size_t)(y + k < height - 1 ? y + k : height - 1) * width;
            for (int x = 0; x < width; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0;
                row[x * 3 + 1] = t1;
                row[x * 3 + 2] = t2;
            }
        }
        /* horizontal part of convolution (replicated borders, double accumulators) */
        for (int x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[width * 3 + x] = row[width * 3 + x - 3];
        }
        for (int x = 0; x < width; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            /* r1 (constant term) is not stored */
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(_row);
    free(kbuf);
}

/* optflowgf.cpp FarnebackUpdateMatrices */
void orc_update_matrices(const float *R0_, const float *R1, const float *flow_, float *M_,
                         int width, int height, int y0_, int y1_)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    size_t step1 = (size_t)width * 5;

    for (int y = y0_; y < y1_; y++) {
        const float *flow = flow_ + (size_t)y * width * 2;
        const float *R0 = R0_ + (size_t)y * width * 5;
        float *M = M_ + (size_t)y * width * 5;
        for (int x = 0; x < width; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = orc_cv_floor(fx), y1 = orc_cv_floor(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1;
            fy -= y1;
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1 < (unsigned)(height - 1)) {
                const float *ptr = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) * (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) * (y >= height - BORDER ? border[height - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;     /* G(1,1) */
            M[x * 5 + 1] = (r4 + r5) * r6;    /* G(1,2) */
            M[x * 5 + 2] = r5 * r5 + r6 * r6; /* G(2,2) */
            M[x * 5 + 3] = r4 * r2 + r6 * r3; /* h(1)   */
            M[x * 5 + 4] = r6 * r2 + r5 * r3; /* h(2)   */
        }
    }
}

static inline void solve_px(double g11, double g12, double g22, double h1, double h2, double scale, float *flow)
{
    double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale, h1_ = h1 * scale, h2_ = h2 * scale;
    double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
    flow[0] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
    flow[1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
}

/* optflowgf.cpp FarnebackUpdateFlow_Blur, OpenCV's own running-sum evaluation */
static void update_flow_blur_faithful(const float *R0, const float *R1, float *flow_, float *matM,
                                      int width, int height, int block_size, int update_matrices)
{
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = (1 << 10) / width > block_size ? (1 << 10) / width : block_size;
    double scale = 1. / (block_size * block_size);
    double *_vsum = (double *)malloc(sizeof(double) * (size_t)(width + m * 2 + 2) * 5);
    double *vsum = _vsum + (m + 1) * 5;

    const float *srow0 = matM;
    for (int x = 0; x < width * 5; x++) vsum[x] = srow0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        srow0 = matM + (size_t)(y < height - 1 ? y : height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] += srow0[x];
    }
    for (int y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float *flow = flow_ + (size_t)y * width * 2;
        srow0 = matM + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * width * 5;
        const float *srow1 = matM + (size_t)(y + m < height - 1 ? y + m : height - 1) * width * 5;
        /* vertical blur: NOTE the row difference is rounded to float before it is accumulated */
        for (int x = 0; x < width * 5; x++) vsum[x] += srow1[x] - srow0[x];
        /* update borders */
        for (int x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2);
        g12 = vsum[1] * (m + 2);
        g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2);
        h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; x++) {
            g11 += vsum[x * 5];
            g12 += vsum[x * 5 + 1];
            g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3];
            h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < width; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            solve_px(g11, g12, g22, h1, h2, scale, flow + x * 2);
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update_matrices && (y1 == height || y1 >= y0 + min_update_stripe)) {
            orc_update_matrices(R0, R1, flow_, matM, width, height, y0, y1);
            y0 = y1;
        }
    }
    free(_vsum);
}

/*
 * Same box window, evaluated without running sums: per pixel
 *   hs(y,x) = sum_{j=-m..m} (double)M(y, clamp(x+j))        left to right
 *   G(y,x)  = sum_{i=-m..m} hs(clamp(y+i), x)               top to bottom
 * then the identical 2x2 solve; UpdateMatrices is applied to all rows afterwards, which is
 * what the stripe schedule above amounts to.  This is the evaluation order of the HIP kernel.
 */
static void update_flow_blur_direct(const float *R0, const float *R1, float *flow_, float *matM,
                                    int width, int height, int block_size, int update_matrices)
{
    int m = block_size / 2;
    double scale = 1. / (block_size * block_size);
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            double acc[5] = {0, 0, 0, 0, 0};
            for (int i = -m; i <= m; i++) {
                int yy = y + i < 0 ? 0 : (y + i > height - 1 ? height - 1 : y + i);
                const float *row = matM + (size_t)yy * width * 5;
                double hs[5] = {0, 0, 0, 0, 0};
                for (int j = -m; j <= m; j++) {
                    int xx = x + j < 0 ? 0 : (x + j > width - 1 ? width - 1 : x + j);
                    for (int c = 0; c < 5; c++) hs[c] = (j == -m) ? (double)row[xx * 5 + c] : hs[c] + (double)row[xx * 5 + c];
                }
                for (int c = 0; c < 5; c++) acc[c] = (i == -m) ? hs[c] : acc[c] + hs[c];
            }
            solve_px(acc[0], acc[1], acc[2], acc[3], acc[4], scale, flow_ + ((size_t)y * width + x) * 2);
        }
    }
    if (update_matrices) orc_update_matrices(R0, R1, flow_, matM, width, height, 0, height);
}

void orc_update_flow_blur(const float *R0, const float *R1, float *flow, float *M,
                          int w, int h, int block_size, int update_matrices, int mode)
{
    if (mode == ORC_BLUR_DIRECT) update_flow_blur_direct(R0, R1, flow, M, w, h, block_size, update_matrices);
    else update_flow_blur_faithful(R0, R1, flow, M, w, h, block_size, update_matrices);
}

/*
 * optflowgf.cpp FarnebackUpdateFlow_GaussianBlur (flags & OPTFLOW_FARNEBACK_GAUSSIAN): the window is a separable
 * Gaussian, sigma = (block_size/2) * 0.3, taps normalised in double and stored as float; both passes accumulate in
 * FLOAT:  v = row[m]*k[0]; for i = 1..m: v += (row[m+i] + row[m-i]) * k[i]   (rows / columns replicated at the
 * border).  The sums enter the 2x2 solve as they are (the taps already sum to one).  UpdateMatrices follows on row
 * stripes that the window has left behind, which is the same as updating all rows after the pass.
 */
void orc_update_flow_gaussian(const float *R0, const float *R1, float *flow_, float *matM, int width, int height,
                              int block_size, int update_matrices)
{
    const int m = block_size / 2;
    const double sigma = m * 0.3;
    double s = 1.;
    float *kernel = (float *)malloc(sizeof(float) * (size_t)(m + 1));
    float *vbuf = (float *)malloc(sizeof(float) * (size_t)(width + m * 2 + 2) * 5);
    float *vsum = vbuf + (m + 1) * 5;
    kernel[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) kernel[i] = (float)(kernel[i] * s);

    for (int y = 0; y < height; y++) {
        float *flow = flow_ + (size_t)y * width * 2;
        const float *rc = matM + (size_t)y * width * 5;
        for (int x = 0; x < width * 5; x++) {
            float s0 = rc[x] * kernel[0];
            for (int i = 1; i <= m; i++) {
                const float *rm = matM + (size_t)(y - i > 0 ? y - i : 0) * width * 5;
                const float *rp = matM + (size_t)(y + i < height - 1 ? y + i : height - 1) * width * 5;
                s0 += (rp[x] + rm[x]) * kernel[i];
            }
            vsum[x] = s0;
        }
        for (int x = 0; x < m * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        for (int x = 0; x < width; x++) {
            float sum[5];
            for (int c = 0; c < 5; c++) {
                float s0 = vsum[x * 5 + c] * kernel[0];
                for (int i = 1; i <= m; i++) s0 += (vsum[(x + i) * 5 + c] + vsum[(x - i) * 5 + c]) * kernel[i];
                sum[c] = s0;
            }
            double g11 = sum[0], g12 = sum[1], g22 = sum[2], h1 = sum[3], h2 = sum[4];
            double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            flow[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
        }
    }
    if (update_matrices) orc_update_matrices(R0, R1, flow_, matM, width, height, 0, height);
    free(kernel);
    free(vbuf);
}

/*
 * imgproc resize(..., INTER_AREA) for f32, as calcOpticalFlowFarneback uses it on the caller's initial flow
 * (OPTFLOW_USE_INITIAL_FLOW: resize(flow0, flow, top-level size, INTER_AREA); flow *= scale).  Shrinking only.
 *  - equal sizes: a copy;
 *  - integer factors (ResizeAreaFast_): float sum over the fx*fy cell, row-major, taken four at a time
 *    (sum += S0 + S1 + S2 + S3) then singly, times (float)(1/(fx*fy));
 *  - otherwise (ResizeArea_ with computeResizeAreaTab): per destination index the covered source cells with weights
 *    [partial left] (ceil(f1) - f1)/cw, full cells 1/cw, [partial right] min(min(f2 - floor(f2), 1), cw)/cw where
 *    f1 = d*scale, f2 = f1 + scale, cw = min(scale, ssize - f1), partial cells only if wider than 1e-3; a source row
 *    is reduced horizontally into buf (float, buf += S*alpha from 0), rows are combined as sum = beta*buf for the
 *    first row of a destination row and sum += beta*buf after.
 */
typedef struct { int si; float alpha; } area_tap;
static int area_taps(int d, int ssize, double scale, area_tap *t)
{
    int n = 0;
    double fsx1 = d * scale, fsx2 = fsx1 + scale;
    double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    if (sx2 > ssize - 1) sx2 = ssize - 1;
    if (sx1 > sx2) sx1 = sx2;
    if (sx1 - fsx1 > 1e-3) { t[n].si = sx1 - 1; t[n++].alpha = (float)((sx1 - fsx1) / cell); }
    for (int sx = sx1; sx < sx2; sx++) { t[n].si = sx; t[n++].alpha = (float)(1.0 / cell); }
    if (fsx2 - sx2 > 1e-3) {
        double a = fsx2 - sx2 < 1. ? fsx2 - sx2 : 1.;
        if (a > cell) a = cell;
        t[n].si = sx2; t[n++].alpha = (float)(a / cell);
    }
    return n;
}

void orc_resize_area_f32(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    if (sw == dw && sh == dh) {
        memcpy(dst, src, sizeof(float) * (size_t)sw * sh * cn);
        return;
    }
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    const int ix = (int)scale_x, iy = (int)scale_y;
    if (fabs(scale_x - ix) < DBL_EPSILON && fabs(scale_y - iy) < DBL_EPSILON) {
        const int area = ix * iy;
        const float scale = 1.f / area;
        int *ofs = (int *)malloc(sizeof(int) * (size_t)area);
        for (int sy = 0, k = 0; sy < iy; sy++)
            for (int sx = 0; sx < ix; sx++) ofs[k++] = (sy * sw + sx) * cn;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    const float *S = src + ((size_t)dy * iy * sw + (size_t)dx * ix) * cn + c;
                    float sum = 0;
                    int k = 0;
                    for (; k <= area - 4; k += 4) sum += S[ofs[k]] + S[ofs[k + 1]] + S[ofs[k + 2]] + S[ofs[k + 3]];
                    for (; k < area; k++) sum += S[ofs[k]];
                    dst[((size_t)dy * dw + dx) * cn + c] = sum * scale;
                }
        free(ofs);
        return;
    }
    area_tap *xt = (area_tap *)malloc(sizeof(area_tap) * (size_t)(ix + 3)), *yt = (area_tap *)malloc(sizeof(area_tap) * (size_t)(iy + 3));
    for (int dy = 0; dy < dh; dy++) {
        int ny = area_taps(dy, sh, scale_y, yt);
        for (int dx = 0; dx < dw; dx++) {
            int nx = area_taps(dx, sw, scale_x, xt);
            for (int c = 0; c < cn; c++) {
                float sum = 0;
                for (int j = 0; j < ny; j++) {
                    const float *S = src + (size_t)yt[j].si * sw * cn + c;
                    float buf = 0;
                    for (int k = 0; k < nx; k++) buf = buf + S[(size_t)xt[k].si * cn] * xt[k].alpha;
                    sum = j == 0 ? yt[j].alpha * buf : sum + yt[j].alpha * buf;
                }
                dst[((size_t)dy * dw + dx) * cn + c] = sum;
            }
        }
    }
    free(xt);
    free(yt);
}

/* level clip of FarnebackOpticalFlowImpl::calc: stop before a side drops under min_size=32 */
int orc_farneback_num_levels(int w, int h, double pyr_scale, int levels)
{
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    return k;
}

void orc_farneback_level_geom(int w, int h, double pyr_scale, int k, int *lw, int *lh, double *sigma, int *ksize)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    double s = (1. / scale - 1) * 0.5;
    int smooth_sz = orc_cv_round(s * 5) | 1;
    if (smooth_sz < 3) smooth_sz = 3;
    *sigma = s;
    *ksize = smooth_sz;
    *lw = orc_cv_round(w * scale);
    *lh = orc_cv_round(h * scale);
}

void orc_farneback_pyr_image(const uint8_t *img, size_t step, int w, int h,
                             int lw, int lh, double sigma, int ksize, float *I)
{
    float *fimg = (float *)malloc(sizeof(float) * (size_t)w * h);
    float *blur = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) fimg[(size_t)y * w + x] = (float)img[(size_t)y * step + x];
    orc_gaussian_blur_f32(fimg, w, h, blur, ksize, sigma);
    orc_resize_linear_f32(blur, w, h, 1, I, lw, lh);
    free(fimg);
    free(blur);
}

/* optflowgf.cpp FarnebackOpticalFlowImpl::calc (CPU path); flags: ORC_OPTFLOW_USE_INITIAL_FLOW, ORC_OPTFLOW_FARNEBACK_GAUSSIAN */
int orc_calc_optical_flow_farneback(const uint8_t *prev, const uint8_t *next, size_t step,
                                    int w, int h, float *flow0,
                                    double pyr_scale, int levels, int winsize, int iterations,
                                    int poly_n, double poly_sigma, int flags, int blur_mode)
{
    if ((flags & ~(ORC_OPTFLOW_USE_INITIAL_FLOW | ORC_OPTFLOW_FARNEBACK_GAUSSIAN)) || !(pyr_scale < 1) || w <= 0 || h <= 0) return -1;
    const uint8_t *img[2] = {prev, next};
    levels = orc_farneback_num_levels(w, h, pyr_scale, levels);

    float *prevFlow = NULL;
    int pw = 0, ph = 0;
    for (int k = levels; k >= 0; k--) {
        int width, height, ksz;
        double sigma, scale = 1;
        for (int i = 0; i < k; i++) scale *= pyr_scale;
        orc_farneback_level_geom(w, h, pyr_scale, k, &width, &height, &sigma, &ksz);
        size_t npx = (size_t)width * height;
        float *flow = k > 0 ? (float *)malloc(sizeof(float) * npx * 2) : flow0;
        if (!prevFlow) {
            if (flags & ORC_OPTFLOW_USE_INITIAL_FLOW) {
                if (k > 0) orc_resize_area_f32(flow0, w, h, 2, flow, width, height); /* k == 0: flow IS flow0 */
                for (size_t i = 0; i < npx * 2; i++) flow[i] = (float)(flow[i] * scale);
            } else {
                memset(flow, 0, sizeof(float) * npx * 2);
            }
        } else {
            orc_resize_linear_f32(prevFlow, pw, ph, 2, flow, width, height);
            double mul = 1. / pyr_scale;
            for (size_t i = 0; i < npx * 2; i++) flow[i] = (float)(flow[i] * mul);
        }
        float *R[2], *I = (float *)malloc(sizeof(float) * npx), *M = (float *)malloc(sizeof(float) * npx * 5);
        for (int i = 0; i < 2; i++) {
            R[i] = (float *)malloc(sizeof(float) * npx * 5);
            orc_farneback_pyr_image(img[i], step, w, h, width, height, sigma, ksz, I);
            orc_polyexp(I, width, height, R[i], poly_n, poly_sigma);
        }
        orc_update_matrices(R[0], R[1], flow, M, width, height, 0, height);
        for (int i = 0; i < iterations; i++) {
            if (flags & ORC_OPTFLOW_FARNEBACK_GAUSSIAN)
                orc_update_flow_gaussian(R[0], R[1], flow, M, width, height, winsize, i < iterations - 1);
            else
                orc_update_flow_blur(R[0], R[1], flow, M, width, height, winsize, i < iterations - 1, blur_mode);
        }
        free(R[0]); free(R[1]); free(I); free(M);
        if (prevFlow) free(prevFlow);
        prevFlow = flow;
        pw = width;
        ph = height;
    }
    return 0;
}
