// farneback.hip -- dense Farneback optical flow for gfx950 (MI355X).
//
// Replaces cv::calcOpticalFlowFarneback as called at VectorGenerator/VectorGenerator.cpp:403
// (pyr_scale 0.5, winsize 3, flags 0; levels / iterations / poly_n / poly_sigma are plugin
// parameters, :390-399).  The stages follow OpenCV's modules/video/src/optflowgf.cpp:
//   pyramid image   convertTo(32F) + GaussianBlur(full res) + resize(INTER_LINEAR)   [F1,F2]
//   polyexp         FarnebackPolyExp: separable (2n+1)^2 weighted quadratic fit      [F3]
//   update          FarnebackUpdateMatrices: warped gather of R1 + border scale      [F4]
//   blur+solve      FarnebackUpdateFlow_Blur: box window of M, 2x2 solve per pixel   [F5]
//   prolongation    resize(prevFlow, INTER_LINEAR) * 1/pyr_scale                     [F6]
//
// Device data layout: every 5-channel field (R0, R1, M) is stored as 5 planes of
// `pitch` x height floats (pitch = width rounded up to 64 floats = 256 B) so that a wavefront
// reading 64 consecutive pixels of one plane issues one fully coalesced 256-byte request.
// The flow field is 2-channel interleaved (float2 per pixel), as OpenCV returns it.
//
// Numerics: every stage evaluates the same IEEE operations in the same order as the CPU code it
// replaces (f32 products and sums where OpenCV uses float, f64 accumulators where it uses
// double); the file is compiled with -ffp-contract=off so no FMA contraction changes a rounding.
// The one deliberate difference: OpenCV's box window keeps running sums (and rounds each row
// difference to f32 before accumulating it); here each pixel sums its own 3x3 window in f64.
#include <cfloat>
#include <chrono>
#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

constexpr int kMaxGaussTaps = 255;
constexpr int kMaxPolyN = 15;
constexpr int kMaxLevels = OFXCV_FB_MAX_LEVELS;
constexpr int kMaxBatch = OFXCV_FB_MAX_BATCH;

// Batched calls: every kernel of the level walk takes the frame pair from the z coordinate of its grid.  Scratch fields of
// consecutive pairs lie a fixed stride apart; what the caller owns (source images, flow fields) comes as a pointer table
// in the kernel arguments.  A single call is a batch of one (grid z = 1, stride unused).
struct ImgTab {   // 8-bit source images: entry 2 * pair + {0 = prev, 1 = next}
    const uint8_t *p[2 * kMaxBatch];
    size_t step[2 * kMaxBatch];
};
struct RgbaTab {  // F7 fused into the last iteration of level 0: per pair an RGBA f32 image that receives flow / render scale in the mapped channels
    float *p[kMaxBatch];        // null: no image for this pair
    ptrdiff_t step[kMaxBatch];  // row bytes
    unsigned mu[kMaxBatch], mv[kMaxBatch];  // bit c: channel c <- flow.x / flow.y (y wins where both are set, as in the reference loop)
    double rsx, rsy;            // render scale
};
struct FlowTab {  // 2-channel flow fields, one per pair (the caller's at level 0, scratch on the coarser levels)
    float *p[kMaxBatch];
    size_t step[kMaxBatch];
};

struct GaussTaps {
    int ksize;
    int fc;  // filter contraction (option "farneback.filter_contraction"): the taps as fused multiply-adds (madd below)
    float k[kMaxGaussTaps];
};

struct PolyCoef {
    int n;
    float g[2 * kMaxPolyN + 1], xg[2 * kMaxPolyN + 1], xxg[2 * kMaxPolyN + 1];  // index k + n
    double ig11, ig03, ig33, ig55;
};

// ------------------------------------------------------------------ host-side coefficient prep

// smooth.cpp getGaussianKernel(n, sigma, CV_32F).  generation 3 (default): OpenCV 2.4 / 3.x -- the taps are cast to float, summed
// (in double) over the float values, then float(tap * 1/sum).  generation 4: OpenCV 4.x (getGaussianKernelBitExact) -- taps and
// their sum stay double, one cast at the end; two taps of the 9- and 19-tap kernels differ by one ulp.  Option
// "farneback.gaussian_kernel_generation"; the oracle's counterpart is orc_set_gaussian_kernel_generation.
void make_gauss_taps(int n, double sigma, GaussTaps &t, int generation = 3) {
    static const float small_tab[4][7] = {{1.f},
                                          {0.25f, 0.5f, 0.25f},
                                          {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                          {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float *fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : nullptr;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    t.ksize = n;
    if (generation >= 4) {
        if (fixed) {
            for (int i = 0; i < n; i++) t.k[i] = fixed[i];
            return;
        }
        double v[kMaxGaussTaps];
        for (int i = 0; i < n; i++) {
            double x = i - (n - 1) * 0.5;
            v[i] = std::exp(scale2X * x * x);
            sum += v[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < n; i++) t.k[i] = (float)(v[i] * sum);
        return;
    }
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double v = fixed ? (double)fixed[i] : std::exp(scale2X * x * x);
        t.k[i] = (float)v;
        sum += t.k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) t.k[i] = (float)(t.k[i] * sum);
}

// optflowgf.cpp FarnebackPrepareGaussian: 1-D weights and the four entries of inv(G) that matter.
// G is block structured; its inverse is obtained with a Cholesky factorisation like G.inv(DECOMP_CHOLESKY).
void make_poly_coef(int n, double sigma, PolyCoef &pc) {
    pc.n = n;
    float *g = pc.g + n, *xg = pc.xg + n, *xxg = pc.xxg + n;
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6] = {};
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
    double L[6][6] = {}, Li[6][6] = {};
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double a = G[i][j];
            for (int k = 0; k < j; k++) a -= L[i][k] * L[j][k];
            L[i][j] = i == j ? std::sqrt(a) : a / L[j][j];
        }
    for (int c = 0; c < 6; c++)
        for (int i = 0; i < 6; i++) {
            double a = i == c ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) a -= L[i][k] * Li[k][c];
            Li[i][c] = a / L[i][i];
        }
    auto inv = [&](int i, int j) {
        double a = 0;
        for (int k = 0; k < 6; k++) a += Li[k][i] * Li[k][j];
        return a;
    };
    pc.ig11 = inv(1, 1);
    pc.ig03 = inv(0, 3);
    pc.ig33 = inv(3, 3);
    pc.ig55 = inv(5, 5);
}

// ------------------------------------------------------------------ device helpers

__device__ __forceinline__ int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One tap of a separable filter / of resize's vertical lerp: a * b + c with two roundings (fc = 0: the scalar loops of OpenCV 2.4 / 3.x and the
// oracle's default) or as ONE fused multiply-add (fc = 1: what OpenCV 4.x's universal-intrinsics paths compute with v_muladd -- SymmRowSmallVec_32f,
// RowVec_32f, SymmColumnSmallVec_32f, SymmColumnVec_32f, VResizeLinearVec_32f).  Option "farneback.filter_contraction"; oracle: orc_set_filter_contraction.
// The pyramid kernels without a GaussTaps argument carry the flag in bit 4 of their `area` argument.
__device__ __forceinline__ float madd(float a, float b, float c, int fc) { return fc ? __builtin_fmaf(a, b, c) : a * b + c; }

// imgwarp.cpp resize(INTER_LINEAR) coefficient rule for destination index d
__device__ __forceinline__ void lerp_coef_scaled(int d, int ssize, double scale, int &s, float &a0, float &a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    a0 = 1.f - f;
    a1 = f;
}
__device__ __forceinline__ void lerp_coef(int d, int ssize, int dsize, int &s, float &a0, float &a1) {
    double scale = (double)ssize / dsize;
    float f = (float)((d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    a0 = 1.f - f;
    a1 = f;
}

// The last step of resize(INTER_LINEAR): the four filtered samples an output sample lies between.  `area` != 0 only when the
// level is EXACTLY half the frame in both directions: cv::resize then rewrites INTER_LINEAR to INTER_AREA ("INTER_AREA (fast)
// also is equal to INTER_LINEAR", imgwarp.cpp / resize.cpp) and resizeAreaFast_ sums the 2x2 block and multiplies by 0.25f --
// the same value up to the association of the three float additions:
//   0  (t00*.5 + t01*.5)*.5 + (t10*.5 + t11*.5)*.5 = ((t00+t01) + (t10+t11)) / 4   bilinear = the 4.x universal-intrinsics row pairs
//   1  ((t00 + t01) + t10) + t11                                                   the scalar loop (2.4.x; builds without SIMD)
//   2  (t00 + t10) + (t01 + t11)                                                   ResizeAreaFastVec_SIMD_32f of 3.x (SSE2: rows first)
// Option "farneback.resize_generation"; the oracle's counterpart is orc_set_resize_generation.
__device__ __forceinline__ float resize_combine(float t00, float t01, float t10, float t11, float ax0, float ax1, float b0, float b1, int area, int fc) {
    if (area == 1) return (((t00 + t01) + t10) + t11) * 0.25f;
    if (area == 2) return ((t00 + t10) + (t01 + t11)) * 0.25f;
    const float r0 = t00 * ax0 + t01 * ax1, r1 = t10 * ax0 + t11 * ax1;  // (HResizeLinear has no float vector path: never contracted)
    return madd(r0, b0, r1 * b1, fc);
}

// Workgroup -> tile mapping.  The dispatcher is observed to place workgroup b on XCD b % 8 and every XCD has its own
// L2, so with the plain mapping two neighbouring tiles -- which share halo rows/columns and the cache lines of the
// R1 samples -- never share an L2.  This bijective remap hands every XCD a contiguous row-major run of tiles
// (cdna_hip_programming.md T1).  It only changes which workgroup computes which tile: results are unaffected.
// With a batch in the grid's z dimension the run continues across pairs (z-major), so the pair index comes out of the remap too.
__device__ __forceinline__ void xcd_tile(int &bx, int &by, int &bz) {
    const unsigned gx = gridDim.x, gxy = gx * gridDim.y, nwg = gxy * gridDim.z, id = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x;
    const unsigned xcd = id & 7u, q = nwg >> 3, r = nwg & 7u;
    unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    bz = (int)(t / gxy);
    t -= (unsigned)bz * gxy;
    by = (int)(t / gx);
    bx = (int)(t - (unsigned)by * gx);
}
__device__ __forceinline__ void xcd_tile(int &bx, int &by) {
    int bz;
    xcd_tile(bx, by, bz);
}

// ------------------------------------------------------------------ F1/F2 pyramid image
//
// OpenCV blurs at full resolution and then decimates; only the two source columns / rows that
// each output sample interpolates between are ever used, so the row filter is evaluated only at
// those columns (T1: `ntap` samples per output column, all source rows) and the column filter
// only at the needed rows.  Values are identical to blur-then-resize because the column filter
// never mixes columns.

// (grid z = frame first + z of the table; its half-blurred rows at T1 + z * H * ncol)
__global__ __launch_bounds__(256) void pyr_hblur_kernel(ImgTab imgs, int first, int W, int H,
                                                        int lw, int ntap, GaussTaps gk, float *__restrict__ T1) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    int ncol = lw * ntap;
    if (c >= ncol) return;
    const uint8_t *__restrict__ img = imgs.p[first + blockIdx.z];
    const size_t step = imgs.step[first + blockIdx.z];
    T1 += (size_t)blockIdx.z * H * ncol;
    int sx;
    if (ntap == 1) {
        sx = c;
    } else {
        float a0, a1;
        lerp_coef(c >> 1, W, lw, sx, a0, a1);
        sx = min(sx + (c & 1), W - 1);
    }
    const uint8_t *S = img + (size_t)y * step;
    const int ksize = gk.ksize, r = ksize >> 1;
    float s;
    if (ksize == 3) {
        s = madd((float)S[reflect101(sx - 1, W)] + (float)S[reflect101(sx + 1, W)], gk.k[2], (float)S[sx] * gk.k[1], gk.fc);
    } else if (ksize == 5) {
        s = madd((float)S[reflect101(sx - 2, W)] + (float)S[reflect101(sx + 2, W)], gk.k[4],
                 madd((float)S[reflect101(sx - 1, W)] + (float)S[reflect101(sx + 1, W)], gk.k[3], (float)S[sx] * gk.k[2], gk.fc), gk.fc);
    } else {
        s = gk.k[0] * (float)S[reflect101(sx - r, W)];
        for (int j = 1; j < ksize; j++) s = madd((float)S[reflect101(sx - r + j, W)], gk.k[j], s, gk.fc);
    }
    T1[(size_t)y * ncol + c] = s;
}

__device__ __forceinline__ float col_filter(const float *__restrict__ T1, int ncol, int c, int y, int H, const GaussTaps &gk) {
    const int ksize = gk.ksize, r = ksize >> 1;
    const float *kc = gk.k + r;
    if (ksize == 3)
        return madd(T1[(size_t)reflect101(y - 1, H) * ncol + c] + T1[(size_t)reflect101(y + 1, H) * ncol + c], kc[1], T1[(size_t)y * ncol + c] * kc[0], gk.fc);
    float s = kc[0] * T1[(size_t)y * ncol + c];
    for (int k = 1; k <= r; k++)
        s = madd(T1[(size_t)reflect101(y + k, H) * ncol + c] + T1[(size_t)reflect101(y - k, H) * ncol + c], kc[k], s, gk.fc);
    return s;
}

__global__ __launch_bounds__(256) void pyr_vblur_resize_kernel(const float *__restrict__ T1, int W, int H, int lw, int lh,
                                                               int ntap, GaussTaps gk, float *__restrict__ I, size_t I_stride, int area) {
    int dx = blockIdx.x * blockDim.x + threadIdx.x;
    int dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= lw || dy >= lh) return;
    int ncol = lw * ntap;
    T1 += (size_t)blockIdx.z * H * ncol;
    I += (size_t)blockIdx.z * I_stride;
    float out;
    if (ntap == 1) {
        out = col_filter(T1, ncol, dx, dy, H, gk);
    } else {
        int sx, sy;
        float ax0, ax1, b0, b1;
        lerp_coef(dx, W, lw, sx, ax0, ax1);
        lerp_coef(dy, H, lh, sy, b0, b1);
        int sy1 = min(sy + 1, H - 1);
        float t00 = col_filter(T1, ncol, dx * 2, sy, H, gk), t10 = col_filter(T1, ncol, dx * 2, sy1, H, gk);
        if (sx + 1 < W) {
            float t01 = col_filter(T1, ncol, dx * 2 + 1, sy, H, gk), t11 = col_filter(T1, ncol, dx * 2 + 1, sy1, H, gk);
            out = resize_combine(t00, t01, t10, t11, ax0, ax1, b0, b1, area, gk.fc);
        } else {
            const float r0 = t00 * 1.f, r1 = t10 * 1.f;
            out = madd(r0, b0, r1 * b1, gk.fc);
        }
    }
    I[(size_t)dy * lw + dx] = out;
}

// Fused pyramid image: one workgroup produces an OW x OH tile of the level image.  The 8-bit source footprint
// of the tile (plus the blur radius, borders reflected on load) is staged in LDS once; the row filter is evaluated
// at the two source columns every output column interpolates between, into a second LDS buffer; the column filter
// and the two lerps finish the tile.  Same operations and order as the two-kernel form above (which remains the
// fall-back when a footprint does not fit in LDS), without the round trip of the half-blurred rows through HBM.
struct PyrTile {
    int ow, oh;    // output tile
    int cw, rh;    // staged source footprint (columns, rows), upper bounds
};

__global__ __launch_bounds__(256) void pyr_fused_kernel(ImgTab imgs, int W, int H, int lw, int lh, int ntap,
                                                        GaussTaps gk, PyrTile t, float *__restrict__ I, size_t I_stride, int area) {
    extern __shared__ unsigned char pyr_lds[];
    const int ksize = gk.ksize, r = ksize >> 1;
    const int ncolh = t.ow * ntap;                 // row-filtered columns kept per source row
    int *s_xs = (int *)pyr_lds;                    // [ow] source column of each output column
    float *s_xa = (float *)(s_xs + t.ow);          // [ow][2] horizontal lerp weights
    int *s_ys = (int *)(s_xa + 2 * t.ow);          // [oh]
    float *s_yb = (float *)(s_ys + t.oh);          // [oh][2]
    float *s_h = s_yb + 2 * t.oh;                  // [rh][ncolh] row-filtered samples
    unsigned char *s_src = (unsigned char *)(s_h + (size_t)t.rh * ncolh);  // [rh][cw] source bytes
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    const uint8_t *__restrict__ img = imgs.p[tbz];
    const size_t step = imgs.step[tbz];
    I += (size_t)tbz * I_stride;
    const int ox0 = tbx * t.ow, oy0 = tby * t.oh;
    const int tid = threadIdx.x;

    if (tid < t.ow) {
        int d = min(ox0 + tid, lw - 1), sx;
        float a0 = 1.f, a1 = 0.f;
        if (ntap == 1) sx = d;
        else lerp_coef(d, W, lw, sx, a0, a1);
        s_xs[tid] = sx;
        s_xa[2 * tid] = a0;
        s_xa[2 * tid + 1] = a1;
    } else if (tid >= 64 && tid < 64 + t.oh) {  // (ow <= 64)
        int q = tid - 64, d = min(oy0 + q, lh - 1), sy;
        float b0 = 1.f, b1 = 0.f;
        if (ntap == 1) sy = d;
        else lerp_coef(d, H, lh, sy, b0, b1);
        s_ys[q] = sy;
        s_yb[2 * q] = b0;
        s_yb[2 * q + 1] = b1;
    }
    __syncthreads();
    const int c_lo = s_xs[0] - r, r_lo = s_ys[0] - r;
    const int cw = min(s_xs[t.ow - 1] + (ntap - 1) + r - c_lo + 1, t.cw), rh = min(s_ys[t.oh - 1] + (ntap - 1) + r - r_lo + 1, t.rh);

    for (int e = tid; e < rh * cw; e += 256) {
        int ry = e / cw, rx = e - ry * cw;
        s_src[ry * t.cw + rx] = img[(size_t)reflect101(r_lo + ry, H) * step + reflect101(c_lo + rx, W)];
    }
    __syncthreads();
    // row filter at the needed columns (the second of a pair is clamped to W-1 like the unfused kernel)
    for (int e = tid; e < rh * ncolh; e += 256) {
        int ry = e / ncolh, j = e - ry * ncolh;
        int sx = s_xs[ntap == 1 ? j : (j >> 1)];
        if (ntap == 2) sx = min(sx + (j & 1), W - 1);
        const unsigned char *S = s_src + ry * t.cw + (sx - c_lo);  // S[i] = source column sx + i (reflected)
        float v;
        if (ksize == 3) v = madd((float)S[-1] + (float)S[1], gk.k[2], (float)S[0] * gk.k[1], gk.fc);
        else if (ksize == 5) v = madd((float)S[-2] + (float)S[2], gk.k[4], madd((float)S[-1] + (float)S[1], gk.k[3], (float)S[0] * gk.k[2], gk.fc), gk.fc);
        else {
            v = gk.k[0] * (float)S[-r];
            for (int q = 1; q < ksize; q++) v = madd((float)S[q - r], gk.k[q], v, gk.fc);
        }
        s_h[ry * ncolh + j] = v;
    }
    __syncthreads();
    const float *kc = gk.k + r;
    auto colf = [&](int j, int sy) -> float {  // column filter at source row sy, filtered column j
        const float *C = s_h + (sy - r_lo) * ncolh + j;
        if (ksize == 3) return madd(C[-ncolh] + C[ncolh], kc[1], C[0] * kc[0], gk.fc);
        float v = kc[0] * C[0];
        for (int q = 1; q <= r; q++) v = madd(C[q * ncolh] + C[-q * ncolh], kc[q], v, gk.fc);
        return v;
    };
    for (int e = tid; e < t.ow * t.oh; e += 256) {
        int ty = e / t.ow, tx = e - ty * t.ow;
        int dx = ox0 + tx, dy = oy0 + ty;
        if (dx >= lw || dy >= lh) continue;
        float out;
        if (ntap == 1) {
            out = colf(tx, s_ys[ty]);
        } else {
            const int sx = s_xs[tx], sy = s_ys[ty], sy1 = min(sy + 1, H - 1);
            const float ax0 = s_xa[2 * tx], ax1 = s_xa[2 * tx + 1], b0 = s_yb[2 * ty], b1 = s_yb[2 * ty + 1];
            float t00 = colf(2 * tx, sy), t10 = colf(2 * tx, sy1);
            if (sx + 1 < W) {
                float t01 = colf(2 * tx + 1, sy), t11 = colf(2 * tx + 1, sy1);
                out = resize_combine(t00, t01, t10, t11, ax0, ax1, b0, b1, area, gk.fc);
            } else {
                const float r0 = t00 * 1.f, r1 = t10 * 1.f;
                out = madd(r0, b0, r1 * b1, gk.fc);
            }
        }
        I[(size_t)dy * lw + dx] = out;
    }
}

// The fused tile kernel for the coarse levels of the default pyramid: the level is the frame divided by S = 4 or 8 in both
// directions exactly, so every output sample lies half-way between source columns S*d + S/2 - 1 and S*d + S/2 (rows alike) and
// the two row-filtered columns of an output column share KS - 1 of their KS + 1 source bytes.  Same operations in the same order
// as pyr_fused_kernel; what changes is how the bytes travel:
//  * the footprint is staged with aligned dword loads (the host checks base and row step; dwords that touch the image edge
//    take the byte path with reflected columns),
//  * a lane filters BOTH columns of an output column from one run of KS + 1 bytes: aligned LDS dwords, re-aligned by the
//    tile-uniform byte offset (v_alignbyte_b32), bytes converted with v_cvt_f32_ubyteN -- 6 LDS reads for 38 taps at KS = 19
//    instead of 38 byte reads,
//  * the column filter reads the two filtered columns of a row as one 8-byte LDS word and evaluates the four filtered samples
//    of an output sample (rows sy, sy + 1) from the 2r + 2 rows they share.
template <int S, int KS>
struct PyrAl {
    static constexpr int OW = 32, OH = 8, R = KS / 2;
    static constexpr int SPAN_C = (OW - 1) * S + 2 + 2 * R, SPAN_R = (OH - 1) * S + 2 + 2 * R;  // source columns / rows a tile touches
    static constexpr int ND = ((3 + SPAN_C + 3) / 4) | 1;  // staged dwords per row (origin aligned down by up to 3 bytes); odd: rows of a wavefront's two half-rows fall on different banks
    static constexpr int NB = (3 + KS + 1 + 3) / 4;        // aligned dwords that hold a lane's KS + 1 bytes at any byte offset
    static constexpr size_t lds_bytes = (size_t)SPAN_R * ND * 4 + (size_t)SPAN_R * OW * 2 * 4;
};
template <int S, int KS>
__global__ __launch_bounds__(256) void pyr_fused_al_kernel(ImgTab imgs, int W, int H, int lw, int lh, GaussTaps gk, float *__restrict__ I, size_t I_stride) {
    using G = PyrAl<S, KS>;
    constexpr int R = G::R, ND = G::ND, NB = G::NB, OW = G::OW, OH = G::OH;
    __shared__ unsigned s_src[G::SPAN_R * ND];
    __shared__ float s_h[G::SPAN_R * OW * 2];
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    const uint8_t *__restrict__ img = imgs.p[tbz];
    const size_t step = imgs.step[tbz];
    I += (size_t)tbz * I_stride;
    const int ox0 = tbx * OW, oy0 = tby * OH, tid = threadIdx.x;
    const int c_first = S * ox0 + S / 2 - 1 - R, r_lo = S * oy0 + S / 2 - 1 - R;  // first source column / row of the footprint
    const int c_lo = c_first & ~3, m = c_first - c_lo;                            // staging origin (a multiple of 4, may be negative)

    for (int e = tid; e < G::SPAN_R * ND; e += 256) {
        const int ry = e / ND, k = e - ry * ND, c = c_lo + 4 * k;
        const uint8_t *Srow = img + (size_t)reflect101(r_lo + ry, H) * step;
        unsigned v;
        if (c >= 0 && c + 3 < W) {
            v = *(const unsigned *)(Srow + c);
        } else {
            v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= (unsigned)Srow[reflect101(c + b, W)] << (8 * b);
        }
        s_src[e] = v;
    }
    __syncthreads();
    // row filter: lane (ry, tx) -> the filtered samples at source columns sx = S*(ox0+tx) + S/2 - 1 and sx + 1 of row ry
    for (int e = tid; e < G::SPAN_R * OW; e += 256) {
        const int ry = e / OW, tx = e - ry * OW;
        const unsigned *Wd = s_src + ry * ND + tx * (S / 4);  // the dword that holds byte (sx - R) - c_lo = m + S*tx
        unsigned w[NB], b[NB - 1];
#pragma unroll
        for (int i = 0; i < NB; i++) w[i] = Wd[i];
#pragma unroll
        for (int i = 0; i < NB - 1; i++) b[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], (unsigned)m);  // byte q of the run = byte q & 3 of b[q >> 2]
        auto B = [&](int q) __attribute__((always_inline)) { return (float)((b[q >> 2] >> (8 * (q & 3))) & 255u); };
        float va = gk.k[0] * B(0), vb = gk.k[0] * B(1);
#pragma unroll
        for (int q = 1; q < KS; q++) {
            va = madd(B(q), gk.k[q], va, gk.fc);
            vb = madd(B(q + 1), gk.k[q], vb, gk.fc);
        }
        *(float2 *)(s_h + (size_t)e * 2) = make_float2(va, vb);
    }
    __syncthreads();
    {
        const int ty = tid / OW, tx = tid - ty * OW;
        const int dx = ox0 + tx, dy = oy0 + ty;
        if (dx >= lw || dy >= lh) return;
        int sx, sy;
        float ax0, ax1, b0, b1;
        lerp_coef(dx, W, lw, sx, ax0, ax1);
        lerp_coef(dy, H, lh, sy, b0, b1);
        // rows sy - R .. sy + 1 + R of the two filtered columns: footprint rows S*ty .. S*ty + 2R + 1
        float2 c[2 * R + 2];
#pragma unroll
        for (int i = 0; i < 2 * R + 2; i++) c[i] = *(const float2 *)(s_h + ((size_t)(S * ty + i) * OW + tx) * 2);
        const float *kc = gk.k + R;
        float t00 = kc[0] * c[R].x, t01 = kc[0] * c[R].y, t10 = kc[0] * c[R + 1].x, t11 = kc[0] * c[R + 1].y;
#pragma unroll
        for (int q = 1; q <= R; q++) {
            t00 = madd(c[R + q].x + c[R - q].x, kc[q], t00, gk.fc);
            t01 = madd(c[R + q].y + c[R - q].y, kc[q], t01, gk.fc);
            t10 = madd(c[R + 1 + q].x + c[R + 1 - q].x, kc[q], t10, gk.fc);
            t11 = madd(c[R + 1 + q].y + c[R + 1 - q].y, kc[q], t11, gk.fc);
        }
        I[(size_t)dy * lw + dx] = resize_combine(t00, t01, t10, t11, ax0, ax1, b0, b1, 0, gk.fc);
    }
}

// 3-tap levels (k = 0: sigma 0 -> [1/4 1/2 1/4], identity resize; k = 1: sigma 0.5, half size): the footprint of an
// output sample is at most 4x4 source bytes, so each lane simply reads it through the L1 -- no staging, no barriers.
// Same operations in the same order as the generic kernels.
__global__ __launch_bounds__(256) void pyr_direct3_kernel(ImgTab imgs, int W, int H, int lw, int lh,
                                                          int ntap, float k0, float k1, double scale_x, double scale_y,
                                                          float *__restrict__ I, size_t I_stride, int area_fc) {
    const int area = area_fc & 15, fc = area_fc >> 4;  // (bit 4: filter contraction, see madd)
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    const uint8_t *__restrict__ img = imgs.p[tbz];
    const size_t step = imgs.step[tbz];
    I += (size_t)tbz * I_stride;
    const int dx = tbx * 64 + threadIdx.x, dy = tby * 4 + threadIdx.y;
    if (dx >= lw || dy >= lh) return;
    int sx = dx, sy = dy;
    float ax0 = 1.f, ax1 = 0.f, b0 = 1.f, b1 = 0.f;
    if (ntap == 2) {
        lerp_coef_scaled(dx, W, scale_x, sx, ax0, ax1);
        lerp_coef_scaled(dy, H, scale_y, sy, b0, b1);
    }
    // row filter of source row `ry` at column `cx`:  S[0]*k0 + (S[-1] + S[1])*k1
    auto rowf = [&](int ry, int cx) -> float {
        const uint8_t *S = img + (size_t)reflect101(ry, H) * step;
        return madd((float)S[reflect101(cx - 1, W)] + (float)S[reflect101(cx + 1, W)], k1, (float)S[cx] * k0, fc);
    };
    // column filter at source row `cy`:  (T[-1] + T[1])*k1 + T[0]*k0
    auto colf = [&](int cy, int cx) -> float { return madd(rowf(cy - 1, cx) + rowf(cy + 1, cx), k1, rowf(cy, cx) * k0, fc); };
    float out;
    if (ntap == 1) {
        out = colf(sy, sx);
    } else {
        const int sy1 = min(sy + 1, H - 1);
        float t00 = colf(sy, sx), t10 = colf(sy1, sx);
        if (sx + 1 < W) {
            const int sx1 = min(sx + 1, W - 1);
            float t01 = colf(sy, sx1), t11 = colf(sy1, sx1);
            out = resize_combine(t00, t01, t10, t11, ax0, ax1, b0, b1, area, fc);
        } else {
            const float r0 = t00 * 1.f, r1 = t10 * 1.f;
            out = madd(r0, b0, r1 * b1, fc);
        }
    }
    I[(size_t)dy * lw + dx] = out;
}

// Dword form of pyr_direct3_kernel for the two shapes the default pyramid has: k = 0 (same size) and k = 1 when the
// level is exactly half the frame.  A byte load costs the texture addresser as much per lane as a dword load, and
// the byte kernel issues 9 (k = 0) or 36 (k = 1) of them per output sample; here a lane reads three aligned dwords
// per source row -- the four source columns it owns plus the neighbour byte on either side -- and produces four
// (k = 0) or two (k = 1) horizontally adjacent samples from them: 2.25 / 6 loads per sample.  Lanes whose dwords
// would cross the image edge take the byte path with reflected columns.  Arithmetic and order as in the byte kernel.
template <int NTAP>
__global__ __launch_bounds__(256) void pyr_direct3v_kernel(ImgTab imgs, int W, int H, int lw, int lh,
                                                           float k0, float k1, float *__restrict__ I, size_t I_stride, int area_fc) {
    const int area = area_fc & 15, fc = area_fc >> 4;  // (bit 4: filter contraction, see madd)
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    const uint8_t *__restrict__ img = imgs.p[tbz];
    const size_t step = imgs.step[tbz];
    I += (size_t)tbz * I_stride;
    const int c0 = (tbx * 64 + threadIdx.x) * 4;  // first source column of this lane
    const int dy = tby * 4 + threadIdx.y;
    if (c0 >= W || dy >= lh) return;
    constexpr int NR = NTAP == 1 ? 3 : 4;          // source rows: sy-1 .. sy+1 (+ sy+2)
    const int sy = NTAP == 1 ? dy : 2 * dy;
    const bool fast = c0 >= 4 && c0 + 8 <= W;
    float rf[NR][4];  // row-filtered samples at columns c0 .. c0+3
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint8_t *S = img + (size_t)reflect101(sy - 1 + r, H) * step;
        float b[6];  // columns c0-1 .. c0+4
        if (fast) {
            const unsigned d0 = *(const unsigned *)(S + c0 - 4), d1 = *(const unsigned *)(S + c0), d2 = *(const unsigned *)(S + c0 + 4);
            b[0] = (float)(d0 >> 24);
            b[1] = (float)(d1 & 255u);
            b[2] = (float)((d1 >> 8) & 255u);
            b[3] = (float)((d1 >> 16) & 255u);
            b[4] = (float)(d1 >> 24);
            b[5] = (float)(d2 & 255u);
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) b[i] = (float)S[reflect101(min(c0 - 1 + i, W), W)];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) rf[r][j] = madd(b[j] + b[j + 2], k1, b[j + 1] * k0, fc);
    }
    // column filter at source row sy (+ sy+1):  (T[-1] + T[1])*k1 + T[0]*k0
    if (NTAP == 1) {
        float *out = I + (size_t)dy * lw + c0;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = madd(rf[0][j] + rf[2][j], k1, rf[1][j] * k0, fc);
        if ((lw & 3) == 0 && (((uintptr_t)I) & 15) == 0) {  // c0 is a multiple of 4: one aligned 16-byte store
            *(float4 *)out = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (c0 + j < lw) out[j] = v[j];
        }
    } else {
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t0[j] = madd(rf[0][j] + rf[2][j], k1, rf[1][j] * k0, fc);
            t1[j] = madd(rf[1][j] + rf[3][j], k1, rf[2][j] * k0, fc);
        }
        float *out = I + (size_t)dy * lw + (c0 >> 1);
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; q++) v[q] = resize_combine(t0[2 * q], t0[2 * q + 1], t1[2 * q], t1[2 * q + 1], 0.5f, 0.5f, 0.5f, 0.5f, area, fc);
        if ((lw & 1) == 0 && (((uintptr_t)I) & 7) == 0) {
            *(float2 *)out = make_float2(v[0], v[1]);
        } else {
#pragma unroll
            for (int q = 0; q < 2; q++)
                if ((c0 >> 1) + q < lw) out[q] = v[q];
        }
    }
}

// Wavefront-row form of pyr_direct3v_kernel (round 5).  That kernel is bound by the texture addresser, not by memory: 9 (k = 0) / 12 (k = 1) dword
// loads per lane for four / two samples, 166 MB of a level-0 launch at 2.3 TB/s.  Here a lane loads ONE dword per source row -- its own four
// columns -- and takes the byte on either side from its neighbour lanes (DPP wave shifts; lanes 0 and 63 only carry those bytes, 62 lanes x 4 = 248
// columns per wavefront, which is the same eight wavefronts across 1920 columns), and it walks ROWS output rows top to bottom so that every
// source row is loaded and row-filtered once per wavefront instead of three (k = 0) or two (k = 1) times: 0.31 / 1.25 loads per sample instead
// of 2.25 / 6.  Frames whose width is a multiple of four; the arithmetic and its order are pyr_direct3v_kernel's.
template <int NTAP, int ROWS>
__global__ __launch_bounds__(256) void pyr_direct3w_kernel(ImgTab imgs, int W, int H, int lw, int lh, float k0, float k1, float *__restrict__ I,
                                                           size_t I_stride, int area_fc) {
    const int area = area_fc & 15, fc = area_fc >> 4;  // (bit 4: filter contraction, see madd)
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    const uint8_t *__restrict__ img = imgs.p[tbz];
    const size_t step = imgs.step[tbz];
    I += (size_t)tbz * I_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = tbx * 248 + (lane - 1) * 4;          // first source column of this lane (lanes 0 / 63: the neighbours' bytes only)
    const int cl = min(max(c0, 0), W - 4);              // the dword it loads
    const bool left_edge = c0 == 0, right_edge = c0 + 4 == W;
    const bool own = lane >= 1 && lane <= 62 && c0 < W;
    const int dy0 = (tby * 4 + wave) * ROWS;             // first output row of this wavefront
    if (dy0 >= lh) return;                               // (wave-uniform)
    constexpr int NS = NTAP == 1 ? ROWS + 2 : 2 * ROWS + 2;  // source rows sy0 - 1 ..
    const int sy0 = NTAP == 1 ? dy0 : 2 * dy0;
    unsigned d[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) d[r] = *(const unsigned *)(img + (size_t)reflect101(min(sy0 - 1 + r, H), H) * step + cl);
    float rf[NS][4];  // row-filtered samples at columns c0 .. c0+3
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned hi = d[r] >> 24, lo = d[r] & 255u;
        const unsigned from_left = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x138, 0xf, 0xf, true);   // lane i <- lane i-1: column c0 - 1
        const unsigned from_right = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xf, 0xf, true);  // lane i <- lane i+1: column c0 + 4
        float b[6];
        b[1] = (float)lo;
        b[2] = (float)((d[r] >> 8) & 255u);
        b[3] = (float)((d[r] >> 16) & 255u);
        b[4] = (float)hi;
        b[0] = left_edge ? b[2] : (float)from_left;      // reflect101: column -1 is column 1
        b[5] = right_edge ? b[3] : (float)from_right;    // ... column W is column W - 2
#pragma unroll
        for (int j = 0; j < 4; j++) rf[r][j] = madd(b[j] + b[j + 2], k1, b[j + 1] * k0, fc);
    }
    if (!own) return;
    // column filter at source row sy (+ sy+1):  (T[-1] + T[1])*k1 + T[0]*k0
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const int dy = dy0 + i;
        if (dy >= lh) break;
        if (NTAP == 1) {
            float *out = I + (size_t)dy * lw + c0;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = madd(rf[i][j] + rf[i + 2][j], k1, rf[i + 1][j] * k0, fc);
            if ((((uintptr_t)I) & 15) == 0) {  // lw is a multiple of 4, c0 too: one aligned 16-byte store
                *(float4 *)out = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) out[j] = v[j];
            }
        } else {
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                t0[j] = madd(rf[2 * i][j] + rf[2 * i + 2][j], k1, rf[2 * i + 1][j] * k0, fc);
                t1[j] = madd(rf[2 * i + 1][j] + rf[2 * i + 3][j], k1, rf[2 * i + 2][j] * k0, fc);
            }
            float *out = I + (size_t)dy * lw + (c0 >> 1);
            float v[2];
#pragma unroll
            for (int q = 0; q < 2; q++) v[q] = resize_combine(t0[2 * q], t0[2 * q + 1], t1[2 * q], t1[2 * q + 1], 0.5f, 0.5f, 0.5f, 0.5f, area, fc);
            if ((lw & 1) == 0 && (((uintptr_t)I) & 7) == 0) {
                *(float2 *)out = make_float2(v[0], v[1]);
            } else {
                out[0] = v[0];
                out[1] = v[1];
            }
        }
    }
}

// ------------------------------------------------------------------ F3 polynomial expansion
//
// One 64x16 output tile per 256-thread block.  The tile of I plus an n-pixel halo is staged in
// LDS (rows and columns clamped = the replicate border of the reference), the vertical pass
// writes its three f32 sums per (row, column) back to LDS, and the horizontal pass accumulates
// the six moments in f64 exactly like the reference's inner loop.

constexpr int kPeTW = 64, kPeTH = 16;
typedef float ofxcv_f2 __attribute__((ext_vector_type(2)));
typedef float ofxcv_f4 __attribute__((ext_vector_type(4)));

// NT > 0: poly_n known at compile time (loops fully unrolled); NT == 0: run-time poly_n
template <int NT>
__global__ __launch_bounds__(256) void polyexp_kernel(const float *__restrict__ I, int w, int h, float *__restrict__ R,
                                                      int pitch, PolyCoef pc, size_t I_stride, size_t pair_stride, size_t field, int pack_odd) {
    extern __shared__ float lds[];
    const int n = NT > 0 ? NT : pc.n;
    const int cw = kPeTW + 2 * n;          // staged columns
    const int ldw = cw | 1;                // odd row stride: conflict-free column walks
    const int ih = kPeTH + 2 * n;          // staged rows
    float *sI = lds;                       // [ih][ldw]
    float *sV = lds + ih * ldw;            // [3][kPeTH][ldw]
    const int tid = threadIdx.x, lx = tid & 63, tq = tid >> 6;
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);  // z = frame: 2 * pair + {0 = prev, 1 = next}
    I += (size_t)tbz * I_stride;
    R += (size_t)(tbz >> 1) * pair_stride + (size_t)(tbz & 1) * field;
    const int x0 = tbx * kPeTW, y0 = tby * kPeTH;

    // stage I (rows and columns clamped = replicated border); lanes < 2n also fetch the extra halo columns
    const int gx0 = clampi(x0 + lx - n, 0, w - 1), gx1 = clampi(x0 + 64 + lx - n, 0, w - 1);
    for (int ry = tq; ry < ih; ry += 4) {
        const float *row = I + (size_t)clampi(y0 + ry - n, 0, h - 1) * w;
        sI[ry * ldw + lx] = row[gx0];
        if (lx < 2 * n) sI[ry * ldw + 64 + lx] = row[gx1];
    }
    __syncthreads();

    const float *g = pc.g + pc.n, *xg = pc.xg + pc.n, *xxg = pc.xxg + pc.n;
    // vertical pass (float): staged rows were clamped on load, so offsets +-k see replicated rows
    auto vertical = [&](int ty, int cx) {
        const float *col = sI + (ty + n) * ldw + cx;
        float t0 = col[0] * g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 1; k <= n; k++) {
            float s0 = col[-k * ldw], s1 = col[k * ldw];
            float p = s0 + s1;
            t0 = t0 + g[k] * p;
            t1 = t1 + xg[k] * (s1 - s0);
            t2 = t2 + xxg[k] * p;
        }
        sV[(0 * kPeTH + ty) * ldw + cx] = t0;
        sV[(1 * kPeTH + ty) * ldw + cx] = t1;
        sV[(2 * kPeTH + ty) * ldw + cx] = t2;
    };
    for (int ty = tq; ty < kPeTH; ty += 4) {
        vertical(ty, lx);
        if (lx < 2 * n) vertical(ty, 64 + lx);
    }
    __syncthreads();

    const size_t plane = (size_t)pitch * h;
    for (int ty = tq; ty < kPeTH; ty += 4) {
        int x = x0 + lx, y = y0 + ty;
        if (x >= w || y >= h) continue;
        const float *r0 = sV + (0 * kPeTH + ty) * ldw + lx + n;
        const float *r1 = sV + (1 * kPeTH + ty) * ldw + lx + n;
        const float *r2 = sV + (2 * kPeTH + ty) * ldw + lx + n;
        float g0 = g[0];
        double b1 = r0[0] * g0, b2 = 0, b3 = r1[0] * g0, b4 = 0, b5 = r2[0] * g0, b6 = 0;
#pragma unroll
        for (int k = 1; k <= n; k++) {
            double tg = r0[k] + r0[-k];
            g0 = g[k];
            b1 += tg * g0;
            b4 += tg * xxg[k];
            b2 += (r0[k] - r0[-k]) * xg[k];
            b3 += (r1[k] + r1[-k]) * g0;
            b6 += (r1[k] - r1[-k]) * xg[k];
            b5 += (r2[k] + r2[-k]) * g0;
        }
        size_t o = (size_t)y * pitch + x;
        const float c1 = (float)(b2 * pc.ig11), c0 = (float)(b3 * pc.ig11), c3 = (float)(b1 * pc.ig03 + b4 * pc.ig33),
                    c2 = (float)(b1 * pc.ig03 + b5 * pc.ig33), c4 = (float)(b6 * pc.ig55);
        if (pack_odd && (tbz & 1)) {  // the second frame of a pair: the packed form of an R1 field (TapsQ)
            ((ofxcv_f4 *)R)[o] = ofxcv_f4{c0, c1, c2, c3};
            R[o + 4 * plane] = c4;
        } else {
            R[o + 0 * plane] = c0;
            R[o + 1 * plane] = c1;
            R[o + 2 * plane] = c2;
            R[o + 3 * plane] = c3;
            R[o + 4 * plane] = c4;
        }
    }
}

// Persistent form for the compile-time neighbourhoods (poly_n 5 = the plugin default, 7).  Same arithmetic, in the
// same order per output sample, as polyexp_kernel; what changes is how it is scheduled:
//  * a workgroup loops over tiles (a contiguous run per XCD, interleaved between that XCD's workgroups) and requests the
//    next tile's samples into registers before it computes the current one, so the loads overlap the arithmetic;
//  * the vertical pass handles two neighbouring columns per lane with packed f32 instructions and stores its three sums
//    per column as one 16-byte LDS word {t0, t1, t1, t2}; the horizontal pass then needs one 16-byte LDS read per tap
//    and forms (t0,t1) differences and (t1,t2) sums with packed instructions before they enter the f64 accumulators.
// The kernel is VALU-bound (about 165 vector instructions per sample, a third of them f64).

template <int N, int TH>
__global__ __launch_bounds__(256) void polyexp_persistent_kernel(const float *__restrict__ Ib, int w, int h, float *__restrict__ Rb,
                                                                 int pitch, PolyCoef pc, int tiles_x, int ntiles_img, int nimg, size_t I_stride,
                                                                 size_t pair_stride, size_t field, int pack_odd) {
    constexpr int CW = kPeTW + 2 * N, LDW = CW + 2, IH = TH + 2 * N;  // staged columns / row stride (even) / rows
    constexpr int NSR = (IH + 3) / 4;                                 // staged rows per wavefront
    constexpr int NV = (CW / 2) * TH;                                 // column pairs x rows of the vertical pass
    __shared__ float sI[IH * LDW];
    __shared__ ofxcv_f4 sV[TH * CW];
    const int tid = threadIdx.x, lx = tid & 63, tq = tid >> 6;
    // XCD b % 8 works through tiles [lo, hi); its workgroups take them round-robin
    // (a batch puts the tiles of its 2 * n frames one after the other: tile t belongs to frame t / ntiles_img)
    const unsigned xcd = blockIdx.x & 7u, per = gridDim.x >> 3;
    const int ntiles = ntiles_img * nimg;
    const int lo = (int)((long)ntiles * xcd / 8), hi = (int)((long)ntiles * (xcd + 1) / 8);
    const float *g = pc.g + pc.n, *xg = pc.xg + pc.n, *xxg = pc.xxg + pc.n;
    const size_t plane = (size_t)pitch * h;

    // staging: wavefront tq fetches rows tq, tq + 4, ...; lane lx column lx, lanes < 2N also column 64 + lx
    float pre[NSR], pre2[NSR];
    auto request = [&](int tt) {
        const int im = tt / ntiles_img, t = tt - im * ntiles_img;
        const float *I = Ib + (size_t)im * I_stride;
        const int ty0 = t / tiles_x, x0 = (t - ty0 * tiles_x) * kPeTW, y0 = ty0 * TH;
        const int gx0 = clampi(x0 + lx - N, 0, w - 1), gx1 = clampi(x0 + 64 + lx - N, 0, w - 1);
#pragma unroll
        for (int i = 0; i < NSR; i++) {
            const int ry = tq + 4 * i;
            const float *row = I + (size_t)clampi(y0 + ry - N, 0, h - 1) * w;
            pre[i] = row[gx0];
            pre2[i] = lx < 2 * N ? row[gx1] : 0.f;
        }
    };
    int t = lo + (int)(blockIdx.x >> 3);
    if (t < hi) request(t);
    while (t < hi) {
#pragma unroll
        for (int i = 0; i < NSR; i++) {
            const int ry = tq + 4 * i;
            if (ry < IH) {
                sI[ry * LDW + lx] = pre[i];
                if (lx < 2 * N) sI[ry * LDW + 64 + lx] = pre2[i];
            }
        }
        __syncthreads();
        const int tn = t + (int)per;
        if (tn < hi) request(tn);

        // vertical pass (f32), two columns per lane
        for (int e = tid; e < NV; e += 256) {
            const int row = e / (CW / 2), c2 = (e - row * (CW / 2)) * 2;
            const float *col = sI + (row + N) * LDW + c2;
            const ofxcv_f2 v0 = *(const ofxcv_f2 *)col;
            ofxcv_f2 t0 = v0 * g[0], t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const ofxcv_f2 s0 = *(const ofxcv_f2 *)(col - k * LDW), s1 = *(const ofxcv_f2 *)(col + k * LDW);
                const ofxcv_f2 p = s0 + s1;
                t0 = t0 + g[k] * p;
                t1 = t1 + xg[k] * (s1 - s0);
                t2 = t2 + xxg[k] * p;
            }
            sV[row * CW + c2] = ofxcv_f4{t0.x, t1.x, t1.x, t2.x};
            sV[row * CW + c2 + 1] = ofxcv_f4{t0.y, t1.y, t1.y, t2.y};
        }
        __syncthreads();

        // horizontal pass (f64 accumulators)
        const int im = t / ntiles_img, tl = t - im * ntiles_img;
        float *R = Rb + (size_t)(im >> 1) * pair_stride + (size_t)(im & 1) * field;
        const int ty0 = tl / tiles_x, x0 = (tl - ty0 * tiles_x) * kPeTW, y0 = ty0 * TH;
        const int x = x0 + lx;
#pragma unroll
        for (int i = 0; i < TH / 4; i++) {
            const int ty = tq + 4 * i, y = y0 + ty;
            const ofxcv_f4 *v = sV + ty * CW + lx + N;
            const ofxcv_f4 c = v[0];
            const float g0 = g[0];
            double b1 = c.x * g0, b2 = 0, b3 = c.y * g0, b4 = 0, b5 = c.w * g0, b6 = 0;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const ofxcv_f4 A = v[k], B = v[-k];
                const double tg = A.x + B.x;
                const ofxcv_f2 d = (ofxcv_f2{A.x, A.y} - ofxcv_f2{B.x, B.y}) * xg[k];
                const ofxcv_f2 sm = (ofxcv_f2{A.z, A.w} + ofxcv_f2{B.z, B.w}) * g[k];
                b1 += tg * g[k];
                b4 += tg * xxg[k];
                b2 += d.x;
                b3 += sm.x;
                b6 += d.y;
                b5 += sm.y;
            }
            if (x < w && y < h) {
                const size_t o = (size_t)y * pitch + x;
                const float c1 = (float)(b2 * pc.ig11), c0 = (float)(b3 * pc.ig11), c3 = (float)(b1 * pc.ig03 + b4 * pc.ig33),
                            c2 = (float)(b1 * pc.ig03 + b5 * pc.ig33), c4 = (float)(b6 * pc.ig55);
                if (pack_odd && (im & 1)) {  // the second frame of a pair: the packed form of an R1 field (TapsQ), one 16-byte store per lane
                    ((ofxcv_f4 *)R)[o] = ofxcv_f4{c0, c1, c2, c3};
                    R[o + 4 * plane] = c4;
                } else {
                    R[o + 0 * plane] = c0;
                    R[o + 1 * plane] = c1;
                    R[o + 2 * plane] = c2;
                    R[o + 3 * plane] = c3;
                    R[o + 4 * plane] = c4;
                }
            }
        }
        __syncthreads();
        t = tn;
    }
}

// ------------------------------------------------------------------ F4 update matrices (per pixel)

struct M5 {
    float v[5];
};

// Buffer addressing: the 128-bit descriptor and the row/plane part of every address are wave-uniform
// (SGPRs: descriptor + soffset), the lane's column is a 32-bit voffset -- no 64-bit vector address math.
// Out-of-range offsets are bounds-checked by the hardware (loads return 0, stores are dropped).
struct Buf {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ Buf make_buf(const void *p, size_t bytes) {
    Buf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
    return b;
}
// AUX: cache policy bits of the instruction (gfx94x / gfx950: 1 = sc0, 2 = nt, 16 = sc1); 0 everywhere except where a kernel streams a field once
template <int AUX = 0>
__device__ __forceinline__ float buf_ld(const Buf &b, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)voff_bytes, (int)soff_bytes, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void buf_st(const Buf &b, float v, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, (int)voff_bytes, (int)soff_bytes, AUX);
}
#ifndef OFXCV_COL_LD_AUX
#define OFXCV_COL_LD_AUX 0
#endif
#ifndef OFXCV_COL_R0_AUX
#define OFXCV_COL_R0_AUX 0
#endif
#ifndef OFXCV_COL_ST_AUX
#define OFXCV_COL_ST_AUX 0
#endif
// two horizontally adjacent taps of one plane.  Written as two dword loads; the compiler merges each pair into one
// buffer_load_dwordx2.  Measured on the fused iteration kernel: keeping them apart (20 gather instructions per pixel
// instead of 10) makes the launch 46 -> 56 us -- for gathers the per-instruction address work dominates, unlike the
// coalesced streaming loads where a dword wave-load is the cheapest form (tools/ubench/l1rate.hip).
struct TapPair {
    float a, b;
};

// R1 taps of one pixel: the 2x2 bilinear footprint of all five planes.  Pixels whose sample falls
// outside the image load a dummy in-range address instead of branching; `inb` selects afterwards.
struct Taps {
    TapPair t[5], b[5];
    float fx, fy;
    bool inb;
};

__device__ __forceinline__ Taps gather_taps(const Buf &R1, int x, int y, int w, int h, int pitch, unsigned plane_bytes,
                                            float dx, float dy) {
    Taps tp;
    float fx = x + dx, fy = y + dy;
    int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    tp.fx = fx - x1;
    tp.fy = fy - y1;
    tp.inb = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
    const unsigned o0 = tp.inb ? ((unsigned)y1 * (unsigned)pitch + (unsigned)x1) * 4u : 0u, o1 = o0 + (unsigned)pitch * 4u;
#pragma unroll
    for (int c = 0; c < 5; c++) {  // wave-uniform plane offset in soffset, 32-bit lane byte offset in voffset
        tp.t[c].a = buf_ld(R1, o0, c * plane_bytes);
        tp.t[c].b = buf_ld(R1, o0 + 4u, c * plane_bytes);
        tp.b[c].a = buf_ld(R1, o1, c * plane_bytes);
        tp.b[c].b = buf_ld(R1, o1 + 4u, c * plane_bytes);
    }
    return tp;
}

// The same footprint from the PACKED form of an R1 field -- what a call's polynomial expansion writes for the SECOND frame of a pair (the
// field that is only ever gathered; R0 is streamed and stays planar): per pixel a float4 {c0, c1, c2, c3} ([h][pitch] float4), then plane 4
// ([h][pitch] floats) -- the same 5 * pitch * h floats as the planar field.  Four 16-byte gathers + two 8-byte ones per pixel instead of ten
// 8-byte ones: the texture addresser is the busiest unit of the iteration kernels (TA_BUSY 74-83 % of the two-iteration launch, every gather ~37
// of its cycles: profiles/r05_experiments.md), and a dwordx4 wave-load costs it what a dwordx2 one does (tools/ubench/l1rate.hip).
// Measured: 416 -> 377 us per (iterate, iterate) launch of 8 x 1080p, 1 601 -> 1 471 us at 3840x2160, same bits.
// The stage-level entry points (ofxcv_farneback_polyexp / _update_matrices / _update_flow_blur) keep planar fields: their kernels take the
// layout as a flag.
struct TapsQ {
    ofxcv_f4 t0, t1, b0, b1;
    TapPair t4, b4;
    float fx, fy;
    bool inb;
};
__device__ __forceinline__ ofxcv_f4 buf_ld4(const Buf &b, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(ofxcv_f4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ TapsQ gather_taps_q(const Buf &R1, int x, int y, int w, int h, int pitch, unsigned plane_bytes, float dx, float dy) {
    TapsQ tp;
    float fx = x + dx, fy = y + dy;
    int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    tp.fx = fx - x1;
    tp.fy = fy - y1;
    tp.inb = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
    const unsigned o = tp.inb ? (unsigned)y1 * (unsigned)pitch + (unsigned)x1 : 0u;
    const unsigned oq = o * 16u, rq = (unsigned)pitch * 16u, o4 = o * 4u, r4 = (unsigned)pitch * 4u;
    tp.t0 = buf_ld4(R1, oq, 0);
    tp.t1 = buf_ld4(R1, oq + 16u, 0);
    tp.b0 = buf_ld4(R1, oq + rq, 0);
    tp.b1 = buf_ld4(R1, oq + rq + 16u, 0);
    tp.t4.a = buf_ld(R1, o4, 4 * plane_bytes);
    tp.t4.b = buf_ld(R1, o4 + 4u, 4 * plane_bytes);
    tp.b4.a = buf_ld(R1, o4 + r4, 4 * plane_bytes);
    tp.b4.b = buf_ld(R1, o4 + r4 + 4u, 4 * plane_bytes);
    return tp;
}
// the four taps of channel c in the order (top left, top right, bottom left, bottom right)
__device__ __forceinline__ void tap4(const Taps &tp, int c, float &ta, float &tb, float &ba, float &bb) {
    ta = tp.t[c].a; tb = tp.t[c].b; ba = tp.b[c].a; bb = tp.b[c].b;
}
__device__ __forceinline__ void tap4(const TapsQ &tp, int c, float &ta, float &tb, float &ba, float &bb) {
    if (c < 4) { ta = tp.t0[c]; tb = tp.t1[c]; ba = tp.b0[c]; bb = tp.b1[c]; }
    else { ta = tp.t4.a; tb = tp.t4.b; ba = tp.b4.a; bb = tp.b4.b; }
}

// F4 in three parts (the column-owning kernel applies the border scale in its own, hoisted form):
// the warped sample and r2..r6 before the border scale ...
template <typename TAPS>
__device__ __forceinline__ void um_sample(const float r0v[5], const TAPS &tp, float dx, float dy, float (&r)[5]) {
    const float fx = tp.fx, fy = tp.fy;
    float r2, r3, r4, r5, r6;
    {
        float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        float rr[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            float ta, tb, ba, bb;
            tap4(tp, c, ta, tb, ba, bb);
            rr[c] = a00 * ta + a01 * tb + a10 * ba + a11 * bb;
        }
        r2 = rr[0]; r3 = rr[1]; r4 = rr[2]; r5 = rr[3]; r6 = rr[4];
        r4 = (r0v[2] + r4) * 0.5f;
        r5 = (r0v[3] + r5) * 0.5f;
        r6 = (r0v[4] + r6) * 0.25f;
    }
    if (!tp.inb) {
        r2 = r3 = 0.f;
        r4 = r0v[2];
        r5 = r0v[3];
        r6 = r0v[4] * 0.5f;
    }
    r2 = (r0v[0] - r2) * 0.5f;
    r3 = (r0v[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    r[0] = r2; r[1] = r3; r[2] = r4; r[3] = r5; r[4] = r6;
}
// ... border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472} by the distance d to an image edge ...
constexpr int kUmBorder = 5;
__device__ __forceinline__ float um_border(int d) { return d < 2 ? 0.14f : (d < kUmBorder ? 0.4472f : 1.f); }
// ... and the five products
__device__ __forceinline__ M5 um_products(const float (&r)[5]) {
    const float r2 = r[0], r3 = r[1], r4 = r[2], r5 = r[3], r6 = r[4];
    M5 m;
    m.v[0] = r4 * r4 + r6 * r6;
    m.v[1] = (r4 + r5) * r6;
    m.v[2] = r5 * r5 + r6 * r6;
    m.v[3] = r4 * r2 + r6 * r3;
    m.v[4] = r6 * r2 + r5 * r3;
    return m;
}

// either layout into the packed structure (packed: wave-uniform flag)
__device__ __forceinline__ TapsQ gather_taps_any(const Buf &R1, bool packed, int x, int y, int w, int h, int pitch, unsigned plane_bytes, float dx, float dy) {
    if (packed) return gather_taps_q(R1, x, y, w, h, pitch, plane_bytes, dx, dy);
    const Taps p = gather_taps(R1, x, y, w, h, pitch, plane_bytes, dx, dy);
    TapsQ q;
    q.t0 = ofxcv_f4{p.t[0].a, p.t[1].a, p.t[2].a, p.t[3].a};
    q.t1 = ofxcv_f4{p.t[0].b, p.t[1].b, p.t[2].b, p.t[3].b};
    q.b0 = ofxcv_f4{p.b[0].a, p.b[1].a, p.b[2].a, p.b[3].a};
    q.b1 = ofxcv_f4{p.b[0].b, p.b[1].b, p.b[2].b, p.b[3].b};
    q.t4 = p.t[4];
    q.b4 = p.b[4];
    q.fx = p.fx;
    q.fy = p.fy;
    q.inb = p.inb;
    return q;
}

template <typename TAPS>
__device__ __forceinline__ M5 update_matrices_finish(const float r0v[5], const TAPS &tp, int x, int y, int w, int h, float dx, float dy) {
    float r[5];
    um_sample(r0v, tp, dx, dy, r);
    constexpr int BORDER = kUmBorder;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        const float scale = um_border(x) * um_border(w - x - 1) * um_border(y) * um_border(h - y - 1);
#pragma unroll
        for (int c = 0; c < 5; c++) r[c] *= scale;
    }
    return um_products(r);
}

__device__ __forceinline__ M5 update_matrices_core(const float r0v[5], const float *__restrict__ R1, int x, int y, int w, int h,
                                                   int pitch, size_t plane, float dx, float dy, bool r1q) {
    const TapsQ tp = gather_taps_any(make_buf(R1, 5 * plane * sizeof(float)), r1q, x, y, w, h, pitch, (unsigned)(plane * 4), dx, dy);
    return update_matrices_finish(r0v, tp, x, y, w, h, dx, dy);
}

// r1q: R1 is in its packed form (TapsQ; wave-uniform)
__device__ __forceinline__ M5 update_matrices_px(const float *__restrict__ R0, const float *__restrict__ R1, int x, int y,
                                                 int w, int h, int pitch, float dx, float dy, bool r1q) {
    const size_t plane = (size_t)pitch * h;
    const size_t o = (size_t)y * pitch + x;
    float r0v[5];
#pragma unroll
    for (int c = 0; c < 5; c++) r0v[c] = R0[o + c * plane];
    return update_matrices_core(r0v, R1, x, y, w, h, pitch, plane, dx, dy, r1q);
}

// F6: the flow of the coarser level at pixel (x, y) of this one: resize INTER_LINEAR, then * 1/pyr_scale
struct Prolong {
    int pw, ph;                             // size of the coarser level
    double inv_pyr_scale, scale_x, scale_y;  // scale = (double)pw / w, divided once on the host
    int fc = 0;                              // filter contraction (madd): resize's vertical lerp as a fused multiply-add
};
__device__ __forceinline__ void prolong_flow(const float *__restrict__ flow, size_t flow_step, const Prolong &pr, int x, int y, float &dx, float &dy) {
    int sx, sy;
    float ax0, ax1, b0, b1;
    lerp_coef_scaled(x, pr.pw, pr.scale_x, sx, ax0, ax1);
    lerp_coef_scaled(y, pr.ph, pr.scale_y, sy, b0, b1);
    int sy1 = min(sy + 1, pr.ph - 1);
    const float2 *S0 = (const float2 *)((const char *)flow + (size_t)sy * flow_step);
    const float2 *S1 = (const float2 *)((const char *)flow + (size_t)sy1 * flow_step);
    float r0x, r0y, r1x, r1y;
    if (sx + 1 < pr.pw) {
        float2 a = S0[sx], b = S0[sx + 1], c = S1[sx], d = S1[sx + 1];
        r0x = a.x * ax0 + b.x * ax1; r0y = a.y * ax0 + b.y * ax1;
        r1x = c.x * ax0 + d.x * ax1; r1y = c.y * ax0 + d.y * ax1;
    } else {
        float2 a = S0[sx], c = S1[sx];
        r0x = a.x * 1.f; r0y = a.y * 1.f;
        r1x = c.x * 1.f; r1y = c.y * 1.f;
    }
    dx = (float)((double)madd(r0x, b0, r1x * b1, pr.fc) * pr.inv_pyr_scale);
    dy = (float)((double)madd(r0y, b0, r1y * b1, pr.fc) * pr.inv_pyr_scale);
}

// F6 + first F4 of a level.  MODE 0: zero initial flow (coarsest level); MODE 1: flow prolongated from
// the coarser level (resize INTER_LINEAR, then * 1/pyr_scale); MODE 2: explicit interleaved flow.
template <int MODE>
__global__ __launch_bounds__(256) void update_matrices_kernel(const float *__restrict__ R0, const float *__restrict__ R1,
                                                              FlowTab flows, int pw, int ph,
                                                              double inv_pyr_scale, double scale_x, double scale_y, int w, int h,
                                                              int pitch, float *__restrict__ M, size_t pair_stride, int r1q) {
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    int x = tbx * 64 + threadIdx.x;
    int y = tby * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    R0 += (size_t)tbz * pair_stride;
    R1 += (size_t)tbz * pair_stride;
    M += (size_t)tbz * pair_stride;
    const float *__restrict__ flow = MODE ? flows.p[tbz] : nullptr;
    const size_t flow_step = MODE ? flows.step[tbz] : 0;
    float dx = 0.f, dy = 0.f;
    if (MODE == 1) {
        const Prolong pr = {pw, ph, inv_pyr_scale, scale_x, scale_y, r1q >> 1};  // (bit 1 of the layout flag: filter contraction)
        prolong_flow(flow, flow_step, pr, x, y, dx, dy);
    } else if (MODE == 2) {
        float2 f = *(const float2 *)((const char *)flow + (size_t)y * flow_step + (size_t)x * 8);
        dx = f.x;
        dy = f.y;
    }
    M5 m = update_matrices_px(R0, R1, x, y, w, h, pitch, dx, dy, (r1q & 1) != 0);
    const size_t plane = (size_t)pitch * h, o = (size_t)y * pitch + x;
#pragma unroll
    for (int c = 0; c < 5; c++) M[o + c * plane] = m.v[c];
}

// ------------------------------------------------------------------ F5 (+F4) one iteration
//
// flow = solve(box(M_in)); if UPDATE, M_out = UpdateMatrices(R0, R1, flow) in the same pass so the
// flow never leaves the registers.  Box sums: horizontal f64 sum of each window row, left to right,
// then the rows top to bottom (replicated borders = clamped coordinates).
template <bool UPDATE>
__global__ __launch_bounds__(256) void blur_solve_update_kernel(const float *__restrict__ R0, const float *__restrict__ R1,
                                                                const float *__restrict__ Min, float *__restrict__ Mout,
                                                                float *__restrict__ flow, size_t flow_step, int w, int h,
                                                                int pitch, int m, double scale, int r1q) {
    int tbx, tby;
    xcd_tile(tbx, tby);
    int x = tbx * 64 + threadIdx.x;
    int y = tby * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)pitch * h;
    double acc[5];
    for (int i = -m; i <= m; i++) {
        const float *row = Min + (size_t)clampi(y + i, 0, h - 1) * pitch;
        double hs[5];
        for (int j = -m; j <= m; j++) {
            int xx = clampi(x + j, 0, w - 1);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                double v = (double)row[xx + c * plane];
                hs[c] = (j == -m) ? v : hs[c] + v;
            }
        }
#pragma unroll
        for (int c = 0; c < 5; c++) acc[c] = (i == -m) ? hs[c] : acc[c] + hs[c];
    }
    double g11_ = acc[0] * scale, g12_ = acc[1] * scale, g22_ = acc[2] * scale, h1_ = acc[3] * scale, h2_ = acc[4] * scale;
    double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
    float fxv = (float)((g11_ * h2_ - g12_ * h1_) * idet);
    float fyv = (float)((g22_ * h1_ - g12_ * h2_) * idet);
    if (flow) *(float2 *)((char *)flow + (size_t)y * flow_step + (size_t)x * 8) = make_float2(fxv, fyv);
    if (UPDATE) {
        M5 mm = update_matrices_px(R0, R1, x, y, w, h, pitch, fxv, fyv, r1q != 0);
        const size_t o = (size_t)y * pitch + x;
#pragma unroll
        for (int c = 0; c < 5; c++) Mout[o + c * plane] = mm.v[c];
    }
}

// ------------------------------------------------------------------ OPTFLOW_FARNEBACK_GAUSSIAN window
//
// FarnebackUpdateFlow_GaussianBlur: separable Gaussian window (sigma = (winsize/2) * 0.3), both passes accumulate in
// f32 in the reference's order  v = c*k[0]; for i = 1..m: v += (plus_i + minus_i) * k[i],  borders replicated.  Two
// kernels per iteration: the vertical pass writes its five sums per pixel to a scratch field, the horizontal pass
// finishes the window, solves and (UPDATE) evaluates the next M in the same pass.
constexpr int kMaxWinTaps = 64;  // winsize <= 127
struct WinTaps {
    int m;
    float k[kMaxWinTaps];
};

__global__ __launch_bounds__(256) void gauss_vpass_kernel(const float *__restrict__ M, int w, int h, int pitch, WinTaps t, float *__restrict__ V) {
    int tbx, tby;
    xcd_tile(tbx, tby);
    const int x = tbx * 64 + threadIdx.x, y = tby * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)pitch * h;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const float *P = M + c * plane + x;
        float v = P[(size_t)y * pitch] * t.k[0];
        for (int i = 1; i <= t.m; i++) v += (P[(size_t)min(y + i, h - 1) * pitch] + P[(size_t)max(y - i, 0) * pitch]) * t.k[i];
        V[c * plane + (size_t)y * pitch + x] = v;
    }
}

template <bool UPDATE>
__global__ __launch_bounds__(256) void gauss_hpass_solve_kernel(const float *__restrict__ R0, const float *__restrict__ R1,
                                                                const float *__restrict__ V, float *__restrict__ Mout,
                                                                float *__restrict__ flow, size_t flow_step, int w, int h, int pitch,
                                                                WinTaps t, int r1q) {
    int tbx, tby;
    xcd_tile(tbx, tby);
    const int x = tbx * 64 + threadIdx.x, y = tby * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)pitch * h;
    float sum[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const float *row = V + c * plane + (size_t)y * pitch;
        float v = row[x] * t.k[0];
        for (int i = 1; i <= t.m; i++) v += (row[min(x + i, w - 1)] + row[max(x - i, 0)]) * t.k[i];
        sum[c] = v;
    }
    const double g11 = sum[0], g12 = sum[1], g22 = sum[2], h1 = sum[3], h2 = sum[4];
    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
    const float fxv = (float)((g11 * h2 - g12 * h1) * idet), fyv = (float)((g22 * h1 - g12 * h2) * idet);
    if (flow) *(float2 *)((char *)flow + (size_t)y * flow_step + (size_t)x * 8) = make_float2(fxv, fyv);
    if (UPDATE) {
        M5 mm = update_matrices_px(R0, R1, x, y, w, h, pitch, fxv, fyv, r1q != 0);
        const size_t o = (size_t)y * pitch + x;
#pragma unroll
        for (int c = 0; c < 5; c++) Mout[o + c * plane] = mm.v[c];
    }
}

// ------------------------------------------------------------------ OPTFLOW_USE_INITIAL_FLOW
//
// Top pyramid level: flow = resize(flow0, level size, INTER_AREA) * scale (imgproc resize.cpp, f32, shrinking).
// Integer factors: ResizeAreaFast_ (row-major cell sum, four at a time, times 1/area); other factors: ResizeArea_ with
// the computeResizeAreaTab weights.  One thread per destination pixel, both channels.
struct AreaTaps {
    int first;          // first source cell
    int n;              // number of cells
    float a0, am, a1;   // weight of the first, the middle and the last cell
    bool has0, has1;    // partial first / last cell present
};
__device__ __forceinline__ void area_taps(int d, int ssize, double scale, int &sx1, int &sx2, bool &left, float &al, float &am, bool &right, float &ar) {
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, ssize - f1);
    sx1 = (int)ceil(f1);
    sx2 = min((int)floor(f2), ssize - 1);
    sx1 = min(sx1, sx2);
    left = sx1 - f1 > 1e-3;
    al = (float)((sx1 - f1) / cell);
    am = (float)(1.0 / cell);
    right = f2 - sx2 > 1e-3;
    ar = (float)(fmin(fmin(f2 - sx2, 1.), cell) / cell);
}

__global__ __launch_bounds__(256) void initial_flow_kernel(const float *__restrict__ flow0, size_t flow0_step, int W, int H,
                                                           float *__restrict__ dst, int w, int h, double mul) {
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= w || dy >= h) return;
    auto src = [&](int sy, int sx, int c) { return ((const float *)((const char *)flow0 + (size_t)sy * flow0_step))[(size_t)sx * 2 + c]; };
    float out[2];
    if (W == w && H == h) {
        out[0] = src(dy, dx, 0);
        out[1] = src(dy, dx, 1);
    } else {
        const double scale_x = (double)W / w, scale_y = (double)H / h;
        const int ix = (int)scale_x, iy = (int)scale_y;
        if (fabs(scale_x - ix) < DBL_EPSILON && fabs(scale_y - iy) < DBL_EPSILON) {
            const int area = ix * iy;
            const float scale = 1.f / area;
            for (int c = 0; c < 2; c++) {
                auto S = [&](int k) { return src(dy * iy + k / ix, dx * ix + k % ix, c); };
                float sum = 0;
                int k = 0;
                for (; k <= area - 4; k += 4) sum += S(k) + S(k + 1) + S(k + 2) + S(k + 3);
                for (; k < area; k++) sum += S(k);
                out[c] = sum * scale;
            }
        } else {
            int x1, x2, y1, y2;
            bool xl, xr, yl, yr;
            float xal, xam, xar, yal, yam, yar;
            area_taps(dx, W, scale_x, x1, x2, xl, xal, xam, xr, xar);
            area_taps(dy, H, scale_y, y1, y2, yl, yal, yam, yr, yar);
            for (int c = 0; c < 2; c++) {
                auto hrow = [&](int sy) {
                    float buf = 0;
                    if (xl) buf = buf + src(sy, x1 - 1, c) * xal;
                    for (int sx = x1; sx < x2; sx++) buf = buf + src(sy, sx, c) * xam;
                    if (xr) buf = buf + src(sy, x2, c) * xar;
                    return buf;
                };
                float sum = 0;
                bool first = true;
                auto vadd = [&](int sy, float beta) {
                    const float b = hrow(sy);
                    sum = first ? beta * b : sum + beta * b;
                    first = false;
                };
                if (yl) vadd(y1 - 1, yal);
                for (int sy = y1; sy < y2; sy++) vadd(sy, yam);
                if (yr) vadd(y2, yar);
                out[c] = sum;
            }
        }
    }
    dst[((size_t)dy * w + dx) * 2] = (float)(out[0] * mul);
    dst[((size_t)dy * w + dx) * 2 + 1] = (float)(out[1] * mul);
}

// ------------------------------------------------------------------ OpenCV-rounding mode of the box window
//
// FarnebackUpdateFlow_Blur keeps a running vertical sum per column and channel,
//     vsum(y) = vsum(y-1) + (double)(float)(M[min(y+1,h-1)] - M[max(y-2,0)]),   vsum(-1) = (double)(float)(3 * M[0]),
// i.e. every row difference is rounded to f32 before it is accumulated in f64.  That rounding noise is part of
// OpenCV's result; at ill-conditioned pixels it is amplified past 1e-4.  The default kernels above sum each window
// directly (no such noise).  With the context option "farneback.opencv_rounding" the iteration is evaluated the
// reference's way instead: one thread per (column, channel) walks the rows sequentially -- the recurrence is a true
// serial dependency -- and stores vsum(y) as f64 planes; a second kernel adds the three columns and does the solve
// and the matrix update.  This is a validation mode (about 20x slower), used by the parity tests to show that the
// GPU path matches the faithful oracle at every sample once the same rounding is applied.
__global__ __launch_bounds__(256) void strict_colscan_kernel(const float *__restrict__ M, int w, int h, int pitch, double *__restrict__ V) {
    const int x = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (x >= w) return;
    const float *m = M + (size_t)c * pitch * h + x;
    double *v = V + (size_t)c * pitch * h + x;
    double acc = (double)(m[0] * 3.f);  // vsum[x] = srow0[x]*(m+2), a float product
    for (int y = 0; y < h; y++) {
        const float a = m[(size_t)min(y + 1, h - 1) * pitch], b = m[(size_t)max(y - 2, 0) * pitch];
        acc += (double)(a - b);
        v[(size_t)y * pitch] = acc;
    }
}

template <bool UPDATE>
__global__ __launch_bounds__(256) void strict_solve_kernel(const float *__restrict__ R0, const float *__restrict__ R1, const double *__restrict__ V,
                                                           float *__restrict__ Mout, float *__restrict__ flow, size_t flow_step, int w, int h,
                                                           int pitch, double scale, int r1q) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)pitch * h;
    const int xm = max(x - 1, 0), xp = min(x + 1, w - 1);
    double acc[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const double *row = V + c * plane + (size_t)y * pitch;
        acc[c] = (row[xm] + row[x]) + row[xp];  // the reference's horizontal running sum, evaluated per pixel (f64 on f64)
    }
    double g11_ = acc[0] * scale, g12_ = acc[1] * scale, g22_ = acc[2] * scale, h1_ = acc[3] * scale, h2_ = acc[4] * scale;
    double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
    float fxv = (float)((g11_ * h2_ - g12_ * h1_) * idet);
    float fyv = (float)((g22_ * h1_ - g12_ * h2_) * idet);
    if (flow) *(float2 *)((char *)flow + (size_t)y * flow_step + (size_t)x * 8) = make_float2(fxv, fyv);
    if (UPDATE) {
        M5 mm = update_matrices_px(R0, R1, x, y, w, h, pitch, fxv, fyv, r1q != 0);
        const size_t o = (size_t)y * pitch + x;
#pragma unroll
        for (int c = 0; c < 5; c++) Mout[o + c * plane] = mm.v[c];
    }
}

// ------------------------------------------------------------------ OpenCV-order box window, strip-parallel: shared helpers
//
// The running sum above is vsum(y) = c0 + sum_{t<=y} (double)d_t with d_t = (float)(M[min(t+1,h-1)] - M[max(t-2,0)]) and
// c0 = (double)(3.f * M[0]): a column prefix of row differences that were rounded to f32.  The strip-parallel forms below
// reproduce the d_t exactly and only re-associate the f64 additions (partial sums per wavefront, strip or round: errors of
// 1e-16 relative to the partial sums, nine orders of magnitude below the f32 rounding of the d_t themselves).  The left / right
// column sums of the 3-column window come from the neighbouring lanes by DPP wave shifts (two dwords per f64).
// bound_ctrl form with a zero `old` operand: ONE v_mov_b32_dpp per dword (the form update_dpp(x, x, ...) costs a copy first: 4 instead of 2
// instructions per f64 shift, 20 of the ~200 vector instructions of a pixel row).  The lane without a source (0 / 63) receives 0: it is
// a halo lane in every kernel that uses these, its window sum is never used.
__device__ __forceinline__ double dpp64_from_left(double v) {  // lane i <- lane i-1
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp64_from_right(double v) {  // lane i <- lane i+1
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int kSsW = 62;  // columns a wavefront of the overlapped-strip form owns (lanes 1..62; lanes 0 and 63 carry the halo columns)

// Rows of a strip over the wavefronts of its workgroup.  A strip of `sh` rows (NW*(RW-1) < sh <= NW*RW) is cut into NW
// wavefronts of RW-1 or RW rows: the first sh - NW*(RW-1) wavefronts take RW.  With sh = NW*RW every wavefront has RW rows
// (VAR = false: known at compile time).  The strip height is free because the launch should not end with a nearly empty
// round of workgroups: 1080 rows in 32-row strips are 34 x 31 = 1054 workgroups on 512 resident slots (two full rounds and
// a third of 30 workgroups, 5.5 us of 41); in 33-row strips they are 33 x 31 = 1023.
template <int RW, int NW, bool VAR>
__device__ __forceinline__ void wave_rows(int sh, int wave, int &off, int &nr) {
    if (!VAR) {
        off = wave * RW;
        nr = RW;
        return;
    }
    const int extra = sh - NW * (RW - 1);
    off = wave * (RW - 1) + min(wave, extra);
    nr = RW - 1 + (wave < extra ? 1 : 0);
}

// ------------------------------------------------------------------ OpenCV-order window, overlapped strips: ONE launch per iteration
//
// The three row differences that straddle a strip boundary, d_t = (float)(M'[t+1] - M'[t-2]) of the NEW field, need rows of two
// workgroups (rounds 2 and 3 spent a second launch per iteration on them).  Here the strips overlap instead: a workgroup that owns the output rows [A, A + SO) of the new M computes the rows [A - 2, A + SO] -- three
// more, not stored -- so that every difference d_t with t in [A, A + SO) has both of its rows in this workgroup.  It leaves
//     T_s  = sum of d_t over t in [A_s, A_s + SO)          (ascending t: wavefront sums, then the wavefronts in order)
//     T'_s = the same without the last two (t < A_s + SO - 2)
// and the next launch's strip s starts its column chain one row above ITS first computed row A_s - 2 with
//     vsum(A_s - 3) = c0 + T_0 + ... + T_{s-2} + T'_{s-1},      c0 = (double)(3.f * M[0])
// summed in that order in the kernel's prologue (c0 + T_0 and c0 + T'_0 are what the top strip stores: it owns row 0; <= 15
// values per column and channel at 1080 rows with eight wavefronts per strip, one channel per wavefront, while the rows of
// the field are in flight).  No kernel reads what another workgroup of the same
// launch wrote; nothing but f64 additions is re-associated, as in the other strip-parallel forms.  Price: SO + 3 rows are
// computed for SO stored (4.3 % at 72-row strips), against one launch and ~18 MB of boundary rows per iteration saved.
struct HaloArgs {
    const double *Tin;    // [2][nstrips][5][pitch]  T (first half) and T' (second half) of the field the launch reads
    double *Tout;         // the same for the field it writes
    const float *Ein;     // [3][5][pitch]  edge rows of M: row 0, row max(h-3, 0), row h-1 (see halo_tile)
    float *Eout;
    int nstrips;
    int so;               // output rows per strip (the strip computes so + 3)
    size_t pair_vsum;     // batched calls: doubles between the T / edge arrays of consecutive pairs
    __device__ __forceinline__ void select_pair(int z) {
        if (Tin) Tin += (size_t)z * pair_vsum;
        Tout += (size_t)z * pair_vsum;
        if (Ein) Ein += (size_t)z * pair_vsum * 2;
        Eout += (size_t)z * pair_vsum * 2;
    }
};

// KIND: what the flow of a row comes from, and what leaves the kernel
//   kHaloLast    solve of box(M_in); the flow of the stored rows goes to `flows` (last iteration of a level), no M_out
//   kHaloIter    solve of box(M_in); M_out = UpdateMatrices(R0, R1, flow) + T / T' of M_out
//   kHaloZero / kHaloCoarse / kHaloGiven   the FIRST M of a level (+ its T / T'): zero flow (coarsest level), the coarser
//                level's flow prolongated (F6, `flows` = that flow), the caller's flow (`flows`, USE_INITIAL_FLOW at level 0)
// In the tall forms what travels between launches is not M but its ROW DIFFERENCES: row t of the field `Min` / `Mout` holds
//     d_t = (float)(M[min(t+1, h-1)] - M[max(t-2, 0)])          (the reference's srow1[x] - srow0[x])
// -- all an iteration ever uses of M besides vsum(-1).  The producer has every d_t of its strip anyway (it sums them for T);
// the consumer reads ITS OWN rows only (no three neighbour rows per wavefront to re-read or exchange) and starts its chain
// directly.  d_{h-1} = M[h-1] - M[max(h-3, 0)] has no row below it to be computed from three rows later, and vsum(-1) needs
// row 0 itself: those three rows of M travel in a small side array (`Ein` / `Eout`).
enum { kHaloLast = 0, kHaloIter = 1, kHaloZero = 2, kHaloCoarse = 3, kHaloGiven = 4 };

// DEEP: the gathers of ALL rows of the wavefront are in flight before the first row is finished (one memory latency per
// wavefront instead of one per row; ~200 registers) -- the form of the small levels, whose launches have at most two
// wavefronts per SIMD and are bound by their critical path, not by throughput
// LROWS (short wavefronts, RW < 5): every row of Mout goes through LDS and the row differences are taken from there after the
// rows are complete -- no constraint on the rows per wavefront.  A launch of a small level has less than one wavefront per
// SIMD and its duration is the instruction stream of ONE wavefront: two or three rows per wavefront instead of five.
template <int RW, int NW, bool LROWS>
struct HaloLds {
    double s_w[NW][5][64];        // wavefront sums: of Min's row differences first, of Mout's afterwards
    double s_kin[5][64];          // vsum of Min one row above the strip's first computed row
    double s_ip[5][64];           // the last wavefront's sum without the strip's last two differences
    float s_first[LROWS ? 1 : NW][3][5][64];   // the first three rows of Mout of every wavefront (for the wavefront above)
    float s_rows[LROWS ? NW * RW : 1][5][64];  // LROWS: all computed rows of Mout
};

// One workgroup's tile (tile column tbx, strip tby, pair tbz).
template <int KIND, int RW, int NW, bool VAR, bool DEEP, bool LROWS>
__device__ __forceinline__ void halo_tile(HaloLds<RW, NW, LROWS> &lds, const float *__restrict__ R0, const float *__restrict__ R1,
                                          const float *__restrict__ Min, float *__restrict__ Mout, const FlowTab &flows, const Prolong &pr,
                                          int w, int h, int pitch, double scale, HaloArgs ha, size_t pair_stride, int tbx, int tby, int tbz,
                                          const RgbaTab *rg = nullptr) {
    constexpr bool UPDATE = KIND != kHaloLast, SOLVE = KIND <= kHaloIter;
    // DF: the field between launches holds the row differences (tall forms).  The short-wavefront forms keep M itself: their
    // differences only exist after the workgroup's last barrier, and stores issued that late lengthen a launch whose duration
    // IS its critical path (measured +0.9 us on 10-12 us at 480x270 / 240x135); their wavefronts re-read the three neighbour rows.
    constexpr bool DF = !LROWS;
    // the last two differences of a strip must be differences inside the last wavefront (T' is its sum without them)
    static_assert(LROWS ? (!VAR && RW >= 2) : (VAR ? RW >= 6 : RW >= 5), "at least five rows per wavefront unless the rows go through LDS");
    auto &s_w = lds.s_w;
    auto &s_kin = lds.s_kin;
    auto &s_ip = lds.s_ip;
    auto &s_first = lds.s_first;
    auto &s_rows = lds.s_rows;
    R0 += (size_t)tbz * pair_stride;
    R1 += (size_t)tbz * pair_stride;
    if (SOLVE) Min += (size_t)tbz * pair_stride;
    if (UPDATE) Mout += (size_t)tbz * pair_stride;
    ha.select_pair(tbz);
    float *__restrict__ flow = flows.p[tbz];  // kHaloLast: out; kHaloCoarse / kHaloGiven: in; otherwise unused
    const size_t flow_step = flows.step[tbz];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x0 = tbx * kSsW;
    const int SC = VAR ? ha.so + 3 : RW * NW, SO = SC - 3;  // rows computed / stored per strip
    int off, nr;  // this wavefront's rows inside the computed strip (wave-uniform)
    wave_rows<RW, NW, VAR>(SC, wave, off, nr);
    const int A = tby * SO, a = A - 2 + off;  // a < 0 only for the first wavefront of the top strip (a = -2)
    const bool top = a < 0;
    const int xr = x0 - 1 + lane, x = clampi(xr, 0, w - 1);  // clamped = the replicated border columns of the reference
    const bool own = lane >= 1 && lane <= kSsW && xr < w;
    const size_t plane = (size_t)pitch * h;
    const unsigned pb = (unsigned)(plane * 4), rb = (unsigned)pitch * 4u, vx = 4u * (unsigned)x;
    const Buf bM = make_buf(Min, SOLVE ? 5 * plane * sizeof(float) : 0), bR0 = make_buf(R0, 5 * plane * sizeof(float)),
              bR1 = make_buf(R1, 5 * plane * sizeof(float)), bMo = make_buf(Mout, UPDATE ? 5 * plane * sizeof(float) : 0);
    const Buf bEi = make_buf(ha.Ein, SOLVE ? (size_t)15 * pitch * sizeof(float) : 0), bEo = make_buf(ha.Eout, UPDATE ? (size_t)15 * pitch * sizeof(float) : 0);
    const unsigned eb = (unsigned)pitch * 20u;  // bytes between the edge rows (5 channels each)
    auto valid = [&](int j) { return j < nr && a + j >= 0 && a + j < h; };                  // a row of the image (wave-uniform)
    auto stored = [&](int j) { return valid(j) && a + j >= A && a + j < A + SO; };         // ... that this strip owns

    float fxs[RW], fys[RW];
    if (SOLVE) {
        // prologue: the chain's start value from the strip sums the previous launch left (one channel per wavefront); issued
        // ahead of the rows of M (loads return in order: the sums are added up while the rows are still in flight)
        for (int c = wave; c < 5; c += NW) {
            // vsum(-1) = srow0 * (m + 2), a float product of row 0: the top strip reads it; for the others it is part of the top
            // strip's sums
            double k = 0.;
            if (tby == 0) {  // row 0 of M
                if (DF) k = (double)(buf_ld(bEi, vx, c * rb) * 3.f);
                else k = (double)(buf_ld(bM, vx, c * pb) * 3.f);
            } else {
                const size_t kst = (size_t)5 * pitch;
                const double *T = ha.Tin + (size_t)c * pitch + x;
                const int n = tby - 1;  // T of the strips 0 .. tby-2, then T' of strip tby-1
                constexpr int CH = 16;  // one batch of loads up to 17 strips
                const double tl = T[(size_t)(ha.nstrips + n) * kst];
                for (int s0 = 0; s0 < n; s0 += CH) {
                    double t[CH];
#pragma unroll
                    for (int i = 0; i < CH; i++) t[i] = s0 + i < n ? T[(size_t)(s0 + i) * kst] : 0.;
#pragma unroll
                    for (int i = 0; i < CH; i++)
                        if (s0 + i < n) k += t[i];
                }
                k += tl;
            }
            s_kin[c][lane] = k;
        }
        // this wavefront's rows of the difference field (the reference's srow1[x] - srow0[x]); the wavefront that holds the last
        // image row takes d_{h-1} from the edge rows
        float d[RW][5];
        if (!DF) {
            // rows a-2 .. a+nr of M (index r <-> image row clamp(a - 2 + r)): d_t = row[t+1] - row[t-2]
            float m[RW + 3][5];
#pragma unroll
            for (int r = 0; r < RW + 3; r++) {
                const unsigned so = (unsigned)clampi(a - 2 + r, 0, h - 1) * rb;
#pragma unroll
                for (int c = 0; c < 5; c++) m[r][c] = buf_ld(bM, vx, so + c * pb);
            }
#pragma unroll
            for (int j = 0; j < RW; j++)
#pragma unroll
                for (int c = 0; c < 5; c++) d[j][c] = m[j + 3][c] - m[j][c];
        }
#pragma unroll
        for (int j = 0; DF && j < RW; j++) {
            const int y = a + j;
            if (!valid(j)) {
#pragma unroll
                for (int c = 0; c < 5; c++) d[j][c] = 0.f;
            } else if (y == h - 1) {
#pragma unroll
                for (int c = 0; c < 5; c++)  // rows h-1 and max(h-3, 0) of M
                    d[j][c] = buf_ld(bEi, vx, 2 * eb + c * rb) - buf_ld(bEi, vx, eb + c * rb);
            } else {
#pragma unroll
                for (int c = 0; c < 5; c++) d[j][c] = buf_ld(bM, vx, (unsigned)y * rb + c * pb);
            }
        }
#pragma unroll
        for (int c = 0; c < 5; c++) {
            double t = 0.;
#pragma unroll
            for (int j = 0; j < RW; j++)
                if (valid(j)) t += (double)d[j][c];
            s_w[wave][c][lane] = t;
        }
        __syncthreads();
        double D[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            D[c] = s_kin[c][lane];
            for (int u = 0; u < wave; u++) D[c] += s_w[u][c][lane];  // vsum just above this wavefront's first row
        }
        __syncthreads();  // s_w is reused for the sums of Mout

        // all solves of the wavefront first: they only depend on the column sums (independent chains the SIMD can interleave)
#pragma unroll
        for (int j = 0; j < RW; j++) {
            if (!valid(j)) continue;  // wave-uniform
            double acc[5];
#pragma unroll
            for (int c = 0; c < 5; c++) D[c] += (double)d[j][c];  // the reference's vsum[x] += srow1[x] - srow0[x]
            if (!UPDATE && !stored(j)) continue;                   // the last iteration of a level has no use for the halo rows
#pragma unroll
            for (int c = 0; c < 5; c++) acc[c] = (dpp64_from_left(D[c]) + D[c]) + dpp64_from_right(D[c]);
            double g11_ = acc[0] * scale, g12_ = acc[1] * scale, g22_ = acc[2] * scale, h1_ = acc[3] * scale, h2_ = acc[4] * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            fxs[j] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            fys[j] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
            if (!UPDATE && flow && own) *(float2 *)((char *)flow + (size_t)(a + j) * flow_step + (size_t)xr * 8) = make_float2(fxs[j], fys[j]);
            if (!UPDATE && rg && rg->p[tbz] && own) {  // F7 (VectorGenerator.cpp:494-519) on the flow still in registers
                const float u = (float)(fxs[j] / rg->rsx), v = (float)(fys[j] / rg->rsy);
                const unsigned mu = rg->mu[tbz], mv = rg->mv[tbz];
                float *d = (float *)((char *)rg->p[tbz] + (ptrdiff_t)(a + j) * rg->step[tbz]) + (size_t)xr * 4;
                if (((mu | mv) & 15u) == 15u && (((uintptr_t)d) & 15) == 0) {
                    *(float4 *)d = make_float4((mv & 1u) ? v : u, (mv & 2u) ? v : u, (mv & 4u) ? v : u, (mv & 8u) ? v : u);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c += 2) {  // channel pairs: one 8-byte store where both are mapped
                        const unsigned m2 = ((mu | mv) >> c) & 3u;
                        if (m2 == 3u && (((uintptr_t)d) & 7) == 0) *(float2 *)(d + c) = make_float2((mv >> c) & 1u ? v : u, (mv >> (c + 1)) & 1u ? v : u);
                        else {
                            if (m2 & 1u) d[c] = (mv >> c) & 1u ? v : u;
                            if (m2 & 2u) d[c + 1] = (mv >> (c + 1)) & 1u ? v : u;
                        }
                    }
                }
            }
        }
        if (!UPDATE) return;
    } else {
#pragma unroll
        for (int j = 0; j < RW; j++) {
            fxs[j] = fys[j] = 0.f;
            if (KIND == kHaloZero || !valid(j)) continue;
            if (KIND == kHaloCoarse) {
                prolong_flow(flow, flow_step, pr, x, a + j, fxs[j], fys[j]);
            } else {
                const float2 f = *(const float2 *)((const char *)flow + (size_t)(a + j) * flow_step + (size_t)x * 8);
                fxs[j] = f.x;
                fys[j] = f.y;
            }
        }
    }

    struct Px {
        TapsQ tp;
        float r0v[5];
        float fxv, fyv;
    };
    Px prev;
    auto st_e = [&](float v, unsigned soff) {  // an edge row of M_out (own lanes: `st`)
        buf_st(bEo, v, vx, soff);
    };
    auto st_d = [&](float dv, int t, int c) {  // row t of the difference field of M_out
        if (!DF || !own) return;
        buf_st(bMo, dv, vx, (unsigned)t * rb + c * pb);
    };
    float mo[RW][5];   // rows of Mout as they are produced (only the last three finished ones stay live)
    double I[5] = {0., 0., 0., 0., 0.};   // row differences of Mout with both rows in this wavefront, ascending t
    double Ip[5] = {0., 0., 0., 0., 0.};  // last wavefront: I before the strip's last two differences
    auto finish = [&](const Px &p, int j) {
        const int y = a + j;
        M5 mm = update_matrices_finish(p.r0v, p.tp, x, y, w, h, p.fxv, p.fyv);
        const bool st = own && y >= A && y < A + SO;  // this strip owns the row: its edge rows
#pragma unroll
        for (int c = 0; c < 5; c++) {
            if (!DF && st) {
                buf_st(bMo, mm.v[c], vx, (unsigned)y * rb + c * pb);
            }
            if (DF && st && (y == 0 || y == h - 1 || y == max(h - 3, 0))) {
                if (y == 0) st_e(mm.v[c], c * rb);
                if (y == max(h - 3, 0)) st_e(mm.v[c], eb + c * rb);
                if (y == h - 1) st_e(mm.v[c], 2 * eb + c * rb);
            }
            if (LROWS) {
                s_rows[off + j][c][lane] = mm.v[c];
                continue;
            }
            mo[j][c] = mm.v[c];
            if (j < 3) s_first[wave][j][c][lane] = mm.v[c];
            // rows above row 0 are row 0 (t = 0, 1 are row 1 - row 0, row 2 - row 0): the top wavefront's rows -2, -1
            if (j == 2 && top) {
                mo[0][c] = mo[1][c] = mm.v[c];
                s_kin[c][lane] = (double)(mm.v[c] * 3.f);  // vsum(-1) of the NEW field: goes into the top strip's sums
            }
            if (j >= 3) {   // t = y-1: rows y, y-3, both in this wavefront
                if (wave == NW - 1 && j == nr - 2) Ip[c] = I[c];
                const float dv = mm.v[c] - mo[j - 3][c];
                I[c] += (double)dv;
                st_d(dv, y - 1, c);
            }
        }
    };
    // the valid rows of a wavefront are consecutive (rows -2, -1 of the top wavefront, rows below the image and the missing
    // row of a short wavefront lie at its ends); all conditions are wave-uniform
    if (DEEP) {
        Px all[RW];
#pragma unroll
        for (int j = 0; j < RW; j++) {
            if (!valid(j)) continue;
            all[j].fxv = fxs[j];
            all[j].fyv = fys[j];
#pragma unroll
            for (int c = 0; c < 5; c++) all[j].r0v[c] = buf_ld(bR0, vx, (unsigned)(a + j) * rb + c * pb);
            all[j].tp = gather_taps_q(bR1, x, a + j, w, h, pitch, pb, fxs[j], fys[j]);
        }
#pragma unroll
        for (int j = 0; j < RW; j++)
            if (valid(j)) finish(all[j], j);
    } else {
#pragma unroll
        for (int j = 0; j < RW; j++) {
            if (!valid(j)) continue;
            const int y = a + j;
            Px cur;
            cur.fxv = fxs[j];
            cur.fyv = fys[j];
#pragma unroll
            for (int c = 0; c < 5; c++) cur.r0v[c] = buf_ld(bR0, vx, (unsigned)y * rb + c * pb);
            cur.tp = gather_taps_q(bR1, x, y, w, h, pitch, pb, cur.fxv, cur.fyv);
            if (j > 0 && valid(j - 1)) finish(prev, j - 1);  // the gather of row j is in flight while the row before it is finished
            prev = cur;
        }
#pragma unroll
        for (int j = 0; j < RW; j++)
            if (valid(j) && !(j + 1 < RW && valid(j + 1))) finish(prev, j);
    }
    __syncthreads();  // every wavefront's first three rows (LROWS: all rows) are in LDS
    if (LROWS) {
        // this wavefront's differences (later row = one of its rows, strip row q = off + j >= 3; in the top strip rows above
        // row 0 are row 0 = strip row 2), ascending; the strip's last two are the last two of the last wavefront
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int q = off + j;
            if (q < 3 || !valid(j)) continue;
            const int qe = tby == 0 ? max(q - 3, 2) : q - 3;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                if (wave == NW - 1 && j == RW - 2) Ip[c] = I[c];
                const float dv = s_rows[q][c][lane] - s_rows[qe][c][lane];
                I[c] += (double)dv;
                st_d(dv, a + j - 1, c);
            }
        }
    }
    // the three differences across the boundary to the wavefront below (t = b-1, b, b+1 with b its first row): its rows
    // 0..2 against this wavefront's last three
#pragma unroll
    for (int c = 0; c < 5; c++) {
        double sum = I[c];
        if (LROWS) {
            if (wave == NW - 1) s_ip[c][lane] = Ip[c];
        } else if (wave < NW - 1) {
            const bool full = !VAR || nr == RW;  // a short wavefront's last three rows are one index earlier
            const float l0 = full ? mo[RW - 3][c] : mo[RW - 4][c], l1 = full ? mo[RW - 2][c] : mo[RW - 3][c],
                        l2 = full ? mo[RW - 1][c] : mo[RW - 2][c];
            const float dv0 = s_first[wave + 1][0][c][lane] - l0, dv1 = s_first[wave + 1][1][c][lane] - l1, dv2 = s_first[wave + 1][2][c][lane] - l2;
            sum += (double)dv0;
            sum += (double)dv1;
            sum += (double)dv2;
            const int yb = a + nr;  // first row of the wavefront below: the differences t = yb-1, yb, yb+1 (where its rows exist)
            if (yb < h) st_d(dv0, yb - 1, c);
            if (yb + 1 < h) st_d(dv1, yb, c);
            if (yb + 2 < h) st_d(dv2, yb + 1, c);
        } else {
            s_ip[c][lane] = Ip[c];
        }
        s_w[wave][c][lane] = sum;
    }
    __syncthreads();
    if (wave < 2 && own) {  // wavefront 0 writes T, wavefront 1 T'
        const size_t tq = (size_t)ha.nstrips * 5 * pitch;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            double sum = s_w[0][c][lane];
            for (int u = 1; u < NW - 1; u++) sum += s_w[u][c][lane];
            sum += wave == 0 ? s_w[NW - 1][c][lane] : s_ip[c][lane];
            if (tby == 0) sum = (LROWS ? (double)(s_rows[2][c][lane] * 3.f) : s_kin[c][lane]) + sum;  // the top strip's sums carry vsum(-1)
            double *o = ha.Tout + (wave ? tq : 0) + ((size_t)tby * 5 + c) * pitch + xr;
            *o = sum;
        }
    }
}


template <int KIND, int RW, int NW, bool VAR, bool DEEP = false, bool LROWS = (RW < 5)>
__global__ __launch_bounds__(64 * NW, DEEP && !LROWS ? 2 : 4) void iterate3h_kernel(const float *__restrict__ R0, const float *__restrict__ R1,
                                                               const float *__restrict__ Min, float *__restrict__ Mout,
                                                               FlowTab flows, Prolong pr, int w, int h, int pitch, double scale,
                                                               HaloArgs ha, size_t pair_stride, RgbaTab rg) {
    __shared__ HaloLds<RW, NW, LROWS> lds;
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    halo_tile<KIND, RW, NW, VAR, DEEP, LROWS>(lds, R0, R1, Min, Mout, flows, pr, w, h, pitch, scale, ha, pair_stride, tbx, tby, tbz,
                                                     KIND == kHaloLast ? &rg : nullptr);
}

// ------------------------------------------------------------------ OpenCV-order window: column-owning workgroups, TWO steps per launch
//
// The column prefix of iteration k+1 needs every row of M_{k+1} above it, so two iterations cannot be fused inside a strip.
// A workgroup that owns a tile column over the FULL height can: it walks the column top to bottom in rounds of S rows
// (NW wavefronts of RW-1 or RW rows each), runs step 1 (the solve of iteration k + the new matrices M') on the rows
// [A, A+S) of a round and step 2 (the solve of iteration k+1 from the row differences of M' + the matrices M'') one row
// behind it, on [A-1, A+S-1): d'_t = M'[t+1] - M'[t-2] needs the row below.  M' never leaves the registers (three boundary
// rows per wavefront through LDS); the running column sums of both steps are handed from wavefront to wavefront (and from round
// to round) through one LDS slot each, so the launch needs no strip sums at all and the next launch starts its chain at
// vsum(-1) = 3 * row 0 again.  Per two iterations a pixel costs M-in 20 + 2 x (R0 20 + R1 gather 20) + M-out 20 = 120 B instead of 160, and
// the second reads of R0 / R1 hit the L2.  64 lanes -> 62 valid columns after step 1 -> 60 after step 2 (6.7 % redundancy).
//
// Steps of a level: first M (zero / prolongated / given flow), iterations - 1 x iterate, last (flow + F7 out) -- paired up
// (first, iterate) (iterate, iterate) ... (iterate, last); an odd count ends with (last, -).  The field between launches is the
// difference field of the overlapped-strip form (rows t = 0 .. h-2 of d, the edge rows 0, h-3, h-1 of M beside it), so both
// forms can follow each other inside a level.
//
// No barrier after the prologue: every hand-off is point to point (LDS data + a monotonic LDS counter, release / acquire at
// workgroup scope on the LDS address space only -- loads and stores to memory stay in flight across it):
//   p[s]    the running f64 column sum of step s: a token chain over (round, wavefront); wavefront u of round r adds the sum
//           of its own row differences and passes it on (seq[s] = tickets served)
//   b[s][u] the last three rows of M' / M'' of wavefront u, for the differences across the boundary to the wavefront below
//           (wavefront 0 takes those of the last wavefront of the round before); wr / rd count writes and reads of a slot
// A wavefront only ever waits for wavefronts of its own workgroup (all resident) along an acyclic order (smaller ticket, or
// the reader of its own slot one round earlier), so the waits terminate; they are bounded all the same (`spin`), and a wait
// that runs out raises the sticky `abort` word (ofxcv_ctx_get_option "farneback.col_aborts").
constexpr int kColW = 60;   // columns a workgroup stores (lanes 2..61)
constexpr size_t kColFlagBytes = 256 + 64 * 16 * 16 * 8;  // abort word + trace area (64 rounds x 16 wavefronts x 16 stamps)
enum { kColNone = -1 };

struct ColArgs {
    const float *Ein;   // [3][5][pitch] edge rows (0, max(h-3, 0), h-1) of the M the launch reads as differences
    float *Eout;        // the same for the M it writes
    size_t pair_vsum;   // doubles between the scratch of consecutive pairs (the edge rows are floats inside it)
    int S, rounds;      // step-1 rows per round (NW * (RW-1) .. NW * RW), rounds (S * rounds >= h + 1)
    unsigned *abort;
    unsigned spin;
    unsigned long long *trace;  // [rounds][NW][16] shader-clock stamps of one workgroup (TRACE instantiation), or null
    __device__ __forceinline__ void select_pair(int z) {
        if (Ein) Ein += (size_t)z * pair_vsum * 2;
        if (Eout) Eout += (size_t)z * pair_vsum * 2;
    }
};

template <int NW>
struct ColLds {
    // the boundary rows of the two steps: a buffer each
    static constexpr int NB = 2;
    float b[NB][NW][3][5][64];
    // the token of step s: the running column sums of five channels per lane as {P0, P1} {P2, P3} {P4} and, written LAST and read FIRST, the ticket
    // they are for.  LDS executes a wavefront's accesses in issue order, so a reader that finds the tag finds the sums behind it: no fence, no
    // separate flag, one LDS round trip per link.
    struct alignas(16) Token {
        double a[64][2], b[64][2], c[64];
        int tag[64];
    } tok[2];
    int wr[NB][NW], rd[NB][NW];
};

__device__ __forceinline__ int lds_flag_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// wait until *p >= target (every lane polls the same word: a broadcast read); then LDS reads may follow
__device__ __forceinline__ void lds_wait(const int *p, int target, const ColArgs &ca) {
    if (lds_flag_ld(p) < target) {
        unsigned n = 0;
        do {
            __builtin_amdgcn_s_sleep(1);
            if (++n > ca.spin) {
                __hip_atomic_store(ca.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // pinned host memory (ctx->fb_col_abort)
                break;
            }
        } while (lds_flag_ld(p) < target);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// LDS data written before, then the counter (one lane)
__device__ __forceinline__ void lds_post(int *p, int v, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) lds_flag_st(p, v);
}

// F7 (VectorGenerator.cpp:494-519) on a flow still in registers
__device__ __forceinline__ void f7_store(const RgbaTab &rg, int z, int xr, int y, float fx, float fy) {
    const float u = (float)(fx / rg.rsx), v = (float)(fy / rg.rsy);
    const unsigned mu = rg.mu[z], mv = rg.mv[z];
    float *d = (float *)((char *)rg.p[z] + (ptrdiff_t)y * rg.step[z]) + (size_t)xr * 4;
    if (((mu | mv) & 15u) == 15u && (((uintptr_t)d) & 15) == 0) {
        *(float4 *)d = make_float4((mv & 1u) ? v : u, (mv & 2u) ? v : u, (mv & 4u) ? v : u, (mv & 8u) ? v : u);
    } else {
#pragma unroll
        for (int c = 0; c < 4; c += 2) {  // channel pairs: one 8-byte store where both are mapped
            const unsigned m2 = ((mu | mv) >> c) & 3u;
            if (m2 == 3u && (((uintptr_t)d) & 7) == 0) *(float2 *)(d + c) = make_float2((mv >> c) & 1u ? v : u, (mv >> (c + 1)) & 1u ? v : u);
            else {
                if (m2 & 1u) d[c] = (mv >> c) & 1u ? v : u;
                if (m2 & 2u) d[c + 1] = (mv >> (c + 1)) & 1u ? v : u;
            }
        }
    }
}

// R1 WINDOW IN LDS (RING; round 5).  Counters (profiles/r05_pmc_attr_iterate_col_baseline.txt): the texture addresser is the busiest unit of this
// kernel -- TA_BUSY 74-83 % of the launch, ~37 of its cycles per gather -- while the vector ALU is 45 % busy and the LDS pipe 3 %.  So the
// workgroup keeps the R1 rows its two steps can reach in an LDS ring and gathers from there:
//   * ring: kRingRows = 64 image rows x kRingCols = 64 + 2 D columns (the tile column's lanes +- D), packed like the field itself (float4 of four
//     channels per pixel + plane 4): 90 KB beside the 65 KB of hand-off rows -- one workgroup per CU either way;
//   * fill: LDS-DMA (buffer_load ... lds: no registers, no ds_write), in groups of four rows.  The wavefront with ticket t (rows 4t .. 4t+3) issues
//     the fill of group t + L (rows 4(t+L) ..) right after it has taken the step-1 token, L = 6 groups = three quarters of a round ahead of the first
//     wavefront that needs it (ticket t + 5); it publishes `filled[wave] = round + 1` once its loads have landed (s_waitcnt vmcnt(0) at the end
//     of its step 1, where nothing else is in flight);
//   * why a 64-row ring is enough and never overwritten too early: the eight active tickets span at most 8 x 4 rows, a ticket reads rows
//     [4t - 1 - D, 4t + 3 + D], the newest group in flight is t_fastest + L: 33 + 4 L + D = 61 rows.  Group g overwrites group g - 16, last read by
//     ticket g - 14; the filler (ticket g - 6) holds the step-1 token, which it can only have got after ticket g - 7 -- the next round of ticket g - 15's
//     wavefront -- started, and it is itself the next round of ticket g - 14: every reader of the old rows is done;
//   * a gather whose 64 lanes all sample within +- D of their own pixel (wave-uniform test, one ballot) reads the ring (four ds_read_b128 + two
//     ds_read2_b32 per pixel); otherwise the whole wavefront-row takes the global gather as before -- same values either way.
constexpr int kRingD = 4, kRingRows = 64, kRingCols = 64 + 2 * kRingD, kRingLead = 6;
struct ColRing {
    ofxcv_f4 q[kRingRows * kRingCols];
    float c[kRingRows * kRingCols];
    int filled[16];
};
typedef unsigned ofxcv_u4 __attribute__((ext_vector_type(4)));
// The fill's loads are LDS-DMA (buffer_load ... offen lds: 64 lanes x 4 or 16 bytes from a buffer into LDS at M0 + lane * size), issued from inline
// assembly: M0 is compiler-reserved, so it is saved and restored inside the statement, and the loads are invisible to the compiler's wait-count
// bookkeeping on purpose (a load it tracked would make it wait for the fill in front of every ring read): completion is the filler's own
// `s_waitcnt vmcnt(0)` before it publishes the group.
template <int K1, int K2, int RW, int NW, bool RING = false, bool TRACE = false>
__global__ __launch_bounds__(64 * NW) void iterate_col_kernel(const float *__restrict__ R0, const float *__restrict__ R1,
                                                              const float *__restrict__ Din, float *__restrict__ Dout, FlowTab fin, FlowTab fout, Prolong pr,
                                                              int w, int h, int pitch, double scale, ColArgs ca, size_t pair_stride, RgbaTab rg) {
    constexpr bool SOLVE1 = K1 <= kHaloIter, LAST1 = K1 == kHaloLast, TWO = K2 != kColNone, LAST2 = K2 == kHaloLast;
    constexpr bool OUT = !LAST1 && !LAST2;  // the launch leaves a field
    static_assert(K2 == kColNone || K2 == kHaloIter || K2 == kHaloLast, "step 2 iterates or ends the level");
    static_assert(LAST1 != TWO, "nothing follows the last step; every other step has a partner");
    static_assert(RW >= 3, "the three boundary rows");
    constexpr int DEPTH = 1;  // rows whose samples are in flight before the first is consumed (2 and 4 measured the same: r05_experiments.md)
    __shared__ ColLds<NW> lds;
    static_assert(!RING || (K1 == kHaloIter && RW == 4 && NW == 8), "the ring's fill schedule rides on the step-1 token of eight wavefronts of four rows");
    // Fill groups are four image rows: ticket t fills group t + 6.  (A second geometry, twelve wavefronts of three rows with the boundary rows of both
    // steps in one LDS buffer, ran 2.5 % faster in round 5 and was never the default; removed in round 6 -- profiles/r05_experiments.md 14 has it.)
    constexpr int kLead = kRingLead;
    __shared__ typename std::conditional<RING, ColRing, int>::type ring;
    int tbx, tby, tbz;
    xcd_tile(tbx, tby, tbz);
    R0 += (size_t)tbz * pair_stride;
    R1 += (size_t)tbz * pair_stride;
    if (SOLVE1) Din += (size_t)tbz * pair_stride;
    if (OUT) Dout += (size_t)tbz * pair_stride;
    ca.select_pair(tbz);
    const float *__restrict__ flow = fin.p[tbz];   // coarse / given: the flow the level starts from
    const size_t flow_step = fin.step[tbz];
    float *__restrict__ oflow = fout.p[tbz];       // last: the level's flow
    const size_t oflow_step = fout.step[tbz];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < 128) lds.tok[threadIdx.x >> 6].tag[threadIdx.x & 63] = 0;
    if (threadIdx.x < ColLds<NW>::NB * NW) {
        (&lds.wr[0][0])[threadIdx.x] = 0;
        (&lds.rd[0][0])[threadIdx.x] = 0;
    }
    if constexpr (RING) {
        if (threadIdx.x < 16) ring.filled[threadIdx.x] = 0;
    }
    const int x0 = tbx * kColW;
    const int xr = x0 - 2 + lane, x = clampi(xr, 0, w - 1);  // clamped = the replicated border columns of the reference
    const bool own = lane >= 2 && lane < 2 + kColW && xr < w;
    const size_t plane = (size_t)pitch * h;
    const unsigned pb = (unsigned)(plane * 4), rb = (unsigned)pitch * 4u, vx = 4u * (unsigned)x;
    const Buf bD = make_buf(Din, SOLVE1 ? 5 * plane * sizeof(float) : 0), bR0 = make_buf(R0, 5 * plane * sizeof(float)),
              bR1 = make_buf(R1, 5 * plane * sizeof(float)), bDo = make_buf(Dout, OUT ? 5 * plane * sizeof(float) : 0);
    const Buf bEi = make_buf(ca.Ein, SOLVE1 ? (size_t)5 * pitch * sizeof(float) : 0), bEo = make_buf(ca.Eout, OUT ? (size_t)5 * pitch * sizeof(float) : 0);
    // lanes beyond the image edge repeat the border column: after step 1 they must hold the BORDER pixel's flow (their own box
    // window is not the border pixel's), so that their M' is the replicated border column step 2 sums over
    const int lane_r = __builtin_amdgcn_readfirstlane(min(w + 1 - x0, 63));
    // ---- the R1 ring (RING): this lane's part of a fill group and the group fill itself
    const int xw0 = x0 - 2 - kRingD;  // image column of ring column 0
    [[maybe_unused]] unsigned ring_q_addr = 0, ring_c_addr = 0;
    [[maybe_unused]] int frow[5];
    [[maybe_unused]] unsigned fvo[5];  // this lane's element of each of a group's five loads: (row in the group) * pitch + image column (clamped to the field's rows)
    [[maybe_unused]] ofxcv_u4 r1rsrc = {0, 0, 0, 0};
    if constexpr (RING) {
        ring_q_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)ring.q);
        ring_c_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)ring.c);
        const unsigned long long ba = (unsigned long long)(size_t)R1;
        r1rsrc = ofxcv_u4{(unsigned)ba, (unsigned)(ba >> 32) & 0xffffu, (unsigned)(5 * plane * sizeof(float)), 0x00020000u};
#pragma unroll
        for (int i = 0; i < 5; i++) {  // element e = i * 64 + lane of a group of 4 rows x kRingCols columns (the fifth load: 32 lanes)
            const int e = i * 64 + lane;
            frow[i] = e / kRingCols;
            fvo[i] = (unsigned)(frow[i] * pitch + clampi(xw0 + e - frow[i] * kRingCols, 0, pitch - 1));
        }
    }
    // group g = image rows 4g .. 4g+3 -> ring rows (4g .. 4g+3) & 63, ten LDS-DMA loads (five of 16 bytes per lane, five of 4) in ONE statement: M0 is
    // saved once, stepped from load to load and restored.  Columns outside the field's rows are clamped (never sampled: such a tap is out of
    // bounds); rows below the image repeat row h-1; a group entirely below the image is not filled at all (the late tickets' clamped rows still read
    // the rows just above the image's last row, which such a fill would overwrite).
    auto ring_fill = [&](int g) __attribute__((always_inline)) {
        if constexpr (RING) {
            if (4 * g >= h) return;
            const unsigned qa = ring_q_addr + (unsigned)((4 * g) & (kRingRows - 1)) * (kRingCols * 16u);
            const unsigned ca4 = ring_c_addr + (unsigned)((4 * g) & (kRingRows - 1)) * (kRingCols * 4u);
            unsigned vo[5];
            unsigned sq, sc;
            if (4 * g + 3 < h) {  // (wave-uniform) every row of the group inside the image: the lane's constant part + the group's rows as scalar offsets
#pragma unroll
                for (int i = 0; i < 5; i++) vo[i] = fvo[i];
                sq = (unsigned)(4 * g) * (unsigned)pitch * 16u;
                sc = (unsigned)(4 * g) * (unsigned)pitch * 4u + 4u * pb;
            } else {
#pragma unroll
                for (int i = 0; i < 5; i++) vo[i] = fvo[i] - (unsigned)(frow[i] * pitch) + (unsigned)(min(4 * g + frow[i], h - 1) * pitch);
                sq = 0u;
                sc = 4u * pb;
            }
            unsigned keep;
            asm volatile(
                "s_mov_b32 %[k], m0\n\t"
                "s_mov_b32 m0, %[qa]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[q0], %[rs], %[sq] offen lds\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[q1], %[rs], %[sq] offen lds\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[q2], %[rs], %[sq] offen lds\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[q3], %[rs], %[sq] offen lds\n\t"
                "s_mov_b32 m0, %[ca]\n\ts_nop 0\n\tbuffer_load_dword %[c0], %[rs], %[sc] offen lds\n\t"
                "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tbuffer_load_dword %[c1], %[rs], %[sc] offen lds\n\t"
                "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tbuffer_load_dword %[c2], %[rs], %[sc] offen lds\n\t"
                "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tbuffer_load_dword %[c3], %[rs], %[sc] offen lds\n\t"
                "s_mov_b32 m0, %[k]"
                : [k] "=&s"(keep)
                : [qa] "s"(qa), [ca] "s"(ca4), [rs] "s"(r1rsrc), [sq] "s"(sq), [sc] "s"(sc), [q0] "v"(vo[0] * 16u), [q1] "v"(vo[1] * 16u), [q2] "v"(vo[2] * 16u),
                  [q3] "v"(vo[3] * 16u), [c0] "v"(vo[0] * 4u), [c1] "v"(vo[1] * 4u), [c2] "v"(vo[2] * 4u), [c3] "v"(vo[3] * 4u)
                : "memory", "scc");
            if (lane < 4 * kRingCols - 256) {  // the last 32 elements of the group
                asm volatile(
                    "s_mov_b32 %[k], m0\n\t"
                    "s_mov_b32 m0, %[qa]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[q4], %[rs], %[sq] offen lds\n\t"
                    "s_mov_b32 m0, %[ca]\n\ts_nop 0\n\tbuffer_load_dword %[c4], %[rs], %[sc] offen lds\n\t"
                    "s_mov_b32 m0, %[k]"
                    : [k] "=&s"(keep)
                    : [qa] "s"(qa + 4096u), [ca] "s"(ca4 + 1024u), [rs] "s"(r1rsrc), [sq] "s"(sq), [sc] "s"(sc), [q4] "v"(vo[4] * 16u), [c4] "v"(vo[4] * 4u)
                    : "memory");
            }
        }
    };
    if constexpr (RING) {
        for (int g0 = wave; g0 < kLead; g0 += NW) ring_fill(g0);  // groups 0 .. L-1: what the first tickets need before any of them has filled anything
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int off = wave * RW;
    const int pw = wave == 0 ? NW - 1 : wave - 1;  // whose boundary rows this wavefront takes
    // TRACE (option farneback.col_trace): one workgroup writes the shader clock at the phase boundaries of every round
    const bool tracing = TRACE && ca.trace && tbx == 3 && tbz == 0;
    auto stamp = [&](int r, int k) __attribute__((always_inline)) {
        if (TRACE && tracing && lane == 0) ca.trace[(size_t)(r * NW + wave) * 16 + k] = __builtin_amdgcn_s_memtime();
    };

    struct Px {
        TapsQ tp;
        float r0v[5];
    };
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto gather = [&](int gx, int gy, float dx, float dy) __attribute__((always_inline)) {
        if constexpr (!RING) {
            return gather_taps_q(bR1, gx, gy, w, h, pitch, pb, dx, dy);
        } else {
            TapsQ tp;
            const float fx = gx + dx, fy = gy + dy;
            const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
            tp.fx = fx - x1;
            tp.fy = fy - y1;
            tp.inb = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
            // every lane's sample within +- D of its own pixel: the ring holds the footprint (wave-uniform decision)
            const bool inw = (unsigned)(x1 - gx + kRingD) < 2u * kRingD && (unsigned)(y1 - gy + kRingD) < 2u * kRingD;
            if (__builtin_amdgcn_ballot_w64(!inw) == 0) {
                const int cx = x1 - xw0;
                const int e0 = (y1 & (kRingRows - 1)) * kRingCols + cx, e1 = ((y1 + 1) & (kRingRows - 1)) * kRingCols + cx;
                tp.t0 = ring.q[e0];
                tp.t1 = ring.q[e0 + 1];
                tp.b0 = ring.q[e1];
                tp.b1 = ring.q[e1 + 1];
                tp.t4.a = ring.c[e0];
                tp.t4.b = ring.c[e0 + 1];
                tp.b4.a = ring.c[e1];
                tp.b4.b = ring.c[e1 + 1];
            } else {
                const unsigned o = tp.inb ? (unsigned)y1 * (unsigned)pitch + (unsigned)x1 : 0u;
                const unsigned oq = o * 16u, rq = (unsigned)pitch * 16u, o4 = o * 4u, r4 = (unsigned)pitch * 4u;
                tp.t0 = buf_ld4(bR1, oq, 0);
                tp.t1 = buf_ld4(bR1, oq + 16u, 0);
                tp.b0 = buf_ld4(bR1, oq + rq, 0);
                tp.b1 = buf_ld4(bR1, oq + rq + 16u, 0);
                tp.t4.a = buf_ld(bR1, o4, 4 * pb);
                tp.t4.b = buf_ld(bR1, o4 + 4u, 4 * pb);
                tp.b4.a = buf_ld(bR1, o4 + r4, 4 * pb);
                tp.b4.b = buf_ld(bR1, o4 + r4 + 4u, 4 * pb);
            }
            return tp;
        }
    };
    // RING: before a ticket's first gather, the groups its rows can reach (<= (4t + 3 + D) / 4) must have landed: the fills of the tickets up to
    // T = that group - L.  Wavefront j has then published at least (T - j) / 8 + 1 fills: lanes 0 .. 7 each check one wavefront's counter.
    auto ring_wait = [&](int ticket) __attribute__((always_inline)) {
        if constexpr (RING) {
            // the last group this ticket's rows can reach, and the ticket that fills it
            const int T = (RW * ticket + RW - 1 + kRingD) / 4 - kLead;
            const int l = fresh_lane();
            const int need = (l < NW && T >= l) ? (T - l) / NW + 1 : 0;
            unsigned n = 0;
            while (__builtin_amdgcn_ballot_w64(lds_flag_ld(&ring.filled[l & 15]) < need) != 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++n > ca.spin) {
                    __hip_atomic_store(ca.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    };
    auto solve = [&](const double (&D)[5], float &fx, float &fy) __attribute__((always_inline)) {
        double acc[5];
#pragma unroll
        for (int c = 0; c < 5; c++) acc[c] = (dpp64_from_left(D[c]) + D[c]) + dpp64_from_right(D[c]);
        const double g11_ = acc[0] * scale, g12_ = acc[1] * scale, g22_ = acc[2] * scale, h1_ = acc[3] * scale, h2_ = acc[4] * scale;
        const double det = g11_ * g22_ - g12_ * g12_ + 1e-3;
        // 1 / det as the compiler's own correctly rounded sequence WITHOUT its range scaling (v_div_scale x 2, the multiplication by the
        // scaled numerator 1.0, v_div_fmas): those only act on operands near the ends of the f64 exponent range, and det is a sum of
        // products of 8-bit-image moments plus 1e-3 -- the same bits for every normal det with |det| in [2^-700, 2^700]; zero, infinity
        // and NaN go through v_div_fixup as before.  8 instead of 12 instructions per solve.
        double y0 = __builtin_amdgcn_rcp(det);
        double e = __builtin_fma(-det, y0, 1.0);
        y0 = __builtin_fma(y0, e, y0);
        e = __builtin_fma(-det, y0, 1.0);
        y0 = __builtin_fma(y0, e, y0);
        e = __builtin_fma(-det, y0, 1.0);
        const double idet = __builtin_amdgcn_div_fixup(__builtin_fma(e, y0, y0), det, 1.0);
        fx = (float)((g11_ * h2_ - g12_ * h1_) * idet);
        fy = (float)((g22_ * h1_ - g12_ * h2_) * idet);
    };
    // F4 of one row from its samples: the border scale as one wave-uniform condition -- the lane's factor of the two vertical image edges is
    // hoisted, a row's factors are scalars, and scale 1 is applied as a multiplication (exact) to the lanes of a border row / border workgroup
    // that are not themselves within five pixels of an edge; rows and workgroups away from the edges skip it.  The reference's test
    // `(unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10)` is kept to the letter: below ten columns its
    // first half wraps and only holds at x == 4, so a lane's column factors count in a border ROW always, elsewhere only where that half holds.
    const bool cx_in = (unsigned)(x - kUmBorder) >= (unsigned)(w - 2 * kUmBorder);
    const float sxc = um_border(x) * um_border(w - x - 1), sx_only = cx_in ? sxc : 1.f;
    const bool wg_edge_x = x0 - 2 < kUmBorder || x0 + 61 >= w - kUmBorder || w < 2 * kUmBorder;  // wave-uniform: some lane's cx_in may hold
    auto finish = [&](const auto &qq, int y, float dx, float dy) __attribute__((always_inline)) {
        float rr[5];
        um_sample(qq.r0v, qq.tp, dx, dy, rr);
        const bool cy = (unsigned)(y - kUmBorder) >= (unsigned)(h - 2 * kUmBorder);  // wave-uniform
        if (wg_edge_x || cy) {
            const float sc = (cy ? sxc : sx_only) * um_border(y) * um_border(h - y - 1);
#pragma unroll
            for (int c = 0; c < 5; c++) rr[c] *= sc;
        }
        return um_products(rr);
    };
    auto flow_out = [&](int y, float fx, float fy) __attribute__((always_inline)) {
        if (!own || y < 0 || y >= h) return;
        if (oflow) *(float2 *)((char *)oflow + (size_t)y * oflow_step + (size_t)xr * 8) = make_float2(fx, fy);
        if (rg.p[tbz]) f7_store(rg, tbz, xr, y, fx, fy);
    };
    // hand the running column sum of step s on: P = the sum just above this wavefront's first row of the step
    // the lane index, recomputed where a hand-off needs it: an LDS address kept in a register across a round is what the
    // register allocator spills first, and a reload from scratch inside the token's critical section costs every wavefront
    // behind this one a memory round trip (measured: two reloads = 5 000 cycles per link, the whole launch chain-bound)
    auto chain = [&](int s, int ticket, const double (&sum)[5], double (&P)[5]) __attribute__((always_inline)) {
        // the sums must be complete BEFORE the token is taken: whatever they wait for (the rows of the difference field still in
        // flight, the last rows of M') would otherwise be waited for while every wavefront behind this one waits for the token
        asm volatile("" ::"v"(sum[0]), "v"(sum[1]), "v"(sum[2]), "v"(sum[3]), "v"(sum[4]) : "memory");
        typedef double tok_d2 __attribute__((ext_vector_type(2)));
        const int l = fresh_lane();
        const unsigned ta = (unsigned)(size_t)(__attribute__((address_space(3))) void *)&lds.tok[s];  // a[], b[] at 16 bytes per lane, c[] at 8, tag[] at 4
        const unsigned a16 = ta + 16u * (unsigned)l, a8 = ta + 2048u + 8u * (unsigned)l, a4 = ta + 2560u + 4u * (unsigned)l;
        if (ticket != 0) {
            tok_d2 A, B;
            double C;
            int tag;
            unsigned n = 0;
            do {  // (a busy poll: the token is what every wavefront behind this one waits for)
                asm volatile("ds_read_b32 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:1024\n\tds_read_b64 %3, %6\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(tag), "=&v"(A), "=&v"(B), "=&v"(C) : "v"(a4), "v"(a16), "v"(a8) : "memory");
                if (__builtin_amdgcn_readfirstlane(tag) == ticket) break;
                __builtin_amdgcn_s_sleep(1);
                if (++n > ca.spin) {
                    __hip_atomic_store(ca.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            } while (true);
            P[0] = A.x; P[1] = A.y; P[2] = B.x; P[3] = B.y; P[4] = C;
        }
        {
            const tok_d2 A = {P[0] + sum[0], P[1] + sum[1]}, B = {P[2] + sum[2], P[3] + sum[3]};
            const double C = P[4] + sum[4];
            asm volatile("ds_write_b128 %0, %3\n\tds_write_b128 %0, %4 offset:1024\n\tds_write_b64 %1, %5\n\tds_write_b32 %2, %6"
                         :: "v"(a16), "v"(a8), "v"(a4), "v"(A), "v"(B), "v"(C), "v"(ticket + 1) : "memory");
        }
    };
    // the last three rows of this wavefront's step-s field for the wavefront below
    auto put_boundary = [&](int s, int r, const float (&m)[RW][5]) __attribute__((always_inline)) {
        const int sb = s, seq = r;
        lds_wait(&lds.rd[sb][wave], seq, ca);  // the reader is done with what was here before
        const int l = fresh_lane();
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 5; c++) lds.b[sb][wave][k][c][l] = m[RW - 3 + k][c];
        lds_post(&lds.wr[sb][wave], seq + 1, l);
    };
    auto get_boundary = [&](int s, int r, float (&pv)[3][5]) __attribute__((always_inline)) {
        const int rr = wave == 0 ? r - 1 : r;  // wavefront 0 takes what the last wavefront left in the round before
        const int sb = s, seq = rr;
        lds_wait(&lds.wr[sb][pw], seq + 1, ca);
        const int l = fresh_lane();
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 5; c++) pv[k][c] = lds.b[sb][pw][k][c][l];
        lds_post(&lds.rd[sb][pw], seq + 1, l);
    };

    // This wavefront's rows of the difference field (the reference's srow1[x] - srow0[x]) of a round, requested one round ahead (20
    // registers at four rows per round).  Rows below the image count as zero.
    // NO per-row branches anywhere in a round: rows outside the image are computed at the clamped row index and masked out
    // where they would count (they only occur in the first and the last round), so a round is straight-line code.
    float d[RW][5];
    auto load_d = [&](int r) __attribute__((always_inline)) {
        const int a = r * ca.S + off;
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int y = a + j;
            const unsigned so = (unsigned)min(y, h - 1) * rb;
            // rows below the image: an out-of-range offset, the bounds check returns 0.  (A select behind each load makes the compiler wait for
            // every load where it is issued as soon as register pressure rises: 20 serial round trips per round, measured 414 -> 598 us.)
            const unsigned vo = y < h ? vx : 0xC0000000u;
#pragma unroll
            for (int c = 0; c < 5; c++) d[j][c] = buf_ld<OFXCV_COL_LD_AUX>(bD, vo, so + c * pb);
        }
    };

    // The R0 samples of a round -- rows a-1 .. a+RW-1: step 2 starts one row above step 1 -- are requested a round ahead as well (after the
    // step 1 before; R0 does not depend on the flow).  With the R1 taps coming from LDS, a row then waits for nothing that is further away than LDS.
    [[maybe_unused]] float r0n[RW + 1][5];
    auto load_r0 = [&](int r) __attribute__((always_inline)) {
        const int a = r * ca.S + off;
#pragma unroll
        for (int i = 0; i <= RW; i++) {
            const unsigned so = (unsigned)clampi(a - 1 + i, 0, h - 1) * rb;
#pragma unroll
            for (int c = 0; c < 5; c++) r0n[i][c] = buf_ld<OFXCV_COL_R0_AUX>(bR0, vx, so + c * pb);
        }
    };
    if (!LAST1) load_r0(0);

    // lanes beyond the image edge take the border pixel's flow (see fix_l / fix_r); branch-free
    const bool wg_left = x0 < 2, wg_right = w + 1 - x0 < 63;  // wave-uniform: the workgroup has lanes left / right of the image
    auto border_flow = [&](float &fx, float &fy) __attribute__((always_inline)) {
        // only the first and the last tile column have such lanes; lanes 0, 1 <- lane 2 and (full last tile) lanes 62, 63 <- lane 61 as
        // one DPP quad permutation each, confined to the quad by the row / bank masks
        if (wg_left) {
            fx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fx), __builtin_bit_cast(int, fx), 0xEA, 0x1, 0x1, false));
            fy = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fy), __builtin_bit_cast(int, fy), 0xEA, 0x1, 0x1, false));
        }
        if (wg_right) {
            if (lane_r == 61) {
                fx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fx), __builtin_bit_cast(int, fx), 0x54, 0x8, 0x8, false));
                fy = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fy), __builtin_bit_cast(int, fy), 0x54, 0x8, 0x8, false));
            } else {
                const float rx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx), lane_r));
                const float ry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fy), lane_r));
                fx = xr >= w ? rx : fx;
                fy = xr >= w ? ry : fy;
            }
        }
    };
    for (int r = 0; r < ca.rounds; r++) {
        const int a = r * ca.S + off;  // first step-1 row of this wavefront in this round
        const int ticket = r * NW + wave;
        const bool topw = ticket == 0;  // owns row 0
        stamp(r, 0);
        // ---------------------------------------------------------------- step 1, rows a .. a+RW-1 top to bottom: per row the column sums
        // advance by the row's differences, the 2x2 solve gives its flow, its R0 samples and R1 taps are requested, and the row
        // DEPTH rows earlier -- whose samples have arrived meanwhile -- becomes a row of M'.  The solve of a row (f64 arithmetic,
        // no memory) runs while the gathers of the rows before it are in flight.
        double P[5];
        if (SOLVE1) {
            if (r == 0) load_d(0);  // later rounds: requested while the round before was in its second step
            double sum[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                double t = 0.;
#pragma unroll
                for (int j = 0; j < RW; j++) t += (double)d[j][c];
                sum[c] = t;
                P[c] = 0.;
            }
            if (topw) {
#pragma unroll
                for (int c = 0; c < 5; c++) P[c] = (double)(buf_ld(bEi, vx, c * rb) * 3.f);  // vsum(-1) = srow0 * (m + 2)
            }
            stamp(r, 1);   // rows of the difference field requested
            chain(0, ticket, sum, P);
            stamp(r, 2);   // chain of step 1 passed
            // (holding the token: every reader of the rows this overwrites is done)
            ring_fill(ticket + kLead);
            ring_wait(ticket);
            stamp(r, 3);   // fill issued, the rows this ticket reads have landed
        }
        // Rows below the image repeat the last row (zero differences -> the same column sums -> the same flow -> the same M'):
        // exactly what d'_{h-1} = M'[h-1] - M'[h-3] wants of the row below the image.
        [[maybe_unused]] float r0c[RW + 1][5];
        if (!LAST1) {
#pragma unroll
            for (int i = 0; i <= RW; i++)
#pragma unroll
                for (int c = 0; c < 5; c++) r0c[i][c] = r0n[i][c];
        }
        float m1[RW][5];
        float d2[RW][5];  // d'_t, t = a - 1 + i: rows a+i and a+i-3 of M'
        {
            Px q[RW];
            float fx1[RW], fy1[RW];
#pragma unroll
            for (int p = 0; p < RW + DEPTH; p++) {
                if (p < RW) {
                    const int j = p, y = min(a + j, h - 1);
                    if (SOLVE1) {
#pragma unroll
                        for (int c = 0; c < 5; c++) P[c] += (double)d[j][c];  // the reference's vsum[x] += srow1[x] - srow0[x]
                        solve(P, fx1[j], fy1[j]);
                        if (LAST1) flow_out(a + j, fx1[j], fy1[j]);
                        else border_flow(fx1[j], fy1[j]);
                    } else {
                        fx1[j] = fy1[j] = 0.f;
                        if (K1 == kHaloCoarse) {
                            prolong_flow(flow, flow_step, pr, x, y, fx1[j], fy1[j]);
                        } else if (K1 == kHaloGiven) {
                            const float2 f = *(const float2 *)((const char *)flow + (size_t)y * flow_step + (size_t)x * 8);
                            fx1[j] = f.x;
                            fy1[j] = f.y;
                        }
                    }
                    if (!LAST1) {
#pragma unroll
                        for (int c = 0; c < 5; c++) q[j].r0v[c] = r0c[j + 1][c];
                        q[j].tp = gather(x, y, fx1[j], fy1[j]);
                    }
                }
                if (!LAST1 && p >= DEPTH) {
                    const int j = p - DEPTH, y = min(a + j, h - 1);
                    const M5 mm = finish(q[j], y, fx1[j], fy1[j]);
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        m1[j][c] = mm.v[c];
                        if (j >= 3) d2[j][c] = mm.v[c] - m1[j - 3][c];
                    }
                }
            }
        }
        if constexpr (RING) {
            // this wavefront's fill is published here, a whole step after its issue.  (Where the memory system is busy -- the launch moves its 1.45 GB
            // at 4.5 TB/s -- the loads take thousands of cycles to land; publishing two rows into step 2 instead only moved the wait: r05_experiments.md.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int l = fresh_lane();
            lds_post(&ring.filled[wave], r + 1, l);
        }
        if (SOLVE1 && r + 1 < ca.rounds) load_d(r + 1);  // this round's rows are used up: the next round's arrive during step 2
        if (!LAST1 && r + 1 < ca.rounds) load_r0(r + 1);
        if (LAST1) continue;
        stamp(r, 4);   // M' complete
        put_boundary(0, r, m1);  // rows RW-3 .. RW-1 for the wavefront below
        {
            float pv[3][5];
            if (!topw) get_boundary(0, r, pv);
#pragma unroll
            for (int c = 0; c < 5; c++)
#pragma unroll
                for (int i = 0; i < 3; i++) d2[i][c] = m1[i][c] - (topw ? m1[0][c] : pv[i][c]);  // rows above row 0 are row 0
        }
        // ---------------------------------------------------------------- step 2, rows a-1 .. a+RW-2, the same way from the column sums of M'
        {
            double sum[5];
            if (a < 1 || a - 1 + RW > h) {  // (wave-uniform: only the first and the last rounds have such rows)
#pragma unroll
                for (int i = 0; i < RW; i++) {
                    const int t = a - 1 + i;
                    const bool valid = t >= 0 && t < h;  // wave-uniform
#pragma unroll
                    for (int c = 0; c < 5; c++) d2[i][c] = valid ? d2[i][c] : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < 5; c++) {
                double t = 0.;
#pragma unroll
                for (int i = 0; i < RW; i++) t += (double)d2[i][c];
                sum[c] = t;
                P[c] = topw ? (double)(m1[0][c] * 3.f) : 0.;
            }
            stamp(r, 5);   // boundary rows from above arrived, differences summed
            chain(1, ticket, sum, P);
            stamp(r, 6);   // chain of step 2 passed
        }
        float m2[RW][5];
        auto st_d3 = [&](float dv, int i, int c) __attribute__((always_inline)) {  // d''_t, t = a - 2 + i: rows a-1+i and a-4+i of M''
            const int t = a - 2 + i;
            // lanes that own no column and rows outside the image store to an out-of-range offset: dropped by the bounds check, no branch
            buf_st<OFXCV_COL_ST_AUX>(bDo, dv, (own && t >= 0 && t < h) ? vx : 0xC0000000u, (unsigned)clampi(t, 0, h - 1) * rb + c * pb);
        };
        {
            Px q[RW];
            float fx2[RW], fy2[RW];
#pragma unroll
            for (int p = 0; p < RW + DEPTH; p++) {
                if (p < RW) {
                    const int i = p, y = clampi(a - 1 + i, 0, h - 1);
#pragma unroll
                    for (int c = 0; c < 5; c++) P[c] += (double)d2[i][c];
                    solve(P, fx2[i], fy2[i]);
                    if (LAST2) {
                        flow_out(a - 1 + i, fx2[i], fy2[i]);
                    } else {
                        // row a-1+i: its R0 samples are step 1's of row i-1 unless the row index was clamped there or here (first / last round)
#pragma unroll
                        for (int c = 0; c < 5; c++) q[i].r0v[c] = r0c[i][c];
                        q[i].tp = gather(x, y, fx2[i], fy2[i]);
                    }
                }
                if (!LAST2 && p >= DEPTH) {
                    const int i = p - DEPTH, y = clampi(a - 1 + i, 0, h - 1);
                    const M5 mm = finish(q[i], y, fx2[i], fy2[i]);
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        m2[i][c] = mm.v[c];
                        if (i == 1 && topw) m2[0][c] = mm.v[c];  // the row above row 0 is row 0
                        if (i >= 3) st_d3(mm.v[c] - m2[i - 3][c], i, c);
                    }
                }
            }
        }
        stamp(r, 7);
        if (LAST2) continue;
        stamp(r, 8);   // M'' complete
        put_boundary(1, r, m2);
        if (topw && own) {  // row 0 of M'' for the next launch's vsum(-1)
#pragma unroll
            for (int c = 0; c < 5; c++) buf_st(bEo, m2[1][c], vx, c * rb);
        }
        {
            float pv[3][5];
            if (!topw) get_boundary(1, r, pv);
#pragma unroll
            for (int c = 0; c < 5; c++)
#pragma unroll
                for (int i = 0; i < 3; i++) st_d3(m2[i][c] - (topw ? m2[0][c] : pv[i][c]), i, c);
        }
        stamp(r, 9);   // end of the round
    }
}

// ------------------------------------------------------------------ host-side geometry (optflowgf.cpp calc())

int num_levels(int w, int h, double pyr_scale, int levels) {
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    return k;
}

void level_geom(int w, int h, double pyr_scale, int k, int &lw, int &lh, double &sigma, int &ksize) {
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    sigma = (1. / scale - 1) * 0.5;
    ksize = ofxcv_cv_round(sigma * 5) | 1;
    if (ksize < 3) ksize = 3;
    lw = ofxcv_cv_round(w * scale);
    lh = ofxcv_cv_round(h * scale);
}

inline int plane_pitch(int w) { return (w + 63) & ~63; }

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// How the scratch of one (batched) call is laid out: every pair has the same layout, consecutive pairs lie a fixed
// stride apart, and the kernels pick their pair from the grid's z coordinate.
struct Layout {
    int n = 1;             // frame pairs of the call
    size_t field0 = 0;     // floats of one 5-plane field at level 0
    size_t rtotal = 0;     // floats of R0 + R1 over all levels
    size_t planes = 0;     // floats between the plane scratch (M ping, M pong, R of every level) of consecutive pairs
    size_t t1 = 0;         // floats of the two-pass pyramid fall-back's row buffer (shared, used sequentially)
    size_t img = 0;        // floats between the pyramid images of consecutive frames
    size_t cflow = 0;      // floats of ONE coarse flow field (two per pair)
    size_t vsum = 0;       // doubles between the column-sum scratch of consecutive pairs
    double *vsum_ptr = nullptr;    // column-sum scratch of the first pair of the launch group (set by the level walk)
    size_t planes_bytes() const { return sizeof(float) * planes * n; }
    size_t tmp_bytes() const { return sizeof(float) * (t1 + 2 * (size_t)n * img); }
    size_t flow_bytes() const { return sizeof(float) * 2 * cflow * n; }
    size_t vsum_bytes() const { return sizeof(double) * vsum * n; }
};



// f64 scratch of the OpenCV-order / Gaussian window kernels for one pair at geometry w x h: the larger of
//   serial column scan / Gaussian vertical pass   5 * pitch * h values
//   overlapped strips                             4 * (strips of >= 9 rows + 1) * 5 * pitch + the edge rows (the column-owning form uses the edge rows only)
size_t vsum_doubles(int w, int h) {
    const size_t pitch = (size_t)plane_pitch(w);
    const size_t a = 5 * pitch * h;
    const size_t d = 4 * (size_t)(ofxcv_div_up(h, 9) + 1) * 5 * pitch + 16 * pitch;  // overlapped strips: T, T' and the edge rows, two buffers
    return round_up(std::max(a, d), 32);
}

int make_layout(ofxcv_ctx *ctx, int n, int width, int height, double pyr_scale, int levels, Layout &L) {
    L.n = n;
    L.field0 = 5 * (size_t)plane_pitch(width) * height;
    L.rtotal = 0;
    for (int k = levels; k >= 0; k--) {
        int w, h, ksz;
        double sigma;
        level_geom(width, height, pyr_scale, k, w, h, sigma, ksz);
        if (ksz > kMaxGaussTaps) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "pyramid blur of %d taps exceeds %d", ksz, kMaxGaussTaps);
        L.rtotal += 2 * 5 * (size_t)plane_pitch(w) * h;
    }
    L.planes = 2 * L.field0 + L.rtotal;
    L.t1 = round_up((size_t)2 * (width + 4) * height, 64);  // (twice a frame's rows at level 0: all 32 frames of a call at the levels the fall-back serves)
    L.img = round_up((size_t)width * height, 64);
    L.cflow = 0;
    if (levels > 0) {
        int lw, lh, ks;
        double sg;
        level_geom(width, height, pyr_scale, 1, lw, lh, sg, ks);
        L.cflow = round_up((size_t)lw * lh * 2, 64);
    }
    L.vsum = vsum_doubles(width, height);
    return OFXCV_OK;
}

// F1/F2 for `nimg` frames in one launch (grid z = frame); I of frame i at d_I + i * I_stride
int launch_pyr_image(ofxcv_ctx *ctx, hipStream_t s, const ImgTab &imgs, int nimg, int W, int H, int lw, int lh, double sigma, int ksize,
                     float *d_T1, size_t t1_floats, float *d_I, size_t I_stride) {
    if (ksize > kMaxGaussTaps) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "pyramid blur of %d taps exceeds %d", ksize, kMaxGaussTaps);
    GaussTaps gk;
    make_gauss_taps(ksize, sigma, gk, ctx->fb_gauss_generation);
    gk.fc = ctx->fb_filter_contraction;
    const int fcb = ctx->fb_filter_contraction << 4;  // the same flag for the kernels that take the two taps as scalars (bit 4 of `area`)
    int ntap = (lw == W && lh == H) ? 1 : 2;
    const int area = (W == 2 * lw && H == 2 * lh) ? ctx->fb_resize_generation : 0;  // cv::resize's exact-2x rewrite (resize_combine)
    const bool no_fused = ctx->fb_pyr_mode == 0, pyr_rows = ctx->fb_pyr_mode != 3, pyr_bytewise = ctx->fb_pyr_mode == 2;
    bool aligned = true;
    for (int i = 0; i < nimg; i++) aligned = aligned && ((uintptr_t)imgs.p[i] & 3) == 0 && (imgs.step[i] & 3) == 0;
    const bool dword_ok = !no_fused && ksize == 3 && W >= 16 && H >= 2 && aligned && (I_stride & 3) == 0;
    if (dword_ok && pyr_rows && (W & 3) == 0 && H >= 4 && (ntap == 1 || (W == 2 * lw && H == 2 * lh))) {
        // eight (k = 0) / four (k = 1) output rows per wavefront, four wavefronts per workgroup
        if (ntap == 1)
            hipLaunchKernelGGL((pyr_direct3w_kernel<1, 8>), dim3(ofxcv_div_up(W, 248), ofxcv_div_up(lh, 32), nimg), dim3(256), 0, s, imgs, W, H, lw, lh, gk.k[1], gk.k[2], d_I,
                               I_stride, fcb);
        else
            hipLaunchKernelGGL((pyr_direct3w_kernel<2, 4>), dim3(ofxcv_div_up(W, 248), ofxcv_div_up(lh, 16), nimg), dim3(256), 0, s, imgs, W, H, lw, lh, gk.k[1], gk.k[2], d_I,
                               I_stride, area | fcb);
        OFXCV_LAUNCH_CHECK(ctx, "pyr_direct3w_kernel");
        return OFXCV_OK;
    }
    if (dword_ok && (ntap == 1 || (W == 2 * lw && H == 2 * lh))) {
        dim3 grid(ofxcv_div_up(ofxcv_div_up(W, 4), 64), ofxcv_div_up(lh, 4), nimg), block(64, 4);
        if (ntap == 1) hipLaunchKernelGGL(pyr_direct3v_kernel<1>, grid, block, 0, s, imgs, W, H, lw, lh, gk.k[1], gk.k[2], d_I, I_stride, fcb);
        else hipLaunchKernelGGL(pyr_direct3v_kernel<2>, grid, block, 0, s, imgs, W, H, lw, lh, gk.k[1], gk.k[2], d_I, I_stride, area | fcb);
        OFXCV_LAUNCH_CHECK(ctx, "pyr_direct3v_kernel");
        return OFXCV_OK;
    }
    if (!no_fused && ksize == 3 && W >= 2 && H >= 2) {
        hipLaunchKernelGGL(pyr_direct3_kernel, dim3(ofxcv_div_up(lw, 64), ofxcv_div_up(lh, 4), nimg), dim3(64, 4), 0, s, imgs, W, H, lw, lh, ntap,
                           gk.k[1], gk.k[2], (double)W / lw, (double)H / lh, d_I, I_stride, area | fcb);
        OFXCV_LAUNCH_CHECK(ctx, "pyr_direct3_kernel");
        return OFXCV_OK;
    }
    // the default pyramid's coarse levels: exactly a quarter / an eighth of the frame, 9 / 19 taps
    if (!no_fused && !pyr_bytewise && aligned && ntap == 2 && W >= 64 && H >= 64) {
        const dim3 g(ofxcv_div_up(lw, 32), ofxcv_div_up(lh, 8), nimg);
        if (W == 4 * lw && H == 4 * lh && ksize == 9) {
            hipLaunchKernelGGL((pyr_fused_al_kernel<4, 9>), g, dim3(256), 0, s, imgs, W, H, lw, lh, gk, d_I, I_stride);
            OFXCV_LAUNCH_CHECK(ctx, "pyr_fused_al_kernel");
            return OFXCV_OK;
        }
        if (W == 8 * lw && H == 8 * lh && ksize == 19) {
            hipLaunchKernelGGL((pyr_fused_al_kernel<8, 19>), g, dim3(256), 0, s, imgs, W, H, lw, lh, gk, d_I, I_stride);
            OFXCV_LAUNCH_CHECK(ctx, "pyr_fused_al_kernel");
            return OFXCV_OK;
        }
    }
    // fused tile kernel when the source footprint of a 32x8 (64x8 for small decimation) tile fits in LDS
    PyrTile t;
    t.ow = (double)W / lw <= 2.01 ? 64 : 32;
    t.oh = 8;
    // coarse levels: smaller tiles until there are enough workgroups to spread over the 256 CUs
    while ((long)ofxcv_div_up(lw, t.ow) * ofxcv_div_up(lh, t.oh) * nimg < 512 && (t.ow > 16 || t.oh > 2)) {
        if (t.ow > 16 && t.ow >= 4 * t.oh) t.ow >>= 1;
        else if (t.oh > 2) t.oh >>= 1;
        else t.ow >>= 1;
    }
    const int r = ksize / 2;
    t.cw = ((int)std::ceil((double)(t.ow - 1) * W / lw) + 2 * r + 4 + 3) & ~3;
    t.rh = (int)std::ceil((double)(t.oh - 1) * H / lh) + 2 * r + 4;
    const size_t lds = sizeof(int) * (t.ow + t.oh) + sizeof(float) * 2 * (t.ow + t.oh) + sizeof(float) * (size_t)t.rh * t.ow * ntap +
                       (size_t)t.rh * t.cw;
    if (!no_fused && lds <= 60 * 1024 && lw >= 2 && lh >= 2) {
        hipLaunchKernelGGL(pyr_fused_kernel, dim3(ofxcv_div_up(lw, t.ow), ofxcv_div_up(lh, t.oh), nimg), dim3(256), lds, s, imgs, W, H, lw, lh, ntap,
                           gk, t, d_I, I_stride, area);
        OFXCV_LAUNCH_CHECK(ctx, "pyr_fused_kernel");
        return OFXCV_OK;
    }
    // Two-pass fall-back (the levels beyond 1/8 of a deeper pyramid: 39 taps and more): as many frames per launch as the row buffer
    // holds -- at these levels a frame's half-blurred rows are W * H / 8 floats or less, so the 2n frames of a call are one or two
    // launch pairs (they were 2n pairs of launches, one frame at a time: 1.4 ms of a 6.8 ms call of 8 pairs at levels = 5).
    const int ncol = lw * ntap;
    const size_t per_frame = (size_t)H * ncol;
    const int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)nimg, t1_floats / std::max<size_t>(per_frame, 1)));
    for (int i = 0; i < nimg; i += group) {
        const int g = std::min(group, nimg - i);
        hipLaunchKernelGGL(pyr_hblur_kernel, dim3(ofxcv_div_up(ncol, 256), H, g), dim3(256), 0, s, imgs, i, W, H, lw, ntap, gk, d_T1);
        OFXCV_LAUNCH_CHECK(ctx, "pyr_hblur_kernel");
        hipLaunchKernelGGL(pyr_vblur_resize_kernel, dim3(ofxcv_div_up(lw, 64), ofxcv_div_up(lh, 4), g), dim3(64, 4), 0, s, d_T1, W, H,
                           lw, lh, ntap, gk, d_I + (size_t)i * I_stride, I_stride, area);
        OFXCV_LAUNCH_CHECK(ctx, "pyr_vblur_resize_kernel");
    }
    return OFXCV_OK;
}

// F3 for `nimg` frames in one launch: I of frame i at d_I + i * I_stride, R of frame i at d_R + (i / 2) * pair_stride + (i % 2) * field
int launch_polyexp(ofxcv_ctx *ctx, hipStream_t s, const float *d_I, int w, int h, float *d_R, int poly_n, double poly_sigma, int nimg,
                   size_t I_stride, size_t pair_stride, size_t field, bool pack_odd) {
    const int po = pack_odd ? 1 : 0;  // the odd frames (the second frame of every pair) in the packed form of an R1 field
    if (poly_n < 1 || poly_n > kMaxPolyN) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "poly_n %d outside 1..%d", poly_n, kMaxPolyN);
    PolyCoef pc;
    make_poly_coef(poly_n, poly_sigma, pc);
    int cw = kPeTW + 2 * poly_n, ldw = cw | 1, ih = kPeTH + 2 * poly_n;
    size_t lds = sizeof(float) * ((size_t)ih * ldw + 3 * kPeTH * ldw);
    dim3 grid(ofxcv_div_up(w, kPeTW), ofxcv_div_up(h, kPeTH), nimg);
    if (poly_n == 5 || poly_n == 7) {
        // persistent workgroups, 64 x 16 tiles, six workgroups per CU (the measured best of the round-2/3 variants: tiles of 8 rows and four / five
        // workgroups per CU lost, the one-tile-per-workgroup kernel too: profiles/r03_experiments.md)
        const int th = 16, wgs_per_cu = 6;
        const int tiles_x = (int)grid.x, tiles_y = ofxcv_div_up(h, th), ntiles = tiles_x * tiles_y;
        const int nwg = std::min((ntiles * nimg + 7) & ~7, ctx->num_cus * wgs_per_cu & ~7);  // a multiple of the 8 XCDs
#define OFXCV_LAUNCH_PE(N, TH)                                                                                                          \
    hipLaunchKernelGGL((polyexp_persistent_kernel<N, TH>), dim3(nwg), dim3(256), 0, s, d_I, w, h, d_R, plane_pitch(w), pc, tiles_x, ntiles, \
                       nimg, I_stride, pair_stride, field, po)
        if (poly_n == 5) OFXCV_LAUNCH_PE(5, 16);
        else OFXCV_LAUNCH_PE(7, 16);
#undef OFXCV_LAUNCH_PE
        OFXCV_LAUNCH_CHECK(ctx, "polyexp_persistent_kernel");
        return OFXCV_OK;
    }
    hipLaunchKernelGGL(polyexp_kernel<0>, grid, dim3(256), lds, s, d_I, w, h, d_R, plane_pitch(w), pc, I_stride, pair_stride, field, po);
    OFXCV_LAUNCH_CHECK(ctx, "polyexp_kernel");
    return OFXCV_OK;
}

// One blur+solve(+update) iteration for the n pairs of a call.  R0 / R1 / Min / Mout are pair 0's fields (pair z lies
// z * L.planes floats further), `flows` the per-pair flow outputs (null pointers: the flow stays on chip).  The kernels of
// the default mode (OpenCV-order 3x3 box) take all pairs in one launch; the other window forms are launched pair by pair.
int launch_iteration(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Min, float *Mout, const FlowTab &flows,
                     int w, int h, int winsize, bool update, const Layout &L, bool r1_packed) {
    const int r1q = r1_packed ? 1 : 0;  // R1 in its packed form (whole calls) or planar (the stage-level entry point)
    int m = winsize / 2;
    double scale = 1. / (winsize * winsize);
    const int pitch = plane_pitch(w);
    for (int z = 0; z < L.n; z++) {  // the other window forms: pair by pair
        const float *r0 = R0 ? R0 + (size_t)z * L.planes : nullptr, *r1 = R1 ? R1 + (size_t)z * L.planes : nullptr, *mi = Min + (size_t)z * L.planes;
        float *mo = Mout ? Mout + (size_t)z * L.planes : nullptr, *flow = flows.p[z];
        const size_t flow_step = flows.step[z];
        if (ctx->fb_opencv_rounding && winsize == 3) {  // OpenCV's order as a serial column scan: mode 2 (cross-check of the strip-parallel forms) and the stage-level entry point
            double *V = L.vsum_ptr + (size_t)z * L.vsum;  // reserved by the caller
            hipLaunchKernelGGL(strict_colscan_kernel, dim3(ofxcv_div_up(w, 256), 5), dim3(256), 0, s, mi, w, h, pitch, V);
            OFXCV_LAUNCH_CHECK(ctx, "strict_colscan_kernel");
            dim3 grid(ofxcv_div_up(w, 64), ofxcv_div_up(h, 4)), block(64, 4);
            if (update)
                hipLaunchKernelGGL(strict_solve_kernel<true>, grid, block, 0, s, r0, r1, (const double *)V, mo, flow, flow_step, w, h, pitch, scale, r1q);
            else
                hipLaunchKernelGGL(strict_solve_kernel<false>, grid, block, 0, s, r0, r1, (const double *)V, mo, flow, flow_step, w, h, pitch, scale, r1q);
            OFXCV_LAUNCH_CHECK(ctx, "strict_solve_kernel");
        } else {
            dim3 grid(ofxcv_div_up(w, 64), ofxcv_div_up(h, 4)), block(64, 4);
            if (update)
                hipLaunchKernelGGL(blur_solve_update_kernel<true>, grid, block, 0, s, r0, r1, mi, mo, flow, flow_step, w, h, pitch, m, scale, r1q);
            else
                hipLaunchKernelGGL(blur_solve_update_kernel<false>, grid, block, 0, s, r0, r1, mi, mo, flow, flow_step, w, h, pitch, m, scale, r1q);
            OFXCV_LAUNCH_CHECK(ctx, "blur_solve_update_kernel");
        }
    }
    return OFXCV_OK;
}

int launch_gauss_iteration(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Min, float *Mout, const FlowTab &flows,
                           int w, int h, int winsize, bool update, const Layout &L) {
    WinTaps t;
    t.m = winsize / 2;
    if (t.m + 1 > kMaxWinTaps) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "Gaussian window of %d exceeds %d", winsize, 2 * kMaxWinTaps - 1);
    const double sigma = t.m * 0.3;
    double sum = 1.;
    t.k[0] = 1.f;
    for (int i = 1; i <= t.m; i++) {
        t.k[i] = (float)std::exp(-i * i / (2 * sigma * sigma));
        sum += t.k[i] * 2;
    }
    sum = 1. / sum;
    for (int i = 0; i <= t.m; i++) t.k[i] = (float)(t.k[i] * sum);
    const int pitch = plane_pitch(w);
    dim3 grid(ofxcv_div_up(w, 64), ofxcv_div_up(h, 4)), block(64, 4);
    for (int z = 0; z < L.n; z++) {
        const float *r0 = R0 + (size_t)z * L.planes, *r1 = R1 + (size_t)z * L.planes, *mi = Min + (size_t)z * L.planes;
        float *mo = Mout + (size_t)z * L.planes, *V = (float *)(L.vsum_ptr + (size_t)z * L.vsum);  // reserved by the caller
        hipLaunchKernelGGL(gauss_vpass_kernel, grid, block, 0, s, mi, w, h, pitch, t, V);
        OFXCV_LAUNCH_CHECK(ctx, "gauss_vpass_kernel");
        if (update)
            hipLaunchKernelGGL(gauss_hpass_solve_kernel<true>, grid, block, 0, s, r0, r1, (const float *)V, mo, flows.p[z], flows.step[z], w, h, pitch, t, 1);  // (whole calls only: R1 packed)
        else
            hipLaunchKernelGGL(gauss_hpass_solve_kernel<false>, grid, block, 0, s, r0, r1, (const float *)V, mo, flows.p[z], flows.step[z], w, h, pitch, t, 1);
        OFXCV_LAUNCH_CHECK(ctx, "gauss_hpass_solve_kernel");
    }
    return OFXCV_OK;
}

// OpenCV-order window, overlapped strips (iterate3h_kernel): one launch per iteration.  Strip geometry by the number of
// workgroups the launch has over the whole batch: eight wavefronts of 8 or 9 rows (65..72 computed rows per strip) where that
// still fills the chip, four of 8 or 9 (33..36) below that, four of 5 rows on the small levels (their launches are latency-bound).
struct HaloGeom {
    int rw, nw, tiles_x, nstrips, so;  // so = output rows per strip (computed rows - 3)
};
// Thresholds (workgroups over the whole batch) measured in rounds 3 - 5: eight tall wavefronts from 300 workgroups of 69 rows (a launch of exactly 256
// tall workgroups is one round at half occupancy), eight wavefronts of 5 rows from 200 workgroups of 37 stored rows (960x540 of a single pair).
// Test hook, option "farneback.halo_geom" (never changes a result): 0 by size; low nibble 1 small form, 2 four tall wavefronts, 3 eight tall;
// bits 4..6 the small form's wavefronts (0 / 3: eight of 3 rows, 2: eight of 2, 4: four of 3, 5: four of 5, 6: eight of 5); bits 8.. the computed rows
// of a tall strip (33..36 / 65..72) instead of the choice by launch rounds.
constexpr int kHaloMin8 = 300, kHaloMin5 = 200, kHaloDeep = 2;
HaloGeom halo_geom(const ofxcv_ctx *ctx, int w, int h, int n) {
    HaloGeom g;
    g.tiles_x = ofxcv_div_up(w, kSsW);
    const long t = (long)g.tiles_x * n;
    const int hook = ctx->fb_halo_geom, hook_small = (hook >> 4) & 7, hook_strip = hook >> 8;
    int form = hook & 15;  // 0 = by size, 1 small, 2 four tall wavefronts, 3 eight
    if (form < 1 || form > 3) form = t * ofxcv_div_up(h, 69) >= kHaloMin8 ? 3 : 1;
    if (form == 1) {
        // small levels: eight wavefronts of 3 rows (21 stored rows per strip), and eight of 5 rows (difference field, no rows through LDS) on a level
        // in between: 960x540 of a single pair, 240 such workgroups (16.2 against 17.3 us; on the levels below it the longer wavefronts lose)
        int f = hook_small ? hook_small : 3;
        if (f == 3 && !hook_small && t * ofxcv_div_up(h, 37) >= kHaloMin5) f = 6;
        g.nw = (f == 5 || f == 4) ? 4 : 8;
        g.rw = (f == 5 || f == 6) ? 5 : (f == 2 ? 2 : 3);
        g.so = g.nw * g.rw - 3;
    } else {
        g.nw = form == 3 ? 8 : 4;
        g.rw = 9;
        // computed rows per strip: nw * 8 + 1 .. nw * 9, by the rounds the launch makes over the resident workgroup slots
        // (16 wavefronts per CU): a round that is nearly empty costs almost a full one
        const double slots = 16.0 / g.nw * ctx->num_cus;
        double best_cost = 0;
        int best = g.nw * 9;
        for (int sc = g.nw * 8 + 1; sc <= g.nw * 9; sc++) {
            if (hook_strip > 0 && sc != hook_strip && hook_strip > g.nw * 8 && hook_strip <= g.nw * 9) continue;
            const double r = (double)t * ofxcv_div_up(h, sc - 3) / slots, full = std::floor(r), frac = r - full;
            const double cost = sc * (full + (frac > 0.02 ? 0.3 + 0.7 * frac : 0.0));
            if (best_cost == 0 || cost <= best_cost) {
                best_cost = cost;
                best = sc;
            }
        }
        g.so = best - 3;
    }
    g.nstrips = ofxcv_div_up(h, g.so);
    return g;
}
struct HaloScratch {  // carved from ctx->fb_vsum by the caller (pair 0's; pair z lies L.vsum doubles further)
    double *T[2];
    float *E[2];  // edge rows of M ([3][5][pitch] floats each)
};
size_t halo_edge_doubles(int w) { return 8 * (size_t)plane_pitch(w); }  // 15 * pitch floats, rounded up
size_t halo_scratch_doubles(int w0, int h0) {  // one buffer: T and T' for strips of >= 9 output rows + the edge rows
    return 2 * (size_t)(ofxcv_div_up(h0, 9) + 1) * 5 * plane_pitch(w0) + halo_edge_doubles(w0);
}
HaloScratch halo_scratch(int w0, int h0, const Layout &L) {  // sized for the level-0 geometry (the largest)
    HaloScratch hs;
    const size_t n = halo_scratch_doubles(w0, h0), e = halo_edge_doubles(w0);
    for (int i = 0; i < 2; i++) {
        hs.T[i] = L.vsum_ptr + i * n;
        hs.E[i] = (float *)(hs.T[i] + (n - e));
    }
    return hs;
}
// kind: kHaloLast / kHaloIter = one iteration (Min -> flows / Mout); kHaloZero / kHaloCoarse / kHaloGiven = the first M of a
// level together with its strip sums (Min unused; `flows` = the coarser level's / the caller's flow)
int launch_halo_iteration(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Min, float *Mout, const FlowTab &flows,
                          const Prolong &pr, int w, int h, int kind, const HaloScratch &hs, int slot, const Layout &L, const RgbaTab *rgba = nullptr) {
    RgbaTab rg = {};
    if (rgba && kind == kHaloLast) rg = *rgba;
    const HaloGeom g = halo_geom(ctx, w, h, L.n);
    HaloArgs ha = {hs.T[slot], hs.T[slot ^ 1], hs.E[slot], hs.E[slot ^ 1], g.nstrips, g.so, L.vsum};
    if (kind >= kHaloZero) {  // writes the strip sums / edge rows of the M it produces into slot `slot`
        ha.Ein = nullptr;
        ha.Eout = hs.E[slot];
        ha.Tin = nullptr;
        ha.Tout = hs.T[slot];
    }
    dim3 grid(g.tiles_x, g.nstrips, L.n);
    const int pitch = plane_pitch(w);
    const double scale = 1. / 9.;
    int rc;
    const int mark = ctx->prof_now ? ctx->prof_on : 0;
    if (mark == 1 && (rc = ofxcv_prof_mark(ctx, s))) return rc;
#define OFXCV_LAUNCH_HALO_K(KIND, RW, NW, VAR, DEEP) \
    hipLaunchKernelGGL((iterate3h_kernel<KIND, RW, NW, VAR, DEEP>), grid, dim3(64 * NW), 0, s, R0, R1, Min, Mout, flows, pr, w, h, pitch, scale, ha, L.planes, rg)
#define OFXCV_LAUNCH_HALO(RW, NW, VAR, DEEP)                                      \
    do {                                                                          \
        if (kind == kHaloIter) OFXCV_LAUNCH_HALO_K(kHaloIter, RW, NW, VAR, DEEP);  \
        else if (kind == kHaloLast) OFXCV_LAUNCH_HALO_K(kHaloLast, RW, NW, VAR, false); \
        else if (kind == kHaloZero) OFXCV_LAUNCH_HALO_K(kHaloZero, RW, NW, VAR, DEEP); \
        else if (kind == kHaloCoarse) OFXCV_LAUNCH_HALO_K(kHaloCoarse, RW, NW, VAR, DEEP); \
        else OFXCV_LAUNCH_HALO_K(kHaloGiven, RW, NW, VAR, DEEP);                   \
    } while (0)
    // small form: every gather of a wavefront in flight at once while the launch has at most two wavefronts per SIMD
    const bool deep = g.rw == 5 && (long)g.tiles_x * g.nstrips * L.n * 4 <= (long)kHaloDeep * 4 * ctx->num_cus;
    if (g.rw == 5 && g.nw == 8) OFXCV_LAUNCH_HALO(5, 8, false, false);
    else if (g.rw == 3 && g.nw == 8) OFXCV_LAUNCH_HALO(3, 8, false, true);
    else if (g.rw == 2) OFXCV_LAUNCH_HALO(2, 8, false, true);
    else if (g.rw == 3) OFXCV_LAUNCH_HALO(3, 4, false, true);
    else if (g.rw == 5 && deep) OFXCV_LAUNCH_HALO(5, 4, false, true);
    else if (g.rw == 5) OFXCV_LAUNCH_HALO(5, 4, false, false);
    else if (g.nw == 4) OFXCV_LAUNCH_HALO(9, 4, true, false);
    else OFXCV_LAUNCH_HALO(9, 8, true, false);
#undef OFXCV_LAUNCH_HALO
#undef OFXCV_LAUNCH_HALO_K
    OFXCV_LAUNCH_CHECK(ctx, "iterate3h_kernel");
    if (mark == 1 && (rc = ofxcv_prof_mark(ctx, s))) return rc;
    return OFXCV_OK;
}

// Column-owning form (iterate_col_kernel): two steps of a level per launch, every pair of the group in the grid's z.
struct ColGeom {
    int nw, rw, S, rounds, tiles_x;
};
ColGeom col_geom(int w, int h) {
    ColGeom g;
    g.nw = 8;  // eight wavefronts of four rows: 32-row rounds
    g.rw = 4;
    g.tiles_x = ofxcv_div_up(w, kColW);
    g.S = g.nw * g.rw;
    g.rounds = ofxcv_div_up(h + 2, g.S);  // step 2 runs one row behind step 1, the differences it stores another row behind, and d_{h-1} needs the row below the image
    return g;
}
// a level takes the column-owning form when its launches have enough workgroups (one per tile column and pair) to occupy the chip
bool col_level(const ofxcv_ctx *ctx, int w, int h, int n, bool halo) {
    if (!halo || !ctx->fb_col || h < 64) return false;
    return (long)ofxcv_div_up(w, kColW) * n >= ctx->fb_col_min;
}
// How many of the n pairs of a call walk a w x h level in the column-owning form (the first that many; the others keep the
// overlapped strips -- the forms are per pair, their fields never meet).  One workgroup per tile column and pair, one workgroup per
// CU: a launch lasts ceil(workgroups / CUs) rounds, so 33 tile columns x 8 pairs = 264 workgroups on 256 CUs would be TWO rounds
// (a 1921-pixel-wide frame: 0.94 against 0.65 ms per pair at 1920), and 4 x 32 = 128 workgroups leave half the chip idle for a
// whole round.  Cost model in rounds of the column-owning launch: a pair in strips costs 0.196 x w / 1920 of a round (2 x 39.7 us
// against 405 us at 1920x1080; both scale with the level's height) -- it reproduces where the form was measured to pay
// (profiles/r04_experiments.md: 1080p from 6 pairs, 3840x2160 from 3, not 1080p x 4 or 5).  A farneback.col_min below the default
// (tests) forces the form wherever it reaches that many workgroups.
constexpr int kColMinDefault = 128;
int col_pairs(const ofxcv_ctx *ctx, int w, int h, int n, bool halo) {
    if (!col_level(ctx, w, h, n, halo)) return 0;
    const long T = ofxcv_div_up(w, kColW), cus = std::max(1, ctx->num_cus);
    const double strip_cost = 0.196 * w / 1920.0;
    const bool forced = ctx->fb_col_min < kColMinDefault;
    int ncol = 0;
    double best = forced ? 1e30 : n * strip_cost;  // (all pairs in strips)
    for (int g = n; g >= 1 && T * g >= ctx->fb_col_min; g--) {
        const double cost = (double)ofxcv_div_up(T * g, cus) + (n - g) * strip_cost;
        if (cost < best - 1e-9) {
            best = cost;
            ncol = g;
        }
    }
    return ncol;
}
int launch_col_steps(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Din, float *Dout, const FlowTab &fin, const FlowTab &fout,
                     const Prolong &pr, int w, int h, int k1, int k2, const HaloScratch &hs, int slot, const Layout &L, const RgbaTab *rgba = nullptr) {
    RgbaTab rg = {};
    if (rgba) rg = *rgba;
    const bool iter_pair = k1 == kHaloIter && k2 == kHaloIter;
    const ColGeom g = col_geom(w, h);
    ColArgs ca = {hs.E[slot], hs.E[slot ^ 1], L.vsum, g.S, g.rounds, ctx->fb_col_abort, (unsigned)ctx->fb_col_spin,
                  ctx->fb_col_trace ? (unsigned long long *)((char *)ctx->fb_col_flag.ptr + 256) : nullptr};
    dim3 grid(g.tiles_x, 1, L.n);
    const int pitch = plane_pitch(w);
    const double scale = 1. / 9.;
#define OFXCV_LAUNCH_COL_K(K1, K2, RW, NW, RING, TRACE) \
    hipLaunchKernelGGL((iterate_col_kernel<K1, K2, RW, NW, RING, TRACE>), grid, dim3(64 * NW), 0, s, R0, R1, Din, Dout, fin, fout, pr, w, h, pitch, scale, ca, L.planes, rg)
    // the R1 ring in LDS (option farneback.col_ring, default on): the steps pairs that open with an iteration -- the ring's fill schedule rides on the
    // step-1 token -- in the eight-by-four geometry; everything else gathers from memory
    // (the ring needs ColLds + ColRing = 159 KB of LDS and 16-byte LDS-DMA -- gfx950; anywhere else the launches gather from memory)
    const bool ring_ok = ctx->is_gfx950 && (size_t)ctx->max_lds >= sizeof(ColLds<8>) + sizeof(ColRing);
    const bool ring = ring_ok && ctx->fb_col_ring;
    if (iter_pair && ctx->fb_col_trace && ring) OFXCV_LAUNCH_COL_K(kHaloIter, kHaloIter, 4, 8, true, true);
    else if (k1 == kHaloIter && k2 == kHaloIter && ring) OFXCV_LAUNCH_COL_K(kHaloIter, kHaloIter, 4, 8, true, false);
    else if (k1 == kHaloIter && k2 == kHaloIter) OFXCV_LAUNCH_COL_K(kHaloIter, kHaloIter, 4, 8, false, false);
    else if (k1 == kHaloIter && k2 == kHaloLast && ring) OFXCV_LAUNCH_COL_K(kHaloIter, kHaloLast, 4, 8, true, false);
    else if (k1 == kHaloIter && k2 == kHaloLast) OFXCV_LAUNCH_COL_K(kHaloIter, kHaloLast, 4, 8, false, false);
    else if (k1 == kHaloLast && k2 == kColNone) OFXCV_LAUNCH_COL_K(kHaloLast, kColNone, 4, 8, false, false);
    else if (k1 == kHaloZero && k2 == kHaloIter) OFXCV_LAUNCH_COL_K(kHaloZero, kHaloIter, 4, 8, false, false);
    else if (k1 == kHaloZero && k2 == kHaloLast) OFXCV_LAUNCH_COL_K(kHaloZero, kHaloLast, 4, 8, false, false);
    else if (k1 == kHaloCoarse && k2 == kHaloIter) OFXCV_LAUNCH_COL_K(kHaloCoarse, kHaloIter, 4, 8, false, false);
    else if (k1 == kHaloCoarse && k2 == kHaloLast) OFXCV_LAUNCH_COL_K(kHaloCoarse, kHaloLast, 4, 8, false, false);
    else if (k1 == kHaloGiven && k2 == kHaloIter) OFXCV_LAUNCH_COL_K(kHaloGiven, kHaloIter, 4, 8, false, false);
    else if (k1 == kHaloGiven && k2 == kHaloLast) OFXCV_LAUNCH_COL_K(kHaloGiven, kHaloLast, 4, 8, false, false);
    else return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "iterate_col_kernel: no such pair of steps (%d, %d)", k1, k2);
#undef OFXCV_LAUNCH_COL_K
    OFXCV_LAUNCH_CHECK(ctx, "iterate_col_kernel");
    return OFXCV_OK;
}

FlowTab one_flow(float *p, size_t step) {
    FlowTab t = {};
    t.p[0] = p;
    t.step[0] = step;
    return t;
}
Layout one_pair_layout() { return Layout(); }

}  // namespace

extern "C" {

int ofxcv_farneback_plane_pitch(int width) { return plane_pitch(width); }

// measurement aid (not part of the public header): the shader-clock stamps the traced workgroup of the last iterate_col_kernel
// launch left (option "farneback.col_trace" 1); n 64-bit words
int ofxcv_debug_col_trace(ofxcv_ctx *ctx, unsigned long long *out, int n) {
    if (!ctx || !out || !ctx->fb_col_flag.ptr || (size_t)n * 8 + 256 > ctx->fb_col_flag.bytes) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
    int rc = ofxcv_ctx_quiesce(ctx);
    if (rc) return rc;
    OFXCV_HIP_CHECK(ctx, hipMemcpy(out, (char *)ctx->fb_col_flag.ptr + 256, (size_t)n * 8, hipMemcpyDeviceToHost));
    return OFXCV_OK;
}

int ofxcv_farneback_num_levels(int width, int height, double pyr_scale, int levels) {
    return num_levels(width, height, pyr_scale, levels);
}

int ofxcv_farneback_level_geom(int width, int height, double pyr_scale, int k, int *lw, int *lh, double *sigma, int *ksize) {
    if (!lw || !lh || !sigma || !ksize || width <= 0 || height <= 0 || k < 0) return OFXCV_ERR_INVALID;
    level_geom(width, height, pyr_scale, k, *lw, *lh, *sigma, *ksize);
    return OFXCV_OK;
}

int ofxcv_farneback_pyr_image(ofxcv_ctx *ctx, const uint8_t *d_img, size_t step, int width, int height, int lw, int lh,
                              double sigma, int ksize, float *d_I, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_img || !d_I || width <= 0 || height <= 0 || lw <= 0 || lh <= 0 || ksize < 1 || !(ksize & 1) || step < (size_t)width)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_pyr_image: bad argument");
    int rc = ofxcv_reserve(ctx, ctx->fb_tmp, sizeof(float) * ((size_t)(2 * lw + 2) * height));
    if (rc) return rc;
    ImgTab imgs = {};
    imgs.p[0] = d_img;
    imgs.step[0] = step;
    return launch_pyr_image(ctx, ofxcv_stream(ctx, stream), imgs, 1, width, height, lw, lh, sigma, ksize, (float *)ctx->fb_tmp.ptr,
                            (size_t)(2 * lw + 2) * height, d_I, 0);
}

int ofxcv_farneback_polyexp(ofxcv_ctx *ctx, const float *d_I, int width, int height, float *d_R, int poly_n, double poly_sigma,
                            void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_I || !d_R || width <= 0 || height <= 0) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_polyexp: bad argument");
    return launch_polyexp(ctx, ofxcv_stream(ctx, stream), d_I, width, height, d_R, poly_n, poly_sigma, 1, 0, 0, 0, false);
}

int ofxcv_farneback_update_matrices(ofxcv_ctx *ctx, const float *d_R0, const float *d_R1, const float *d_flow, size_t flow_step,
                                    int width, int height, float *d_M, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_R0 || !d_R1 || !d_flow || !d_M || width <= 0 || height <= 0 || (flow_step & 7))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_update_matrices: bad argument");
    hipLaunchKernelGGL(update_matrices_kernel<2>, dim3(ofxcv_div_up(width, 64), ofxcv_div_up(height, 4)), dim3(64, 4), 0,
                       ofxcv_stream(ctx, stream), d_R0, d_R1, one_flow(const_cast<float *>(d_flow), flow_step), 0, 0, 1.0, 1.0, 1.0, width, height,
                       plane_pitch(width), d_M, (size_t)0, 0);
    OFXCV_LAUNCH_CHECK(ctx, "update_matrices_kernel");
    return OFXCV_OK;
}

int ofxcv_farneback_update_flow_blur(ofxcv_ctx *ctx, const float *d_R0, const float *d_R1, const float *d_M_in, float *d_M_out,
                                     float *d_flow, size_t flow_step, int width, int height, int winsize, int update, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_M_in || width <= 0 || height <= 0 || winsize < 1 || !(winsize & 1) || (d_flow && (flow_step & 7)) ||
        (update && (!d_R0 || !d_R1 || !d_M_out || d_M_out == d_M_in)) || (!update && !d_flow))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "farneback_update_flow_blur: bad argument");
    Layout L = one_pair_layout();
    L.vsum = vsum_doubles(width, height);
    if (ctx->fb_opencv_rounding) {
        int rc = ofxcv_reserve(ctx, ctx->fb_vsum, sizeof(double) * L.vsum);
        if (rc) return rc;
    }
    L.vsum_ptr = (double *)ctx->fb_vsum.ptr;
    return launch_iteration(ctx, ofxcv_stream(ctx, stream), d_R0, d_R1, d_M_in, d_M_out, one_flow(d_flow, flow_step), width, height, winsize,
                            update != 0, L, false);
}

// The launch sequence of one call (n frame pairs).  The pyramid images and polynomial expansions of ALL levels depend only
// on the input frames, so they run on the context's preparation stream (coarsest level first) while the main stream
// walks the levels; an event per level hands R0/R1 over.  The coarse levels of a single pair cannot fill the chip (a
// 240x135 level is 127 workgroups on 256 CUs): their launches are latency-bound, which is what a batch amortises --
// every launch of the walk carries all n pairs in its grid's z dimension.
static int enqueue_farneback(ofxcv_ctx *ctx, hipStream_t s, hipStream_t sp, const Layout &L, const ImgTab &imgs, const FlowTab &out,
                             int width, int height, double pyr_scale, int levels, int winsize,
                             int iterations, int poly_n, double poly_sigma, int flags, bool profile, const RgbaTab *rgba = nullptr) {
    int rc;
    const int n = L.n;
    // scratch carving (sizes were reserved by the caller); pointers are pair 0's
    float *base = (float *)ctx->fb_planes.ptr;
    float *Mbuf[2] = {base, base + L.field0};
    float *Rk = base + 2 * L.field0;  // R0/R1 of level levels, levels-1, ..., 0 packed one after the other
    float *T1 = (float *)ctx->fb_tmp.ptr;
    float *I = T1 + L.t1;
    float *cflow[2] = {(float *)ctx->fb_flow.ptr, (float *)ctx->fb_flow.ptr + L.cflow};

    // fork: the preparation stream starts once the inputs are ready on the main stream
    OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, s));
    OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(sp, ctx->ev_fork, 0));
    float *R[kMaxLevels + 1][2];
    {
        float *p = Rk;
        for (int k = levels; k >= 0; k--) {
            int w, h, ksz;
            double sigma;
            level_geom(width, height, pyr_scale, k, w, h, sigma, ksz);
            const size_t field = 5 * (size_t)plane_pitch(w) * h;
            R[k][0] = p;
            R[k][1] = p + field;
            p += 2 * field;
            // (measurement probe farneback.reuse_prep: the pyramid images and expansions a previous call on the SAME frames left in the scratch are used
            // as they are -- the upper bound of what keeping a named frame's expansions on the device could save; results are unchanged)
            if (!ctx->fb_reuse_prep) {
                rc = launch_pyr_image(ctx, sp, imgs, 2 * n, width, height, w, h, sigma, ksz, T1, L.t1, I, L.img);
                if (rc) return rc;
                rc = launch_polyexp(ctx, sp, I, w, h, R[k][0], poly_n, poly_sigma, 2 * n, L.img, L.planes, field, true);  // R1 = the odd frames, packed
                if (rc) return rc;
            }
            OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_level[k], sp));
        }
    }
    // Launch groups.  A level whose working set (M ping + M pong + R0 + R1 = 80 B/px) for the whole batch stays inside the
    // Infinity Cache is walked with all pairs in every launch; a larger level is walked in groups of as many pairs as fit
    // (at least one), group after group, so that the fields an iteration re-reads are still on the die when it comes back
    // to them (measured at 1920x1080, level 0: 42.5 us per pair and iteration alone, 46.8 us in a batch of three).
    const size_t budget = (size_t)std::max(1, ctx->fb_batch_mb) << 20;
    const size_t pair_cflow = 2 * L.cflow;
    auto sub_tab = [&](const FlowTab &t, int z0, int gn) {
        FlowTab r = {};
        for (int z = 0; z < gn; z++) {
            r.p[z] = t.p[z0 + z];
            r.step[z] = t.step[z0 + z];
        }
        return r;
    };
    auto coarse_tab = [&](float *p0, size_t step) {
        FlowTab t = {};
        for (int z = 0; z < n; z++) {
            t.p[z] = p0 + (size_t)z * pair_cflow;
            t.step[z] = step;
        }
        return t;
    };
    const FlowTab no_flow = {};
    FlowTab prev_all = {};
    bool have_prev = false;
    int pw = 0, ph = 0;
    bool rgba_fused[kMaxBatch] = {};  // F7 of the pair was done by the last level-0 launch (overlapped-strip form); otherwise it follows as its own launch
    hipStream_t s_main = s;
    const bool use_coarse = ctx->coarse && levels > 0 && sp != s && !profile;
    for (int k = levels; k >= 0; k--) {
        int w, h, ksz;
        double sigma;
        level_geom(width, height, pyr_scale, k, w, h, sigma, ksz);
        // the coarse levels may run on a high-priority stream of their own: their short launches are then not queued behind
        // another call's full-size kernels (several calls in flight on one device)
        if (use_coarse && k == levels) OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->coarse, ctx->ev_fork, 0));
        if (use_coarse && k == 0) {
            OFXCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev_coarse, ctx->coarse));
            OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(s_main, ctx->ev_coarse, 0));
        }
        hipStream_t s = (use_coarse && k > 0) ? ctx->coarse : s_main;
        OFXCV_HIP_CHECK(ctx, hipStreamWaitEvent(s, ctx->ev_level[k], 0));  // join (level 0's wait closes the fork)
        const int pitch = plane_pitch(w);
        const size_t level_bytes = 4 * sizeof(float) * 5 * (size_t)pitch * h;
        const int fit = (int)std::min<size_t>((size_t)n, std::max<size_t>(1, budget / level_bytes));
        const bool gaussian = (flags & OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN) != 0;
        // the column-owning form streams every pair of the call through one launch (a workgroup per tile column and pair; the
        // fields are read once per two iterations, so the Infinity-Cache grouping below has nothing to keep on the die)
        // (one exception: a single iteration on a caller-supplied flow at level 0 would pair "first M from the given flow" with "last: flow
        // out" in ONE launch over the SAME buffer -- a workgroup's halo lanes read columns its neighbour overwrites; found by tests/perf/fuzz_halo.py)
        const bool given_in_place = k == 0 && !have_prev && (flags & OFXCV_OPTFLOW_USE_INITIAL_FLOW) && iterations == 1;
        const int ncol = given_in_place ? 0 : col_pairs(ctx, w, h, n, ctx->fb_opencv_rounding == 1 && winsize == 3 && !gaussian);
        struct Group {
            int z0, gn;
            bool col;
        };
        Group plan[kMaxBatch + 1];
        int ngroups = 0;
        if (ncol) plan[ngroups++] = {0, ncol, true};
        if (ncol < n) {
            const int rest = n - ncol;
            const int per_group = ofxcv_div_up(rest, ofxcv_div_up(rest, std::min(fit, rest)));  // groups of equal size (4 pairs, 3 fit: 2 + 2, not 3 + 1)
            for (int z = ncol; z < n; z += per_group) plan[ngroups++] = {z, std::min(per_group, n - z), false};
        }
        const FlowTab out_all = k == 0 ? out : coarse_tab(cflow[k & 1], (size_t)w * 8);
        for (int gi = 0; gi < ngroups; gi++) {
            const int z0 = plan[gi].z0, gn = plan[gi].gn;
            const bool col = plan[gi].col;
            Layout G = L;  // this group's view of the scratch: its first pair is "pair 0" of every launch
            G.n = gn;
            G.vsum_ptr = (double *)ctx->fb_vsum.ptr + (size_t)z0 * L.vsum;
            const size_t po = (size_t)z0 * L.planes;
            float *M0 = Mbuf[0] + po, *M1 = Mbuf[1] + po;
            const float *R0 = R[k][0] + po, *R1 = R[k][1] + po;
            float *Mg[2] = {M0, M1};
            dim3 grid(ofxcv_div_up(w, 64), ofxcv_div_up(h, 4), gn), block(64, 4);
            const FlowTab out_tab = sub_tab(out_all, z0, gn);
            // OpenCV-order window (the library default): overlapped strips, or -- above -- column-owning workgroups
            const bool halo = ctx->fb_opencv_rounding == 1 && winsize == 3 && !gaussian;
            // halo: the level's first field, its strip sums and edge rows come from the iteration kernel's "first" forms in one launch
            const bool halo_first = halo;
            HaloScratch hs = {};
            if (halo) hs = halo_scratch(width, height, G);
            const Prolong no_pr = {0, 0, 1.0, 1.0, 1.0};
            if (col) {
                int fk = kHaloZero;
                FlowTab ftab = no_flow;
                Prolong prc = no_pr;
                if (!have_prev && (flags & OFXCV_OPTFLOW_USE_INITIAL_FLOW)) {
                    // the caller's flow, area-resized to the top level and scaled; at k == 0 it is the flow buffer itself
                    fk = kHaloGiven;
                    ftab = sub_tab(out, z0, gn);
                    if (k > 0) {
                        double scale = 1;
                        for (int i = 0; i < k; i++) scale *= pyr_scale;
                        ftab = sub_tab(coarse_tab(cflow[(k & 1) ^ 1], (size_t)w * 8), z0, gn);
                        for (int z = 0; z < gn; z++) {
                            hipLaunchKernelGGL(initial_flow_kernel, dim3(grid.x, grid.y), block, 0, s, (const float *)out.p[z0 + z], out.step[z0 + z], width, height,
                                               ftab.p[z], w, h, scale);
                            OFXCV_LAUNCH_CHECK(ctx, "initial_flow_kernel");
                        }
                    }
                } else if (have_prev) {
                    fk = kHaloCoarse;
                    ftab = sub_tab(prev_all, z0, gn);
                    prc = {pw, ph, 1. / pyr_scale, (double)pw / w, (double)ph / h, ctx->fb_filter_contraction};
                }
                // steps of the level: first M, iterations - 1 x iterate, last -- two per launch
                const int nsteps = iterations + 1;
                int cur = 0;
                for (int st = 0; st < nsteps; st += 2) {
                    const int k1 = st == 0 ? fk : (st == nsteps - 1 ? kHaloLast : kHaloIter);
                    const int k2 = st + 1 >= nsteps ? kColNone : (st + 1 == nsteps - 1 ? kHaloLast : kHaloIter);
                    RgbaTab rg = {};
                    const bool sink = rgba && k == 0 && (k1 == kHaloLast || k2 == kHaloLast);  // F7 rides on the launch that produces the final flow
                    if (sink) {
                        rg.rsx = rgba->rsx;
                        rg.rsy = rgba->rsy;
                        for (int z = 0; z < gn; z++) {
                            rg.p[z] = rgba->p[z0 + z];
                            rg.step[z] = rgba->step[z0 + z];
                            rg.mu[z] = rgba->mu[z0 + z];
                            rg.mv[z] = rgba->mv[z0 + z];
                            rgba_fused[z0 + z] = true;
                        }
                    }
                    const bool prof = profile && k == 0 && k1 == kHaloIter && k2 == kHaloIter;  // the dominant kernel's launches
                    if (prof && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                    rc = launch_col_steps(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ftab, out_tab, prc, w, h, k1, k2, hs, cur, G, sink ? &rg : nullptr);
                    if (rc) return rc;
                    if (prof && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                    cur ^= 1;
                }
                continue;
            }
            if (!have_prev && (flags & OFXCV_OPTFLOW_USE_INITIAL_FLOW)) {
                // the caller's flow, area-resized to the top level and scaled; at k == 0 it is the flow buffer itself
                FlowTab init = sub_tab(out, z0, gn);
                if (k > 0) {
                    double scale = 1;
                    for (int i = 0; i < k; i++) scale *= pyr_scale;
                    init = sub_tab(coarse_tab(cflow[(k & 1) ^ 1], (size_t)w * 8), z0, gn);
                    for (int z = 0; z < gn; z++) {
                        hipLaunchKernelGGL(initial_flow_kernel, dim3(grid.x, grid.y), block, 0, s, (const float *)out.p[z0 + z], out.step[z0 + z], width, height,
                                           init.p[z], w, h, scale);
                        OFXCV_LAUNCH_CHECK(ctx, "initial_flow_kernel");
                    }
                }
                if (halo_first) rc = launch_halo_iteration(ctx, s, R0, R1, nullptr, M0, init, no_pr, w, h, kHaloGiven, hs, 0, G);
                else hipLaunchKernelGGL(update_matrices_kernel<2>, grid, block, 0, s, R0, R1, init, 0, 0, 1.0, 1.0, 1.0, w, h, pitch, M0, L.planes, 1);
            } else if (!have_prev) {
                if (halo_first) rc = launch_halo_iteration(ctx, s, R0, R1, nullptr, M0, no_flow, no_pr, w, h, kHaloZero, hs, 0, G);
                else hipLaunchKernelGGL(update_matrices_kernel<0>, grid, block, 0, s, R0, R1, no_flow, 0, 0, 1.0, 1.0, 1.0, w, h, pitch, M0, L.planes, 1);
            } else {
                const Prolong pr = {pw, ph, 1. / pyr_scale, (double)pw / w, (double)ph / h, ctx->fb_filter_contraction};
                if (halo_first) rc = launch_halo_iteration(ctx, s, R0, R1, nullptr, M0, sub_tab(prev_all, z0, gn), pr, w, h, kHaloCoarse, hs, 0, G);
                else hipLaunchKernelGGL(update_matrices_kernel<1>, grid, block, 0, s, R0, R1, sub_tab(prev_all, z0, gn), pw, ph, pr.inv_pyr_scale, pr.scale_x,
                                        pr.scale_y, w, h, pitch, M0, L.planes, 1 | (pr.fc << 1));
            }
            if (halo_first && rc) return rc;
            OFXCV_LAUNCH_CHECK(ctx, "update_matrices_kernel");
            int cur = 0;
            for (int i = 0; i < iterations;) {
                const bool prof = profile && k == 0 && i < iterations - 1;  // the dominant kernel's launches
                const bool inner = prof && halo;  // marks set around the kernel inside launch_halo_iteration
                ctx->prof_now = inner;
                if (prof && !inner && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                {
                    bool update = i < iterations - 1;
                    const FlowTab &ft = update ? no_flow : out_tab;
                    if (gaussian)
                        rc = launch_gauss_iteration(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ft, w, h, winsize, update, G);
                    else if (halo) {
                        // F7 rides on the level-0 launch that produces the final flow
                        RgbaTab rg = {};
                        const bool sink = rgba && k == 0 && !update;
                        if (sink) {
                            rg.rsx = rgba->rsx;
                            rg.rsy = rgba->rsy;
                            for (int z = 0; z < gn; z++) {
                                rg.p[z] = rgba->p[z0 + z];
                                rg.step[z] = rgba->step[z0 + z];
                                rg.mu[z] = rgba->mu[z0 + z];
                                rg.mv[z] = rgba->mv[z0 + z];
                                rgba_fused[z0 + z] = true;
                            }
                        }
                        rc = launch_halo_iteration(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ft, no_pr, w, h, update ? kHaloIter : kHaloLast, hs, cur, G, sink ? &rg : nullptr);
                    } else
                        rc = launch_iteration(ctx, s, R0, R1, Mg[cur], Mg[cur ^ 1], ft, w, h, winsize, update, G, true);
                    i += 1;
                }
                ctx->prof_now = false;
                if (rc) return rc;
                if (prof && !inner && (rc = ofxcv_prof_mark(ctx, s))) return rc;
                cur ^= 1;
            }
        }
        prev_all = out_all;
        have_prev = true;
        pw = w;
        ph = h;
    }
    if (rgba) {
        for (int z = 0; z < n; z++) {
            if (!rgba->p[z] || rgba_fused[z]) continue;
            rc = ofxcv_launch_flow_to_rgba(ctx, s_main, out.p[z], out.step[z], width, height, rgba->p[z], rgba->step[z], rgba->mu[z], rgba->mv[z], rgba->rsx, rgba->rsy);
            if (rc) return rc;
        }
    }
    return OFXCV_OK;
}

int ofxcv_calc_optical_flow_farneback_batch(ofxcv_ctx *ctx, int n, const uint8_t *const *d_prev, const size_t *prev_step,
                                            const uint8_t *const *d_next, const size_t *next_step, float *const *d_flow,
                                            const size_t *flow_step, int width, int height, double pyr_scale, int levels, int winsize,
                                            int iterations, int poly_n, double poly_sigma, int flags, void *stream) {
    return ofxcv_calc_optical_flow_farneback_batch_rgba(ctx, n, d_prev, prev_step, d_next, next_step, d_flow, flow_step, width, height, pyr_scale, levels,
                                                        winsize, iterations, poly_n, poly_sigma, flags, nullptr, nullptr, nullptr, nullptr, 1.0, 1.0, stream);
}

int ofxcv_calc_optical_flow_farneback_batch_rgba(ofxcv_ctx *ctx, int n, const uint8_t *const *d_prev, const size_t *prev_step,
                                                 const uint8_t *const *d_next, const size_t *next_step, float *const *d_flow,
                                                 const size_t *flow_step, int width, int height, double pyr_scale, int levels, int winsize,
                                                 int iterations, int poly_n, double poly_sigma, int flags, float *const *d_rgba,
                                                 const ptrdiff_t *rgba_row_bytes, const unsigned *chan_u_mask, const unsigned *chan_v_mask,
                                                 double render_scale_x, double render_scale_y, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (n < 1 || n > kMaxBatch) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: batch of %d pairs outside 1..%d", n, kMaxBatch);
    if (!d_prev || !prev_step || !d_next || !next_step || !d_flow || !flow_step || width <= 0 || height <= 0)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad argument");
    for (int z = 0; z < n; z++)
        if (!d_prev[z] || !d_next[z] || !d_flow[z] || prev_step[z] < (size_t)width || next_step[z] < (size_t)width ||
            flow_step[z] < (size_t)width * 8 || (flow_step[z] & 7) || (((uintptr_t)d_flow[z]) & 7))
            return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad argument (pair %d)", z);
    if (flags & ~(OFXCV_OPTFLOW_USE_INITIAL_FLOW | OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN))
        return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "calc_optical_flow_farneback: flags 0x%x not supported (USE_INITIAL_FLOW, FARNEBACK_GAUSSIAN)", flags);
    if (!(pyr_scale > 0 && pyr_scale < 1) || levels < 0 || iterations < 1 || winsize < 1 || !(winsize & 1))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad parameter");
    // a 5-plane field is addressed through one buffer resource with 32-bit byte offsets: 5 * pitch * height * 4 < 2^31
    if ((size_t)plane_pitch(width) * height * 20 >= (size_t)1 << 31)
        return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "calc_optical_flow_farneback: frames above %d pixels exceed the 32-bit buffer offsets",
                          (int)(((size_t)1 << 31) / 20));
    if (poly_n < 1 || poly_n > kMaxPolyN) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "poly_n %d outside 1..%d", poly_n, kMaxPolyN);
    RgbaTab rgba = {};
    bool have_rgba = false;
    if (d_rgba) {
        if (!rgba_row_bytes || !chan_u_mask || !chan_v_mask || render_scale_x == 0 || render_scale_y == 0)
            return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: bad RGBA argument");
        rgba.rsx = render_scale_x;
        rgba.rsy = render_scale_y;
        for (int z = 0; z < n; z++) {
            if (!d_rgba[z]) continue;
            if ((((uintptr_t)d_rgba[z]) & 3) || (rgba_row_bytes[z] & 3))
                return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "calc_optical_flow_farneback: RGBA image of pair %d is not float-aligned", z);
            rgba.p[z] = d_rgba[z];
            rgba.step[z] = rgba_row_bytes[z];
            rgba.mu[z] = chan_u_mask[z] & 15u;
            rgba.mv[z] = chan_v_mask[z] & 15u;
            have_rgba = true;
        }
    }
    const RgbaTab *rgba_p = have_rgba ? &rgba : nullptr;
    hipStream_t s = ofxcv_stream(ctx, stream);
    // measurement probe (tools/reuse_prep_probe.py), deliberately NOT an option of the C ABI: with OFXCV_DEBUG_REUSE_PREP=1 in the
    // environment a call skips its pyramid images and polynomial expansions and reads what the previous call left in the scratch -- only
    // meaningful when the frames, geometry and batch are those of that call
    {
        const char *e = std::getenv("OFXCV_DEBUG_REUSE_PREP");
        ctx->fb_reuse_prep = e && e[0] == '1';
    }
    levels = num_levels(width, height, pyr_scale, levels);
    if (levels > kMaxLevels) levels = kMaxLevels;

    // scratch per pair: M ping + M pong (level-0 size) + R0/R1 of every level, two coarse flows, the f64 column-sum scratch;
    // per frame: one pyramid image; shared: the row buffer of the two-pass pyramid fall-back
    Layout L;
    int rc = make_layout(ctx, n, width, height, pyr_scale, levels, L);
    if (rc) return rc;
    // (a context that serves calls of varying batch size -- the per-device batch context of the host path, vectorgen.hip -- sizes its scratch for
    // fb_reserve_pairs once: a scratch that grows is freed and re-allocated, and the captured launch sequences hold its addresses)
    Layout Lr = L;
    Lr.n = std::max(n, std::min(ctx->fb_reserve_pairs, kMaxBatch));
    rc = ofxcv_reserve(ctx, ctx->fb_planes, Lr.planes_bytes());
    if (rc) return rc;
    rc = ofxcv_reserve(ctx, ctx->fb_tmp, Lr.tmp_bytes());
    if (rc) return rc;
    if (levels > 0) {
        rc = ofxcv_reserve(ctx, ctx->fb_flow, Lr.flow_bytes());
        if (rc) return rc;
    }
    const bool need_vsum = ctx->fb_opencv_rounding || (flags & OFXCV_OPTFLOW_FARNEBACK_GAUSSIAN);
    if (need_vsum) {
        rc = ofxcv_reserve(ctx, ctx->fb_vsum, Lr.vsum_bytes());
        if (rc) return rc;
    }
    if (ctx->fb_col && !ctx->fb_col_flag.ptr) {  // the trace area of iterate_col_kernel
        rc = ofxcv_reserve(ctx, ctx->fb_col_flag, kColFlagBytes);
        if (rc) return rc;
        OFXCV_HIP_CHECK(ctx, hipMemsetAsync(ctx->fb_col_flag.ptr, 0, kColFlagBytes, s));
    }
    if (ctx->fb_col && !ctx->fb_col_abort) {  // its abort word: pinned, host-coherent (the host reads it at its synchronisation points without a copy)
        std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        void *p = nullptr;
        OFXCV_HIP_CHECK(ctx, hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(p, 0, 64);
        ctx->fb_col_abort = (unsigned *)p;
    }
    rc = ofxcv_farneback_streams(ctx);
    if (rc) return rc;
    hipStream_t sp = ctx->fb_one_stream ? s : ctx->prep;

    ImgTab imgs = {};
    FlowTab out = {};
    for (int z = 0; z < n; z++) {
        imgs.p[2 * z] = d_prev[z];
        imgs.step[2 * z] = prev_step[z];
        imgs.p[2 * z + 1] = d_next[z];
        imgs.step[2 * z + 1] = next_step[z];
        out.p[z] = d_flow[z];
        out.step[z] = flow_step[z];
    }

    // hipGraph replay: the launches of a call are captured once per (pointers, geometry, parameters) and replayed
    // with one hipGraphLaunch.  The measurement hook needs its event pairs between launches and therefore uses the eager path.
    const bool use_graph = !ctx->prof_on && !ctx->fb_no_graph;
    if (!use_graph)
        return enqueue_farneback(ctx, s, sp, L, imgs, out, width, height, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags,
                                 ctx->prof_on != 0, rgba_p);
    FbGraphKey key;
    std::memset(&key, 0, sizeof(key));
    key.n = n;
    key.width = width; key.height = height; key.levels = levels; key.winsize = winsize; key.iterations = iterations; key.poly_n = poly_n; key.flags = flags;
    key.pyr_scale = pyr_scale; key.poly_sigma = poly_sigma;
    key.planes = ctx->fb_planes.ptr; key.tmp = ctx->fb_tmp.ptr; key.cflow = ctx->fb_flow.ptr; key.vsum = need_vsum ? ctx->fb_vsum.ptr : nullptr;
    for (int z = 0; z < n; z++) {
        key.prev[z] = d_prev[z]; key.next[z] = d_next[z]; key.flow[z] = d_flow[z];
        key.prev_step[z] = prev_step[z]; key.next_step[z] = next_step[z]; key.flow_step[z] = flow_step[z];
        key.rgba[z] = rgba.p[z]; key.rgba_step[z] = rgba.step[z]; key.rgba_mu[z] = rgba.mu[z]; key.rgba_mv[z] = rgba.mv[z];
    }
    key.rsx = rgba.rsx; key.rsy = rgba.rsy;
    FbGraph *g = nullptr;
    for (FbGraph &c : ctx->fb_graphs)
        if (c.exec && !std::memcmp(&c.key, &key, sizeof(key))) g = &c;
    if (!g) {
        FbGraph *slot = &ctx->fb_graphs[0];  // an empty slot, else the least recently replayed one
        for (FbGraph &c : ctx->fb_graphs)
            if (!c.exec) { slot = &c; break; }
            else if (c.used < slot->used) slot = &c;
        // Relaxed mode: the captured region itself makes no capture-unsafe call, and other host threads (each with
        // its own context: allocations, synchronising copies) must not be able to invalidate this capture.  If the
        // capture cannot be completed anyway, the call falls back to plain launches and stops using graphs.
        hipGraph_t graph = nullptr;
        // one capture at a time per device, and no device allocation / free / context teardown of another host thread
        // during it (ofxcv_capture_mutex): either was seen to invalidate a capture on ROCm 7.2.  Launches, copies
        // and graph replays of other threads stay concurrent.
        std::unique_lock<std::shared_mutex> capture_lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        const auto hold0 = std::chrono::steady_clock::now();
        if (slot->exec) {  // evicted entry: destroyed under the exclusive lock (see common.h)
            (void)hipGraphExecDestroy(slot->exec);
            slot->exec = nullptr;
        }
        bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
            rc = enqueue_farneback(ctx, s, sp, L, imgs, out, width, height, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags, false, rgba_p);
            ok = hipStreamEndCapture(s, &graph) == hipSuccess && rc == OFXCV_OK && graph != nullptr;
            if (ok) ok = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
        }
        ctx->lock_hold_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - hold0).count();
        ctx->lock_holds++;
        capture_lock.unlock();
        if (!ok) {
            // plain launches from here on, all on the caller's stream (the preparation stream may have been left in
            // the abandoned capture)
            slot->exec = nullptr;
            (void)hipGetLastError();  // clear the sticky capture error
            ctx->fb_no_graph = true;
            ctx->fb_one_stream = true;
            ctx->err[0] = 0;
            return enqueue_farneback(ctx, s, s, L, imgs, out, width, height, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags, false, rgba_p);
        }
        slot->key = key;
        g = slot;
    }
    {
        // exclusive as well: two threads inside hipGraphLaunch at once (different graph execs, different streams) crashed in
        // hip::Graph::UpdateStreams on ROCm 7.2 -- about 1 in 30 runs of four concurrent render threads, backtrace under
        // rocgdb with every other locked operation parked on this lock
        std::unique_lock<std::shared_mutex> launch_lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        const auto hold0 = std::chrono::steady_clock::now();
        g->used = ++ctx->fb_graph_clock;
        const hipError_t le = hipGraphLaunch(g->exec, s);
        ctx->lock_hold_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - hold0).count();
        ctx->lock_holds++;
        OFXCV_HIP_CHECK(ctx, le);
    }
    return OFXCV_OK;
}

int ofxcv_farneback_col_pairs(const ofxcv_ctx *ctx, int width, int height, int n) {
    if (!ctx || width <= 0 || height <= 0 || n <= 0) return 0;
    return col_pairs(ctx, width, height, n, ctx->fb_opencv_rounding == 1);
}

int ofxcv_calc_optical_flow_farneback(ofxcv_ctx *ctx, const uint8_t *d_prev, size_t prev_step, const uint8_t *d_next,
                                      size_t next_step, float *d_flow, size_t flow_step, int width, int height, double pyr_scale,
                                      int levels, int winsize, int iterations, int poly_n, double poly_sigma, int flags, void *stream) {
    // a batch of one: the same launch sequence with a grid z of 1
    return ofxcv_calc_optical_flow_farneback_batch(ctx, 1, &d_prev, &prev_step, &d_next, &next_step, &d_flow, &flow_step, width, height, pyr_scale,
                                                   levels, winsize, iterations, poly_n, poly_sigma, flags, stream);
}

}  // extern "C"
