// fb_pyramid.hip -- F1/F2: convertTo(32F) + GaussianBlur + resize(INTER_LINEAR), the pyramid image of one level (optflowgf.cpp calc(), smooth.cpp, imgwarp.cpp)
// (one translation unit of the Farneback path; shared declarations: fb.h)
#include "fb.h"

namespace ofxcv_fb {

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


}  // namespace ofxcv_fb
