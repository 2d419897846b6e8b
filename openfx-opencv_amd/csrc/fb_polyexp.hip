// fb_polyexp.hip -- F3: FarnebackPolyExp + FarnebackPrepareGaussian (optflowgf.cpp)
// (one translation unit of the Farneback path; shared declarations: fb.h)
#include "fb.h"

namespace ofxcv_fb {

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

// ------------------------------------------------------------------ F3 polynomial expansion
//
// One 64x16 output tile per 256-thread block.  The tile of I plus an n-pixel halo is staged in
// LDS (rows and columns clamped = the replicate border of the reference), the vertical pass
// writes its three f32 sums per (row, column) back to LDS, and the horizontal pass accumulates
// the six moments in f64 exactly like the reference's inner loop.

constexpr int kPeTW = 64, kPeTH = 16;
typedef float ofxcv_f2 __attribute__((ext_vector_type(2)));

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


}  // namespace ofxcv_fb
