// fb_window.hip -- F4 as a launch of its own (the first matrices of a level), the generic box window, the Gaussian window, the initial flow, the serial column scan
// (one translation unit of the Farneback path; shared declarations: fb.h)
#include "fb.h"

namespace ofxcv_fb {

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

// F6 + first F4 of a level as a launch of its own (the window forms that are not strip-parallel; the stage-level entry point).
// mode 0: zero flow, 1: `flows` prolongated from the coarser level (pr), 2: `flows` as they are
int launch_update_matrices(ofxcv_ctx *ctx, hipStream_t s, int mode, const float *R0, const float *R1, const FlowTab &flows, const Prolong &pr, int w, int h, float *M,
                           size_t pair_stride, int npairs, bool r1_packed) {
    const dim3 grid(ofxcv_div_up(w, 64), ofxcv_div_up(h, 4), npairs), block(64, 4);
    const int pitch = plane_pitch(w), r1q = (r1_packed ? 1 : 0) | (pr.fc << 1);
    if (mode == 0) hipLaunchKernelGGL(update_matrices_kernel<0>, grid, block, 0, s, R0, R1, flows, 0, 0, 1.0, 1.0, 1.0, w, h, pitch, M, pair_stride, r1q);
    else if (mode == 1)
        hipLaunchKernelGGL(update_matrices_kernel<1>, grid, block, 0, s, R0, R1, flows, pr.pw, pr.ph, pr.inv_pyr_scale, pr.scale_x, pr.scale_y, w, h, pitch, M, pair_stride, r1q);
    else hipLaunchKernelGGL(update_matrices_kernel<2>, grid, block, 0, s, R0, R1, flows, 0, 0, 1.0, 1.0, 1.0, w, h, pitch, M, pair_stride, r1q);
    OFXCV_LAUNCH_CHECK(ctx, "update_matrices_kernel");
    return OFXCV_OK;
}
// OPTFLOW_USE_INITIAL_FLOW: the caller's flow, area-resized to a w x h level and scaled
int launch_initial_flow(ofxcv_ctx *ctx, hipStream_t s, const float *flow0, size_t flow0_step, int W, int H, float *flow, int w, int h, double scale) {
    hipLaunchKernelGGL(initial_flow_kernel, dim3(ofxcv_div_up(w, 64), ofxcv_div_up(h, 4)), dim3(64, 4), 0, s, flow0, flow0_step, W, H, flow, w, h, scale);
    OFXCV_LAUNCH_CHECK(ctx, "initial_flow_kernel");
    return OFXCV_OK;
}

}  // namespace ofxcv_fb
