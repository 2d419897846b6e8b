// fb.h -- what the translation units of the Farneback path share (internal; the C ABI is include/ofxcv_hip.h):
// the kernel-argument tables of a batched call, device helpers (buffer addressing, the R1 taps of a pixel, FarnebackUpdateMatrices of one
// pixel, flow prolongation, the f64 wave shifts of the 3-column window), the scratch layout, and the launchers each unit exports.
//   fb_pyramid.hip  F1/F2 pyramid images        fb_polyexp.hip  F3 polynomial expansion       fb_window.hip  first matrices, generic / Gaussian
//   fb_strips.hip   OpenCV-order window, overlapped strips (one iteration per launch)          window, initial flow, serial column scan
//   fb_column.hip   OpenCV-order window, column-owning workgroups (two steps per launch)        farneback.hip  geometry, level walk, entry points
#pragma once
#include <cfloat>
#include <chrono>
#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace ofxcv_fb {

typedef float ofxcv_f4 __attribute__((ext_vector_type(4)));


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

// ------------------------------------------------------------------ host side: geometry, scratch layout, the units' launchers
int num_levels(int w, int h, double pyr_scale, int levels);
void level_geom(int w, int h, double pyr_scale, int k, int &lw, int &lh, double &sigma, int &ksize);
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



size_t vsum_doubles(int w, int h);
int make_layout(ofxcv_ctx *ctx, int n, int width, int height, double pyr_scale, int levels, Layout &L);

// fb_pyramid.hip / fb_polyexp.hip
int launch_pyr_image(ofxcv_ctx *ctx, hipStream_t s, const ImgTab &imgs, int nimg, int W, int H, int lw, int lh, double sigma, int ksize,
                     float *d_T1, size_t t1_floats, float *d_I, size_t I_stride);
int launch_polyexp(ofxcv_ctx *ctx, hipStream_t s, const float *d_I, int w, int h, float *d_R, int poly_n, double poly_sigma, int nimg,
                   size_t I_stride, size_t pair_stride, size_t field, bool pack_odd);
// fb_window.hip
int launch_update_matrices(ofxcv_ctx *ctx, hipStream_t s, int mode, const float *R0, const float *R1, const FlowTab &flows, const Prolong &pr, int w, int h, float *M,
                           size_t pair_stride, int npairs, bool r1_packed);
int launch_initial_flow(ofxcv_ctx *ctx, hipStream_t s, const float *flow0, size_t flow0_step, int W, int H, float *flow, int w, int h, double scale);
int launch_iteration(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Min, float *Mout, const FlowTab &flows,
                     int w, int h, int winsize, bool update, const Layout &L, bool r1_packed);
int launch_gauss_iteration(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Min, float *Mout, const FlowTab &flows,
                           int w, int h, int winsize, bool update, const Layout &L);
// fb_strips.hip
struct HaloScratch {  // carved from ctx->fb_vsum by the caller (pair 0's; pair z lies L.vsum doubles further)
    double *T[2];
    float *E[2];  // edge rows of M ([3][5][pitch] floats each)
};
inline size_t halo_edge_doubles(int w) { return 8 * (size_t)plane_pitch(w); }  // 15 * pitch floats, rounded up
inline size_t halo_scratch_doubles(int w0, int h0) {  // one buffer: T and T' for strips of >= 9 output rows + the edge rows
    return 2 * (size_t)(ofxcv_div_up(h0, 9) + 1) * 5 * plane_pitch(w0) + halo_edge_doubles(w0);
}
inline HaloScratch halo_scratch(int w0, int h0, const Layout &L) {  // sized for the level-0 geometry (the largest)
    HaloScratch hs;
    const size_t n = halo_scratch_doubles(w0, h0), e = halo_edge_doubles(w0);
    for (int i = 0; i < 2; i++) {
        hs.T[i] = L.vsum_ptr + i * n;
        hs.E[i] = (float *)(hs.T[i] + (n - e));
    }
    return hs;
}int launch_halo_iteration(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Min, float *Mout, const FlowTab &flows,
                          const Prolong &pr, int w, int h, int kind, const HaloScratch &hs, int slot, const Layout &L, const RgbaTab *rgba = nullptr);
// fb_column.hip
enum { kColNone = -1 };
constexpr size_t kColFlagBytes = 256 + 64 * 16 * 16 * 8;  // abort word + trace area (64 rounds x 16 wavefronts x 16 stamps)
int col_pairs(const ofxcv_ctx *ctx, int w, int h, int n, bool halo);
int launch_col_steps(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Din, float *Dout, const FlowTab &fin, const FlowTab &fout,
                     const Prolong &pr, int w, int h, int k1, int k2, const HaloScratch &hs, int slot, const Layout &L, const RgbaTab *rgba = nullptr);

}  // namespace ofxcv_fb
