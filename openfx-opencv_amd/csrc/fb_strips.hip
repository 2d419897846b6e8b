// fb_strips.hip -- F5 + F4, OpenCV-order 3x3 window, overlapped strips: one iteration per launch (iterate3h_kernel)
// (one translation unit of the Farneback path; shared declarations: fb.h)
#include "fb.h"

namespace ofxcv_fb {

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
// kind: kHaloLast / kHaloIter = one iteration (Min -> flows / Mout); kHaloZero / kHaloCoarse / kHaloGiven = the first M of a
// level together with its strip sums (Min unused; `flows` = the coarser level's / the caller's flow)
int launch_halo_iteration(ofxcv_ctx *ctx, hipStream_t s, const float *R0, const float *R1, const float *Min, float *Mout, const FlowTab &flows,
                          const Prolong &pr, int w, int h, int kind, const HaloScratch &hs, int slot, const Layout &L, const RgbaTab *rgba) {
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


}  // namespace ofxcv_fb
