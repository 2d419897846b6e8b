// fb_column.hip -- F5 + F4, OpenCV-order 3x3 window, column-owning workgroups: TWO steps of a level per launch (iterate_col_kernel)
// (one translation unit of the Farneback path; shared declarations: fb.h)
#include "fb.h"

// Wavefront priorities inside iterate_col_kernel (s_setprio; profiles/r06_experiments.md 5).  A workgroup's eight wavefronts sit two per SIMD
// (u and u + 4) and hand a token down the line; who issues when both wavefronts of a SIMD are ready decides how long the line waits.  A wavefront
// working on its step-2 rows goes first (P2), then one on its step-1 rows -- the later wavefront of the SIMD (u + 4: the earlier one's results
// are already on their way down the line) before the earlier one (P1HI > P1LO) -- and everything else (the next round's loads, the token of
// step 1) at 0.  Measured on 8 x 1080p: 325 - 330 us without priorities, 298 - 302 us with (1, 2, 3).  Define OFXCV_COL_NOPRIO to compile them out.
#ifndef OFXCV_COL_P1LO
#define OFXCV_COL_P1LO 1
#endif
#ifndef OFXCV_COL_P1HI
#define OFXCV_COL_P1HI 2
#endif
#ifndef OFXCV_COL_P2
#define OFXCV_COL_P2 3
#endif
#ifdef OFXCV_COL_NOPRIO
#define OFXCV_SETPRIO(p) do { } while (0)
#else
#define OFXCV_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#endif

namespace ofxcv_fb {

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
#ifndef OFXCV_COL_LEAD
#define OFXCV_COL_LEAD 6
#endif
constexpr int kRingD = 4, kRingRows = 64, kRingCols = 64 + 2 * kRingD, kRingLead = OFXCV_COL_LEAD;
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
#ifndef OFXCV_COL_DEPTH
#define OFXCV_COL_DEPTH 4
#endif
    constexpr int DEPTH = OFXCV_COL_DEPTH;  // rows whose samples are in flight before the first is consumed (round 6, with the priorities: 4 is 2 % faster than 1, 207 VGPRs, no spill)
    __shared__ ColLds<NW> lds;
    static_assert(!RING || (TWO && RW == 4 && NW == 8), "the ring's fill schedule rides on the step-1 token of eight wavefronts of four rows");
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
            if (wave < NW / 2) OFXCV_SETPRIO(OFXCV_COL_P1LO);
            else OFXCV_SETPRIO(OFXCV_COL_P1HI);
            stamp(r, 2);   // chain of step 1 passed
            // (holding the token: every reader of the rows this overwrites is done)
            ring_fill(ticket + kLead);
            ring_wait(ticket);
            stamp(r, 3);   // fill issued, the rows this ticket reads have landed
        } else {
            if constexpr (RING) {
                // the first step of a level has no column sums to hand on, but the ring's fill schedule rides on the step-1 token (whoever holds
                // it knows that every reader of the rows its fill overwrites is done): an empty token takes its place
                double none[5] = {0., 0., 0., 0., 0.};
#pragma unroll
                for (int c = 0; c < 5; c++) P[c] = 0.;
                chain(0, ticket, none, P);
            }
            if (wave < NW / 2) OFXCV_SETPRIO(OFXCV_COL_P1LO);
            else OFXCV_SETPRIO(OFXCV_COL_P1HI);
            if constexpr (RING) {
                ring_fill(ticket + kLead);
                ring_wait(ticket);
            }
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
        if (LAST1) {
            OFXCV_SETPRIO(0);
            continue;
        }
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
        OFXCV_SETPRIO(OFXCV_COL_P2);
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
        if (LAST2) {
            OFXCV_SETPRIO(0);
            continue;
        }
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
        OFXCV_SETPRIO(0);
        stamp(r, 9);   // end of the round
    }
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
// whole round.  Cost model in rounds of the column-owning launch: a pair in strips costs 0.237 x w / 1920 of a round (2 x 35.8 us
// against 302 us at 1920x1080; both scale with the level's height) -- it reproduces where the form was measured to pay
// (profiles/r04_experiments.md: 1080p from 6 pairs, 3840x2160 from 3, not 1080p x 4 or 5).  A farneback.col_min below the default
// (tests) forces the form wherever it reaches that many workgroups.
constexpr int kColMinDefault = 128;
int col_pairs(const ofxcv_ctx *ctx, int w, int h, int n, bool halo) {
    if (!col_level(ctx, w, h, n, halo)) return 0;
    const long T = ofxcv_div_up(w, kColW), cus = std::max(1, ctx->num_cus);
    const double strip_cost = 0.237 * w / 1920.0;  // (round 6: 2 x 35.8 us against the 302 us round; rounds 4 - 5: 0.196 = 2 x 39.7 / 405)
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
                     const Prolong &pr, int w, int h, int k1, int k2, const HaloScratch &hs, int slot, const Layout &L, const RgbaTab *rgba) {
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
    else if (k1 == kHaloCoarse && k2 == kHaloIter && ring) OFXCV_LAUNCH_COL_K(kHaloCoarse, kHaloIter, 4, 8, true, false);
    else if (k1 == kHaloCoarse && k2 == kHaloIter) OFXCV_LAUNCH_COL_K(kHaloCoarse, kHaloIter, 4, 8, false, false);
    else if (k1 == kHaloCoarse && k2 == kHaloLast) OFXCV_LAUNCH_COL_K(kHaloCoarse, kHaloLast, 4, 8, false, false);
    else if (k1 == kHaloGiven && k2 == kHaloIter) OFXCV_LAUNCH_COL_K(kHaloGiven, kHaloIter, 4, 8, false, false);
    else if (k1 == kHaloGiven && k2 == kHaloLast) OFXCV_LAUNCH_COL_K(kHaloGiven, kHaloLast, 4, 8, false, false);
    else return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "iterate_col_kernel: no such pair of steps (%d, %d)", k1, k2);
#undef OFXCV_LAUNCH_COL_K
    OFXCV_LAUNCH_CHECK(ctx, "iterate_col_kernel");
    return OFXCV_OK;
}


}  // namespace ofxcv_fb
