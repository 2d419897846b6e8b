// inpaint.hip -- the OpenCV calls of the opencv2fx/inpaint render() body (opencv2fx/inpaint/inpaint.cpp:303-318):
//   I0  cvCvtColor(RGBA2RGB) x2, cvCvtColor(RGBA2GRAY)            :303-305
//   I1  cvThreshold(mask, mask, 0, 255, CV_THRESH_BINARY_INV)       :307
//   I2  cvDilate(mask, mask, NULL, (int)t2)                         :309
//   I3  cvInpaint set-up: band / ring maps on the 1-pixel padded image
//   I4  cvInpaint(image0, mask, image1, t1, CV_INPAINT_TELEA)       :311-318  (photo/src/inpaint.cpp)
//
// Split of the Telea algorithm (see DESIGN.md "inpaint"):
//   * I0-I2 are exact-integer pixel kernels (one coalesced dword per lane).
//   * The fast-marching front (distance map T and the order in which hole pixels are filled) depends only
//     on the hole mask, never on colours, and is a strictly sequential priority-queue recurrence: a pixel
//     receives its T once, from the state at the moment its first neighbour is accepted.  It touches only
//     hole pixels (a few % of the frame) and runs on the host thread that owns the context.
//   * The colour fill (>90 % of the arithmetic: a (2r+1)^2 weighted window per pixel and channel) runs on
//     the GPU.  A pixel may only be computed after every earlier-filled pixel inside its window; pixels
//     are grouped into dependency levels, holes that cannot influence each other form independent
//     components, and one persistent workgroup per component walks its levels with a barrier in between.
//     One wavefront fills one pixel: it stages the (2r+3)^2 neighbourhood (distance, fill order, colours as
//     the sequential algorithm would see them at that moment) in LDS, evaluates the window taps one per
//     lane, and then accumulates them strictly in the reference's row-major order (float addition is not
//     associative and the result is rounded twice, so the order matters for bit-exact colours).
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <queue>
#include <vector>

#include <chrono>
#include <cstdlib>

#include "common.h"
#include "telea_march.h"

namespace {

using namespace ofxcv_telea;            // flags, March, march_begin / march_advance (telea_march.h)
constexpr int kFillWavesHost = 16;     // wavefronts per fill workgroup (kFillThreads / 64); 8 for the large-window instantiation

// ------------------------------------------------------------------ I0-I2 kernels

// RGBA -> hole mask: gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14 (cvCvtColor RGBA2GRAY, channel 0 = R),
// mask = gray > 0 ? 0 : 255 (THRESH_BINARY_INV at 0)
__global__ __launch_bounds__(256) void rgba_to_mask_kernel(const uint8_t *__restrict__ src, ptrdiff_t row_bytes, int w, int h,
                                                           uint8_t *__restrict__ mask, ptrdiff_t mask_step) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint32_t p = *(const uint32_t *)(src + (ptrdiff_t)y * row_bytes + (size_t)x * 4);
    int r = p & 255, g = (p >> 8) & 255, b = (p >> 16) & 255;
    int gray = (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14;
    mask[(ptrdiff_t)y * mask_step + x] = gray > 0 ? 0 : 255;
}

// `iters` 3x3-rect dilations == one (2*iters+1)^2 max; pixels outside the image never contribute
__global__ __launch_bounds__(256) void dilate_rect_kernel(const uint8_t *__restrict__ src, ptrdiff_t src_step, int w, int h, int r,
                                                          uint8_t *__restrict__ dst, ptrdiff_t dst_step) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint8_t m = 0;
    for (int dy = -r; dy <= r; dy++) {
        int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        const uint8_t *row = src + (ptrdiff_t)yy * src_step;
        for (int dx = -r; dx <= r; dx++) {
            int xx = x + dx;
            if (xx >= 0 && xx < w) m = max(m, row[xx]);
        }
    }
    dst[(ptrdiff_t)y * dst_step + x] = m;
}

// ------------------------------------------------------------------ I3/I4 front march (host): telea_march.h

// level(p) = 1 + max level of the pixels filled before p within Chebyshev distance range+2 (every pixel whose
// colour p can read: window range, +1 for the image-gradient taps, +1 for the row/column-1 sample quirk).
// Pixels further apart than 2*(range+2) never influence each other: connected groups of occupied coarse
// cells are independent components, each walked by its own workgroup.
void build_levels(March &m, bool dataflow) {
    const int ec = m.w + 2, er = m.h + 2, R = m.range + 2;
    const int n = (int)m.pix.size();
    m.level.assign(n, 1);
    std::vector<int> lvl_map((size_t)ec * er, 0);
    const bool windowed = (long)(2 * R + 1) * (2 * R + 1) * n <= 400000000L;
    for (int k = 0; k < n && !dataflow; k++) {
        const int p = m.pix[k], i = p / ec, j = p % ec;
        int lv = 0;
        if (windowed) {
            for (int a = std::max(i - R, 1); a <= std::min(i + R, er - 2); a++) {
                const int *row = lvl_map.data() + (size_t)a * ec;
                for (int b = std::max(j - R, 1); b <= std::min(j + R, ec - 2); b++) lv = std::max(lv, row[b]);
            }
        } else {
            lv = k;  // very large radius: plain sequential order
        }
        m.level[k] = lv + 1;
        lvl_map[p] = lv + 1;
    }
    // components: cells of 2R+1 pixels; two hole pixels closer than 2R+1 lie in the same or in 8-adjacent cells
    const int cs = 2 * R + 1, gw = (ec + cs - 1) / cs, gh = (er + cs - 1) / cs;
    std::vector<int> cell((size_t)gw * gh, -1);
    for (int k = 0; k < n; k++) cell[(size_t)(m.pix[k] / ec / cs) * gw + (m.pix[k] % ec) / cs] = -2;  // occupied
    int ncomp = 0;
    std::vector<int> stack;
    if (!windowed) {  // sequential order: a single component
        for (int &c : cell)
            if (c == -2) c = 0;
        ncomp = n > 0 ? 1 : 0;
    } else {
        for (int c0 = 0; c0 < gw * gh; c0++) {
            if (cell[c0] != -2) continue;
            cell[c0] = ncomp;
            stack.assign(1, c0);
            while (!stack.empty()) {
                int c = stack.back();
                stack.pop_back();
                int cy = c / gw, cx = c % gw;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        int yy = cy + dy, xx = cx + dx;
                        if (yy < 0 || xx < 0 || yy >= gh || xx >= gw || cell[yy * gw + xx] != -2) continue;
                        cell[yy * gw + xx] = ncomp;
                        stack.push_back(yy * gw + xx);
                    }
            }
            ncomp++;
        }
    }
    // sort pixels by (component, level, order) and build the two-level CSR
    std::vector<int> comp(n), idx(n);
    for (int k = 0; k < n; k++) {
        comp[k] = cell[(size_t)(m.pix[k] / ec / cs) * gw + (m.pix[k] % ec) / cs];
        idx[k] = k;
    }
    if (dataflow) {  // counting sort by component keeps the fill order inside each component
        m.cmp_off.assign(ncomp + 1, 0);
        for (int k = 0; k < n; k++) m.cmp_off[comp[k] + 1]++;
        for (int c = 0; c < ncomp; c++) m.cmp_off[c + 1] += m.cmp_off[c];
        std::vector<int> fillp(m.cmp_off.begin(), m.cmp_off.end() - 1);
        m.cmp_pix.resize(n);
        m.cmp_ord.resize(n);
        for (int k = 0; k < n; k++) {
            const int q = fillp[comp[k]]++;
            m.cmp_pix[q] = m.pix[k];
            m.cmp_ord[q] = k + 1;
        }
        // workgroups per component: a wavefront spends a few microseconds per pixel even when it never waits, so a large
        // component is spread over up to 8 workgroups (they poll each other's results through the L2)
        m.cmp_wg.clear();
        for (int c = 0; c < ncomp; c++) {
            const int nc = m.cmp_off[c + 1] - m.cmp_off[c];
            const int g = std::min(8, std::max(1, (nc + 1999) / 2000));
            for (int r = 0; r < g; r++) {
                const int rec[4] = {m.cmp_off[c], m.cmp_off[c + 1], r * kFillWavesHost, g * kFillWavesHost};
                m.cmp_wg.insert(m.cmp_wg.end(), rec, rec + 4);
            }
        }
        return;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return comp[a] != comp[b] ? comp[a] < comp[b] : m.level[a] < m.level[b]; });
    m.lvl_pix.resize(n);
    m.lvl_ord.resize(n);
    m.lvl_off.clear();
    m.comp_off.assign(1, 0);
    for (int q = 0; q < n; q++) {
        const int k = idx[q];
        m.lvl_pix[q] = m.pix[k];
        m.lvl_ord[q] = k + 1;
        const bool new_comp = q == 0 || comp[k] != comp[idx[q - 1]];
        if (new_comp && q > 0) m.comp_off.push_back((int)m.lvl_off.size());
        if (new_comp || m.level[k] != m.level[idx[q - 1]]) m.lvl_off.push_back(q);
    }
    m.lvl_off.push_back(n);
    m.comp_off.push_back((int)m.lvl_off.size() - 1);
    if (n == 0) m.comp_off.assign(1, 0);
}

// Dataflow schedule of ONE portion of the fill order, pixels [k0, k1): their components (connected groups of occupied
// coarse cells, as in build_levels) in fill order, 1..8 workgroups per component.  Appends to the call-wide arrays
// (sched_pix / sched_ord hold the pixels of all portions one after the other; a workgroup record addresses them
// absolutely); returns the number of workgroups added.  Pixels of earlier portions are complete when this portion's
// launch starts (stream order), later ones still carry kNeverFilled in the order map: the polls only ever wait inside
// the portion.
int build_dataflow_portion(const March &m, int k0, int k1, std::vector<int> &sched_pix, std::vector<int> &sched_ord, std::vector<int> &sched_wg,
                           std::vector<int> &cell, std::vector<int> &stack, int per_wg, int max_wg, int waves = kFillWavesHost) {
    const int ec = m.w + 2, er = m.h + 2, R = m.range + 2;
    const int cs = 2 * R + 1, gw = (ec + cs - 1) / cs, gh = (er + cs - 1) / cs;
    const int n = k1 - k0;
    if (n <= 0) return 0;
    if (cell.size() != (size_t)gw * gh) cell.assign((size_t)gw * gh, -1);
    std::vector<int> occupied;
    for (int k = k0; k < k1; k++) {
        const int c = (m.pix[k] / ec / cs) * gw + (m.pix[k] % ec) / cs;
        if (cell[c] == -1) {
            cell[c] = -2;
            occupied.push_back(c);
        }
    }
    int ncomp = 0;
    for (int c0 : occupied) {
        if (cell[c0] != -2) continue;
        cell[c0] = ncomp;
        stack.assign(1, c0);
        while (!stack.empty()) {
            int c = stack.back();
            stack.pop_back();
            int cy = c / gw, cx = c % gw;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = cy + dy, xx = cx + dx;
                    if (yy < 0 || xx < 0 || yy >= gh || xx >= gw || cell[yy * gw + xx] != -2) continue;
                    cell[yy * gw + xx] = ncomp;
                    stack.push_back(yy * gw + xx);
                }
        }
        ncomp++;
    }
    // counting sort by component keeps the fill order inside each component
    std::vector<int> off(ncomp + 1, 0), comp(n);
    for (int k = 0; k < n; k++) {
        comp[k] = cell[(m.pix[k0 + k] / ec / cs) * gw + (m.pix[k0 + k] % ec) / cs];
        off[comp[k] + 1]++;
    }
    for (int c = 0; c < ncomp; c++) off[c + 1] += off[c];
    const int base = (int)sched_pix.size();
    sched_pix.resize(base + n);
    sched_ord.resize(base + n);
    std::vector<int> fillp(off.begin(), off.end() - 1);
    for (int k = 0; k < n; k++) {
        const int q = base + fillp[comp[k]]++;
        sched_pix[q] = m.pix[k0 + k];
        sched_ord[q] = k0 + k + 1;
    }
    int nwg = 0;
    for (int c = 0; c < ncomp; c++) {
        const int nc = off[c + 1] - off[c];
        const int g = std::min(max_wg, std::max(1, (nc + per_wg - 1) / per_wg));
        for (int r = 0; r < g; r++) {
            const int rec[4] = {base + off[c], base + off[c + 1], r * waves, g * waves};
            sched_wg.insert(sched_wg.end(), rec, rec + 4);
            nwg++;
        }
    }
    for (int c : occupied) cell[c] = -1;  // the grid is reused by the next portion
    return nwg;
}

// Tile schedule of ONE portion (round 4): the portion's pixels grouped by square tiles of `ts` padded-image pixels, one workgroup per
// occupied tile, its 16 wavefronts taking the tile's pixels round-robin in fill order.  Dependencies are spatially local (window
// radius range + 2), so most of a pixel's predecessors lie in its own tile and reach it through the workgroup's LDS slots
// (telea_fill_kernel: `ts`); only those across a tile edge go through the L2.  Every workgroup of the launch can wait for any
// other, so all of them must be resident at once: the caller caps the number of tiles (`max_tiles`: this call's share of the
// kTileBudget workgroups of 1024 threads the chip holds one per CU with room to spare) and the smallest tile size that stays under
// the cap is taken; returns -1 if even the largest does not do.  A launch takes as long as its fullest tile (16 wavefronts, ~10 us
// per pixel and wavefront): more and smaller tiles shorten it (48 -> 192 tiles, 16-pixel tiles allowed: the launches of the
// middle of a 1080p fill 240-390 -> 110-230 us; the last two, the dense centres of the holes, stay chain-bound at ~600 us each).
constexpr int kTileBudget = 192, kMinTileGroups = 48;
int build_tile_portion(March &m, int k0, int k1, std::vector<int> &sched_pix, std::vector<int> &sched_ord, std::vector<int> &sched_wg,
                       std::vector<int> &cell, int &ts_out, int max_tiles, int waves = kFillWavesHost) {
    const int ec = m.w + 2, er = m.h + 2;
    const int n = k1 - k0;
    if (n <= 0) return 0;
    static const int sizes[] = {16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512};
    // row and column of every pixel once (one division each), the tile of a row / column from a table per tile size: the passes
    // below cost no division per pixel (they had four each: 1.2 of the 11.5 ms of a 1080p call)
    std::vector<int> &pr = m.sc_r, &pc = m.sc_c, &rt = m.sc_rt, &ct = m.sc_ct, &tile = m.sc_tile;  // scratch kept with the context
    pr.resize(n); pc.resize(n); tile.resize(n);
    for (int k = 0; k < n; k++) {
        const int p = m.pix[k0 + k], i = p / ec;
        pr[k] = i;
        pc[k] = p - i * ec;
    }
    rt.resize(er); ct.resize(ec);
    for (int ts : sizes) {
        const int gw = (ec + ts - 1) / ts, gh = (er + ts - 1) / ts;
        for (int i = 0; i < er; i++) rt[i] = (i / ts) * gw;
        for (int j = 0; j < ec; j++) ct[j] = j / ts;
        cell.assign((size_t)gw * gh, -1);
        int ntile = 0, k = 0;
        for (; k < n && ntile <= max_tiles; k++) {
            int &c = cell[(size_t)rt[pr[k]] + ct[pc[k]]];
            if (c < 0) c = ntile++;
            tile[k] = c;
        }
        if (ntile > max_tiles) continue;
        std::vector<int> off(ntile + 1, 0);
        for (k = 0; k < n; k++) off[tile[k] + 1]++;
        for (int t = 0; t < ntile; t++) off[t + 1] += off[t];
        const int base = (int)sched_pix.size();
        sched_pix.resize(base + n);
        sched_ord.resize(base + n);
        std::vector<int> fillp(off.begin(), off.end() - 1);
        for (k = 0; k < n; k++) {  // ascending k: fill order is kept inside a tile
            const int q = base + fillp[tile[k]]++;
            sched_pix[q] = m.pix[k0 + k];
            sched_ord[q] = k0 + k + 1;
        }
        for (int t = 0; t < ntile; t++) {
            const int rec[4] = {base + off[t], base + off[t + 1], 0, waves};
            sched_wg.insert(sched_wg.end(), rec, rec + 4);
        }
        ts_out = ts;
        return ntile;
    }
    return -1;
}

// ------------------------------------------------------------------ I4 colour fill (device)

constexpr int kFillThreads = 1024;             // 16 wavefronts: 16 pixels of a level in flight per workgroup
constexpr int kFillWaves = kFillThreads / 64;
static_assert(kFillWaves == kFillWavesHost, "the host schedule assumes 16 wavefronts per workgroup");
constexpr int kAcc = 10;                       // Ia[3], Jx[3], Jy[3], s
constexpr int kMaxLdsRange = 5;                // (2r+3)^2 <= 169 neighbourhood entries staged in LDS: sixteen wavefronts per workgroup
constexpr int kBigLdsRange = 12;               // ... <= 729 entries: the large-window instantiation, eight wavefronts per workgroup (radius 6 .. 12)
constexpr int kBigFillWaves = 8;
constexpr int kFillSlots = 8192;               // tile schedule: one LDS dword per fill-order pixel of a launch (the default portion)

// The fill works on 4-byte pixels (R | G<<8 | B<<16 | X<<24), w*h dwords: one aligned load / store per pixel.
struct FillArgs {
    const float *t;         // padded distance map
    const int *ord;         // padded order map
    const uint32_t *src;    // original image
    uint32_t *out;          // image being filled (initialised to src)
    int w, h, range;
    const int *lvl_pix, *lvl_ord, *lvl_off, *comp_off;   // level schedule (large radius)
    const int *cmp_pix, *cmp_ord, *cmp_off;              // dataflow schedule: the pixels of each component in fill order
    int *err;                                            // dataflow: set when a poll gave up (see the poll loop)
    unsigned long long *trace;                           // measurement hook (OFXCV_FILL_TRACE): 8 shader-clock stamps per fill-order pixel, or null
    int spin_limit;                                      // polls a wavefront spends on one awaited colour before it gives up
    int k0;                                              // tile schedule: the order numbers of this launch's pixels are k0+1 .. k0+kFillSlots at most
    int ts;                                              // tile schedule: tile size (padded pixels); 0 = component schedule (all polls through the L2)
};

__device__ __forceinline__ void wave_lds_sync() {  // LDS hand-over between lanes of ONE wavefront
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <int CN>
__global__ __launch_bounds__(256) void pack_rgbx_kernel(const uint8_t *__restrict__ src, ptrdiff_t step, int w, int h,
                                                        uint32_t *__restrict__ a, uint32_t *__restrict__ b) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t *p = src + (ptrdiff_t)y * step + (size_t)x * CN;
    uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | (CN == 4 ? (uint32_t)p[3] << 24 : 0u);
    a[(size_t)y * w + x] = v;
    b[(size_t)y * w + x] = v & 0x00ffffffu;  // top byte of the working copy: "filled" tag
}
template <int CN>
__global__ __launch_bounds__(256) void unpack_rgbx_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ alpha, int w, int h,
                                                          uint8_t *__restrict__ dst, ptrdiff_t step) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint32_t v = (src[(size_t)y * w + x] & 0x00ffffffu) | (alpha[(size_t)y * w + x] & 0xff000000u);
    uint8_t *p = dst + (ptrdiff_t)y * step + (size_t)x * CN;
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    if (CN == 4) p[3] = (uint8_t)(v >> 24);
}

// LDSWIN: the (2r+3)^2 neighbourhood of the pixel is staged in LDS (range <= kMaxLdsRange); otherwise every
// access goes to global memory (any range up to 100).
//
// NS: the colour rule of icvNSInpaintFMM instead of icvTeleaInpaintFMM (weights from the isophote direction, built
// from absolute byte differences, no level-set term): per channel the accumulators are Ia (lanes 0..2) and the weight
// sum s (lanes 3..5); everything else -- fill order, staging, ordered accumulation -- is shared.
//
// Scheduling.  A filled pixel is written as R | G<<8 | B<<16 | 1<<24: the top byte of `out` is a "filled" tag (0 in the
// initial copy), so one 4-byte load returns the colour together with the fact that it is final.
//  * LDSWIN (dataflow): a component gets 1..8 workgroups (16 wavefronts each) according to its size; their wavefronts
//    take the component's pixels round-robin in fill order.  While staging its neighbourhood a wavefront polls exactly those entries that the sequential algorithm
//    would have filled before its own pixel (fill-order number smaller than its own) until their tag is set.  Every
//    pixel a wavefront can wait for has a smaller number, and the smallest unfinished pixel is always being worked on
//    by its owner without waiting: no deadlock, no barrier, and the cost of a dependency is one store -> load round
//    trip through the L2 instead of a workgroup-wide level barrier.
//  * otherwise (radius above kMaxLdsRange): the pixels of a component are grouped into dependency levels by the host
//    and a workgroup barrier separates the levels.
template <bool LDSWIN, bool NS, int RMAX = kMaxLdsRange, int NWAVES = kFillWaves>
__global__ __launch_bounds__(64 * NWAVES) void telea_fill_kernel(FillArgs a) {
    constexpr int kWinMax = (2 * RMAX + 3) * (2 * RMAX + 3), kTapMax = (2 * RMAX + 1) * (2 * RMAX + 1);
    __shared__ float s_terms[NWAVES][64][kAcc + 1];  // +1: odd stride, conflict-free column walks
    __shared__ int s_word[LDSWIN ? NWAVES : 1][LDSWIN ? kWinMax : 1];
    __shared__ float s_wt[LDSWIN ? NWAVES : 1][LDSWIN ? kWinMax : 1];
    __shared__ uint32_t s_wrgb[LDSWIN ? NWAVES : 1][LDSWIN ? kWinMax : 1];
    // Tile schedule: slot q - k0 - 1 receives the colour | tag of the pixel with fill-order number q the moment it is final; the
    // wavefronts of this workgroup that need it poll the slot -- an LDS round trip where the store -> L2 -> poll path costs
    // microseconds per link of the dependency chain.  Only pixels of this workgroup's tile are ever looked up here.
    __shared__ uint32_t s_slot[LDSWIN ? kFillSlots : 1];
    // Telea's distance factor of a tap, 1 / (|r|^2 |r|), only depends on the integer |r|^2 <= range^2: computed once per workgroup by
    // the same double-precision operations instead of a double square root and a double division per tap and pixel
    __shared__ float s_dst[LDSWIN && !NS ? RMAX * RMAX + 1 : 1];
    if (LDSWIN && !NS) {
        for (int e = threadIdx.x; e <= RMAX * RMAX; e += 64 * NWAVES) {
            const float vl = (float)e;
            s_dst[e] = e ? (float)(1. / (vl * sqrt((double)vl))) : 0.f;
        }
        __syncthreads();
    }
    const int ts = LDSWIN ? a.ts : 0;
    if (ts) {
        for (int e = threadIdx.x; e < kFillSlots; e += 64 * NWAVES) s_slot[e] = 0;
        __syncthreads();
    }
    const int ec = a.w + 2, er = a.h + 2, range = a.range;
    const int side = 2 * range + 1, ntap = side * side, ws = side + 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    // colour of padded pixel (r,c) as the sequential algorithm sees it while pixel number `o` is filled: the
    // filled value if it was filled earlier (agent-scope load: written by another wave, must not come from a
    // stale L1 line), the original otherwise.  Both loads are issued; `q` only selects.
    auto resolve = [&](int r, int c, int q, int o) -> uint32_t {
        const size_t at = (size_t)(r - 1) * a.w + (c - 1);
        const uint32_t orig = a.src[at];
        const uint32_t cur = __hip_atomic_load(a.out + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (q != 0 && q < o) ? cur : orig;
    };

    // n / d for the indices of a window (n < 2^16, d <= 255): (n + 0.5) / d is at least 0.5 / d away from an integer, far more than
    // the rounding of a float product -- an integer division by a run-time divisor is ~25 vector instructions, and a pixel had ten of them
    const float inv_ws = 1.0f / (float)ws, inv_side = 1.0f / (float)side;
    auto div_small = [](int n, float inv_d) __attribute__((always_inline)) { return (int)(((float)n + 0.5f) * inv_d); };
    // one pixel (padded index p, fill-order number o) by one wavefront
    auto fill_pixel = [&](const int p, const int o) {
            {
                {
                const int i = p / ec, j = p - i * ec;
                const int wi0 = i - range - 1, wj0 = j - range - 1;  // padded coordinates of the staged window's corner
                // measurement hook: lane 0 stamps the phases of this pixel (0 start, 1 maps staged + weights, 2 awaited colours there,
                // 3 colours staged, 4 terms in LDS, 5 sums done, 6 pixel stored; 7 = the largest order number it waited for)
                auto stamp = [&](int kx) __attribute__((always_inline)) {
                    if (a.trace && lane == 0) a.trace[(size_t)(o - 1) * 8 + kx] = __builtin_amdgcn_s_memtime();
                };
                stamp(0);
                auto ORD = [&](int r, int c) -> int { return LDSWIN ? s_word[wave][(r - wi0) * ws + (c - wj0)] : a.ord[r * ec + c]; };
                auto TT = [&](int r, int c) -> float { return LDSWIN ? s_wt[wave][(r - wi0) * ws + (c - wj0)] : a.t[r * ec + c]; };
                // What a window tap (one per lane and chunk of 64) needs of the maps alone: is it a tap at all, which neighbours
                // of it count as known, and -- Telea -- its weight.  With the window in LDS this runs before the polls.
                // bit 0: a tap at all; bits 1..4: right / left / down / up neighbour not known yet; bit 5: inside the circle of the window
                // (whatever the maps say), bits 8..13: its rank among the chunk's lanes inside the circle.  Only those taps can
                // contribute, and only they are summed: 29 of the 49 window positions at radius 3 -- the ordered sum is the longest
                // stretch of the pixel -> pixel chain, and skipping a term that is zero by construction does not change it.
                struct TapSetup {
                    unsigned bits;
                    float wgt;
                };
                constexpr int kChunks = LDSWIN ? (kTapMax + 63) / 64 : 1;
                TapSetup pre[kChunks];
                float Tij = 0.f, gTx = 0.f, gTy = 0.f;
                auto tap_setup = [&](int tap) -> TapSetup {  // (called by all lanes of the wavefront together: it takes a ballot)
                    TapSetup ts_;
                    ts_.bits = 0;
                    ts_.wgt = 0.f;
                    const int tq = div_small(tap, inv_side);
                    const int k = i - range + tq, l = j - range + (tap - tq * side);
                    const bool circ = tap < ntap && (l - j) * (l - j) + (k - i) * (k - i) <= range * range;
                    const unsigned long long cm = __builtin_amdgcn_ballot_w64(circ);
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                    if (!circ) return ts_;
                    ts_.bits = 32u | (rank << 8);
                    if (!(k > 0 && l > 0 && k < er - 1 && l < ec - 1 && ORD(k, l) < o)) return ts_;
                    ts_.bits |= 1u | (ORD(k, l + 1) >= o ? 2u : 0u) | (ORD(k, l - 1) >= o ? 4u : 0u) | (ORD(k + 1, l) >= o ? 8u : 0u) | (ORD(k - 1, l) >= o ? 16u : 0u);
                    if (!NS) {
                        const float ry = (float)(i - k), rx = (float)(j - l);
                        const float vl = rx * rx + ry * ry;
                        const float dst = LDSWIN ? s_dst[(int)vl] : (float)(1. / (vl * sqrt((double)vl)));
                        const float lev = (float)(1. / (1 + fabsf(TT(k, l) - Tij)));
                        float dir = rx * gTx + ry * gTy;
                        if (fabs(dir) <= 0.01) dir = 0.000001f;
                        ts_.wgt = (float)fabs(dst * lev * dir);
                    }
                    return ts_;
                };
                auto centre = [&]() {  // the pixel's own distance and the gradient of the distance map at it
                    Tij = TT(i, j);
                    const bool r_in = ORD(i, j + 1) >= o, l_in = ORD(i, j - 1) >= o, d_in = ORD(i + 1, j) >= o, u_in = ORD(i - 1, j) >= o;
                    if (!r_in) gTx = !l_in ? (TT(i, j + 1) - TT(i, j - 1)) * 0.5f : (TT(i, j + 1) - Tij);
                    else gTx = !l_in ? (Tij - TT(i, j - 1)) : 0.f;
                    if (!d_in) gTy = !u_in ? (TT(i + 1, j) - TT(i - 1, j)) * 0.5f : (TT(i + 1, j) - Tij);
                    else gTy = !u_in ? (Tij - TT(i - 1, j)) : 0.f;
                };
                auto pre_taps = [&]() {
                    centre();
#pragma unroll
                    for (int c = 0; c < kChunks; c++)
                        if (64 * c < ntap) pre[c] = tap_setup(64 * c + lane);
                };
                if (LDSWIN) {
                    // all loads that do not depend on other pixels' results first (maps, original colours), then the
                    // polls: the only thing on the dependency chain is the round trip of the awaited colours
                    constexpr int kEl = (kWinMax + 63) / 64;
                    int q[kEl];
                    float tv[kEl];
                    uint32_t rgb[kEl];
                    const uint32_t *wait_on[kEl];  // entries the sequential algorithm fills before pixel o: polled through the L2 ...
                    int wait_slot[kEl];            // ... or, inside this workgroup's tile, in their LDS slot
                    const int ti0 = ts ? (i / ts) * ts : 0, tj0 = ts ? (j / ts) * ts : 0;  // this pixel's tile (= this workgroup's): [ti0, ti0 + ts) x [tj0, tj0 + ts)
#pragma unroll
                    for (int u = 0; u < kEl; u++) {
                        const int e = lane + 64 * u;
                        q[u] = 0;
                        tv[u] = 0.f;
                        rgb[u] = 0;
                        wait_on[u] = nullptr;
                        wait_slot[u] = -1;
                        if (e < ws * ws) {
                            const int eq = div_small(e, inv_ws);
                            const int r = wi0 + eq, c = wj0 + (e - eq * ws);
                            if (r >= 0 && c >= 0 && r < er && c < ec) {
                                q[u] = a.ord[r * ec + c];
                                tv[u] = a.t[r * ec + c];
                                if (r >= 1 && c >= 1 && r <= a.h && c <= a.w) {
                                    const size_t at = (size_t)(r - 1) * a.w + (c - 1);
                                    if (q[u] != 0 && q[u] < o) {
                                        if (ts && q[u] <= a.k0) rgb[u] = a.out[at];  // filled by an earlier launch: final in memory
                                        else if (ts && (unsigned)(r - ti0) < (unsigned)ts && (unsigned)(c - tj0) < (unsigned)ts) wait_slot[u] = q[u] - a.k0 - 1;  // by this workgroup: its slot
                                        else wait_on[u] = a.out + at;               // by another workgroup of this launch
                                    } else {
                                        rgb[u] = a.src[at];
                                    }
                                }
                            }
                        }
                    }
                    // order numbers and distances do not depend on anybody's colour: staged now, so that everything the taps need of
                    // them (the weights with their two double divisions and the square root, Telea) is computed BEFORE the polls --
                    // off the dependency chain between a pixel and the pixels that wait for it
#pragma unroll
                    for (int u = 0; u < kEl; u++) {
                        const int e = lane + 64 * u;
                        if (e < ws * ws) {
                            s_word[wave][e] = q[u];
                            s_wt[wave][e] = tv[u];
                        }
                    }
                    wave_lds_sync();
                    pre_taps();
                    if (a.trace) {  // the latest pixel this one waits for
                        int wmax = 0;
#pragma unroll
                        for (int u = 0; u < kEl; u++)
                            if (wait_slot[u] >= 0 || wait_on[u]) wmax = max(wmax, q[u]);
                        for (int sh = 32; sh >= 1; sh >>= 1) wmax = max(wmax, __shfl_xor(wmax, sh));
                        if (lane == 0) a.trace[(size_t)(o - 1) * 8 + 7] = (unsigned long long)wmax;
                    }
                    stamp(1);
#pragma unroll
                    for (int u = 0; u < kEl; u++) {
                        int spins = 0;
                        while (wait_slot[u] >= 0) {
                            const uint32_t v = __hip_atomic_load(&s_slot[wait_slot[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (v >> 24) {
                                rgb[u] = v;
                                wait_slot[u] = -1;
                            } else if (++spins > a.spin_limit || ((spins & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                                __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // bounded like the L2 polls below
                                wait_slot[u] = -1;
                            } else {
                                __builtin_amdgcn_s_sleep(1);
                            }
                        }
                        while (wait_on[u]) {  // agent-scope load: straight from the L2, never a stale L1 line
                            const uint32_t v = __hip_atomic_load(wait_on[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (v >> 24) {
                                rgb[u] = v;
                                wait_on[u] = nullptr;
                            } else if (++spins > a.spin_limit || ((spins & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                                // Bounded wait.  The owner of the awaited pixel is a wavefront of another workgroup of this launch; should
                                // it not be running (a partner workgroup that never became resident), give up instead of
                                // spinning for ever inside the host's render thread: flag the launch, let every wavefront
                                // drain, and the host repeats the fill with the barrier-scheduled kernel.
                                __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                wait_on[u] = nullptr;
                            } else {
                                __builtin_amdgcn_s_sleep(1);
                            }
                        }
                    }
                    stamp(2);
                    // From here to the pixel's store a wavefront is on the pixel -> pixel chain; the other wavefronts of its SIMD are mostly
                    // staging maps and computing weights for pixels whose turn has not come: the chain goes first in the issue arbitration.
                    __builtin_amdgcn_s_setprio(3);
#pragma unroll
                    for (int u = 0; u < kEl; u++) {
                        const int e = lane + 64 * u;
                        if (e < ws * ws) s_wrgb[wave][e] = rgb[u];
                    }
                    wave_lds_sync();
                    stamp(3);
                }
                if (!LDSWIN) centre();
                // lanes 0..9: the sequential accumulator they own (s starts at 1e-20)
                float run = (NS ? (lane >= 3 && lane < 6) : lane == kAcc - 1) ? 1.0e-20f : 0.f;
                auto chunk = [&](const int t0, const TapSetup &tp) {
                    // phase 1: one window tap per lane
                    float term[kAcc];
#pragma unroll
                    for (int q = 0; q < kAcc; q++) term[q] = 0.f;
                    if (tp.bits & 1u) {
                        const int tap = t0 + lane;
                        const int tq = div_small(tap, inv_side);
                        const int k = i - range + tq, l = j - range + (tap - tq * side);
                        const int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2);
                        const int lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                        const float ry = (float)(i - k), rx = (float)(j - l);
                        const float vl = rx * rx + ry * ry;
                        const bool r_in = tp.bits & 2u, l_in = tp.bits & 4u, d_in = tp.bits & 8u, u_in = tp.bits & 16u;
                        // The tap's pixel and its six neighbours, one dword each (three channels per load), loaded whatever the flags say --
                        // they all lie inside the staged window -- and the gradient forms chosen by selects: the lanes of a wavefront
                        // have different flags, and as branches the four forms per direction ran one after the other (the phase trace
                        // had this stretch at 2 500 of the 4 300 cycles a pixel needs after its colours have arrived).
                        auto PIX = [&](int ir, int ic) -> uint32_t {
                            return LDSWIN ? s_wrgb[wave][(ir + 1 - wi0) * ws + (ic + 1 - wj0)] : resolve(ir + 1, ic + 1, a.ord[(ir + 1) * ec + ic + 1], o);
                        };
                        const uint32_t p_c = PIX(km, lm), p_xl = PIX(km, lm - 1), p_xr0 = PIX(km, lp), p_xr1 = PIX(km, lp + 1);
                        const uint32_t p_yu = PIX(km - 1, lm), p_yd0 = PIX(kp, lm), p_yd1 = PIX(kp + 1, lm);
                        if (NS) {
                            const float dst = 1 / (vl * vl + 1);
#pragma unroll
                            for (int ch = 0; ch < 3; ch++) {
                                auto CI = [&](uint32_t v) { return (int)((v >> (8 * ch)) & 255u); };
                                const int c = CI(p_c), xl = CI(p_xl), xr1 = CI(p_xr1), yu = CI(p_yu), yd0 = CI(p_yd0), yd1 = CI(p_yd1);
                                float gx = !d_in ? (!u_in ? (float)(abs(yd1 - yd0) + abs(yd0 - yu)) : (float)(abs(yd1 - yd0)) * 2.0f)
                                                 : (!u_in ? (float)(abs(yd0 - yu)) * 2.0f : 0.f);
                                const float gy = !r_in ? (!l_in ? (float)(abs(xr1 - c) + abs(c - xl)) : (float)(abs(xr1 - c)) * 2.0f)
                                                       : (!l_in ? (float)(abs(c - xl)) * 2.0f : 0.f);
                                gx = -gx;
                                float dir = rx * gx + ry * gy;
                                if (fabs(dir) <= 0.01) dir = 0.000001f;
                                else dir = (float)fabs((rx * gx + ry * gy) / sqrt((double)(vl * (gx * gx + gy * gy))));
                                const float wgt = dst * dir;
                                term[ch] = wgt * (float)c;
                                term[3 + ch] = wgt;
                            }
                        } else {
                            const float wgt = tp.wgt;
#pragma unroll
                            for (int ch = 0; ch < 3; ch++) {
                                auto CF = [&](uint32_t v) { return (float)((v >> (8 * ch)) & 255u); };
                                const float c = CF(p_c), xl = CF(p_xl), xr0 = CF(p_xr0), xr1 = CF(p_xr1), yu = CF(p_yu), yd0 = CF(p_yd0), yd1 = CF(p_yd1);
                                const float gIx = !r_in ? (!l_in ? (xr1 - xl) * 2.0f : (xr1 - c)) : (!l_in ? (xr0 - xl) : 0.f);
                                const float gIy = !d_in ? (!u_in ? (yd1 - yu) * 2.0f : (yd1 - c)) : (!u_in ? (yd0 - yu) : 0.f);
                                term[ch] = wgt * c;
                                term[3 + ch] = -(wgt * (gIx * rx));  // Jx, Jy are accumulated with -=: run - t is run + (-t) exactly, and the
                                term[6 + ch] = -(wgt * (gIy * ry));  // sign is applied here, by the tap's lane, not inside the ordered sum
                            }
                            term[9] = wgt;
                        }
                    }
                    // the terms of the taps inside the circle, compacted in tap order: row = rank; rows up to the next multiple of
                    // sixteen are zero (the ordered sum below takes whole batches of sixteen)
                    const int nc = __builtin_popcountll(__builtin_amdgcn_ballot_w64((tp.bits & 32u) != 0));
                    const int nrows = (nc + 15) & ~15;
                    if (tp.bits & 32u) {
                        const int row = (int)((tp.bits >> 8) & 63u);
#pragma unroll
                        for (int q = 0; q < kAcc; q++) s_terms[wave][row][q] = term[q];
                    }
                    if (lane >= nc && lane < nrows) {
#pragma unroll
                        for (int q = 0; q < kAcc; q++) s_terms[wave][lane][q] = 0.f;
                    }
                    wave_lds_sync();
                    stamp(4);
                    // phase 2: lane q adds this chunk's terms of accumulator q in tap (row-major) order
                    if (lane < kAcc) {
                        // The ordered adds are the chain; the LDS reads are not: batch b+1 is requested before batch b is added up.
                        // Rows nc .. nrows-1 hold zeros: whole batches, no per-row conditions.
                        constexpr int kB = 16;
                        const int nb = nrows / kB;
                        float v0[kB], v1[kB];
#pragma unroll
                        for (int u = 0; u < kB; u++) v0[u] = s_terms[wave][u][lane];
                        for (int b = 0; b < nb; b++) {
                            if (b + 1 < nb) {
#pragma unroll
                                for (int u = 0; u < kB; u++) v1[u] = s_terms[wave][(b + 1) * kB + u][lane];
                            }
#pragma unroll
                            for (int u = 0; u < kB; u++) run = run + v0[u];  // one dependent add per tap: the chain
#pragma unroll
                            for (int u = 0; u < kB; u++) v0[u] = v1[u];
                        }
                    }
                    wave_lds_sync();
                };
                if (LDSWIN) {
#pragma unroll
                    for (int c = 0; c < kChunks; c++)  // (constant indices into `pre`: it stays in registers)
                        if (64 * c < ntap) chunk(64 * c, pre[c]);
                } else {
                    for (int t0 = 0; t0 < ntap; t0 += 64) chunk(t0, tap_setup(t0 + lane));
                }
                // phase 3: lane ch finishes channel ch, lane 0 stores the pixel as one dword.  The sums travel between lanes by DPP row
                // shifts and readlane, the three bytes by readlane: this stretch is on the pixel -> pixel chain (it went through LDS
                // and ds_bpermute before).
                auto from_lane_plus = [&](float v, int d) {  // lane l <- lane l + d (d = 3 or 6, inside a row of 16)
                    const int iv = __builtin_bit_cast(int, v);
                    return __builtin_bit_cast(float, d == 3 ? __builtin_amdgcn_update_dpp(0, iv, 0x103, 0xf, 0xf, true)
                                                            : __builtin_amdgcn_update_dpp(0, iv, 0x106, 0xf, 0xf, true));
                };
                stamp(5);
                uint32_t byte = 0;
                if (NS) {
                    const float sw = from_lane_plus(run, 3);
                    if (lane < 3) {
                        int iv = (int)rint((double)run / sw);  // saturate_cast<uchar>(double)
                        byte = (uint32_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
                    }
                } else {
                    const float Jx = from_lane_plus(run, 3), Jy = from_lane_plus(run, 6);
                    const float sw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, run), 9));
                    if (lane < 3) {
                        const float Ia = run;
                        const float sat = (float)((Ia / sw + (Jx + Jy) / (sqrtf(Jx * Jx + Jy * Jy) + 1.0e-20f) + 0.5f));
                        int iv = (int)rintf(sat);  // cvRound, then saturate
                        byte = (uint32_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
                    }
                }
                const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)byte, 1), bl = (uint32_t)__builtin_amdgcn_readlane((int)byte, 2);
                if (lane == 0) {
                    const size_t at = (size_t)(i - 1) * a.w + (j - 1);
                    const uint32_t px = byte | (g << 8) | (bl << 16) | 0x01000000u;
                    if (ts) __hip_atomic_store(&s_slot[o - a.k0 - 1], px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // first: it is on the chain
                    __hip_atomic_store(a.out + at, px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (LDSWIN) __builtin_amdgcn_s_setprio(0);
                stamp(6);
                }
            }
    };

    if (LDSWIN) {
        // the wavefronts of all workgroups of a component take its pixels round-robin in fill order: the smallest unfinished
        // pixel is always the one its owner is working on (all smaller ones are finished), so the polls cannot deadlock
        // cmp_off holds one record per workgroup: {first pixel, end, this workgroup's first wavefront slot, slots in total}
        const int cbeg = a.cmp_off[4 * blockIdx.x], cend = a.cmp_off[4 * blockIdx.x + 1];
        const int first = a.cmp_off[4 * blockIdx.x + 2], stride = a.cmp_off[4 * blockIdx.x + 3];
        // (round 4/5 also had a dynamic form -- a wavefront takes the tile's next pixel from an LDS counter -- which bought nothing: removed in round 6)
        for (int id = cbeg + first + wave; id < cend; id += stride) fill_pixel(a.cmp_pix[id], a.cmp_ord[id]);
    } else {
        const int seg_beg = a.comp_off[blockIdx.x], seg_end = a.comp_off[blockIdx.x + 1];
        int beg = seg_beg < seg_end ? a.lvl_off[seg_beg] : 0;
        for (int seg = seg_beg; seg < seg_end; seg++) {
            const int end = a.lvl_off[seg + 1];
            for (int base = beg; base < end; base += NWAVES) {
                const int id = base + wave;
                if (id < end) fill_pixel(a.lvl_pix[id], a.lvl_ord[id]);  // wave-uniform
            }
            beg = end;
            // every colour of this level is written (one workgroup = one CU; later reads are agent-scope loads that
            // bypass the L1) before the next level of this component starts
            __threadfence_block();
            __syncthreads();
        }
    }
}

// render body: alpha forced to 255 (inpaint.cpp:349-352) on the device image, before it goes back to the host
__global__ __launch_bounds__(256) void opaque_alpha_kernel(uint32_t *__restrict__ img, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) img[i] |= 0xff000000u;
}

// ---- persistent device maps of the march (distance, order number): defaults everywhere, sparse updates per call
__global__ __launch_bounds__(256) void map_init_kernel(float *__restrict__ t, int *__restrict__ ord, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        t[i] = 1.0e6f;
        ord[i] = 0;
    }
}
__global__ __launch_bounds__(256) void map_scatter_kernel(const int *__restrict__ idx, const float *__restrict__ tv, const int *__restrict__ ov, int n,
                                                          float *__restrict__ t, int *__restrict__ ord) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        t[idx[i]] = tv[i];
        ord[idx[i]] = ov[i];
    }
}
// fill-order pixels k0+1 .. k0+n (their order numbers are consecutive)
__global__ __launch_bounds__(256) void map_scatter_front_kernel(const int *__restrict__ idx, const float *__restrict__ tv, int first_order, int n,
                                                                float *__restrict__ t, int *__restrict__ ord) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        t[idx[i]] = tv[i];
        ord[idx[i]] = first_order + i;
    }
}
__global__ __launch_bounds__(256) void map_reset_kernel(const int *__restrict__ idx, int n, float *__restrict__ t, int *__restrict__ ord) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        t[idx[i]] = 1.0e6f;
        ord[idx[i]] = 0;
    }
}
// padded order map -> width x height index map of the C ABI (1-based fill order, 0 = not filled)
__global__ __launch_bounds__(256) void order_map_kernel(const int *__restrict__ ord, int w, int h, int *__restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int o = ord[(size_t)(y + 1) * (w + 2) + x + 1];
    out[(size_t)y * w + x] = o == kNeverFilled ? 0 : o;
}

constexpr int kFillPortion = 8192;  // fill-order pixels per portion of the pipelined dataflow fill

// Termination of the dataflow fill.  A wavefront only ever waits for pixels of its own component with a smaller order
// number.  Components with ONE workgroup (up to 2000 pixels -- all of them for dust-like masks, however many
// components there are) therefore wait only inside their workgroup, whose wavefronts are co-resident by construction:
// the size of the grid does not matter for them.  A larger component is spread over 2..8 workgroups with consecutive
// ids; those do wait for each other, so a workgroup can spin while a partner is still in the dispatch queue.
// Workgroups are dispatched in id order, so per launch at most 7 resident workgroups can be in that state (the ones
// at the dispatch frontier); all others have every partner resident, run to completion and free their slots.  A
// device-wide stall would need every resident slot held by such frontier waiters, i.e. ~70 concurrent fills on 512
// slots; kMaxConcurrentFills bounds the fills of this process, and for everything that reasoning does not cover
// (other processes sharing the GPU, a dispatcher that does not go in order) the polls are BOUNDED: a wavefront that
// has spun kSpinLimit times raises FillArgs::err, every wavefront drains, and the host repeats the fill with the
// barrier-scheduled kernel (one workgroup per component, no cross-workgroup waits).
constexpr int kMaxConcurrentFills = 4;
constexpr int kSpinLimit = 1 << 21;  // ~2 s of polling one colour (a dependency normally resolves in microseconds)
class FillSlot {
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::condition_variable &cv() { static std::condition_variable c; return c; }
    static int &in_flight() { static int n = 0; return n; }
public:
    FillSlot() {
        std::unique_lock<std::mutex> lk(mu());
        cv().wait(lk, [] { return in_flight() < kMaxConcurrentFills; });
        ++in_flight();
    }
    ~FillSlot() {
        { std::lock_guard<std::mutex> lk(mu()); --in_flight(); }
        cv().notify_one();
    }
    static int active() {  // fills in flight right now, this one included
        std::lock_guard<std::mutex> lk(mu());
        return in_flight();
    }
    FillSlot(const FillSlot &) = delete;
    FillSlot &operator=(const FillSlot &) = delete;
};

int inpaint_mask_device(ofxcv_ctx *ctx, hipStream_t s, const uint8_t *d_rgba, ptrdiff_t row_bytes, int w, int h, int iters,
                        uint8_t *d_mask, ptrdiff_t mask_step, uint8_t *d_tmp) {
    dim3 block(256), grid(ofxcv_div_up(w, 256), h);
    uint8_t *first = iters > 0 ? d_tmp : d_mask;
    ptrdiff_t first_step = iters > 0 ? (ptrdiff_t)w : mask_step;
    hipLaunchKernelGGL(rgba_to_mask_kernel, grid, block, 0, s, d_rgba, row_bytes, w, h, first, first_step);
    OFXCV_LAUNCH_CHECK(ctx, "rgba_to_mask_kernel");
    if (iters > 0) {
        hipLaunchKernelGGL(dilate_rect_kernel, grid, block, 0, s, (const uint8_t *)d_tmp, (ptrdiff_t)w, w, h, iters, d_mask, mask_step);
        OFXCV_LAUNCH_CHECK(ctx, "dilate_rect_kernel");
    }
    return OFXCV_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int ofxcv_inpaint_mask(ofxcv_ctx *ctx, const uint8_t *d_rgba, ptrdiff_t row_bytes, int width, int height, int dilate_iters,
                       uint8_t *d_mask, ptrdiff_t mask_step, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_rgba || !d_mask || width <= 0 || height <= 0 || dilate_iters < 0 || mask_step < width || (((uintptr_t)d_rgba | (uintptr_t)row_bytes) & 3))
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "inpaint_mask: bad argument");
    int rc = ofxcv_reserve(ctx, ctx->ip_tmp, (size_t)width * height);
    if (rc) return rc;
    return inpaint_mask_device(ctx, ofxcv_stream(ctx, stream), d_rgba, row_bytes, width, height, dilate_iters, d_mask, mask_step,
                               (uint8_t *)ctx->ip_tmp.ptr);
}

int ofxcv_inpaint_telea(ofxcv_ctx *ctx, const uint8_t *d_src, ptrdiff_t src_step, int channels, const uint8_t *d_mask,
                        ptrdiff_t mask_step, int width, int height, double radius, uint8_t *d_dst, ptrdiff_t dst_step,
                        float *d_t_map, int *d_order_map, void *stream) {
    return ofxcv_inpaint(ctx, d_src, src_step, channels, d_mask, mask_step, width, height, radius, OFXCV_INPAINT_TELEA, d_dst, dst_step,
                         d_t_map, d_order_map, stream);
}

int ofxcv_inpaint(ofxcv_ctx *ctx, const uint8_t *d_src, ptrdiff_t src_step, int channels, const uint8_t *d_mask,
                  ptrdiff_t mask_step, int width, int height, double radius, int method, uint8_t *d_dst, ptrdiff_t dst_step,
                  float *d_t_map, int *d_order_map, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_src || !d_mask || !d_dst || width <= 0 || height <= 0 || (channels != 3 && channels != 4) || d_src == d_dst)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "inpaint: bad argument");
    if (method != OFXCV_INPAINT_NS && method != OFXCV_INPAINT_TELEA)
        return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "inpaint: method %d (CV_INPAINT_NS = 0, CV_INPAINT_TELEA = 1)", method);
    const bool ns = method == OFXCV_INPAINT_NS;
    hipStream_t s = ofxcv_stream(ctx, stream);
    int range = ofxcv_cv_round(radius);
    range = std::min(std::max(range, 1), 100);
    const int w = width, h = height, ec = w + 2, er = h + 2;
    const size_t en = (size_t)ec * er;
    static const bool trace = getenv("OFXCV_TRACE_INPAINT") != nullptr;  // debug: phase times on stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = trace ? now() : 0;

    // dst = src (cvCopy(input_img, output_img)); the mask comes back to the host for the front march
    OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, dst_step, d_src, src_step, (size_t)w * channels, h, hipMemcpyDeviceToDevice, s));
    std::vector<uint8_t> mask((size_t)w * h);
    OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(mask.data(), w, d_mask, mask_step, w, h, hipMemcpyDeviceToHost, s));

    // persistent device maps (distance, order): defaults everywhere, touched only at the hole, its band and the ring
    int rc;
    if (ctx->ip_map_w != w || ctx->ip_map_h != h || !ctx->ip_tmap.ptr) {
        rc = ofxcv_reserve(ctx, ctx->ip_tmap, en * 4);
        if (rc) return rc;
        rc = ofxcv_reserve(ctx, ctx->ip_omap, en * 4);
        if (rc) return rc;
        hipLaunchKernelGGL(map_init_kernel, dim3((unsigned)((en + 255) / 256)), dim3(256), 0, s, (float *)ctx->ip_tmap.ptr, (int *)ctx->ip_omap.ptr, en);
        OFXCV_LAUNCH_CHECK(ctx, "map_init_kernel");
        ctx->ip_map_w = w;
        ctx->ip_map_h = h;
    }
    float *d_t = (float *)ctx->ip_tmap.ptr;
    int *d_ord = (int *)ctx->ip_omap.ptr;
    // The device maps are only reset (sparsely, at the indices this call touches) on the normal ways out.  Any other return
    // below -- a failed reserve, a HIP error, a schedule that outgrows its scratch -- leaves them dirty: forget their size
    // then, so that the next call on this context starts from map_init_kernel instead of stale distances and order numbers.
    struct MapsGuard {
        ofxcv_ctx *c;
        bool clean = false;
        ~MapsGuard() {
            if (!clean) c->ip_map_w = c->ip_map_h = 0;
        }
    } maps_guard{ctx};
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));  // the mask is on the host

    if (!ctx->ip_host_state) {
        ctx->ip_host_state = new March();
        ctx->ip_host_state_free = [](void *p) { delete (March *)p; };
    }
    March &m = *(March *)ctx->ip_host_state;
    m.prepare(w, h, range);
    const bool dataflow = range <= kBigLdsRange;  // the window fits LDS: sixteen wavefronts per workgroup up to radius 5, eight up to 12
    const bool big_window = range > kMaxLdsRange;
    const int fill_waves = big_window ? kBigFillWaves : kFillWaves;
    const bool any = march_begin(mask.data(), !ns, m);
    // large holes in several pieces: the pieces' fronts are marched side by side on host threads and their pop sequences merged
    // into the exact sequential fill order (telea_march.h); march_advance() below hands that order out in the same portions
    const double t_par0 = trace ? now() : 0;
    if (any && ctx->ip_parallel_march) (void)march_parallel_run(m, ctx->ip_parallel_march > 1 ? ctx->ip_parallel_march : 8192);
    if (trace && m.par_on) fprintf(stderr, "ofxcv inpaint: %zu components marched side by side in %.2f ms\n", m.par->comps.size(), now() - t_par0);
    const double t_setup = trace ? now() : 0;

    // every index the march touches outside the inward front: holes (order "not yet"), band seeds and ring (their distances)
    const int n_holes = (int)m.holes.size(), n_static = n_holes + (int)m.seeds.size() + (int)m.ring_px.size();
    std::vector<int> &st_idx = m.up_idx, &st_ord = m.up_ord;
    std::vector<float> &st_t = m.up_t;
    st_idx.clear(); st_ord.clear(); st_t.clear();
    st_idx.reserve(n_static); st_ord.reserve(n_static); st_t.reserve(n_static);
    for (int p : m.holes) { st_idx.push_back(p); st_t.push_back(1.0e6f); st_ord.push_back(m.ord[p]); }
    for (int p : m.seeds) { st_idx.push_back(p); st_t.push_back(m.t[p]); st_ord.push_back(0); }
    for (int p : m.ring_px) { st_idx.push_back(p); st_t.push_back(m.t[p]); st_ord.push_back(0); }
    // device scratch: static entries, then per-pixel entries of the front (index, distance, order number), the schedule
    // arrays (pixels, order numbers: n_holes each) and the workgroup records (4 ints each, at most one per pixel)
    const size_t a256 = 256;
    const size_t off_si = 0, off_st = align_up(off_si + (size_t)n_static * 4, a256), off_so = align_up(off_st + (size_t)n_static * 4, a256),
                 off_fi = align_up(off_so + (size_t)n_static * 4, a256), off_ft = align_up(off_fi + (size_t)n_holes * 4, a256),
                 off_fo = align_up(off_ft + (size_t)n_holes * 4, a256), off_pix = align_up(off_fo + (size_t)n_holes * 4, a256),
                 off_po = align_up(off_pix + (size_t)n_holes * 4, a256), off_wg = align_up(off_po + (size_t)n_holes * 4, a256),
                 off_lo = align_up(off_wg + (size_t)n_holes * 16 + 16, a256), off_stream = align_up(off_lo + ((size_t)n_holes + 2) * 8 + 256, a256);
    // pipelined fill: everything a portion hands over (pixel indices, distances, schedule pixels, their order numbers, workgroup
    // records -- five arrays, each 16-byte aligned) is packed one portion after the other, so that a portion is ONE copy
    const int portion_px = ctx->ip_portion > 0 ? ctx->ip_portion : kFillPortion;
    const size_t stream_bytes = (size_t)n_holes * 32 + ((size_t)n_holes / portion_px + 2) * 5 * 16 + 256;
    const size_t total = align_up(off_stream + stream_bytes, a256);
    rc = ofxcv_reserve(ctx, ctx->ip_maps, total);
    if (rc) return rc;
    char *dp = (char *)ctx->ip_maps.ptr;
    // pinned host mirror of the per-pixel part of that scratch: the portions of the pipelined fill are handed over with
    // true asynchronous copies (a copy from pageable memory costs the host ~20 us and there are five per portion)
    if (ctx->ip_pinned_bytes < total) {
        std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        if (ctx->ip_pinned) {
            OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));
            (void)hipHostFree(ctx->ip_pinned);
            ctx->ip_pinned = nullptr;
            ctx->ip_pinned_bytes = 0;
        }
        OFXCV_HIP_CHECK(ctx, hipHostMalloc(&ctx->ip_pinned, total + (total >> 2), hipHostMallocDefault));
        ctx->ip_pinned_bytes = total + (total >> 2);
    }
    char *hp = (char *)ctx->ip_pinned;
    auto reset_maps = [&]() -> int {  // device maps back to their defaults (the host maps are reset at the next prepare())
        if (n_static) {
            hipLaunchKernelGGL(map_reset_kernel, dim3((n_static + 255) / 256), dim3(256), 0, s, (const int *)(dp + off_si), n_static, d_t, d_ord);
            OFXCV_LAUNCH_CHECK(ctx, "map_reset_kernel");
        }
        return OFXCV_OK;
    };
    if (n_static) {
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_si, st_idx.data(), (size_t)n_static * 4, hipMemcpyHostToDevice, s));
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_st, st_t.data(), (size_t)n_static * 4, hipMemcpyHostToDevice, s));
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_so, st_ord.data(), (size_t)n_static * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(map_scatter_kernel, dim3((n_static + 255) / 256), dim3(256), 0, s, (const int *)(dp + off_si), (const float *)(dp + off_st),
                           (const int *)(dp + off_so), n_static, d_t, d_ord);
        OFXCV_LAUNCH_CHECK(ctx, "map_scatter_kernel");
    }
    auto write_maps = [&]() -> int {  // optional parity outputs, from the device maps
        if (d_t_map) OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(d_t_map, d_t, en * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (d_order_map) {
            hipLaunchKernelGGL(order_map_kernel, dim3(ofxcv_div_up(w, 256), h), dim3(256), 0, s, (const int *)d_ord, w, h, d_order_map);
            OFXCV_LAUNCH_CHECK(ctx, "order_map_kernel");
        }
        return OFXCV_OK;
    };
    if (!any || n_holes == 0) {  // nothing to fill (no hole, or the reference's Out->Init failure)
        if ((rc = write_maps())) return rc;
        if ((rc = reset_maps())) return rc;
        OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));
        maps_guard.clean = true;
        return OFXCV_OK;
    }

    rc = ofxcv_reserve(ctx, ctx->ip_work, 2 * (size_t)w * h * 4);
    if (rc) return rc;
    uint32_t *work_src = (uint32_t *)ctx->ip_work.ptr, *work_out = work_src + (size_t)w * h;
    dim3 pblock(256), pgrid(ofxcv_div_up(w, 256), h);
    auto pack = [&]() {
        if (channels == 4) hipLaunchKernelGGL(pack_rgbx_kernel<4>, pgrid, pblock, 0, s, d_src, src_step, w, h, work_src, work_out);
        else hipLaunchKernelGGL(pack_rgbx_kernel<3>, pgrid, pblock, 0, s, d_src, src_step, w, h, work_src, work_out);
    };
    pack();
    OFXCV_LAUNCH_CHECK(ctx, "pack_rgbx_kernel");
    FillArgs fa;
    fa.t = d_t;
    fa.ord = d_ord;
    fa.src = work_src;
    fa.out = work_out;
    fa.w = w;
    fa.h = h;
    fa.range = range;
    fa.err = nullptr;
    fa.trace = nullptr;
    fa.spin_limit = 0;
    fa.k0 = 0;
    fa.ts = 0;
    fa.lvl_pix = fa.cmp_pix = (const int *)(dp + off_pix);
    fa.lvl_ord = fa.cmp_ord = (const int *)(dp + off_po);
    fa.lvl_off = fa.cmp_off = (const int *)(dp + off_wg);
    fa.comp_off = (const int *)(dp + off_lo);
    FillSlot slot;  // released when the call returns (after its last stream synchronisation)

    // uploads the distances / order numbers of fill-order pixels [k0, k1) into the device maps
    auto scatter_front = [&](int k0, int k1) -> int {
        const int n = k1 - k0;
        if (n <= 0) return OFXCV_OK;
        int *pi = (int *)(hp + off_fi) + k0;
        float *pt = (float *)(hp + off_ft) + k0;
        for (int k = 0; k < n; k++) {
            pi[k] = m.pix[k0 + k];
            pt[k] = m.t[m.pix[k0 + k]];
        }
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_fi + (size_t)k0 * 4, pi, (size_t)n * 4, hipMemcpyHostToDevice, s));
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_ft + (size_t)k0 * 4, pt, (size_t)n * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(map_scatter_front_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const int *)(dp + off_fi) + k0, (const float *)(dp + off_ft) + k0, k0 + 1, n,
                           d_t, d_ord);
        OFXCV_LAUNCH_CHECK(ctx, "map_scatter_front_kernel");
        return OFXCV_OK;
    };
    double t_march = 0, t_sched = 0;
    int launches = 0;
    bool timed_out = false;
    if (dataflow) {
        // The front is strictly sequential on the host; the colours are not.  The host moves the front on by a portion,
        // hands that portion's pixels to the GPU (their distances and order numbers into the device maps, their dataflow
        // schedule, one fill launch) and goes on marching while the GPU fills: the fill of a 1080p frame hides behind the march.
        rc = ofxcv_reserve(ctx, ctx->ip_flag, 256);
        if (rc) return rc;
        fa.err = (int *)ctx->ip_flag.ptr;
        static const char *trace_file = getenv("OFXCV_FILL_TRACE");  // measurement hook: phase stamps of every pixel of the fill into this file
        if (trace_file) {
            rc = ofxcv_reserve(ctx, ctx->ip_trace, (size_t)n_holes * 64);
            if (rc) return rc;
            OFXCV_HIP_CHECK(ctx, hipMemsetAsync(ctx->ip_trace.ptr, 0, (size_t)n_holes * 64, s));
            fa.trace = (unsigned long long *)ctx->ip_trace.ptr;
        }
        fa.spin_limit = ctx->ip_spin_limit >= 0 ? ctx->ip_spin_limit : kSpinLimit;
        OFXCV_HIP_CHECK(ctx, hipMemsetAsync(fa.err, 0, sizeof(int), s));
        std::vector<int> &sp = m.cmp_pix, &so = m.cmp_ord, &sw = m.cmp_wg;
        sp.clear(); so.clear(); sw.clear();
        // no reallocation while asynchronous copies from these arrays may be in flight
        sp.reserve(n_holes); so.reserve(n_holes); sw.reserve((size_t)n_holes * 4 + 4);
        m.pix.reserve(n_holes);
        const int portion = portion_px;
        size_t cursor = 0;  // into the stream area, the same on the pinned mirror and on the device
        char *const hs = hp + off_stream, *const ds = dp + off_stream;
        for (;;) {
            const double ta = trace ? now() : 0;
            const int k0 = (int)m.pix.size();
            const int got = march_advance(m, portion);
            const double tb = trace ? now() : 0;
            t_march += tb - ta;
            if (got <= 0) break;
            const int k1 = k0 + got;
            const size_t wg0 = sw.size() / 4;
            // tile schedule (default): a workgroup per occupied tile, hand-offs inside a tile through LDS; the component schedule
            // (1..8 workgroups per connected group of pixels, every hand-off through the L2) where the tiles do not fit
            int nwg = -1, ts = 0;
            // This call's share of the chip, re-read per portion: 192 resident workgroups over the fills in flight (four at most:
            // 48 each).  A fill that starts while another one's larger launch is still running finds some of its workgroups queued
            // behind it for the rest of that launch (a fraction of a millisecond: the older launch is resident and waits for nobody).
            const int max_tiles = ctx->ip_max_tiles > 0 ? ctx->ip_max_tiles : std::max(kMinTileGroups, kTileBudget / std::max(1, FillSlot::active()));
            if (ctx->ip_tiles && got <= kFillSlots) nwg = build_tile_portion(m, k0, k1, sp, so, sw, m.cell, ts, max_tiles, fill_waves);
            if (nwg < 0) {
                ts = 0;
                m.cell.clear();  // the tile pass leaves the grid in another geometry
                nwg = build_dataflow_portion(m, k0, k1, sp, so, sw, m.cell, m.stack, 256,
                                             8, fill_waves);
            }
            fa.k0 = k0;
            fa.ts = ts;
            // the portion's block: [pixel index | distance | schedule pixel | schedule order number] x got, [workgroup record] x nwg
            const size_t a16 = 16, q = align_up((size_t)got * 4, a16);
            const size_t b_idx = cursor, b_t = b_idx + q, b_pix = b_t + q, b_ord = b_pix + q, b_wg = b_ord + q, b_end = b_wg + align_up((size_t)nwg * 16, a16);
            if (b_end > stream_bytes) return ofxcv_fail(ctx, OFXCV_ERR_HIP, "inpaint: portion schedule larger than its scratch");
            {
                int *pi = (int *)(hs + b_idx);
                float *pt = (float *)(hs + b_t);
                const float *tm = m.t.data();
                for (int k = 0; k < got; k++) {
                    const int px = m.pix[k0 + k];
                    pi[k] = px;
                    pt[k] = tm[px];
                }
            }
            std::memcpy(hs + b_pix, sp.data() + k0, (size_t)got * 4);
            std::memcpy(hs + b_ord, so.data() + k0, (size_t)got * 4);
            std::memcpy(hs + b_wg, sw.data() + wg0 * 4, (size_t)nwg * 16);
            t_sched += trace ? now() - tb : 0;
            OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(ds + cursor, hs + cursor, b_end - cursor, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(map_scatter_front_kernel, dim3((got + 255) / 256), dim3(256), 0, s, (const int *)(ds + b_idx), (const float *)(ds + b_t), k0 + 1, got,
                               d_t, d_ord);
            OFXCV_LAUNCH_CHECK(ctx, "map_scatter_front_kernel");
            // the workgroup records address the schedule arrays of the whole call (entry k0 is this block's first)
            fa.lvl_pix = fa.cmp_pix = (const int *)((uintptr_t)(ds + b_pix) - (uintptr_t)k0 * 4);
            fa.lvl_ord = fa.cmp_ord = (const int *)((uintptr_t)(ds + b_ord) - (uintptr_t)k0 * 4);
            fa.lvl_off = fa.cmp_off = (const int *)(ds + b_wg);
            cursor = b_end;
            if (big_window) {
                if (ns) hipLaunchKernelGGL((telea_fill_kernel<true, true, kBigLdsRange, kBigFillWaves>), dim3(nwg), dim3(64 * kBigFillWaves), 0, s, fa);
                else hipLaunchKernelGGL((telea_fill_kernel<true, false, kBigLdsRange, kBigFillWaves>), dim3(nwg), dim3(64 * kBigFillWaves), 0, s, fa);
            } else if (ns) hipLaunchKernelGGL((telea_fill_kernel<true, true>), dim3(nwg), dim3(kFillThreads), 0, s, fa);
            else hipLaunchKernelGGL((telea_fill_kernel<true, false>), dim3(nwg), dim3(kFillThreads), 0, s, fa);
            OFXCV_LAUNCH_CHECK(ctx, "telea_fill_kernel");
            launches++;
        }
        int flag = 0;
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(&flag, fa.err, sizeof(int), hipMemcpyDeviceToHost, s));
        OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));
        timed_out = flag != 0;
        if (timed_out) ctx->ip_fallbacks++;
        if (fa.trace) {
            std::vector<unsigned long long> tr((size_t)n_holes * 8);
            OFXCV_HIP_CHECK(ctx, hipMemcpy(tr.data(), fa.trace, tr.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(trace_file, "wb")) {
                fwrite(tr.data(), 8, tr.size(), f);
                fclose(f);
            }
            fa.trace = nullptr;
        }
    } else {
        const double ta = trace ? now() : 0;
        m.pix.reserve(n_holes);
        while (march_advance(m, 1 << 30) > 0) {}
        t_march = trace ? now() - ta : 0;
        if ((rc = scatter_front(0, (int)m.pix.size()))) return rc;
    }
    const int n = (int)m.pix.size();
    if (n > 0 && (!dataflow || timed_out)) {
        // barrier-scheduled fill: dependency levels per component, one workgroup per component, nothing waits across
        // workgroups (large radii; and the repeat of a dataflow fill whose poll gave up -- same arithmetic, same colours)
        build_levels(m, false);
        if (timed_out) pack();  // working copy back to "nothing filled yet"
        if (m.lvl_off.size() > (size_t)n_holes + 2 || m.comp_off.size() > (size_t)n_holes + 2)
            return ofxcv_fail(ctx, OFXCV_ERR_HIP, "inpaint: level schedule larger than its scratch");
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_pix, m.lvl_pix.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_po, m.lvl_ord.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_wg, m.lvl_off.data(), m.lvl_off.size() * 4, hipMemcpyHostToDevice, s));
        OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(dp + off_lo, m.comp_off.data(), m.comp_off.size() * 4, hipMemcpyHostToDevice, s));
        fa.lvl_pix = fa.cmp_pix = (const int *)(dp + off_pix);  // (the pipelined fill had pointed these at its portions)
        fa.lvl_ord = fa.cmp_ord = (const int *)(dp + off_po);
        fa.lvl_off = fa.cmp_off = (const int *)(dp + off_wg);
        fa.err = nullptr;
        const int nwg = (int)m.comp_off.size() - 1;
        if (ns) hipLaunchKernelGGL((telea_fill_kernel<false, true>), dim3(nwg), dim3(kFillThreads), 0, s, fa);
        else hipLaunchKernelGGL((telea_fill_kernel<false, false>), dim3(nwg), dim3(kFillThreads), 0, s, fa);
        OFXCV_LAUNCH_CHECK(ctx, "telea_fill_kernel");
        launches++;
    }
    if (channels == 4)
        hipLaunchKernelGGL(unpack_rgbx_kernel<4>, pgrid, pblock, 0, s, (const uint32_t *)work_out, (const uint32_t *)work_src, w, h, d_dst, dst_step);
    else
        hipLaunchKernelGGL(unpack_rgbx_kernel<3>, pgrid, pblock, 0, s, (const uint32_t *)work_out, (const uint32_t *)work_src, w, h, d_dst, dst_step);
    OFXCV_LAUNCH_CHECK(ctx, "unpack_rgbx_kernel");
    if ((rc = write_maps())) return rc;
    if ((rc = reset_maps())) return rc;
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));  // the host arrays of the march must outlive the copies
    maps_guard.clean = true;
    if (trace)
        fprintf(stderr, "ofxcv inpaint: %d hole pixels, %d filled: set-up + ring %.2f ms, inward march %.2f ms, schedules %.2f ms, %d fill launches, total %.2f ms\n",
                n_holes, n, t_setup - t_begin, t_march, t_sched, launches, now() - t_begin);
    return OFXCV_OK;
}

int ofxcv_inpaint_render_host(ofxcv_ctx *ctx, const uint8_t *h_src, ptrdiff_t src_row_bytes, int width, int height, double radius,
                              double dilation, uint8_t *h_dst, ptrdiff_t dst_row_bytes, uint8_t *h_mask_out) {
    if (!ctx) return OFXCV_ERR_INVALID;
    if (!h_src || !h_dst || width <= 0 || height <= 0) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "inpaint_render_host: bad argument");
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
    hipStream_t s = ctx->compute;
    const int w = width, h = height;
    const size_t row = (size_t)w * 4, img = align_up(row * h, 256), msk = align_up((size_t)w * h, 256);
    int rc = ofxcv_reserve(ctx, ctx->ip_img, 2 * img + msk);
    if (rc) return rc;
    uint8_t *d_src = (uint8_t *)ctx->ip_img.ptr, *d_dst = d_src + img, *d_mask = d_dst + img;
    rc = ofxcv_upload_rows(ctx, d_src, row, h_src, src_row_bytes, h, s);
    if (rc) return rc;
    rc = ofxcv_inpaint_mask(ctx, d_src, (ptrdiff_t)row, w, h, dilation > 0 ? (int)dilation : 0, d_mask, w, s);
    if (rc) return rc;
    rc = ofxcv_inpaint_telea(ctx, d_src, (ptrdiff_t)row, 4, d_mask, w, w, h, radius, d_dst, (ptrdiff_t)row, nullptr, nullptr, s);
    if (rc) return rc;
    // write-back of inpaint.cpp:320-358 for noise == 0: RGB copied, alpha forced to 255 (the caller applies the
    // libc rand() noise of :336-347 itself when the noise parameter is non-zero, using h_mask_out)
    hipLaunchKernelGGL(opaque_alpha_kernel, dim3((unsigned)(((size_t)w * h + 255) / 256)), dim3(256), 0, s, (uint32_t *)d_dst, (size_t)w * h);
    OFXCV_LAUNCH_CHECK(ctx, "opaque_alpha_kernel");
    rc = ofxcv_download_rows(ctx, h_dst, dst_row_bytes, d_dst, row, h, s);
    if (rc) return rc;
    if (h_mask_out) OFXCV_HIP_CHECK(ctx, hipMemcpy2DAsync(h_mask_out, w, d_mask, w, w, h, hipMemcpyDeviceToHost, s));
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));
    return OFXCV_OK;
}

}  // extern "C"
