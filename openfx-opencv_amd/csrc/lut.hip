// lut.hip -- F0 (f32 linear RGB(A) -> 8-bit sRGB luma) and F7 (flow -> RGBA write-back).
//
// F0 replaces OFX::Color::Lut::to_byte_grayscale_nodither (openfx-supportext ofxsLut.h) as called
// from GenericOpenCVPlugin::fetchCVImage8UGrayscale (OpenCV/GenericOpenCVPlugin.cpp:223-265).
// The LUT is the supportext one: 65536 entries indexed by the high half of the float bit
// pattern, each holding the sRGB-encoded value in 8.8 fixed point; byte = (v + 0x80) >> 8.
// It is built once on the host (powf) and lives in HBM (128 KiB, L2-resident).
//
// F7 replaces the write-back loop of VectorGenerator/VectorGenerator.cpp:494-519.
// Both kernels are pure streaming: one coalesced 16-byte load or store per lane.
#include <cfloat>
#include <cmath>
#include <vector>

#include "common.h"

namespace {

// Rec.709 luma weights of the supportext grayscale conversion.
constexpr float kLumaR = 0.2126f, kLumaG = 0.7152f, kLumaB = 0.0722f;
// ... and Rec.601 (option "lut.luma" 601): openfx-supportext's source is not in the reference tree (empty submodule), so which set its
// to_byte_grayscale_nodither uses cannot be verified here ("make it one named constant") -- both are a switch, like the OpenCV generations
constexpr float kLuma601R = 0.299f, kLuma601G = 0.587f, kLuma601B = 0.114f;

float srgb_encode(float v) {
    if (v < 0.0031308f) return (v < 0.0f) ? 0.0f : v * 12.92f;
    return 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
}
float srgb_decode(float v) {
    if (v < 0.04045f) return (v < 0.0f) ? 0.0f : v * (1.0f / 12.92f);
    return std::pow((v + 0.055f) * (1.0f / 1.055f), 2.4f);
}
uint16_t high_half(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t)(u >> 16);
}
float bucket_midpoint(uint16_t i) {
    if (i < 0x80 || (i >= 0x8000 && i < 0x8080)) return 0.f;  // zeros, denormals
    if (i >= 0x7f80 && i < 0x8000) return FLT_MAX;            // +inf / NaN
    if (i >= 0xff80) return -FLT_MAX;                         // -inf / NaN
    uint32_t u = ((uint32_t)i << 16) | 0x8000u;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
int to_fixed_8_8(float v) {
    if (v <= 0) return 0;
    if (v >= 1.) return 0xff00;
    return (int)(v * 0xff00 + 0.5);
}

int ensure_lut(ofxcv_ctx *ctx, hipStream_t s) {
    if (ctx->d_srgb_lut) return OFXCV_OK;
    std::vector<uint16_t> lut(0x10000);
    for (int i = 0; i < 0x10000; ++i) lut[i] = (uint16_t)to_fixed_8_8(srgb_encode(bucket_midpoint((uint16_t)i)));
    for (int b = 0; b < 256; ++b) lut[high_half(srgb_decode(b / 255.0f))] = (uint16_t)(b << 8);
    {
        std::unique_lock<std::shared_mutex> lock(ofxcv_capture_mutex(ofxcv_lock_index(ctx)));
        OFXCV_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_srgb_lut, lut.size() * sizeof(uint16_t)));
    }
    OFXCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_srgb_lut, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    OFXCV_HIP_CHECK(ctx, hipStreamSynchronize(s));  // `lut` is pageable and dies with this scope
    return OFXCV_OK;
}

// (the table pointer's lowest bit carries the luma choice: the table is 2-byte aligned data, bit 0 of its address is free)
__device__ __forceinline__ uint8_t lut_byte(const uint16_t *__restrict__ lut_tagged, float r, float g, float b) {
    const bool r601 = ((uintptr_t)lut_tagged & 1u) != 0;
    const uint16_t *__restrict__ lut = (const uint16_t *)((uintptr_t)lut_tagged & ~(uintptr_t)1);
    float l = r601 ? kLuma601R * r + kLuma601G * g + kLuma601B * b : kLumaR * r + kLumaG * g + kLumaB * b;
    return (uint8_t)((lut[__float_as_uint(l) >> 16] + 0x80) >> 8);
}

// One pixel per lane and instruction: consecutive lanes read consecutive RGBA pixels (a coalesced 1 KiB request per
// wavefront for RGBA), and the gray bytes of a wavefront leave as one 64-byte row segment.  A lane can walk NSEG such
// segments, 256 pixels apart, with the loads of all of them in flight together; NSEG = 1 measured best.
template <int NCOMP, int NSEG>
__global__ __launch_bounds__(256) void gray_lut_kernel(const float *__restrict__ src, ptrdiff_t src_row_bytes,
                                                       int width, int height, uint8_t *__restrict__ dst,
                                                       ptrdiff_t dst_row_bytes, const uint16_t *__restrict__ lut) {
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * (256 * NSEG) + threadIdx.x;
    const float *srow = (const float *)((const char *)src + (ptrdiff_t)y * src_row_bytes);
    uint8_t *drow = dst + (ptrdiff_t)y * dst_row_bytes;
    const bool vec = NCOMP == 4 && (((uintptr_t)srow) & 15) == 0;
    float r[NSEG], g[NSEG], b[NSEG];
#pragma unroll
    for (int i = 0; i < NSEG; i++) {
        const int x = x0 + i * 256;
        r[i] = g[i] = b[i] = 0.f;
        if (x < width) {
            if (vec) {
                float4 p = *(const float4 *)(srow + (size_t)x * 4);
                r[i] = p.x; g[i] = p.y; b[i] = p.z;
            } else {
                r[i] = srow[(size_t)x * NCOMP]; g[i] = srow[(size_t)x * NCOMP + 1]; b[i] = srow[(size_t)x * NCOMP + 2];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NSEG; i++) {
        const int x = x0 + i * 256;
        if (x < width) drow[x] = lut_byte(lut, r[i], g[i], b[i]);
    }
}

// Four consecutive RGBA pixels per lane: four independent 16-byte loads in flight, the four gray bytes leave as one dword
// (a wavefront stores 256 contiguous bytes).  Needs 16-byte aligned source rows, a 4-byte aligned destination and a width
// that is a multiple of 4; everything else takes gray_lut_kernel.
__global__ __launch_bounds__(256) void gray_lut4_kernel(const float *__restrict__ src, ptrdiff_t src_row_bytes, int width, int height,
                                                        uint8_t *__restrict__ dst, ptrdiff_t dst_row_bytes, const uint16_t *__restrict__ lut) {
    const int y = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;  // group of four pixels
    if (q * 4 >= width) return;
    const float4 *srow = (const float4 *)((const char *)src + (ptrdiff_t)y * src_row_bytes) + (size_t)q * 4;
    const float4 p0 = srow[0], p1 = srow[1], p2 = srow[2], p3 = srow[3];
    const unsigned b0 = lut_byte(lut, p0.x, p0.y, p0.z), b1 = lut_byte(lut, p1.x, p1.y, p1.z), b2 = lut_byte(lut, p2.x, p2.y, p2.z),
                   b3 = lut_byte(lut, p3.x, p3.y, p3.z);
    *(unsigned *)(dst + (ptrdiff_t)y * dst_row_bytes + (size_t)q * 4) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

// The same for the frames of a batched call (grid z = frame): one launch for the 2n frames of n pairs.
constexpr int kLutBatch = 2 * OFXCV_FB_MAX_BATCH;
struct LutTab {
    const float *src[kLutBatch];
    uint8_t *dst[kLutBatch];
    ptrdiff_t src_rb[kLutBatch], dst_rb[kLutBatch];
};
__global__ __launch_bounds__(256) void gray_lut4_batch_kernel(LutTab t, int width, int height, const uint16_t *__restrict__ lut) {
    const int y = blockIdx.y, z = blockIdx.z;
    const int q = blockIdx.x * 256 + threadIdx.x;  // group of four pixels
    if (q * 4 >= width) return;
    const float4 *srow = (const float4 *)((const char *)t.src[z] + (ptrdiff_t)y * t.src_rb[z]) + (size_t)q * 4;
    const float4 p0 = srow[0], p1 = srow[1], p2 = srow[2], p3 = srow[3];
    const unsigned b0 = lut_byte(lut, p0.x, p0.y, p0.z), b1 = lut_byte(lut, p1.x, p1.y, p1.z), b2 = lut_byte(lut, p2.x, p2.y, p2.z),
                   b3 = lut_byte(lut, p3.x, p3.y, p3.z);
    *(unsigned *)(t.dst[z] + (ptrdiff_t)y * t.dst_rb[z] + (size_t)q * 4) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

__global__ __launch_bounds__(256) void flow_to_rgba_kernel(const float *__restrict__ flow, size_t flow_step, int width,
                                                           int height, float *__restrict__ dst, ptrdiff_t dst_row_bytes,
                                                           unsigned mu, unsigned mv, double rsx, double rsy) {
    int y = blockIdx.y;
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    const float2 f = *(const float2 *)((const char *)flow + (size_t)y * flow_step + (size_t)x * 8);
    float u = (float)(f.x / rsx), v = (float)(f.y / rsy);
    float *d = (float *)((char *)dst + (ptrdiff_t)y * dst_row_bytes) + (size_t)x * 4;
    if (((mu | mv) & 15u) == 15u && (((uintptr_t)d) & 15) == 0) {
        float4 o;
        o.x = (mv & 1u) ? v : u;
        o.y = (mv & 2u) ? v : u;
        o.z = (mv & 4u) ? v : u;
        o.w = (mv & 8u) ? v : u;
        *(float4 *)d = o;
    } else {
        for (int c = 0; c < 4; c++) {
            if (mv & (1u << c)) d[c] = v;  // coord 1 is written after coord 0 in the reference loop
            else if (mu & (1u << c)) d[c] = u;
        }
    }
}

}  // namespace

int ofxcv_launch_flow_to_rgba(ofxcv_ctx *ctx, hipStream_t s, const float *d_flow, size_t flow_step, int width, int height, float *d_dst,
                              ptrdiff_t dst_row_bytes, unsigned chan_u_mask, unsigned chan_v_mask, double render_scale_x, double render_scale_y) {
    dim3 block(256), grid(ofxcv_div_up(width, 256), height);
    hipLaunchKernelGGL(flow_to_rgba_kernel, grid, block, 0, s, d_flow, flow_step, width, height, d_dst, dst_row_bytes,
                       chan_u_mask & 15u, chan_v_mask & 15u, render_scale_x, render_scale_y);
    OFXCV_LAUNCH_CHECK(ctx, "flow_to_rgba_kernel");
    return OFXCV_OK;
}

extern "C" {

int ofxcv_to_byte_grayscale(ofxcv_ctx *ctx, const float *d_src, ptrdiff_t src_row_bytes, int ncomp, int width,
                            int height, uint8_t *d_dst, ptrdiff_t dst_row_bytes, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_src || !d_dst || width <= 0 || height <= 0) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "to_byte_grayscale: bad argument");
    if (ncomp != 3 && ncomp != 4) return ofxcv_fail(ctx, OFXCV_ERR_UNSUPPORTED, "to_byte_grayscale: RGB or RGBA only");
    hipStream_t s = ofxcv_stream(ctx, stream);
    int rc = ensure_lut(ctx, s);
    if (rc) return rc;
    // one pixel per thread measured best (9.6 us per 1080p RGBA frame; 2 / 4 / 8 pixels per thread: 10.4 / 11.6 / 10.8 us)
    constexpr int kSeg = 1;
    const uint16_t *lut_arg = (const uint16_t *)((uintptr_t)ctx->d_srgb_lut | (ctx->lut_luma601 ? 1u : 0u));  // (bit 0: Rec.601 luma weights, lut_byte)
    dim3 block(256), grid(ofxcv_div_up(width, 256 * kSeg), height);
    if (ncomp == 4 && !(width & 3) && !(((uintptr_t)d_src) & 15) && !(src_row_bytes & 15) && !(((uintptr_t)d_dst) & 3) && !(dst_row_bytes & 3))
        hipLaunchKernelGGL(gray_lut4_kernel, dim3(ofxcv_div_up(width / 4, 256), height), block, 0, s, d_src, src_row_bytes, width, height, d_dst, dst_row_bytes,
                           lut_arg);
    else if (ncomp == 4)
        hipLaunchKernelGGL((gray_lut_kernel<4, kSeg>), grid, block, 0, s, d_src, src_row_bytes, width, height, d_dst, dst_row_bytes, lut_arg);
    else
        hipLaunchKernelGGL((gray_lut_kernel<3, kSeg>), grid, block, 0, s, d_src, src_row_bytes, width, height, d_dst, dst_row_bytes, lut_arg);
    OFXCV_LAUNCH_CHECK(ctx, "gray_lut_kernel");
    return OFXCV_OK;
}

int ofxcv_to_byte_grayscale_batch(ofxcv_ctx *ctx, int n, const float *const *d_src, const ptrdiff_t *src_row_bytes, int ncomp, int width,
                                  int height, uint8_t *const *d_dst, const ptrdiff_t *dst_row_bytes, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    if (n <= 0 || !d_src || !src_row_bytes || !d_dst || !dst_row_bytes) return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "to_byte_grayscale_batch: bad argument");
    bool fast = ncomp == 4 && n <= kLutBatch && width > 0 && height > 0 && !(width & 3);
    for (int i = 0; fast && i < n; i++)
        fast = d_src[i] && d_dst[i] && !(((uintptr_t)d_src[i]) & 15) && !(src_row_bytes[i] & 15) && !(((uintptr_t)d_dst[i]) & 3) && !(dst_row_bytes[i] & 3);
    if (!fast) {  // whatever the four-pixel kernel cannot take (RGB, odd widths, unaligned rows), and every argument check: frame by frame
        for (int i = 0; i < n; i++) {
            int rc = ofxcv_to_byte_grayscale(ctx, d_src[i], src_row_bytes[i], ncomp, width, height, d_dst[i], dst_row_bytes[i], stream);
            if (rc) return rc;
        }
        return OFXCV_OK;
    }
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));
    hipStream_t s = ofxcv_stream(ctx, stream);
    int rc = ensure_lut(ctx, s);
    if (rc) return rc;
    LutTab t = {};
    for (int i = 0; i < n; i++) {
        t.src[i] = d_src[i];
        t.dst[i] = d_dst[i];
        t.src_rb[i] = src_row_bytes[i];
        t.dst_rb[i] = dst_row_bytes[i];
    }
    hipLaunchKernelGGL(gray_lut4_batch_kernel, dim3(ofxcv_div_up(width / 4, 256), height, n), dim3(256), 0, s, t, width, height, (const uint16_t *)((uintptr_t)ctx->d_srgb_lut | (ctx->lut_luma601 ? 1u : 0u)));
    OFXCV_LAUNCH_CHECK(ctx, "gray_lut4_batch_kernel");
    return OFXCV_OK;
}

int ofxcv_flow_to_rgba(ofxcv_ctx *ctx, const float *d_flow, size_t flow_step, int width, int height, float *d_dst,
                       ptrdiff_t dst_row_bytes, unsigned chan_u_mask, unsigned chan_v_mask, double render_scale_x,
                       double render_scale_y, void *stream) {
    if (!ctx) return OFXCV_ERR_INVALID;
    OFXCV_HIP_CHECK(ctx, hipSetDevice(ctx->hip_device));  // a thread may hold contexts on several devices
    if (!d_flow || !d_dst || width <= 0 || height <= 0 || (flow_step & 7) || render_scale_x == 0 || render_scale_y == 0)
        return ofxcv_fail(ctx, OFXCV_ERR_INVALID, "flow_to_rgba: bad argument");
    return ofxcv_launch_flow_to_rgba(ctx, ofxcv_stream(ctx, stream), d_flow, flow_step, width, height, d_dst, dst_row_bytes, chan_u_mask, chan_v_mask,
                                     render_scale_x, render_scale_y);
}

}  // extern "C"
