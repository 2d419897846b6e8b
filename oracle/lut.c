/*
 * lut.c -- CPU oracle for the two marshalling stages either side of Farneback:
 *   F0  GenericOpenCVPlugin::fetchCVImage8UGrayscale (OpenCV/GenericOpenCVPlugin.cpp:223-265)
 *       -> OFX::Color::Lut::to_byte_grayscale_nodither of openfx-supportext ofxsLut.h
 *   F7  the flow -> RGBA write-back loop (VectorGenerator/VectorGenerator.cpp:494-519)
 *
 * TEST INFRASTRUCTURE ONLY -- see ofxcv_oracle.h.  PARITY UNPINNED: SupportExt/ is an empty
 * submodule in the reference tree, so the sRGB LUT is restated from the published
 * openfx-supportext source: a 65536-entry table indexed by the high 16 bits of the float,
 * holding the sRGB-encoded value as 8.8 fixed point; byte = (v + 0x80) >> 8.
 * The luma weights are Rec.709 (ORC_LUMA_*), kept as named constants because the source that
 * fixes them is not in the tree.
 */
#include "ofxcv_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

#define ORC_LUMA_R 0.2126f
#define ORC_LUMA_G 0.7152f
#define ORC_LUMA_B 0.0722f
/* ... or Rec.601 (orc_set_luma(601)): which set supportext uses cannot be verified here, so it is a switch of the oracle and of the library
 * (option "lut.luma"), like the OpenCV generation switches of farneback.c */
#define ORC_LUMA601_R 0.299f
#define ORC_LUMA601_G 0.587f
#define ORC_LUMA601_B 0.114f
static int g_luma601 = 0;
void orc_set_luma(int standard) { g_luma601 = standard == 601; }
int orc_get_luma(void) { return g_luma601 ? 601 : 709; }

static float to_func_srgb(float v)
{
    if (v < 0.0031308f) return (v < 0.0f) ? 0.0f : v * 12.92f;
    return 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
}
static float from_func_srgb(float v)
{
    if (v < 0.04045f) return (v < 0.0f) ? 0.0f : v * (1.0f / 12.92f);
    return powf((v + 0.055f) * (1.0f / 1.055f), 2.4f);
}
static uint16_t hipart(float f)
{
    uint32_t i;
    memcpy(&i, &f, 4);
    return (uint16_t)(i >> 16);
}
static float index_to_float(uint16_t i)
{
    /* zeros and denormals -> 0; NaN/inf -> +-FLT_MAX; otherwise the mid-point of the bucket */
    if (i < 0x80 || (i >= 0x8000 && i < 0x8080)) return 0;
    if (i >= 0x7f80 && i < 0x8000) return FLT_MAX;
    if (i >= 0xff80) return -FLT_MAX;
    uint32_t u = ((uint32_t)i << 16) | 0x8000u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static int float_to_int_ff01(float value) /* Color::floatToInt<0xff01> */
{
    if (value <= 0) return 0;
    if (value >= 1.) return 0xff00;
    return (int)(value * 0xff00 + 0.5);
}

void orc_srgb_lut_build(uint16_t *lut)
{
    for (int i = 0; i < 0x10000; ++i) lut[i] = (uint16_t)float_to_int_ff01(to_func_srgb(index_to_float((uint16_t)i)));
    /* make toFunc(fromFunc(b)) the identity on bytes */
    for (int b = 0; b < 256; ++b) {
        float f = from_func_srgb(b / 255.0f);
        lut[hipart(f)] = (uint16_t)(b << 8);
    }
}

void orc_to_byte_grayscale(const float *src, ptrdiff_t src_row_bytes, int ncomp, int w, int h,
                           uint8_t *dst, ptrdiff_t dst_row_bytes)
{
    static uint16_t lut[0x10000];
    static int init = 0;
    if (!init) { orc_srgb_lut_build(lut); init = 1; }
    for (int y = 0; y < h; y++) {
        const float *s = (const float *)((const char *)src + y * src_row_bytes);
        uint8_t *d = dst + y * dst_row_bytes;
        for (int x = 0; x < w; x++, s += ncomp) {
            float l = g_luma601 ? ORC_LUMA601_R * s[0] + ORC_LUMA601_G * s[1] + ORC_LUMA601_B * s[2] : ORC_LUMA_R * s[0] + ORC_LUMA_G * s[1] + ORC_LUMA_B * s[2];
            d[x] = (uint8_t)((lut[hipart(l)] + 0x80) >> 8);
        }
    }
}

void orc_flow_to_rgba(const float *flow, int w, int h, float *dst, ptrdiff_t dst_row_bytes,
                      const int chan_u[4], const int chan_v[4], double rs_x, double rs_y)
{
    for (int y = 0; y < h; y++) {
        float *d = (float *)((char *)dst + y * dst_row_bytes);
        const float *s = flow + (size_t)y * w * 2;
        for (int x = 0; x < w; x++) {
            float u = (float)(s[x * 2] / rs_x), v = (float)(s[x * 2 + 1] / rs_y);
            for (int c = 0; c < 4; c++) {
                if (chan_u[c]) d[x * 4 + c] = u;
                if (chan_v[c]) d[x * 4 + c] = v; /* coord 1 is written after coord 0 (:508-516) */
            }
        }
    }
}
