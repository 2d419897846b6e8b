// inpaint.cpp -- OFX plugin "uk.org.bratwurstandhaggis:cvInpaint" v0.5, MI355X back-end.
//
// Drop-in for opencv2fx/inpaint/inpaint.cpp: same plugin identifier and version (:584-593), same describe()
// properties (:468-515), same clips and parameter set (:383-464), same action dispatch and error convention
// (:534-573).  The OpenCV calls of render() (:303-318) go to libofxcv_hip (ofxcv_inpaint_render_host); the
// noise / write-back loop (:320-358), which uses libc rand(), stays on the host exactly as in the reference.
#include <cstdlib>
#include <vector>

#include "ofxcv_plugin.h"

using namespace ofxcv_plugin;

#define INPAINT_RADIUS "threshold1"
#define DILATION "threshold2"
#define INPAINT_NOISE "inpaintnoise"
#define PLUGIN_GROUPING "Draw"  // opencv2fx.h:10

// the reference's property value (inpaint.cpp:40-67): one credit line, then the licence notice
static const char *kDescription = "OpenCV inpaint. Wrapper provided by Bernd Porr -- http://www.berndporr.me.uk\n\n" OFXCV_OFX_LICENCE_NOTICE;

namespace {

Suites g;

struct InstanceData {  // inpaint.cpp:107-112
    OfxParamHandle threshold1 = nullptr, threshold2 = nullptr, inpaintNoise = nullptr;
    int isGeneralEffect = 0;
};

InstanceData *instance_data(OfxImageEffectHandle effect) {  // :115-131
    OfxPropertySetHandle props = nullptr;
    check(g.effect->getPropertySet(effect, &props));
    InstanceData *d = nullptr;
    check(g.prop->propGetPointer(props, kOfxPropInstanceData, 0, (void **)&d));
    return d;
}

OfxStatus create_instance(OfxImageEffectHandle effect) {  // :135-174
    OfxPropertySetHandle props = nullptr;
    check(g.effect->getPropertySet(effect, &props));
    OfxParamSetHandle params = nullptr;
    check(g.effect->getParamSet(effect, &params));
    InstanceData *d = new InstanceData;
    char *context = nullptr;
    check(g.prop->propGetString(props, kOfxImageEffectPropContext, 0, &context));
    d->isGeneralEffect = context && std::strcmp(context, kOfxImageEffectContextGeneral) == 0;
    try {
        check(g.param->paramGetHandle(params, INPAINT_RADIUS, &d->threshold1, nullptr));
        check(g.param->paramGetHandle(params, DILATION, &d->threshold2, nullptr));
        check(g.param->paramGetHandle(params, INPAINT_NOISE, &d->inpaintNoise, nullptr));
        check(g.prop->propSetPointer(props, kOfxPropInstanceData, 0, d));
    } catch (...) {
        delete d;
        throw;
    }
    return kOfxStatOK;
}

OfxStatus destroy_instance(OfxImageEffectHandle effect) {  // :178-190
    delete instance_data(effect);
    return kOfxStatOK;
}

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

OfxStatus render(OfxImageEffectHandle instance, OfxPropertySetHandle inArgs, OfxPropertySetHandle) {  // :209-376
    InstanceData *d = instance_data(instance);
    OfxTime time;
    OfxRectI rw;
    check(g.prop->propGetDouble(inArgs, kOfxPropTime, 0, &time));
    check(g.prop->propGetIntN(inArgs, kOfxImageEffectPropRenderWindow, 4, &rw.x1));

    OfxImageClipHandle outClip = nullptr, srcClip = nullptr;
    check(g.effect->clipGetHandle(instance, kOfxImageEffectOutputClipName, &outClip, nullptr));
    ImageGuard out(g, outClip, time);
    check(g.effect->clipGetHandle(instance, kOfxImageEffectSimpleSourceClipName, &srcClip, nullptr));
    ImageGuard src(g, srcClip, time);
    if (out.img.depth != kOfxBitDepthByte || src.img.depth != kOfxBitDepthByte) return kOfxStatErrImageFormat;

    double t1, t2, ng;
    check(g.param->paramGetValueAtTime(d->threshold1, time, &t1));
    check(g.param->paramGetValueAtTime(d->threshold2, time, &t2));
    check(g.param->paramGetValueAtTime(d->inpaintNoise, time, &ng));

    const int w = src.img.width(), h = src.img.height();
    if (w <= 0 || h <= 0) return kOfxStatFailed;
    // image1 (RGBA, alpha 255) and the dilated mask, computed on the GPU (:303-318)
    std::vector<unsigned char> image1((size_t)w * h * 4), mask((size_t)w * h);
    ThreadContext::Lease ctx = ThreadContext::get();
    check_hip(ctx, ofxcv_inpaint_render_host(ctx, (const uint8_t *)src.img.data, src.img.row_bytes, w, h, t1, t2, image1.data(),
                                             (ptrdiff_t)w * 4, mask.data()));

    // write-back with optional noise (:320-358), same loop structure and libc rand() sequence
    const int noise_flag = ng > 0;
    int noise_div = 1000;
    if (noise_flag) noise_div = (int)(1 / ng);
    if (noise_div == 0) noise_div = 1;
    int rs = 0;
    for (int y = rw.y1; y < rw.y1 + h; y++) {
        if (g.effect->abort(instance)) break;
        if (y < out.img.bounds.y1 || y >= out.img.bounds.y2 || y - rw.y1 >= h || y < 0 || y >= h) continue;
        OfxRGBAColourB *dstPix = (OfxRGBAColourB *)((char *)out.img.data + (ptrdiff_t)(y - out.img.bounds.y1) * out.img.row_bytes) +
                                 (rw.x1 - out.img.bounds.x1);
        const unsigned char *srcPix = image1.data() + ((size_t)y * w + rw.x1) * 4;
        const unsigned char *maskP = mask.data() + (size_t)y * w + rw.x1;
        for (int x = rw.x1; x < rw.x1 + w && x < w; x++) {
            if ((y % 4) == 0) srand(rs);
            int a = 0;
            if (noise_flag && maskP[0] > 0 && (x % 4) == 0) {
                a = ((rand() % 10) - 5) / noise_div;
                rs = (rs + srcPix[0]) % 256;
            }
            if (x >= out.img.bounds.x1 && x < out.img.bounds.x2) {
                dstPix->r = (unsigned char)clampi(srcPix[0] + a, 0, 255);
                dstPix->g = (unsigned char)clampi(srcPix[1] + a, 0, 255);
                dstPix->b = (unsigned char)clampi(srcPix[2] + a, 0, 255);
                dstPix->a = 255;
            }
            dstPix++;
            srcPix += 4;
            maskP++;
        }
    }
    return kOfxStatOK;
}

OfxStatus describe_in_context(OfxImageEffectHandle effect, OfxPropertySetHandle) {  // :383-464
    OfxPropertySetHandle props = nullptr;
    check(g.effect->clipDefine(effect, kOfxImageEffectOutputClipName, &props));
    check(g.prop->propSetString(props, kOfxImageEffectPropSupportedComponents, 0, kOfxImageComponentRGBA));
    check(g.effect->clipDefine(effect, kOfxImageEffectSimpleSourceClipName, &props));
    check(g.prop->propSetString(props, kOfxImageEffectPropSupportedComponents, 0, kOfxImageComponentRGBA));

    OfxStatus st = g.fetch_all();
    if (st != kOfxStatOK) return st;

    OfxParamSetHandle params = nullptr;
    check(g.effect->getParamSet(effect, &params));
    define_double_param(g, params, INPAINT_RADIUS, "Radius", "Sets the inpaint radius", 1, 10, 3);
    define_double_param(g, params, DILATION, "Dilation", "Sets the size of the boundary of intact pixels taken for the inpainting", 1, 5, 1);
    define_double_param(g, params, INPAINT_NOISE, "Inpaint noise", "Sets additional noise to fake camera noise", 0, 1, 0);

    check(g.param->paramDefine(params, kOfxParamTypePage, "Main", &props));
    check(g.prop->propSetString(props, kOfxParamPropPageChild, 0, INPAINT_RADIUS));
    check(g.prop->propSetString(props, kOfxParamPropPageChild, 1, DILATION));
    return kOfxStatOK;
}

OfxStatus describe(OfxImageEffectHandle effect) {  // :468-515
    OfxPropertySetHandle p = nullptr;
    check(g.effect->getPropertySet(effect, &p));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsMultipleClipDepths, 0, 0));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedPixelDepths, 0, kOfxBitDepthByte));
    check(g.prop->propSetString(p, kOfxPropLabel, 0, "openCV Inpaint"));
    check(g.prop->propSetString(p, kOfxImageEffectPluginPropGrouping, 0, PLUGIN_GROUPING));
    check(g.prop->propSetString(p, kOfxPropPluginDescription, 0, kDescription));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedContexts, 0, kOfxImageEffectContextFilter));
    check(g.prop->propSetInt(p, kOfxImageEffectPluginPropSingleInstance, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPluginPropHostFrameThreading, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsMultiResolution, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsTiles, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPropTemporalClipAccess, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPluginPropFieldRenderTwiceAlways, 0, 1));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsMultipleClipPARs, 0, 0));
    return kOfxStatOK;
}

OfxStatus plugin_main(const char *action, const void *handle, OfxPropertySetHandle inArgs, OfxPropertySetHandle outArgs) {  // :534-573
    return guarded([&]() -> OfxStatus {
        OfxImageEffectHandle effect = (OfxImageEffectHandle)handle;
        if (!std::strcmp(action, kOfxActionLoad)) return g.fetch_basic();
        if (!std::strcmp(action, "OfxActionUnload")) {  // not handled by the reference (reply default); device contexts are released
            ThreadContext::release_all();
            return kOfxStatReplyDefault;
        }
        if (!std::strcmp(action, kOfxActionDescribe)) return describe(effect);
        if (!std::strcmp(action, kOfxImageEffectActionDescribeInContext)) return describe_in_context(effect, inArgs);
        if (!std::strcmp(action, kOfxImageEffectActionRender)) return render(effect, inArgs, outArgs);
        if (!std::strcmp(action, kOfxActionCreateInstance)) return create_instance(effect);
        if (!std::strcmp(action, kOfxActionDestroyInstance)) return destroy_instance(effect);
        return kOfxStatReplyDefault;
    });
}

void set_host(OfxHost *h) { g.host = h; }

OfxPlugin plugin = {kOfxImageEffectPluginApi, 1, "uk.org.bratwurstandhaggis:cvInpaint", 0, 5, set_host, plugin_main};

}  // namespace

extern "C" {
OfxExport OfxPlugin *OfxGetPlugin(int nth) { return nth == 0 ? &plugin : nullptr; }
OfxExport int OfxGetNumberOfPlugins(void) { return 1; }
}
