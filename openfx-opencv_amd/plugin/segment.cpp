// segment.cpp -- OFX plugin "uk.org.bratwurstandhaggis:cvPyrSegmentation" v0.5, MI355X back-end.
//
// Drop-in for opencv2fx/segment/segment.cpp: same identifier / version (:533-542), describe() (:418-464), clips
// and parameters (:344-414), action dispatch (:483-522).  The reference's cvPyrSegmentation call (:296-302,
// OpenCV <= 2.4 legacy pyramid linking, source not in the reference tree) is served by the mean-shift
// segmentation BASELINE.json defines for this workload: pyramid level 2 as in the reference (:280),
// `threshold 2` (the reference's cluster-merge colour threshold) as the colour radius and `threshold 1` (the
// reference's link threshold, 1..255, default 250) as the spatial radius, scaled so that the default gives the
// spatial radius 10 of BASELINE.json: sp = threshold1 / 25.  The substitution is announced to the host through the
// message suite (log message, once per instance) -- the plugin identifier and parameter set are the reference's, the
// pixels are mean-shift's.
#include <vector>

#include "ofxcv_plugin.h"

using namespace ofxcv_plugin;

#define THRESHOLD1 "threshold1"
#define THRESHOLD2 "threshold2"
#define PLUGIN_GROUPING "Draw"

static const char *kDescription = OFXCV_OFX_LICENCE_NOTICE;  // the reference's property value (segment.cpp:41-66)

namespace {

Suites g;

struct InstanceData {  // segment.cpp:100-105 (the CvMemStorage / CvSeq members have no counterpart here)
    OfxParamHandle threshold1 = nullptr, threshold2 = nullptr;
    int isGeneralEffect = 0;
    bool announced = false;  // the substitution notice has been logged
};

InstanceData *instance_data(OfxImageEffectHandle effect) {
    OfxPropertySetHandle props = nullptr;
    check(g.effect->getPropertySet(effect, &props));
    InstanceData *d = nullptr;
    check(g.prop->propGetPointer(props, kOfxPropInstanceData, 0, (void **)&d));
    return d;
}

OfxStatus create_instance(OfxImageEffectHandle effect) {  // :128-165
    OfxPropertySetHandle props = nullptr;
    check(g.effect->getPropertySet(effect, &props));
    OfxParamSetHandle params = nullptr;
    check(g.effect->getParamSet(effect, &params));
    InstanceData *d = new InstanceData;
    char *context = nullptr;
    check(g.prop->propGetString(props, kOfxImageEffectPropContext, 0, &context));
    d->isGeneralEffect = context && std::strcmp(context, kOfxImageEffectContextGeneral) == 0;
    try {
        check(g.param->paramGetHandle(params, THRESHOLD1, &d->threshold1, nullptr));
        check(g.param->paramGetHandle(params, THRESHOLD2, &d->threshold2, nullptr));
        check(g.prop->propSetPointer(props, kOfxPropInstanceData, 0, d));
    } catch (...) {
        delete d;
        throw;
    }
    return kOfxStatOK;
}

OfxStatus destroy_instance(OfxImageEffectHandle effect) {
    delete instance_data(effect);
    return kOfxStatOK;
}

OfxStatus render(OfxImageEffectHandle instance, OfxPropertySetHandle inArgs, OfxPropertySetHandle) {  // :197-338
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

    InstanceData *d = instance_data(instance);
    double t1, t2;
    check(g.param->paramGetValueAtTime(d->threshold1, time, &t1));
    check(g.param->paramGetValueAtTime(d->threshold2, time, &t2));
    if (!d->announced && g.message) {
        d->announced = true;
        g.message->message(instance, kOfxMessageLog, "ofxcv.segment.substitution",
                           "cvPyrSegmentation(level 2, threshold 1, threshold 2) is rendered as pyramid mean-shift filtering on the GPU: "
                           "spatial radius = threshold 1 / 25 (= %g), colour radius = threshold 2 (= %d), pyramid level 2",
                           t1 / 25.0, (int)t2 > 0 ? (int)t2 : 1);
    }

    const int level = 2;                                   // :280
    const int w = src.img.width() & -(1 << level);         // :283-284: the processed rectangle is rounded down
    const int h = src.img.height() & -(1 << level);
    if (w <= 0 || h <= 0) return kOfxStatOK;
    std::vector<unsigned char> image1((size_t)w * h * 4);
    ThreadContext::Lease ctx = ThreadContext::get();
    const double sr = (int)t2 > 0 ? (double)(int)t2 : 1.0;
    const double sp = t1 / 25.0 >= 1.0 ? t1 / 25.0 : 1.0;   // threshold 1 (1..255, default 250) -> spatial radius (default 10)
    check_hip(ctx, ofxcv_segment_render_host(ctx, (const uint8_t *)src.img.data, src.img.row_bytes, w, h, sp, sr, level, image1.data(),
                                             (ptrdiff_t)w * 4));
    // write-back of the reduced rectangle, alpha 255 (:307-323)
    for (int y = rw.y1; y < rw.y1 + h; y++) {
        if (g.effect->abort(instance)) break;
        if (y < out.img.bounds.y1 || y >= out.img.bounds.y2 || y < 0 || y >= h) continue;
        OfxRGBAColourB *dstPix = (OfxRGBAColourB *)((char *)out.img.data + (ptrdiff_t)(y - out.img.bounds.y1) * out.img.row_bytes) +
                                 (rw.x1 - out.img.bounds.x1);
        const unsigned char *srcPix = image1.data() + ((size_t)y * w + rw.x1) * 4;
        for (int x = rw.x1; x < rw.x1 + w && x < w; x++, dstPix++, srcPix += 4) {
            if (x < out.img.bounds.x1 || x >= out.img.bounds.x2) continue;
            dstPix->r = srcPix[0];
            dstPix->g = srcPix[1];
            dstPix->b = srcPix[2];
            dstPix->a = 255;
        }
    }
    return kOfxStatOK;
}

OfxStatus describe_in_context(OfxImageEffectHandle effect, OfxPropertySetHandle) {  // :344-414
    OfxPropertySetHandle props = nullptr;
    check(g.effect->clipDefine(effect, kOfxImageEffectOutputClipName, &props));
    check(g.prop->propSetString(props, kOfxImageEffectPropSupportedComponents, 0, kOfxImageComponentRGBA));
    check(g.effect->clipDefine(effect, kOfxImageEffectSimpleSourceClipName, &props));
    check(g.prop->propSetString(props, kOfxImageEffectPropSupportedComponents, 0, kOfxImageComponentRGBA));
    OfxStatus st = g.fetch_all();
    if (st != kOfxStatOK) return st;
    OfxParamSetHandle params = nullptr;
    check(g.effect->getParamSet(effect, &params));
    define_double_param(g, params, THRESHOLD1, "threshold 1", "Sets the threshold #1", 1, 255, 250);
    define_double_param(g, params, THRESHOLD2, "threshold 2", "Sets the threshold #2", 1, 255, 30);
    check(g.param->paramDefine(params, kOfxParamTypePage, "Main", &props));
    check(g.prop->propSetString(props, kOfxParamPropPageChild, 0, THRESHOLD1));
    check(g.prop->propSetString(props, kOfxParamPropPageChild, 1, THRESHOLD2));
    return kOfxStatOK;
}

OfxStatus describe(OfxImageEffectHandle effect) {  // :418-464
    OfxPropertySetHandle p = nullptr;
    check(g.effect->getPropertySet(effect, &p));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsMultipleClipDepths, 0, 0));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedPixelDepths, 0, kOfxBitDepthByte));
    check(g.prop->propSetString(p, kOfxPropLabel, 0, "openCV Segment"));
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

OfxStatus plugin_main(const char *action, const void *handle, OfxPropertySetHandle inArgs, OfxPropertySetHandle outArgs) {  // :483-522
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

OfxPlugin plugin = {kOfxImageEffectPluginApi, 1, "uk.org.bratwurstandhaggis:cvPyrSegmentation", 0, 5, set_host, plugin_main};

}  // namespace

extern "C" {
OfxExport OfxPlugin *OfxGetPlugin(int nth) { return nth == 0 ? &plugin : nullptr; }
OfxExport int OfxGetNumberOfPlugins(void) { return 1; }
}
