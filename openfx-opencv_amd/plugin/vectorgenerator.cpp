// vectorgenerator.cpp -- OFX plugin "net.sf.openfx.VectorGenerator" v1.0, MI355X back-end.
//
// Drop-in for VectorGenerator/VectorGenerator.cpp (+ OpenCV/GenericOpenCVPlugin.cpp).  The reference is written
// against the OFX C++ Support library, which is an empty submodule in the reference tree; this file speaks the
// OFX C API directly (like the opencv2fx plugins do) and reproduces what the Support library would declare:
//   identifier / version         VectorGenerator.cpp:98-103, 939
//   describe()                   GenericOpenCVPlugin.cpp:327-358 via VectorGenerator.cpp:700-704
//   clips + the parameter set    VectorGenerator.cpp:706-930 (OpenCV >= 3 build: no "Simple flow" option / params)
//   render()                     :523-640, getFramesNeeded :675-695, changedParam / updateVisibility :642-673
// The Farneback branch of calcOpticalFlow (:374-406 + write-back :494-519) runs on the GPU through
// ofxcv_vectorgen_flow_host; the SimpleFlow and Dual TV-L1 methods are outside the accelerated path and fail
// with a message.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <string>

#include "ofxcv_plugin.h"

using namespace ofxcv_plugin;

#define kPluginName "VectorGeneratorOFX"
#define kPluginGrouping "Time"
#define kPluginDescription "Compute optical flow for the input sequence, using OpenCV."
#define kPluginIdentifier "net.sf.openfx.VectorGenerator"

namespace {

Suites g;

struct ChoiceDef { const char *name, *label, *hint; int def; };
const ChoiceDef kChannels[4] = {
    {"rChannel", "R channel", "Selects which component of the motion vectors to set in the red channel of the output image", 1},
    {"gChannel", "G channel", "Selects which component of the motion vectors to set in the green channel of the output image", 2},
    {"bChannel", "B channel", "Selects which component of the motion vectors to set in the blue channel of the output image", 3},
    {"aChannel", "A channel", "Selects which component of the motion vectors to set in the alpha channel of the output image", 4}};
const char *kChannelOptions[5] = {"0", "forward.u", "forward.v", "backward.u", "backward.v"};
// per-option help texts of appendOption(name, hint), VectorGenerator.cpp:126-137 (the last one says "x flow" in the reference too)
const char *kChannelOptionHints[5] = {"0 constant channel", "x flow (in pixels) to the next frame.", "y flow (in pixels) to the next frame.",
                                      "x flow (in pixels) to the previous frame.", "x flow (in pixels) to the previous frame."};

enum Method { eFarneback = 0, eSimpleFlow = 1, eDualTVL1 = 2 };  // VectorGenerator.cpp:219-224

struct NumDef { const char *name, *label, *hint; bool is_int; double def; };
const NumDef kNums[] = {
    {"levels", "Levels", "Number of pyramid levels including initial image. If 1 that means no extra layer will be created and only the original images are used.", true, 3},
    {"iterations", "Iterations", "Number of iterations the algorithm uses at each pyramid level", true, 15},
    {"neighborhood", "Neighborhood",
     "Size of the pixel neighborhood used to find the polynomial expansion in each pixel. Larger values mean that the image will "
     "be approximated with smoother surfaces, yielding a more robust algorithm and more blurred motion field.", true, 5},
    {"sigma", "Sigma",
     "Standard deviation of the Gaussian used to smooth derivatives used as a basis of the  polynomial expansion. For a Neighborhood of 5 "
     "you can set Sigma to 1.1. For a Neighborhood of 7, a good value for sigma would be 1.5.", false, 1.1},
    {"tau", "Tau", "Time step of the numerical scheme", false, 0.25},
    {"lambda", "Lambda", "This determines the smoothness of the output. The smaller the parameter is, the smoother the solutions we obtain.", false, 0.15},
    {"theta", "Theta",
     "It serves as a link between the attachment and the regularization terms. It should have a small value in order to maintain both parts in "
     "correspondance.", false, 0.3},
    {"nScales", "N. Scales", "Number of scales used to create the pyramid image", true, 5},
    {"warps", "Warps", "Number of warpings per scale. This affects the stability of the method at the expense of running time.", true, 5},
    {"epsilon", "Epsilon", "Stopping criterion theshold which is a trade-off between accuracy and running time. A small value will yield more accurate solutions.", false, 0.01}};

struct InstanceData {
    OfxImageClipHandle dstClip = nullptr, srcClip = nullptr;
    OfxParamHandle channel[4] = {nullptr, nullptr, nullptr, nullptr};
    OfxParamHandle method = nullptr, levels = nullptr, iterations = nullptr, neighborhood = nullptr, sigma = nullptr;
    OfxParamSetHandle params = nullptr;
};

InstanceData *instance_data(OfxImageEffectHandle effect) {
    OfxPropertySetHandle props = nullptr;
    check(g.effect->getPropertySet(effect, &props));
    InstanceData *d = nullptr;
    check(g.prop->propGetPointer(props, kOfxPropInstanceData, 0, (void **)&d));
    if (!d) throw SuiteError(kOfxStatErrBadHandle);
    return d;
}

void set_labels(OfxPropertySetHandle p, const char *label) {  // ParamDescriptor::setLabels(label, label, label)
    check(g.prop->propSetString(p, kOfxPropLabel, 0, label));
    check(g.prop->propSetString(p, kOfxPropShortLabel, 0, label));
    check(g.prop->propSetString(p, kOfxPropLongLabel, 0, label));
}

OfxStatus describe(OfxImageEffectHandle effect) {  // genericCVDescribe, GenericOpenCVPlugin.cpp:327-358
    OfxPropertySetHandle p = nullptr;
    check(g.effect->getPropertySet(effect, &p));
    set_labels(p, kPluginName);
    check(g.prop->propSetString(p, kOfxImageEffectPluginPropGrouping, 0, kPluginGrouping));
    check(g.prop->propSetString(p, kOfxPropPluginDescription, 0, kPluginDescription));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedContexts, 0, kOfxImageEffectContextFilter));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedContexts, 1, kOfxImageEffectContextGeneral));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedPixelDepths, 0, kOfxBitDepthFloat));
    check(g.prop->propSetInt(p, kOfxImageEffectPluginPropSingleInstance, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPluginPropHostFrameThreading, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsMultiResolution, 0, 1));  // kSupportsMultiResolution, :106
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsTiles, 0, 0));            // kSupportsTiles, :105
    check(g.prop->propSetInt(p, kOfxImageEffectPropTemporalClipAccess, 0, 1));
    check(g.prop->propSetInt(p, kOfxImageEffectPluginPropFieldRenderTwiceAlways, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsMultipleClipPARs, 0, 0));
    check(g.prop->propSetString(p, kOfxImageEffectPluginRenderThreadSafety, 0, kOfxImageEffectRenderFullySafe));  // :108
    return kOfxStatOK;
}

OfxStatus describe_in_context(OfxImageEffectHandle effect, OfxPropertySetHandle) {  // :706-930
    OfxStatus st = g.fetch_all();
    if (st != kOfxStatOK) return st;
    OfxPropertySetHandle p = nullptr;
    check(g.effect->clipDefine(effect, kOfxImageEffectSimpleSourceClipName, &p));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedComponents, 0, kOfxImageComponentRGBA));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedComponents, 1, kOfxImageComponentRGB));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedComponents, 2, kOfxImageComponentAlpha));
    check(g.prop->propSetInt(p, kOfxImageEffectPropTemporalClipAccess, 0, 1));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsTiles, 0, 0));
    check(g.prop->propSetInt(p, kOfxImageClipPropIsMask, 0, 0));
    check(g.effect->clipDefine(effect, kOfxImageEffectOutputClipName, &p));
    check(g.prop->propSetString(p, kOfxImageEffectPropSupportedComponents, 0, kOfxImageComponentRGBA));
    check(g.prop->propSetInt(p, kOfxImageEffectPropSupportsTiles, 0, 0));

    OfxParamSetHandle params = nullptr;
    check(g.effect->getParamSet(effect, &params));
    OfxPropertySetHandle page = nullptr;
    check(g.param->paramDefine(params, kOfxParamTypePage, "Controls", &page));
    int child = 0;
    for (const ChoiceDef &c : kChannels) {
        check(g.param->paramDefine(params, kOfxParamTypeChoice, c.name, &p));
        set_labels(p, c.label);
        // ChoiceParamDescriptor::appendOption(name, hint): the option help goes into the host's per-option label property
        // and, on hosts without it, is appended to the parameter hint as "name: hint" lines
        std::string hint = c.hint;
        for (int i = 0; i < 5; i++) {
            check(g.prop->propSetString(p, kOfxParamPropChoiceOption, i, kChannelOptions[i]));
            if (g.prop->propSetString(p, kOfxParamPropChoiceLabelOption, i, kChannelOptionHints[i]) != kOfxStatOK)
                hint += std::string("\n") + kChannelOptions[i] + ": " + kChannelOptionHints[i];
        }
        check(g.prop->propSetString(p, kOfxParamPropHint, 0, hint.c_str()));
        check(g.prop->propSetInt(p, kOfxParamPropDefault, 0, c.def));
        check(g.prop->propSetInt(p, kOfxParamPropAnimates, 0, 1));
        check(g.prop->propSetString(page, kOfxParamPropPageChild, child++, c.name));
    }
    check(g.param->paramDefine(params, kOfxParamTypeChoice, "method", &p));
    set_labels(p, "Method");
    check(g.prop->propSetString(p, kOfxParamPropHint, 0, ""));
    check(g.prop->propSetString(p, kOfxParamPropChoiceOption, 0, "Farneback"));
    check(g.prop->propSetString(p, kOfxParamPropChoiceOption, 1, "Dual TV L1"));  // OpenCV >= 3: "Simple flow" is not appended (:790-793)
    check(g.prop->propSetInt(p, kOfxParamPropDefault, 0, (int)eFarneback));
    check(g.prop->propSetInt(p, kOfxParamPropAnimates, 0, 0));
    check(g.prop->propSetString(page, kOfxParamPropPageChild, child++, "method"));
    for (const NumDef &n : kNums) {
        check(g.param->paramDefine(params, n.is_int ? kOfxParamTypeInteger : kOfxParamTypeDouble, n.name, &p));
        set_labels(p, n.label);
        check(g.prop->propSetString(p, kOfxParamPropHint, 0, n.hint));
        if (n.is_int) check(g.prop->propSetInt(p, kOfxParamPropDefault, 0, (int)n.def));
        else check(g.prop->propSetDouble(p, kOfxParamPropDefault, 0, n.def));
        check(g.prop->propSetInt(p, kOfxParamPropAnimates, 0, 1));
        check(g.prop->propSetString(page, kOfxParamPropPageChild, child++, n.name));
    }
    return kOfxStatOK;
}

int get_int(OfxParamHandle h, OfxTime t) {
    int v = 0;
    check(g.param->paramGetValueAtTime(h, t, &v));
    return v;
}
double get_double(OfxParamHandle h, OfxTime t) {
    double v = 0;
    check(g.param->paramGetValueAtTime(h, t, &v));
    return v;
}

void set_secret(OfxParamSetHandle params, const char *name, bool secret) {
    OfxParamHandle h = nullptr;
    OfxPropertySetHandle p = nullptr;
    check(g.param->paramGetHandle(params, name, &h, &p));
    if (!p) check(g.param->paramGetPropertySet(h, &p));
    check(g.prop->propSetInt(p, kOfxParamPropSecret, 0, secret ? 1 : 0));
}

void update_visibility(InstanceData *d, int method) {  // :642-662
    set_secret(d->params, "levels", method != eFarneback);
    set_secret(d->params, "iterations", method != eFarneback && method != eDualTVL1);
    set_secret(d->params, "neighborhood", method != eFarneback);
    set_secret(d->params, "sigma", method != eFarneback);
    for (const char *n : {"tau", "lambda", "theta", "nScales", "warps", "epsilon"}) set_secret(d->params, n, method != eDualTVL1);
}

OfxStatus create_instance(OfxImageEffectHandle effect) {  // VectorGeneratorPlugin ctor :231-262 + GenericOpenCVPlugin ctor
    OfxPropertySetHandle props = nullptr;
    check(g.effect->getPropertySet(effect, &props));
    InstanceData *d = new InstanceData;
    try {
        check(g.effect->getParamSet(effect, &d->params));
        check(g.effect->clipGetHandle(effect, kOfxImageEffectOutputClipName, &d->dstClip, nullptr));
        check(g.effect->clipGetHandle(effect, kOfxImageEffectSimpleSourceClipName, &d->srcClip, nullptr));
        for (int i = 0; i < 4; i++) check(g.param->paramGetHandle(d->params, kChannels[i].name, &d->channel[i], nullptr));
        check(g.param->paramGetHandle(d->params, "method", &d->method, nullptr));
        check(g.param->paramGetHandle(d->params, "levels", &d->levels, nullptr));
        check(g.param->paramGetHandle(d->params, "iterations", &d->iterations, nullptr));
        check(g.param->paramGetHandle(d->params, "neighborhood", &d->neighborhood, nullptr));
        check(g.param->paramGetHandle(d->params, "sigma", &d->sigma, nullptr));
        check(g.prop->propSetPointer(props, kOfxPropInstanceData, 0, d));
        int method = 0;
        check(g.param->paramGetValue(d->method, &method));
        update_visibility(d, method);
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

void fail_with_message(OfxImageEffectHandle effect, const char *text) {  // setPersistentMessage + throw kOfxStatFailed (:531-536)
    if (g.message) g.message->message(effect, kOfxMessageError, "", "%s", text);
    throw SuiteError(kOfxStatFailed);
}

// one direction of calcOpticalFlow (:353-520): `mu`/`mv` = RGBA channels receiving flow.x / flow.y
void calc_optical_flow(OfxImageEffectHandle effect, double time, const Image &ref, const Image &other, const Image &dst, const OfxRectI &rw,
                       double rsx, double rsy, unsigned mu, unsigned mv, int levels, int iterations, int poly_n, double poly_sigma) {
    if (ref.depth != kOfxBitDepthFloat || other.depth != kOfxBitDepthFloat || dst.depth != kOfxBitDepthFloat) throw SuiteError(kOfxStatErrImageFormat);
    int ncomp = ref.components == kOfxImageComponentRGBA ? 4 : (ref.components == kOfxImageComponentRGB ? 3 : 0);
    if (!ncomp || other.components != ref.components) fail_with_message(effect, "VectorGenerator: the Farneback method needs RGB or RGBA sources");
    if (dst.components != kOfxImageComponentRGBA) throw SuiteError(kOfxStatErrImageFormat);
    // the flow field covers the reference bounds (:365-372; the union with `other` is a no-op in the reference)
    const int w = ref.width(), h = ref.height();
    if (other.width() != w || other.height() != h) fail_with_message(effect, "VectorGenerator: source frames differ in size");
    // the reference's write-back is defined for windows anchored at the image origin (:494-519); general windows
    // are served by writing the part of the render window that lies inside both the flow field and the output image
    const int x1 = std::max(std::max(rw.x1, ref.bounds.x1), dst.bounds.x1), x2 = std::min(std::min(rw.x2, ref.bounds.x2), dst.bounds.x2);
    const int y1 = std::max(std::max(rw.y1, ref.bounds.y1), dst.bounds.y1), y2 = std::min(std::min(rw.y2, ref.bounds.y2), dst.bounds.y2);
    if (x1 >= x2 || y1 >= y2) return;
    // named frames: the device of this frame time's block (they are found again there whatever thread renders the neighbouring frames)
    ThreadContext::Lease ctx = ThreadContext::get(ref.unique_id.empty() ? -1 : ThreadContext::device_for_time(time));
    if (x1 == ref.bounds.x1 && y1 == ref.bounds.y1 && x2 == ref.bounds.x2 && y2 == ref.bounds.y2) {
        float *d0 = (float *)((char *)dst.data + (ptrdiff_t)(y1 - dst.bounds.y1) * dst.row_bytes) + (size_t)(x1 - dst.bounds.x1) * 4;
        // (the frames with the names the host gives their pixels, as in the two-direction call of render())
        check_hip(ctx, ofxcv_vectorgen_flows_host_keyed(ctx, (const float *)ref.data, ref.row_bytes, (const float *)other.data, other.row_bytes, nullptr, 0,
                                                        ncomp, w, h, d0, dst.row_bytes, mu, mv, 0, 0, rsx, rsy, levels, iterations, poly_n, poly_sigma,
                                                        ref.unique_id.c_str(), other.unique_id.c_str(), nullptr));
    } else {
        std::vector<float> tmp((size_t)w * h * 4);
        // pre-load the window so unmapped channels keep the output image's content
        for (int y = y1; y < y2; y++)
            std::memcpy(&tmp[((size_t)(y - ref.bounds.y1) * w + (x1 - ref.bounds.x1)) * 4],
                        (char *)dst.data + (ptrdiff_t)(y - dst.bounds.y1) * dst.row_bytes + (size_t)(x1 - dst.bounds.x1) * 16, (size_t)(x2 - x1) * 16);
        check_hip(ctx, ofxcv_vectorgen_flow_host(ctx, (const float *)ref.data, ref.row_bytes, (const float *)other.data, other.row_bytes, ncomp, w, h,
                                                 tmp.data(), (ptrdiff_t)w * 16, mu, mv, rsx, rsy, levels, iterations, poly_n, poly_sigma));
        for (int y = y1; y < y2; y++)
            std::memcpy((char *)dst.data + (ptrdiff_t)(y - dst.bounds.y1) * dst.row_bytes + (size_t)(x1 - dst.bounds.x1) * 16,
                        &tmp[((size_t)(y - ref.bounds.y1) * w + (x1 - ref.bounds.x1)) * 4], (size_t)(x2 - x1) * 16);
    }
}

OfxStatus render(OfxImageEffectHandle effect, OfxPropertySetHandle inArgs, OfxPropertySetHandle) {  // :523-640
    InstanceData *d = instance_data(effect);
    OfxTime time;
    OfxRectI rw;
    double rs[2] = {1, 1};
    char *field = nullptr;
    check(g.prop->propGetDouble(inArgs, kOfxPropTime, 0, &time));
    check(g.prop->propGetIntN(inArgs, kOfxImageEffectPropRenderWindow, 4, &rw.x1));
    check(g.prop->propGetDoubleN(inArgs, kOfxImageEffectPropRenderScale, 2, rs));
    if (g.prop->propGetString(inArgs, kOfxImageEffectPropFieldToRender, 0, &field) != kOfxStatOK) field = nullptr;

    ImageGuard dst(g, d->dstClip, time);
    if (!dst.img.data) throw SuiteError(kOfxStatFailed);
    if (dst.img.scale_x != rs[0] || dst.img.scale_y != rs[1] || (field && !dst.img.field.empty() && dst.img.field != field))
        fail_with_message(effect, "OFX Host gave image with wrong scale or field properties");
    ImageGuard ref(g, d->srcClip, time);
    if (!ref.img.data) throw SuiteError(kOfxStatFailed);

    int ch[4];
    for (int i = 0; i < 4; i++) ch[i] = get_int(d->channel[i], time);
    bool forward = false, backward = false;
    for (int i = 0; i < 4; i++) {
        forward |= ch[i] == 1 || ch[i] == 2;
        backward |= ch[i] == 3 || ch[i] == 4;
    }
    const int method = get_int(d->method, time);
    if ((forward || backward) && method != eFarneback)
        fail_with_message(effect, "VectorGenerator: only the Farneback method is implemented by the MI355X back-end");
    const int levels = get_int(d->levels, time), iterations = get_int(d->iterations, time), poly_n = get_int(d->neighborhood, time);
    const double poly_sigma = get_double(d->sigma, time);
    // both directions over the whole reference frame: one library call stages and converts the reference once and
    // overlaps the second frame pair with the first flow
    if (forward && backward) {
        ImageGuard next(g, d->srcClip, time + 1), prev(g, d->srcClip, time - 1);
        if (!next.img.data || !prev.img.data) throw SuiteError(kOfxStatFailed);
        const Image &r = ref.img, &o = dst.img;
        const bool whole = rw.x1 <= r.bounds.x1 && rw.y1 <= r.bounds.y1 && rw.x2 >= r.bounds.x2 && rw.y2 >= r.bounds.y2 &&
                           o.bounds.x1 <= r.bounds.x1 && o.bounds.y1 <= r.bounds.y1 && o.bounds.x2 >= r.bounds.x2 && o.bounds.y2 >= r.bounds.y2;
        const int ncomp = r.components == kOfxImageComponentRGBA ? 4 : (r.components == kOfxImageComponentRGB ? 3 : 0);
        const bool same = ncomp && next.img.components == r.components && prev.img.components == r.components &&
                          next.img.width() == r.width() && next.img.height() == r.height() && prev.img.width() == r.width() &&
                          prev.img.height() == r.height() && r.depth == kOfxBitDepthFloat && next.img.depth == kOfxBitDepthFloat &&
                          prev.img.depth == kOfxBitDepthFloat && o.depth == kOfxBitDepthFloat && o.components == kOfxImageComponentRGBA;
        if (whole && same) {
            unsigned fu = 0, fv = 0, bu = 0, bv = 0;
            for (int i = 0; i < 4; i++) {
                if (ch[i] == 1) fu |= 1u << i;
                if (ch[i] == 2) fv |= 1u << i;
                if (ch[i] == 3) bu |= 1u << i;
                if (ch[i] == 4) bv |= 1u << i;
            }
            ThreadContext::Lease ctx = ThreadContext::get(r.unique_id.empty() ? -1 : ThreadContext::device_for_time(time));
            float *d0 = (float *)((char *)o.data + (ptrdiff_t)(r.bounds.y1 - o.bounds.y1) * o.row_bytes) + (size_t)(r.bounds.x1 - o.bounds.x1) * 4;
            // The frames travel with the names the host gives their pixels (kOfxImagePropUniqueIdentifier; "" = none): rendering
            // frame t+1 after frame t finds two of its three source frames on the device already.  A name covers the whole image:
            // it is only passed where the image IS the frame the library sees (same bounds as the reference, which `same` checked
            // by size, and the data pointer at its origin).
            check_hip(ctx, ofxcv_vectorgen_flows_host_keyed(ctx, (const float *)r.data, r.row_bytes, (const float *)next.img.data, next.img.row_bytes,
                                                            (const float *)prev.img.data, prev.img.row_bytes, ncomp, r.width(), r.height(), d0,
                                                            o.row_bytes, fu, fv, bu, bv, rs[0], rs[1], levels, iterations, poly_n, poly_sigma,
                                                            r.unique_id.c_str(), next.img.unique_id.c_str(), prev.img.unique_id.c_str()));
            return kOfxStatOK;
        }
    }
    if (forward) {
        ImageGuard other(g, d->srcClip, time + 1);
        if (!other.img.data) throw SuiteError(kOfxStatFailed);
        unsigned mu = 0, mv = 0;
        for (int i = 0; i < 4; i++) {
            if (ch[i] == 1) mu |= 1u << i;
            if (ch[i] == 2) mv |= 1u << i;
        }
        calc_optical_flow(effect, time, ref.img, other.img, dst.img, rw, rs[0], rs[1], mu, mv, levels, iterations, poly_n, poly_sigma);
    }
    if (backward) {
        ImageGuard other(g, d->srcClip, time - 1);
        if (!other.img.data) throw SuiteError(kOfxStatFailed);
        unsigned mu = 0, mv = 0;
        for (int i = 0; i < 4; i++) {
            if (ch[i] == 3) mu |= 1u << i;
            if (ch[i] == 4) mv |= 1u << i;
        }
        calc_optical_flow(effect, time, ref.img, other.img, dst.img, rw, rs[0], rs[1], mu, mv, levels, iterations, poly_n, poly_sigma);
    }
    return kOfxStatOK;
}

OfxStatus get_frames_needed(OfxImageEffectHandle effect, OfxPropertySetHandle inArgs, OfxPropertySetHandle outArgs) {  // :675-695
    InstanceData *d = instance_data(effect);
    OfxTime time;
    check(g.prop->propGetDouble(inArgs, kOfxPropTime, 0, &time));
    bool forward = false, backward = false;
    for (int i = 0; i < 4; i++) {
        int c = get_int(d->channel[i], time);
        forward |= c == 1 || c == 2;
        backward |= c == 3 || c == 4;
    }
    if (forward || backward) {
        double range[2] = {time - (int)backward, time + (int)forward};
        check(g.prop->propSetDoubleN(outArgs, kOfxImageClipPropFrameRangePrefix kOfxImageEffectSimpleSourceClipName, 2, range));
    }
    return kOfxStatOK;
}

OfxStatus instance_changed(OfxImageEffectHandle effect, OfxPropertySetHandle inArgs) {  // changedParam :664-673
    char *type = nullptr, *name = nullptr;
    if (g.prop->propGetString(inArgs, kOfxPropType, 0, &type) != kOfxStatOK || !type || std::strcmp(type, kOfxTypeParameter)) return kOfxStatReplyDefault;
    check(g.prop->propGetString(inArgs, kOfxPropName, 0, &name));
    if (name && !std::strcmp(name, "method")) {
        InstanceData *d = instance_data(effect);
        int method = 0;
        check(g.param->paramGetValue(d->method, &method));
        update_visibility(d, method);
        return kOfxStatOK;
    }
    return kOfxStatReplyDefault;
}

OfxStatus plugin_main(const char *action, const void *handle, OfxPropertySetHandle inArgs, OfxPropertySetHandle outArgs) {
    return guarded([&]() -> OfxStatus {
        OfxImageEffectHandle effect = (OfxImageEffectHandle)handle;
        if (!std::strcmp(action, kOfxActionLoad)) return g.fetch_basic();
        if (!std::strcmp(action, kOfxActionUnload)) {  // the reference frees its LUT manager here (:697); this one its device contexts
            ThreadContext::release_all();
            return kOfxStatOK;
        }
        if (!std::strcmp(action, kOfxActionDescribe)) return describe(effect);
        if (!std::strcmp(action, kOfxImageEffectActionDescribeInContext)) return describe_in_context(effect, inArgs);
        if (!std::strcmp(action, kOfxActionCreateInstance)) return create_instance(effect);
        if (!std::strcmp(action, kOfxActionDestroyInstance)) return destroy_instance(effect);
        if (!std::strcmp(action, kOfxImageEffectActionRender)) return render(effect, inArgs, outArgs);
        if (!std::strcmp(action, kOfxImageEffectActionGetFramesNeeded)) return get_frames_needed(effect, inArgs, outArgs);
        if (!std::strcmp(action, kOfxActionInstanceChanged)) return instance_changed(effect, inArgs);
        return kOfxStatReplyDefault;
    });
}

void set_host(OfxHost *h) { g.host = h; }

OfxPlugin plugin = {kOfxImageEffectPluginApi, 1, kPluginIdentifier, 1, 0, set_host, plugin_main};

}  // namespace

extern "C" {
OfxExport OfxPlugin *OfxGetPlugin(int nth) { return nth == 0 ? &plugin : nullptr; }
OfxExport int OfxGetNumberOfPlugins(void) { return 1; }
}
