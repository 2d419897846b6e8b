// ofxcv_plugin.h -- helpers shared by the three OFX plugins: suite pointers, the status -> exception -> status
// convention of the reference (opencv2fx/opencv2fx.cpp:8-31, inpaint.cpp:537-569), parameter definition helpers
// (opencv2fx.cpp:60-94) and a per-thread lease of libofxcv_hip contexts.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ofx_min.h"
#include "ofxcv_hip.h"

namespace ofxcv_plugin {

struct SuiteError : std::exception {
    OfxStatus status;
    explicit SuiteError(OfxStatus s) : status(s) {}
    const char *what() const noexcept override { return "OFX suite error"; }
};

// opencv2fx.cpp:8-25: OK / Reply* pass, ErrMemory -> bad_alloc, anything else -> Suite exception
inline void check(OfxStatus stat) {
    switch (stat) {
        case kOfxStatOK:
        case kOfxStatReplyYes:
        case kOfxStatReplyNo:
        case kOfxStatReplyDefault: return;
        case kOfxStatErrMemory: throw std::bad_alloc();
        default: throw SuiteError(stat);
    }
}

struct Suites {
    OfxHost *host = nullptr;
    const OfxImageEffectSuiteV1 *effect = nullptr;
    const OfxPropertySuiteV1 *prop = nullptr;
    const OfxParameterSuiteV1 *param = nullptr;
    const OfxMemorySuiteV1 *memory = nullptr;
    const OfxMultiThreadSuiteV1 *thread = nullptr;
    const OfxMessageSuiteV1 *message = nullptr;  // optional
    const void *interact = nullptr;              // optional

    // inpaint.cpp:519-530 (onLoad): effect + property suites are mandatory
    OfxStatus fetch_basic() {
        if (!host) return kOfxStatErrMissingHostFeature;
        effect = (const OfxImageEffectSuiteV1 *)host->fetchSuite(host->host, kOfxImageEffectSuite, 1);
        prop = (const OfxPropertySuiteV1 *)host->fetchSuite(host->host, kOfxPropertySuite, 1);
        return (effect && prop) ? kOfxStatOK : kOfxStatErrMissingHostFeature;
    }
    // inpaint.cpp:404-415: the five mandatory suites, message / interact optional
    OfxStatus fetch_all() {
        if (!host) return kOfxStatErrMissingHostFeature;
        effect = (const OfxImageEffectSuiteV1 *)host->fetchSuite(host->host, kOfxImageEffectSuite, 1);
        prop = (const OfxPropertySuiteV1 *)host->fetchSuite(host->host, kOfxPropertySuite, 1);
        param = (const OfxParameterSuiteV1 *)host->fetchSuite(host->host, kOfxParameterSuite, 1);
        memory = (const OfxMemorySuiteV1 *)host->fetchSuite(host->host, kOfxMemorySuite, 1);
        thread = (const OfxMultiThreadSuiteV1 *)host->fetchSuite(host->host, kOfxMultiThreadSuite, 1);
        message = (const OfxMessageSuiteV1 *)host->fetchSuite(host->host, kOfxMessageSuite, 1);
        interact = host->fetchSuite(host->host, kOfxInteractSuite, 1);
        if (!effect || !prop || !param || !memory || !thread) return kOfxStatErrMissingHostFeature;
        return kOfxStatOK;
    }
};

// the try / catch ladder of pluginMain (inpaint.cpp:554-569): no exception crosses the C ABI
template <class F>
inline OfxStatus guarded(F &&f) {
    try {
        return f();
    } catch (const SuiteError &e) {
        std::printf("OFX Plugin Suite error: %d\n", e.status);
        return e.status;
    } catch (const std::bad_alloc &) {
        std::printf("OFX Plugin Memory error.\n");
        return kOfxStatErrMemory;
    } catch (const std::exception &e) {
        std::printf("OFX Plugin error: %s\n", e.what());
        return kOfxStatErrUnknown;
    } catch (...) {
        std::printf("OFX Plugin error\n");
        return kOfxStatErrUnknown;
    }
}

// opencv2fx.cpp:60-94
inline void define_double_param(const Suites &s, OfxParamSetHandle params, const char *name, const char *label, const char *hint,
                                double dmin, double dmax, double initial) {
    OfxPropertySetHandle props = nullptr;
    check(s.param->paramDefine(params, kOfxParamTypeDouble, name, &props));
    check(s.prop->propSetString(props, kOfxParamPropDoubleType, 0, kOfxParamDoubleTypeScale));
    check(s.prop->propSetDouble(props, kOfxParamPropDefault, 0, initial));
    check(s.prop->propSetDouble(props, kOfxParamPropMin, 0, 0.0));
    check(s.prop->propSetDouble(props, kOfxParamPropDisplayMin, 0, dmin));
    check(s.prop->propSetDouble(props, kOfxParamPropDisplayMax, 0, dmax));
    check(s.prop->propSetString(props, kOfxParamPropHint, 0, hint));
    check(s.prop->propSetString(props, kOfxParamPropScriptName, 0, name));
    check(s.prop->propSetString(props, kOfxPropLabel, 0, label));
}

// image properties of one fetched OFX image (inpaint.cpp:233-251)
struct Image {
    OfxPropertySetHandle handle = nullptr;
    void *data = nullptr;
    int row_bytes = 0;
    OfxRectI bounds = {0, 0, 0, 0};
    std::string depth, components;
    double scale_x = 1, scale_y = 1;
    std::string field;
    int width() const { return bounds.x2 - bounds.x1; }
    int height() const { return bounds.y2 - bounds.y1; }
};

// fetches an image and releases it on scope exit (the reference leaks images on exceptions; this does not)
struct ImageGuard {
    const Suites &s;
    Image img;
    ImageGuard(const Suites &su, OfxImageClipHandle clip, OfxTime time) : s(su) {
        check(s.effect->clipGetImage(clip, time, nullptr, &img.handle));
        char *str = nullptr;
        check(s.prop->propGetPointer(img.handle, kOfxImagePropData, 0, &img.data));
        check(s.prop->propGetInt(img.handle, kOfxImagePropRowBytes, 0, &img.row_bytes));
        check(s.prop->propGetString(img.handle, kOfxImageEffectPropPixelDepth, 0, &str));
        img.depth = str ? str : "";
        check(s.prop->propGetIntN(img.handle, kOfxImagePropBounds, 4, &img.bounds.x1));
        if (s.prop->propGetString(img.handle, kOfxImageEffectPropComponents, 0, &str) == kOfxStatOK && str) img.components = str;
        double sc[2] = {1, 1};
        if (s.prop->propGetDoubleN(img.handle, kOfxImageEffectPropRenderScale, 2, sc) == kOfxStatOK) {
            img.scale_x = sc[0];
            img.scale_y = sc[1];
        }
        if (s.prop->propGetString(img.handle, kOfxImagePropField, 0, &str) == kOfxStatOK && str) img.field = str;
    }
    ~ImageGuard() {
        if (img.handle) s.effect->clipReleaseImage(img.handle);
    }
    ImageGuard(const ImageGuard &) = delete;
    ImageGuard &operator=(const ImageGuard &) = delete;
};

// One libofxcv_hip context per calling host thread (render may be called concurrently: VectorGenerator is
// eRenderFullySafe).  Threads are spread round-robin over the visible devices; the context lives as long as
// the thread.  A HIP failure maps to kOfxStatFailed / kOfxStatErrMemory.
class ThreadContext {
  public:
    static ofxcv_ctx *get() {
        thread_local Holder h;
        if (!h.ctx) {
            static std::atomic<int> next{0};
            int n = ofxcv_device_count();
            if (n <= 0) throw SuiteError(kOfxStatFailed);
            int rc = ofxcv_ctx_create(next.fetch_add(1) % n, &h.ctx);
            if (rc == OFXCV_ERR_MEMORY) throw std::bad_alloc();
            if (rc != OFXCV_OK) throw SuiteError(kOfxStatFailed);
        }
        return h.ctx;
    }

  private:
    struct Holder {
        ofxcv_ctx *ctx = nullptr;
        ~Holder() {
            if (ctx) ofxcv_ctx_destroy(ctx);
        }
    };
};

inline void check_hip(ofxcv_ctx *ctx, int rc) {
    if (rc == OFXCV_OK) return;
    std::printf("ofxcv: %s\n", ofxcv_last_error(ctx));
    if (rc == OFXCV_ERR_MEMORY) throw std::bad_alloc();
    throw SuiteError(kOfxStatFailed);
}

}  // namespace ofxcv_plugin
