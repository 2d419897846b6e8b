// ofxcv_plugin.h -- helpers shared by the three OFX plugins: suite pointers, the status -> exception -> status
// convention of the reference (opencv2fx/opencv2fx.cpp:8-31, inpaint.cpp:537-569), parameter definition helpers
// (opencv2fx.cpp:60-94) and a per-thread lease of libofxcv_hip contexts.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdlib>
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
// kOfxPropPluginDescription of the two opencv2fx plugins is the Open Effects Association's BSD licence notice
// (segment.cpp:41-66; inpaint.cpp:40-67 puts one credit line in front).  The property value is part of the drop-in
// contract, so it is reproduced character for character -- including the reference's missing line breaks in the
// disclaimer paragraph.
#define OFXCV_OFX_LICENCE_NOTICE                                                                                          \
    "Copyright (c) 2003, The Open Effects Association Ltd. All rights reserved.\n\n"                                        \
    "Redistribution and use in source and binary forms, with or without\nmodification, are permitted provided that the "  \
    "following conditions are met:\n\n"                                                                                   \
    "    * Redistributions of source code must retain the above copyright notice,\n      this list of conditions and "    \
    "the following disclaimer.\n"                                                                                         \
    "    * Redistributions in binary form must reproduce the above copyright notice,\n      this list of conditions "     \
    "and the following disclaimer in the documentation\n      and/or other materials provided with the distribution.\n"  \
    "    * Neither the name The Open Effects Association Ltd, nor the names of its\n      contributors may be used to "   \
    "endorse or promote products derived from this\n      software without specific prior written permission.\n\n"       \
    "THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS \"AS IS\" AND" "ANY EXPRESS OR IMPLIED "          \
    "WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED" "WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A "           \
    "PARTICULAR PURPOSE ARE" "DISCLAIMED. IN NO EVENT SHALL THE COPYRIGHT OWNER OR CONTRIBUTORS BE LIABLE FOR" "ANY "     \
    "DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES" "(INCLUDING, BUT NOT LIMITED TO, "       \
    "PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES;" "LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER "    \
    "CAUSED AND ON" "ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT" "(INCLUDING NEGLIGENCE OR " \
    "OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS" "SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE."

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
    std::string unique_id;  // kOfxImagePropUniqueIdentifier where the host provides it ("" otherwise): names the pixels
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
        // (tagged with the clip it came from: a host that hands out the same identifiers on different clips or instances does not get one clip's frames
        // for another's -- the device-side cache trusts the name; frames of one clip of one instance, the playback case, still find each other)
        if (s.prop->propGetString(img.handle, kOfxImagePropUniqueIdentifier, 0, &str) == kOfxStatOK && str && *str) {
            char tag[40];
            std::snprintf(tag, sizeof tag, "%p:", (void *)clip);
            img.unique_id = std::string(tag) + str;
        }
    }
    ~ImageGuard() {
        if (img.handle) s.effect->clipReleaseImage(img.handle);
    }
    ImageGuard(const ImageGuard &) = delete;
    ImageGuard &operator=(const ImageGuard &) = delete;
};

// Contexts of libofxcv_hip are LEASED per render() call from a pool per device (render may be called concurrently: VectorGenerator is
// eRenderFullySafe, VectorGenerator.cpp:108; a context serves one call at a time).  A device holds as many contexts as it has ever had renders
// in flight at once -- not one per render thread and device (named frames are routed by frame time, so with per-thread
// contexts every render thread ended up with streams, staging and Farneback scratch on every GPU).  Which device: a thread's unnamed
// renders go to its home device (threads are spread round-robin over the visible devices) -- unless the caller asks for one (get(device)):
// frames that travel with names (kOfxImagePropUniqueIdentifier) stay on the device they were uploaded to, so VectorGenerator sends BLOCKS
// of consecutive frame times to the same device whatever thread renders them (device_for_time) and a sequence's frames are found again
// on an 8-GPU node as they are on one GPU.  A HIP failure maps to kOfxStatFailed / kOfxStatErrMemory.
class ThreadContext {
  public:
    static constexpr int kMaxDevices = 64;
    // (the runtime is asked once -- hipGetDeviceCount is the expensive part -- and again only when OFXCV_VIRTUAL_DEVICES has changed or after OfxActionUnload)
    static int device_count() {
        const char *e = std::getenv("OFXCV_VIRTUAL_DEVICES");
        std::lock_guard<std::mutex> lock(count_mu());
        CountCache &c = count_cache();
        if (c.n < 0 || c.env != (e ? e : "")) {
            c.n = ofxcv_device_count();
            c.env = e ? e : "";
        }
        return c.n;
    }
    // the device of a named frame time: blocks of OFXCV_FRAMES_PER_DEVICE (default 16) consecutive frames per device, round-robin
    static int device_for_time(double time) {
        const int n = device_count();
        if (n <= 1) return 0;
        static const int per = [] {
            const char *e = std::getenv("OFXCV_FRAMES_PER_DEVICE");
            const int v = e ? std::atoi(e) : 0;
            return v > 0 ? v : 16;
        }();
        const long blk = (long)std::floor(time / per);
        return (int)(((blk % n) + n) % n);
    }
    // a context of `device` for the duration of one render() (converts to ofxcv_ctx *); back to the pool on scope exit
    class Lease {
      public:
        Lease(ofxcv_ctx *c, int d) : ctx_(c), dev_(d) {}
        Lease(Lease &&o) noexcept : ctx_(o.ctx_), dev_(o.dev_) { o.ctx_ = nullptr; }
        Lease(const Lease &) = delete;
        Lease &operator=(const Lease &) = delete;
        ~Lease() {
            if (!ctx_) return;
            Pool &p = pool(dev_);
            std::lock_guard<std::mutex> lock(p.mu);
            p.idle.push_back(ctx_);
        }
        operator ofxcv_ctx *() const { return ctx_; }

      private:
        ofxcv_ctx *ctx_;
        int dev_;
    };
    static Lease get(int device = -1) {
        const int n = device_count();
        if (n <= 0) throw SuiteError(kOfxStatFailed);
        if (device < 0) {
            thread_local int home = -1;  // this thread's place in the round-robin over the devices
            if (home < 0) {
                static std::atomic<int> next{0};
                home = next.fetch_add(1);
            }
            device = home % n;
        }
        device = (device % n) % kMaxDevices;
        Pool &p = pool(device);
        {
            std::lock_guard<std::mutex> lock(p.mu);
            if (!p.idle.empty()) {
                ofxcv_ctx *c = p.idle.back();  // the most recently used: its scratch is sized and warm
                p.idle.pop_back();
                return Lease(c, device);
            }
        }
        ofxcv_ctx *c = nullptr;
        int rc = ofxcv_ctx_create(device, &c);
        if (rc == OFXCV_ERR_MEMORY) throw std::bad_alloc();
        if (rc != OFXCV_OK) throw SuiteError(kOfxStatFailed);
        return Lease(c, device);
    }
    // OfxActionUnload: the host guarantees no render is in flight, so every context this plugin binary created is idle in its pool:
    // all are destroyed (streams, scratch, pinned staging), with the named frames kept on their devices; renders after that re-create theirs
    static void release_all() {
        for (int d = 0; d < kMaxDevices; d++) {
            Pool &p = pool(d);
            std::lock_guard<std::mutex> lock(p.mu);
            bool first = true;
            for (ofxcv_ctx *c : p.idle) {
                if (first) (void)ofxcv_host_cache_clear(c);  // (per device: the named frames and the submission queue's batch contexts)
                first = false;
                ofxcv_ctx_destroy(c);
            }
            p.idle.clear();
        }
        {
            std::lock_guard<std::mutex> lock(count_mu());
            count_cache().n = -1;
        }
    }

  private:
    struct Pool {
        std::mutex mu;
        std::vector<ofxcv_ctx *> idle;
    };
    static Pool &pool(int device) {
        static Pool *pools = new Pool[kMaxDevices];  // intentionally leaked: no static destructor runs while a host unloads the plugin
        return pools[device];
    }
    struct CountCache {
        int n = -1;
        std::string env;
    };
    static CountCache &count_cache() { static CountCache *c = new CountCache(); return *c; }
    static std::mutex &count_mu() { static std::mutex *m = new std::mutex(); return *m; }
};

inline void check_hip(ofxcv_ctx *ctx, int rc) {
    if (rc == OFXCV_OK) return;
    std::printf("ofxcv: %s\n", ofxcv_last_error(ctx));
    if (rc == OFXCV_ERR_MEMORY) throw std::bad_alloc();
    throw SuiteError(kOfxStatFailed);
}

}  // namespace ofxcv_plugin
