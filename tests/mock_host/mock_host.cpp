// mock_host.cpp -- a minimal OpenFX host for tests: implements the suites the plugins fetch (ImageEffect,
// Property, Parameter, Memory, MultiThread, Message), dlopen()s a .ofx binary and drives
// Load -> Describe -> DescribeInContext -> CreateInstance -> Render / GetFramesNeeded / InstanceChanged ->
// DestroyInstance.  The reference ships no host and no tests; this is the only caller of
// the drop-in boundary available here.  Exposes a small C interface for ctypes (tests/test_ofx_boundary.py).
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "ofx_min.h"

namespace {

struct Value {
    enum Kind { None, Int, Double, String, Pointer } kind = None;
    int i = 0;
    double d = 0;
    std::string s;
    void *p = nullptr;
};

struct PropSet {
    std::map<std::string, std::vector<Value>> props;
    std::vector<std::string> order;  // first-set order, for stable dumps
    Value &slot(const char *name, int index) {
        auto it = props.find(name);
        if (it == props.end()) {
            order.push_back(name);
            it = props.emplace(name, std::vector<Value>()).first;
        }
        if ((int)it->second.size() <= index) it->second.resize(index + 1);
        return it->second[index];
    }
    const Value *find(const char *name, int index) const {
        auto it = props.find(name);
        if (it == props.end() || index < 0 || index >= (int)it->second.size()) return nullptr;
        return &it->second[index];
    }
};

struct Param {
    std::string name, type;
    PropSet props;
    Value value;  // current value (Int / Double)
    bool has_value = false;
};

struct ParamSet {
    std::vector<std::unique_ptr<Param>> params;
    PropSet props;
    Param *find(const char *name) {
        for (auto &p : params)
            if (p->name == name) return p.get();
        return nullptr;
    }
};

struct Clip {
    std::string name;
    PropSet props;
    std::map<double, std::unique_ptr<PropSet>> images;  // time -> image property set
    int fetched = 0, released = 0;
};

struct Effect {
    PropSet props;
    ParamSet params;
    std::vector<std::unique_ptr<Clip>> clips;
    bool descriptor = true;
    std::string last_message;
    int abort_flag = 0;
    Clip *find_clip(const char *name) {
        for (auto &c : clips)
            if (c->name == name) return c.get();
        return nullptr;
    }
};

std::mutex g_lock;  // render may be called from several threads on one instance
std::map<PropSet *, Clip *> g_image_owner;

// ---------------- property suite
#define PS(h) reinterpret_cast<PropSet *>(h)
OfxStatus propSetPointer(OfxPropertySetHandle h, const char *n, int i, void *v) { if (!h) return kOfxStatErrBadHandle; Value &x = PS(h)->slot(n, i); x.kind = Value::Pointer; x.p = v; return kOfxStatOK; }
OfxStatus propSetString(OfxPropertySetHandle h, const char *n, int i, const char *v) { if (!h) return kOfxStatErrBadHandle; Value &x = PS(h)->slot(n, i); x.kind = Value::String; x.s = v ? v : ""; return kOfxStatOK; }
OfxStatus propSetDouble(OfxPropertySetHandle h, const char *n, int i, double v) { if (!h) return kOfxStatErrBadHandle; Value &x = PS(h)->slot(n, i); x.kind = Value::Double; x.d = v; return kOfxStatOK; }
OfxStatus propSetInt(OfxPropertySetHandle h, const char *n, int i, int v) { if (!h) return kOfxStatErrBadHandle; Value &x = PS(h)->slot(n, i); x.kind = Value::Int; x.i = v; return kOfxStatOK; }
OfxStatus propSetPointerN(OfxPropertySetHandle h, const char *n, int c, void *const *v) { for (int i = 0; i < c; i++) propSetPointer(h, n, i, v[i]); return kOfxStatOK; }
OfxStatus propSetStringN(OfxPropertySetHandle h, const char *n, int c, const char *const *v) { for (int i = 0; i < c; i++) propSetString(h, n, i, v[i]); return kOfxStatOK; }
OfxStatus propSetDoubleN(OfxPropertySetHandle h, const char *n, int c, const double *v) { for (int i = 0; i < c; i++) propSetDouble(h, n, i, v[i]); return kOfxStatOK; }
OfxStatus propSetIntN(OfxPropertySetHandle h, const char *n, int c, const int *v) { for (int i = 0; i < c; i++) propSetInt(h, n, i, v[i]); return kOfxStatOK; }
OfxStatus propGetPointer(OfxPropertySetHandle h, const char *n, int i, void **v) { if (!h) return kOfxStatErrBadHandle; const Value *x = PS(h)->find(n, i); if (!x) { if (!std::strcmp(n, kOfxPropInstanceData)) { *v = nullptr; return kOfxStatOK; } return kOfxStatErrUnknown; } *v = x->p; return kOfxStatOK; }
OfxStatus propGetString(OfxPropertySetHandle h, const char *n, int i, char **v) { if (!h) return kOfxStatErrBadHandle; const Value *x = PS(h)->find(n, i); if (!x || x->kind != Value::String) return kOfxStatErrUnknown; *v = const_cast<char *>(x->s.c_str()); return kOfxStatOK; }
OfxStatus propGetDouble(OfxPropertySetHandle h, const char *n, int i, double *v) { if (!h) return kOfxStatErrBadHandle; const Value *x = PS(h)->find(n, i); if (!x) return kOfxStatErrUnknown; *v = x->kind == Value::Int ? x->i : x->d; return kOfxStatOK; }
OfxStatus propGetInt(OfxPropertySetHandle h, const char *n, int i, int *v) { if (!h) return kOfxStatErrBadHandle; const Value *x = PS(h)->find(n, i); if (!x) return kOfxStatErrUnknown; *v = x->kind == Value::Double ? (int)x->d : x->i; return kOfxStatOK; }
OfxStatus propGetPointerN(OfxPropertySetHandle h, const char *n, int c, void **v) { for (int i = 0; i < c; i++) { OfxStatus s = propGetPointer(h, n, i, v + i); if (s) return s; } return kOfxStatOK; }
OfxStatus propGetStringN(OfxPropertySetHandle h, const char *n, int c, char **v) { for (int i = 0; i < c; i++) { OfxStatus s = propGetString(h, n, i, v + i); if (s) return s; } return kOfxStatOK; }
OfxStatus propGetDoubleN(OfxPropertySetHandle h, const char *n, int c, double *v) { for (int i = 0; i < c; i++) { OfxStatus s = propGetDouble(h, n, i, v + i); if (s) return s; } return kOfxStatOK; }
OfxStatus propGetIntN(OfxPropertySetHandle h, const char *n, int c, int *v) { for (int i = 0; i < c; i++) { OfxStatus s = propGetInt(h, n, i, v + i); if (s) return s; } return kOfxStatOK; }
OfxStatus propReset(OfxPropertySetHandle h, const char *n) { if (!h) return kOfxStatErrBadHandle; PS(h)->props.erase(n); return kOfxStatOK; }
OfxStatus propGetDimension(OfxPropertySetHandle h, const char *n, int *c) { if (!h) return kOfxStatErrBadHandle; auto it = PS(h)->props.find(n); *c = it == PS(h)->props.end() ? 0 : (int)it->second.size(); return kOfxStatOK; }
OfxPropertySuiteV1 g_prop = {propSetPointer, propSetString, propSetDouble, propSetInt, propSetPointerN, propSetStringN, propSetDoubleN, propSetIntN,
                             propGetPointer, propGetString, propGetDouble, propGetInt, propGetPointerN, propGetStringN, propGetDoubleN, propGetIntN,
                             propReset, propGetDimension};

// ---------------- image effect suite
#define EF(h) reinterpret_cast<Effect *>(h)
OfxStatus getPropertySet(OfxImageEffectHandle e, OfxPropertySetHandle *p) { if (!e) return kOfxStatErrBadHandle; *p = (OfxPropertySetHandle)&EF(e)->props; return kOfxStatOK; }
OfxStatus getParamSet(OfxImageEffectHandle e, OfxParamSetHandle *p) { if (!e) return kOfxStatErrBadHandle; *p = (OfxParamSetHandle)&EF(e)->params; return kOfxStatOK; }
OfxStatus clipDefine(OfxImageEffectHandle e, const char *name, OfxPropertySetHandle *p) {
    if (!e || !EF(e)->descriptor) return kOfxStatErrBadHandle;
    if (EF(e)->find_clip(name)) return kOfxStatErrExists;
    EF(e)->clips.emplace_back(new Clip);
    EF(e)->clips.back()->name = name;
    if (p) *p = (OfxPropertySetHandle)&EF(e)->clips.back()->props;
    return kOfxStatOK;
}
OfxStatus clipGetHandle(OfxImageEffectHandle e, const char *name, OfxImageClipHandle *c, OfxPropertySetHandle *p) {
    if (!e) return kOfxStatErrBadHandle;
    Clip *cl = EF(e)->find_clip(name);
    if (!cl) return kOfxStatErrBadHandle;
    *c = (OfxImageClipHandle)cl;
    if (p) *p = (OfxPropertySetHandle)&cl->props;
    return kOfxStatOK;
}
OfxStatus clipGetPropertySet(OfxImageClipHandle c, OfxPropertySetHandle *p) { if (!c) return kOfxStatErrBadHandle; *p = (OfxPropertySetHandle)&reinterpret_cast<Clip *>(c)->props; return kOfxStatOK; }
OfxStatus clipGetImage(OfxImageClipHandle c, OfxTime time, const OfxRectD *, OfxPropertySetHandle *img) {
    if (!c) return kOfxStatErrBadHandle;
    std::lock_guard<std::mutex> lk(g_lock);
    Clip *cl = reinterpret_cast<Clip *>(c);
    auto it = cl->images.find(time);
    if (it == cl->images.end()) return kOfxStatFailed;
    cl->fetched++;
    *img = (OfxPropertySetHandle)it->second.get();
    g_image_owner[it->second.get()] = cl;
    return kOfxStatOK;
}
OfxStatus clipReleaseImage(OfxPropertySetHandle img) {
    std::lock_guard<std::mutex> lk(g_lock);
    auto it = g_image_owner.find(PS(img));
    if (it == g_image_owner.end()) return kOfxStatErrBadHandle;
    it->second->released++;
    return kOfxStatOK;
}
OfxStatus clipGetRegionOfDefinition(OfxImageClipHandle, OfxTime, OfxRectD *) { return kOfxStatErrUnsupported; }
int effect_abort(OfxImageEffectHandle e) { return e ? EF(e)->abort_flag : 0; }
OfxStatus imageMemoryAlloc(OfxImageEffectHandle, size_t n, OfxImageMemoryHandle *h) { void *p = std::malloc(n); if (!p) return kOfxStatErrMemory; *h = (OfxImageMemoryHandle)p; return kOfxStatOK; }
OfxStatus imageMemoryFree(OfxImageMemoryHandle h) { std::free(h); return kOfxStatOK; }
OfxStatus imageMemoryLock(OfxImageMemoryHandle h, void **p) { *p = h; return kOfxStatOK; }
OfxStatus imageMemoryUnlock(OfxImageMemoryHandle) { return kOfxStatOK; }
OfxImageEffectSuiteV1 g_effect = {getPropertySet, getParamSet, clipDefine, clipGetHandle, clipGetPropertySet, clipGetImage, clipReleaseImage,
                                  clipGetRegionOfDefinition, effect_abort, imageMemoryAlloc, imageMemoryFree, imageMemoryLock, imageMemoryUnlock};

// ---------------- parameter suite
OfxStatus paramDefine(OfxParamSetHandle ps, const char *type, const char *name, OfxPropertySetHandle *p) {
    if (!ps) return kOfxStatErrBadHandle;
    ParamSet *s = reinterpret_cast<ParamSet *>(ps);
    if (s->find(name)) return kOfxStatErrExists;
    static const char *known[] = {kOfxParamTypeInteger, kOfxParamTypeDouble, kOfxParamTypeChoice, kOfxParamTypePage};
    bool ok = false;
    for (const char *k : known) ok |= !std::strcmp(k, type);
    if (!ok) return kOfxStatErrUnsupported;
    s->params.emplace_back(new Param);
    s->params.back()->name = name;
    s->params.back()->type = type;
    if (p) *p = (OfxPropertySetHandle)&s->params.back()->props;
    return kOfxStatOK;
}
OfxStatus paramGetHandle(OfxParamSetHandle ps, const char *name, OfxParamHandle *h, OfxPropertySetHandle *p) {
    if (!ps) return kOfxStatErrBadHandle;
    Param *q = reinterpret_cast<ParamSet *>(ps)->find(name);
    if (!q) return kOfxStatErrUnknown;
    *h = (OfxParamHandle)q;
    if (p) *p = (OfxPropertySetHandle)&q->props;
    return kOfxStatOK;
}
OfxStatus paramSetGetPropertySet(OfxParamSetHandle ps, OfxPropertySetHandle *p) { *p = (OfxPropertySetHandle)&reinterpret_cast<ParamSet *>(ps)->props; return kOfxStatOK; }
OfxStatus paramGetPropertySet(OfxParamHandle h, OfxPropertySetHandle *p) { *p = (OfxPropertySetHandle)&reinterpret_cast<Param *>(h)->props; return kOfxStatOK; }
OfxStatus param_get(Param *q, va_list ap) {
    if (!q) return kOfxStatErrBadHandle;
    Value v = q->value;
    if (!q->has_value) {
        const Value *d = q->props.find(kOfxParamPropDefault, 0);
        if (d) v = *d;
    }
    if (q->type == kOfxParamTypeDouble) { double *out = va_arg(ap, double *); *out = v.kind == Value::Int ? v.i : v.d; return kOfxStatOK; }
    if (q->type == kOfxParamTypeInteger || q->type == kOfxParamTypeChoice) { int *out = va_arg(ap, int *); *out = v.kind == Value::Double ? (int)v.d : v.i; return kOfxStatOK; }
    return kOfxStatErrUnsupported;
}
OfxStatus paramGetValue(OfxParamHandle h, ...) { va_list ap; va_start(ap, h); OfxStatus s = param_get(reinterpret_cast<Param *>(h), ap); va_end(ap); return s; }
OfxStatus paramGetValueAtTime(OfxParamHandle h, OfxTime t, ...) { va_list ap; va_start(ap, t); OfxStatus s = param_get(reinterpret_cast<Param *>(h), ap); va_end(ap); return s; }
OfxStatus paramUnsupportedV(OfxParamHandle, OfxTime, ...) { return kOfxStatErrUnsupported; }
OfxStatus paramGetIntegral(OfxParamHandle, OfxTime, OfxTime, ...) { return kOfxStatErrUnsupported; }
OfxStatus paramSetValue(OfxParamHandle h, ...) {
    Param *q = reinterpret_cast<Param *>(h);
    va_list ap;
    va_start(ap, h);
    if (q->type == kOfxParamTypeDouble) { q->value.kind = Value::Double; q->value.d = va_arg(ap, double); }
    else { q->value.kind = Value::Int; q->value.i = va_arg(ap, int); }
    q->has_value = true;
    va_end(ap);
    return kOfxStatOK;
}
OfxStatus paramGetNumKeys(OfxParamHandle, unsigned int *n) { *n = 0; return kOfxStatOK; }
OfxStatus paramGetKeyTime(OfxParamHandle, unsigned int, OfxTime *) { return kOfxStatErrBadIndex; }
OfxStatus paramGetKeyIndex(OfxParamHandle, OfxTime, int, int *) { return kOfxStatFailed; }
OfxStatus paramDeleteKey(OfxParamHandle, OfxTime) { return kOfxStatErrBadIndex; }
OfxStatus paramDeleteAllKeys(OfxParamHandle) { return kOfxStatOK; }
OfxStatus paramCopy(OfxParamHandle, OfxParamHandle, OfxTime, const OfxRangeD *) { return kOfxStatErrUnsupported; }
OfxStatus paramEditBegin(OfxParamSetHandle, const char *) { return kOfxStatOK; }
OfxStatus paramEditEnd(OfxParamSetHandle) { return kOfxStatOK; }
OfxParameterSuiteV1 g_param = {paramDefine, paramGetHandle, paramSetGetPropertySet, paramGetPropertySet, paramGetValue, paramGetValueAtTime,
                               paramUnsupportedV, paramGetIntegral, paramSetValue, paramUnsupportedV, paramGetNumKeys, paramGetKeyTime,
                               paramGetKeyIndex, paramDeleteKey, paramDeleteAllKeys, paramCopy, paramEditBegin, paramEditEnd};

// ---------------- memory / multithread / message suites
OfxStatus memoryAlloc(void *, size_t n, void **p) { *p = std::malloc(n); return *p ? kOfxStatOK : kOfxStatErrMemory; }
OfxStatus memoryFree(void *p) { std::free(p); return kOfxStatOK; }
OfxMemorySuiteV1 g_memory = {memoryAlloc, memoryFree};

OfxStatus mtMultiThread(OfxThreadFunctionV1 f, unsigned int n, void *arg) { for (unsigned i = 0; i < n; i++) f(i, n, arg); return kOfxStatOK; }
OfxStatus mtNumCPUs(unsigned int *n) { *n = 1; return kOfxStatOK; }
OfxStatus mtIndex(unsigned int *i) { *i = 0; return kOfxStatOK; }
int mtIsSpawned(void) { return 0; }
OfxStatus mtMutexCreate(OfxMutexHandle *m, int) { *m = (OfxMutexHandle) new std::recursive_mutex; return kOfxStatOK; }
OfxStatus mtMutexDestroy(const OfxMutexHandle m) { delete reinterpret_cast<std::recursive_mutex *>(m); return kOfxStatOK; }
OfxStatus mtMutexLock(const OfxMutexHandle m) { reinterpret_cast<std::recursive_mutex *>(m)->lock(); return kOfxStatOK; }
OfxStatus mtMutexUnLock(const OfxMutexHandle m) { reinterpret_cast<std::recursive_mutex *>(m)->unlock(); return kOfxStatOK; }
OfxStatus mtMutexTryLock(const OfxMutexHandle m) { return reinterpret_cast<std::recursive_mutex *>(m)->try_lock() ? kOfxStatOK : kOfxStatFailed; }
OfxMultiThreadSuiteV1 g_thread = {mtMultiThread, mtNumCPUs, mtIndex, mtIsSpawned, mtMutexCreate, mtMutexDestroy, mtMutexLock, mtMutexUnLock, mtMutexTryLock};

OfxStatus msgMessage(void *handle, const char *type, const char *, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (handle) EF(handle)->last_message = std::string(type ? type : "") + ": " + buf;
    return kOfxStatOK;
}
OfxMessageSuiteV1 g_message = {msgMessage};

struct HostState {
    bool hide_param_suite = false;  // to test kOfxStatErrMissingHostFeature
};
HostState g_state;
PropSet g_host_props;

const void *fetchSuite(OfxPropertySetHandle, const char *name, int version) {
    if (version != 1) return nullptr;
    if (!std::strcmp(name, kOfxImageEffectSuite)) return &g_effect;
    if (!std::strcmp(name, kOfxPropertySuite)) return &g_prop;
    if (!std::strcmp(name, kOfxParameterSuite)) return g_state.hide_param_suite ? nullptr : &g_param;
    if (!std::strcmp(name, kOfxMemorySuite)) return &g_memory;
    if (!std::strcmp(name, kOfxMultiThreadSuite)) return &g_thread;
    if (!std::strcmp(name, kOfxMessageSuite)) return &g_message;
    return nullptr;  // no interact suite: the plugins must tolerate that (inpaint.cpp:414)
}
OfxHost g_host = {(OfxPropertySetHandle)&g_host_props, fetchSuite};

struct Plugin {
    void *dl = nullptr;
    OfxPlugin *plugin = nullptr;
    int count = 0;
    std::unique_ptr<Effect> descriptor;   // after Describe + DescribeInContext
    std::string context;
};

std::string json_escape(const std::string &s) {
    std::string o;
    for (char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += c; }
        else if (c == '\n') o += "\\n";
        else if ((unsigned char)c < 32) o += ' ';
        else o += c;
    }
    return o;
}
void dump_props(std::ostringstream &o, const PropSet &p) {
    o << "{";
    bool first = true;
    for (const std::string &k : p.order) {
        auto it = p.props.find(k);
        if (it == p.props.end()) continue;
        if (!first) o << ",";
        first = false;
        o << "\"" << json_escape(k) << "\":[";
        for (size_t i = 0; i < it->second.size(); i++) {
            const Value &v = it->second[i];
            if (i) o << ",";
            switch (v.kind) {
                case Value::Int: o << v.i; break;
                case Value::Double: { char b[64]; snprintf(b, sizeof b, "%.17g", v.d); o << b; break; }
                case Value::String: o << "\"" << json_escape(v.s) << "\""; break;
                case Value::Pointer: o << "\"<pointer>\""; break;
                default: o << "null";
            }
        }
        o << "]";
    }
    o << "}";
}

Effect *clone_descriptor(const Effect &d) {
    Effect *e = new Effect;
    e->descriptor = false;
    e->props = d.props;
    e->params.props = d.params.props;
    for (auto &p : d.params.params) {
        e->params.params.emplace_back(new Param);
        Param &q = *e->params.params.back();
        q.name = p->name;
        q.type = p->type;
        q.props = p->props;
    }
    for (auto &c : d.clips) {
        e->clips.emplace_back(new Clip);
        e->clips.back()->name = c->name;
        e->clips.back()->props = c->props;
    }
    return e;
}

}  // namespace

extern "C" {

void *mh_open(const char *path) {
    Plugin *pl = new Plugin;
    pl->dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!pl->dl) { std::fprintf(stderr, "mock host: dlopen failed: %s\n", dlerror()); delete pl; return nullptr; }
    auto nfn = (int (*)(void))dlsym(pl->dl, "OfxGetNumberOfPlugins");
    auto gfn = (OfxPlugin * (*)(int)) dlsym(pl->dl, "OfxGetPlugin");
    if (!nfn || !gfn) { dlclose(pl->dl); delete pl; return nullptr; }
    pl->count = nfn();
    pl->plugin = gfn(0);
    if (!pl->plugin) { dlclose(pl->dl); delete pl; return nullptr; }
    return pl;
}
int mh_plugin_count(void *h) { return ((Plugin *)h)->count; }
int mh_get_plugin_is_null(void *h, int nth) {
    auto gfn = (OfxPlugin * (*)(int)) dlsym(((Plugin *)h)->dl, "OfxGetPlugin");
    return gfn(nth) == nullptr;
}
const char *mh_plugin_identifier(void *h) { return ((Plugin *)h)->plugin->pluginIdentifier; }
const char *mh_plugin_api(void *h) { return ((Plugin *)h)->plugin->pluginApi; }
int mh_plugin_api_version(void *h) { return ((Plugin *)h)->plugin->apiVersion; }
int mh_plugin_version(void *h, int minor) { return minor ? ((Plugin *)h)->plugin->pluginVersionMinor : ((Plugin *)h)->plugin->pluginVersionMajor; }
void mh_hide_param_suite(int hide) { g_state.hide_param_suite = hide != 0; }

// setHost (optional: pass 0 to test the missing-host path) + Load
int mh_load(void *h, int set_host) {
    Plugin *pl = (Plugin *)h;
    if (set_host) pl->plugin->setHost(&g_host);
    return pl->plugin->mainEntry(kOfxActionLoad, nullptr, nullptr, nullptr);
}
int mh_action_raw(void *h, const char *action) { return ((Plugin *)h)->plugin->mainEntry(action, nullptr, nullptr, nullptr); }

// Describe + DescribeInContext on a fresh descriptor; returns the DescribeInContext status (or the Describe one if it failed)
int mh_describe(void *h, const char *context) {
    Plugin *pl = (Plugin *)h;
    pl->descriptor.reset(new Effect);
    pl->context = context;
    int st = pl->plugin->mainEntry(kOfxActionDescribe, pl->descriptor.get(), nullptr, nullptr);
    if (st != kOfxStatOK && st != kOfxStatReplyDefault) return st;
    PropSet in;
    propSetString((OfxPropertySetHandle)&in, kOfxImageEffectPropContext, 0, context);
    return pl->plugin->mainEntry(kOfxImageEffectActionDescribeInContext, pl->descriptor.get(), (OfxPropertySetHandle)&in, nullptr);
}

// JSON: {"props":{...},"clips":[{"name":..,"props":{..}}],"params":[{"name":..,"type":..,"props":{..}}]}
int mh_dump(void *h, char *out, int cap) {
    Plugin *pl = (Plugin *)h;
    if (!pl->descriptor) return -1;
    std::ostringstream o;
    o << "{\"props\":";
    dump_props(o, pl->descriptor->props);
    o << ",\"clips\":[";
    for (size_t i = 0; i < pl->descriptor->clips.size(); i++) {
        if (i) o << ",";
        o << "{\"name\":\"" << pl->descriptor->clips[i]->name << "\",\"props\":";
        dump_props(o, pl->descriptor->clips[i]->props);
        o << "}";
    }
    o << "],\"params\":[";
    for (size_t i = 0; i < pl->descriptor->params.params.size(); i++) {
        if (i) o << ",";
        const Param &p = *pl->descriptor->params.params[i];
        o << "{\"name\":\"" << p.name << "\",\"type\":\"" << p.type << "\",\"props\":";
        dump_props(o, p.props);
        o << "}";
    }
    o << "]}";
    std::string s = o.str();
    if ((int)s.size() + 1 > cap) return -(int)s.size() - 1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

void *mh_create_instance(void *h, int *status) {
    Plugin *pl = (Plugin *)h;
    if (!pl->descriptor) { *status = kOfxStatFailed; return nullptr; }
    Effect *e = clone_descriptor(*pl->descriptor);
    propSetString((OfxPropertySetHandle)&e->props, kOfxImageEffectPropContext, 0, pl->context.c_str());
    *status = pl->plugin->mainEntry(kOfxActionCreateInstance, e, nullptr, nullptr);
    if (*status != kOfxStatOK) { delete e; return nullptr; }
    return e;
}
int mh_destroy_instance(void *h, void *inst) {
    int st = ((Plugin *)h)->plugin->mainEntry(kOfxActionDestroyInstance, inst, nullptr, nullptr);
    {
        std::lock_guard<std::mutex> lk(g_lock);
        for (auto &c : ((Effect *)inst)->clips)
            for (auto &im : c->images) g_image_owner.erase(im.second.get());
    }
    delete (Effect *)inst;
    return st;
}
int mh_set_param_double(void *inst, const char *name, double v) {
    Param *p = ((Effect *)inst)->params.find(name);
    if (!p) return -1;
    p->value.kind = p->type == kOfxParamTypeDouble ? Value::Double : Value::Int;
    p->value.d = v;
    p->value.i = (int)v;
    p->has_value = true;
    return 0;
}
int mh_get_param_secret(void *inst, const char *name) {
    Param *p = ((Effect *)inst)->params.find(name);
    if (!p) return -1;
    const Value *v = p->props.find(kOfxParamPropSecret, 0);
    return v ? v->i : 0;
}
// registers the image a clip returns at `time`
int mh_set_image(void *inst, const char *clip, double time, void *data, int x1, int y1, int x2, int y2, int row_bytes, const char *depth,
                 const char *components, double rsx, double rsy) {
    Clip *c = ((Effect *)inst)->find_clip(clip);
    if (!c) return -1;
    std::unique_ptr<PropSet> im(new PropSet);
    OfxPropertySetHandle p = (OfxPropertySetHandle)im.get();
    propSetPointer(p, kOfxImagePropData, 0, data);
    int b[4] = {x1, y1, x2, y2};
    propSetIntN(p, kOfxImagePropBounds, 4, b);
    propSetInt(p, kOfxImagePropRowBytes, 0, row_bytes);
    propSetString(p, kOfxImageEffectPropPixelDepth, 0, depth);
    propSetString(p, kOfxImageEffectPropComponents, 0, components);
    double rs[2] = {rsx, rsy};
    propSetDoubleN(p, kOfxImageEffectPropRenderScale, 2, rs);
    propSetString(p, kOfxImagePropField, 0, kOfxImageFieldNone);
    std::lock_guard<std::mutex> lk(g_lock);
    c->images[time] = std::move(im);
    return 0;
}
// the name of the pixels of the image a clip returns at `time` (kOfxImagePropUniqueIdentifier; hosts change it with the pixels)
int mh_set_image_id(void *inst, const char *clip, double time, const char *id) {
    Clip *c = ((Effect *)inst)->find_clip(clip);
    if (!c) return -1;
    std::lock_guard<std::mutex> lk(g_lock);
    auto it = c->images.find(time);
    if (it == c->images.end()) return -1;
    propSetString((OfxPropertySetHandle)it->second.get(), kOfxImagePropUniqueIdentifier, 0, id);
    return 0;
}
int mh_render(void *h, void *inst, double time, int x1, int y1, int x2, int y2, double rsx, double rsy) {
    PropSet in;
    OfxPropertySetHandle p = (OfxPropertySetHandle)&in;
    propSetDouble(p, kOfxPropTime, 0, time);
    int rw[4] = {x1, y1, x2, y2};
    propSetIntN(p, kOfxImageEffectPropRenderWindow, 4, rw);
    double rs[2] = {rsx, rsy};
    propSetDoubleN(p, kOfxImageEffectPropRenderScale, 2, rs);
    propSetString(p, kOfxImageEffectPropFieldToRender, 0, kOfxImageFieldNone);
    PropSet out;
    return ((Plugin *)h)->plugin->mainEntry(kOfxImageEffectActionRender, inst, p, (OfxPropertySetHandle)&out);
}
int mh_get_frames_needed(void *h, void *inst, double time, double *range, int *have) {
    PropSet in, out;
    propSetDouble((OfxPropertySetHandle)&in, kOfxPropTime, 0, time);
    int st = ((Plugin *)h)->plugin->mainEntry(kOfxImageEffectActionGetFramesNeeded, inst, (OfxPropertySetHandle)&in, (OfxPropertySetHandle)&out);
    const Value *a = out.find("OfxImageClipPropFrameRange_Source", 0), *b = out.find("OfxImageClipPropFrameRange_Source", 1);
    *have = a && b;
    if (*have) { range[0] = a->d; range[1] = b->d; }
    return st;
}
int mh_instance_changed(void *h, void *inst, const char *param) {
    PropSet in;
    propSetString((OfxPropertySetHandle)&in, kOfxPropType, 0, kOfxTypeParameter);
    propSetString((OfxPropertySetHandle)&in, kOfxPropName, 0, param);
    propSetString((OfxPropertySetHandle)&in, kOfxPropChangeReason, 0, kOfxChangeUserEdited);
    return ((Plugin *)h)->plugin->mainEntry(kOfxActionInstanceChanged, inst, (OfxPropertySetHandle)&in, nullptr);
}
int mh_clip_balance(void *inst, const char *clip) {  // images fetched minus released
    Clip *c = ((Effect *)inst)->find_clip(clip);
    return c ? c->fetched - c->released : -999;
}
const char *mh_last_message(void *inst) { return ((Effect *)inst)->last_message.c_str(); }
void mh_close(void *h) {
    Plugin *pl = (Plugin *)h;
    if (pl->plugin) pl->plugin->mainEntry(kOfxActionUnload, nullptr, nullptr, nullptr);
    // the .ofx stays mapped: its thread-local GPU contexts are destroyed at thread exit
    delete pl;
}

}  // extern "C"
