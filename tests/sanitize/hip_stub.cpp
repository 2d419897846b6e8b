// hip_stub.cpp -- a stand-in for the HIP runtime, TEST INFRASTRUCTURE ONLY (tests/sanitize): "device" memory is host memory, streams run
// everything at once in the calling thread, events are flags, kernel launches do nothing.  It lets the HOST side of libofxcv_hip -- contexts,
// scratch management, the named-frame cache, the submission queue, the level plan and launch sequence, the Telea front march -- and of the
// three OFX plugins run on a CPU under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer (SURVEY section 5's plan; the reference
// has only assert()s: VectorGenerator.cpp:400-401).  Results are meaningless (no kernel runs); memory and thread safety of the host code is
// what is checked.  Never linked into the product: the Makefile beside it builds private copies of the objects with --cuda-host-only.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <chrono>

namespace {
struct Stream { int id; };
struct Event { std::atomic<int> recorded{0}; };
thread_local hipError_t g_last = hipSuccess;
thread_local struct { dim3 g, b; size_t sh; hipStream_t s; } g_cfg;
std::atomic<int> g_streams{0};
}  // namespace

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) {
    *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : (a == hipDeviceAttributeMaxSharedMemoryPerBlock ? 160 * 1024 : 0);
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) {
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t) new Stream{g_streams.fetch_add(1)}; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipStreamDestroy(hipStream_t s) { delete (Stream *)s; return hipSuccess; }
// (a wait takes a moment, as on a device: without it calls are over before the next render thread has entered and nothing ever coalesces)
static void device_time() { std::this_thread::sleep_for(std::chrono::microseconds(300)); }
hipError_t hipStreamSynchronize(hipStream_t) { device_time(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return g_last = hipErrorNotSupported; }  // (the library falls back to plain launches)
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return g_last = hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, hipGraphNode_t *, char *, size_t) { return g_last = hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return g_last = hipErrorNotSupported; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t) new Event(); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { delete (Event *)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { ((Event *)e)->recorded.store(1); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { device_time(); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : (g_last = hipErrorOutOfMemory); }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t y = 0; y < h; y++) std::memmove((char *)d + y * dp, (const char *)s + y * sp, w);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorNotSupported ? "not supported (stub runtime)" : "error (stub runtime)"); }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { return hipSuccess; }  // no device: nothing runs
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s) { g_cfg.g = g; g_cfg.b = b; g_cfg.sh = sh; g_cfg.s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *sh, hipStream_t *s) { *g = g_cfg.g; *b = g_cfg.b; *sh = g_cfg.sh; *s = g_cfg.s; return hipSuccess; }
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipUnregisterFatBinary(void **) {}
}
