// repro_graph_launch.hip -- does the HIP runtime itself survive concurrent hipGraphLaunch calls of several host threads on ONE device?
// No code of libofxcv_hip: every thread has its own two streams, captures ONE graph shaped like a Farneback call (a fork to a second stream,
// ~100 small kernels, event joins), and replays it in a loop (hipGraphLaunch + hipStreamSynchronize).  With --lock the launches are serialised
// by a mutex (what the library's process-wide runtime lock does with farneback.graph 1).  A watchdog reports threads that stopped making
// progress and exits 3 after --stall seconds without any; a crash shows as a signal.  (is the round-5 soak that did
// not return -- OFXCV_LOCK_PER_DEVICE=1 over OFXCV_VIRTUAL_DEVICES -- the runtime or the library's shared state?)
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/repro_graph_launch tools/repro_graph_launch.hip -lpthread
// run:   timeout -s KILL 120 /tmp/repro_graph_launch [--threads 8] [--seconds 20] [--lock] [--nodes 100] [--eager]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            std::_Exit(2);                                                                      \
        }                                                                                       \
    } while (0)

__global__ void busy(float *p, int n, int spin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int k = 0; k < spin; k++) v = v * 1.0001f + 0.5f;
    p[i] = v;
}

static std::mutex g_launch;
static std::atomic<long> g_done[64];
static std::atomic<bool> g_stop{false};

static void enqueue(hipStream_t s, hipStream_t sp, hipEvent_t fork, hipEvent_t join, float *a, float *b, int nodes) {
    CK(hipEventRecord(fork, s));
    CK(hipStreamWaitEvent(sp, fork, 0));
    for (int k = 0; k < nodes / 4; k++) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, sp, b, 65536, 20);
    CK(hipEventRecord(join, sp));
    for (int k = 0; k < nodes / 4; k++) hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, s, a, 16384, 50);
    CK(hipStreamWaitEvent(s, join, 0));
    for (int k = 0; k < nodes / 2; k++) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s, a, 262144, 10);
}

static void worker(int id, int nodes, bool lock, bool eager, bool capture_each) {
    CK(hipSetDevice(0));
    hipStream_t s, sp;
    hipEvent_t fork, join;
    {
        std::lock_guard<std::mutex> lk(g_launch);  // (creation is serialised in every mode: the question is the launch)
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    }
    float *a, *b;
    {
        // (allocations and the legacy-stream memsets under the mutex too: beside another thread's capture they fail with "operation would make the
        // legacy stream depend on a capturing blocking stream" -- the first thing this program ran into, and why the library allocates under its lock)
        std::lock_guard<std::mutex> lk(g_launch);
        CK(hipMalloc(&a, 262144 * 4));
        CK(hipMalloc(&b, 65536 * 4));
        CK(hipMemset(a, 0, 262144 * 4));
        CK(hipMemset(b, 0, 65536 * 4));
        CK(hipDeviceSynchronize());
    }
    hipGraphExec_t exec = nullptr;
    auto capture = [&]() {
        std::lock_guard<std::mutex> lk(g_launch);  // (captures are serialised in every mode as well)
        if (exec) CK(hipGraphExecDestroy(exec));
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        enqueue(s, sp, fork, join, a, b, nodes);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    };
    if (!eager) capture();
    long it = 0;
    while (!g_stop.load(std::memory_order_relaxed)) {
        if (eager) {
            enqueue(s, sp, fork, join, a, b, nodes);
        } else {
            if (capture_each && (it % 16) == 15) capture();
            if (lock) {
                std::lock_guard<std::mutex> lk(g_launch);
                CK(hipGraphLaunch(exec, s));
            } else {
                CK(hipGraphLaunch(exec, s));
            }
        }
        CK(hipStreamSynchronize(s));
        g_done[id].fetch_add(1, std::memory_order_relaxed);
        it++;
    }
}

int main(int argc, char **argv) {
    int threads = 8, nodes = 100, seconds = 20, stall = 15;
    bool lock = false, eager = false, capture_each = false;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--threads")) threads = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--nodes")) nodes = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--seconds")) seconds = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--stall")) stall = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--lock")) lock = true;
        else if (!std::strcmp(argv[i], "--eager")) eager = true;
        else if (!std::strcmp(argv[i], "--recapture")) capture_each = true;
    }
    if (threads > 64) threads = 64;
    int rt = 0;
    CK(hipRuntimeGetVersion(&rt));
    std::printf("HIP runtime %d; %d threads, %d nodes per graph, %s, %d s\n", rt, threads, nodes,
                eager ? "eager launches" : (lock ? "hipGraphLaunch under one mutex" : "hipGraphLaunch concurrent"), seconds);
    std::vector<std::thread> th;
    for (int i = 0; i < threads; i++) th.emplace_back(worker, i, nodes, lock, eager, capture_each);
    long last[64] = {0};
    int quiet = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int sec = 0; sec < seconds; sec++) {
        std::this_thread::sleep_for(std::chrono::seconds(1));
        bool any = false;
        int stuck = 0;
        for (int i = 0; i < threads; i++) {
            const long d = g_done[i].load();
            if (d != last[i]) any = true;
            else stuck++;
            last[i] = d;
        }
        quiet = any ? 0 : quiet + 1;
        if (stuck && sec > 1) std::printf("  t=%2d s: %d of %d threads made no progress in the last second\n", sec + 1, stuck, threads);
        if (quiet >= stall) {
            std::printf("STALLED: no thread has completed a replay for %d s\n", stall);
            std::fflush(stdout);
            std::_Exit(3);
        }
    }
    g_stop.store(true);
    for (auto &t : th) t.join();
    long total = 0;
    for (int i = 0; i < threads; i++) total += g_done[i].load();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("ok: %ld replays in %.1f s (%.0f per second)\n", total, el, total / el);
    return 0;
}
