// l1rate.hip -- vector-memory instruction throughput per CU on gfx950 for the access shapes the Farneback
// kernels use (L2-resident source, so the limit seen is the TA/L1 path, not HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct __attribute__((packed, aligned(4))) P2 { float a, b; };
struct __attribute__((packed, aligned(4))) P4 { float a, b, c, d; };

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int iters, int span_floats) {
    // each wave walks its own 4 KiB-aligned region; lane offsets per MODE
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const float* base = src + (size_t)(wave % 256) * 1024;
    float acc = 0;
    int off = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { acc += base[off + lane]; }                                         // dword coalesced
        if (MODE == 1) { float2 v = *(const float2*)(base + off + lane * 2); acc += v.x + v.y; }       // dwordx2 aligned
        if (MODE == 2) { float4 v = *(const float4*)(base + off + lane * 4); acc += v.x + v.y + v.z + v.w; } // dwordx4 aligned
        if (MODE == 3) { P2 v = *(const P2*)(base + off + lane + 1); acc += v.a + v.b; }    // dwordx2, lane stride 4 B (overlapping pairs), misaligned
        if (MODE == 4) { acc += base[off + lane + 1]; }                                     // dword misaligned by 4 B
        if (MODE == 5) { P4 v = *(const P4*)(base + off + lane + 1); acc += v.a + v.d; }    // dwordx4, lane stride 4 B
        if (MODE == 6) { acc += base[off + lane * 5]; }                                     // dword stride 20 B (AoS5 one channel)
        if (MODE == 7) { P4 v = *(const P4*)(base + off + lane * 5); acc += v.a + v.d; }    // dwordx4 stride 20 B (AoS5)
        off = (off + 256) & (span_floats - 1);
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE> void run(const char* name, const float* d, float* o, int bytes_per_lane) {
    int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, 64, 512);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, iters, 512);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters;
    double clk = ms * 1e-3 * 2.1e9;  // assume ~2.1 GHz sustained
    printf("%-46s %8.3f ms  %6.1f clk/wave-instr/CU  %6.1f B/clk/CU  %7.2f TB/s\n", name, ms, clk / (winstr / 256), winstr * 64 * bytes_per_lane / 256 / clk,
           winstr * 64 * bytes_per_lane / ms / 1e9);
}
int main() {
    float *d, *o; hipMalloc(&d, 16 << 20); hipMalloc(&o, 4096); hipMemset(d, 0, 16 << 20);
    run<0>("dword, coalesced", d, o, 4);
    run<1>("dwordx2, coalesced aligned", d, o, 8);
    run<2>("dwordx4, coalesced aligned", d, o, 16);
    run<4>("dword, coalesced, +4 B misaligned", d, o, 4);
    run<3>("dwordx2, lane stride 4 B (gather pairs)", d, o, 8);
    run<5>("dwordx4, lane stride 4 B", d, o, 16);
    run<6>("dword, lane stride 20 B", d, o, 4);
    run<7>("dwordx4, lane stride 20 B (AoS5)", d, o, 16);
    return 0;
}
