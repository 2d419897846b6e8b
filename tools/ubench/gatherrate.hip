// gatherrate.hip -- what a dword wave-load costs the TA / L1 when the lanes' addresses are ALMOST consecutive: the shape of the R1 taps of
// FarnebackUpdateMatrices on a smooth flow (x1 of neighbouring lanes differs by one except at a few jumps; a few lanes on the row below).
// The lane -> offset table comes from memory so the compiler knows nothing about the pattern.  L2-resident source.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WIDTH, bool SPARSE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, const int* __restrict__ pat, float* out, int iters, int span_floats) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const float* base = src + (size_t)(wave % 64) * 8192;
    const int lo = pat[lane];
    float acc = 0;
    int off = 0;
    for (int it = 0; it < iters; it++) {
        if (!SPARSE || pat[64 + lane]) {
            acc += base[off + lo];
        }
        off = (off + 256) & (span_floats - 1);
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <bool SPARSE> void run(const char* name, const float* d, const int* dp, float* o, const std::vector<int>& pat) {
    hipMemcpy((void*)dp, pat.data(), 128 * sizeof(int), hipMemcpyHostToDevice);
    int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<1, SPARSE>), dim3(blocks), dim3(256), 0, 0, d, dp, o, 64, 2048);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<1, SPARSE>), dim3(blocks), dim3(256), 0, 0, d, dp, o, iters, 2048);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters;
    double clk = ms * 1e-3 * 2.1e9;  // assume ~2.1 GHz sustained
    printf("%-64s %8.3f ms  %6.1f clk/wave-instr/CU\n", name, ms, clk / (winstr / 256));
}
int main() {
    float *d, *o; int* dp;
    hipMalloc(&d, 16 << 20); hipMalloc(&o, 4096); hipMalloc(&dp, 1024); hipMemset(d, 0, 16 << 20);
    std::vector<int> p(128, 1);
    auto fill = [&](auto f) { for (int l = 0; l < 64; l++) p[l] = f(l); };
    fill([](int l) { return l; });                          run<false>("consecutive lanes", d, dp, o, p);
    fill([](int l) { return l + 1; });                      run<false>("consecutive, +4 B misaligned", d, dp, o, p);
    fill([](int l) { return l + (l >= 32); });              run<false>("consecutive with ONE jump of +1 at lane 32", d, dp, o, p);
    fill([](int l) { return l + (l >= 37); });              run<false>("consecutive with ONE jump of +1 at lane 37", d, dp, o, p);
    fill([](int l) { return l - (l >= 37); });              run<false>("consecutive with ONE repeat (jump of 0) at lane 37", d, dp, o, p);
    fill([](int l) { return l + l / 16; });                 run<false>("consecutive with a jump every 16 lanes", d, dp, o, p);
    fill([](int l) { return l + l / 4; });                  run<false>("consecutive with a jump every 4 lanes", d, dp, o, p);
    fill([](int l) { return l + (l >= 40 ? 1920 : 0); });   run<false>("consecutive, lanes >= 40 on the next row (+1920 floats)", d, dp, o, p);
    fill([](int l) { return l + ((l / 8) & 1) * 1920; });   run<false>("consecutive, groups of 8 lanes alternate between two rows", d, dp, o, p);
    fill([](int l) { return l * 5; });                      run<false>("lane stride 20 B", d, dp, o, p);
    fill([](int l) { return (l * 37) & 1023; });            run<false>("scattered (37 * lane mod 1024)", d, dp, o, p);
    fill([](int l) { return l; });
    for (int l = 0; l < 64; l++) p[64 + l] = (l == 5 || l == 41);   run<true>("2 active lanes of 64 (exec-masked)", d, dp, o, p);
    for (int l = 0; l < 64; l++) p[64 + l] = (l % 8 == 0);          run<true>("8 active lanes of 64 (exec-masked)", d, dp, o, p);
    for (int l = 0; l < 64; l++) p[64 + l] = (l < 16);              run<true>("lanes 0..15 active (exec-masked)", d, dp, o, p);
    return 0;
}
