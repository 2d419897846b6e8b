// valurate.hip -- issue rate of the integer vector instructions the mean-shift tap loop can be built from (gfx950).
// Each wave runs long chains of one instruction (8 independent chains per lane); 4 waves per SIMD keep the pipe full.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i;
    unsigned b = seed * 3 + 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = a[i] + b;                                                      // v_add_u32
            if (MODE == 1) a[i] = a[i] * b;                                                      // v_mul_lo_u32
            if (MODE == 2) a[i] = (unsigned)__mul24((int)a[i], (int)b);                          // v_mul_i32_i24
            if (MODE == 3) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);                  // v_dot4_u32_u8
            if (MODE == 4) a[i] = (a[i] >> 8) & 255u;                                            // v_bfe_u32
            if (MODE == 5) a[i] = a[i] > b ? a[i] - 1 : b;                                       // v_cmp + v_cndmask (+ sub)
            if (MODE == 6) a[i] = (unsigned)__builtin_amdgcn_sad_u8(a[i], b, a[i]);              // v_sad_u8
            if (MODE == 7) a[i] = (unsigned)__mul24((int)a[i], (int)b) + a[i];                   // v_mad_i32_i24
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) out[0] = s;
}

template <int MODE> void run(const char *name, unsigned *o, int per) {
    const int iters = 2048, blocks = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, 16, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 8 * per;       // wave-instructions issued
    const double clk = ms * 1e-3 * 2.4e9;                              // assume 2.4 GHz
    printf("%-34s %7.3f ms  %5.2f clk per wave-instruction per SIMD\n", name, ms, clk / (winstr / 1024.0));
}

int main() {
    unsigned *o; hipMalloc(&o, 64);
    run<0>("v_add_u32", o, 1);
    run<1>("v_mul_lo_u32", o, 1);
    run<2>("v_mul_i32_i24", o, 1);
    run<3>("v_dot4_u32_u8", o, 1);
    run<4>("v_bfe_u32 (shift+and)", o, 1);
    run<5>("cmp + cndmask + sub (3 instr)", o, 3);
    run<6>("v_sad_u8", o, 1);
    run<7>("v_mad_i32_i24", o, 1);
    return 0;
}
