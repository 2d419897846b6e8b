// valurate_fp.hip -- issue rate of the floating-point vector instructions the Farneback iteration kernels are built from (gfx950):
// f32 add / mul / fma, packed f32, f64 add / mul / fma, a DPP move, f32 <-> f64 conversion.  Same scheme as valurate.hip: long chains of one
// instruction (8 independent chains per lane), 4 waves per SIMD, inline asm so the compiler cannot fold or re-associate anything.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[8];
    double da[8];
    for (int i = 0; i < 8; i++) {
        a[i] = seed + threadIdx.x * 0.001f + i;
        da[i] = a[i];
    }
    float b = seed * 1.0001f;
    double db = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[i]) : "v"(db));
            if (MODE == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da[i]) : "v"(db));
            if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(da[i]) : "v"(db));
            if (MODE == 6) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(da[i]) : "v"(a[i]));
            if (MODE == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(da[i]) : "v"(db));
            if (MODE == 9) asm volatile("v_rcp_f64 %0, %0" : "+v"(da[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (float)da[i];
    if (s == 0.12345f) out[0] = s;
}

template <int MODE> void run(const char *name, float *o) {
    const int iters = 2048, blocks = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, 16, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 8;             // wave-instructions issued
    const double clk = ms * 1e-3 * 2.4e9;                              // assume 2.4 GHz
    printf("%-34s %7.3f ms  %5.2f clk per wave-instruction per SIMD\n", name, ms, clk / (winstr / 1024.0));
}

int main() {
    float *o; hipMalloc(&o, 64);
    run<0>("v_add_f32", o);
    run<1>("v_mul_f32", o);
    run<2>("v_fma_f32", o);
    run<3>("v_add_f64", o);
    run<4>("v_mul_f64", o);
    run<5>("v_fma_f64", o);
    run<6>("v_mov_b32_dpp wave_shr:1", o);
    run<7>("v_cvt_f64_f32", o);
    run<8>("v_pk_fma_f32 (2 lanes of f32)", o);
    run<9>("v_rcp_f64", o);
    return 0;
}
