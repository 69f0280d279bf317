// VALU issue rate on gfx950: plain v_fma_f32 vs v_pk_fma_f32 at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[8]; float2_ y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = float2_{x[i], x[i] + 1.f}; }
    float2_ a2{a, a}, b2{b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
                else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                else if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
                else asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int MODE> void run(const char* name, float* d) {
    for (int bpc : {1, 2, 4, 8}) {
        int iters = 4000; dim3 grid(256 * bpc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<grid, 256>>>(d, 100, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0); k<MODE><<<grid, 256>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double insts_per_simd = (double)bpc * iters * 64.0;   // waves per SIMD x instructions per wave
        double cyc = ms * 1e-3 * 2.4e9;
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles(2.4GHz) per wave-instruction per SIMD\n", name, bpc, ms, cyc / insts_per_simd);
    }
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_exp_f32", d); run<3>("v_pk_mul_f32", d); run<4>("v_mov_dpp", d);
    return 0;
}
