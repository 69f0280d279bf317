#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.999f;
    float s[8]; float2v p[8];
    for (int i = 0; i < 8; ++i) { s[i] = i * 0.1f; p[i] = (float2v){i * 0.1f, i * 0.2f}; }
    float2v a2 = {a, a * 1.01f}, b2 = {b, b * 0.99f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(a), "v"(b));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(a2), "v"(b2));
            }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += s[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4096, blocks = 2048;
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)blocks * 4 * iters * 32;     // wave-instructions
        if (rep) printf("mode %s: %.3f ms, cycles per wave-instruction per SIMD @2.4GHz: %.2f\n", mode ? "v_pk_fma_f32" : "v_fma_f32", ms, ms * 1e-3 * 2.4e9 / (instr / 1024));
    }
    return 0;
}
