#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template<int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f + 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nacc : {1, 4}) for (int blocks : {1024, 2344}) {
        int total = 1728; int iters = total / (8 * nacc);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mf = (double)blocks * 4 * iters * 8 * nacc;
            printf("nacc %d blocks %d: %.3f ms  %.1f TFLOP/s  cycles/MFMA/SIMD@2.4GHz %.1f\n", nacc, blocks, ms, mf * 4096 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
        }
    }
    return 0;
}
