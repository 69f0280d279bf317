#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template<int NACC, int WPS>
__global__ void __launch_bounds__(64, WPS) k(float* out, const u32x4* in, int iters) {
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a = in[threadIdx.x], b = in[64 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 65536 * 64 * 4);
    u32x4* in; hipMalloc(&in, 128 * 16);
    unsigned h[512]; for (int i = 0; i < 512; ++i) h[i] = 0x3f803f80u + ((i * 2654435761u) & 0x007f007fu);   // bf16 values in [1,2)
    hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int total = 2592;
    for (int cfg = 0; cfg < 6; ++cfg) {
        int nacc = (cfg % 3 == 0) ? 1 : (cfg % 3 == 1) ? 2 : 8;
        int wps = cfg < 3 ? 1 : 2;
        int blocks = 4688;
        int iters = total / (8 * nacc);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (wps == 1) {
                if (nacc == 1) hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
                else if (nacc == 2) hipLaunchKernelGGL((k<2, 1>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
                else hipLaunchKernelGGL((k<8, 1>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
            } else {
                if (nacc == 1) hipLaunchKernelGGL((k<1, 2>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
                else if (nacc == 2) hipLaunchKernelGGL((k<2, 2>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
                else hipLaunchKernelGGL((k<8, 2>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mf = (double)blocks * iters * 8 * nacc;
            if (rep == 2) printf("nacc %d waves/simd %d blocks %d: %.3f ms  %.1f TFLOP/s  cycles/MFMA/SIMD@2.4GHz %.1f\n", nacc, wps, blocks, ms, mf * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
        }
    }
    return 0;
}
