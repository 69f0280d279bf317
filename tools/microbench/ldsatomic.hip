#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// LDS atomic throughput: MODE 0 float add distinct, 1 float add 4 random texels x 16 ch, 2 same address, 3 u32 add distinct, 4 u32 random texel
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, int size) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < size; i += 512) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, ch = lane & 15;
    unsigned rng = threadIdx.x / 16 * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        rng = rng * 1664525u + 1013904223u;
        int addr;
        if (MODE == 0 || MODE == 3) addr = (threadIdx.x + it * 512) % size;
        else if (MODE == 1 || MODE == 4) addr = ((rng >> 8) % (size / 16)) * 16 + ch;
        else addr = 0;
        if (MODE >= 3) atomicAdd(reinterpret_cast<unsigned*>(lds) + addr, 1u);
        else atomicAdd(lds + addr, 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[17];
}
template <int MODE> void run(const char* name) {
    float* out; hipMalloc(&out, 4096);
    const int size = 23040, iters = 2000, blocks = 512;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, size * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 512, size * 4>>>(out, 10, size);
    hipEventRecord(a);
    k<MODE><<<blocks, 512, size * 4>>>(out, iters, size);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 2 blocks sequentially (1 per CU by LDS) -> lane-ops per CU = 2 * 512 * iters
    double per_cu_ops = 2.0 * 512 * iters;
    printf("%-28s %.3f ms  -> %.2f lane-atomics / cycle / CU (2.4 GHz)\n", name, ms, per_cu_ops / (ms * 1e-3 * 2.4e9));
}
int main() {
    run<0>("f32 distinct");
    run<1>("f32 4 random texels x16");
    run<2>("f32 same address");
    run<3>("u32 distinct");
    run<4>("u32 4 random texels x16");
    return 0;
}
