#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// MODE 0 f32 rtn, 1 f64, 2 u64, 3 f32 no-rtn (baseline), 4 global f32 random texels (region floats), 5 plain RMW (non atomic)
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, float* gbuf, int iters, int size, int gsize) {
    extern __shared__ double ldsd[];
    float* lds = reinterpret_cast<float*>(ldsd);
    for (int i = threadIdx.x; i < size * 2; i += 512) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, ch = lane & 15;
    unsigned rng = threadIdx.x / 16 * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        rng = rng * 1664525u + 1013904223u;
        const int addr = ((rng >> 8) % (size / 16)) * 16 + ch;
        if (MODE == 0) acc += atomicAdd(lds + addr, 1.0f);
        else if (MODE == 1) atomicAdd(ldsd + addr, 1.0);
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(ldsd) + addr, 1ull);
        else if (MODE == 3) atomicAdd(lds + addr, 1.0f);
        else if (MODE == 4) atomicAdd(gbuf + ((rng >> 4) % (gsize / 16)) * 16 + ch, 1.0f);
        else { lds[addr] += 1.0f; }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[17] + acc;
}
template <int MODE> void run(const char* name) {
    float* out; (void)hipMalloc(&out, 4096);
    const int gsize = 3 * 128 * 128 * 16;
    float* gbuf; (void)hipMalloc(&gbuf, gsize * 4); (void)hipMemset(gbuf, 0, gsize * 4);
    const int size = 8064, iters = 2000, blocks = 512;
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, size * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<blocks, 512, size * 8>>>(out, gbuf, 10, size, gsize);
    (void)hipEventRecord(a);
    k<MODE><<<blocks, 512, size * 8>>>(out, gbuf, iters, size, gsize);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double per_cu_ops = 2.0 * 512 * iters;
    printf("%-28s %.3f ms  -> %.2f lane-ops / cycle / CU (2.4 GHz)\n", name, ms, per_cu_ops / (ms * 1e-3 * 2.4e9));
}
int main() {
    run<3>("lds f32 no-rtn");
    run<0>("lds f32 rtn");
    run<1>("lds f64");
    run<2>("lds u64");
    run<4>("global f32 random texel");
    run<5>("lds plain rmw (racy)");
    return 0;
}
