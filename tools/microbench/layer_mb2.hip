#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define BMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
struct Split3 { u32x4 h, m, l; };
__device__ __forceinline__ unsigned top16_pair(float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }
__device__ __forceinline__ float below_top16(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ Split3 split_acc(const floatx16& a, int jj) {
    float x[8], r1[8], r2[8]; Split3 s;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaxf(a[8 * jj + e], 0.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.h[i] = top16_pair(x[2 * i], x[2 * i + 1]);
#pragma unroll
    for (int e = 0; e < 8; ++e) r1[e] = below_top16(x[e]);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.m[i] = top16_pair(r1[2 * i], r1[2 * i + 1]);
#pragma unroll
    for (int e = 0; e < 8; ++e) r2[e] = below_top16(r1[e]);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.l[i] = top16_pair(r2[2 * i], r2[2 * i + 1]);
    return s;
}
__device__ __forceinline__ floatx16 mma_split(const Split3& a, const Split3& b, floatx16 c) {
    c = BMFMA(a.m, b.m, c); c = BMFMA(a.l, b.h, c); c = BMFMA(a.h, b.l, c);
    c = BMFMA(a.m, b.h, c); c = BMFMA(a.h, b.m, c); c = BMFMA(a.h, b.h, c);
    return c;
}
__device__ __forceinline__ Split3 sfrag_load(const u32x4* packed, int ib, int J, int j, int lane) {
    const u32x4* q = packed + (size_t)((ib * J + j) * 3) * 64 + lane;
    Split3 s; s.h = q[0]; s.m = q[64]; s.l = q[128]; return s;
}

// V: 0 = constant operands, 1 = + fragment loads, 2 = + on-the-fly split.  CB point blocks (32 points each) per wave,
// WPS waves per SIMD asked of the compiler.
template <int V, int CB, int WPS>
__global__ void __launch_bounds__(64, WPS) k(float* out, const u32x4* w, int layers) {
    const int lane = threadIdx.x;
    floatx16 x[CB][4], acc[CB][4];
    for (int cb = 0; cb < CB; ++cb) for (int nb = 0; nb < 4; ++nb) for (int r = 0; r < 16; ++r) { x[cb][nb][r] = 0.01f * (lane + r + nb) - 0.3f; acc[cb][nb][r] = 0.f; }
    Split3 c0 = sfrag_load(w, 0, 8, 0, lane);
    for (int L = 0; L < layers; ++L) {
        const u32x4* packed = w + (size_t)L * 4 * 8 * 3 * 64;
        Split3 f[3], xs[2][CB];
        if (V >= 1) { f[0] = sfrag_load(packed, 0, 8, 0, lane); f[1] = sfrag_load(packed, 1, 8, 0, lane); }
        else { f[0] = f[1] = f[2] = c0; }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) xs[0][cb] = xs[1][cb] = (V == 2) ? split_acc(x[cb][0], 0) : c0;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int j = t >> 2, nb = t & 3;
            if (V >= 1 && t + 2 < 32) f[(t + 2) % 3] = sfrag_load(packed, (t + 2) & 3, 8, (t + 2) >> 2, lane);
            if (V == 2 && j + 1 < 8 && nb < CB) xs[(j + 1) & 1][nb] = split_acc(x[nb][(j + 1) >> 1], (j + 1) & 1);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[cb][nb] = mma_split(f[t % 3], xs[j & 1][cb], acc[cb][nb]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V == 2) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) for (int nb = 0; nb < 4; ++nb) x[cb][nb] = acc[cb][nb] * 1e-3f;
        }
    }
    float s = 0;
    for (int cb = 0; cb < CB; ++cb) for (int nb = 0; nb < 4; ++nb) for (int r = 0; r < 16; ++r) s += acc[cb][nb][r];
    out[blockIdx.x * 64 + lane] = s;
}
template <int V, int CB, int WPS>
void run(const char* name, float* out, const u32x4* w, int layers, int points) {
    const int blocks = points / (32 * CB);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, CB, WPS>), dim3(blocks), dim3(64), 0, 0, out, w, layers);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double mf = (double)blocks * layers * 192 * CB;
    printf("%-40s %.3f ms  cycles/MFMA/SIMD@2.4GHz %.1f\n", name, ms, ms * 1e-3 * 2.4e9 / (mf / 1024));
}
int main() {
    const int layers = 11, points = 4688 * 64;
    float* out; (void)hipMalloc(&out, (size_t)points * 2 * 4);
    size_t wn = (size_t)layers * 4 * 8 * 3 * 64 * 4;
    unsigned* hw = (unsigned*)malloc(wn * 4);
    for (size_t i = 0; i < wn; ++i) hw[i] = 0x3c003c00u + (unsigned)((i * 2654435761u) & 0x007f007fu);
    u32x4* w; (void)hipMalloc(&w, wn * 4); (void)hipMemcpy(w, hw, wn * 4, hipMemcpyHostToDevice);
    run<0, 2, 1>("64 pts/wave, 1 wave/SIMD, constants", out, w, layers, points);
    run<1, 2, 1>("64 pts/wave, 1 wave/SIMD, +loads", out, w, layers, points);
    run<2, 2, 1>("64 pts/wave, 1 wave/SIMD, +loads+split", out, w, layers, points);
    run<0, 1, 2>("32 pts/wave, 2 waves/SIMD, constants", out, w, layers, points);
    run<1, 1, 2>("32 pts/wave, 2 waves/SIMD, +loads", out, w, layers, points);
    run<2, 1, 2>("32 pts/wave, 2 waves/SIMD, +loads+split", out, w, layers, points);
    run<2, 1, 1>("32 pts/wave, 1 wave/SIMD, +loads+split", out, w, layers, points);
    run<2, 1, 3>("32 pts/wave, 3 waves/SIMD, +loads+split", out, w, layers, points);
    return 0;
}
