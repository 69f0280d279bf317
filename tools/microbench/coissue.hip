// Does a SIMD run vector-ALU instructions while its matrix pipe works on an MFMA?
// One wave issues, in a fixed program order, one v_mfma_f32_32x32x16_bf16 followed by KV independent v_fma_f32
// (registers disjoint from the MFMA's), 1, 2 or 4 waves per SIMD.  If the two overlap, cycles per MFMA stay flat
// until the KV instructions fill the MFMA's shadow; if they do not, every v_fma adds its issue time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: accumulators in vector registers; 1: in accumulation registers (a[..]); 2: two 16x16x32 MFMAs (the same
// flop count, 4-register accumulators) in place of one 32x32x16
#define MFMA_V(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFMA_16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
typedef float floatx4 __attribute__((ext_vector_type(4)));
// MODE 5 / 6: every MFMA reads another pair of operand registers (eight pairs in turn), held in v / in a registers
#define MFMA_AV(acc, x, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y))
#define MFMA_AA(acc, x, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(x), "a"(y))
#define FMA(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z))

template <int KV, int WPS, int MODE>
__global__ void __launch_bounds__(64, WPS) k(float* out, const u32x4* in, int iters) {
    floatx16 acc[4];
    floatx4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    u32x4 a = in[threadIdx.x], b = in[64 + threadIdx.x];
    u32x4 av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = a + (unsigned)i; bv[i] = b + (unsigned)(i << 16); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i);
    float y = 1.0f + 1e-7f * threadIdx.x, z = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) MFMA_V(acc[u & 3]);
            else if (MODE == 1) MFMA_A(acc[u & 3]);
            else if (MODE == 3) MFMA_A(acc[0]);              // every MFMA accumulates into the previous one's result
            else if (MODE == 4) MFMA_A(acc[u & 1]);          // two accumulation chains, alternating
            else if (MODE == 5) MFMA_AV(acc[u & 3], av[u], bv[u]);
            else if (MODE == 6) MFMA_AA(acc[u & 3], av[u], bv[u]);
            else MFMA_16(acc4[(2 * u) & 7]);
#pragma unroll
            for (int v = 0; v < KV / 2; ++v) FMA(x[v & 7]);
            if (MODE == 2) MFMA_16(acc4[(2 * u + 1) & 7]);
#pragma unroll
            for (int v = KV / 2; v < KV; ++v) FMA(x[v & 7]);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// VALU only (no MFMA): the cost of the KV instructions alone
template <int KV, int WPS>
__global__ void __launch_bounds__(64, WPS) kv(float* out, int iters) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i);
    float y = 1.0f + 1e-7f * threadIdx.x, z = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int v = 0; v < KV; ++v) FMA(x[v & 7]);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

static float* out;
static u32x4* in;
static hipEvent_t e0, e1;

template <int KV, int WPS, int MODE = 0>
static void run(int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 3000;
    float ms = 0, msv = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KV, WPS, MODE>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (KV > 0) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((kv<KV, WPS>), dim3(blocks), dim3(64), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&msv, e0, e1);
        }
    }
    const double mfma_per_simd = (double)waves_per_simd * iters * 8;
    printf("%s  waves/SIMD %d  v_fma per MFMA %2d: %.3f ms = %5.1f cycles per MFMA (2.4 GHz);  the v_fma alone: %5.1f cycles\n",
           MODE == 0 ? "32x32x16 acc in v" : MODE == 1 ? "32x32x16 acc in a" : MODE == 2 ? "2 x 16x16x32     " : MODE == 3 ? "acc in a, 1 chain" : MODE == 4 ? "acc in a, 2 chain" : MODE == 5 ? "8 operand pairs v" : "8 operand pairs a", waves_per_simd, KV, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd, msv * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
    hipMalloc(&out, 4096 * 64 * 4);
    hipMalloc(&in, 128 * 16);
    unsigned h[512];
    for (int i = 0; i < 512; ++i) h[i] = 0x3f803f80u + ((i * 2654435761u) & 0x007f007fu);   // bf16 values in [1,2)
    hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
    hipEventCreate(&e0); hipEventCreate(&e1);
    run<0, 1>(1); run<2, 1>(1); run<4, 1>(1); run<6, 1>(1); run<8, 1>(1); run<12, 1>(1); run<16, 1>(1);
    run<0, 2>(2); run<4, 2>(2); run<8, 2>(2); run<12, 2>(2);
    run<0, 4>(4); run<8, 4>(4); run<12, 4>(4);
    run<0, 1, 1>(1); run<4, 1, 1>(1); run<8, 1, 1>(1); run<12, 1, 1>(1); run<16, 1, 1>(1);
    run<0, 2, 1>(2); run<4, 2, 1>(2); run<8, 2, 1>(2); run<12, 2, 1>(2);
    run<0, 1, 2>(1); run<4, 1, 2>(1); run<8, 1, 2>(1); run<12, 1, 2>(1); run<16, 1, 2>(1);
    run<0, 2, 2>(2); run<4, 2, 2>(2); run<8, 2, 2>(2); run<12, 2, 2>(2);
    run<0, 1, 3>(1); run<4, 1, 3>(1); run<8, 1, 3>(1); run<0, 2, 3>(2);
    run<0, 1, 4>(1); run<4, 1, 4>(1); run<8, 1, 4>(1); run<0, 2, 4>(2);
    run<0, 1, 5>(1); run<4, 1, 5>(1); run<8, 1, 5>(1); run<0, 2, 5>(2);
    run<0, 1, 6>(1); run<4, 1, 6>(1); run<8, 1, 6>(1); run<0, 2, 6>(2);
    return 0;
}
