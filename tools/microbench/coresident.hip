// Do a wave's vector-ALU results depend on what ANOTHER wave on the same compute unit executes?
//
// A "victim" kernel runs a long, data-dependent chain of one vector instruction per lane and writes the final
// values; it is run alone (the reference) and then again while a "neighbour" kernel occupies the same compute
// units from a second stream.  The chains are deterministic, so the two outputs must be identical bit for bit.
// Victims:    0 v_pk_fma_f32   1 v_fma_f32   2 v_pk_mul_f32 + v_pk_add_f32   3 v_fma_f64   4 v_pk_fma_f32 with its
//             operands re-read from LDS every 64 steps   5 v_pk_fma_f32 op_sel_hi:[0,1,1] (first operand's low half
//             for both results)   6 v_pk_add_f32 with neg_lo / neg_hi + v_pk_mul_f32 with an inline constant
//             7 v_pk_fma_f32 whose operands are re-read from memory (16-byte loads) every 8 steps
// Neighbours: 0 v_mfma_f32_32x32x16_bf16 (accumulators in a-registers)   1 v_mfma_f32_32x32x2_f32
//             2 v_mfma_f32_32x32x16_bf16 (accumulators in v-registers)   3 v_fma_f32 only (no matrix instruction)
//             4 v_mfma_f32_16x16x32_bf16
// The neighbour allocates 220 registers per wave (156 + 64, the shape of this library's weight-gradient kernel), so
// two of its waves leave room for victim waves on every SIMD.
// Output: per (victim, neighbour) the number of victim lanes whose result differs, by 16-lane quarter of the wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 coresident.hip -o coresident && ./coresident [rounds]
// Two more modes pair one side with a real workload in ANOTHER process (tools/share_probe.sh; the flag files
// <dir>/ready and <dir>/done are the ones of tools/fwd_stale_probe.py):
//   ./coresident neighbour <C> <dir>     runs neighbour C again and again until <dir>/done exists
//   ./coresident victims <dir> <seconds> takes every victim's reference alone, creates <dir>/ready, then repeats the
//                                        victims for <seconds> and counts differing lanes; creates <dir>/done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <chrono>
#include <unistd.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define NV 8
template <int V>
__global__ void __launch_bounds__(256) victim(float* out, int iters, const floatx4* src) {
    __shared__ float lds[256 * 4];
    const int t = threadIdx.x, lane = t & 63;
    floatx2 acc = {1.0f + 1e-3f * lane, 2.0f - 1e-3f * lane};
    floatx2 a = {0.99990f + 1e-6f * lane, 1.00010f - 1e-6f * lane};
    floatx2 b = {1e-4f * (lane + 1), -1e-4f * (lane + 1)};
    double dacc = 1.0 + 1e-3 * lane, da = 0.9999 + 1e-6 * lane, db = 1e-4 * (lane + 1);
    lds[4 * t] = a.x; lds[4 * t + 1] = a.y; lds[4 * t + 2] = b.x; lds[4 * t + 3] = b.y;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (V == 4 && (it & 63) == 0) {
            const floatx4 v = *reinterpret_cast<const floatx4*>(lds + 4 * t);
            a = floatx2{v.x, v.y}; b = floatx2{v.z, v.w};
        }
        if (V == 7) {
            const floatx4 v = src[(size_t)((blockIdx.x * 256 + t) * 7 + it) & 0xffff];
            a = floatx2{v.x, v.y}; b = floatx2{v.z, v.w};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (V == 0 || V == 4 || V == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
            else if (V == 5) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
            else if (V == 6) {
                asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(b));
                asm volatile("v_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]" : "+v"(acc));
                asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(acc));
            }
            else if (V == 1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.x) : "v"(a.x), "v"(b.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.y) : "v"(a.y), "v"(b.y));
            } else if (V == 2) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "v"(a));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(b));
            } else if (V == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dacc) : "v"(da), "v"(db));
        }
    }
    const size_t i = (size_t)blockIdx.x * 256 + t;
    if (V == 3) { out[2 * i] = (float)dacc; out[2 * i + 1] = (float)(dacc * 1e3 - (double)(float)(dacc * 1e3)); }
    else { out[2 * i] = acc.x; out[2 * i + 1] = acc.y; }
}

template <int C>
__global__ void __launch_bounds__(256) neighbour(float* out, const u32x4* in, int iters) {
    floatx16 acc[4];
    floatx4 acc4[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    const u32x4 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i);
    const float y = 1.0f + 1e-7f * threadIdx.x, z = 1e-3f;
    const floatx2 a2 = {__uint_as_float(a.x), __uint_as_float(a.y)}, b2 = {__uint_as_float(b.x), __uint_as_float(b.y)};
    asm volatile("" ::: "v155");                 // 156 vector registers + the accumulators: 220 per wave with C = 0
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (C == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a), "v"(b));
            else if (C == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a2.x), "v"(b2.x));
            else if (C == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
            else if (C == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc4[u & 3]) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[u]) : "v"(y), "v"(z));
            if (C == 3) {
#pragma unroll
                for (int v = 0; v < 8; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v]) : "v"(y), "v"(z));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

static const floatx4* g_src;
template <int V> static void launch_victim(float* out, int blocks, int iters, hipStream_t s) {
    hipLaunchKernelGGL(victim<V>, dim3(blocks), dim3(256), 0, s, out, iters, g_src);
}
template <int C> static void launch_neighbour(float* out, const u32x4* in, int blocks, int iters, hipStream_t s) {
    hipLaunchKernelGGL(neighbour<C>, dim3(blocks), dim3(256), 0, s, out, in, iters);
}
typedef void (*victim_fn)(float*, int, int, hipStream_t);
typedef void (*neighbour_fn)(float*, const u32x4*, int, int, hipStream_t);

static bool exists(const std::string& p) { return access(p.c_str(), F_OK) == 0; }
static void touch(const std::string& p) { FILE* f = fopen(p.c_str(), "w"); if (f) fclose(f); }

int main(int argc, char** argv) {
    const bool mode_n = argc > 3 && !strcmp(argv[1], "neighbour"), mode_v = argc > 3 && !strcmp(argv[1], "victims");
    const int rounds = (argc > 1 && !mode_n && !mode_v) ? atoi(argv[1]) : 10;
    // [cus]: the number of compute units the process may use (ROC_GLOBAL_CU_MASK); the grids are sized to it
    const int cus = (argc > 2 && !mode_n && !mode_v) ? atoi(argv[2]) : (getenv("CORESIDENT_CUS") ? atoi(getenv("CORESIDENT_CUS")) : 256);
    const int vblocks = 8 * cus, viters = 4000, nblocks = 2 * cus, niters = 40000;
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const size_t vn = (size_t)vblocks * 256 * 2;
    float *ref, *got, *nout;
    u32x4* nin;
    CHECK(hipMalloc(&ref, vn * 4)); CHECK(hipMalloc(&got, vn * 4));
    CHECK(hipMalloc(&nout, (size_t)nblocks * 256 * 4)); CHECK(hipMalloc(&nin, 128 * 16));
    std::vector<unsigned> h(128 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f803f80u + (unsigned)(i * 0x00010001u);     // bf16 pairs near 1
    CHECK(hipMemcpy(nin, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    {
        std::vector<float> hs(65536 * 4);
        for (size_t i = 0; i < hs.size(); i += 4) {
            hs[i] = 0.9999f + 1e-9f * (float)(i % 1000); hs[i + 1] = 1.0001f - 1e-9f * (float)(i % 777);
            hs[i + 2] = 1e-4f * (float)(1 + i % 13); hs[i + 3] = -1e-4f * (float)(1 + i % 7);
        }
        floatx4* d;
        CHECK(hipMalloc(&d, hs.size() * 4));
        CHECK(hipMemcpy(d, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        g_src = d;
    }
    victim_fn victims[NV] = {launch_victim<0>, launch_victim<1>, launch_victim<2>, launch_victim<3>, launch_victim<4>,
                             launch_victim<5>, launch_victim<6>, launch_victim<7>};
    neighbour_fn neighbours[5] = {launch_neighbour<0>, launch_neighbour<1>, launch_neighbour<2>, launch_neighbour<3>,
                                  launch_neighbour<4>};
    const char* vname[NV] = {"v_pk_fma_f32", "v_fma_f32", "v_pk_mul+add_f32", "v_fma_f64", "v_pk_fma_f32+lds",
                             "v_pk_fma op_sel_hi", "v_pk_add neg + mul c", "v_pk_fma_f32+loads"};
    const char* nname[5] = {"mfma 32x32x16 bf16 (a)", "mfma 32x32x2 f32", "mfma 32x32x16 bf16 (v)", "v_fma_f32 only",
                            "mfma 16x16x32 bf16"};
    std::vector<float> hr(vn), hg(vn);
    if (mode_n) {
        const int c = atoi(argv[2]);
        const std::string dir = argv[3];
        while (!exists(dir + "/ready")) usleep(100000);
        long long launches = 0;
        while (!exists(dir + "/done")) {
            for (int k = 0; k < 4; ++k) neighbours[c](nout, nin, nblocks, niters / 8, s1);
            CHECK(hipStreamSynchronize(s1));
            launches += 4;
        }
        printf("load: neighbour %s launches %lld\n", nname[c], launches);
        return 0;
    }
    if (mode_v) {
        const std::string dir = argv[2];
        const double seconds = atof(argv[3]);
        std::vector<std::vector<float>> refs(NV, std::vector<float>(vn));
        for (int v = 0; v < NV; ++v) {
            victims[v](ref, vblocks, viters / 4, s2);
            CHECK(hipStreamSynchronize(s2));
            CHECK(hipMemcpy(refs[v].data(), ref, vn * 4, hipMemcpyDeviceToHost));
        }
        touch(dir + "/ready");
        long long q[NV][4] = {}, runs[NV] = {}, bad_runs[NV] = {};
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int v = 0; v < NV; ++v) {
                victims[v](got, vblocks, viters / 4, s2);
                CHECK(hipStreamSynchronize(s2));
                CHECK(hipMemcpy(hg.data(), got, vn * 4, hipMemcpyDeviceToHost));
                bool bad = false;
                for (size_t i = 0; i < vn / 2; ++i)
                    if (memcmp(&refs[v][2 * i], &hg[2 * i], 8) != 0) { q[v][(i & 63) >> 4]++; bad = true; }
                runs[v]++; bad_runs[v] += bad;
            }
        }
        touch(dir + "/done");
        for (int v = 0; v < NV; ++v)
            printf("victim %-20s deviating runs %lld/%lld   differing lanes by quarter %lld %lld %lld %lld\n", vname[v],
                   bad_runs[v], runs[v], q[v][0], q[v][1], q[v][2], q[v][3]);
        return 0;
    }
    printf("%-20s %-24s %10s   differing lanes by quarter (0-15, 16-31, 32-47, 48-63)\n", "victim", "neighbour", "runs");
    for (int v = 0; v < NV; ++v) {
        victims[v](ref, vblocks, viters, s2);
        CHECK(hipStreamSynchronize(s2));
        CHECK(hipMemcpy(hr.data(), ref, vn * 4, hipMemcpyDeviceToHost));
        {   // alone, again: the chain reproduces itself
            victims[v](got, vblocks, viters, s2);
            CHECK(hipStreamSynchronize(s2));
            CHECK(hipMemcpy(hg.data(), got, vn * 4, hipMemcpyDeviceToHost));
            if (memcmp(hr.data(), hg.data(), vn * 4) != 0) printf("%-20s differs from itself when run ALONE\n", vname[v]);
        }
        for (int c = 0; c < 5; ++c) {
            long long q[4] = {0, 0, 0, 0}, bad_runs = 0;
            for (int r = 0; r < rounds; ++r) {
                neighbours[c](nout, nin, nblocks, niters, s1);
                victims[v](got, vblocks, viters, s2);
                victims[v](got, vblocks, viters, s2);          // a second pass while the neighbour is surely running
                CHECK(hipStreamSynchronize(s2));
                CHECK(hipMemcpy(hg.data(), got, vn * 4, hipMemcpyDeviceToHost));
                CHECK(hipStreamSynchronize(s1));
                bool bad = false;
                for (size_t i = 0; i < vn / 2; ++i)
                    if (memcmp(&hr[2 * i], &hg[2 * i], 8) != 0) { q[(i & 63) >> 4]++; bad = true; }
                bad_runs += bad;
            }
            printf("%-20s %-24s %4lld/%-4d   %lld %lld %lld %lld\n", vname[v], nname[c], bad_runs, rounds, q[0], q[1], q[2], q[3]);
            fflush(stdout);
        }
    }
    return 0;
}
