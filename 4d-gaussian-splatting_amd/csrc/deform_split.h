// Split-operand matrix products for the deformation MLP (included by deform.hip).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the rate of v_mfma_f32_32x32x16_bf16.
// An fp32 value x is written EXACTLY as the sum of three bf16 numbers
//     x = h + m + l,  h = top 8 significant bits, m = next 8, l = last 8
// (truncation: h = bits & 0xffff0000, m likewise of the exact remainder x - h, ...).
// A product x*y is then the sum of nine exact bf16 x bf16 products; the six with
// weight >= 2^-16 (hh, hm, mh, hl, lh, mm) are issued as six bf16 MFMAs that
// accumulate in fp32, the three dropped ones (ml, lm, ll) are below 2^-23 |xy|,
// i.e. under the rounding error of a single fp32 multiply.  Six bf16 MFMAs of
// K = 16 cost 192 cycles per SIMD against 512 for the eight fp32 MFMAs they replace.
//
// Orientation is the one of the fp32 kernels: activations transposed in the
// accumulator layout (register r of a 32-row block holds feature row(r,h), lane
// c = point).  For the bf16 instruction lane (c, g = lane>>5) supplies, as B
// operand, eight K values: accumulator registers 8*jj .. 8*jj+7 of a block are
// exactly such a group, so K chunk j = 2*block + jj carries the features
//     kidx(j, g, e) = 16 j + 8 (e>>2) + 4 g + (e&3),   e = 0..7
// and the weights are packed to that order (three 1-KiB pieces per fragment).
#pragma once
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BMFMA(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

struct Split3 { u32x4 h, m, l; };        // eight fp32 values as three bf16x8 operands

// bf16 pair: low half = top 16 bits of lo, high half = top 16 bits of hi
__device__ __forceinline__ unsigned top16_pair(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float below_top16(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

template <bool RELU>
__device__ __forceinline__ Split3 split8(const float* v) {
    float x[8], r1[8], r2[8];
    Split3 s;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = RELU ? fmaxf(v[e], 0.0f) : v[e];
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

// eight consecutive registers of an accumulator block
template <bool RELU>
__device__ __forceinline__ Split3 split_acc(const floatx16& a, int jj) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a[8 * jj + e];
    return split8<RELU>(v);
}
// same, but the split cannot be scheduled before this point of the program: pure arithmetic is otherwise
// free to move across sched_barrier, and the compiler gathers all splits of a layer in front of its first MFMA
template <bool RELU>
__device__ __forceinline__ Split3 split_acc_here(const floatx16& a, int jj) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = a[8 * jj + e];
        asm volatile("" : "+v"(v[e]));
    }
    return split8<RELU>(v);
}

// c += a * b, smallest terms first
__device__ __forceinline__ floatx16 mma_split(const Split3& a, const Split3& b, floatx16 c) {
    c = BMFMA(a.m, b.m, c);
    c = BMFMA(a.l, b.h, c);
    c = BMFMA(a.h, b.l, c);
    c = BMFMA(a.m, b.h, c);
    c = BMFMA(a.h, b.m, c);
    c = BMFMA(a.h, b.h, c);
    return c;
}

// ---------------------------------------------------------------------------
// Weight fragments, split form: for row block ib and K chunk j three u32x4 per
// lane (pieces h, m, l), lane (c, g) holding element(32 ib + c, kidx(j, g, e)).
// ---------------------------------------------------------------------------
#define FRAG_BIAS 8          // not a matrix: a bias vector, copied and zero-padded
#define FRAG_PLANE 16        // not a matrix: a feature plane [16][I texels], rewritten channel-last [I][16]
#define FRAG_ZERO 32         // not a matrix: I 16-byte words at dst are cleared (src unused; dst 16-byte aligned)
struct SFragDesc { const float* src; u32x4* dst; int I; int K; int ld; int mode; };
struct SFragTable { SFragDesc d[FRAG_MAX]; int n; };

__host__ __device__ __forceinline__ int sfrag_vec4(int I, int K) { return ((I + 31) / 32) * ((K + 15) / 16) * 3 * 64; }

__global__ void __launch_bounds__(256)
pack_split_kernel(SFragTable t) {
    SFragDesc d = t.d[0];
#pragma unroll
    for (int i = 1; i < FRAG_MAX; ++i)
        if ((int)blockIdx.y == i) d = t.d[i];
    if ((int)blockIdx.y >= t.n) return;
    if (d.mode & FRAG_PLANE) {
        float* dst = reinterpret_cast<float*>(d.dst);
        for (int tx = blockIdx.x * blockDim.x + threadIdx.x; tx < d.I; tx += gridDim.x * blockDim.x) {
            float v[16];
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) v[ch] = d.src[(size_t)ch * d.I + tx];
            float4* q = reinterpret_cast<float4*>(dst + (size_t)tx * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        }
        return;
    }
    if (d.mode & FRAG_ZERO) {            // a clear that rides in this launch instead of taking one of its own
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < d.I; idx += gridDim.x * blockDim.x) d.dst[idx] = z;
        return;
    }
    if (d.mode & FRAG_BIAS) {            // plain copy of I floats, zero-padded to K
        float* dst = reinterpret_cast<float*>(d.dst);
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < d.K; idx += gridDim.x * blockDim.x)
            dst[idx] = idx < d.I ? d.src[idx] : 0.0f;
        return;
    }
    const int J = (d.K + 15) / 16, IB = (d.I + 31) / 32;
    const int total = IB * J * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int l = idx & 63, j = (idx >> 6) % J, ib = (idx >> 6) / J;
        const int c = l & 31, g = l >> 5;
        const int i = 32 * ib + c;
        const int col = (d.mode & FRAG_PERMROWS) ? (16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)) : i;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = (d.mode & FRAG_FEATMAP) ? (16 * g + 8 * j + e) : (16 * j + 8 * (e >> 2) + 4 * g + (e & 3));
            float x = 0.0f;
            if (i < d.I && kk < d.K)
                x = (d.mode & FRAG_TRANSPOSE) ? d.src[(size_t)kk * d.ld + col] : d.src[(size_t)col * d.ld + kk];
            v[e] = x;
        }
        const Split3 s = split8<false>(v);
        u32x4* q = d.dst + (size_t)((ib * J + j) * 3) * 64 + l;
        q[0] = s.h; q[64] = s.m; q[128] = s.l;
    }
}

__device__ __forceinline__ Split3 sfrag_load(const u32x4* packed, int ib, int J, int j, int lane) {
    const u32x4* q = packed + (size_t)((ib * J + j) * 3) * 64 + lane;
    Split3 s;
    s.h = q[0]; s.m = q[64]; s.l = q[128];
    return s;
}

// Same with the fragment address formed as (wave-uniform base in scalar registers) + (lane * 16): no 64-bit
// per-lane address is ever materialised.  Inside a loop over point tiles the compiler otherwise hoists one
// address pair per fragment out of the loop (100+ pairs), spills them, and every reload waits for all loads.
typedef __attribute__((address_space(1))) const char* gptr_t;    // global address space: global_load, not flat_load
__device__ __forceinline__ gptr_t uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ Split3 sfrag_load_u(const u32x4* packed, int frag, int lane) {
    typedef __attribute__((address_space(1))) const u32x4* gvec_t;
    gptr_t b = uniform_ptr(packed) + (size_t)frag * 3072;
    const unsigned off = (unsigned)lane * 16u;
    Split3 s;
    s.h = *(gvec_t)(b + off);
    s.m = *(gvec_t)(b + off + 1024);
    s.l = *(gvec_t)(b + off + 2048);
    return s;
}

#ifndef SCB
#define SCB 2                            // column blocks (32 points each) per wave
#endif

#define SGB_MFMA 0x008
#define SGB_VALU 0x002
// scheduling hint for the block just written: n times (one MFMA, then v vector instructions)
#define INTERLEAVE(n, v)                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                      \
        __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(SGB_VALU, (v), 0);               \
    }

// acc[cb][nb] += W[32 nb ..][:] * relu?(X[cb]) for all four row blocks of a 128x128 layer.  X is in accumulator
// form; K chunk j+1 is split (pure VALU work) while the 48 MFMAs of chunk j run.
template <bool RELU>
__device__ __forceinline__ void split_layer128x4(const u32x4* packed, int lane, const floatx16 (*x)[4],
                                                 floatx16 (*acc)[4]) {
    // weight fragments in the order t = 4 j + nb, requested two fragments (24 MFMAs) ahead of their use
    Split3 f[3], xs[2][SCB];
    f[0] = sfrag_load(packed, 0, 8, 0, lane);
    f[1] = sfrag_load(packed, 1, 8, 0, lane);
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb) xs[0][cb] = split_acc<RELU>(x[cb][0], 0);
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const int j = t >> 2, nb = t & 3;
        if (t + 2 < 32) f[(t + 2) % 3] = sfrag_load(packed, (t + 2) & 3, 8, (t + 2) >> 2, lane);
        if (j + 1 < 8 && nb < SCB) xs[(j + 1) & 1][nb] = split_acc_here<RELU>(x[nb][(j + 1) >> 1], (j + 1) & 1);
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) acc[cb][nb] = mma_split(f[t % 3], xs[j & 1][cb], acc[cb][nb]);
        INTERLEAVE(6 * SCB, 5)
        __builtin_amdgcn_sched_barrier(0);   // keeps the prefetch distance (bounds the live registers)
    }
}

// same with X already split (K chunks xs[cb][j]): no vector work at all besides the fragment loads
__device__ __forceinline__ void split_layer128x4_pre(const u32x4* packed, int lane, const Split3 (*xs)[8],
                                                     floatx16 (*acc)[4]) {
    Split3 f[3];
    f[0] = sfrag_load(packed, 0, 8, 0, lane);
    f[1] = sfrag_load(packed, 1, 8, 0, lane);
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const int j = t >> 2, nb = t & 3;
        if (t + 2 < 32) f[(t + 2) % 3] = sfrag_load(packed, (t + 2) & 3, 8, (t + 2) >> 2, lane);
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) acc[cb][nb] = mma_split(f[t % 3], xs[cb][j], acc[cb][nb]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc[cb][ob] += W[32 ob ..][:] * relu?(X[cb]) for the first NOB row blocks, same pipelining
template <bool RELU, int NOB>
__device__ __forceinline__ void split_layer128_acc(const u32x4* packed, int lane, const floatx16 (*x)[4],
                                                   floatx16 (*acc)[NOB]) {
    // fragments in the order t = NOB j + ob
    Split3 f[3], xs[2][SCB];
    f[0] = sfrag_load(packed, 0, 8, 0, lane);
    f[1] = sfrag_load(packed, 1 % NOB, 8, 1 / NOB, lane);
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb) xs[0][cb] = split_acc<RELU>(x[cb][0], 0);
#pragma unroll
    for (int t = 0; t < 8 * NOB; ++t) {
        const int j = t / NOB, ob = t % NOB;
        if (t + 2 < 8 * NOB) f[(t + 2) % 3] = sfrag_load(packed, (t + 2) % NOB, 8, (t + 2) / NOB, lane);
        if (j + 1 < 8) {
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb)
                if ((cb * NOB) / SCB == ob)      // spread the splits of chunk j+1 over the NOB steps of chunk j
                    xs[(j + 1) & 1][cb] = split_acc_here<RELU>(x[cb][(j + 1) >> 1], (j + 1) & 1);
        }
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) acc[cb][ob] = mma_split(f[t % 3], xs[j & 1][cb], acc[cb][ob]);
        INTERLEAVE(6 * SCB, NOB == 1 ? 10 : 5)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Output layer of a head with k <= 4 rows as plain fp32 dot products on the vector ALU: lane (c, h) holds 64 of
// the 128 features of its point, the two lane halves are summed with one cross-lane add per output.  Cheaper
// than a 32-row matrix block of which at most four rows are used.  w2 is the [k,128] weight in LDS (the
// rows are re-read by every point tile; from LDS their latency is short enough not to need a prefetch).
__device__ __forceinline__ void small_head(const float* w2, int k, int h, floatx16 (*x)[4], float (*res)[4]) {
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[cb][nb][r] = fmaxf(x[cb][nb][r], 0.0f);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float s[SCB];
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) s[cb] = 0.0f;
        if (o < k) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *reinterpret_cast<const float4*>(w2 + o * DW + 32 * nb + 8 * q + 4 * h);
#pragma unroll
                    for (int cb = 0; cb < SCB; ++cb) {
                        s[cb] = fmaf(x[cb][nb][4 * q], w.x, s[cb]);
                        s[cb] = fmaf(x[cb][nb][4 * q + 1], w.y, s[cb]);
                        s[cb] = fmaf(x[cb][nb][4 * q + 2], w.z, s[cb]);
                        s[cb] = fmaf(x[cb][nb][4 * q + 3], w.w, s[cb]);
                    }
                }
        }
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) res[cb][o] = s[cb] + __shfl_xor(s[cb], 32);
    }
}

// padded bias vector -> accumulator block nb (register 4q+i of lane half h is row 32 nb + 8 q + 4 h + i)
__device__ __forceinline__ floatx16 bias_block(const float* __restrict__ padded, int nb, int h) {
    floatx16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(padded + 32 * nb + 8 * q + 4 * h);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    return v;
}

// ---------------------------------------------------------------------------
// Forward, kernel 1: HexPlane features at full occupancy.  Gather phase: 16 lanes
// per point (lane = channel), the two levels in turn, so that every texel access
// of a point is one contiguous 64-byte segment (four segments per wave
// instruction; with one lane per point an instruction touches 64 of them and the
// kernel is bound by the cache-line rate of the vector memory path).  The
// features go through an LDS tile; in the second phase lane (c, h) of a wave
// reads the 16 features of level h of point c of a 32-point group and writes
// them already split, in the B-operand order of the first Linear (group G,
// chunk j, piece: 64 lanes x 16 bytes contiguous).
// ---------------------------------------------------------------------------
struct HexFeatSK {
    int n;
    PlaneSet ps;
    const float* aabb;
    const float* xyz; const float* time; float time_scalar;
    u32x4* feat_split;                   // [ceil(n/32)][2][3][64]
    float* feat_f32;                     // optional [n,32]
    u32x4* feat_y;                       // optional: the features split, points-as-K order (B operand of the dW0
                                         // product of the backward): [ceil(n/16)][3][64]
};

#define HF_PTS 128                       // points per workgroup (4 waves x 32)
#define HF_STRIDE 36                     // floats per point row of the tile: 32 + 4, rows of 16 lanes hit 64 distinct banks

__global__ void __launch_bounds__(256)
hexplane_features_split_kernel(HexFeatSK a) {
    __shared__ __attribute__((aligned(16))) float tile[HF_PTS * HF_STRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // Four lanes per point, each four of the sixteen channels: a texel is one 64-byte segment read by four 16-byte
        // loads.  (With sixteen lanes per point, one channel each, every lane repeated the point's coordinate,
        // index and weight arithmetic: 70 % of the kernel's vector instructions, and halving its texel loads did
        // not change its time.)  One level at a time: 24 loads of 16 bytes in flight per lane.
        const int c4 = lane & 3, sub = lane >> 2;
        const int p0 = blockIdx.x * HF_PTS;
#pragma unroll 1
        for (int ib = wave * 16; ib < HF_PTS; ib += 64) {
            const int pl = ib + sub;
            const int p = min(p0 + pl, a.n - 1);
            float co[4], sc3[3];
            point_coords(a.xyz, a.aabb, a.time, a.time_scalar, p, co, sc3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                AxisLookup axl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) axl[q] = make_axis(co[q], h ? a.ps.res[1][q] : a.ps.res[0][q]);
                float4 tex[6][4];
#pragma unroll
                for (int ci = 0; ci < 6; ++ci) {
                    const int ax = plane_a(ci), bx = plane_b(ci);
                    const int Wd = h ? a.ps.res[1][ax] : a.ps.res[0][ax];
                    // wave-uniform base + 32-bit lane offset (in 16-byte units: texel * 4 + channel quad; a plane is
                    // far below 4 GB): one address instruction per load instead of a 64-bit multiply-add
                    const float4* __restrict__ pl_ = reinterpret_cast<const float4*>(h ? a.ps.pl[1][ci] : a.ps.pl[0][ci]);
                    const unsigned o00 = (unsigned)((axl[bx].i0 * Wd + axl[ax].i0) * (DF / 4) + c4);
                    const unsigned ox = (unsigned)(axl[ax].d * (DF / 4)), oy = (unsigned)(axl[bx].d * Wd * (DF / 4));
                    tex[ci][0] = pl_[o00];
                    tex[ci][1] = pl_[o00 + ox];
                    tex[ci][2] = pl_[o00 + oy];
                    tex[ci][3] = pl_[o00 + oy + ox];
                }
                float4 feat = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
#pragma unroll
                for (int ci = 0; ci < 6; ++ci) {
                    const float fx = axl[plane_a(ci)].f, fy = axl[plane_b(ci)].f;
                    const float w0 = (1.0f - fx) * (1.0f - fy), w1 = fx * (1.0f - fy), w2 = (1.0f - fx) * fy, w3 = fx * fy;
                    feat.x *= tex[ci][0].x * w0 + tex[ci][1].x * w1 + tex[ci][2].x * w2 + tex[ci][3].x * w3;
                    feat.y *= tex[ci][0].y * w0 + tex[ci][1].y * w1 + tex[ci][2].y * w2 + tex[ci][3].y * w3;
                    feat.z *= tex[ci][0].z * w0 + tex[ci][1].z * w1 + tex[ci][2].z * w2 + tex[ci][3].z * w3;
                    feat.w *= tex[ci][0].w * w0 + tex[ci][1].w * w1 + tex[ci][2].w * w2 + tex[ci][3].w * w3;
                }
                *reinterpret_cast<float4*>(tile + pl * HF_STRIDE + 16 * h + 4 * c4) = feat;
            }
        }
    }
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    if (a.feat_y) {
        // the same tile read the other way: lane (feature c, half h) takes points 8 h .. 8 h + 7 of a 16-point
        // chunk; two chunks per wave.  Points beyond n contribute zeros.
        const int total_chunks = (a.n + 15) / 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int lc = wave * 2 + q;                     // chunk within the workgroup's 128 points
            const int chunk = blockIdx.x * (HF_PTS / 16) + lc;
            if (chunk < total_chunks) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int pl = 16 * lc + 8 * h + e;
                    v[e] = (blockIdx.x * HF_PTS + pl < a.n) ? tile[pl * HF_STRIDE + c] : 0.0f;
                }
                const Split3 s3 = split8<false>(v);
                u32x4* dst = a.feat_y + (size_t)(chunk * 3) * 64 + lane;
                dst[0] = s3.h; dst[64] = s3.m; dst[128] = s3.l;
            }
        }
    }
    const int G = blockIdx.x * 4 + wave;
    if (G * 32 >= a.n) return;
    const int p = G * 32 + c;
    float feat[DF];
    {
        const float4* row = reinterpret_cast<const float4*>(tile + (wave * 32 + c) * HF_STRIDE + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = row[q];
            feat[4 * q] = t.x; feat[4 * q + 1] = t.y; feat[4 * q + 2] = t.z; feat[4 * q + 3] = t.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const Split3 s = split8<false>(feat + 8 * j);
        u32x4* q = a.feat_split + (size_t)((G * 2 + j) * 3) * 64 + lane;
        q[0] = s.h; q[64] = s.m; q[128] = s.l;
    }
    if (a.feat_f32 && p < a.n) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(a.feat_f32 + (size_t)p * 32 + 16 * h + 4 * q) =
                make_float4(feat[4 * q], feat[4 * q + 1], feat[4 * q + 2], feat[4 * q + 3]);
    }
}

// ReLU masks.  Lane (c, h) owns 64 of a point's 128 features (accumulator registers r of blocks nb);
// bit 31 - (16 (nb & 1) + r) of word nb >> 1 is SET when the value is NOT positive.  A positive float has
// an integer image n in [1, 0x7f800000]; for every other float (zeros of both signs, negatives) n <= 0 as a
// signed integer, and bit 31 of n | (n - 1) is exactly "n <= 0".  v_alignbit shifts that bit in: three vector
// instructions per value to build, two (bfe + and) to apply.
__device__ __forceinline__ uint2 relu_mask_build(const floatx16* x) {
    unsigned w[2] = {0u, 0u};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const unsigned n = __float_as_uint(x[nb][r]);
            w[nb >> 1] = __builtin_amdgcn_alignbit(w[nb >> 1], n | (n - 1u), 31);
        }
    return make_uint2(w[0], w[1]);
}
__device__ __forceinline__ void relu_mask_apply(uint2 m, floatx16* x) {
    const unsigned w[2] = {m.x, m.y};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int neg = __builtin_amdgcn_sbfe((int)w[nb >> 1], 31 - (16 * (nb & 1) + r), 1);   // -1 if masked
            x[nb][r] = __uint_as_float(__float_as_uint(x[nb][r]) & ~(unsigned)neg);
        }
}

// ---------------------------------------------------------------------------
// Forward, kernel 2: the MLP.  One wave = 64 points (two column blocks), whole
// register file, no LDS.
// ---------------------------------------------------------------------------
// The last-layer rows of the four heads with few outputs (3 + 3 + 4 + 1 rows of 128 floats) into LDS: all eight
// 16-byte loads of a lane are requested before the first is stored (a copy loop goes load, wait, store, and a wave
// that has its SIMD to itself pays one memory latency per trip).
__device__ __forceinline__ void stage_small_w2(const float* const* w2, float (*w2s)[4 * DW], int lane) {
    float4 v[4][2];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        const int vecs = head_dim(hd) * (DW / 4);
        const float4* __restrict__ src = reinterpret_cast<const float4*>(sel5(w2, hd));
#pragma unroll
        for (int it = 0; it < 2; ++it) v[hd][it] = src[min(lane + 64 * it, vecs - 1)];
    }
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        const int vecs = head_dim(hd) * (DW / 4);
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (lane + 64 * it < vecs) reinterpret_cast<float4*>(w2s[hd])[lane + 64 * it] = v[hd][it];
    }
}

struct DeformFwdSK {
    int n;
    const u32x4* feat_split;
    const u32x4* pw0; const float* pb0;              // packed first Linear, padded bias
    const u32x4* pw1[5]; const float* pb1[5]; const u32x4* pw2[5]; const float* pb2[5];
    const float* w2[5]; const float* b2[5];          // unpacked output layers (used for the heads with k <= 4)
    const float* in[5];                              // residual inputs: xyz, scales, rotations, opacity, shs
    const float* in_rest;                            // NULL, or in[4] is f_dc [n,3] and this is f_rest [n,45]
    float* out[5];
    float* save_hidden;                              // optional [n,128]: hidden, pre-activation
    u32x4* save_act;                                 // relu(hidden) as split operand pieces: [ceil(n/32)][8][3][64]
    uint2* save_mask;                                // optional [6][n][2]: ReLU masks of the heads' h1 and of hidden
    u32x4* save_yhid;                                // optional: relu(hidden) split, points-as-K order (the B operand of
                                                     // the dW1 products): [ceil(n/16)][4][3][64]
};

// hidden^T = W0 f^T + b0 for the SCB point groups of this wave
__device__ __forceinline__ void split_first_linear(const u32x4* feat_split, const u32x4* pw0, const float* pb0,
                                                   int group0, int ngroups, int lane, int h, floatx16 (*hid)[4]) {
    Split3 fs[SCB][2];
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb) {
        const int G = min(group0 + cb, ngroups - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) fs[cb][j] = sfrag_load(feat_split, G, 2, j, lane);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const floatx16 b = bias_block(pb0, nb, h);
        const Split3 w_0 = sfrag_load(pw0, nb, 2, 0, lane), w_1 = sfrag_load(pw0, nb, 2, 1, lane);
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) {
            hid[cb][nb] = mma_split(w_0, fs[cb][0], b);
            hid[cb][nb] = mma_split(w_1, fs[cb][1], hid[cb][nb]);
        }
    }
}

// SAVE: training variant, stores hidden and relu(h1) for the backward
template <bool SAVE>
__global__ void __launch_bounds__(64, SCB == 1 ? 2 : 1)
deform_fwd_split_kernel(DeformFwdSK a) {
    const int lane = threadIdx.x;
    const int c = lane & 31, h = lane >> 5;
    const int base = blockIdx.x * (32 * SCB);

    // output layers of the four small heads (3+3+4+1 rows of 128) staged in LDS, one row group per head
    __shared__ float w2s[4][4 * DW];
    stage_small_w2(a.w2, w2s, lane);
    __syncthreads();
    floatx16 hid[SCB][4];
    split_first_linear(a.feat_split, a.pw0, a.pb0, blockIdx.x * SCB, (a.n + 31) / 32, lane, h, hid);
    if (SAVE) {
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) {
            const int p = base + 32 * cb + c;
            if (p < a.n) {
                a.save_mask[((size_t)5 * a.n + p) * 2 + h] = relu_mask_build(hid[cb]);
                if (a.save_hidden) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(a.save_hidden + (size_t)p * DW + 32 * nb + 8 * q + 4 * h) =
                                make_float4(hid[cb][nb][4 * q], hid[cb][nb][4 * q + 1], hid[cb][nb][4 * q + 2],
                                            hid[cb][nb][4 * q + 3]);
                }
            }
        }
        if (a.save_yhid) {
            // relu(hidden) once more, transposed: the dW1 products of the backward contract over the POINTS, so
            // their operand needs eight consecutive points of one feature per lane.  32 points x 128 features go
            // through an LDS tile (row = feature, 36 floats: 16-byte aligned, conflict-free for both accesses);
            // lane (n, kg) then reads points 8 kg .. 8 kg + 7 of feature 32 nb + n for each 16-point chunk.
            __shared__ __attribute__((aligned(16))) float ytile[DW * 36];
            const int total_chunks = (a.n + 15) / 16;
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb) {
                const bool ok = base + 32 * cb + c < a.n;
                __syncthreads();
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ytile[(32 * nb + drow(r, h)) * 36 + c] = ok ? hid[cb][nb][r] : 0.0f;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int chunk = (base + 32 * cb) / 16 + q;
                    if (chunk < total_chunks) {
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) {
                            const float4* src = reinterpret_cast<const float4*>(ytile + (32 * nb + c) * 36 + 16 * q + 8 * h);
                            const float4 v0 = src[0], v1 = src[1];
                            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                            const Split3 s3 = split8<true>(v);
                            u32x4* dst = a.save_yhid + (size_t)((chunk * 4 + nb) * 3) * 64 + lane;
                            dst[0] = s3.h; dst[64] = s3.m; dst[128] = s3.l;
                        }
                    }
                }
            }
        }
    }
    Split3 hs[SCB][8];                   // relu(hidden) split once, shared by the five heads
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb)
#pragma unroll
        for (int j = 0; j < 8; ++j) hs[cb][j] = split_acc<true>(hid[cb][j >> 1], j & 1);
    if (SAVE) {                          // the same pieces are the A operand of the dW2 recompute in the backward
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) {
            const int G = blockIdx.x * SCB + cb;
            if (G * 32 < a.n) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u32x4* q = a.save_act + (size_t)((G * 8 + j) * 3) * 64 + lane;
                    q[0] = hs[cb][j].h; q[64] = hs[cb][j].m; q[128] = hs[cb][j].l;
                }
            }
        }
    }
    for (int head = 0; head < 5; ++head) {
        const int k = head_dim(head);
        const u32x4* pw1 = sel5(a.pw1, head); const float* pb1 = sel5(a.pb1, head);
        const u32x4* pw2 = sel5(a.pw2, head); const float* pb2 = sel5(a.pb2, head);
        const float* __restrict__ b2 = sel5(a.b2, head);
        const float* __restrict__ in = sel5(a.in, head); float* __restrict__ out = sel5(a.out, head);
        floatx16 h1[SCB][4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const floatx16 b = bias_block(pb1, nb, h);
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb) h1[cb][nb] = b;
        }
        split_layer128x4_pre(pw1, lane, hs, h1);
        auto save_act = [&]() {
            if (SAVE) {
#pragma unroll
                for (int cb = 0; cb < SCB; ++cb) {
                    const int p = base + 32 * cb + c;
                    if (p < a.n) {
                        a.save_mask[((size_t)head * a.n + p) * 2 + h] = relu_mask_build(h1[cb]);
                    }
                }
            }
        };
        // issued below, after this head's residual-input loads: a wait on those loads then does not also wait
        // for the 32 stores (loads and stores share one in-order counter)
        if (k <= 4) {
            float iv[SCB][4], res[SCB][4];
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb) {
                const int p = min(base + 32 * cb + c, a.n - 1);
#pragma unroll
                for (int o = 0; o < 4; ++o) {    // unconditional loads (clamped index), then the select: see below
                    const int oc = min(o, k - 1);
                    const float v = in[(size_t)p * k + oc] + b2[oc];
                    iv[cb][o] = o < k ? v : 0.0f;
                }
            }
            save_act();
            small_head(w2s[head], k, h, h1, res);
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb) {
                const int p = base + 32 * cb + c;
                if (p < a.n && h == 0) {
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < k) out[(size_t)p * k + o] = iv[cb][o] + res[cb][o];
                }
            }
        } else {                          // shs: 48 rows = two row blocks sharing the splits of h1
            float4 iv[SCB][2][4];
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb) {
                const int p = min(base + 32 * cb + c, a.n - 1);
                const float* __restrict__ row0 = a.in_rest ? a.in_rest + (size_t)p * 45 - 3 : in + (size_t)p * k;
                const float* __restrict__ dc = a.in_rest ? in + (size_t)p * 3 : nullptr;
                // The residual inputs are requested here and used after the second layer.  Every load is
                // unconditional: a load inside a branch (a lane-dependent one included) makes the compiler wait for
                // ALL outstanding loads at the join, and the twelve requests of a point then go out one memory
                // latency apart on a wave that has its SIMD to itself.  Which of the 8-row groups exist (48 rows) is
                // known at compile time; the first group of the two-tensor layout takes its first three floats from
                // the dc tensor by a select on the lane half.
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (32 * ob + 8 * q >= 48) {
                            iv[cb][ob][q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            continue;
                        }
                        const int row = 32 * ob + 8 * q + 4 * h;
                        if (ob == 0 && q == 0) {
                            const fdgs_f4u hi = *reinterpret_cast<const fdgs_f4u*>(row0 + 4);
                            const float* __restrict__ first = dc ? dc : row0;
                            const float f0 = first[0], f1 = first[1], f2 = first[2], f3 = row0[3];
                            iv[cb][ob][q] = h ? make_float4(hi.x, hi.y, hi.z, hi.w) : make_float4(f0, f1, f2, f3);
                        } else {
                            const fdgs_f4u v = *reinterpret_cast<const fdgs_f4u*>(row0 + row);
                            iv[cb][ob][q] = make_float4(v.x, v.y, v.z, v.w);
                        }
                    }
            }
            floatx16 o[SCB][2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const floatx16 b = bias_block(pb2, ob, h);
#pragma unroll
                for (int cb = 0; cb < SCB; ++cb) o[cb][ob] = b;
            }
            save_act();
            split_layer128_acc<true, 2>(pw2, lane, h1, o);
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb) {
                const int p = base + 32 * cb + c;
                if (p >= a.n) continue;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 32 * ob + 8 * q + 4 * h;
                        if (row < k)
                            *reinterpret_cast<float4*>(out + (size_t)p * k + row) =
                                make_float4(iv[cb][ob][q].x + o[cb][ob][4 * q], iv[cb][ob][q].y + o[cb][ob][4 * q + 1],
                                            iv[cb][ob][q].z + o[cb][ob][4 * q + 2], iv[cb][ob][q].w + o[cb][ob][4 * q + 3]);
                    }
            }
        }
    }
}

// ===========================================================================
// Backward, pass 1 (split form).  One wave = 64 points.  Nothing is recomputed:
// the forward saved hidden and relu(h1) of every head.  Per head
//     dh1 = (W2^T g) * [a1 > 0]        (vector ALU for k <= 4, MFMA for shs)
//     dacc += W1^T dh1
// then dhid = dacc * [hidden > 0] and dfeat = W0^T dhid.  dh1, dhid, dfeat go to
// the workspace (point-major) for the weight-gradient products and the
// HexPlane backward.
// ===========================================================================
struct DeformBwdSK {
    int n;
    const u32x4* pw1t[5];                        // split W1^T
    const u32x4* pw2t_shs;                       // split W2^T of the shs head: [128 x 48]
    const u32x4* pw0t;                           // split W0^T, rows permuted so that lane half h receives level h
    const float* w2[5];                          // unpacked W2 (small heads)
    const uint2* mask;                           // [6][n_total][2] ReLU masks saved by the forward
    int n_total; int p_first;                    // the call's point count and this launch's first point
    const float* g[5];                           // output cotangents
    float* ws_dh1; float* ws_dhid; float* ws_df;  // dh1: [5][dh1_rows][128], one contiguous stream per head
    size_t dh1_rows;
};

__global__ void __launch_bounds__(64, 1)
deform_bwd_points_split_kernel(DeformBwdSK a) {
    __shared__ float w2s[4][4 * DW];
    const int lane = threadIdx.x;
    const int c = lane & 31, h = lane >> 5;
    stage_small_w2(a.w2, w2s, lane);
    __syncthreads();

    int p[SCB], pc[SCB]; bool pvalid[SCB];
    floatx16 dacc[SCB][4];
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb) {
        p[cb] = blockIdx.x * (32 * SCB) + 32 * cb + c;
        pvalid[cb] = p[cb] < a.n;
        pc[cb] = pvalid[cb] ? p[cb] : a.n - 1;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[cb][nb][r] = 0.0f;
    }
    // ReLU masks of the five heads and of hidden: 8 bytes per lane each, coalesced
    uint2 msk[SCB][5], hmsk[SCB];
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb) {
#pragma unroll
        for (int hd = 0; hd < 5; ++hd) msk[cb][hd] = a.mask[((size_t)hd * a.n_total + a.p_first + pc[cb]) * 2 + h];
        hmsk[cb] = a.mask[((size_t)5 * a.n_total + a.p_first + pc[cb]) * 2 + h];
    }
    for (int head = 0; head < 5; ++head) {
        const int k = head_dim(head);
        const u32x4* pw1t = sel5(a.pw1t, head);
        const float* __restrict__ g = sel5(a.g, head);
        floatx16 t[SCB][4];      // dh1
        // dh1 = W2^T g
        if (k <= 4) {
            float gv[SCB][4];
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb)
#pragma unroll
                for (int o = 0; o < 4; ++o) gv[cb][o] = (o < k && pvalid[cb]) ? g[(size_t)p[cb] * k + o] : 0.0f;
            const float* w2 = w2s[head];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 acc4[SCB];
#pragma unroll
                    for (int cb = 0; cb < SCB; ++cb) acc4[cb] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (o < k) {
                            const float4 w = *reinterpret_cast<const float4*>(w2 + o * DW + 32 * nb + 8 * q + 4 * h);
#pragma unroll
                            for (int cb = 0; cb < SCB; ++cb) {
                                acc4[cb].x = fmaf(w.x, gv[cb][o], acc4[cb].x); acc4[cb].y = fmaf(w.y, gv[cb][o], acc4[cb].y);
                                acc4[cb].z = fmaf(w.z, gv[cb][o], acc4[cb].z); acc4[cb].w = fmaf(w.w, gv[cb][o], acc4[cb].w);
                            }
                        }
                    }
#pragma unroll
                    for (int cb = 0; cb < SCB; ++cb) {
                        t[cb][nb][4 * q] = acc4[cb].x; t[cb][nb][4 * q + 1] = acc4[cb].y;
                        t[cb][nb][4 * q + 2] = acc4[cb].z; t[cb][nb][4 * q + 3] = acc4[cb].w;
                    }
                }
        } else {
            Split3 gsp[SCB][3];
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float4 g0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g1 = g0;
                    if (pvalid[cb]) {
                        g0 = *reinterpret_cast<const float4*>(g + (size_t)p[cb] * 48 + 16 * j + 4 * h);
                        g1 = *reinterpret_cast<const float4*>(g + (size_t)p[cb] * 48 + 16 * j + 8 + 4 * h);
                    }
                    const float v[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                    gsp[cb][j] = split8<false>(v);
                }
#pragma unroll
            for (int cb = 0; cb < SCB; ++cb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[cb][nb][r] = 0.0f;
            // the twelve W2^T fragments, requested three steps ahead (a fragment consumed right behind its request
            // costs this wave - alone on its SIMD - one L2 latency per step)
            Split3 wq[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) wq[i] = sfrag_load(a.pw2t_shs, i & 3, 3, i >> 2, lane);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int j = i >> 2, nb = i & 3;
                if (i + 3 < 12) wq[(i + 3) & 3] = sfrag_load(a.pw2t_shs, (i + 3) & 3, 3, (i + 3) >> 2, lane);
#pragma unroll
                for (int cb = 0; cb < SCB; ++cb) t[cb][nb] = mma_split(wq[i & 3], gsp[cb][j], t[cb][nb]);
            }
        }
        // ReLU mask, and dh1 out
#pragma unroll
        for (int cb = 0; cb < SCB; ++cb) {
            relu_mask_apply(sel5(msk[cb], head), t[cb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (pvalid[cb] && head == 4)     // the small heads' dh1 is re-formed inside the dW1 product
                        *reinterpret_cast<float4*>(a.ws_dh1 + ((size_t)head * a.dh1_rows + p[cb]) * DW + 32 * nb + 8 * q + 4 * h) =
                            make_float4(t[cb][nb][4 * q], t[cb][nb][4 * q + 1], t[cb][nb][4 * q + 2], t[cb][nb][4 * q + 3]);
        }
        split_layer128x4<false>(pw1t, lane, t, dacc);
    }
    floatx16 df[SCB][1];
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb) {
        relu_mask_apply(hmsk[cb], dacc[cb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (pvalid[cb])
                    *reinterpret_cast<float4*>(a.ws_dhid + (size_t)p[cb] * DW + 32 * nb + 8 * q + 4 * h) =
                        make_float4(dacc[cb][nb][4 * q], dacc[cb][nb][4 * q + 1], dacc[cb][nb][4 * q + 2],
                                    dacc[cb][nb][4 * q + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) df[cb][0][r] = 0.0f;
    }
    split_layer128_acc<false, 1>(a.pw0t, lane, dacc, df);
#pragma unroll
    for (int cb = 0; cb < SCB; ++cb)
        if (pvalid[cb]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(a.ws_df + (size_t)p[cb] * 32 + 16 * h + 4 * q) =
                    make_float4(df[cb][0][4 * q], df[cb][0][4 * q + 1], df[cb][0][4 * q + 2], df[cb][0][4 * q + 3]);
        }
}

// ===========================================================================
// Backward, pass 2 (split form): C[m][n] += sum_p X[p][m] * relu(Y[p][n]) for the
// 128 x 128 products (dW1 of the five heads; Y = hidden is the same for all of
// them), db[m] += sum_p X[p][m].  The points are the MFMA K dimension: for a K
// chunk of 16 points lane (c, g) holds the eight points pb + 8 g + e of column c.
//   prep   relu(Y) is split once into B-operand order (chunk, column block, piece:
//          64 lanes x 16 bytes) - every head and row block re-reads it.
//   main   workgroup = (point range, head), wave = row block of 32: per chunk
//          eight coalesced dword loads of X (one 128-byte row segment per lane
//          half), one split, twelve 1-KiB piece loads, 24 MFMAs.
// ===========================================================================
// prep: Y[p][0..ncols) (ncols <= 32 nb) -> split pieces, order (chunk, column block, piece, lane); optionally
// relu first; optionally the column sums of Y accumulated into colsum[0..ncols) (bias gradients)
struct WGradPrepDesc { const float* Y; int ldy; int ncols; int nb; int relu; u32x4* out; float* colsum; };
#define WPREP_MAX 8
struct WGradPrepK { int n; WGradPrepDesc d[WPREP_MAX]; };

__global__ void __launch_bounds__(256)
wgrad_prep_split_kernel(WGradPrepK a) {
    WGradPrepDesc d = a.d[0];
#pragma unroll
    for (int i = 1; i < WPREP_MAX; ++i)
        if ((int)blockIdx.y == i) d = a.d[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, g = lane >> 5;
    const int total_chunks = (a.n + 15) / 16;
    float csum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int chunk = blockIdx.x * 4 + wave; chunk < total_chunks; chunk += gridDim.x * 4) {
        const int pb = chunk * 16;
        float v[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = pb + 8 * g + e;
            const bool ok = p < a.n;
            const float* row = d.Y + (size_t)(ok ? p : pb) * d.ldy;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float x = 0.0f;
                if (nb < d.nb) {
                    const int col = 32 * nb + c;
                    x = (ok && col < d.ncols) ? row[col < d.ncols ? col : 0] : 0.0f;
                    csum[nb] += x;
                    if (d.relu) x = fmaxf(x, 0.0f);
                }
                v[nb][e] = x;
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb < d.nb) {
                const Split3 s3 = split8<false>(v[nb]);
                u32x4* q = d.out + (size_t)((chunk * d.nb + nb) * 3) * 64 + lane;
                q[0] = s3.h; q[64] = s3.m; q[128] = s3.l;
            }
        }
    }
    if (d.colsum) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            if (nb < d.nb && 32 * nb + c < d.ncols) atomicAdd(d.colsum + 32 * nb + c, csum[nb]);
    }
}

// main: C(m, n) += sum_p X[p][m] * Ysplit[p][n], m < 128 (wave = row block), n < 32 NB; element (m, n) lives at
// C + m*sm + n*sn and only n < nvalid is written; rowsum[m] += sum_p X[p][m] if given
struct WGradSDesc {
    const float* X; int ldx;
    const u32x4* ysplit;
    float* C; int sm; int sn; int nvalid;
    float* rowsum;
    // a small C (few cache lines) would take one atomic per line from every workgroup, and atomics to one line
    // serialise in its L2 channel: workgroup b then adds into copy b % slabs at C + (b % slabs) * slab_stride
    // (rowsum follows at its own stride) and wgrad_slab_reduce_kernel folds the copies; slabs <= 1: no copies
    int slabs; int slab_stride; int rowsum_slab_stride;
    // X not stored but formed here (the heads with k <= 4 outputs): X[p][m] = [h1[p][m] > 0] sum_o w2[o][m] cot[p][o],
    // from the head's cotangent [n,k], its last-layer weights [k,128] and the forward's ReLU bit masks (layout of
    // relu_mask_build, mask[p * 2 + half]).  cot == NULL: X is read from memory
    const float* cot; int k; const float* w2; const uint2* mask;
};
#define WGS_MAX 6
struct WGradSK {
    int n; int chunks_per_block;
    // heads_together > 0: a one-dimensional grid of 8 * heads_together * ceil(ranges / 8) workgroups in which the
    // workgroups of one point range (one per product) are consecutive multiples of 8 apart: workgroups go to the
    // eight XCDs in turn, so these run side by side on ONE XCD and the Y pieces of the range, which all of them
    // read, come from HBM once and from that XCD's L2 afterwards.  0: grid (ranges, products)
    int heads_together;
    WGradSDesc d[WGS_MAX];
};

template <int NB>
__global__ void __launch_bounds__(256)
deform_wgrad_split_kernel(WGradSK a) {
    int product = blockIdx.y, range = blockIdx.x;
    if (a.heads_together > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        product = j % a.heads_together;
        range = (j / a.heads_together) * 8 + xcd;
    }
    WGradSDesc d = a.d[0];
#pragma unroll
    for (int i = 1; i < WGS_MAX; ++i)
        if (product == i) d = a.d[i];
    if (!d.C) return;
    const int lane = threadIdx.x & 63, mb = threadIdx.x >> 6;
    const int c = lane & 31, g = lane >> 5;
    const int total_chunks = (a.n + 15) / 16;
    const int ch_begin = range * a.chunks_per_block;
    const int ch_end = min(ch_begin + a.chunks_per_block, total_chunks);
    if (ch_begin >= ch_end) return;
    floatx16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    float bsum = 0.0f;
    const float* __restrict__ X = d.X + 32 * mb + c;
    // X is unique to the chunk and the wave (HBM): its eight dwords are requested XA chunks ahead (8 registers
    // per stage).  The Y pieces of a chunk are the same for the four waves of the workgroup: each wave fetches
    // one column block from L2 and they are exchanged through LDS (double buffered, one barrier per chunk);
    // fetched per wave they make this kernel L2-bandwidth bound (12 KiB per 24 MFMAs).
    constexpr int XA = 4;
    __shared__ u32x4 ybuf[2][4 * 3 * 64];
    // on-the-fly X: per wave, the 16 x k cotangents and the 32 mask words of a chunk pass through LDS so that
    // every lane can read those of its eight points (broadcast reads).  They are staged TRANSPOSED - cotangent
    // (point, o) at o * 16 + point, mask word (point, lane half) at half * 16 + point - so that a lane's eight points
    // are eight consecutive words: two 16-byte reads per output and per mask instead of eight 4-byte ones.
    __shared__ __align__(16) float fly_cot[4][64];
    __shared__ __align__(16) unsigned fly_msk[4][32];
    const bool fly = d.cot != nullptr;
    // four scalars, not an array: captured by the loop's lambdas an array stayed in scratch memory, and its reload
    // in every chunk - a vector-memory operation, the youngest one - made the wave wait for every prefetch in flight
    float w2v0 = 0.0f, w2v1 = 0.0f, w2v2 = 0.0f, w2v3 = 0.0f;
    if (fly) {                                   // w2 == NULL: unit weights (X = mask * cot, k = 1)
        const float* __restrict__ wsrc = d.w2 ? d.w2 + 32 * mb + c : nullptr;
        if (0 < d.k) w2v0 = wsrc ? wsrc[0 * DW] : 1.0f;
        if (1 < d.k) w2v1 = wsrc ? wsrc[1 * DW] : 1.0f;
        if (2 < d.k) w2v2 = wsrc ? wsrc[2 * DW] : 1.0f;
        if (3 < d.k) w2v3 = wsrc ? wsrc[3 * DW] : 1.0f;
    }
    // feature 32 mb + c sits in the mask of lane half (c >> 2) & 1, word mb >> 1, at bit 31 - (16 (mb & 1) + r)
    // with r = (c & 3) + 4 ((c >> 3) & 3)
    const int fly_half = (c >> 2) & 1;
    const int fly_bit = 31 - (16 * (mb & 1) + (c & 3) + 4 * ((c >> 3) & 3));
    // staging slots of the words this lane loads: cotangent word `lane` is (point lane / k, output lane % k), mask
    // word `lane` (< 32) is (point lane >> 1, lane half lane & 1); lanes beyond 16 k hold zeros and keep slot `lane`,
    // which no reader uses (readers touch slots below 16 k)
    int cot_slot = lane;
    if (fly && lane < 16 * d.k) cot_slot = (lane % d.k) * 16 + lane / d.k;
    const int msk_slot = ((lane & 1) << 4) | ((lane >> 1) & 15);
    // Every prefetch below is an unconditional load from a clamped address (a chunk index beyond the range's end
    // repeats the last chunk, a point beyond n repeats the last point; such values are zeroed where they are
    // consumed), the loop body has no branch around a load, and the two ways of getting X are two instances of the
    // loop under one branch at the top.  Loads inside branches - even uniform ones - made the compiler's wait-count
    // bookkeeping pessimistic at the joins: it waited for ALL outstanding loads once per chunk, and a chunk then
    // took one memory latency however far ahead its operands were requested.  The price: a range whose length is
    // not a multiple of XA runs up to XA - 1 chunks of zeros (the host makes the ranges multiples of XA, so this
    // happens in the last range only).
    const int ch_last = ch_end - 1;
    int it = 0;
    auto run = [&](auto fly_c) {
        constexpr bool FLY = decltype(fly_c)::value;
        float cotr[XA];
        uint2 mskr[XA];
        float xr[XA][8];
        // the request keeps the words as they arrive; what is valid is decided where they are used, XA chunks later
        // (a select at the request would wait for the load at once)
        const long long cot_last = (long long)a.n * d.k - 1, msk_last = 2ll * a.n - 1;
        auto load_fly = [&](int ch, float& cv, uint2& mv) {
            const int pb = min(ch, ch_last) * 16;
            const long long ci = (long long)pb * d.k + lane, mi = (long long)pb * 2 + lane;
            cv = d.cot[ci < cot_last ? ci : cot_last];
            mv = d.mask[mi < msk_last ? mi : msk_last];
        };
        auto use_fly = [&](int ch, float cv, uint2 mv, float& co, unsigned& mo) {
            const bool real = ch <= ch_last;
            const long long ci = (long long)ch * 16 * d.k + lane, mi = (long long)ch * 32 + lane;
            co = (real && lane < 16 * d.k && ci <= cot_last) ? cv : 0.0f;
            mo = (real && lane < 32 && mi <= msk_last) ? ((mb >> 1) ? mv.y : mv.x) : 0u;
        };
        auto load_x = [&](int ch, float* xo) {
            const int pb = min(ch, ch_last) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) xo[e] = X[(size_t)min(pb + 8 * g + e, a.n - 1) * d.ldx];
        };
#pragma unroll
        for (int i = 0; i < XA; ++i) {
            if constexpr (FLY) load_fly(ch_begin + i, cotr[i], mskr[i]);
            else load_x(ch_begin + i, xr[i]);
        }
        // This wave's share (column block mb) of the Y pieces of the next XA chunks, requested as far ahead as X
        // (loads return in order: a piece requested in the iteration that hands it to LDS would make that iteration
        // wait for every older request too).  Ring slot k % XA holds chunk ch_begin + k.
        Split3 yst[XA];
        if (NB == 4 || mb < NB) {
            const Split3 y0 = sfrag_load(d.ysplit, ch_begin, NB, mb, lane);
#pragma unroll
            for (int k = 1; k <= XA; ++k) yst[k % XA] = sfrag_load(d.ysplit, min(ch_begin + k, ch_last), NB, mb, lane);
            ybuf[0][(mb * 3 + 0) * 64 + lane] = y0.h;
            ybuf[0][(mb * 3 + 1) * 64 + lane] = y0.m;
            ybuf[0][(mb * 3 + 2) * 64 + lane] = y0.l;
        }
        __syncthreads();
        // X of chunk `ch` as split operand pieces, from ring stage i (static); refills the stage for chunk ch + XA.
        // It is called for the NEXT chunk right after the current chunk's MFMAs are issued, so that its vector and
        // LDS work runs under the matrix pipeline.
        auto fly_x = [&](auto kc, float* xv) {                 // kc: the head's output count as a compile-time value
            constexpr int K = decltype(kc)::value;
            const float4* cq = reinterpret_cast<const float4*>(&fly_cot[mb][0]) + 2 * g;
            const uint4* mq = reinterpret_cast<const uint4*>(&fly_msk[mb][0]) + 4 * fly_half + 2 * g;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.0f;
#pragma unroll
            for (int o = 0; o < K; ++o) {
                const float4 lo = cq[4 * o], hi = cq[4 * o + 1];
                const float cv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const float wo = o == 0 ? w2v0 : o == 1 ? w2v1 : o == 2 ? w2v2 : w2v3;     // o is a compile-time index
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaf(wo, cv[e], x[e]);
            }
            const uint4 mlo = mq[0], mhi = mq[1];
            const unsigned mw[8] = {mlo.x, mlo.y, mlo.z, mlo.w, mhi.x, mhi.y, mhi.z, mhi.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int masked = __builtin_amdgcn_sbfe((int)mw[e], fly_bit, 1);
                xv[e] = __uint_as_float(__float_as_uint(x[e]) & ~(unsigned)masked);
            }
        };
        auto make_x = [&](int ch, int i) -> Split3 {
            float xv[8];
            if constexpr (FLY) {
                float co;
                unsigned mo;
                use_fly(ch, cotr[i], mskr[i], co, mo);
                fly_cot[mb][cot_slot] = co;
                if (lane < 32) fly_msk[mb][msk_slot] = mo;
                load_fly(ch + XA, cotr[i], mskr[i]);
                if (d.k == 4) fly_x(std::integral_constant<int, 4>{}, xv);
                else if (d.k == 3) fly_x(std::integral_constant<int, 3>{}, xv);
                else if (d.k == 2) fly_x(std::integral_constant<int, 2>{}, xv);
                else fly_x(std::integral_constant<int, 1>{}, xv);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = xr[i][e];
                if (ch * 16 + 16 > a.n || ch > ch_last) {    // a partial or a padding chunk: rows beyond the data are zero
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ch > ch_last || ch * 16 + 8 * g + e >= a.n) xv[e] = 0.0f;
                }
                load_x(ch + XA, xr[i]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum += xv[e];
            return split8<false>(xv);
        };
        Split3 xs = make_x(ch_begin, 0);
        for (int ch0 = ch_begin; ch0 < ch_end; ch0 += XA) {
#pragma unroll
            for (int i = 0; i < XA; ++i) {           // ring position i is static: the stages stay in registers
                const int ch = ch0 + i;
                const int cur = it & 1;
                Split3 ys[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    ys[nb].h = ybuf[cur][(nb * 3 + 0) * 64 + lane];
                    ys[nb].m = ybuf[cur][(nb * 3 + 1) * 64 + lane];
                    ys[nb].l = ybuf[cur][(nb * 3 + 2) * 64 + lane];
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = mma_split(xs, ys[nb], acc[nb]);
                xs = make_x(ch + 1, (i + 1) % XA);
                if (NB == 4 || mb < NB) {
                    Split3& yn = yst[(i + 1) % XA];             // chunk ch + 1, requested XA chunks ago
                    ybuf[cur ^ 1][(mb * 3 + 0) * 64 + lane] = yn.h;
                    ybuf[cur ^ 1][(mb * 3 + 1) * 64 + lane] = yn.m;
                    ybuf[cur ^ 1][(mb * 3 + 2) * 64 + lane] = yn.l;
                    yn = sfrag_load(d.ysplit, min(ch + 1 + XA, ch_last), NB, mb, lane);
                }
                __syncthreads();
                ++it;
            }
        }
    };
    if (fly) run(std::true_type{});
    else run(std::false_type{});
    const int slab = d.slabs > 1 ? (int)((unsigned)range % (unsigned)d.slabs) : 0;
    float* __restrict__ Cs = d.C + (size_t)slab * d.slab_stride;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = 32 * nb + c;
        if (n < d.nvalid) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                atomicAdd(Cs + (size_t)(32 * mb + drow(r, g)) * d.sm + (size_t)n * d.sn, acc[nb][r]);
        }
    }
    if (d.rowsum) atomicAdd(d.rowsum + (size_t)slab * d.rowsum_slab_stride + 32 * mb + c, bsum);
}

// out[e][i] += sum over the copies of slab[e][s * stride[e] + i]  (i < count[e]) for up to SLAB_ENTRIES destinations
#define SLAB_ENTRIES 12
struct SlabReduceK { const float* slab[SLAB_ENTRIES]; float* out[SLAB_ENTRIES]; int count[SLAB_ENTRIES];
                     int stride[SLAB_ENTRIES]; int slabs; };
__global__ void __launch_bounds__(256)
wgrad_slab_reduce_kernel(SlabReduceK a) {
    const float* __restrict__ src = a.slab[0]; float* __restrict__ dst = a.out[0];
    int count = a.count[0], stride = a.stride[0];
#pragma unroll
    for (int e = 1; e < SLAB_ENTRIES; ++e)
        if ((int)blockIdx.y == e) { src = a.slab[e]; dst = a.out[e]; count = a.count[e]; stride = a.stride[e]; }
    if (!dst) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        float s = 0.0f;
        for (int k = 0; k < a.slabs; ++k) s += src[(size_t)k * stride + i];
        dst[i] += s;
    }
}


// ===========================================================================
// Backward, pass 2b (split form): dW2_h[o][f] = sum_p g_h[p][o] * a1_h[p][f] with
// a1_h = relu(W1_h relu(hidden) + b1_h) RECOMPUTED in the orientation the product
// needs, so that relu(h1) is never stored:
//   C1[m = point][n = f] = sum_k relu(hidden)[point][k] W1[f][k]  A = the split relu(hidden) pieces the forward
//                                                                 saved, B = the forward's W1 fragments
// leaves lane (f, g) holding, in accumulator registers 8jj..8jj+7, eight points
// of column f: exactly the B operand (K = points) of
//   C2[m = o][n = f] += sum_points g[point][o] a1[point][f]      A = g columns, lane o
// One wave = (range of points, head); 64 points per step.
// ===========================================================================
struct WGrad2SK {
    int n; int tiles_per_block;                  // 64-point tiles per wave
    const u32x4* hsplit;                         // relu(hidden) pieces saved by the forward: [ceil(n/32)][8][3][64]
    const u32x4* pw1n[5];                        // split W1 (the forward's fragments)
    const float* b1[5];
    const float* g[5];                           // cotangents [n,k]
    float* dw2[5];                               // [k,128]
    // private copies of the 59 x 128 block of all five dW2 (rows 0-2 pos, 3-5 scales, 6-9 rotations, 10 opacity,
    // 11-58 shs): wave b adds into copy b % slabs (every wave hits every line of its heads, and atomics to one
    // line serialise in its L2 channel); wgrad_slab_reduce_kernel folds them into dw2.  NULL: straight into dw2
    float* dw2_slabs; int slabs;
    int small_heads;                             // NOB = 1 walks heads 0 .. small_heads - 1 (3: the opacity head's dW2
                                                 // comes from deform_bwd_finish_kernel instead)
};

// NOB = 2: the shs head alone (48 rows = two row blocks).  NOB = 1: the four small heads TOGETHER, their
// 3 + 3 + 4 + 1 rows packed into one 32-row block (row offsets 0, 3, 6, 10): the A operand of a head is zero
// outside its rows, so the four products accumulate into the same block and the saved relu(hidden) pieces of a
// tile are fetched from HBM once for the four heads.
#ifndef FDGS_W2_FRAG_AHEAD
#define FDGS_W2_FRAG_AHEAD 2
#endif
template <int NOB>
__global__ void __launch_bounds__(64, 1)
deform_wgrad2_split_kernel(WGrad2SK a) {
    const int lane = threadIdx.x;
    const int c = lane & 31, g = lane >> 5;
    const int total_tiles = (a.n + 63) / 64;
    const int t_begin = blockIdx.x * a.tiles_per_block;
    const int t_end = min(t_begin + a.tiles_per_block, total_tiles);
    if (t_begin >= t_end) return;
    const int ngroups = (a.n + 31) / 32;
    const int head_first = NOB == 1 ? 0 : 4, head_last = NOB == 1 ? a.small_heads - 1 : 4;
    floatx16 c2[NOB][4];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) c2[ob][fb][r] = 0.0f;
    // product 1, K = 128 input features; both operands arrive as ready pieces (1 KiB coalesced loads)
    // the relu(hidden) pieces are unique to the tile (HBM latency): requested HS_AHEAD K chunks ahead
    constexpr int HS_AHEAD = 3;
    constexpr int FA = FDGS_W2_FRAG_AHEAD;     // weight fragments requested FA steps ahead
    Split3 f[FA + 1], hs[HS_AHEAD + 1][2];
    // The first pieces of a (tile, head) step are requested during the second product of the step before it (and
    // here for the first one): a wave has its SIMD to itself, and loads consumed right behind their request cost it
    // a memory latency at the head of every step.
    auto request_first = [&](int tile, int head) {
        const u32x4* pwn = sel5(a.pw1n, head);
#pragma unroll
        for (int t = 0; t < FA; ++t) f[t] = sfrag_load_u(pwn, (t & 3) * 8 + (t >> 2), lane);
#pragma unroll
        for (int jj = 0; jj < HS_AHEAD; ++jj)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                hs[jj][mb] = sfrag_load_u(a.hsplit, min(tile * 2 + mb, ngroups - 1) * 8 + jj, lane);
    };
    request_first(t_begin, head_first);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int p0 = tile * 64;
        int G[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) G[mb] = min(tile * 2 + mb, ngroups - 1);
        for (int head = head_first; head <= head_last; ++head) {
            const int k = head_dim(head);
            const int ro = NOB == 1 ? (head == 0 ? 0 : head == 1 ? 3 : head == 2 ? 6 : 10) : 0;   // first row
            const u32x4* pw = sel5(a.pw1n, head);
            const float* __restrict__ gsrc = sel5(a.g, head);
            const float* __restrict__ b1 = sel5(a.b1, head);
            float gv[NOB][4][8];
            floatx16 c1[2][4];                   // [point block][feature block]
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const float bias = b1[32 * fb + c];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) c1[mb][fb][r] = bias;
            }
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const int j = t >> 2, fb = t & 3;
                if (t + FA < 32) f[(t + FA) % (FA + 1)] = sfrag_load_u(pw, ((t + FA) & 3) * 8 + ((t + FA) >> 2), lane);
                if (fb == 0 && j + HS_AHEAD < 8) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        hs[(j + HS_AHEAD) % (HS_AHEAD + 1)][mb] = sfrag_load_u(a.hsplit, G[mb] * 8 + j + HS_AHEAD, lane);
                }
                if (t == 22) {
                    // this step's cotangents: row o of the block, eight points per K chunk (mb, jj); requested late
                    // in product 1 so that they are not live across all of it
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) {
                        const int o = 32 * ob + c - ro;
#pragma unroll
                        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int p = p0 + 32 * (ch >> 1) + drow(8 * (ch & 1) + e, g);
                                gv[ob][ch][e] = (o >= 0 && o < k && p < a.n) ? gsrc[(size_t)p * k + o] : 0.0f;
                            }
                    }
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) c1[mb][fb] = mma_split(hs[j % (HS_AHEAD + 1)][mb], f[t % (FA + 1)], c1[mb][fb]);
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // the step after this one (the last one of the wave repeats itself: a few unused loads)
                const bool more_heads = head < head_last;
                const int tile_n = more_heads ? tile : min(tile + 1, t_end - 1);
                const int head_n = more_heads ? head + 1 : (tile + 1 < t_end ? head_first : head);
                request_first(tile_n, head_n);
            }
            // product 2, K = the 64 points of this step: chunk (mb, jj) = points p0 + 32 mb + drow(8 jj + e, g)
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                Split3 gs[NOB];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) gs[ob] = split8<false>(gv[ob][ch]);
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    const Split3 as = split_acc_here<true>(c1[ch >> 1][fb], ch & 1);
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) c2[ob][fb] = mma_split(gs[ob], as, c2[ob][fb]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * ob + drow(r, g);
                int head = 4, o = row;
                if (NOB == 1) {
                    head = row < 3 ? 0 : row < 6 ? 1 : row < 10 ? 2 : 3;
                    o = row - (head == 0 ? 0 : head == 1 ? 3 : head == 2 ? 6 : 10);
                }
                float* dw2 = sel5(a.dw2, head);
                if (row < (NOB == 1 ? 11 : 48) && dw2) {
                    if (a.dw2_slabs) {
                        float* dst = a.dw2_slabs + (size_t)(blockIdx.x % (unsigned)a.slabs) * (59 * DW) +
                                     (size_t)((NOB == 1 ? 0 : 11) + row) * DW;
                        atomicAdd(dst + 32 * fb + c, c2[ob][fb][r]);
                    } else {
                        atomicAdd(dw2 + (size_t)o * DW + 32 * fb + c, c2[ob][fb][r]);
                    }
                }
            }
}

// A head with ONE output needs no recompute of h1 for its dW2.  With X[p][f] = [h1[p][f] > 0] g[p] (unit weights in
// the dW1 product above) the kernel above yields Q[f][n] = sum_p X[p][f] relu(hidden)[p][n] and T[f] = sum_p X[p][f],
// and, h1 being W1 relu(hidden) + b1 where the mask is set,
//   dW1[f][n] = w2[f] Q[f][n],   db1[f] = w2[f] T[f],   dW2[f] = sum_n W1[f][n] Q[f][n] + b1[f] T[f].
// One workgroup per row f (the second half of deform_bwd_finish_kernel).
struct OpacityFoldK { const float* Q; const float* T; const float* w1; const float* b1; const float* w2;
                      float* dw1; float* db1; float* dw2; };

// wgrad_slab_reduce_kernel and the fold above in one launch (they end the backward and touch disjoint outputs,
// provided the opacity head's dW2 is not among the slab destinations): workgroups (x, y < SLAB_ENTRIES) fold the
// copies of entry y, workgroups (x, y >= SLAB_ENTRIES) take row f = 8 (y - SLAB_ENTRIES) + x of the opacity fold.
__global__ void __launch_bounds__(256)
deform_bwd_finish_kernel(SlabReduceK a, OpacityFoldK o) {
    __shared__ float part[2];
    if ((int)blockIdx.y < SLAB_ENTRIES) {
        const float* __restrict__ src = a.slab[0]; float* __restrict__ dst = a.out[0];
        int count = a.count[0], stride = a.stride[0];
#pragma unroll
        for (int e = 1; e < SLAB_ENTRIES; ++e)
            if ((int)blockIdx.y == e) { src = a.slab[e]; dst = a.out[e]; count = a.count[e]; stride = a.stride[e]; }
        if (!dst) return;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
            float s = 0.0f;
            for (int k = 0; k < a.slabs; ++k) s += src[(size_t)k * stride + i];
            dst[i] += s;
        }
        return;
    }
    const int f = 8 * ((int)blockIdx.y - SLAB_ENTRIES) + (int)blockIdx.x, n = threadIdx.x;
    const bool live = f < DW && n < DW;                       // uniform per wave: DW is a multiple of 64
    float s = 0.0f, w2f = 0.0f;
    if (live) {
        const float q = o.Q[f * DW + n];
        w2f = o.w2[f];
        if (o.dw1) o.dw1[f * DW + n] += w2f * q;
        s = o.w1[f * DW + n] * q;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (live && (n & 63) == 0) part[n >> 6] = s;
    __syncthreads();
    if (live && n == 0) {
        const float t = o.T[f];
        if (o.dw2) o.dw2[f] += part[0] + part[1] + o.b1[f] * t;
        if (o.db1) o.db1[f] += w2f * t;
    }
}

// column sums of a small row-major matrix: out[j] += sum_p X[p][j], j < ncols <= 64 (the b2 gradients)
struct ColSumK { const float* X[5]; int ncols[5]; float* out[5]; int n;
                 float* slabs_base; int slabs; };   // copies [slabs][5][64] instead of out (same reason as above)
__global__ void __launch_bounds__(256)
colsum_kernel(ColSumK a) {
    const int hd = blockIdx.y;
    float* out = sel5(a.out, hd);
    if (!out) return;
    const float* __restrict__ X = sel5(a.X, hd);
    const int nc = sel5(a.ncols, hd);
    __shared__ float acc[64];
    if (threadIdx.x < 64) acc[threadIdx.x] = 0.0f;
    __syncthreads();
    // flat walk with a stride that is a multiple of the column count: a thread stays in one column
    const long long total = (long long)a.n * nc;
    const long long stride = ((long long)gridDim.x * 256 / nc) * nc;
    const long long tg = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tg < stride) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;     // four loads in flight per thread
        long long i = tg;
        for (; i + 3 * stride < total; i += 4 * stride) {
            s0 += X[i]; s1 += X[i + stride]; s2 += X[i + 2 * stride]; s3 += X[i + 3 * stride];
        }
        for (; i < total; i += stride) s0 += X[i];
        atomicAdd(&acc[tg % nc], (s0 + s1) + (s2 + s3));
    }
    __syncthreads();
    if ((int)threadIdx.x < nc) {
        float* dst = a.slabs_base ? a.slabs_base + ((size_t)(blockIdx.x % (unsigned)a.slabs) * 5 + hd) * 64 : out;
        atomicAdd(dst + threadIdx.x, acc[threadIdx.x]);
    }
}
