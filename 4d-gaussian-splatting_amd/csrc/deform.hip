// HexPlane deformation field for gfx950: fused plane interpolation + MLP heads.
//
// Orientation.  Every activation tile is kept TRANSPOSED in the MFMA accumulator
// layout: rows = features (register index), columns = points (lane & 31).  A wave
// owns 32 points; lane l = (c = l & 31, h = l >> 5) belongs to point c.  With
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, D = A*B + C, K = 2 per
// instruction) lane l supplies A[i = c][k = h] and B[k = h][j = c], and holds
// D[row(r,h)][j = c] in register r with row(r,h) = (r&3) + 8*(r>>2) + 4*h.
// A layer Y^T = W X^T therefore takes the previous layer's accumulator registers
// directly as its B operand (register r of the k-th 32-row block is the K pair
// {32k+row(r,0), 32k+row(r,1)}), and the weights W[n][k] as the A operand: four
// consecutive steps r = 4t..4t+3 need W[n][32k + 8t + 4h + 0..3], one 16-byte load
// from a copy of the weights re-packed in exactly that fragment order.
//
// The two halves of a wave also split the HexPlane work: lane (c,h) interpolates
// the 16 channels of resolution level h for point c, which is exactly the K pair
// {s, 16+s} of the first Linear, so the features never move between lanes.
//
// Planes are gathered from a channel-last copy [H][W][16] (one 64-byte texel per
// lookup instead of 16 strided lines), rebuilt from the parameter layout
// [16][H][W] at the start of every call (9.5 MB, a few microseconds).
#include "fdgs_common.h"
#include <cstdlib>
#include <type_traits>

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define DW FDGS_DEFORM_WIDTH             // 128
#define DF FDGS_DEFORM_FEAT              // 16
#define TILE_PTS 128                     // points per workgroup (4 waves x 32)

// plane ci pairs axes (a,b) = itertools.combinations(range(4), 2)[ci]  (src/model/hexplane.py:96-97):
// (0,1) (0,2) (0,3) (1,2) (1,3) (2,3).  constexpr so that unrolled loops index registers statically.
__host__ __device__ __forceinline__ constexpr int plane_a(int ci) { return ci < 3 ? 0 : (ci < 5 ? 1 : 2); }
__host__ __device__ __forceinline__ constexpr int plane_b(int ci) {
    return ci == 0 ? 1 : ci == 1 ? 2 : ci == 2 ? 3 : ci == 3 ? 2 : 3;
}

__device__ __forceinline__ int drow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <typename T>
__device__ __forceinline__ T sel5(const T* v, int i) {
    T r = v[0];
    r = i == 1 ? v[1] : r; r = i == 2 ? v[2] : r; r = i == 3 ? v[3] : r; r = i == 4 ? v[4] : r;
    return r;
}
__device__ __forceinline__ int head_dim(int head) { return head == 0 ? 3 : head == 1 ? 3 : head == 2 ? 4 : head == 3 ? 1 : 48; }

// ---------------------------------------------------------------------------
// [16][H][W] -> [H][W][16] for all twelve planes in one launch, and the reverse
// (accumulating, +=) for the gradients.  blockIdx.y selects the plane.
// ---------------------------------------------------------------------------
struct PlaneTable {
    const float* src[12];
    float* dst[12];
    int hw[12];
};

__global__ void __launch_bounds__(256)
planes_to_channel_last_kernel(PlaneTable t) {
    const float* __restrict__ src = t.src[0]; float* __restrict__ dst = t.dst[0]; int hw = t.hw[0];
#pragma unroll
    for (int i = 1; i < 12; ++i)
        if ((int)blockIdx.y == i) { src = t.src[i]; dst = t.dst[i]; hw = t.hw[i]; }
    for (int tx = blockIdx.x * blockDim.x + threadIdx.x; tx < hw; tx += gridDim.x * blockDim.x) {
        float v[DF];
#pragma unroll
        for (int ch = 0; ch < DF; ++ch) v[ch] = src[(size_t)ch * hw + tx];
        float4* d = reinterpret_cast<float4*>(dst + (size_t)tx * DF);
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

__global__ void __launch_bounds__(256)
planes_grad_from_channel_last_kernel(PlaneTable t) {   // src = channel-last gradients, dst = parameter-layout gradients
    const float* __restrict__ src = t.src[0]; float* __restrict__ dst = t.dst[0]; int hw = t.hw[0];
#pragma unroll
    for (int i = 1; i < 12; ++i)
        if ((int)blockIdx.y == i) { src = t.src[i]; dst = t.dst[i]; hw = t.hw[i]; }
    if (!dst) return;
    for (int tx = blockIdx.x * blockDim.x + threadIdx.x; tx < hw; tx += gridDim.x * blockDim.x) {
        const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)tx * DF);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = s4[q];
            dst[(size_t)(4 * q + 0) * hw + tx] += v.x;
            dst[(size_t)(4 * q + 1) * hw + tx] += v.y;
            dst[(size_t)(4 * q + 2) * hw + tx] += v.z;
            dst[(size_t)(4 * q + 3) * hw + tx] += v.w;
        }
    }
}

// ---------------------------------------------------------------------------
// Plane addressing shared by forward and backward
// ---------------------------------------------------------------------------
struct PlaneSet {                        // two levels x six planes, channel-last
    const float* pl[2][6];
    int res[2][4];
};

struct Bilinear {                        // one lookup: 4 texel offsets (in texels) and weights
    int o00, o01, o10, o11;
    int x0, y0;                          // texel coordinates of corner 00
    float w00, w01, w10, w11;
    float fx, fy;                        // fractional parts
    bool in_x, in_y;                     // coordinate strictly inside (gradient flows)
};

// align_corners=True un-normalisation + border clipping of F.grid_sample
// (the call at src/model/hexplane.py:42-46).
__device__ __forceinline__ void unnorm_clip(float c, int size, float& x, bool& inside) {
    x = ((c + 1.0f) * 0.5f) * (float)(size - 1);
    inside = (x > 0.0f) && (x < (float)(size - 1));
    x = fminf(fmaxf(x, 0.0f), (float)(size - 1));
    if (!(x == x)) x = 0.0f;
}

__device__ __forceinline__ Bilinear make_bilinear(float ca, float cb, int Wd, int Hd) {
    Bilinear b;
    float x, y;
    unnorm_clip(ca, Wd, x, b.in_x);
    unnorm_clip(cb, Hd, y, b.in_y);
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = min(x0 + 1, Wd - 1), y1 = min(y0 + 1, Hd - 1);
    b.fx = x - x0f;
    b.fy = y - y0f;
    b.w00 = (1.0f - b.fx) * (1.0f - b.fy);
    b.w01 = b.fx * (1.0f - b.fy);
    b.w10 = (1.0f - b.fx) * b.fy;
    b.w11 = b.fx * b.fy;
    b.o00 = y0 * Wd + x0; b.o01 = y0 * Wd + x1; b.o10 = y1 * Wd + x0; b.o11 = y1 * Wd + x1;
    b.x0 = x0; b.y0 = y0;
    return b;
}

// one coordinate axis of a bilinear lookup: lower texel, step to the upper one (0 at the border), fraction,
// and whether the coordinate is strictly inside (same arithmetic as make_bilinear)
struct AxisLookup { int i0, d; float f; bool in; };
__device__ __forceinline__ AxisLookup make_axis(float c, int size) {
    AxisLookup a;
    float x;
    unnorm_clip(c, size, x, a.in);
    const float x0f = floorf(x);
    a.i0 = (int)x0f;
    a.d = min(a.i0 + 1, size - 1) - a.i0;
    a.f = x - x0f;
    return a;
}

// normalised 4-D coordinate of a point: (xyz - aabb[0]) * (2/(aabb[1]-aabb[0])) - 1, then raw time
// (src/model/hexplane.py:14-17,185-186).
__device__ __forceinline__ void point_coords(const float* __restrict__ xyz, const float* __restrict__ aabb,
                                             const float* __restrict__ time, float time_scalar, int p, float* co,
                                             float* scale3) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float s = 2.0f / (aabb[3 + k] - aabb[k]);
        scale3[k] = s;
        co[k] = (xyz[3 * (size_t)p + k] - aabb[k]) * s - 1.0f;
    }
    co[3] = time ? time[p] : time_scalar;
}

// 16 channels of one level for one point: product over the six planes.
// vals (optional) receives the per-plane interpolated values for the backward.
__device__ __forceinline__ void level_features(const PlaneSet& ps, int h, const float* co, float* feat,
                                               float (*vals)[DF]) {
#pragma unroll
    for (int ch = 0; ch < DF; ++ch) feat[ch] = 1.0f;
#pragma unroll
    for (int ci = 0; ci < 6; ++ci) {
        const int a = plane_a(ci), b = plane_b(ci);
        const int Wd = h ? ps.res[1][a] : ps.res[0][a];
        const int Hd = h ? ps.res[1][b] : ps.res[0][b];
        const float4* __restrict__ pl = reinterpret_cast<const float4*>(h ? ps.pl[1][ci] : ps.pl[0][ci]);
        const Bilinear bl = make_bilinear(co[a], co[b], Wd, Hd);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t00 = pl[(size_t)bl.o00 * 4 + q], t01 = pl[(size_t)bl.o01 * 4 + q];
            const float4 t10 = pl[(size_t)bl.o10 * 4 + q], t11 = pl[(size_t)bl.o11 * 4 + q];
            float v[4];
            v[0] = t00.x * bl.w00 + t01.x * bl.w01 + t10.x * bl.w10 + t11.x * bl.w11;
            v[1] = t00.y * bl.w00 + t01.y * bl.w01 + t10.y * bl.w10 + t11.y * bl.w11;
            v[2] = t00.z * bl.w00 + t01.z * bl.w01 + t10.z * bl.w10 + t11.z * bl.w11;
            v[3] = t00.w * bl.w00 + t01.w * bl.w01 + t10.w * bl.w10 + t11.w * bl.w11;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                feat[4 * q + k] *= v[k];
                if (vals) vals[ci][4 * q + k] = v[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Weight fragments.  At the start of every call each weight matrix is re-packed
// (one small launch, 0.4-0.7 MB) into the order the MFMA A operand consumes it:
// for row block ib (32 output rows) and K group T (8 reduction indices) the 64
// lanes' float4 are contiguous, lane (c,h) holding element(32 ib + c, kk) for
// kk = 8T + 4h + q, q = 0..3 (the K pairs {row(r,0), row(r,1)} of four
// consecutive accumulator registers).  Every A-operand fetch is then one fully
// coalesced 1-KiB global_load_dwordx4 from L2; no LDS, no workgroup barrier.
// ---------------------------------------------------------------------------
#define FRAG_TRANSPOSE 1     // element(i,kk) = src[kk*ld + i]   (else src[i*ld + kk])
#define FRAG_FEATMAP 2       // kk = 16h + 4T + q  (first Linear: lane half h owns level h's 16 features)
#define FRAG_PERMROWS 4      // row i carries column perm(i) = 16*((i>>2)&1) + 4*(i>>3) + (i&3)

struct FragDesc { const float* src; float* dst; int I; int K; int ld; int mode; };
#define FRAG_MAX 36
struct FragTable { FragDesc d[FRAG_MAX]; int n; };

__host__ __device__ __forceinline__ int frag_floats(int I, int K) { return ((I + 31) / 32) * ((K + 7) / 8) * 256; }

__global__ void __launch_bounds__(256)
pack_fragments_kernel(FragTable t) {
    FragDesc d = t.d[0];
#pragma unroll
    for (int i = 1; i < FRAG_MAX; ++i)
        if ((int)blockIdx.y == i) d = t.d[i];
    if ((int)blockIdx.y >= t.n) return;
    const int G = (d.K + 7) / 8, IB = (d.I + 31) / 32;
    const int total = IB * G * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int l = idx & 63, T = (idx >> 6) % G, ib = (idx >> 6) / G;
        const int c = l & 31, h = l >> 5;
        const int i = 32 * ib + c;
        const int col = (d.mode & FRAG_PERMROWS) ? (16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)) : i;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = (d.mode & FRAG_FEATMAP) ? (16 * h + 4 * T + q) : (8 * T + 4 * h + q);
            float x = 0.0f;
            if (i < d.I && kk < d.K)
                x = (d.mode & FRAG_TRANSPOSE) ? d.src[(size_t)kk * d.ld + col] : d.src[(size_t)col * d.ld + kk];
            v[q] = x;
        }
        reinterpret_cast<float4*>(d.dst)[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float4 frag_load(const float* packed, int ib, int G, int T, int lane) {
    return reinterpret_cast<const float4*>(packed)[(size_t)(ib * G + T) * 64 + lane];
}

// acc (+)= W[row block ib] * relu?(X) for a 128-wide X held in x[4] (accumulator layout)
template <bool RELU>
__device__ __forceinline__ void layer128(const float* packed, int ib, int lane, const floatx16* x, floatx16& acc) {
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        const float4 a4 = frag_load(packed, ib, 16, T, lane);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float xv = x[T >> 2][4 * (T & 3) + q];
            if (RELU) xv = fmaxf(xv, 0.0f);
            acc = MFMA(av[q], xv, acc);
        }
    }
}

// all four 32-row blocks of a 128x128 layer at once: one B value feeds four independent accumulators
template <bool RELU>
__device__ __forceinline__ void layer128x4(const float* packed, int lane, const floatx16* x, floatx16* acc) {
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        float av[4][4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const float4 a4 = frag_load(packed, nb, 16, T, lane);
            av[nb][0] = a4.x; av[nb][1] = a4.y; av[nb][2] = a4.z; av[nb][3] = a4.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float xv = x[T >> 2][4 * (T & 3) + q];
            if (RELU) xv = fmaxf(xv, 0.0f);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = MFMA(av[nb][q], xv, acc[nb]);
        }
    }
}

__device__ __forceinline__ floatx16 bias_init(const float* __restrict__ bias, int nb, int h, int valid_rows) {
    floatx16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * nb + drow(r, h);
        v[r] = row < valid_rows ? bias[row] : 0.0f;
    }
    return v;
}

// ---------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------
struct DeformFwdK {
    int n;
    PlaneSet ps;
    const float* aabb;
    const float* pw0; const float* b0;              // packed first Linear
    const float* pw1[5]; const float* b1[5]; const float* pw2[5]; const float* b2[5];
    const float* xyz; const float* in[5];           // in[0] = xyz ... in[4] = shs (residual inputs)
    const float* in_rest;                           // NULL, or in[4] is f_dc [n,3] and this is f_rest [n,45]
    const float* time; float time_scalar;
    float* out[5];
    float* save_hidden;
};

__global__ void __launch_bounds__(256, 2)
deform_fwd_kernel(DeformFwdK a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int p = blockIdx.x * TILE_PTS + wave * 32 + c;
    const bool pvalid = p < a.n;
    const int pc = pvalid ? p : a.n - 1;

    float co[4], sc3[3], feat[DF];
    point_coords(a.xyz, a.aabb, a.time, a.time_scalar, pc, co, sc3);
    level_features(a.ps, h, co, feat, nullptr);

    // hidden^T = W0 f^T + b0
    floatx16 hid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        hid[nb] = bias_init(a.b0, nb, h, DW);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const float4 a4 = frag_load(a.pw0, nb, 4, T, lane);
            hid[nb] = MFMA(a4.x, feat[4 * T + 0], hid[nb]);
            hid[nb] = MFMA(a4.y, feat[4 * T + 1], hid[nb]);
            hid[nb] = MFMA(a4.z, feat[4 * T + 2], hid[nb]);
            hid[nb] = MFMA(a4.w, feat[4 * T + 3], hid[nb]);
        }
    }
    if (a.save_hidden && pvalid) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(a.save_hidden + (size_t)p * DW + 32 * nb + 8 * q + 4 * h) =
                    make_float4(hid[nb][4 * q], hid[nb][4 * q + 1], hid[nb][4 * q + 2], hid[nb][4 * q + 3]);
    }

    for (int head = 0; head < 5; ++head) {
        const int k = head_dim(head);
        const float* pw1 = sel5(a.pw1, head); const float* b1 = sel5(a.b1, head);
        const float* pw2 = sel5(a.pw2, head); const float* b2 = sel5(a.b2, head);
        const float* in = sel5(a.in, head); float* out = sel5(a.out, head);
        floatx16 h1[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) h1[nb] = bias_init(b1, nb, h, DW);
        layer128x4<true>(pw1, lane, hid, h1);
        const int nblk = (k + 31) >> 5;
        for (int ob = 0; ob < nblk; ++ob) {
            floatx16 o = bias_init(b2, ob, h, k);
            layer128<true>(pw2, ob, lane, h1, o);
            if (pvalid) {
                if ((k & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 32 * ob + 8 * q + 4 * h;
                        if (row < k) {
                            const bool two = head == 4 && a.in_rest != nullptr;
                            const float4 iv = fdgs_load_sh4(two ? a.in_rest + (size_t)p * 45 - 3 : in + (size_t)p * k,
                                                            two ? in + (size_t)p * 3 : nullptr, row);
                            *reinterpret_cast<float4*>(out + (size_t)p * k + row) =
                                make_float4(iv.x + o[4 * q], iv.y + o[4 * q + 1], iv.z + o[4 * q + 2], iv.w + o[4 * q + 3]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {       // k <= 3: rows 0..2 live in registers 0..2 of half 0
                        const int row = drow(r, h);
                        if (row < k) out[(size_t)p * k + row] = in[(size_t)p * k + row] + o[r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Wide layers (used by the backward): one wave owns 64 points (two column
// blocks) and the whole 512-register file.  Every weight fragment fetched from
// L2 feeds two MFMAs per accumulator row block, and the next K group's
// fragments are requested one full group (32 MFMAs) ahead.
// ---------------------------------------------------------------------------
#define NCB 2
template <bool RELU>
__device__ __forceinline__ void wide_layer128x4(const float* packed, int lane, const floatx16 (*x)[4],
                                                floatx16 (*acc)[4]) {
    float4 cur[4], nxt[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) cur[nb] = frag_load(packed, nb, 16, 0, lane);
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        if (T + 1 < 16) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) nxt[nb] = frag_load(packed, nb, 16, T + 1, lane);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float xv = x[cb][T >> 2][4 * (T & 3) + q];
                if (RELU) xv = fmaxf(xv, 0.0f);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const float av = q == 0 ? cur[nb].x : q == 1 ? cur[nb].y : q == 2 ? cur[nb].z : cur[nb].w;
                    acc[cb][nb] = MFMA(av, xv, acc[cb][nb]);
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) cur[nb] = nxt[nb];
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distance at one K group (bounds live registers)
    }
}

// same, with the B operand (relu of a [128 features][64 points] tile) read from this wave's LDS tile
__device__ __forceinline__ void wide_layer128x4_ldsb(const float* packed, int lane, const float* tile,
                                                     floatx16 (*acc)[4]) {
    const int c = lane & 31, h = lane >> 5;
    float4 cur[4], nxt[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) cur[nb] = frag_load(packed, nb, 16, 0, lane);
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        if (T + 1 < 16) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) nxt[nb] = frag_load(packed, nb, 16, T + 1, lane);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float xv = fmaxf(tile[(8 * T + 4 * h + q) * 64 + 32 * cb + c], 0.0f);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const float av = q == 0 ? cur[nb].x : q == 1 ? cur[nb].y : q == 2 ? cur[nb].z : cur[nb].w;
                    acc[cb][nb] = MFMA(av, xv, acc[cb][nb]);
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) cur[nb] = nxt[nb];
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distance at one K group (bounds live registers)
    }
}

template <bool RELU>
__device__ __forceinline__ void wide_layer128(const float* packed, int ib, int lane, const floatx16 (*x)[4],
                                              floatx16* acc) {
    float4 cur = frag_load(packed, ib, 16, 0, lane), nxt = cur;
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        if (T + 1 < 16) nxt = frag_load(packed, ib, 16, T + 1, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float av = q == 0 ? cur.x : q == 1 ? cur.y : q == 2 ? cur.z : cur.w;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float xv = x[cb][T >> 2][4 * (T & 3) + q];
                if (RELU) xv = fmaxf(xv, 0.0f);
                acc[cb] = MFMA(av, xv, acc[cb]);
            }
        }
        cur = nxt;
    }
}

#include "deform_split.h"

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
static bool params_ok(const fdgs_deform_params* p) {
    if (!p || p->num_levels != 2 || !p->aabb || !p->w0 || !p->b0) return false;   // built for multires of length 2
    for (int l = 0; l < 2; ++l)
        for (int ci = 0; ci < 6; ++ci)
            if (!p->planes[l][ci]) return false;
    for (int l = 0; l < 2; ++l)
        for (int k = 0; k < 4; ++k)
            if (p->res[l][k] < 2) return false;
    for (int h = 0; h < 5; ++h)
        if (!p->w1[h] || !p->b1[h] || !p->w2[h] || !p->b2[h]) return false;
    if (p->matmul != FDGS_MATMUL_F32 && p->matmul != FDGS_MATMUL_SPLIT_BF16) return false;
    return true;
}


static size_t plane_texels(const fdgs_deform_params* p, int l, int ci) {
    return (size_t)p->res[l][plane_a(ci)] * (size_t)p->res[l][plane_b(ci)];
}
static size_t total_plane_floats(const fdgs_deform_params* p) {
    size_t t = 0;
    for (int l = 0; l < p->num_levels; ++l)
        for (int ci = 0; ci < 6; ++ci) t += plane_texels(p, l, ci) * DF;
    return t;
}

static const int h_head_dim[5] = {3, 3, 4, 1, 48};

static size_t fwd_packed_floats(int matmul) {
    if (matmul == FDGS_MATMUL_SPLIT_BF16) {
        size_t t = sfrag_vec4(DW, 32);
        for (int h = 0; h < 5; ++h) t += sfrag_vec4(DW, DW) + sfrag_vec4(h_head_dim[h], DW);
        return 4 * t;
    }
    size_t t = frag_floats(DW, 32);
    for (int h = 0; h < 5; ++h) t += frag_floats(DW, DW) + frag_floats(h_head_dim[h], DW);
    return t;
}
static size_t bwd_packed_floats(int matmul) {
    if (matmul == FDGS_MATMUL_SPLIT_BF16)     // W0^T, five W1^T, five W1 (natural K order), W2^T of the shs head
        return 4 * (size_t)(sfrag_vec4(32, DW) + 10 * sfrag_vec4(DW, DW) + sfrag_vec4(DW, 48));
    size_t t = frag_floats(32, DW);
    for (int h = 0; h < 5; ++h) t += 2 * frag_floats(DW, DW) + frag_floats(DW, h_head_dim[h]);
    return t;
}

static size_t fwd_split_bias_floats();
static size_t fwd_split_feat_floats(int n);

extern "C" size_t fdgs_deform_fwd_workspace_bytes(const fdgs_deform_params* p, int32_t num_points) {
    if (!p || p->num_levels < 1 || p->num_levels > FDGS_DEFORM_MAX_LEVELS || num_points < 0) return 0;
    size_t floats = total_plane_floats(p) + fwd_packed_floats(p->matmul);
    if (p->matmul == FDGS_MATMUL_SPLIT_BF16) floats += fwd_split_bias_floats() + fwd_split_feat_floats(num_points);
    return floats * sizeof(float);
}

static int launch_pack(const FragTable& t, hipStream_t s) {
    hipLaunchKernelGGL(pack_fragments_kernel, dim3(16, t.n), dim3(256), 0, s, t);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

// fills ps with the channel-last plane addresses inside ws and (unless the caller
// says ws already holds them) rebuilds the copies with one launch
struct SFragTable;
static void add_plane_copy(SFragTable* into, const float* src, float* dst, int texels);
// With `into`, the copies are left as twelve entries of that table (the split path packs its weights with the
// same launch) and nothing is launched here.
static int make_plane_set(const fdgs_deform_params* p, float* ws, PlaneSet* ps, bool rebuild, hipStream_t s,
                          SFragTable* into = nullptr) {
    PlaneTable t;
    size_t off = 0;
    int max_hw = 0;
    for (int l = 0; l < 2; ++l) {
        for (int k = 0; k < 4; ++k) ps->res[l][k] = p->res[l][k];
        for (int ci = 0; ci < 6; ++ci) {
            const int hw = (int)plane_texels(p, l, ci);
            t.src[6 * l + ci] = p->planes[l][ci];
            t.dst[6 * l + ci] = ws + off;
            t.hw[6 * l + ci] = hw;
            ps->pl[l][ci] = ws + off;
            if (rebuild && into) add_plane_copy(into, p->planes[l][ci], ws + off, hw);
            off += (size_t)hw * DF;
            if (hw > max_hw) max_hw = hw;
        }
    }
    if (rebuild && !into) {
        hipLaunchKernelGGL(planes_to_channel_last_kernel, dim3((max_hw + 255) / 256, 12), dim3(256), 0, s, t);
        FDGS_CHECK_LAUNCH();
    }
    return FDGS_OK;
}

static int launch_pack_split(const SFragTable& t, hipStream_t s, int blocks_per_entry = 8) {
    hipLaunchKernelGGL(pack_split_kernel, dim3(blocks_per_entry, t.n), dim3(256), 0, s, t);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
static void add_plane_copy(SFragTable* into, const float* src, float* dst, int texels) {
    static_assert(DF == 16, "the plane entries of pack_split_kernel move 16 channels");
    into->d[into->n++] = SFragDesc{src, reinterpret_cast<u32x4*>(dst), texels, 0, 0, FRAG_PLANE};
}

// split-mode forward workspace, in floats after the plane copies:
//   weight pieces | padded biases (b0 128, b1 5x128, b2 5x64) | feature pieces (ceil(N/32) groups x 1536 floats)
static size_t fwd_split_bias_floats() { return DW + 5 * DW + 5 * 64; }
static size_t fwd_split_feat_floats(int n) { return (size_t)((n + 31) / 32) * 2 * 3 * 64 * 4; }

static int deform_fwd_split(const fdgs_deform_fwd_args* a, hipStream_t s) {
    const bool saving = a->save_mask != nullptr;
    if (saving != (a->save_act != nullptr) || saving != (a->save_hidden != nullptr || a->save_yhid != nullptr))
        return FDGS_EINVAL;                         // mask, act and (hidden or yhid) together, or none of them
    DeformFwdSK k;
    HexFeatSK g;
    k.n = g.n = a->num_points;
    float* ws = (float*)a->workspace;
    SFragTable t;
    t.n = 0;
    int rc = make_plane_set(&a->p, ws, &g.ps, true, s, &t);    // twelve entries: the channel-last plane copies
    if (rc != FDGS_OK) return rc;
    u32x4* pk = reinterpret_cast<u32x4*>(ws + total_plane_floats(&a->p));
    float* pb = reinterpret_cast<float*>(pk) + fwd_packed_floats(FDGS_MATMUL_SPLIT_BF16);
    u32x4* feat = reinterpret_cast<u32x4*>(pb + fwd_split_bias_floats());
    {
        t.d[t.n++] = SFragDesc{a->p.w0, pk, DW, 32, 32, FRAG_FEATMAP};
        k.pw0 = pk; pk += sfrag_vec4(DW, 32);
        for (int h = 0; h < 5; ++h) {
            t.d[t.n++] = SFragDesc{a->p.w1[h], pk, DW, DW, DW, 0};
            k.pw1[h] = pk; pk += sfrag_vec4(DW, DW);
            t.d[t.n++] = SFragDesc{a->p.w2[h], pk, h_head_dim[h], DW, DW, 0};
            k.pw2[h] = pk; pk += sfrag_vec4(h_head_dim[h], DW);
        }
        // the padded biases ride in the same launch: 12 + 11 + 11 descriptors
        static_assert(FRAG_MAX >= 34, "plane copies, weights and biases of the forward in one table");
        t.d[t.n++] = SFragDesc{a->p.b0, reinterpret_cast<u32x4*>(pb), DW, DW, 0, FRAG_BIAS};
        k.pb0 = pb; pb += DW;
        for (int h = 0; h < 5; ++h) {
            t.d[t.n++] = SFragDesc{a->p.b1[h], reinterpret_cast<u32x4*>(pb), DW, DW, 0, FRAG_BIAS};
            k.pb1[h] = pb; pb += DW;
            t.d[t.n++] = SFragDesc{a->p.b2[h], reinterpret_cast<u32x4*>(pb), h_head_dim[h], 64, 0, FRAG_BIAS};
            k.pb2[h] = pb; pb += 64;
        }
        rc = launch_pack_split(t, s, 64);     // 64 workgroups per entry: the largest plane has tens of thousands of texels
        if (rc != FDGS_OK) return rc;
    }
    g.aabb = a->p.aabb; g.xyz = a->xyz; g.time = a->time; g.time_scalar = a->time_scalar;
    g.feat_split = feat; g.feat_f32 = nullptr; g.feat_y = reinterpret_cast<u32x4*>(a->save_yfeat);
    k.feat_split = feat;
    k.in[0] = a->xyz; k.in[1] = a->scales; k.in[2] = a->rotations; k.in[3] = a->opacity; k.in[4] = a->shs;
    k.in_rest = a->shs_rest;
    k.out[0] = a->out_xyz; k.out[1] = a->out_scales; k.out[2] = a->out_rotations; k.out[3] = a->out_opacity;
    k.out[4] = a->out_shs;
    for (int h = 0; h < 5; ++h) { k.w2[h] = a->p.w2[h]; k.b2[h] = a->p.b2[h]; }
    k.save_hidden = a->save_hidden;
    k.save_act = reinterpret_cast<u32x4*>(a->save_act);
    k.save_mask = reinterpret_cast<uint2*>(a->save_mask);
    k.save_yhid = reinterpret_cast<u32x4*>(a->save_yhid);
    hipLaunchKernelGGL(hexplane_features_split_kernel, dim3((a->num_points + 127) / 128), dim3(256), 0, s, g);
    FDGS_CHECK_LAUNCH();
    const int blocks = (a->num_points + 32 * SCB - 1) / (32 * SCB);
    if (saving) hipLaunchKernelGGL(deform_fwd_split_kernel<true>, dim3(blocks), dim3(64), 0, s, k);
    else hipLaunchKernelGGL(deform_fwd_split_kernel<false>, dim3(blocks), dim3(64), 0, s, k);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

static bool planes_ok(const fdgs_deform_params* p) {
    if (!p || p->num_levels != 2 || !p->aabb) return false;
    for (int l = 0; l < 2; ++l) {
        for (int ci = 0; ci < 6; ++ci)
            if (!p->planes[l][ci]) return false;
        for (int k = 0; k < 4; ++k)
            if (p->res[l][k] < 2) return false;
    }
    return true;
}

extern "C" size_t fdgs_hexplane_features_workspace_bytes(const fdgs_deform_params* p, int32_t num_points) {
    if (!p || p->num_levels < 1 || p->num_levels > FDGS_DEFORM_MAX_LEVELS || num_points < 0) return 0;
    return (total_plane_floats(p) + fwd_split_feat_floats(num_points)) * sizeof(float);
}

extern "C" int fdgs_hexplane_features(const fdgs_hexplane_features_args* a, void* stream) {
    if (!a || a->num_points < 0 || !planes_ok(&a->p)) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->xyz || !a->features || !a->workspace) return FDGS_EINVAL;
    if (a->workspace_bytes < fdgs_hexplane_features_workspace_bytes(&a->p, a->num_points)) return FDGS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    HexFeatSK g;
    g.n = a->num_points;
    float* ws = (float*)a->workspace;
    int rc = make_plane_set(&a->p, ws, &g.ps, true, s);
    if (rc != FDGS_OK) return rc;
    g.aabb = a->p.aabb; g.xyz = a->xyz; g.time = a->time; g.time_scalar = a->time_scalar;
    g.feat_split = reinterpret_cast<u32x4*>(ws + total_plane_floats(&a->p));
    g.feat_f32 = a->features; g.feat_y = nullptr;
    hipLaunchKernelGGL(hexplane_features_split_kernel, dim3((a->num_points + 127) / 128), dim3(256), 0, s, g);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

extern "C" int fdgs_deform_fwd(const fdgs_deform_fwd_args* a, void* stream) {
    if (!a || a->num_points < 0 || !params_ok(&a->p)) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->xyz || !a->scales || !a->rotations || !a->opacity || !a->shs || !a->out_xyz || !a->out_scales ||
        !a->out_rotations || !a->out_opacity || !a->out_shs || !a->workspace)
        return FDGS_EINVAL;
    if (a->workspace_bytes < fdgs_deform_fwd_workspace_bytes(&a->p, a->num_points)) return FDGS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (a->p.matmul == FDGS_MATMUL_SPLIT_BF16) return deform_fwd_split(a, s);
    DeformFwdK k;
    k.n = a->num_points;
    float* ws = (float*)a->workspace;
    int rc = make_plane_set(&a->p, ws, &k.ps, true, s);
    if (rc != FDGS_OK) return rc;
    {   // weight fragments behind the plane copies
        float* pk = ws + total_plane_floats(&a->p);
        FragTable t;
        t.n = 0;
        t.d[t.n++] = FragDesc{a->p.w0, pk, DW, 32, 32, FRAG_FEATMAP};
        k.pw0 = pk; pk += frag_floats(DW, 32);
        for (int h = 0; h < 5; ++h) {
            t.d[t.n++] = FragDesc{a->p.w1[h], pk, DW, DW, DW, 0};
            k.pw1[h] = pk; pk += frag_floats(DW, DW);
            t.d[t.n++] = FragDesc{a->p.w2[h], pk, h_head_dim[h], DW, DW, 0};
            k.pw2[h] = pk; pk += frag_floats(h_head_dim[h], DW);
        }
        rc = launch_pack(t, s);
        if (rc != FDGS_OK) return rc;
    }
    k.aabb = a->p.aabb; k.b0 = a->p.b0;
    for (int h = 0; h < 5; ++h) { k.b1[h] = a->p.b1[h]; k.b2[h] = a->p.b2[h]; }
    k.xyz = a->xyz;
    k.in[0] = a->xyz; k.in[1] = a->scales; k.in[2] = a->rotations; k.in[3] = a->opacity; k.in[4] = a->shs;
    k.in_rest = a->shs_rest;
    k.out[0] = a->out_xyz; k.out[1] = a->out_scales; k.out[2] = a->out_rotations; k.out[3] = a->out_opacity;
    k.out[4] = a->out_shs;
    k.time = a->time; k.time_scalar = a->time_scalar;
    k.save_hidden = a->save_hidden;
    const int blocks = (a->num_points + TILE_PTS - 1) / TILE_PTS;
    hipLaunchKernelGGL(deform_fwd_kernel, dim3(blocks), dim3(256), 0, s, k);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

// ===========================================================================
// Backward
// ===========================================================================
// Pass 1 (per 128-point tile, same orientation as the forward): recompute the
// activations, chain the output cotangents back through the five heads and the
// first Linear, scatter the plane gradients, and leave in the workspace the
// per-point operands of the weight-gradient products:
//   a1  [N,640]  relu(h1) of every head        dh1 [N,640]  dL/dh1 of every head
//   dhid[N,128]  dL/dhidden                    feat[N,32]   HexPlane features
// Pass 2: weight gradients as X^T Y products with the points as the MFMA K
// dimension, split over point chunks, accumulated with float atomics shaped as
// two 128-byte row segments per wave instruction.
struct DeformBwdK {
    int n;
    const float* pw1[5]; const float* b1[5];    // packed W1 (recompute of h1)
    const float* pw2t[5];                       // packed W2^T : A[kk][o]
    const float* pw1t[5];                       // packed W1^T : A[kk][n]
    const float* pw0t;                          // packed W0^T with permuted rows
    const float* hidden;
    const float* g[5];                   // output cotangents
    float* ws_a1; float* ws_dh1; float* ws_dhid; float* ws_df;
};

// Pass 1: one wave per workgroup, 64 points per wave (two column blocks); hidden^T of the wave's points
// lives in LDS (B operand of the h1 recompute), dacc and the h1/dh1 tile in registers.
__global__ void __launch_bounds__(64, 1)
deform_bwd_points_kernel(DeformBwdK a) {
    __shared__ float tile[DW * 64];          // hidden^T of this wave's 64 points: [feature][point], 32 KiB
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;
    int p[NCB]; bool pvalid[NCB];
    floatx16 dacc[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        p[cb] = blockIdx.x * (32 * NCB) + 32 * cb + c;
        pvalid[cb] = p[cb] < a.n;
        const int pc = pvalid[cb] ? p[cb] : a.n - 1;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 32 * nb + 8 * q + 4 * h;
                const float4 v = *reinterpret_cast<const float4*>(a.hidden + (size_t)pc * DW + f);
                tile[(f + 0) * 64 + 32 * cb + c] = v.x; tile[(f + 1) * 64 + 32 * cb + c] = v.y;
                tile[(f + 2) * 64 + 32 * cb + c] = v.z; tile[(f + 3) * 64 + 32 * cb + c] = v.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[cb][nb][r] = 0.0f;
        }
    }
    for (int head = 0; head < 5; ++head) {
        const int k = head_dim(head);
        const float* pw1 = sel5(a.pw1, head); const float* b1 = sel5(a.b1, head);
        const float* pw2t = sel5(a.pw2t, head); const float* pw1t = sel5(a.pw1t, head);
        const float* g = sel5(a.g, head);
        floatx16 t[NCB][4];      // h1, then dh1
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) t[cb][nb] = bias_init(b1, nb, h, DW);
        wide_layer128x4_ldsb(pw1, lane, tile, t);
        unsigned long long pos_mask[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            pos_mask[cb] = 0ull;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t[cb][nb][r] > 0.0f) pos_mask[cb] |= (1ull << (16 * nb + r));
                if (pvalid[cb]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(a.ws_a1 + (size_t)p[cb] * 640 + head * DW + 32 * nb + 8 * q + 4 * h) =
                            make_float4(fmaxf(t[cb][nb][4 * q], 0.f), fmaxf(t[cb][nb][4 * q + 1], 0.f),
                                        fmaxf(t[cb][nb][4 * q + 2], 0.f), fmaxf(t[cb][nb][4 * q + 3], 0.f));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) t[cb][nb][r] = 0.0f;
            }
        }
        const int G2 = (k + 7) >> 3;
        for (int T = 0; T < G2; ++T) {
            float gv[NCB][4];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 8 * T + 4 * h + q;
                    gv[cb][q] = (o < k && pvalid[cb]) ? g[(size_t)p[cb] * k + o] : 0.0f;
                }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const float4 a4 = frag_load(pw2t, nb, G2, T, lane);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    t[cb][nb] = MFMA(a4.x, gv[cb][0], t[cb][nb]);
                    t[cb][nb] = MFMA(a4.y, gv[cb][1], t[cb][nb]);
                    t[cb][nb] = MFMA(a4.z, gv[cb][2], t[cb][nb]);
                    t[cb][nb] = MFMA(a4.w, gv[cb][3], t[cb][nb]);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!((pos_mask[cb] >> (16 * nb + r)) & 1ull)) t[cb][nb][r] = 0.0f;
                if (pvalid[cb]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(a.ws_dh1 + (size_t)p[cb] * 640 + head * DW + 32 * nb + 8 * q + 4 * h) =
                            make_float4(t[cb][nb][4 * q], t[cb][nb][4 * q + 1], t[cb][nb][4 * q + 2], t[cb][nb][4 * q + 3]);
                }
            }
        wide_layer128x4<false>(pw1t, lane, t, dacc);
    }
    floatx16 df[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (!(tile[(32 * nb + drow(r, h)) * 64 + 32 * cb + c] > 0.0f)) dacc[cb][nb][r] = 0.0f;
            if (pvalid[cb]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.ws_dhid + (size_t)p[cb] * DW + 32 * nb + 8 * q + 4 * h) =
                        make_float4(dacc[cb][nb][4 * q], dacc[cb][nb][4 * q + 1], dacc[cb][nb][4 * q + 2],
                                    dacc[cb][nb][4 * q + 3]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) df[cb][r] = 0.0f;
    }
    wide_layer128<false>(a.pw0t, 0, lane, dacc, df);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
        if (pvalid[cb]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(a.ws_df + (size_t)p[cb] * 32 + 16 * h + 4 * q) =
                    make_float4(df[cb][4 * q], df[cb][4 * q + 1], df[cb][4 * q + 2], df[cb][4 * q + 3]);
        }
}

// ---------------------------------------------------------------------------
// Pass 1b: HexPlane backward.  Item = (point, level); 16 lanes per item, lane =
// channel, so that every texel access of an item is one contiguous 64-byte
// segment: four items per wave instruction for loads and for the float atomics
// of the scatter.  Planes with a time axis are touched in two rows only when
// the whole call shares one time stamp; those rows are accumulated in LDS and
// flushed once per workgroup.
// ---------------------------------------------------------------------------
struct HexBwdK {
    int n;
    PlaneSet ps;
    float* gpl[2][6];
    const float* aabb;
    const float* xyz; const float* time; float time_scalar;
    const float* ws_df;                  // [N,32] dL/dfeature
    float* ws_feat;                      // [N,32] features, or NULL (for the first Linear's weight gradient)
    const float* g_xyz; float* d_xyz;
    int lds_time;                        // 1: one time stamp for all points, time-plane rows live in LDS
};

template <int CTRL>
__device__ __forceinline__ float row_shr_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(t);
}
// sum over the 16 lanes of a DPP row, valid in lane 15 of the row
__device__ __forceinline__ float row_sum15(float v) {
    v = row_shr_add<0x111>(v); v = row_shr_add<0x112>(v); v = row_shr_add<0x114>(v); v = row_shr_add<0x118>(v);
    return v;
}

#define HEX_LDS_FLOATS (3 * 2 * 64 * DF + 3 * 2 * 128 * DF)   // upper bound used for static sizing below

__global__ void __launch_bounds__(256)
hexplane_bwd_kernel(HexBwdK a, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int ch = lane & 15, sub = lane >> 4;
    if (a.lds_time) {
        for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) lds[i] = 0.0f;
        __syncthreads();
    }
    // LDS offsets of the three time planes (ci = 2, 4, 5) of each level: [2 rows][res_a][16]
    int lds_base[2][3];
    {
        int off = 0;
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int tp = 0; tp < 3; ++tp) {
                lds_base[l][tp] = off;
                off += 2 * a.ps.res[l][tp] * DF;      // spatial axis of time plane tp is axis tp (x,y,z)
            }
    }
    const int n_items = 2 * a.n;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave_global * 4; base < n_items; base += n_waves * 4) {
        const int item = base + sub;
        const bool valid = item < n_items;
        const int p = valid ? (item >> 1) : (a.n - 1);
        const int h = item & 1;
        float co[4], sc3[3];
        point_coords(a.xyz, a.aabb, a.time, a.time_scalar, p, co, sc3);
        float v[6];
        Bilinear bl[6];
        float t00[6], t01[6], t10[6], t11[6];
        float feat = 1.0f;
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
            const int ax = plane_a(ci), bx = plane_b(ci);
            const int Wd = h ? a.ps.res[1][ax] : a.ps.res[0][ax];
            const int Hd = h ? a.ps.res[1][bx] : a.ps.res[0][bx];
            const float* __restrict__ pl = h ? a.ps.pl[1][ci] : a.ps.pl[0][ci];
            bl[ci] = make_bilinear(co[ax], co[bx], Wd, Hd);
            t00[ci] = pl[(size_t)bl[ci].o00 * DF + ch]; t01[ci] = pl[(size_t)bl[ci].o01 * DF + ch];
            t10[ci] = pl[(size_t)bl[ci].o10 * DF + ch]; t11[ci] = pl[(size_t)bl[ci].o11 * DF + ch];
            v[ci] = t00[ci] * bl[ci].w00 + t01[ci] * bl[ci].w01 + t10[ci] * bl[ci].w10 + t11[ci] * bl[ci].w11;
            feat *= v[ci];
        }
        float dfv = 0.0f;
        if (valid) {
            if (a.ws_feat) a.ws_feat[(size_t)p * 32 + 16 * h + ch] = feat;
            dfv = a.ws_df[(size_t)p * 32 + 16 * h + ch];
        }
        float dco[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
            const int ax = plane_a(ci), bx = plane_b(ci);
            const int Wd = h ? a.ps.res[1][ax] : a.ps.res[0][ax];
            const int Hd = h ? a.ps.res[1][bx] : a.ps.res[0][bx];
            float g = dfv;
#pragma unroll
            for (int cj = 0; cj < 6; ++cj)
                if (cj != ci) g *= v[cj];
            const Bilinear& b = bl[ci];
            const float gx = g * ((t01[ci] - t00[ci]) * (1.0f - b.fy) + (t11[ci] - t10[ci]) * b.fy);
            const float gy = g * ((t10[ci] - t00[ci]) * (1.0f - b.fx) + (t11[ci] - t01[ci]) * b.fx);
            // d(texel coordinate)/d(normalised coordinate) = 0.5*(size-1) where strictly inside
            if (b.in_x) dco[ax] += gx * (0.5f * (float)(Wd - 1));
            if (bx < 3 && b.in_y) dco[bx] += gy * (0.5f * (float)(Hd - 1));
            if (valid) {
                if (bx == 3 && a.lds_time) {
                    // time plane: rows y0 / y1 are shared by every point of the call
                    const int tp = ax;
                    float* L = lds + (h ? lds_base[1][tp] : lds_base[0][tp]);
                    const int y0 = b.o00 / Wd, x0 = b.o00 - y0 * Wd, x1 = b.o01 - y0 * Wd;
                    const int r1 = (b.o10 != b.o00) ? 1 : 0;
                    atomicAdd(L + (0 * Wd + x0) * DF + ch, g * b.w00);
                    atomicAdd(L + (0 * Wd + x1) * DF + ch, g * b.w01);
                    atomicAdd(L + (r1 * Wd + x0) * DF + ch, g * b.w10);
                    atomicAdd(L + (r1 * Wd + x1) * DF + ch, g * b.w11);
                } else {
                    float* __restrict__ gp = h ? a.gpl[1][ci] : a.gpl[0][ci];
                    atomicAdd(gp + (size_t)b.o00 * DF + ch, g * b.w00);
                    atomicAdd(gp + (size_t)b.o01 * DF + ch, g * b.w01);
                    atomicAdd(gp + (size_t)b.o10 * DF + ch, g * b.w10);
                    atomicAdd(gp + (size_t)b.o11 * DF + ch, g * b.w11);
                }
            }
        }
        // sum over channels (DPP row), then over the two levels of the point (rows 2j and 2j+1)
        float dx = row_sum15(dco[0] * sc3[0]), dy = row_sum15(dco[1] * sc3[1]), dz = row_sum15(dco[2] * sc3[2]);
        dx += __shfl_xor(dx, 16); dy += __shfl_xor(dy, 16); dz += __shfl_xor(dz, 16);
        if (valid && h == 0 && ch == 15) {
            a.d_xyz[3 * (size_t)p + 0] = a.g_xyz[3 * (size_t)p + 0] + dx;
            a.d_xyz[3 * (size_t)p + 1] = a.g_xyz[3 * (size_t)p + 1] + dy;
            a.d_xyz[3 * (size_t)p + 2] = a.g_xyz[3 * (size_t)p + 2] + dz;
        }
    }
    if (a.lds_time) {
        __syncthreads();
        // flush: LDS row 0 -> plane row y0, LDS row 1 -> plane row y1 of the shared time stamp
        float x; bool in;
        unnorm_clip(a.time_scalar, a.ps.res[0][3], x, in);
        const int y0 = (int)floorf(x), y1 = min(y0 + 1, a.ps.res[0][3] - 1);
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int tp = 0; tp < 3; ++tp) {
                const int ci = tp == 0 ? 2 : tp == 1 ? 4 : 5;
                const int Wd = a.ps.res[l][tp];
                float* gp = a.gpl[l][ci];
                const float* L = lds + lds_base[l][tp];
                const int row_floats = Wd * DF;
                for (int i = threadIdx.x; i < row_floats; i += blockDim.x) {
                    const float v0 = L[i], v1 = L[row_floats + i];
                    if (v0 != 0.0f) atomicAdd(gp + (size_t)y0 * row_floats + i, v0);
                    if (v1 != 0.0f) atomicAdd(gp + (size_t)y1 * row_floats + i, v1);
                }
            }
    }
}

#include "hexplane_bricks.h"

// ---------------------------------------------------------------------------
// Pass 2: C[m][n] += sum_p X[p][m] * Y[p][n]   (+ db[m] += sum_p X[p][m])
// ---------------------------------------------------------------------------
struct WGradDesc {
    const float* X; int ldx; int M;          // X[p*ldx + m], m < M
    const float* Y; int ldy; int NN;         // Y[p*ldy + n], n < NN (NN = 32 or 128)
    int relu_y;
    float* C; int ldc;                       // C[m*ldc + n]
    float* db;                               // may be NULL
};
#define WG_MAX 11
struct WGradK {
    int n; int chunk;
    WGradDesc d[WG_MAX];
};

__global__ void __launch_bounds__(256)
deform_wgrad_kernel(WGradK a) {
    WGradDesc d = a.d[0];
#pragma unroll
    for (int i = 1; i < WG_MAX; ++i)
        if ((int)blockIdx.y == i) d = a.d[i];
    if (!d.C) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int m0 = 32 * wave;
    if (m0 >= d.M) return;
    const int p_begin = blockIdx.x * a.chunk;
    const int p_end = min(p_begin + a.chunk, a.n);
    const bool mv = (m0 + c) < d.M;
    const int nblk = d.NN >> 5;
    floatx16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    float bsum = 0.0f;
    // 8 K-steps (16 points) per iteration: all loads are issued before the MFMAs that consume them
    for (int pb = p_begin; pb < p_end; pb += 16) {
        float xv[8], yv[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = pb + 2 * u + hh;
            const bool pv = p < p_end;
            xv[u] = (pv && mv) ? d.X[(size_t)p * d.ldx + m0 + c] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                yv[u][nb] = (pv && nb < nblk) ? d.Y[(size_t)p * d.ldy + 32 * nb + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bsum += xv[u];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (nb < nblk) {
                    float y = yv[u][nb];
                    if (d.relu_y) y = fmaxf(y, 0.0f);
                    acc[nb] = MFMA(xv[u], y, acc[nb]);
                }
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb < nblk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + drow(r, hh);
                if (m < d.M) atomicAdd(d.C + (size_t)m * d.ldc + 32 * nb + c, acc[nb][r]);
            }
        }
    }
    if (d.db && mv) atomicAdd(d.db + m0 + c, bsum);
}

// points per backward chunk: bounds the a1 / dh1 workspace (5 KiB per point) for very large point counts;
// smaller chunks measured slower (the chip drains between the launches)
static int bwd_chunk_points(int n) {
    int c = 524288;
    if (c > n) c = n;
    return c < 1 ? 1 : c;
}

// pre-split operands of the weight-gradient products: relu(hidden), 4 column blocks
static size_t bwd_ysplit_floats(int matmul, int n) {
    return matmul == FDGS_MATMUL_SPLIT_BF16 ? (size_t)192 * (bwd_chunk_points(n) + 16) : 0;
}
// plane features pre-split as the B operand of the dW0 product (one column block, 192 bytes per point)
#define DW0_SLABS 16
#define W2_SLABS 16
#define W2_SLAB_FLOATS (59 * DW + 5 * 64)                 // the five dW2 (59 rows) and db2 (5 x 64, padded)
// + DW0_SLABS private copies of dW0 (128 x 32) and db0 (128), + W2_SLABS copies of dW2 / db2
static size_t bwd_yfeat_floats(int matmul, int n) {
    return matmul == FDGS_MATMUL_SPLIT_BF16
               ? (size_t)48 * ((size_t)n + 16) + (size_t)DW0_SLABS * (DW * 32 + DW) + (size_t)W2_SLABS * W2_SLAB_FLOATS +
                     (size_t)(DW * DW + DW)                  // Q and T of the one-output head (deform_bwd_finish_kernel)
               : 0;
}
// brick binning of the HexPlane backward, in 4-byte words: perm [n], brick_of [n] (16 bit, rounded up),
// counts / cursor [BRICK_MAX] each, work count (16 words), work items [3 x max_work]
static int brick_max_work(int n) { return BRICK_MAX + n / BRICK_CH + 1; }
static size_t bwd_brick_words(int n) {
    return (size_t)n + ((size_t)n + 1) / 2 + 2 * BRICK_MAX + 16 + (size_t)3 * brick_max_work(n);
}

extern "C" size_t fdgs_deform_bwd_workspace_bytes(const fdgs_deform_params* p, int32_t num_points) {
    if (!p || num_points < 0 || p->num_levels < 1 || p->num_levels > FDGS_DEFORM_MAX_LEVELS) return 0;
    const size_t pf = total_plane_floats(p);
    const size_t per_point = DW + 32 + 32;
    return (2 * pf + bwd_packed_floats(p->matmul) + per_point * (size_t)num_points +
            (size_t)1280 * bwd_chunk_points(num_points) + bwd_ysplit_floats(p->matmul, num_points) +
            bwd_yfeat_floats(p->matmul, num_points) + bwd_brick_words(num_points)) * sizeof(float);
}

// The HexPlane scatter (atomics-bound) and the weight-gradient products (matrix-core / memory-bound) only
// depend on pass 1, not on each other: when the caller lends a second stream and two events
// (fdgs_deform_bwd_args.side_stream / fork_event / join_event) they are issued on two streams (fork after
// pass 1, join before returning to the caller's stream order; a first fork before pass 1 starts what needs none of
// its results, fork_event being recorded a second time after it).  The library itself owns no stream and no event.
struct SideStream { hipStream_t stream; hipEvent_t fork, join; };

extern "C" int fdgs_deform_bwd(const fdgs_deform_bwd_args* a, void* stream) {
    if (!a || a->num_points < 0 || !params_ok(&a->p)) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->xyz || (!a->hidden && !(a->yhid && a->p.matmul == FDGS_MATMUL_SPLIT_BF16)) || !a->g_xyz || !a->g_scales || !a->g_rotations || !a->g_opacity || !a->g_shs ||
        !a->d_xyz || !a->workspace)
        return FDGS_EINVAL;
    if (a->workspace_bytes < fdgs_deform_bwd_workspace_bytes(&a->p, a->num_points)) return FDGS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int n = a->num_points;
    const size_t pf = total_plane_floats(&a->p);
    float* ws = (float*)a->workspace;
    float* planes_cl = ws;
    float* grads_cl = ws + pf;
    float* ws_packed = grads_cl + pf;
    float* ws_a1 = ws_packed + bwd_packed_floats(a->p.matmul);
    const size_t chunk_cap = (size_t)bwd_chunk_points(n);
    float* ws_dh1 = ws_a1 + (size_t)640 * chunk_cap;
    float* ws_dhid = ws_dh1 + (size_t)640 * chunk_cap;
    float* ws_feat = ws_dhid + (size_t)DW * n;

    float* ws_df = ws_feat + (size_t)32 * n;
    float* ws_ysplit = ws_df + (size_t)32 * n;      // split mode only (sized by fdgs_deform_bwd_workspace_bytes)
    float* ws_yfeat = ws_ysplit + bwd_ysplit_floats(a->p.matmul, n);
    int* ws_brick = reinterpret_cast<int*>(ws_yfeat + bwd_yfeat_floats(a->p.matmul, n));

    HexBwdK hk;
    hk.n = n;
    const bool reuse = a->planes_channel_last != nullptr;
    if (reuse) planes_cl = const_cast<float*>(a->planes_channel_last);
    int rc = make_plane_set(&a->p, planes_cl, &hk.ps, !reuse, s);
    if (rc != FDGS_OK) return rc;
    {   // (grads_cl is cleared where it is first used: at the head of the HexPlane tail, on that tail's stream)
        size_t off = 0;
        for (int l = 0; l < 2; ++l)
            for (int ci = 0; ci < 6; ++ci) {
                hk.gpl[l][ci] = grads_cl + off;
                off += plane_texels(&a->p, l, ci) * DF;
            }
    }
    const bool split = a->p.matmul == FDGS_MATMUL_SPLIT_BF16;
    const u32x4* pw1n[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    DeformBwdK k;
    DeformBwdSK ks;
    k.n = ks.n = n;
    if (split) {
        if (!a->mask || !a->act) return FDGS_EINVAL;
        u32x4* pk = reinterpret_cast<u32x4*>(ws_packed);
        SFragTable t;
        t.n = 0;
        t.d[t.n++] = SFragDesc{a->p.w0, pk, 32, DW, 32, FRAG_TRANSPOSE | FRAG_PERMROWS};
        ks.pw0t = pk; pk += sfrag_vec4(32, DW);
        for (int h = 0; h < 5; ++h) {
            t.d[t.n++] = SFragDesc{a->p.w1[h], pk, DW, DW, DW, FRAG_TRANSPOSE};
            ks.pw1t[h] = pk; pk += sfrag_vec4(DW, DW);
        }
        t.d[t.n++] = SFragDesc{a->p.w2[4], pk, DW, 48, DW, FRAG_TRANSPOSE};
        ks.pw2t_shs = pk; pk += sfrag_vec4(DW, 48);
        for (int h = 0; h < 5; ++h) {
            t.d[t.n++] = SFragDesc{a->p.w1[h], pk, DW, DW, DW, 0};
            pw1n[h] = pk; pk += sfrag_vec4(DW, DW);
        }
        // Three clears ride in this launch (each took a launch of its own before): the private copies of the small
        // accumulators (dW0 / db0, dW2 / db2, the opacity head's Q / T: contiguous), the channel-last plane
        // gradients, and the brick histogram.  Everything that reads them is enqueued behind this launch, on this
        // stream or behind a fork event recorded after it.
        {
            const size_t dw0_floats = (size_t)DW0_SLABS * (DW * 32 + DW);
            float* slabs_first = ws_yfeat + (size_t)48 * ((size_t)n + 16);
            const size_t slab_floats = dw0_floats + (size_t)W2_SLABS * W2_SLAB_FLOATS + DW * DW + DW;
            static_assert(((size_t)DW0_SLABS * (DW * 32 + DW)) % 4 == 0 && ((size_t)W2_SLABS * W2_SLAB_FLOATS + DW * DW + DW) % 4 == 0,
                          "the cleared regions are whole 16-byte words");
            t.d[t.n++] = SFragDesc{nullptr, reinterpret_cast<u32x4*>(slabs_first), (int)(slab_floats / 4), 0, 0, FRAG_ZERO};
            t.d[t.n++] = SFragDesc{nullptr, reinterpret_cast<u32x4*>(grads_cl), (int)(pf / 4), 0, 0, FRAG_ZERO};
            int* brick_counts = ws_brick + n + (n + 1) / 2;
            // the histogram starts at a 4-byte boundary only: clear from the 16-byte boundary below it (the words in
            // between are the tail of brick_of, rewritten by brick_hist_kernel before anything reads them)
            const size_t mis = ((size_t)brick_counts & 15) / 4;
            t.d[t.n++] = SFragDesc{nullptr, reinterpret_cast<u32x4*>(brick_counts - mis), (int)((BRICK_MAX + mis + 3) / 4), 0, 0,
                                   FRAG_ZERO};
        }
        rc = launch_pack_split(t, s, 64);
        if (rc != FDGS_OK) return rc;
        for (int h = 0; h < 5; ++h) ks.w2[h] = a->p.w2[h];
        ks.mask = reinterpret_cast<const uint2*>(a->mask);
        ks.n_total = n;
    } else {
        float* pk = ws_packed;
        FragTable t;
        t.n = 0;
        t.d[t.n++] = FragDesc{a->p.w0, pk, 32, DW, 32, FRAG_TRANSPOSE | FRAG_PERMROWS};
        k.pw0t = pk; pk += frag_floats(32, DW);
        for (int h = 0; h < 5; ++h) {
            t.d[t.n++] = FragDesc{a->p.w1[h], pk, DW, DW, DW, 0};
            k.pw1[h] = pk; pk += frag_floats(DW, DW);
            t.d[t.n++] = FragDesc{a->p.w1[h], pk, DW, DW, DW, FRAG_TRANSPOSE};
            k.pw1t[h] = pk; pk += frag_floats(DW, DW);
            t.d[t.n++] = FragDesc{a->p.w2[h], pk, DW, h_head_dim[h], DW, FRAG_TRANSPOSE};
            k.pw2t[h] = pk; pk += frag_floats(DW, h_head_dim[h]);
        }
        rc = launch_pack(t, s);
        if (rc != FDGS_OK) return rc;
    }
    for (int h = 0; h < 5; ++h) k.b1[h] = a->p.b1[h];
    k.hidden = a->hidden;
    k.g[0] = a->g_xyz; k.g[1] = a->g_scales; k.g[2] = a->g_rotations; k.g[3] = a->g_opacity; k.g[4] = a->g_shs;
    k.ws_a1 = ws_a1; k.ws_dh1 = ks.ws_dh1 = ws_dh1; k.ws_dhid = ks.ws_dhid = ws_dhid;
    k.ws_df = ks.ws_df = ws_df;
    // dW0[nn][f] = sum_p dhid[p][nn] * feat[p][f], db0 = column sums of dhid (split mode): the same split product as
    // the dW1 kernels with one 32-column block.  yfeat: the features as points-as-K pieces from the forward, or
    // NULL to build them from ws_feat (written by the HexPlane backward, so only after it).
    // fold_later: leave the fold of the private copies to the launch that folds the dW2 / db2 copies at the end of
    // the call (a launch of its own in the middle of the weight-gradient chain waited 45 us for a free slot on
    // the busy chip)
    float* dw0_slab_w = nullptr;
    float* dw0_slab_b = nullptr;
    auto launch_dw0_split = [&](hipStream_t hs, const u32x4* yfeat, bool fold_later) -> int {
        const int chunks = (n + 15) / 16;
        if (!yfeat) {
            WGradPrepK pk;
            pk.n = n;
            pk.d[0] = WGradPrepDesc{ws_feat, 32, 32, 1, 0, reinterpret_cast<u32x4*>(ws_yfeat), nullptr};
            for (int i = 1; i < WPREP_MAX; ++i) pk.d[i] = pk.d[0];
            int pblocks = (chunks + 3) / 4;
            if (pblocks > 4096) pblocks = 4096;
            hipLaunchKernelGGL(wgrad_prep_split_kernel, dim3(pblocks, 1), dim3(256), 0, hs, pk);
            FDGS_CHECK_LAUNCH();
            yfeat = reinterpret_cast<const u32x4*>(ws_yfeat);
        }
        int per = (chunks + 1023) / 1024;
        per = (per + 3) / 4 * 4;                // the kernel walks a range in groups of four chunks
        if (per < 4) per = 4;
        WGradSK w1;
        w1.n = n;
        w1.chunks_per_block = per;
        w1.heads_together = 0;
        for (int i = 0; i < WGS_MAX; ++i) w1.d[i].C = nullptr;
        // dW0 is 128 rows of 128 bytes: every workgroup would hit every one of them; private copies instead
        float* slab_w = ws_yfeat + (size_t)48 * ((size_t)n + 16);
        float* slab_b = slab_w + (size_t)DW0_SLABS * DW * 32;
        // fold_later: these copies lie right before the dW2 / db2 copies and were cleared with them
        if (!fold_later &&
            hipMemsetAsync(slab_w, 0, (size_t)DW0_SLABS * (DW * 32 + DW) * sizeof(float), hs) != hipSuccess)
            return FDGS_ELAUNCH;
        w1.d[0] = WGradSDesc{ws_dhid, DW, yfeat, slab_w, 32, 1, 32,
                             a->dp.b0 ? slab_b : nullptr, DW0_SLABS, DW * 32, DW, nullptr, 0, nullptr, nullptr};
        if (!a->dp.w0) w1.d[0].C = nullptr;
        hipLaunchKernelGGL(deform_wgrad_split_kernel<1>, dim3((chunks + per - 1) / per, 1), dim3(256), 0, hs, w1);
        FDGS_CHECK_LAUNCH();
        if (fold_later) {
            dw0_slab_w = a->dp.w0 ? slab_w : nullptr;
            dw0_slab_b = (a->dp.w0 && a->dp.b0) ? slab_b : nullptr;
            return FDGS_OK;
        }
        SlabReduceK sr;
        for (int e = 0; e < SLAB_ENTRIES; ++e) { sr.slab[e] = nullptr; sr.out[e] = nullptr; sr.count[e] = 0; sr.stride[e] = 0; }
        sr.slab[0] = slab_w; sr.out[0] = a->dp.w0; sr.count[0] = DW * 32; sr.stride[0] = DW * 32;
        sr.slab[1] = slab_b; sr.out[1] = a->dp.w0 ? a->dp.b0 : nullptr; sr.count[1] = DW; sr.stride[1] = DW;
        sr.slabs = DW0_SLABS;
        hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(DW * 32 / 256, 2), dim3(256), 0, hs, sr);
        FDGS_CHECK_LAUNCH();
        return FDGS_OK;
    };
    // The brick path of the HexPlane backward: float64 windows of all twelve planes in LDS, on the coarsest brick grid
    // whose windows fit.  Decided before pass 1 because the binning of the points reads nothing but xyz and may run
    // beside that pass.
    hk.lds_time = (a->time == nullptr) ? 1 : 0;
    HexBrickK bk;
    BrickBinK bb;
    int brick_lds_doubles = 0;
    bool bricks = false;
    {
        const char* mode = getenv("FDGS_HEX_BWD");
        if (!(mode && mode[0] == 'f')) {            // FDGS_HEX_BWD=flat keeps the one-atomic-per-corner kernel
            static const int cand[4][3] = {{8, 8, 8}, {8, 8, 16}, {8, 16, 16}, {16, 16, 16}};
            int first = 0, last = 3;
            if (mode && mode[0] >= '0' && mode[0] <= '3') first = last = mode[0] - '0';
            for (int ci = first; ci <= last && !bricks; ++ci) {
                int fl = 0;
                for (int l = 0; l < 2; ++l) {
                    int tw[3];
                    for (int ax = 0; ax < 3; ++ax) tw[ax] = bk.tw[l][ax] = (a->p.res[l][ax] - 1) / cand[ci][ax] + 3;
                    fl += (tw[0] * tw[1] + tw[0] * tw[2] + tw[1] * tw[2]) * DF;
                    if (hk.lds_time) fl += (tw[0] + tw[1] + tw[2]) * DF;
                }
                if ((size_t)fl * sizeof(double) <= 128 * 1024) {
                    bricks = true;
                    brick_lds_doubles = fl;
                    for (int ax = 0; ax < 3; ++ax) bk.nb[ax] = cand[ci][ax];
                }
            }
        }
    }
    if (bricks) {
        bb.n = n; bb.xyz = a->xyz; bb.aabb = a->p.aabb;
        bb.perm = ws_brick;
        bb.brick_of = reinterpret_cast<unsigned short*>(ws_brick + n);
        bb.counts = ws_brick + n + (n + 1) / 2;
        bb.cursor = bb.counts + BRICK_MAX;
        bb.work_count = bb.cursor + BRICK_MAX;
        bb.work = bb.work_count + 16;
        for (int ax = 0; ax < 3; ++ax) bb.nb[ax] = bk.nb[ax];
    }
    bool binned = false;
    auto brick_binning = [&](hipStream_t hs) -> int {
        const int nbr = bk.nb[0] * bk.nb[1] * bk.nb[2];
        if (!split && hipMemsetAsync(bb.counts, 0, (size_t)nbr * sizeof(int), hs) != hipSuccess) return FDGS_ELAUNCH;
        const int sort_blocks = (n + BRICK_SORT_PTS - 1) / BRICK_SORT_PTS;
        hipLaunchKernelGGL(brick_hist_kernel, dim3(sort_blocks), dim3(256), 0, hs, bb);
        FDGS_CHECK_LAUNCH();
        hipLaunchKernelGGL(brick_plan_kernel, dim3(1), dim3(1024), 0, hs, bb);
        FDGS_CHECK_LAUNCH();
        hipLaunchKernelGGL(brick_scatter_kernel, dim3(sort_blocks), dim3(256), 0, hs, bb);
        FDGS_CHECK_LAUNCH();
        binned = true;
        return FDGS_OK;
    };
    // pass 1b + dW0 + plane-gradient layout change: everything downstream of dfeat; runs on stream hs
    bool grads_cleared = split;          // split mode: cleared by the pack launch
    auto hex_tail = [&](hipStream_t hs) -> int {
        if (!grads_cleared && hipMemsetAsync(grads_cl, 0, pf * sizeof(float), hs) != hipSuccess) return FDGS_ELAUNCH;
        // HexPlane backward (plane scatter, coordinate gradient, features for dW0)
        hk.aabb = a->p.aabb; hk.xyz = a->xyz; hk.time = a->time; hk.time_scalar = a->time_scalar;
        hk.ws_df = ws_df; hk.g_xyz = a->g_xyz; hk.d_xyz = a->d_xyz;
        hk.ws_feat = (a->yfeat && a->p.matmul == FDGS_MATMUL_SPLIT_BF16) ? nullptr : ws_feat;
        if (bricks) {
            if (!binned) {
                rc = brick_binning(hs);
                if (rc != FDGS_OK) return rc;
            }
            bk.h = hk; bk.perm = bb.perm; bk.work = bb.work; bk.work_count = bb.work_count;
            const size_t bytes = (size_t)brick_lds_doubles * sizeof(double);
            if (bytes > 64 * 1024 &&
                hipFuncSetAttribute((const void*)hexplane_bwd_brick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bytes) != hipSuccess)
                return FDGS_ELAUNCH;
            int max_work = bk.nb[0] * bk.nb[1] * bk.nb[2] + n / BRICK_CH + 1;     // <= brick_max_work(n)
            if (max_work > n) max_work = n;
            hipLaunchKernelGGL(hexplane_bwd_brick_kernel, dim3(max_work), dim3(BRICK_THREADS), bytes, hs, bk, brick_lds_doubles);
            FDGS_CHECK_LAUNCH();
        } else {
            int hex_lds_floats = 0;
            for (int l = 0; l < 2; ++l)
                for (int tp = 0; tp < 3; ++tp) hex_lds_floats += 2 * a->p.res[l][tp] * DF;
            if ((size_t)hex_lds_floats * sizeof(float) > 150 * 1024) { hk.lds_time = 0; }
            const size_t hex_lds_bytes = hk.lds_time ? (size_t)hex_lds_floats * sizeof(float) : 0;
            if (hex_lds_bytes > 64 * 1024 &&
                hipFuncSetAttribute((const void*)hexplane_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)hex_lds_bytes) != hipSuccess)
                return FDGS_ELAUNCH;
            const long long waves_needed = ((long long)2 * n + 3) / 4;
            long long hb = (waves_needed + 3) / 4;
            if (hb > 512) hb = 512;
            if (hb < 1) hb = 1;
            hipLaunchKernelGGL(hexplane_bwd_kernel, dim3((unsigned)hb), dim3(256), hex_lds_bytes, hs, hk, hex_lds_floats);
            FDGS_CHECK_LAUNCH();
        }
        if (a->p.matmul == FDGS_MATMUL_SPLIT_BF16) {
            if (!a->yfeat) {                        // with the forward's pieces this product is already on its way
                rc = launch_dw0_split(hs, nullptr, false);
                if (rc != FDGS_OK) return rc;
            }
        } else {                                    // dW0[nn][f] = sum_p dhid[p][nn] * feat[p][f]  (fp32 MFMA)
            WGradK w;
            w.n = n;
            w.chunk = 1024;
            for (int i = 0; i < WG_MAX; ++i) {
                w.d[i].C = nullptr; w.d[i].X = nullptr; w.d[i].Y = nullptr; w.d[i].db = nullptr;
                w.d[i].M = 0; w.d[i].NN = 0; w.d[i].ldx = w.d[i].ldy = w.d[i].ldc = 0; w.d[i].relu_y = 0;
            }
            WGradDesc& d = w.d[0];
            d.X = ws_dhid; d.ldx = DW; d.M = DW;
            d.Y = ws_feat; d.ldy = 32; d.NN = 32; d.relu_y = 0;
            d.C = a->dp.w0; d.ldc = 32; d.db = a->dp.b0;
            hipLaunchKernelGGL(deform_wgrad_kernel, dim3((n + w.chunk - 1) / w.chunk, 1), dim3(256), 0, hs, w);
            FDGS_CHECK_LAUNCH();
        }

        // plane gradients back to the parameter layout [16][H][W] (one launch, += into the caller's buffers)
        {
            PlaneTable t;
            int max_hw = 0;
            for (int l = 0; l < 2; ++l)
                for (int ci = 0; ci < 6; ++ci) {
                    const int hw = (int)plane_texels(&a->p, l, ci);
                    t.src[6 * l + ci] = hk.gpl[l][ci];
                    t.dst[6 * l + ci] = a->dp.planes[l][ci];
                    t.hw[6 * l + ci] = hw;
                    if (hw > max_hw) max_hw = hw;
                }
            hipLaunchKernelGGL(planes_grad_from_channel_last_kernel, dim3((max_hw + 255) / 256, 12), dim3(256), 0, hs, t);
            FDGS_CHECK_LAUNCH();
        }
        return FDGS_OK;
    };
    // Points are processed in chunks: each chunk's a1 / dh1 operands are consumed by the weight-gradient
    // products right after they are produced and the workspace holds one chunk only.
    SideStream side_handles{(hipStream_t)a->side_stream, (hipEvent_t)a->fork_event, (hipEvent_t)a->join_event};
    SideStream* side = (a->side_stream && a->fork_event && a->join_event && a->side_stream != stream &&
                        !getenv("FDGS_SINGLE_STREAM")) ? &side_handles : nullptr;
    static const int kdim[5] = {3, 3, 4, 1, 48};
    const float* gs[5] = {a->g_xyz, a->g_scales, a->g_rotations, a->g_opacity, a->g_shs};
    const int chunk_pts = bwd_chunk_points(n);
    // private copies of the small dW2 / db2 accumulators (split mode), folded into the outputs after the loop
    float* w2_slabs = ws_yfeat + (size_t)48 * ((size_t)n + 16) + (size_t)DW0_SLABS * (DW * 32 + DW);
    float* op_q = w2_slabs + (size_t)W2_SLABS * W2_SLAB_FLOATS;      // Q [128][128], then T [128]
    // (split mode: these copies, the dW0 / db0 copies before them and Q / T behind them were cleared by the pack launch)
    // What depends on nothing of pass 1 goes to the side stream first and runs beside that pass (whose one wave per
    // SIMD leaves the last round of its launch partly idle): the binning of the points into bricks and the column
    // sums of the cotangents (db2), over all points at once.  FDGS_NO_EARLY_SIDE keeps the single fork after pass 1.
    const bool early_side = side && !getenv("FDGS_NO_EARLY_SIDE");
    bool colsum_done = false;
    if (early_side) {
        if (hipEventRecord(side->fork, s) != hipSuccess || hipStreamWaitEvent(side->stream, side->fork, 0) != hipSuccess)
            return FDGS_ELAUNCH;
        if (!grads_cleared && hipMemsetAsync(grads_cl, 0, pf * sizeof(float), side->stream) != hipSuccess)
            return FDGS_ELAUNCH;
        grads_cleared = true;
        if (bricks) {
            rc = brick_binning(side->stream);
            if (rc != FDGS_OK) return rc;
        }
        if (split) {
            ColSumK cs;
            cs.n = n;
            for (int h = 0; h < 5; ++h) { cs.X[h] = gs[h]; cs.ncols[h] = kdim[h]; cs.out[h] = a->dp.b2[h]; }
            cs.slabs_base = w2_slabs + (size_t)W2_SLABS * 59 * DW; cs.slabs = W2_SLABS;
            hipLaunchKernelGGL(colsum_kernel, dim3(1024, 5), dim3(256), 0, side->stream, cs);
            FDGS_CHECK_LAUNCH();
            colsum_done = true;
        }
    }
    for (int c0 = 0; c0 < n; c0 += chunk_pts) {
        const int cn = (n - c0) < chunk_pts ? (n - c0) : chunk_pts;
        DeformBwdK kc = k;
        kc.n = cn;
        kc.hidden = a->hidden + (size_t)c0 * DW;
        for (int h = 0; h < 5; ++h) kc.g[h] = gs[h] + (size_t)c0 * kdim[h];
        kc.ws_dhid = ws_dhid + (size_t)c0 * DW;
        kc.ws_df = ws_df + (size_t)c0 * 32;
        if (split) {
            DeformBwdSK sc = ks;
            sc.n = cn;
            sc.p_first = c0;
            sc.dh1_rows = chunk_cap;
            for (int h = 0; h < 5; ++h) sc.g[h] = kc.g[h];
            sc.ws_dhid = kc.ws_dhid; sc.ws_df = kc.ws_df;
            hipLaunchKernelGGL(deform_bwd_points_split_kernel, dim3((cn + 32 * SCB - 1) / (32 * SCB)), dim3(64), 0, s, sc);
        } else {
            hipLaunchKernelGGL(deform_bwd_points_kernel, dim3((cn + 32 * NCB - 1) / (32 * NCB)), dim3(64), 0, s, kc);
        }
        FDGS_CHECK_LAUNCH();
        if (c0 + cn >= n) {                     // pass 1 complete: fork the HexPlane tail
            if (side) {
                if (hipEventRecord(side->fork, s) != hipSuccess ||
                    hipStreamWaitEvent(side->stream, side->fork, 0) != hipSuccess)
                    return FDGS_ELAUNCH;
                rc = hex_tail(side->stream);
                if (rc != FDGS_OK) return rc;
                if (hipEventRecord(side->join, side->stream) != hipSuccess) return FDGS_ELAUNCH;
            }
            if (split && a->yfeat) {            // needs dhid only: runs beside the HexPlane backward, not after it
                rc = launch_dw0_split(s, reinterpret_cast<const u32x4*>(a->yfeat), true);
                if (rc != FDGS_OK) return rc;
            }
        }
        if (split) {
            // dW1_h[m][n] = sum_p dh1_h[p][m] relu(hidden)[p][n]: relu(hidden) pre-split once, dh1 split on the fly
            const int chunks = (cn + 15) / 16;
            const u32x4* y_hidden = reinterpret_cast<const u32x4*>(ws_ysplit);
            if (a->yhid) {                      // written by the forward: chunk c0 / 16 onwards
                y_hidden = reinterpret_cast<const u32x4*>(a->yhid) + (size_t)(c0 / 16) * 4 * 3 * 64;
            } else {
                WGradPrepK pk;
                pk.n = cn;
                pk.d[0] = WGradPrepDesc{kc.hidden, DW, DW, 4, 1, reinterpret_cast<u32x4*>(ws_ysplit), nullptr};
                for (int i = 1; i < WPREP_MAX; ++i) pk.d[i] = pk.d[0];
                int pblocks = (chunks + 3) / 4;
                if (pblocks > 4096) pblocks = 4096;
                hipLaunchKernelGGL(wgrad_prep_split_kernel, dim3(pblocks, 1), dim3(256), 0, s, pk);
                FDGS_CHECK_LAUNCH();
            }
            static const int wg_ranges = getenv("FDGS_WGRAD_RANGES") ? atoi(getenv("FDGS_WGRAD_RANGES")) : 204;
            int per = (chunks + wg_ranges - 1) / wg_ranges;     // about a thousand workgroups of four waves
            per = (per + 3) / 4 * 4;                            // the kernel walks a range in groups of four chunks
            if (per < 4) per = 4;
            WGradSK w4;
            w4.n = cn;
            w4.chunks_per_block = per;
            w4.heads_together = getenv("FDGS_WGRAD_GRID2D") ? 0 : 5;
            for (int i = 0; i < WGS_MAX; ++i) w4.d[i].C = nullptr;
            for (int h = 0; h < 5; ++h)
                {
                    w4.d[h] = WGradSDesc{ws_dh1 + (size_t)h * chunk_cap * DW, DW, y_hidden, a->dp.w1[h], DW, 1, DW,
                                         a->dp.b1[h], 0, 0, 0, nullptr, 0, nullptr, nullptr};
                    if (h < 4) {            // dh1 of the small heads is not stored: formed from g, W2 and the masks
                        w4.d[h].X = nullptr;
                        w4.d[h].cot = kc.g[h]; w4.d[h].k = kdim[h]; w4.d[h].w2 = a->p.w2[h];
                        w4.d[h].mask = reinterpret_cast<const uint2*>(a->mask) + ((size_t)h * n + c0) * 2;
                    }
                    if (h == 3) {           // one output: unit weights, raw sums into Q / T (folded after the loop)
                        w4.d[h].w2 = nullptr;
                        w4.d[h].C = op_q; w4.d[h].rowsum = op_q + DW * DW;
                    }
                }
            const int ranges = (chunks + per - 1) / per;
            const dim3 grid4 = w4.heads_together ? dim3(8 * 5 * ((ranges + 7) / 8), 1) : dim3(ranges, 5);
            hipLaunchKernelGGL(deform_wgrad_split_kernel<4>, grid4, dim3(256), 0, s, w4);
            FDGS_CHECK_LAUNCH();
            // dW2_h = g_h^T relu(h1_h) with h1 recomputed in place (nothing stored), db2_h = column sums of g_h
            WGrad2SK w2k;
            w2k.n = cn;
            const int tiles = (cn + 63) / 64;
            int tpb = (tiles + 203) / 204;
            if (tpb < 1) tpb = 1;
            w2k.tiles_per_block = tpb;
            w2k.hsplit = reinterpret_cast<const u32x4*>(a->act) + (size_t)(c0 / 32) * 8 * 3 * 64;
            ColSumK cs;
            cs.n = cn;
            for (int h = 0; h < 5; ++h) {
                w2k.pw1n[h] = pw1n[h]; w2k.b1[h] = a->p.b1[h]; w2k.g[h] = kc.g[h]; w2k.dw2[h] = a->dp.w2[h];
                cs.X[h] = kc.g[h]; cs.ncols[h] = kdim[h]; cs.out[h] = a->dp.b2[h];
            }
            w2k.tiles_per_block = tpb = (tiles + 1023) / 1024;       // about 1000 waves per launch
            w2k.dw2_slabs = w2_slabs; w2k.slabs = W2_SLABS; w2k.small_heads = 3;
            cs.slabs_base = w2_slabs + (size_t)W2_SLABS * 59 * DW; cs.slabs = W2_SLABS;
            hipLaunchKernelGGL(deform_wgrad2_split_kernel<1>, dim3((tiles + tpb - 1) / tpb), dim3(64), 0, s, w2k);
            FDGS_CHECK_LAUNCH();
            hipLaunchKernelGGL(deform_wgrad2_split_kernel<2>, dim3((tiles + tpb - 1) / tpb), dim3(64), 0, s, w2k);
            FDGS_CHECK_LAUNCH();
            if (!colsum_done) {
                hipLaunchKernelGGL(colsum_kernel, dim3(1024, 5), dim3(256), 0, s, cs);
                FDGS_CHECK_LAUNCH();
            }
            continue;
        }
        WGradK w;
        w.n = cn;
        w.chunk = 1024;
        int nd = 0;
        for (int h = 0; h < 5 && !split; ++h) { // dW1_h[nn][kk] = sum_p dh1[p][nn] * relu(hidden)[p][kk]
            WGradDesc& d = w.d[nd++];
            d.X = ws_dh1 + h * DW; d.ldx = 640; d.M = DW;
            d.Y = kc.hidden; d.ldy = DW; d.NN = DW; d.relu_y = 1;
            d.C = a->dp.w1[h]; d.ldc = DW; d.db = a->dp.b1[h];
        }
        for (int h = 0; h < 5; ++h) {           // dW2_h[o][kk] = sum_p g[p][o] * a1_h[p][kk]
            WGradDesc& d = w.d[nd++];
            d.X = kc.g[h]; d.ldx = kdim[h]; d.M = kdim[h];
            d.Y = ws_a1 + h * DW; d.ldy = 640; d.NN = DW; d.relu_y = 0;
            d.C = a->dp.w2[h]; d.ldc = DW; d.db = a->dp.b2[h];
        }
        for (int i = nd; i < WG_MAX; ++i) {
            w.d[i].C = nullptr; w.d[i].X = nullptr; w.d[i].Y = nullptr; w.d[i].db = nullptr;
            w.d[i].M = 0; w.d[i].NN = 0; w.d[i].ldx = w.d[i].ldy = w.d[i].ldc = 0; w.d[i].relu_y = 0;
        }
        hipLaunchKernelGGL(deform_wgrad_kernel, dim3((cn + w.chunk - 1) / w.chunk, nd), dim3(256), 0, s, w);
        FDGS_CHECK_LAUNCH();
    }
    const bool joined_early = side && colsum_done;
    if (joined_early) {                         // the fold below reads the db2 copies written on the side stream
        if (hipStreamWaitEvent(s, side->join, 0) != hipSuccess) return FDGS_ELAUNCH;
        side = nullptr;
    }
    if (split) {
        OpacityFoldK of;
        of.Q = op_q; of.T = op_q + DW * DW; of.w1 = a->p.w1[3]; of.b1 = a->p.b1[3]; of.w2 = a->p.w2[3];
        of.dw1 = a->dp.w1[3]; of.db1 = a->dp.b1[3]; of.dw2 = a->dp.w2[3];
        static const int row0[5] = {0, 3, 6, 10, 11};
        SlabReduceK sr;
        for (int e = 0; e < SLAB_ENTRIES; ++e) { sr.slab[e] = nullptr; sr.out[e] = nullptr; sr.count[e] = 0; sr.stride[e] = 0; }
        for (int h = 0; h < 5; ++h) {
            sr.slab[h] = w2_slabs + (size_t)row0[h] * DW; sr.out[h] = a->dp.w2[h];
            sr.count[h] = kdim[h] * DW; sr.stride[h] = 59 * DW;
            sr.slab[5 + h] = w2_slabs + (size_t)W2_SLABS * 59 * DW + 64 * h; sr.out[5 + h] = a->dp.b2[h];
            sr.count[5 + h] = kdim[h]; sr.stride[5 + h] = 5 * 64;
        }
        static_assert(DW0_SLABS == W2_SLABS && SLAB_ENTRIES >= 12, "one fold launch for both sets of copies");
        if (dw0_slab_w) { sr.slab[10] = dw0_slab_w; sr.out[10] = a->dp.w0; sr.count[10] = DW * 32; sr.stride[10] = DW * 32; }
        if (dw0_slab_b) { sr.slab[11] = dw0_slab_b; sr.out[11] = a->dp.b0; sr.count[11] = DW; sr.stride[11] = DW; }
        sr.slabs = W2_SLABS;
        // the opacity head's dW2 comes from the fold (its rows of the copies are zero): it is written by one of the
        // two halves of this launch only
        sr.out[3] = nullptr;
        static_assert(DW % 8 == 0 && DW % 64 == 0 && DW <= 256, "rows of the opacity fold over the launch's workgroups");
        hipLaunchKernelGGL(deform_bwd_finish_kernel, dim3(8, SLAB_ENTRIES + DW / 8), dim3(256), 0, s, sr, of);
        FDGS_CHECK_LAUNCH();
    }
    if (side) {                                 // join
        if (hipStreamWaitEvent(s, side->join, 0) != hipSuccess) return FDGS_ELAUNCH;
    } else if (!joined_early) {
        rc = hex_tail(s);
        if (rc != FDGS_OK) return rc;
    }
    return FDGS_OK;
}
