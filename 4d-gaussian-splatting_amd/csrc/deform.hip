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
// consecutive steps r = 4t..4t+3 need W[n][32k + 8t + 4h + 0..3], one 16-byte LDS
// read from a row-major image padded to a 132-float row stride (conflict-free).
//
// The two halves of a wave also split the HexPlane work: lane (c,h) interpolates
// the 16 channels of resolution level h for point c, which is exactly the K pair
// {s, 16+s} of the first Linear, so the features never move between lanes.
//
// Planes are gathered from a channel-last copy [H][W][16] (one 64-byte texel per
// lookup instead of 16 strided lines), rebuilt from the parameter layout
// [16][H][W] at the start of every call (9.5 MB, a few microseconds).
#include "fdgs_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define DW FDGS_DEFORM_WIDTH             // 128
#define DF FDGS_DEFORM_FEAT              // 16
#define WSTRIDE 132                      // padded row stride (floats) of a [rows][128] LDS weight image
#define W0STRIDE 36                      // padded row stride of the [128][32] first-layer image
#define TILE_PTS 128                     // points per workgroup (4 waves x 32)
#define LDS_FLOATS (DW * WSTRIDE)        // 16896 floats = 67584 bytes

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
    return b;
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
// LDS staging of row-major weight matrices
// ---------------------------------------------------------------------------
// [rows][128] global -> LDS with row stride WSTRIDE; rows [rows, pad_rows) are zero filled.
__device__ __forceinline__ void stage_w128(float* lds, const float* __restrict__ w, int rows, int pad_rows) {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(w);
    for (int idx = threadIdx.x; idx < pad_rows * 32; idx += blockDim.x) {
        const int row = idx >> 5, c4 = idx & 31;
        const float4 v = row < rows ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(lds + row * WSTRIDE + c4 * 4) = v;
    }
}
// [128][32] global -> LDS with row stride W0STRIDE
__device__ __forceinline__ void stage_w0(float* lds, const float* __restrict__ w) {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(w);
    for (int idx = threadIdx.x; idx < DW * 8; idx += blockDim.x) {
        const int row = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<float4*>(lds + row * W0STRIDE + c4 * 4) = src[idx];
    }
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// acc[nb] (+)= W[32nb.., :] * relu?(X)   for a 128-wide input held in x[4] (accumulator layout)
template <bool RELU>
__device__ __forceinline__ void layer128(const float* lds_w, int nb, int c, int h, const floatx16* x, floatx16& acc) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 a4 = *reinterpret_cast<const float4*>(lds_w + (32 * nb + c) * WSTRIDE + 32 * b + 8 * t + 4 * h);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float xv = x[b][4 * t + q];
                if (RELU) xv = fmaxf(xv, 0.0f);
                acc = MFMA(av[q], xv, acc);
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
    const float* w0; const float* b0;
    const float* w1[5]; const float* b1[5]; const float* w2[5]; const float* b2[5];
    const float* xyz; const float* in[5];     // in[0] = xyz ... in[4] = shs (residual inputs)
    const float* time; float time_scalar;
    float* out[5];
    float* save_hidden;
};

// two workgroups per CU: one stages weights while the other issues MFMAs
__global__ void __launch_bounds__(256, 2)
deform_fwd_kernel(DeformFwdK a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int p = blockIdx.x * TILE_PTS + wave * 32 + c;
    const bool pvalid = p < a.n;
    const int pc = pvalid ? p : a.n - 1;

    float co[4], sc3[3], feat[DF];
    point_coords(a.xyz, a.aabb, a.time, a.time_scalar, pc, co, sc3);
    level_features(a.ps, h, co, feat, nullptr);

    // hidden^T = W0 f^T + b0
    stage_w0(lds, a.w0);
    __syncthreads();
    floatx16 hid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        hid[nb] = bias_init(a.b0, nb, h, DW);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 a4 = *reinterpret_cast<const float4*>(lds + (32 * nb + c) * W0STRIDE + 16 * h + 4 * t);
            hid[nb] = MFMA(a4.x, feat[4 * t + 0], hid[nb]);
            hid[nb] = MFMA(a4.y, feat[4 * t + 1], hid[nb]);
            hid[nb] = MFMA(a4.z, feat[4 * t + 2], hid[nb]);
            hid[nb] = MFMA(a4.w, feat[4 * t + 3], hid[nb]);
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
        const float* w1 = sel5(a.w1, head); const float* b1 = sel5(a.b1, head);
        const float* w2 = sel5(a.w2, head); const float* b2 = sel5(a.b2, head);
        const float* in = sel5(a.in, head); float* out = sel5(a.out, head);
        __syncthreads();
        stage_w128(lds, w1, DW, DW);
        __syncthreads();
        floatx16 h1[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            h1[nb] = bias_init(b1, nb, h, DW);
            layer128<true>(lds, nb, c, h, hid, h1[nb]);
        }
        __syncthreads();
        const int nblk = (k + 31) >> 5;
        stage_w128(lds, w2, k, 32 * nblk);
        __syncthreads();
        for (int ob = 0; ob < nblk; ++ob) {
            floatx16 o = bias_init(b2, ob, h, k);
            layer128<true>(lds, ob, c, h, h1, o);
            if (pvalid) {
                if ((k & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 32 * ob + 8 * q + 4 * h;
                        if (row < k) {
                            const float4 iv = *reinterpret_cast<const float4*>(in + (size_t)p * k + row);
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

extern "C" size_t fdgs_deform_fwd_workspace_bytes(const fdgs_deform_params* p) {
    if (!p || p->num_levels < 1 || p->num_levels > FDGS_DEFORM_MAX_LEVELS) return 0;
    return total_plane_floats(p) * sizeof(float);
}

// fills ps with the channel-last plane addresses inside ws and (unless the caller
// says ws already holds them) rebuilds the copies with one launch
static int make_plane_set(const fdgs_deform_params* p, float* ws, PlaneSet* ps, bool rebuild, hipStream_t s) {
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
            off += (size_t)hw * DF;
            if (hw > max_hw) max_hw = hw;
        }
    }
    if (rebuild) {
        hipLaunchKernelGGL(planes_to_channel_last_kernel, dim3((max_hw + 255) / 256, 12), dim3(256), 0, s, t);
        FDGS_CHECK_LAUNCH();
    }
    return FDGS_OK;
}

extern "C" int fdgs_deform_fwd(const fdgs_deform_fwd_args* a, void* stream) {
    if (!a || a->num_points < 0 || !params_ok(&a->p)) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->xyz || !a->scales || !a->rotations || !a->opacity || !a->shs || !a->out_xyz || !a->out_scales ||
        !a->out_rotations || !a->out_opacity || !a->out_shs || !a->workspace)
        return FDGS_EINVAL;
    if (a->workspace_bytes < fdgs_deform_fwd_workspace_bytes(&a->p)) return FDGS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    DeformFwdK k;
    k.n = a->num_points;
    int rc = make_plane_set(&a->p, (float*)a->workspace, &k.ps, true, s);
    if (rc != FDGS_OK) return rc;
    k.aabb = a->p.aabb; k.w0 = a->p.w0; k.b0 = a->p.b0;
    for (int h = 0; h < 5; ++h) { k.w1[h] = a->p.w1[h]; k.b1[h] = a->p.b1[h]; k.w2[h] = a->p.w2[h]; k.b2[h] = a->p.b2[h]; }
    k.xyz = a->xyz;
    k.in[0] = a->xyz; k.in[1] = a->scales; k.in[2] = a->rotations; k.in[3] = a->opacity; k.in[4] = a->shs;
    k.out[0] = a->out_xyz; k.out[1] = a->out_scales; k.out[2] = a->out_rotations; k.out[3] = a->out_opacity;
    k.out[4] = a->out_shs;
    k.time = a->time; k.time_scalar = a->time_scalar;
    k.save_hidden = a->save_hidden;
    const int blocks = (a->num_points + TILE_PTS - 1) / TILE_PTS;
    const size_t lds_bytes = LDS_FLOATS * sizeof(float);
    if (hipFuncSetAttribute((const void*)deform_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
        return FDGS_ELAUNCH;
    hipLaunchKernelGGL(deform_fwd_kernel, dim3(blocks), dim3(256), lds_bytes, s, k);
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
    const float* w0;
    const float* w1[5]; const float* b1[5]; const float* w2[5];
    const float* hidden;
    const float* g[5];                   // output cotangents
    float* ws_a1; float* ws_dh1; float* ws_dhid; float* ws_df;
};

// one wave per SIMD: the backward keeps three 64-register accumulator tiles live (no spills at 512 registers)
__global__ void __launch_bounds__(256, 1)
deform_bwd_points_kernel(DeformBwdK a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int p = blockIdx.x * TILE_PTS + wave * 32 + c;
    const bool pvalid = p < a.n;
    const int pc = pvalid ? p : a.n - 1;

    // hidden (saved by the forward) in accumulator layout
    floatx16 hid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(a.hidden + (size_t)pc * DW + 32 * nb + 8 * q + 4 * h);
            hid[nb][4 * q] = v.x; hid[nb][4 * q + 1] = v.y; hid[nb][4 * q + 2] = v.z; hid[nb][4 * q + 3] = v.w;
        }
    floatx16 dacc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[nb][r] = 0.0f;

    for (int head = 0; head < 5; ++head) {
        const int k = head_dim(head);
        const float* w1 = sel5(a.w1, head); const float* b1 = sel5(a.b1, head);
        const float* w2 = sel5(a.w2, head); const float* g = sel5(a.g, head);
        __syncthreads();
        stage_w128(lds, w1, DW, DW);
        __syncthreads();
        // h1 = W1 relu(hidden) + b1 ; keep only the sign mask, store a1 = relu(h1)
        unsigned long long pos_mask = 0ull;       // bit (16*nb + r) : h1[nb][r] > 0
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            floatx16 h1 = bias_init(b1, nb, h, DW);
            layer128<true>(lds, nb, c, h, hid, h1);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (h1[r] > 0.0f) pos_mask |= (1ull << (16 * nb + r));
            if (pvalid) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.ws_a1 + (size_t)p * 640 + head * DW + 32 * nb + 8 * q + 4 * h) =
                        make_float4(fmaxf(h1[4 * q], 0.f), fmaxf(h1[4 * q + 1], 0.f), fmaxf(h1[4 * q + 2], 0.f),
                                    fmaxf(h1[4 * q + 3], 0.f));
            }
        }
        // da1^T[kk][p] = sum_o W2[o][kk] g[p][o]  (A = W2 read from global: lanes run along kk, coalesced)
        floatx16 dh1[4];
        const int osteps = (k + 1) >> 1;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dh1[nb][r] = 0.0f;
        }
        for (int s = 0; s < osteps; ++s) {
            const int o = 2 * s + h;
            const bool ov = o < k;
            const float gv = (ov && pvalid) ? g[(size_t)p * k + o] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const float av = ov ? w2[(size_t)o * DW + 32 * nb + c] : 0.0f;
                dh1[nb] = MFMA(av, gv, dh1[nb]);
            }
        }
        // dh1 = da1 * (h1 > 0); store for the weight gradients
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (!((pos_mask >> (16 * nb + r)) & 1ull)) dh1[nb][r] = 0.0f;
            if (pvalid) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.ws_dh1 + (size_t)p * 640 + head * DW + 32 * nb + 8 * q + 4 * h) =
                        make_float4(dh1[nb][4 * q], dh1[nb][4 * q + 1], dh1[nb][4 * q + 2], dh1[nb][4 * q + 3]);
            }
        }
        // dacc^T[kk][p] += sum_n W1[n][kk] dh1^T[n][p]   (A[i=kk][n] = W1[n][kk] from the LDS image)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float av = lds[(32 * b + drow(r, h)) * WSTRIDE + 32 * kb + c];
                    dacc[kb] = MFMA(av, dh1[b][r], dacc[kb]);
                }
    }

    // dhidden = dacc * (hidden > 0)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (!(hid[nb][r] > 0.0f)) dacc[nb][r] = 0.0f;
        if (pvalid) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(a.ws_dhid + (size_t)p * DW + 32 * nb + 8 * q + 4 * h) =
                    make_float4(dacc[nb][4 * q], dacc[nb][4 * q + 1], dacc[nb][4 * q + 2], dacc[nb][4 * q + 3]);
        }
    }

    // df^T = W0^T dhidden^T with output row i carrying feature perm(i) = 16*((i>>2)&1) + 4*(i>>3) + (i&3),
    // so that register r of lane (c,h) is dL/dfeature[16h + r]: the level this lane interpolates.
    __syncthreads();
    stage_w0(lds, a.w0);
    __syncthreads();
    floatx16 df;
#pragma unroll
    for (int r = 0; r < 16; ++r) df[r] = 0.0f;
    {
        const int pcol = 16 * ((c >> 2) & 1) + 4 * (c >> 3) + (c & 3);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float av = lds[(32 * b + drow(r, h)) * W0STRIDE + pcol];
                df = MFMA(av, dacc[b][r], df);
            }
    }

    if (pvalid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(a.ws_df + (size_t)p * 32 + 16 * h + 4 * q) =
                make_float4(df[4 * q], df[4 * q + 1], df[4 * q + 2], df[4 * q + 3]);
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
    float* ws_feat;                      // [N,32] features (for the first Linear's weight gradient)
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
            a.ws_feat[(size_t)p * 32 + 16 * h + ch] = feat;
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

extern "C" size_t fdgs_deform_bwd_workspace_bytes(const fdgs_deform_params* p, int32_t num_points) {
    if (!p || num_points < 0 || p->num_levels < 1 || p->num_levels > FDGS_DEFORM_MAX_LEVELS) return 0;
    const size_t pf = total_plane_floats(p);
    const size_t per_point = 640 + 640 + DW + 32 + 32;
    return (2 * pf + per_point * (size_t)num_points) * sizeof(float);
}

extern "C" int fdgs_deform_bwd(const fdgs_deform_bwd_args* a, void* stream) {
    if (!a || a->num_points < 0 || !params_ok(&a->p)) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->xyz || !a->hidden || !a->g_xyz || !a->g_scales || !a->g_rotations || !a->g_opacity || !a->g_shs ||
        !a->d_xyz || !a->workspace)
        return FDGS_EINVAL;
    if (a->workspace_bytes < fdgs_deform_bwd_workspace_bytes(&a->p, a->num_points)) return FDGS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int n = a->num_points;
    const size_t pf = total_plane_floats(&a->p);
    float* ws = (float*)a->workspace;
    float* planes_cl = ws;
    float* grads_cl = ws + pf;
    float* ws_a1 = grads_cl + pf;
    float* ws_dh1 = ws_a1 + (size_t)640 * n;
    float* ws_dhid = ws_dh1 + (size_t)640 * n;
    float* ws_feat = ws_dhid + (size_t)DW * n;

    float* ws_df = ws_feat + (size_t)32 * n;

    HexBwdK hk;
    hk.n = n;
    const bool reuse = a->planes_channel_last != nullptr;
    if (reuse) planes_cl = const_cast<float*>(a->planes_channel_last);
    int rc = make_plane_set(&a->p, planes_cl, &hk.ps, !reuse, s);
    if (rc != FDGS_OK) return rc;
    if (hipMemsetAsync(grads_cl, 0, pf * sizeof(float), s) != hipSuccess) return FDGS_ELAUNCH;
    {
        size_t off = 0;
        for (int l = 0; l < 2; ++l)
            for (int ci = 0; ci < 6; ++ci) {
                hk.gpl[l][ci] = grads_cl + off;
                off += plane_texels(&a->p, l, ci) * DF;
            }
    }
    DeformBwdK k;
    k.n = n;
    k.w0 = a->p.w0;
    for (int h = 0; h < 5; ++h) { k.w1[h] = a->p.w1[h]; k.b1[h] = a->p.b1[h]; k.w2[h] = a->p.w2[h]; }
    k.hidden = a->hidden;
    k.g[0] = a->g_xyz; k.g[1] = a->g_scales; k.g[2] = a->g_rotations; k.g[3] = a->g_opacity; k.g[4] = a->g_shs;
    k.ws_a1 = ws_a1; k.ws_dh1 = ws_dh1; k.ws_dhid = ws_dhid; k.ws_df = ws_df;
    const int blocks = (n + TILE_PTS - 1) / TILE_PTS;
    const size_t lds_bytes = LDS_FLOATS * sizeof(float);
    if (hipFuncSetAttribute((const void*)deform_bwd_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
        return FDGS_ELAUNCH;
    hipLaunchKernelGGL(deform_bwd_points_kernel, dim3(blocks), dim3(256), lds_bytes, s, k);
    FDGS_CHECK_LAUNCH();

    // HexPlane backward (plane scatter, coordinate gradient, features for dW0)
    hk.aabb = a->p.aabb; hk.xyz = a->xyz; hk.time = a->time; hk.time_scalar = a->time_scalar;
    hk.ws_df = ws_df; hk.ws_feat = ws_feat; hk.g_xyz = a->g_xyz; hk.d_xyz = a->d_xyz;
    hk.lds_time = (a->time == nullptr) ? 1 : 0;
    int hex_lds_floats = 0;
    for (int l = 0; l < 2; ++l)
        for (int tp = 0; tp < 3; ++tp) hex_lds_floats += 2 * a->p.res[l][tp] * DF;
    if ((size_t)hex_lds_floats * sizeof(float) > 150 * 1024) { hk.lds_time = 0; }
    const size_t hex_lds_bytes = hk.lds_time ? (size_t)hex_lds_floats * sizeof(float) : 0;
    if (hex_lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)hexplane_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)hex_lds_bytes) != hipSuccess)
        return FDGS_ELAUNCH;
    {
        const long long waves_needed = ((long long)2 * n + 3) / 4;
        long long hb = (waves_needed + 3) / 4;
        if (hb > 512) hb = 512;
        if (hb < 1) hb = 1;
        hipLaunchKernelGGL(hexplane_bwd_kernel, dim3((unsigned)hb), dim3(256), hex_lds_bytes, s, hk, hex_lds_floats);
        FDGS_CHECK_LAUNCH();
    }

    // weight gradients
    static const int kdim[5] = {3, 3, 4, 1, 48};
    WGradK w;
    w.n = n;
    w.chunk = 1024;
    int nd = 0;
    for (int h = 0; h < 5; ++h) {           // dW1_h[nn][kk] = sum_p dh1[p][nn] * relu(hidden)[p][kk]
        WGradDesc& d = w.d[nd++];
        d.X = ws_dh1 + h * DW; d.ldx = 640; d.M = DW;
        d.Y = a->hidden; d.ldy = DW; d.NN = DW; d.relu_y = 1;
        d.C = a->dp.w1[h]; d.ldc = DW; d.db = a->dp.b1[h];
        if (!d.C) d.C = nullptr;
    }
    const float* gs[5] = {a->g_xyz, a->g_scales, a->g_rotations, a->g_opacity, a->g_shs};
    for (int h = 0; h < 5; ++h) {           // dW2_h[o][kk] = sum_p g[p][o] * a1_h[p][kk]
        WGradDesc& d = w.d[nd++];
        d.X = gs[h]; d.ldx = kdim[h]; d.M = kdim[h];
        d.Y = ws_a1 + h * DW; d.ldy = 640; d.NN = DW; d.relu_y = 0;
        d.C = a->dp.w2[h]; d.ldc = DW; d.db = a->dp.b2[h];
    }
    {                                       // dW0[nn][f] = sum_p dhid[p][nn] * feat[p][f]
        WGradDesc& d = w.d[nd++];
        d.X = ws_dhid; d.ldx = DW; d.M = DW;
        d.Y = ws_feat; d.ldy = 32; d.NN = 32; d.relu_y = 0;
        d.C = a->dp.w0; d.ldc = 32; d.db = a->dp.b0;
    }
    const int chunks = (n + w.chunk - 1) / w.chunk;
    hipLaunchKernelGGL(deform_wgrad_kernel, dim3(chunks, WG_MAX), dim3(256), 0, s, w);
    FDGS_CHECK_LAUNCH();

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
        hipLaunchKernelGGL(planes_grad_from_channel_last_kernel, dim3((max_hw + 255) / 256, 12), dim3(256), 0, s, t);
        FDGS_CHECK_LAUNCH();
    }
    return FDGS_OK;
}
