// Shared device/host helpers for the gfx950 kernels of libfdgs.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <hip/hip_fp16.h>
#include "fdgs.h"

#define FDGS_WAVE 64

#define FDGS_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return FDGS_ELAUNCH; \
    } while (0)

// Rasterizer constants (the published tile-rasterizer algorithm).
#define FDGS_NEAR_Z 0.2f
#define FDGS_COV_DILATION 0.3f
#define FDGS_ALPHA_MAX 0.99f
#define FDGS_ALPHA_MIN (1.0f / 255.0f)
#define FDGS_T_STOP 0.0001f

// Spherical-harmonics constants, src/utils/sh_utils.py:26-43 of the reference.
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f

// Sort workspace: the radix sort's histogram area, then the area of the tile-binned sort (tile_sort.h): totals,
// first positions, order and (tile, first, length) records of up to FDGS_TILE_SORT_MAX_TILES tiles, and the
// table[workgroups][tiles] of counts.
#define FDGS_RADIX_WS_WORDS (256u * 1024u + 256u)
#define FDGS_TILE_SORT_MAX_TILES 32768
#define FDGS_TILE_SORT_TABLE_WORDS (2u << 20)
#define FDGS_TILE_SORT_WS_WORDS (7u * FDGS_TILE_SORT_MAX_TILES + FDGS_TILE_SORT_TABLE_WORDS + 64u)

static inline int fdgs_tiles_x(int w) { return (w + FDGS_TILE - 1) / FDGS_TILE; }
static inline int fdgs_tiles_y(int h) { return (h + FDGS_TILE - 1) / FDGS_TILE; }

// Tile rectangle touched by a splat of integer radius r centred at (px,py).
// Clamping happens in float before the conversion so that the result is defined
// for any finite input and identical to clamp((int)f, 0, grid) where that is defined.
__device__ __forceinline__ void fdgs_tile_rect(float px, float py, float r, int gx, int gy,
                                               int& x0, int& y0, int& x1, int& y1) {
#pragma clang fp contract(off)
    float fx0 = (px - r) / (float)FDGS_TILE;
    float fy0 = (py - r) / (float)FDGS_TILE;
    float fx1 = (px + r + (float)(FDGS_TILE - 1)) / (float)FDGS_TILE;
    float fy1 = (py + r + (float)(FDGS_TILE - 1)) / (float)FDGS_TILE;
    fx0 = (fx0 != fx0) ? 0.0f : fminf(fmaxf(fx0, 0.0f), (float)gx);
    fy0 = (fy0 != fy0) ? 0.0f : fminf(fmaxf(fy0, 0.0f), (float)gy);
    fx1 = (fx1 != fx1) ? 0.0f : fminf(fmaxf(fx1, 0.0f), (float)gx);
    fy1 = (fy1 != fy1) ? 0.0f : fminf(fmaxf(fy1, 0.0f), (float)gy);
    x0 = (int)fx0; y0 = (int)fy0; x1 = (int)fx1; y1 = (int)fy1;
}

#define FDGS_SH_TILE_STRIDE 49                       // floats per Gaussian row of the LDS tile: odd, so rows do not share banks

// Copies the workgroup's [rows][48] block between memory and the padded LDS tile with 16-byte accesses
// (a thread's own 48 floats are 192 bytes apart from its neighbour's: direct per-thread accesses would touch 64
// cache lines per instruction).
template <bool TO_LDS>
__device__ __forceinline__ void sh_tile_copy(float* __restrict__ tile, float* __restrict__ mem, int rows) {
    const int n4 = rows * 12;
    for (int j = threadIdx.x; j < n4; j += 256) {
        const int row = j / 12, c4 = j - row * 12;
        float* t = tile + row * FDGS_SH_TILE_STRIDE + 4 * c4;
        float4* m = reinterpret_cast<float4*>(mem) + j;
        if (TO_LDS) {
            const float4 v = *m;
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        } else {
            *m = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
}

// The blend backward accumulates into dL_dgeom [N,12] in the scaling its inner loop works in (the conic is staged
// multiplied by -log2(e)); preprocess_bwd multiplies by these constants when it reads the record:
//   [0],[1]  sum g (a2 dx + b1 dy), sum g (c2 dy + b1 dx)  = log2(e) dL/d(x_pix, y_pix)
//   [2],[3],[4]  sum g dx^2, sum g dx dy, sum g dy^2        = -2 dL/dA, -dL/dB, -2 dL/dC
//   [5] dL/d(opacity)   [6],[7],[8] dL/d(r, g, b)           (g = dL/d(exponent) of the pixel-Gaussian pair)
#define FDGS_DGEOM_XY 0.6931471805599453f
#define FDGS_DGEOM_AC (-0.5f)
#define FDGS_DGEOM_B (-1.0f)

// Slots 10 and 11 of the packed per-Gaussian record.
//   [10]: radius in bits 0-27, SH clamp mask in bits 28-30.
//   [11]: half-widths (x in the low, y in the high 16 bits, fp16 rounded up) of the axis-aligned box around the
//         region where the splat's alpha can reach 1/255: the blend kernels' cull needs only four compares per list
//         entry instead of a logarithm, a division and two square roots.  -inf: alpha never reaches 1/255
//         anywhere; +inf: never skip.
#define FDGS_RADIUS_MASK 0x0fffffff
__device__ __forceinline__ int fdgs_pack_radius(int radius, int clampmask) {
    return (radius & FDGS_RADIUS_MASK) | (clampmask << 28);
}
__device__ __forceinline__ int fdgs_unpack_radius(int bits) { return bits & FDGS_RADIUS_MASK; }
__device__ __forceinline__ int fdgs_unpack_clampmask(int bits) { return (bits >> 28) & 7; }

// alpha = o exp(power) >= 1/255  <=>  A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o); the box of that ellipse,
// enlarged by a margin that covers float rounding in the per-pixel evaluation and the fp16 storage.
__device__ __forceinline__ uint32_t fdgs_pack_reach(float A, float B, float C, float o) {
    const float inf = __builtin_inff();
    float hx, hy;
    const float tau = __logf(255.0f * o);
    if (!(tau >= 0.0f)) {                        // also o <= 0 and NaN
        hx = hy = -inf;
    } else {
        const float s = 2.0f * tau * 1.001f + 0.001f;
        const float det = A * C - B * B;
        if (!(det > 0.0f)) {                     // degenerate conic: never skip
            hx = hy = inf;
        } else {
            const float inv = s / det;
            hx = sqrtf(inv * C) + 0.01f;
            hy = sqrtf(inv * A) + 0.01f;
            if (!(hx == hx)) hx = inf;
            if (!(hy == hy)) hy = inf;
        }
    }
    const uint32_t lo = __half_as_ushort(__float2half_ru(hx)), hi = __half_as_ushort(__float2half_ru(hy));
    return lo | (hi << 16);
}
__device__ __forceinline__ bool fdgs_reaches(float gx, float gy, uint32_t reach, float rx0, float ry0, float rx1,
                                             float ry1) {
    const float hx = __half2float(__ushort_as_half((unsigned short)(reach & 0xffffu)));
    const float hy = __half2float(__ushort_as_half((unsigned short)(reach >> 16)));
    return (gx + hx >= rx0) && (gx - hx <= rx1) && (gy + hy >= ry0) && (gy - hy <= ry1);
}

// Four consecutive SH floats [row, row + 4) of a point, row a multiple of 4.  `base` is the point's row in one
// [N,16,3] array (dc == nullptr), or (rest + 45 p - 3) when the coefficients come as the two tensors the reference's
// GaussianModel keeps (src/model/gaussian_model.py:64-65,86-89: dc [N,1,3] and rest [N,15,3], whose concatenation
// along the coefficient axis is the [N,16,3] array): floats 3-47 of the row are then rest's 45 floats, read with
// 4-byte alignment, and floats 0-2 come from dc + 3 p.
struct __attribute__((packed, aligned(4))) fdgs_f4u { float x, y, z, w; };
__device__ __forceinline__ float4 fdgs_load_sh4(const float* __restrict__ base, const float* __restrict__ dc, int row) {
    if (dc != nullptr && row == 0) return make_float4(dc[0], dc[1], dc[2], base[3]);
    const fdgs_f4u v = *reinterpret_cast<const fdgs_f4u*>(base + row);
    return make_float4(v.x, v.y, v.z, v.w);
}
