// SSIM between two image stacks and its gradient (src/model/losses.py:13-53 of the reference: five grouped
// 11 x 11 convolutions with a Gaussian window, zero padding, then the SSIM map and its mean).
// The window is an outer product, so each convolution is two 1-D passes.  One workgroup owns a 32 x 32 tile
// of one plane: the tile and its halo are staged in LDS once, the horizontal pass of all five moments goes
// back to LDS, the vertical pass ends in registers where the map is evaluated.  With `dmaps` the forward also
// stores the three partial derivatives the backward needs (d map / d mu1, d map / d E[x^2], d map / d E[xy]);
// the backward convolves those three maps the same way and combines them with the pixel values:
//   d map_sum / d x(q) = conv(A)(q) + 2 x(q) conv(B)(q) + y(q) conv(C)(q)       (the window is symmetric)
// Traffic: forward reads two images and writes three maps, backward reads five and writes one.
#include "fdgs_common.h"

#define SS_TILE 32
#define SS_MAXR FDGS_SSIM_MAX_RADIUS
#define SS_IN (SS_TILE + 2 * SS_MAXR)            // 46: widest staged tile edge

struct SsimK {
    int planes, h, w, radius;
    float win[2 * SS_MAXR + 1];
    float c1, c2;
    const float* x; const float* y;
    float* partial;                    // [planes][tiles]
    float* dmaps;                      // [3][planes][h][w] or NULL
    const float* plane_scale;          // backward: [planes]
    float* grad;                       // backward: [planes][h][w]
};

__device__ __forceinline__ float ss_block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// Q maps staged with halo, horizontal pass into hbuf[q][row][col], vertical pass returns out[q][k] for the
// thread's four pixels (col = tid % 32, row = tid / 32 + 8 k).
template <int Q, typename Load>
__device__ __forceinline__ void separable(const SsimK& a, float* tile, float* hbuf, Load load, float out[Q][4]) {
    const int R = a.radius, edge = SS_TILE + 2 * R;
    const int x0 = blockIdx.x * SS_TILE - R, y0 = blockIdx.y * SS_TILE - R;
    // stage: tile[s][r][c], s < S source maps (S = 2 for the forward's x, y; 3 for the backward's A, B, C)
    constexpr int S = (Q == 5) ? 2 : 3;
    for (int i = threadIdx.x; i < edge * edge; i += 256) {
        const int r = i / edge, c = i - r * edge;
        const int gy = y0 + r, gx = x0 + c;
        const bool in = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
#pragma unroll
        for (int s = 0; s < S; ++s) tile[(s * SS_IN + r) * SS_IN + c] = in ? load(s, gy, gx) : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < edge * SS_TILE; i += 256) {
        const int r = i / SS_TILE, c = i - r * SS_TILE;
        float acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = 0.0f;
        for (int t = 0; t <= 2 * R; ++t) {
            const float wt = a.win[t];
            if (Q == 5) {
                const float xv = tile[(0 * SS_IN + r) * SS_IN + c + t], yv = tile[(1 * SS_IN + r) * SS_IN + c + t];
                acc[0] += wt * xv; acc[1] += wt * yv; acc[2] += wt * (xv * xv); acc[3] += wt * (yv * yv);
                acc[4] += wt * (xv * yv);
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] += wt * tile[(q * SS_IN + r) * SS_IN + c + t];
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) hbuf[(q * SS_IN + r) * SS_TILE + c] = acc[q];
    }
    __syncthreads();
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k;
#pragma unroll
        for (int q = 0; q < Q; ++q) out[q][k] = 0.0f;
        for (int t = 0; t <= 2 * R; ++t) {
            const float wt = a.win[t];
#pragma unroll
            for (int q = 0; q < Q; ++q) out[q][k] += wt * hbuf[(q * SS_IN + r + t) * SS_TILE + c];
        }
    }
}

__global__ void __launch_bounds__(256)
ssim_fwd_kernel(SsimK a) {
    __shared__ float tile[2 * SS_IN * SS_IN];
    __shared__ float hbuf[5 * SS_IN * SS_TILE];
    __shared__ float sh[4];
    const int plane = blockIdx.z;
    const size_t pbase = (size_t)plane * a.h * a.w;
    const float* __restrict__ X = a.x + pbase;
    const float* __restrict__ Y = a.y + pbase;
    float m[5][4];
    separable<5>(a, tile, hbuf, [&](int s, int gy, int gx) { return (s ? Y : X)[(size_t)gy * a.w + gx]; }, m);
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const int gx = blockIdx.x * SS_TILE + c;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gy = blockIdx.y * SS_TILE + r0 + 8 * k;
        if (gx >= a.w || gy >= a.h) continue;
        const float m1 = m[0][k], m2 = m[1][k];
        const float s1 = m[2][k] - m1 * m1, s2 = m[3][k] - m2 * m2, s12 = m[4][k] - m1 * m2;
        const float n1 = 2.0f * (m1 * m2) + a.c1, n2 = 2.0f * s12 + a.c2;
        const float d1 = m1 * m1 + m2 * m2 + a.c1, d2 = s1 + s2 + a.c2;
        const float inv = 1.0f / (d1 * d2);
        const float f = n1 * n2 * inv;
        sum += f;
        if (a.dmaps) {
            const size_t map = (size_t)a.planes * a.h * a.w;
            const size_t o = pbase + (size_t)gy * a.w + gx;
            a.dmaps[o] = 2.0f * m2 * (n2 - n1) * inv - 2.0f * m1 * f / d1 + 2.0f * m1 * f / d2;
            a.dmaps[map + o] = -f / d2;
            a.dmaps[2 * map + o] = 2.0f * n1 * inv;
        }
    }
    sum = ss_block_sum(sum, sh);
    if (threadIdx.x == 0)
        a.partial[(size_t)plane * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] = sum;
}

__global__ void __launch_bounds__(256)
ssim_plane_sum_kernel(const float* __restrict__ partial, int tiles, float* __restrict__ plane_sum) {
    __shared__ double shd[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < tiles; i += 256) s += (double)partial[(size_t)blockIdx.x * tiles + i];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) plane_sum[blockIdx.x] = (float)shd[0];
}

__global__ void __launch_bounds__(256)
ssim_bwd_kernel(SsimK a) {
    __shared__ float tile[3 * SS_IN * SS_IN];
    __shared__ float hbuf[3 * SS_IN * SS_TILE];
    const int plane = blockIdx.z;
    const size_t pbase = (size_t)plane * a.h * a.w;
    const size_t map = (size_t)a.planes * a.h * a.w;
    const float* __restrict__ D = a.dmaps + pbase;
    float cv[3][4];
    separable<3>(a, tile, hbuf, [&](int s, int gy, int gx) { return D[s * map + (size_t)gy * a.w + gx]; }, cv);
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const int gx = blockIdx.x * SS_TILE + c;
    const float g = a.plane_scale[plane];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gy = blockIdx.y * SS_TILE + r0 + 8 * k;
        if (gx >= a.w || gy >= a.h) continue;
        const size_t o = pbase + (size_t)gy * a.w + gx;
        a.grad[o] = g * (cv[0][k] + 2.0f * a.x[o] * cv[1][k] + a.y[o] * cv[2][k]);
    }
}

extern "C" size_t fdgs_ssim_tiles(int32_t height, int32_t width) {
    if (height <= 0 || width <= 0) return 0;
    return (size_t)((height + SS_TILE - 1) / SS_TILE) * (size_t)((width + SS_TILE - 1) / SS_TILE);
}

static bool ssim_shape_ok(int planes, int h, int w, int radius) {
    return planes > 0 && h > 0 && w > 0 && radius >= 0 && radius <= SS_MAXR && planes <= 65535 &&
           (h + SS_TILE - 1) / SS_TILE <= 65535;
}

extern "C" int fdgs_ssim_fwd(const fdgs_ssim_fwd_args* a, void* stream) {
    if (!a || !ssim_shape_ok(a->planes, a->height, a->width, a->radius)) return FDGS_EINVAL;
    if (!a->img1 || !a->img2 || !a->plane_sum || !a->partial) return FDGS_EINVAL;
    SsimK k;
    k.planes = a->planes; k.h = a->height; k.w = a->width; k.radius = a->radius;
    for (int i = 0; i < 2 * SS_MAXR + 1; ++i) k.win[i] = a->window[i];
    k.c1 = a->c1; k.c2 = a->c2; k.x = a->img1; k.y = a->img2; k.partial = a->partial; k.dmaps = a->dmaps;
    k.plane_scale = nullptr; k.grad = nullptr;
    const dim3 grid((a->width + SS_TILE - 1) / SS_TILE, (a->height + SS_TILE - 1) / SS_TILE, a->planes);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, s, k);
    FDGS_CHECK_LAUNCH();
    hipLaunchKernelGGL(ssim_plane_sum_kernel, dim3(a->planes), dim3(256), 0, s, a->partial, (int)(grid.x * grid.y),
                       a->plane_sum);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

extern "C" int fdgs_ssim_bwd(const fdgs_ssim_bwd_args* a, void* stream) {
    if (!a || !ssim_shape_ok(a->planes, a->height, a->width, a->radius)) return FDGS_EINVAL;
    if (!a->img1 || !a->img2 || !a->dmaps || !a->plane_scale || !a->grad_img1) return FDGS_EINVAL;
    SsimK k;
    k.planes = a->planes; k.h = a->height; k.w = a->width; k.radius = a->radius;
    for (int i = 0; i < 2 * SS_MAXR + 1; ++i) k.win[i] = a->window[i];
    k.c1 = 0.0f; k.c2 = 0.0f; k.x = a->img1; k.y = a->img2; k.partial = nullptr;
    k.dmaps = const_cast<float*>(a->dmaps); k.plane_scale = a->plane_scale; k.grad = a->grad_img1;
    const dim3 grid((a->width + SS_TILE - 1) / SS_TILE, (a->height + SS_TILE - 1) / SS_TILE, a->planes);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, k);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
