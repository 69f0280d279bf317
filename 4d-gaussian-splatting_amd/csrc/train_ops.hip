// The two element-wise tails of the reference's training iteration, each as one pass over its data:
//   * the L1 image loss mean|a - b| (src/model/losses.py:7-8, used at src/runner/fdgs_trainer.py:495) and
//     its gradient;
//   * the densification statistics: max_radii2D, xyz_gradient_accum and denom updated for the visible
//     Gaussians (src/runner/fdgs_trainer.py:181-182, src/model/gaussian_model.py:182-184).
// Memory-bound; in PyTorch they are about thirty small launches per iteration.
#include "fdgs_common.h"
#include <mutex>

#define L1_THREADS 256
#define L1_PER_THREAD 16                      // floats per thread per input (4 x float4)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {   // result valid in thread 0
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void __launch_bounds__(L1_THREADS)
l1_partial_kernel(fdgs_l1_fwd_args a) {
    __shared__ float sh[4];
    const long long per_block = (long long)L1_THREADS * L1_PER_THREAD;
    const long long base = (long long)blockIdx.x * per_block;
    float s = 0.0f, q = 0.0f;
    // optional gradient for an upstream gradient of 1, the expression of l1_bwd_kernel
    const float g = (float)((double)a.scale / (double)a.n);
    auto sgn = [g](float d) { return d > 0.0f ? g : (d < 0.0f ? -g : 0.0f * d); };
    const bool vec = ((((size_t)a.a | (size_t)a.b | (size_t)a.grad_a) & 15) == 0) && base + per_block <= a.n;
    if (vec) {
#pragma unroll
        for (int u = 0; u < L1_PER_THREAD / 4; ++u) {
            const long long i = base + 4 * (threadIdx.x + (long long)L1_THREADS * u);
            const float4 x = *reinterpret_cast<const float4*>(a.a + i), y = *reinterpret_cast<const float4*>(a.b + i);
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            if (a.grad_a) *reinterpret_cast<float4*>(a.grad_a + i) = make_float4(sgn(d0), sgn(d1), sgn(d2), sgn(d3));
        }
    } else {
        for (long long i = base + threadIdx.x; i < base + per_block && i < a.n; i += L1_THREADS) {
            const float d = a.a[i] - a.b[i];
            s += fabsf(d);
            q += d * d;
            if (a.grad_a) a.grad_a[i] = sgn(d);
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
    if (a.psnr) {                                  // uniform branch: the squared differences ride along
        __syncthreads();
        q = block_sum_256(q, sh);
        if (threadIdx.x == 0) a.partial[gridDim.x + blockIdx.x] = q;
    }
}

__global__ void __launch_bounds__(L1_THREADS)
l1_final_kernel(fdgs_l1_fwd_args a, int num_partials) {
    double s = 0.0;                               // a few hundred partials: the double adds cost nothing
    for (int i = threadIdx.x; i < num_partials; i += L1_THREADS) s += (double)a.partial[i];
    __shared__ double shd[L1_THREADS];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = L1_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l1 = (float)(shd[0] / (double)a.n * (double)a.scale);
        a.loss[0] = a.add ? l1 + a.add[0] : l1;
    }
    if (a.psnr) {
        __syncthreads();
        double q = 0.0;
        for (int i = threadIdx.x; i < num_partials; i += L1_THREADS) q += (double)a.partial[num_partials + i];
        shd[threadIdx.x] = q;
        __syncthreads();
        for (int o = L1_THREADS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float mse = (float)(shd[0] / (double)a.n);
            a.psnr[0] = 20.0f * log10f(1.0f / sqrtf(mse));
        }
    }
}

extern "C" size_t fdgs_l1_loss_partials(int64_t n) {
    if (n <= 0) return 1;
    return (size_t)((n + (int64_t)L1_THREADS * L1_PER_THREAD - 1) / ((int64_t)L1_THREADS * L1_PER_THREAD));
}

extern "C" int fdgs_l1_loss_fwd(const fdgs_l1_fwd_args* a, void* stream) {
    if (!a || a->n <= 0 || !a->a || !a->b || !a->loss || !a->partial) return FDGS_EINVAL;
    const size_t blocks = fdgs_l1_loss_partials(a->n);
    if (blocks > 0x7fffffffu) return FDGS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)blocks), dim3(L1_THREADS), 0, s, *a);
    FDGS_CHECK_LAUNCH();
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(L1_THREADS), 0, s, *a, (int)blocks);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

// ---- the same two launches with the partial sums in a scratch array built into the library (no allocation) ----
// One copy of the array per GPU that runs the library; `scratch` itself is only a host-side handle of the
// symbol: its device address is looked up once per device and that pointer is what the kernels receive.
static __device__ __align__(256) float scratch[2 * FDGS_L1_STATIC_MAX_PARTIALS];

#define L1_MAX_DEVICES 64
static std::mutex scratch_mu;
static float* scratch_ptr[L1_MAX_DEVICES];        // device address of `scratch` on each device, 0 until looked up

// The array that belongs to the CURRENT device: hipGetSymbolAddress answers for that one.  A stream of another
// device is refused (its kernels would receive the current device's address).
static int scratch_address(hipStream_t s, float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= L1_MAX_DEVICES) return FDGS_ELAUNCH;
    hipDevice_t sdev = -1;
    if (s && hipStreamGetDevice(s, &sdev) == hipSuccess && (int)sdev != dev) return FDGS_EINVAL;
    std::lock_guard<std::mutex> lock(scratch_mu);
    if (!scratch_ptr[dev]) {
        float* p = nullptr;
        if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return FDGS_ELAUNCH;
        scratch_ptr[dev] = p;
    }
    *out = scratch_ptr[dev];
    return FDGS_OK;
}

extern "C" int fdgs_l1_loss_fwd_static(const fdgs_l1_fwd_args* a, void* stream) {
    if (!a || a->n <= 0 || !a->a || !a->b || !a->loss) return FDGS_EINVAL;
    if (a->n > FDGS_L1_STATIC_MAX_N) return FDGS_ETOOLARGE;
    static_assert((long long)L1_THREADS * L1_PER_THREAD == 4096, "FDGS_L1_STATIC_MAX_N assumes 4096 floats per partial");
    const size_t blocks = fdgs_l1_loss_partials(a->n);          // <= FDGS_L1_STATIC_MAX_PARTIALS by the check above
    if (blocks > FDGS_L1_STATIC_MAX_PARTIALS) return FDGS_ETOOLARGE;
    fdgs_l1_fwd_args k = *a;
    hipStream_t s = (hipStream_t)stream;
    int rc = scratch_address(s, &k.partial);
    if (rc != FDGS_OK) return rc;
    hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)blocks), dim3(L1_THREADS), 0, s, k);
    FDGS_CHECK_LAUNCH();
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(L1_THREADS), 0, s, k, (int)blocks);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

__global__ void __launch_bounds__(L1_THREADS)
l1_bwd_kernel(fdgs_l1_bwd_args a) {
    const float g = a.grad_loss[0] * (float)((double)a.scale / (double)a.n);
    const long long per_block = (long long)L1_THREADS * L1_PER_THREAD;
    const long long base = (long long)blockIdx.x * per_block;
    auto sgn = [g](float d) { return d > 0.0f ? g : (d < 0.0f ? -g : 0.0f * d); };   // NaN stays NaN, as torch.sgn
    const bool vec = ((((size_t)a.a | (size_t)a.b | (size_t)a.grad_a) & 15) == 0) && base + per_block <= a.n;
    if (vec) {
#pragma unroll
        for (int u = 0; u < L1_PER_THREAD / 4; ++u) {
            const long long i = base + 4 * (threadIdx.x + (long long)L1_THREADS * u);
            const float4 x = *reinterpret_cast<const float4*>(a.a + i), y = *reinterpret_cast<const float4*>(a.b + i);
            *reinterpret_cast<float4*>(a.grad_a + i) = make_float4(sgn(x.x - y.x), sgn(x.y - y.y), sgn(x.z - y.z), sgn(x.w - y.w));
        }
    } else {
        for (long long i = base + threadIdx.x; i < base + per_block && i < a.n; i += L1_THREADS)
            a.grad_a[i] = sgn(a.a[i] - a.b[i]);
    }
}

extern "C" int fdgs_l1_loss_bwd(const fdgs_l1_bwd_args* a, void* stream) {
    if (!a || a->n <= 0 || !a->a || !a->b || !a->grad_loss || !a->grad_a) return FDGS_EINVAL;
    const size_t blocks = fdgs_l1_loss_partials(a->n);
    if (blocks > 0x7fffffffu) return FDGS_EINVAL;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3((unsigned)blocks), dim3(L1_THREADS), 0, (hipStream_t)stream, *a);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

__global__ void __launch_bounds__(256)
densify_stats_kernel(fdgs_densify_stats_args a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.num_points) return;
    const int r = a.radii[i];
    if (r <= 0) return;                           // visibility_filter = radii > 0
    const float rf = (float)r;
    if (a.max_radii2D) a.max_radii2D[i] = fmaxf(a.max_radii2D[i], rf);
    const float gx = a.viewspace_grad[3 * (size_t)i], gy = a.viewspace_grad[3 * (size_t)i + 1];
    a.xyz_gradient_accum[i] += sqrtf(gx * gx + gy * gy);
    a.denom[i] += 1.0f;
}

extern "C" int fdgs_densify_stats(const fdgs_densify_stats_args* a, void* stream) {
    if (!a || a->num_points < 0) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->radii || !a->viewspace_grad || !a->xyz_gradient_accum || !a->denom) return FDGS_EINVAL;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((a->num_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

// ---------------------------------------------------------------------------
// Grid regulariser of the deformation field (src/model/deformation_wrapper.py:52-92 with
// src/utils/plane_utils.py:3-9): for every plane [C][h][w] the mean squared second difference along h,
// weighted, plus for the time planes the weighted mean |1 - t|.  One thread per element computes its loss
// terms and, if asked, the complete gradient of that element (the second difference is a 3-tap stencil,
// its adjoint a 5-tap one: five loads along h).  In PyTorch this is about 100 launches forward and 150
// backward per iteration.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
plane_reg_kernel(fdgs_plane_reg_args a, int blocks_per_plane) {
    __shared__ float sh[4];
    const int pi = blockIdx.x / blocks_per_plane;
    const int bx = blockIdx.x - pi * blocks_per_plane;
    // uniform selection without dynamic indexing of the argument arrays
    const float* __restrict__ t = a.plane[0]; float* g = a.grad[0];
    int h = a.h[0], w = a.w[0]; float ws = a.w_smooth[0], wl = a.w_l1[0];
#pragma unroll
    for (int i = 1; i < FDGS_REG_MAX_PLANES; ++i)
        if (pi == i) { t = a.plane[i]; g = a.grad[i]; h = a.h[i]; w = a.w[i]; ws = a.w_smooth[i]; wl = a.w_l1[i]; }
    const int hw = h * w, total = a.channels * hw;
    const float ms = h > 2 ? ws / ((float)a.channels * (float)(h - 2) * (float)w) : 0.0f;   // weight / count of the mean
    const float ml = wl / (float)total;
    const float gs = a.grad_scale ? a.grad_scale[0] : 1.0f;
    float loss = 0.0f;
    for (int idx = bx * 256 + threadIdx.x; idx < total; idx += blocks_per_plane * 256) {
        const int c = idx / hw, r = idx - c * hw;
        const int y = r / w;
        const float* col = t + idx;                       // element (c, y, x); neighbours along h are w apart
        const float t0 = col[0];
        const float tm2 = y >= 2 ? col[-2 * w] : 0.0f, tm1 = y >= 1 ? col[-w] : 0.0f;
        const float tp1 = y + 1 < h ? col[w] : 0.0f, tp2 = y + 2 < h ? col[2 * w] : 0.0f;
        // second differences that contain this element: d2(i) = (t[i+2] - t[i+1]) - (t[i+1] - t[i]), 0 <= i <= h-3
        const float d_y = (y + 2 < h) ? (tp2 - tp1) - (tp1 - t0) : 0.0f;              // i = y
        const float d_y1 = (y >= 1 && y + 1 < h) ? (tp1 - t0) - (t0 - tm1) : 0.0f;    // i = y - 1
        const float d_y2 = (y >= 2) ? (t0 - tm1) - (tm1 - tm2) : 0.0f;                // i = y - 2
        const float one_minus = 1.0f - t0;
        loss += ms * (d_y * d_y) + ml * fabsf(one_minus);
        if (g) {
            const float gsm = 2.0f * (d_y - 2.0f * d_y1 + d_y2);
            const float gl1 = one_minus > 0.0f ? -1.0f : (one_minus < 0.0f ? 1.0f : 0.0f);
            const float v = gs * (ms * gsm + ml * gl1);
            if (a.accumulate) g[idx] += v; else g[idx] = v;
        }
    }
    if (a.loss_partial) {
        loss = block_sum_256(loss, sh);
        if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = loss;
    }
}

__global__ void __launch_bounds__(L1_THREADS)
plane_reg_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ double shd[L1_THREADS];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += L1_THREADS) s += (double)partial[i];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = L1_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)shd[0];
}

#define REG_BLOCKS_PER_PLANE 256

extern "C" size_t fdgs_plane_reg_partials(int32_t num_planes) {
    return (size_t)(num_planes > 0 ? num_planes : 1) * REG_BLOCKS_PER_PLANE;
}

extern "C" int fdgs_plane_regulation(const fdgs_plane_reg_args* a, void* stream) {
    if (!a || a->num_planes < 1 || a->num_planes > FDGS_REG_MAX_PLANES || a->channels < 1) return FDGS_EINVAL;
    if ((a->loss == nullptr) != (a->loss_partial == nullptr)) return FDGS_EINVAL;
    bool any_grad = false;
    for (int i = 0; i < a->num_planes; ++i) {
        if (!a->plane[i] || a->h[i] < 1 || a->w[i] < 1) return FDGS_EINVAL;
        if ((long long)a->channels * a->h[i] * a->w[i] > 0x7fffffffLL) return FDGS_EINVAL;
        any_grad = any_grad || a->grad[i];
    }
    if (!a->loss && !any_grad) return FDGS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(plane_reg_kernel, dim3(a->num_planes * REG_BLOCKS_PER_PLANE), dim3(256), 0, s, *a,
                       REG_BLOCKS_PER_PLANE);
    FDGS_CHECK_LAUNCH();
    if (a->loss) {
        hipLaunchKernelGGL(plane_reg_final_kernel, dim3(1), dim3(L1_THREADS), 0, s, a->loss_partial,
                           a->num_planes * REG_BLOCKS_PER_PLANE, a->loss);
        FDGS_CHECK_LAUNCH();
    }
    return FDGS_OK;
}

// ---------------------------------------------------------------------------
// 8-bit frames (render.py:47,78 of the reference: to8b + transpose to H, W, C).  One thread per pixel: C coalesced
// plane reads, C consecutive bytes written.  HBM-bound: 4 C bytes read, C written per pixel.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
image_to8b_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int C, int64_t plane) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= plane) return;
    for (int c = 0; c < C; ++c) {
        const float v = fminf(fmaxf(img[c * plane + p], 0.0f), 1.0f);      // a NaN gives 0
        out[p * C + c] = (uint8_t)(int)(255.0f * v);
    }
}

extern "C" int fdgs_image_to8b(const float* image_chw, uint8_t* out_hwc, int32_t channels, int32_t height,
                               int32_t width, void* stream) {
    if (channels <= 0 || height <= 0 || width <= 0 || !image_chw || !out_hwc) return FDGS_EINVAL;
    const int64_t plane = (int64_t)height * width;
    hipLaunchKernelGGL(image_to8b_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       image_chw, out_hwc, channels, plane);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
