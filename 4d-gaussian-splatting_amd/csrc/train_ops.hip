// The two element-wise tails of the reference's training iteration, each as one pass over its data:
//   * the L1 image loss mean|a - b| (src/model/losses.py:7-8, used at src/runner/fdgs_trainer.py:495) and
//     its gradient;
//   * the densification statistics: max_radii2D, xyz_gradient_accum and denom updated for the visible
//     Gaussians (src/runner/fdgs_trainer.py:181-182, src/model/gaussian_model.py:182-184).
// Memory-bound; in PyTorch they are about thirty small launches per iteration.
#include "fdgs_common.h"

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
    float s = 0.0f;
    const bool vec = ((((size_t)a.a | (size_t)a.b) & 15) == 0) && base + per_block <= a.n;
    if (vec) {
#pragma unroll
        for (int u = 0; u < L1_PER_THREAD / 4; ++u) {
            const long long i = base + 4 * (threadIdx.x + (long long)L1_THREADS * u);
            const float4 x = *reinterpret_cast<const float4*>(a.a + i), y = *reinterpret_cast<const float4*>(a.b + i);
            s += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
        }
    } else {
        for (long long i = base + threadIdx.x; i < base + per_block && i < a.n; i += L1_THREADS)
            s += fabsf(a.a[i] - a.b[i]);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
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
    if (threadIdx.x == 0) a.loss[0] = (float)(shd[0] / (double)a.n * (double)a.scale);
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
