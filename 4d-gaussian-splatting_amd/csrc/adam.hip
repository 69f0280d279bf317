// Multi-tensor Adam step in one launch (the optimizer of the reference's training loop:
// torch.optim.Adam(groups, lr=0.0, eps=1e-15), src/runner/fdgs_trainer.py:262-274, stepped at :204).
// Memory-bound: 16 bytes read and 12 written per parameter; every tensor is walked with 16-byte
// accesses when its four pointers allow it.
#include "fdgs_common.h"

struct AdamK {
    fdgs_adam_args a;
    int block_start[FDGS_ADAM_MAX_TENSORS + 1];
    float step_size[FDGS_ADAM_MAX_TENSORS];      // lr / (1 - beta1^t), t = the tensor's own step count
    float inv_sqrt_bc2[FDGS_ADAM_MAX_TENSORS];   // 1 / sqrt(1 - beta2^t)
    float omb1, omb2;                            // 1 - beta, formed in double as the reference's Python does
};

#define ADAM_PER_BLOCK 2048                      // elements per workgroup (256 threads x 8)

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2,
                                         float eps, float step_size, float inv_sqrt_bc2) {
    m = m + omb1 * (g - m);
    v = b2 * v + omb2 * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= step_size * m / denom;
}

__global__ void __launch_bounds__(256)
adam_kernel(AdamK k) {
    int t = 0;
#pragma unroll 1
    for (int i = 1; i < k.a.num_tensors; ++i)
        if ((int)blockIdx.x >= k.block_start[i]) t = i;
    float* __restrict__ p = k.a.param[t];
    const float* __restrict__ g = k.a.grad[t];
    float* __restrict__ m = k.a.exp_avg[t];
    float* __restrict__ v = k.a.exp_avg_sq[t];
    const long long n = k.a.numel[t];
    const float o1 = k.omb1, b2 = (float)k.a.beta2, o2 = k.omb2, eps = (float)k.a.eps, ss = k.step_size[t], ib = k.inv_sqrt_bc2[t];
    const long long base = (long long)((int)blockIdx.x - k.block_start[t]) * ADAM_PER_BLOCK;
    const int L = k.a.grad_row_len[t];
    if (L > 0) {
        // the gradient is a column slice of a wider array (row length L, row stride S): read element by element;
        // parameter and moments stay contiguous (16-byte accesses when aligned)
        const long long S = k.a.grad_row_stride[t];
        const bool vec3 = ((((size_t)p | (size_t)m | (size_t)v) & 15) == 0) && base + ADAM_PER_BLOCK <= n;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long i = base + 4 * (threadIdx.x + 256 * u);
            if (i >= n) break;
            long long row = i / L;
            int col = (int)(i - row * L);
            float gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gv[j] = (i + j < n) ? g[row * S + col] : 0.0f;
                if (++col == L) { col = 0; ++row; }
            }
            if (vec3) {
                float4 pp = *reinterpret_cast<float4*>(p + i);
                float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
                adam_one(pp.x, gv[0], mm.x, vv.x, o1, b2, o2, eps, ss, ib);
                adam_one(pp.y, gv[1], mm.y, vv.y, o1, b2, o2, eps, ss, ib);
                adam_one(pp.z, gv[2], mm.z, vv.z, o1, b2, o2, eps, ss, ib);
                adam_one(pp.w, gv[3], mm.w, vv.w, o1, b2, o2, eps, ss, ib);
                *reinterpret_cast<float4*>(p + i) = pp;
                *reinterpret_cast<float4*>(m + i) = mm;
                *reinterpret_cast<float4*>(v + i) = vv;
            } else {
                for (int j = 0; j < 4 && i + j < n; ++j) {
                    float pp = p[i + j], mm = m[i + j], vv = v[i + j];
                    adam_one(pp, gv[j], mm, vv, o1, b2, o2, eps, ss, ib);
                    p[i + j] = pp; m[i + j] = mm; v[i + j] = vv;
                }
            }
        }
        return;
    }
    const bool vec = ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0);
    if (vec && base + ADAM_PER_BLOCK <= n) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long i = base + 4 * (threadIdx.x + 256 * u);
            float4 pp = *reinterpret_cast<float4*>(p + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, o1, b2, o2, eps, ss, ib);
            adam_one(pp.y, gg.y, mm.y, vv.y, o1, b2, o2, eps, ss, ib);
            adam_one(pp.z, gg.z, mm.z, vv.z, o1, b2, o2, eps, ss, ib);
            adam_one(pp.w, gg.w, mm.w, vv.w, o1, b2, o2, eps, ss, ib);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
    } else {
        for (long long i = base + threadIdx.x; i < base + ADAM_PER_BLOCK && i < n; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam_one(pp, g[i], mm, vv, o1, b2, o2, eps, ss, ib);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

extern "C" int fdgs_adam_step(const fdgs_adam_args* a, void* stream) {
    if (!a || a->num_tensors < 0 || a->num_tensors > FDGS_ADAM_MAX_TENSORS) return FDGS_EINVAL;
    if (!(a->beta1 >= 0.0 && a->beta1 < 1.0 && a->beta2 >= 0.0 && a->beta2 < 1.0)) return FDGS_EINVAL;
    AdamK k;
    k.a = *a;
    k.omb1 = (float)(1.0 - a->beta1);
    k.omb2 = (float)(1.0 - a->beta2);
    long long blocks = 0;
    for (int t = 0; t < a->num_tensors; ++t) {
        if (a->numel[t] < 0) return FDGS_EINVAL;
        if (a->numel[t] > 0 && (!a->param[t] || !a->grad[t] || !a->exp_avg[t] || !a->exp_avg_sq[t])) return FDGS_EINVAL;
        if (a->step[t] < 1 || a->grad_row_len[t] < 0) return FDGS_EINVAL;
        if (a->grad_row_len[t] > 0 && a->grad_row_stride[t] < a->grad_row_len[t]) return FDGS_EINVAL;
        const double bc1 = 1.0 - pow(a->beta1, (double)a->step[t]);
        const double bc2 = 1.0 - pow(a->beta2, (double)a->step[t]);
        k.block_start[t] = (int)blocks;
        k.step_size[t] = (float)((double)a->lr[t] / bc1);
        k.inv_sqrt_bc2[t] = (float)(1.0 / sqrt(bc2));
        blocks += (a->numel[t] + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK;
        if (blocks > 0x7fffffffLL) return FDGS_EINVAL;
    }
    k.block_start[a->num_tensors] = (int)blocks;
    if (blocks == 0) return FDGS_OK;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
