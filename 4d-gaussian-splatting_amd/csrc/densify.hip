// Densification / pruning surgery on the per-Gaussian arrays (src/runner/fdgs_trainer.py:304-492 of the
// reference, where it is a chain of boolean-mask gathers, torch.cat and per-tensor optimizer-state edits):
//   classify  - one pass over the statistics: keep / clone / split / drop per Gaussian
//   plan      - stable three-way scan of the classes -> src[new row] = old row
//   apply     - every array (six parameters, twelve Adam moments, statistics, deformation table) moved
//               through the map in ONE launch, zeros for the moments of added rows
//   children  - positions and scales of the split children
// All of it is byte traffic: each array is read once and written once.
#include "fdgs_common.h"

#define SG_THREADS 256
#define SG_PER_THREAD 4
#define SG_BLOCK_PTS (SG_THREADS * SG_PER_THREAD)      // 1024 consecutive points per workgroup

__device__ __forceinline__ float max_scale(const float* __restrict__ scaling, int i) {
    const float a = expf(scaling[3 * (size_t)i]), b = expf(scaling[3 * (size_t)i + 1]),
                c = expf(scaling[3 * (size_t)i + 2]);
    return fmaxf(a, fmaxf(b, c));
}

__global__ void __launch_bounds__(SG_THREADS)
densify_classify_kernel(fdgs_densify_classify_args a) {
    const int i = blockIdx.x * SG_THREADS + threadIdx.x;
    if (i >= a.num_points) return;
    float g = a.xyz_gradient_accum[i] / a.denom[i];
    if (g != g) g = 0.0f;
    const bool big = max_scale(a.scaling, i) > a.size_limit;
    uint8_t c = FDGS_CLS_KEEP;
    if (!big && fabsf(g) >= a.grad_threshold) c = FDGS_CLS_CLONE;
    else if (big && g >= a.grad_threshold) c = FDGS_CLS_SPLIT;
    a.cls[i] = c;
}

__global__ void __launch_bounds__(SG_THREADS)
prune_classify_kernel(fdgs_prune_classify_args a) {
    const int i = blockIdx.x * SG_THREADS + threadIdx.x;
    if (i >= a.num_points) return;
    const float o = 1.0f / (1.0f + expf(-a.opacity[i]));
    bool drop = o < a.min_opacity;
    if (a.max_screen_size > 0.0f)
        drop = drop || a.max_radii2D[i] > a.max_screen_size || max_scale(a.scaling, i) > a.world_limit;
    a.cls[i] = drop ? FDGS_CLS_DROP : FDGS_CLS_KEEP;
}

// ---- plan -------------------------------------------------------------------------------------------
// Three counters travel together in one 64-bit word (21 bits each; a workgroup holds 1024 points).
#define SG_SURV(w) ((uint32_t)((w) & 0x1fffffull))
#define SG_CLONE(w) ((uint32_t)(((w) >> 21) & 0x1fffffull))
#define SG_SPLIT(w) ((uint32_t)(((w) >> 42) & 0x1fffffull))

__device__ __forceinline__ unsigned long long class_word(uint32_t c) {
    return (c < FDGS_CLS_SPLIT ? 1ull : 0ull) | (c == FDGS_CLS_CLONE ? (1ull << 21) : 0ull) |
           (c == FDGS_CLS_SPLIT ? (1ull << 42) : 0ull);
}

// The thread's four classes (invalid points read as DROP) and the exclusive prefix of its packed counts
// inside the workgroup; the workgroup total comes back in *total.
__device__ __forceinline__ unsigned long long block_prefix(const uint8_t* __restrict__ cls, int n, uint32_t c[SG_PER_THREAD],
                                                           unsigned long long* total) {
    __shared__ unsigned long long wsum[SG_THREADS / 64];
    const int first = blockIdx.x * SG_BLOCK_PTS + threadIdx.x * SG_PER_THREAD;
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < SG_PER_THREAD; ++k) {
        c[k] = (first + k < n) ? cls[first + k] : (uint32_t)FDGS_CLS_DROP;
        mine += class_word(c[k]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SG_THREADS / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    *total = tot;
    return base + incl - mine;
}

__global__ void __launch_bounds__(SG_THREADS)
surgery_count_kernel(const uint8_t* __restrict__ cls, int n, uint32_t* __restrict__ block_tot) {
    uint32_t c[SG_PER_THREAD];
    unsigned long long tot;
    block_prefix(cls, n, c, &tot);
    if (threadIdx.x == 0) {
        block_tot[3 * blockIdx.x] = SG_SURV(tot);
        block_tot[3 * blockIdx.x + 1] = SG_CLONE(tot);
        block_tot[3 * blockIdx.x + 2] = SG_SPLIT(tot);
    }
}

// One workgroup: exclusive scan of the per-workgroup totals (in place) and the four counts.
__global__ void __launch_bounds__(1024)
surgery_scan_kernel(uint32_t* __restrict__ block_tot, int blocks, int32_t* __restrict__ counts) {
    __shared__ uint32_t wsum[3][16];
    __shared__ uint32_t carry[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 3) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < blocks; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        uint32_t v[3], incl[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            v[q] = b < blocks ? block_tot[3 * b + q] : 0u;
            incl[q] = v[q];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl[q], d);
                if (lane >= d) incl[q] += o;
            }
            if (lane == 63) wsum[q][wave] = incl[q];
        }
        __syncthreads();
        uint32_t tot[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            uint32_t base = carry[q];
            tot[q] = 0;
            for (int w = 0; w < 16; ++w) {
                if (w < wave) base += wsum[q][w];
                tot[q] += wsum[q][w];
            }
            if (b < blocks) block_tot[3 * b + q] = base + incl[q] - v[q];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) carry[q] += tot[q];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t k = carry[0], c = carry[1], s = carry[2];
        counts[0] = (int32_t)(k + c + 2 * s);
        counts[1] = (int32_t)k;
        counts[2] = (int32_t)c;
        counts[3] = (int32_t)s;
    }
}

__global__ void __launch_bounds__(SG_THREADS)
surgery_map_kernel(const uint8_t* __restrict__ cls, int n, const uint32_t* __restrict__ block_base,
                   const int32_t* __restrict__ counts, int32_t* __restrict__ src) {
    uint32_t c[SG_PER_THREAD];
    unsigned long long tot;
    const unsigned long long pre = block_prefix(cls, n, c, &tot);
    uint32_t ks = block_base[3 * blockIdx.x] + SG_SURV(pre);
    uint32_t cs = block_base[3 * blockIdx.x + 1] + SG_CLONE(pre);
    uint32_t ss = block_base[3 * blockIdx.x + 2] + SG_SPLIT(pre);
    const uint32_t K = (uint32_t)counts[1], C = (uint32_t)counts[2], S = (uint32_t)counts[3];
    const int first = blockIdx.x * SG_BLOCK_PTS + threadIdx.x * SG_PER_THREAD;
#pragma unroll
    for (int k = 0; k < SG_PER_THREAD; ++k) {
        const int i = first + k;
        if (c[k] < FDGS_CLS_SPLIT) src[ks++] = i;
        if (c[k] == FDGS_CLS_CLONE) src[K + cs++] = i;
        if (c[k] == FDGS_CLS_SPLIT) {
            src[K + C + ss] = i;
            src[K + C + S + ss] = i;
            ++ss;
        }
    }
}

// ---- apply ------------------------------------------------------------------------------------------
struct SurgeryApplyK {
    int rows_after, rows_carried;
    const int32_t* src;
    const void* in[FDGS_SURGERY_MAX_ARRAYS];
    void* out[FDGS_SURGERY_MAX_ARRAYS];
    int32_t row_units[FDGS_SURGERY_MAX_ARRAYS];     // words, or bytes when the array moves as bytes
    int32_t flags[FDGS_SURGERY_MAX_ARRAYS];         // bit 0: zero new rows, bit 1: move as bytes
};

template <typename T>
__device__ __forceinline__ void move_rows(const SurgeryApplyK& a, int k) {
    const T* __restrict__ in = reinterpret_cast<const T*>(a.in[k]);
    T* __restrict__ out = reinterpret_cast<T*>(a.out[k]);
    const uint32_t ru = (uint32_t)a.row_units[k];
    const bool zero_new = a.flags[k] & FDGS_SURGERY_ZERO_NEW;
    const unsigned long long total = (unsigned long long)a.rows_after * ru;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint32_t row = (uint32_t)(e / ru), w = (uint32_t)(e - (unsigned long long)row * ru);
        T v = 0;
        if (!(zero_new && (int)row >= a.rows_carried)) v = in[(size_t)a.src[row] * ru + w];
        out[e] = v;
    }
}

__global__ void __launch_bounds__(256)
surgery_apply_kernel(SurgeryApplyK a) {
    const int k = blockIdx.y;
    if (a.flags[k] & 2) move_rows<uint8_t>(a, k);
    else move_rows<uint32_t>(a, k);
}

__global__ void __launch_bounds__(256)
split_children_kernel(fdgs_split_children_args a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.num_children) return;
    const size_t row = (size_t)a.first_child_row + j;
    const size_t i = (size_t)a.src[row];
    const float sx = expf(a.scaling[3 * i]), sy = expf(a.scaling[3 * i + 1]), sz = expf(a.scaling[3 * i + 2]);
    const float vx = a.noise[3 * (size_t)j] * sx, vy = a.noise[3 * (size_t)j + 1] * sy, vz = a.noise[3 * (size_t)j + 2] * sz;
    // src/utils/general_utils.py:101-124: normalise, then the rotation matrix of (r, x, y, z)
    float r = a.rotation[4 * i], x = a.rotation[4 * i + 1], y = a.rotation[4 * i + 2], z = a.rotation[4 * i + 3];
    const float norm = sqrtf(r * r + x * x + y * y + z * z);
    r /= norm; x /= norm; y /= norm; z /= norm;
    const float r00 = 1.0f - 2.0f * (y * y + z * z), r01 = 2.0f * (x * y - r * z), r02 = 2.0f * (x * z + r * y);
    const float r10 = 2.0f * (x * y + r * z), r11 = 1.0f - 2.0f * (x * x + z * z), r12 = 2.0f * (y * z - r * x);
    const float r20 = 2.0f * (x * z - r * y), r21 = 2.0f * (y * z + r * x), r22 = 1.0f - 2.0f * (x * x + y * y);
    a.xyz_out[3 * row] = (r00 * vx + r01 * vy + r02 * vz) + a.xyz[3 * i];
    a.xyz_out[3 * row + 1] = (r10 * vx + r11 * vy + r12 * vz) + a.xyz[3 * i + 1];
    a.xyz_out[3 * row + 2] = (r20 * vx + r21 * vy + r22 * vz) + a.xyz[3 * i + 2];
    const float div = 0.8f * 2.0f;
    a.scaling_out[3 * row] = logf(sx / div);
    a.scaling_out[3 * row + 1] = logf(sy / div);
    a.scaling_out[3 * row + 2] = logf(sz / div);
}

__global__ void __launch_bounds__(256)
reset_opacity_kernel(float* __restrict__ opacity, int n, float cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float o = fminf(1.0f / (1.0f + expf(-opacity[i])), cap);
    opacity[i] = logf(o / (1.0f - o));
}

// ---- entry points -------------------------------------------------------------------------------------
extern "C" int fdgs_densify_classify(const fdgs_densify_classify_args* a, void* stream) {
    if (!a || a->num_points < 0) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->xyz_gradient_accum || !a->denom || !a->scaling || !a->cls) return FDGS_EINVAL;
    hipLaunchKernelGGL(densify_classify_kernel, dim3((a->num_points + SG_THREADS - 1) / SG_THREADS), dim3(SG_THREADS), 0,
                       (hipStream_t)stream, *a);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

extern "C" int fdgs_prune_classify(const fdgs_prune_classify_args* a, void* stream) {
    if (!a || a->num_points < 0) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->opacity || !a->cls) return FDGS_EINVAL;
    if (a->max_screen_size > 0.0f && (!a->max_radii2D || !a->scaling)) return FDGS_EINVAL;
    hipLaunchKernelGGL(prune_classify_kernel, dim3((a->num_points + SG_THREADS - 1) / SG_THREADS), dim3(SG_THREADS), 0,
                       (hipStream_t)stream, *a);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

static int surgery_blocks(int32_t n) { return n > 0 ? (n + SG_BLOCK_PTS - 1) / SG_BLOCK_PTS : 1; }

extern "C" size_t fdgs_surgery_workspace_bytes(int32_t num_points) {
    return (size_t)surgery_blocks(num_points) * 3 * sizeof(uint32_t);
}

extern "C" int fdgs_surgery_plan(const fdgs_surgery_plan_args* a, void* stream) {
    if (!a || a->num_points < 0 || !a->counts) return FDGS_EINVAL;
    if (a->num_points > 0 && (!a->cls || !a->src || !a->workspace)) return FDGS_EINVAL;
    if (a->num_points > 0 && a->workspace_bytes < fdgs_surgery_workspace_bytes(a->num_points)) return FDGS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (a->num_points == 0) {
        if (hipMemsetAsync(a->counts, 0, 4 * sizeof(int32_t), s) != hipSuccess) return FDGS_ELAUNCH;
        return FDGS_OK;
    }
    const int blocks = surgery_blocks(a->num_points);
    uint32_t* tot = reinterpret_cast<uint32_t*>(a->workspace);
    hipLaunchKernelGGL(surgery_count_kernel, dim3(blocks), dim3(SG_THREADS), 0, s, a->cls, a->num_points, tot);
    FDGS_CHECK_LAUNCH();
    hipLaunchKernelGGL(surgery_scan_kernel, dim3(1), dim3(1024), 0, s, tot, blocks, a->counts);
    FDGS_CHECK_LAUNCH();
    hipLaunchKernelGGL(surgery_map_kernel, dim3(blocks), dim3(SG_THREADS), 0, s, a->cls, a->num_points, tot, a->counts,
                       a->src);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

extern "C" int fdgs_surgery_apply(const fdgs_surgery_apply_args* a, void* stream) {
    if (!a || a->rows_after < 0 || a->rows_carried < 0 || a->rows_carried > a->rows_after) return FDGS_EINVAL;
    if (a->num_arrays < 0 || a->num_arrays > FDGS_SURGERY_MAX_ARRAYS) return FDGS_EINVAL;
    if (a->rows_after == 0 || a->num_arrays == 0) return FDGS_OK;
    if (!a->src) return FDGS_EINVAL;
    SurgeryApplyK k;
    k.rows_after = a->rows_after; k.rows_carried = a->rows_carried; k.src = a->src;
    int32_t widest = 1;
    for (int i = 0; i < a->num_arrays; ++i) {
        if (!a->out[i] || a->row_bytes[i] <= 0) return FDGS_EINVAL;
        if (!a->in[i] && !((a->flags[i] & FDGS_SURGERY_ZERO_NEW) && a->rows_carried == 0)) return FDGS_EINVAL;
        const bool words = (a->row_bytes[i] % 4 == 0) && ((((size_t)a->in[i] | (size_t)a->out[i]) & 3) == 0);
        k.in[i] = a->in[i]; k.out[i] = a->out[i];
        k.row_units[i] = words ? a->row_bytes[i] / 4 : a->row_bytes[i];
        k.flags[i] = (a->flags[i] & FDGS_SURGERY_ZERO_NEW) | (words ? 0 : 2);
        if (k.row_units[i] > widest) widest = k.row_units[i];
    }
    for (int i = a->num_arrays; i < FDGS_SURGERY_MAX_ARRAYS; ++i) {
        k.in[i] = nullptr; k.out[i] = nullptr; k.row_units[i] = 0; k.flags[i] = 0;
    }
    long long blocks = ((long long)a->rows_after * widest + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(surgery_apply_kernel, dim3((unsigned)blocks, a->num_arrays), dim3(256), 0, (hipStream_t)stream, k);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

extern "C" int fdgs_split_children(const fdgs_split_children_args* a, void* stream) {
    if (!a || a->num_children < 0 || a->first_child_row < 0) return FDGS_EINVAL;
    if (a->num_children == 0) return FDGS_OK;
    if (!a->src || !a->xyz || !a->scaling || !a->rotation || !a->noise || !a->xyz_out || !a->scaling_out) return FDGS_EINVAL;
    hipLaunchKernelGGL(split_children_kernel, dim3((a->num_children + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

extern "C" int fdgs_reset_opacity(float* opacity, int32_t num_points, float cap, void* stream) {
    if (num_points < 0 || !(cap > 0.0f && cap < 1.0f)) return FDGS_EINVAL;
    if (num_points == 0) return FDGS_OK;
    if (!opacity) return FDGS_EINVAL;
    hipLaunchKernelGGL(reset_opacity_kernel, dim3((num_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, opacity,
                       num_points, cap);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
