// Mean squared distance of every point to its three nearest neighbours: the `simple_knn._C.distCUDA2` call that
// seeds the Gaussian scales from the initial point cloud (src/model/gaussian_model.py:57; the CUDA package itself
// is not part of the reference tree).  Initialisation-only, so the exact all-pairs search is used: candidates are
// staged through LDS in tiles of 256 points, every thread keeps the three smallest squared distances of its
// query in registers.  1.6e11 pair evaluations at N = 400k, a few tens of milliseconds.
#include "fdgs_common.h"

#define KNN_THREADS 256

__global__ void __launch_bounds__(KNN_THREADS)
knn_mean_dist2_kernel(const float* __restrict__ pts, int n, float* __restrict__ out) {
    __shared__ float sx[KNN_THREADS], sy[KNN_THREADS], sz[KNN_THREADS];
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    const int qi = min(i, n - 1);
    const float qx = pts[3 * (size_t)qi], qy = pts[3 * (size_t)qi + 1], qz = pts[3 * (size_t)qi + 2];
    const float inf = __builtin_huge_valf();
    float b0 = inf, b1 = inf, b2 = inf;                 // b0 <= b1 <= b2
    for (int base = 0; base < n; base += KNN_THREADS) {
        const int j = base + threadIdx.x;
        __syncthreads();
        if (j < n) { sx[threadIdx.x] = pts[3 * (size_t)j]; sy[threadIdx.x] = pts[3 * (size_t)j + 1]; sz[threadIdx.x] = pts[3 * (size_t)j + 2]; }
        __syncthreads();
        const int cnt = min(KNN_THREADS, n - base);
        for (int k = 0; k < cnt; ++k) {
            const float dx = sx[k] - qx, dy = sy[k] - qy, dz = sz[k] - qz;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < b2 && base + k != qi) {             // the query itself is skipped by index; duplicates count
                if (d < b1) {
                    b2 = b1;
                    if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
                } else {
                    b2 = d;
                }
            }
        }
    }
    if (i < n) out[i] = (b0 + b1 + b2) / 3.0f;
}

extern "C" int fdgs_knn_mean_dist2(const float* points, int32_t num_points, float* mean_dist2, void* stream) {
    if (num_points < 0) return FDGS_EINVAL;
    if (num_points == 0) return FDGS_OK;
    if (!points || !mean_dist2) return FDGS_EINVAL;
    hipLaunchKernelGGL(knn_mean_dist2_kernel, dim3((num_points + KNN_THREADS - 1) / KNN_THREADS), dim3(KNN_THREADS), 0,
                       (hipStream_t)stream, points, (int)num_points, mean_dist2);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
