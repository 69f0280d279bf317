// Rasterizer stages 1-3 and 5: per-Gaussian preprocess, inclusive scan of
// tiles_touched, duplicateWithKeys, identifyTileRanges.
//
// This translation unit is compiled with -ffp-contract=off: every float
// expression that feeds an integer decision (radius, tile rectangle, depth bits
// of the sort key) is a sequence of single IEEE operations in a fixed order, so
// that the integer stages are reproducible bit for bit by any IEEE
// implementation that evaluates the same sequence.
#include "fdgs_common.h"
#include <cstdlib>

#pragma clang fp contract(off)

// Shared by stage 1 (which leaves per-workgroup sums) and the scan of stage 2.
#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_CHUNK (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < FDGS_WAVE; d <<= 1) {
        uint32_t n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (256 threads); returns the
// exclusive prefix, *total gets the block total.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* total, uint32_t* lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t s = lds4[w];
        if (w < wave) base += s;
    }
    *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return base + incl - v;
}

// Scan workspace: [0] smallest and [1] largest depth bit pattern among the listed Gaussians (0xffffffff / 0 when
// there are none), [2] R = the total, [3] unused, then per group of 256 Gaussians (one workgroup of stage 1):
// sum -> exclusive base, depth minimum, depth maximum.
#define SCAN_HEADER 4
#define SCAN_GROUP 256
#define SCAN_GROUPS_PER_CHUNK (SCAN_CHUNK / SCAN_GROUP)

__device__ __forceinline__ void block_minmax_256(uint32_t& lo, uint32_t& hi, uint32_t* lds8) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { lds8[wave] = lo; lds8[4 + wave] = hi; }
    __syncthreads();
    lo = min(min(lds8[0], lds8[1]), min(lds8[2], lds8[3]));
    hi = max(max(lds8[4], lds8[5]), max(lds8[6], lds8[7]));
}

// ---------------------------------------------------------------------------
// Stage 1: one thread per Gaussian.  HBM-bound: reads 59 floats, writes the
// 48-byte record + radius + tiles_touched.  Each workgroup also leaves the sum of its 256 tile counts and the
// smallest / largest depth bit pattern among its listed Gaussians in the scan workspace: the first level of the scan.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
preprocess_fwd_kernel(fdgs_preprocess_fwd_args a) {
    __shared__ float sh_tile[256 * FDGS_SH_TILE_STRIDE];
    const int first = blockIdx.x * 256;
    const int rows = min(256, a.num_points - first);
    if (!a.colors_precomp) {                        // the workgroup's SH coefficients, read with 16-byte accesses
        sh_tile_copy<true>(sh_tile, const_cast<float*>(a.shs) + (size_t)first * 48, rows);
        __syncthreads();
    }
    // threads past the last Gaussian stay for the workgroup's reductions: they recompute the first row and store nothing
    const bool live = (int)threadIdx.x < rows;
    const int i = live ? first + (int)threadIdx.x : first;
    const int W = a.image_width, H = a.image_height;
    const int gx = (W + FDGS_TILE - 1) / FDGS_TILE, gy = (H + FDGS_TILE - 1) / FDGS_TILE;
    const float* __restrict__ V = a.viewmatrix;
    const float* __restrict__ P = a.projmatrix;

    float4* grec = reinterpret_cast<float4*>(a.geom) + 3 * (size_t)i;
    const float x = a.means3D[3 * i + 0], y = a.means3D[3 * i + 1], z = a.means3D[3 * i + 2];

    const float pvx = V[0] * x + V[4] * y + V[8] * z + V[12];
    const float pvy = V[1] * x + V[5] * y + V[9] * z + V[13];
    const float pvz = V[2] * x + V[6] * y + V[10] * z + V[14];
    bool visible = live && pvz > FDGS_NEAR_Z;

    int radius = 0;
    uint32_t touched = 0;
    float px = 0.f, py = 0.f, conA = 0.f, conB = 0.f, conC = 0.f;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    int clampmask = 0;
    if (visible) {
        const float phx = P[0] * x + P[4] * y + P[8] * z + P[12];
        const float phy = P[1] * x + P[5] * y + P[9] * z + P[13];
        const float phw = P[3] * x + P[7] * y + P[11] * z + P[15];
        const float p_w = 1.0f / (phw + 1e-7f);
        const float ppx = phx * p_w, ppy = phy * p_w;

        // 3-D covariance Sigma = (R S)(R S)^T, R as the reference's build_rotation
        // (src/utils/general_utils.py:108-121) for a normalised (w,x,y,z), unless it is given.
        float c00, c01, c02, c11, c12, c22;
        if (a.cov3D_precomp) {
            const float* __restrict__ cv = a.cov3D_precomp + (size_t)i * 6;
            c00 = cv[0]; c01 = cv[1]; c02 = cv[2]; c11 = cv[3]; c12 = cv[4]; c22 = cv[5];
        } else {
            const float mod = a.scale_modifier;
            float s0 = a.scales[3 * i + 0], s1 = a.scales[3 * i + 1], s2 = a.scales[3 * i + 2];
            if (a.activations & FDGS_ACT_SCALE_EXP) { s0 = expf(s0); s1 = expf(s1); s2 = expf(s2); }
            const float sx = mod * s0, sy = mod * s1, sz = mod * s2;
            float qr = a.rotations[4 * i + 0], qx = a.rotations[4 * i + 1];
            float qy = a.rotations[4 * i + 2], qz = a.rotations[4 * i + 3];
            if (a.activations & FDGS_ACT_ROTATION_NORMALIZE) {
                const float inv = 1.0f / fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);
                qr *= inv; qx *= inv; qy *= inv; qz *= inv;
            }
            const float R00 = 1.0f - 2.0f * (qy * qy + qz * qz);
            const float R01 = 2.0f * (qx * qy - qr * qz);
            const float R02 = 2.0f * (qx * qz + qr * qy);
            const float R10 = 2.0f * (qx * qy + qr * qz);
            const float R11 = 1.0f - 2.0f * (qx * qx + qz * qz);
            const float R12 = 2.0f * (qy * qz - qr * qx);
            const float R20 = 2.0f * (qx * qz - qr * qy);
            const float R21 = 2.0f * (qy * qz + qr * qx);
            const float R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
            const float M00 = R00 * sx, M01 = R01 * sy, M02 = R02 * sz;
            const float M10 = R10 * sx, M11 = R11 * sy, M12 = R12 * sz;
            const float M20 = R20 * sx, M21 = R21 * sy, M22 = R22 * sz;
            c00 = M00 * M00 + M01 * M01 + M02 * M02;
            c01 = M00 * M10 + M01 * M11 + M02 * M12;
            c02 = M00 * M20 + M01 * M21 + M02 * M22;
            c11 = M10 * M10 + M11 * M11 + M12 * M12;
            c12 = M10 * M20 + M11 * M21 + M12 * M22;
            c22 = M20 * M20 + M21 * M21 + M22 * M22;
        }

        // EWA projection
        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float fx = (float)W / (2.0f * a.tanfovx), fy = (float)H / (2.0f * a.tanfovy);
        const float tz = pvz;
        const float tx = fminf(limx, fmaxf(-limx, pvx / tz)) * tz;
        const float ty = fminf(limy, fmaxf(-limy, pvy / tz)) * tz;
        const float J00 = fx / tz;
        const float J02 = -(fx * tx) / (tz * tz);
        const float J11 = fy / tz;
        const float J12 = -(fy * ty) / (tz * tz);
        const float T00 = J00 * V[0] + J02 * V[2];
        const float T01 = J00 * V[4] + J02 * V[6];
        const float T02 = J00 * V[8] + J02 * V[10];
        const float T10 = J11 * V[1] + J12 * V[2];
        const float T11 = J11 * V[5] + J12 * V[6];
        const float T12 = J11 * V[9] + J12 * V[10];
        const float U00 = T00 * c00 + T01 * c01 + T02 * c02;
        const float U01 = T00 * c01 + T01 * c11 + T02 * c12;
        const float U02 = T00 * c02 + T01 * c12 + T02 * c22;
        const float U10 = T10 * c00 + T11 * c01 + T12 * c02;
        const float U11 = T10 * c01 + T11 * c11 + T12 * c12;
        const float U12 = T10 * c02 + T11 * c12 + T12 * c22;
        const float ca = U00 * T00 + U01 * T01 + U02 * T02 + FDGS_COV_DILATION;
        const float cb2 = U00 * T10 + U01 * T11 + U02 * T12;
        const float cc = U10 * T10 + U11 * T11 + U12 * T12 + FDGS_COV_DILATION;
        const float det = ca * cc - cb2 * cb2;
        visible = (det != 0.0f);
        if (visible) {
            const float det_inv = 1.0f / det;
            conA = cc * det_inv;
            conB = -cb2 * det_inv;
            conC = ca * det_inv;
            const float mid = 0.5f * (ca + cc);
            const float disc = sqrtf(fmaxf(mid * mid - det, 0.1f));
            const float lam = fmaxf(mid + disc, mid - disc);
            float rad_f = ceilf(3.0f * sqrtf(lam));
            rad_f = (rad_f != rad_f) ? 0.0f : fminf(rad_f, 16777216.0f);
            px = ((ppx + 1.0f) * (float)W - 1.0f) * 0.5f;
            py = ((ppy + 1.0f) * (float)H - 1.0f) * 0.5f;
            int x0, y0, x1, y1;
            fdgs_tile_rect(px, py, rad_f, gx, gy, x0, y0, x1, y1);
            const int area = (x1 - x0) * (y1 - y0);
            visible = area > 0;
            if (visible) {
                radius = (int)rad_f;
                touched = (uint32_t)area;
            }
        }
        if (visible && a.colors_precomp) {
            cr = a.colors_precomp[3 * i + 0]; cg = a.colors_precomp[3 * i + 1]; cb = a.colors_precomp[3 * i + 2];
        } else if (visible) {
            // colour from SH, basis and signs as src/utils/sh_utils.py:57-100
            const float ddx = x - a.campos[0], ddy = y - a.campos[1], ddz = z - a.campos[2];
            const float nrm = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
            const float ux = ddx / nrm, uy = ddy / nrm, uz = ddz / nrm;
            const float* sh = sh_tile + threadIdx.x * FDGS_SH_TILE_STRIDE;
            float res[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) res[c] = SH_C0 * sh[c];
            if (a.sh_degree > 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    res[c] = res[c] - (SH_C1 * uy) * sh[3 + c] + (SH_C1 * uz) * sh[6 + c] - (SH_C1 * ux) * sh[9 + c];
                if (a.sh_degree > 1) {
                    const float xx = ux * ux, yy = uy * uy, zz = uz * uz;
                    const float xy = ux * uy, yz = uy * uz, xz = ux * uz;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        res[c] = res[c] + (SH_C2_0 * xy) * sh[12 + c] + (SH_C2_1 * yz) * sh[15 + c] +
                                 (SH_C2_2 * (2.0f * zz - xx - yy)) * sh[18 + c] + (SH_C2_3 * xz) * sh[21 + c] +
                                 (SH_C2_4 * (xx - yy)) * sh[24 + c];
                    if (a.sh_degree > 2) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            res[c] = res[c] + ((SH_C3_0 * uy) * (3.0f * xx - yy)) * sh[27 + c] +
                                     ((SH_C3_1 * xy) * uz) * sh[30 + c] +
                                     ((SH_C3_2 * uy) * (4.0f * zz - xx - yy)) * sh[33 + c] +
                                     ((SH_C3_3 * uz) * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh[36 + c] +
                                     ((SH_C3_4 * ux) * (4.0f * zz - xx - yy)) * sh[39 + c] +
                                     ((SH_C3_5 * uz) * (xx - yy)) * sh[42 + c] +
                                     ((SH_C3_6 * ux) * (xx - 3.0f * yy)) * sh[45 + c];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                res[c] = res[c] + 0.5f;
                if (res[c] < 0.0f) { clampmask |= (1 << c); res[c] = 0.0f; }
            }
            cr = res[0]; cg = res[1]; cb = res[2];
        }
    }
    float op = visible ? a.opacities[i] : 0.0f;
    if (visible && (a.activations & FDGS_ACT_OPACITY_SIGMOID)) op = 1.0f / (1.0f + expf(-op));
    if (live) {
        grec[0] = make_float4(px, py, conA, conB);
        grec[1] = make_float4(conC, op, cr, cg);
        grec[2] = make_float4(cb, pvz, __int_as_float(fdgs_pack_radius(radius, clampmask)),
                              __uint_as_float(visible ? fdgs_pack_reach(conA, conB, conC, op) : 0xfc00fc00u));
        a.radii[i] = radius;
        a.tiles_touched[i] = touched;
    }
    // first level of the scan and of the frame's depth range (a listed depth is > 0.2, so its bits are not 0)
    uint32_t* lds = reinterpret_cast<uint32_t*>(sh_tile);    // the SH rows have been consumed
    __syncthreads();
    uint32_t total;
    block_excl_scan_256(touched, &total, lds);
    uint32_t lo = touched ? __float_as_uint(pvz) : 0xffffffffu, hi = touched ? __float_as_uint(pvz) : 0u;
    block_minmax_256(lo, hi, lds + 4);
    if (threadIdx.x == 0) {
        const int n_groups = gridDim.x;
        a.scan_workspace[SCAN_HEADER + blockIdx.x] = total;
        a.scan_workspace[SCAN_HEADER + n_groups + blockIdx.x] = lo;
        a.scan_workspace[SCAN_HEADER + 2 * n_groups + blockIdx.x] = hi;
    }
}

// ---------------------------------------------------------------------------
// Stage 2: inclusive scan of uint32 counts.  Stage 1 has left the sum of every group of 256 counts; one launch
// (scan_apply_fused_kernel) finishes the scan, or two (a single-block scan of the group sums, then per-chunk scan +
// base) when there are more groups than a workgroup should sum for itself.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SCAN_THREADS)
scan_group_bases_kernel(uint32_t* __restrict__ ws, int n_chunks) {
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t lds8[8];
    uint32_t* __restrict__ chunk_sums = ws + SCAN_HEADER;
    uint32_t carry = 0, lo = 0xffffffffu, hi = 0u;
    for (int start = 0; start < n_chunks; start += SCAN_THREADS) {
        const int idx = start + threadIdx.x;
        uint32_t v = idx < n_chunks ? chunk_sums[idx] : 0u;
        if (idx < n_chunks) {
            lo = min(lo, chunk_sums[n_chunks + idx]);
            hi = max(hi, chunk_sums[2 * n_chunks + idx]);
        }
        uint32_t total;
        uint32_t excl = block_excl_scan_256(v, &total, lds4);
        if (idx < n_chunks) chunk_sums[idx] = carry + excl;   // exclusive base of the chunk
        carry += total;
    }
    block_minmax_256(lo, hi, lds8);
    if (threadIdx.x == 0) { ws[0] = lo; ws[1] = hi; ws[2] = carry; ws[3] = 0u; }
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ chunk_bases,
                  uint32_t* __restrict__ out, int n) {
    __shared__ uint32_t lds4[4];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t run = chunk_bases[SCAN_HEADER + blockIdx.x * SCAN_GROUPS_PER_CHUNK] + block_excl_scan_256(s, &total, lds4);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        run += v[k];
        if (base + k < n) out[base + k] = run;
    }
}

// scan_group_bases_kernel and scan_apply_kernel in one launch, for up to SCAN_FUSED_MAX_GROUPS groups: every
// workgroup sums the group sums before its own chunk (a few hundred to a few thousand words) instead of reading a
// base a launch of one workgroup has prepared; the last workgroup also leaves the header (depth range, R).  The
// group sums stay as stage 1 wrote them.
#define SCAN_FUSED_MAX_GROUPS 8192
__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_fused_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ ws, uint32_t* __restrict__ out,
                        int n, int n_groups) {
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t lds8[8];
    const uint32_t* __restrict__ chunk_sums = ws + SCAN_HEADER;
    const int n_chunks = n_groups;                           // the three arrays are n_groups long
    const int groups_before = min((int)blockIdx.x * SCAN_GROUPS_PER_CHUNK, n_groups);
    uint32_t before = 0;
    for (int j = threadIdx.x; j < groups_before; j += SCAN_THREADS) before += chunk_sums[j];
    uint32_t chunk_base;
    block_excl_scan_256(before, &chunk_base, lds4);          // only the total is used: the sum of the earlier chunks
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t run = chunk_base + block_excl_scan_256(s, &total, lds4);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        run += v[k];
        if (base + k < n) out[base + k] = run;
    }
    if (blockIdx.x == gridDim.x - 1) {                       // uniform over the workgroup
        uint32_t lo = 0xffffffffu, hi = 0u;
        for (int j = threadIdx.x; j < n_chunks; j += SCAN_THREADS) {
            lo = min(lo, chunk_sums[n_chunks + j]);
            hi = max(hi, chunk_sums[2 * n_chunks + j]);
        }
        block_minmax_256(lo, hi, lds8);
        if (threadIdx.x == 0) { ws[0] = lo; ws[1] = hi; ws[2] = chunk_base + total; ws[3] = 0u; }
    }
}

// ---------------------------------------------------------------------------
// Stage 3: duplicateWithKeys.  Gaussian i owns the list positions [offsets[i-1], offsets[i]) and gets one
// (key, value) pair per touched tile there, tiles row-major.  key = (tile_id << 32) | bits(depth); when the
// frame's depth range is known, in the order-preserving compressed form
// (tile_id << depth_bits) | (bits(depth) - depth_min), which the sort's last pass expands again.
// A wave takes 64 consecutive Gaussians and writes their positions TOGETHER: lane l writes positions first + l,
// first + l + 64, ... and finds the owner of each by binary search over the 64 scan values (in LDS), so every store
// instruction covers 64 consecutive entries.  (One thread per Gaussian looping over its own tiles stores to 64
// unrelated lines per instruction and runs as long as the wave's largest splat.)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
duplicate_with_keys_kernel(int n, const float* __restrict__ geom, const uint32_t* __restrict__ offsets,
                           uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int gx, int gy,
                           int depth_bits, uint32_t depth_min, uint32_t* __restrict__ ranges) {
    __shared__ uint32_t s_end[4][64], s_start[4][64], s_x0w[4][64], s_y0[4][64], s_depth[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 256 + threadIdx.x;
    // the tile ranges start as (0, 0) = empty; identifyTileRanges, five sort passes later, fills the others
    for (int w = i; w < 2 * gx * gy; w += gridDim.x * 256) ranges[w] = 0u;
    const int ic = i < n ? i : n - 1;
    const uint32_t end = offsets[ic];
    const uint32_t start = i < n ? (i == 0 ? 0u : offsets[i - 1]) : end;
    uint32_t x0w = 0, y0v = 0, dbits = 0;
    if (end > start) {
        const float4 g2 = reinterpret_cast<const float4*>(geom)[3 * (size_t)i + 2];
        const int radius = fdgs_unpack_radius(__float_as_int(g2.z));
        const float4 g0 = reinterpret_cast<const float4*>(geom)[3 * (size_t)i + 0];
        int x0, y0, x1, y1;
        fdgs_tile_rect(g0.x, g0.y, (float)radius, gx, gy, x0, y0, x1, y1);
        x0w = (uint32_t)x0 | ((uint32_t)(x1 - x0) << 16);
        y0v = (uint32_t)y0;
        dbits = __float_as_uint(g2.y) - depth_min;
    }
    s_end[wave][lane] = end;
    s_start[wave][lane] = start;
    s_x0w[wave][lane] = x0w;
    s_y0[wave][lane] = y0v;
    s_depth[wave][lane] = dbits;
    __syncthreads();
    const uint32_t first = s_start[wave][0], last = s_end[wave][63];
    const uint32_t i0 = (uint32_t)(blockIdx.x * 256 + wave * 64);
    for (uint32_t pos = first + lane; pos < last; pos += 64) {
        int l = 0;                                  // the first Gaussian of the wave whose end lies beyond pos
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (s_end[wave][l + step - 1] <= pos) l += step;
        const uint32_t q = pos - s_start[wave][l];
        const uint32_t xw = s_x0w[wave][l];
        const uint32_t w = xw >> 16;
        const uint32_t row = q / w;
        const uint32_t tile = (s_y0[wave][l] + row) * (uint32_t)gx + (xw & 0xffffu) + (q - row * w);
        keys[pos] = ((uint64_t)tile << depth_bits) | (uint64_t)s_depth[wave][l];
        vals[pos] = i0 + (uint32_t)l;
    }
}

// ---------------------------------------------------------------------------
// Stage 5: identifyTileRanges over the sorted keys.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
identify_tile_ranges_kernel(int64_t R, const uint64_t* __restrict__ keys, uint32_t* __restrict__ ranges) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R) return;
    const uint32_t tile = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[2 * tile] = 0u;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (prev != tile) {
            ranges[2 * prev + 1] = (uint32_t)idx;
            ranges[2 * tile] = (uint32_t)idx;
        }
    }
    if (idx == R - 1) ranges[2 * tile + 1] = (uint32_t)R;
}

// ---------------------------------------------------------------------------
// Tile ids by decreasing list length, for the blend launches: workgroups are started in index order, and with the
// tiles in image order the ones with the longest lists can start last and run alone at the end of the launch
// (measured at 1352 x 1014, 300k Gaussians: blend_fwd 0.115-0.129 -> 0.092-0.098 ms, blend_bwd 0.24-0.27 -> 0.215-0.224).
// A counting sort in one workgroup on a logarithmic length class (16 classes per octave): exact order is not needed.
// ---------------------------------------------------------------------------
#define ORDER_CLASSES 512
__device__ __forceinline__ int tile_length_class(uint32_t len) {      // larger length -> smaller class
    if (len == 0) return ORDER_CLASSES - 1;
    const int e = 31 - __clz((int)len);                               // floor(log2(len)), 0 .. 31
    const uint32_t frac = e >= 4 ? (len >> (e - 4)) & 15u : (len << (4 - e)) & 15u;
    const int k = e * 16 + (int)frac;                                  // 0 .. 511, monotone in len
    return ORDER_CLASSES - 1 - k;
}

// One workgroup of THREADS >= ORDER_CLASSES threads; the lengths come from `totals` [tiles] or, when that is null,
// from the ranges.  count: ORDER_CLASSES words of LDS, wsum: THREADS / 64 words.
template <int THREADS>
__device__ __forceinline__ void tile_order_block(const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ totals,
                                                 uint32_t* __restrict__ order, int tiles, uint32_t* count,
                                                 uint32_t* wsum, const uint32_t* starts = nullptr,
                                                 uint4* __restrict__ desc = nullptr) {
    static_assert(THREADS >= ORDER_CLASSES, "one thread per length class in the scan");
    for (int k = threadIdx.x; k < ORDER_CLASSES; k += THREADS) count[k] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += THREADS) {
        const uint32_t len = totals ? totals[t] : ranges[2 * t + 1] - ranges[2 * t];
        atomicAdd(&count[tile_length_class(len)], 1u);
    }
    __syncthreads();
    // exclusive scan of the class counts by the first ORDER_CLASSES threads (8 waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = threadIdx.x < ORDER_CLASSES ? count[threadIdx.x] : 0u;
    uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    __syncthreads();
    if (threadIdx.x < ORDER_CLASSES) count[threadIdx.x] = base + incl - v;   // first slot of the class
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += THREADS) {
        const uint32_t len = totals ? totals[t] : ranges[2 * t + 1] - ranges[2 * t];
        const uint32_t pos = atomicAdd(&count[tile_length_class(len)], 1u);
        order[pos] = (uint32_t)t;
        if (desc) desc[pos] = make_uint4((uint32_t)t, starts[t], len, 0u);     // starts: first list positions (LDS)
    }
}

__global__ void __launch_bounds__(1024)
tile_order_kernel(const uint32_t* __restrict__ ranges, uint32_t* __restrict__ order, int tiles) {
    __shared__ uint32_t count[ORDER_CLASSES];
    __shared__ uint32_t wsum[16];
    tile_order_block<1024>(ranges, nullptr, order, tiles, count, wsum);
}

#include "tile_sort.h"

// ---------------------------------------------------------------------------
// Host entry points
// ---------------------------------------------------------------------------
extern "C" size_t fdgs_raster_scan_workspace_bytes(int32_t num_points) {
    if (num_points < 0) return 0;
    size_t groups = ((size_t)num_points + SCAN_GROUP - 1) / SCAN_GROUP;
    return (SCAN_HEADER + 3 * groups + 1) * sizeof(uint32_t);
}

extern "C" int fdgs_raster_preprocess_fwd(const fdgs_preprocess_fwd_args* a, void* stream) {
    if (!a || a->num_points < 0 || a->image_width <= 0 || a->image_height <= 0) return FDGS_EINVAL;
    if (a->sh_degree < 0 || a->sh_degree > 3) return FDGS_EINVAL;
    const int n = a->num_points;
    if (n == 0) return FDGS_OK;
    if (!a->viewmatrix || !a->projmatrix || !a->campos || !a->means3D || !a->opacities || !a->geom || !a->radii ||
        !a->tiles_touched || !a->point_offsets || !a->scan_workspace)
        return FDGS_EINVAL;
    if (!a->colors_precomp && !a->shs) return FDGS_EINVAL;
    if (!a->cov3D_precomp && (!a->scales || !a->rotations)) return FDGS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int groups = (n + SCAN_GROUP - 1) / SCAN_GROUP;    // = workgroups of stage 1 = entries of the workspace arrays
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(groups), dim3(SCAN_GROUP), 0, s, *a);
    FDGS_CHECK_LAUNCH();
    const int chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    // FDGS_SCAN_FUSED_MAX_GROUPS: the largest group count scanned by one launch (tests force the two-launch path)
    const char* fused_env = getenv("FDGS_SCAN_FUSED_MAX_GROUPS");
    const int fused_max = fused_env ? atoi(fused_env) : SCAN_FUSED_MAX_GROUPS;
    if (groups <= fused_max) {
        hipLaunchKernelGGL(scan_apply_fused_kernel, dim3(chunks), dim3(SCAN_THREADS), 0, s, a->tiles_touched,
                           a->scan_workspace, a->point_offsets, n, groups);
        FDGS_CHECK_LAUNCH();
        return FDGS_OK;
    }
    hipLaunchKernelGGL(scan_group_bases_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, a->scan_workspace, groups);
    FDGS_CHECK_LAUNCH();
    hipLaunchKernelGGL(scan_apply_kernel, dim3(chunks), dim3(SCAN_THREADS), 0, s, a->tiles_touched,
                       a->scan_workspace, a->point_offsets, n);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

// sort lives in radix_sort.hip
int fdgs_radix_sort_passes(int end_bit);
int fdgs_radix_sort_pingpong(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, int64_t n,
                             int end_bit, void* workspace, size_t workspace_bytes, hipStream_t s, int expand_bits,
                             uint32_t expand_add);

extern "C" int fdgs_raster_bin_sort(const fdgs_binsort_args* a, void* stream) {
    if (!a || a->num_points < 0 || a->num_rendered < 0 || a->image_width <= 0 || a->image_height <= 0)
        return FDGS_EINVAL;
    if (a->num_rendered > 0x7fffffffLL) return FDGS_EINVAL;
    if (a->phase < FDGS_BINSORT_ALL || a->phase > FDGS_BINSORT_FINISH) return FDGS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int gx = fdgs_tiles_x(a->image_width), gy = fdgs_tiles_y(a->image_height);
    const int tiles = gx * gy;
    // FDGS_SORT=radix: the least-significant-digit radix sort over all R keys; otherwise the tile-binned sort
    // (tile_sort.h) wherever it applies.  Both give the same keys, values and ranges.
    const char* sort_env = getenv("FDGS_SORT");
    const bool binned = !(sort_env && sort_env[0] == 'r');
    if (a->phase == FDGS_BINSORT_COUNT) {           // before the list length is known: the per-tile counts
        if (a->num_points == 0 || !binned) return FDGS_OK;
        if (!a->geom || !a->point_offsets || !a->sort_workspace) return FDGS_EINVAL;
        if (a->sort_workspace_bytes < fdgs_raster_sort_workspace_bytes(0)) return FDGS_EWORKSPACE;
        const int rc = fdgs_tile_sort(a, (uint32_t*)a->sort_workspace + FDGS_RADIX_WS_WORDS,
                                      a->sort_workspace_bytes / sizeof(uint32_t) - FDGS_RADIX_WS_WORDS, s,
                                      FDGS_BINSORT_COUNT);
        return rc == 1 ? FDGS_OK : rc;
    }
    if (!a->ranges) return FDGS_EINVAL;
    const int64_t R = a->num_rendered;
    if (R == 0 || a->num_points == 0) {
        if (hipMemsetAsync(a->ranges, 0, sizeof(uint32_t) * 2 * (size_t)tiles, s) != hipSuccess) return FDGS_ELAUNCH;
        if (a->tile_order) {
            hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, a->ranges, a->tile_order, tiles);
            FDGS_CHECK_LAUNCH();
        }
        return FDGS_OK;
    }
    if (!a->geom || !a->point_offsets || !a->keys_unsorted || !a->vals_unsorted || !a->keys_sorted ||
        !a->vals_sorted || !a->sort_workspace)
        return FDGS_EINVAL;
    if (a->sort_workspace_bytes < fdgs_raster_sort_workspace_bytes(R)) return FDGS_EWORKSPACE;
    if (binned) {
        const int rc = fdgs_tile_sort(a, (uint32_t*)a->sort_workspace + FDGS_RADIX_WS_WORDS,
                                      a->sort_workspace_bytes / sizeof(uint32_t) - FDGS_RADIX_WS_WORDS, s, a->phase);
        if (rc != 1) return rc;
    }
    int tile_bits = 0;
    while ((1 << tile_bits) < tiles) ++tile_bits;   // keys are < tiles << 32
    // With the frame's depth range (the first two words fdgs_raster_preprocess_fwd leaves in its scan workspace) the
    // keys are sorted as (tile << k) | (depth bits - minimum), k = the bits of (maximum - minimum): the same order
    // on fewer digits; the last pass writes the keys back in the (tile << 32) | depth bits form.
    int depth_bits = 32;
    uint32_t depth_min = 0;
    if (a->depth_bits_max >= a->depth_bits_min && a->depth_bits_max != 0) {
        depth_min = a->depth_bits_min;
        const uint32_t span = a->depth_bits_max - a->depth_bits_min;
        depth_bits = 1;
        while (depth_bits < 32 && (span >> depth_bits) != 0) ++depth_bits;
    }
    const int end_bit = depth_bits + tile_bits;
    // The sort ping-pongs between the two buffer pairs; start in the pair that
    // makes the last pass land in keys_sorted / vals_sorted.
    const bool odd = fdgs_radix_sort_passes(end_bit) & 1;
    uint64_t* ka = odd ? a->keys_unsorted : a->keys_sorted;
    uint32_t* va = odd ? a->vals_unsorted : a->vals_sorted;
    uint64_t* kb = odd ? a->keys_sorted : a->keys_unsorted;
    uint32_t* vb = odd ? a->vals_sorted : a->vals_unsorted;
    hipLaunchKernelGGL(duplicate_with_keys_kernel, dim3((a->num_points + 255) / 256), dim3(256), 0, s, a->num_points,
                       a->geom, a->point_offsets, ka, va, gx, gy, depth_bits, depth_min, a->ranges);
    FDGS_CHECK_LAUNCH();
    int rc = fdgs_radix_sort_pingpong(ka, va, kb, vb, R, end_bit, a->sort_workspace, a->sort_workspace_bytes, s,
                                      depth_bits < 32 || depth_min != 0 ? depth_bits : 0, depth_min);
    if (rc != FDGS_OK) return rc;
    hipLaunchKernelGGL(identify_tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, R,
                       a->keys_sorted, a->ranges);
    FDGS_CHECK_LAUNCH();
    if (a->tile_order) {
        hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, a->ranges, a->tile_order, tiles);
        FDGS_CHECK_LAUNCH();
    }
    return FDGS_OK;
}
