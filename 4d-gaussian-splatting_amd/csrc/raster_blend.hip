// Rasterizer stage 6 (front-to-back alpha blending) and stage 7 (its backward).
//
// gfx950 mapping: one 256-thread workgroup per 16x16 tile, but each of its four
// wavefronts owns one 8x8 pixel quadrant and walks the tile's depth-sorted list
// on its own, with no workgroup barrier.  A batch of 64 list entries is loaded
// one per lane; every lane tests whether its entry's 1/255-alpha ellipse can
// reach the wave's quadrant, a 64-bit ballot collects the survivors and the wave
// iterates over the set bits in list order.  The survivor's record reaches all lanes
// through a wave-private LDS copy of the batch (three 16-byte stores per lane per batch,
// same-address reads per survivor; other waves of the SIMD cover the read's latency - issuing
// the next survivor's reads ahead cost more in register copies than it hid): both kernels are bound by vector-instruction issue, a v_readlane
// per field is a vector instruction and an LDS read is not.  No workgroup barrier: a
// wave's LDS instructions execute in program order.  Entries that cannot reach the
// quadrant cost nothing; a quadrant stops as soon as its 64 pixels are saturated.
#include "fdgs_common.h"

#define BL_REC 2                                   // float4 per staged record: (x y A B) (C o r g); + one float2

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ void __launch_bounds__(256)
blend_fwd_kernel(fdgs_blend_fwd_args a, int tiles_x) {
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qx0 = tx * FDGS_TILE + (wave & 1) * 8, qy0 = ty * FDGS_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const int W = a.image_width, H = a.image_height;
    const bool inside = pxi < W && pyi < H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float rx0 = (float)qx0, ry0 = (float)qy0, rx1 = (float)(qx0 + 7), ry1 = (float)(qy0 + 7);

    const uint32_t r_begin = a.ranges[2 * tile], r_end = a.ranges[2 * tile + 1];
    const float4* __restrict__ geom4 = reinterpret_cast<const float4*>(a.geom);

    __shared__ float4 staged[4][64 * BL_REC];
    __shared__ float2 staged2[4][64];                       // forward: (b, depth)
    float4* __restrict__ my = staged[wave];
    float2* __restrict__ my2 = staged2[wave];

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, D = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t base = r_begin; base < r_end; base += 64) {
        if (__ballot(!done) == 0ull) break;
        const uint32_t e = base + lane;
        const bool valid = e < r_end;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        if (valid) {
            const uint32_t id = a.point_list[e];
            g0 = geom4[3 * (size_t)id + 0];
            g1 = geom4[3 * (size_t)id + 1];
            g2 = geom4[3 * (size_t)id + 2];
        }
        my[lane * BL_REC + 0] = g0;
        my[lane * BL_REC + 1] = g1;
        my2[lane] = make_float2(g2.x, g2.y);
        const bool hit = valid && fdgs_reaches(g0.x, g0.y, __float_as_uint(g2.w), rx0, ry0, rx1, ry1);
        unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        while (true) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const bool more = mask != 0ull;
            const float4 c0 = my[j * BL_REC + 0], c1 = my[j * BL_REC + 1];
            const float2 c2 = my2[j];
            const float ex = c0.x, ey = c0.y, A = c0.z, B = c0.w, C = c1.x, o = c1.y;
            const float dx = ex - pxf, dy = ey - pyf;
            const float power = -0.5f * (A * dx * dx + C * dy * dy) - B * dx * dy;
            const float alpha = fminf(FDGS_ALPHA_MAX, o * __expf(power));
            if (!done && power <= 0.0f && alpha >= FDGS_ALPHA_MIN) {
                const float test_T = T * (1.0f - alpha);
                if (test_T < FDGS_T_STOP) {
                    done = true;
                } else {
                    const float w = alpha * T;
                    Cr += c1.z * w;
                    Cg += c1.w * w;
                    Cb += c2.x * w;
                    D += c2.y * w;
                    T = test_T;
                    last = (base - r_begin) + (uint32_t)j + 1u;
                }
            }
            if (!more || __ballot(!done) == 0ull) break;
        }
    }
    if (inside) {
        const size_t pix = (size_t)pyi * W + pxi, plane = (size_t)W * H;
        a.out_color[pix] = Cr + T * a.bg[0];
        a.out_color[plane + pix] = Cg + T * a.bg[1];
        a.out_color[2 * plane + pix] = Cb + T * a.bg[2];
        a.out_depth[pix] = D;
        a.final_T[pix] = T;
        a.n_contrib[pix] = last;
    }
}

// ---------------------------------------------------------------------------
// Backward
// ---------------------------------------------------------------------------
// Wave-wide sum; the total is returned from lane 63 (as a wave-uniform value).
// Row-level Hillis-Steele steps with row_shr DPP, then the two cross-row
// broadcasts (row_bcast:15, row_bcast:31) of the GCN/CDNA DPP set.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return bcast(v, 63);
}

// x + (the other half-wave's y) arranged so that lanes 0-31 return sum-pairs of a, lanes 32-63 of b
__device__ __forceinline__ float xswap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// same between the 16-lane rows of each half: even rows return pairs of a, odd rows pairs of b
__device__ __forceinline__ float xswap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float ror8(float v) {   // lane i reads lane i^8 of its row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
}

__global__ void __launch_bounds__(256)
blend_bwd_kernel(fdgs_blend_bwd_args a, int tiles_x) {
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qx0 = tx * FDGS_TILE + (wave & 1) * 8, qy0 = ty * FDGS_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const int W = a.image_width, H = a.image_height;
    const bool inside = pxi < W && pyi < H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float rx0 = (float)qx0, ry0 = (float)qy0, rx1 = (float)(qx0 + 7), ry1 = (float)(qy0 + 7);

    const uint32_t r_begin = a.ranges[2 * tile], r_end = a.ranges[2 * tile + 1];
    if (r_begin >= r_end) return;
    const float4* __restrict__ geom4 = reinterpret_cast<const float4*>(a.geom);

    float T_final = 1.0f, dLr = 0.f, dLg = 0.f, dLb = 0.f;
    int ncon = 0;
    if (inside) {
        const size_t pix = (size_t)pyi * W + pxi, plane = (size_t)W * H;
        T_final = a.final_T[pix];
        ncon = (int)a.n_contrib[pix];
        dLr = a.dL_dcolor[pix];
        dLg = a.dL_dcolor[plane + pix];
        dLb = a.dL_dcolor[2 * plane + pix];
    }
    const float bg_dot = a.bg[0] * dLr + a.bg[1] * dLg + a.bg[2] * dLb;
    float T = T_final;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, last_alpha = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f;

    int wave_last = ncon;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, d));
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);

    __shared__ float4 staged[4][64 * BL_REC];
    __shared__ float2 staged2[4][64];                       // backward: (b, Gaussian index)
    float4* __restrict__ my = staged[wave];
    float2* __restrict__ my2 = staged2[wave];

    for (int hi = wave_last; hi > 0; hi -= 64) {
        const int pos = hi - 1 - lane;        // 0-based position in the tile list, back to front
        const bool valid = pos >= 0;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
        float2 g2 = make_float2(0.f, 0.f);
        uint32_t reach = 0xfc00fc00u;         // (-inf, -inf): reaches nothing
        if (valid) {
            const uint32_t id = a.point_list[r_begin + (uint32_t)pos];
            g0 = geom4[3 * (size_t)id + 0];
            g1 = geom4[3 * (size_t)id + 1];
            g2.x = a.geom[FDGS_GEOM_FLOATS * (size_t)id + 8];
            reach = __float_as_uint(a.geom[FDGS_GEOM_FLOATS * (size_t)id + 11]);
            g2.y = __uint_as_float(id);       // the Gaussian's index travels with the record
        }
        my[lane * BL_REC + 0] = g0;
        my[lane * BL_REC + 1] = g1;
        my2[lane] = g2;
        const bool hit = valid && fdgs_reaches(g0.x, g0.y, reach, rx0, ry0, rx1, ry1);
        unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        while (true) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const bool more = mask != 0ull;
            const float4 c0 = my[j * BL_REC + 0], c1 = my[j * BL_REC + 1];
            const float2 c2 = my2[j];
            const int p = hi - 1 - j;
            const float ex = c0.x, ey = c0.y, A = c0.z, B = c0.w, C = c1.x, o = c1.y;
            const float dx = ex - pxf, dy = ey - pyf;
            const float power = -0.5f * (A * dx * dx + C * dy * dy) - B * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(FDGS_ALPHA_MAX, o * G);
            const bool contrib = (p < ncon) && power <= 0.0f && alpha >= FDGS_ALPHA_MIN;
            if (__ballot(contrib) != 0ull) {
                // No divergent branch: a lane that does not contribute runs the same arithmetic with alpha = 0 and
                // G = 0, which leaves its state unchanged bit for bit (T * rcp(1) = T; the pending (last_alpha,
                // last colour) pair is folded into acc now instead of at the next contributor - the same single
                // expression on the same operands - and the pair it leaves, alpha 0, folds as 0 * c + 1 * acc = acc)
                // and makes all nine of its partial sums exact zeros.
                const float al = contrib ? alpha : 0.0f;
                const float Ge = contrib ? G : 0.0f;
                const float cr = c1.z, cg = c1.w, cb = c2.x;
                const float inv_one_m = __builtin_amdgcn_rcpf(1.0f - al);   // 1 ulp; alpha <= 0.99
                T = T * inv_one_m;
                const float w = al * T;
                acc_r = last_alpha * last_r + (1.0f - last_alpha) * acc_r;
                acc_g = last_alpha * last_g + (1.0f - last_alpha) * acc_g;
                acc_b = last_alpha * last_b + (1.0f - last_alpha) * acc_b;
                last_r = cr; last_g = cg; last_b = cb;
                float dL_dalpha = (cr - acc_r) * dLr + (cg - acc_g) * dLg + (cb - acc_b) * dLb;
                const float v_r = w * dLr, v_g = w * dLg, v_b = w * dLb;
                dL_dalpha *= T;
                last_alpha = al;
                dL_dalpha += (-T_final * inv_one_m) * bg_dot;
                const float dL_dG = o * dL_dalpha;
                const float gdx = Ge * dx, gdy = Ge * dy;
                const float v_mx = dL_dG * (-gdx * A - gdy * B);
                const float v_my = dL_dG * (-gdy * C - gdx * B);
                const float v_A = -0.5f * gdx * dx * dL_dG;
                const float v_B = -gdx * dy * dL_dG;
                const float v_C = -0.5f * gdy * dy * dL_dG;
                const float v_o = Ge * dL_dalpha;
                // Eight of the nine partial sums go through a transposing butterfly: each exchange step halves
                // the number of live registers (v_permlane32_swap / v_permlane16_swap trade one register's upper
                // half for another's lower half; row_ror:8 does the same inside a 16-lane row), so that the 8-lane
                // group g = lane>>3 ends up owning the total of value g.  The ninth takes the plain DPP reduction.
                float t0 = xswap32_add(v_mx, v_C), t1 = xswap32_add(v_my, v_o);   // lanes <32: v[i], >=32: v[i+4]
                float t2 = xswap32_add(v_A, v_r), t3 = xswap32_add(v_B, v_g);
                const float u0 = xswap16_add(t0, t2), u1 = xswap16_add(t1, t3);  // rows with bit4: v[i+2]
                const float a8 = u0 + ror8(u0), b8 = u1 + ror8(u1);
                float r = (lane & 8) ? b8 : a8;                                   // bit3: v[i+1]
                r = dpp_add<0x111, 0xf>(r);                                       // sum of each 8-lane group ...
                r = dpp_add<0x112, 0xf>(r);
                r = dpp_add<0x114, 0xf>(r);                                       // ... lands in lanes 8g+7
                const float s8 = wave_sum(v_b);
                const uint32_t gid = __float_as_uint(c2.y);
                if ((lane & 7) == 7 || lane == 0) {
                    const int comp = lane == 0 ? 8 : (lane >> 3);
                    atomicAdd(a.dL_dgeom + (size_t)gid * FDGS_GEOM_FLOATS + comp, lane == 0 ? s8 : r);
                }
            }
            if (!more) break;
        }
    }
}

extern "C" int fdgs_raster_blend_fwd(const fdgs_blend_fwd_args* a, void* stream) {
    if (!a || a->image_width <= 0 || a->image_height <= 0) return FDGS_EINVAL;
    if (!a->ranges || !a->bg || !a->out_color || !a->out_depth || !a->final_T || !a->n_contrib)
        return FDGS_EINVAL;
    // point_list may be NULL when every range is empty, and then (a scene without Gaussians) so may geom
    if (!a->geom && a->point_list) return FDGS_EINVAL;
    const int gx = fdgs_tiles_x(a->image_width), gy = fdgs_tiles_y(a->image_height);
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(gx * gy), dim3(256), 0, (hipStream_t)stream, *a, gx);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}

extern "C" int fdgs_raster_blend_bwd(const fdgs_blend_bwd_args* a, void* stream) {
    if (!a || a->image_width <= 0 || a->image_height <= 0) return FDGS_EINVAL;
    if (!a->ranges || !a->bg || !a->final_T || !a->n_contrib || !a->dL_dcolor) return FDGS_EINVAL;
    if ((!a->geom || !a->dL_dgeom) && a->point_list) return FDGS_EINVAL;   // both may be NULL with no Gaussians
    const int gx = fdgs_tiles_x(a->image_width), gy = fdgs_tiles_y(a->image_height);
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(gx * gy), dim3(256), 0, (hipStream_t)stream, *a, gx);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
