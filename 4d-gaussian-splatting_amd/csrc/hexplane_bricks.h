// HexPlane backward over spatial bricks (included by deform.hip after the flat scatter kernel).
//
// The flat kernel sends every bilinear corner of every point to memory as a float atomic: 24 x 16
// atomics per point and level pair, which is what bounds it.  Here the points are first binned into
// nb^3 bricks of the normalised box (counting sort: histogram, plan, scatter).  A workgroup then takes up
// to BRICK_CH points of ONE brick; every texel those points can touch lies in a small window of each
// plane (about res/nb + 2 texels per axis), so the window lives in LDS, corners are accumulated with LDS
// atomics and each window is flushed to memory once per workgroup.  A corner that falls outside the
// window (rounding at a brick face) goes to memory directly, so the result does not depend on how exactly
// the binning agrees with the texel arithmetic.
//
// The windows are float64: on gfx950 an LDS float32 atomic add retires about one lane every three cycles
// per CU, a float64 one at least twelve times as many (measured), and the sums come out more accurate.
// With one time stamp for the whole call every point has the same two time rows and the same time
// fraction, so a time plane needs one window row (the corner weights without their time factor); the two
// plane rows are that row times (1 - ft) and ft at the flush.
#pragma once

#define BRICK_MAX 4096               // bricks per call: at most 16^3
#define BRICK_CH 1024                // points per work item (one workgroup)
#define BRICK_THREADS 1024           // workgroup of the scatter kernel: 16 waves, four per SIMD (<= 128 VGPRs)
#define BRICK_SORT_PTS 4096          // points per workgroup of the binning kernels (16 per thread)

struct BrickBinK {
    int n, nb[3];                    // bricks per axis
    const float* xyz; const float* aabb;
    unsigned short* brick_of;        // [n]
    int* counts;                     // [nb^3], zero on entry of the histogram
    int* cursor;                     // [nb^3] running write position per brick
    int* perm;                       // [n] point indices grouped by brick
    int* work;                       // [max_work][3]: brick, begin, end (positions in perm)
    int* work_count;
};

__device__ __forceinline__ int brick_axis(float co, int nb) {
    float u = (co + 1.0f) * 0.5f * (float)nb;
    u = fminf(fmaxf(u, 0.0f), (float)(nb - 1));      // NaN -> 0
    return (int)u;
}

__global__ void __launch_bounds__(256)
brick_hist_kernel(BrickBinK a) {
    __shared__ int hist[BRICK_MAX];
    const int nbr = a.nb[0] * a.nb[1] * a.nb[2];
    for (int i = threadIdx.x; i < nbr; i += 256) hist[i] = 0;
    __syncthreads();
    const int p0 = blockIdx.x * BRICK_SORT_PTS;
#pragma unroll 4
    for (int u = 0; u < BRICK_SORT_PTS / 256; ++u) {
        const int p = p0 + u * 256 + threadIdx.x;
        if (p < a.n) {
            int bk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float s = 2.0f / (a.aabb[3 + k] - a.aabb[k]);
                bk[k] = brick_axis((a.xyz[3 * (size_t)p + k] - a.aabb[k]) * s - 1.0f, a.nb[k]);
            }
            const int b = (bk[2] * a.nb[1] + bk[1]) * a.nb[0] + bk[0];
            a.brick_of[p] = (unsigned short)b;
            atomicAdd(&hist[b], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbr; i += 256)
        if (hist[i]) atomicAdd(a.counts + i, hist[i]);
}

// one workgroup: exclusive scan of the brick counts, and the list of work items (a brick with c points
// becomes ceil(c / BRICK_CH) items)
__global__ void __launch_bounds__(1024)
brick_plan_kernel(BrickBinK a) {
    __shared__ int s_cnt[1024], s_blk[1024];
    const int nbr = a.nb[0] * a.nb[1] * a.nb[2];
    const int per = (nbr + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    int cnt = 0, blk = 0;
    for (int i = 0; i < per; ++i)
        if (b0 + i < nbr) {
            const int c = a.counts[b0 + i];
            cnt += c;
            blk += (c + BRICK_CH - 1) / BRICK_CH;
        }
    s_cnt[threadIdx.x] = cnt; s_blk[threadIdx.x] = blk;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int c = 0, b = 0;
        if ((int)threadIdx.x >= d) { c = s_cnt[threadIdx.x - d]; b = s_blk[threadIdx.x - d]; }
        __syncthreads();
        s_cnt[threadIdx.x] += c; s_blk[threadIdx.x] += b;
        __syncthreads();
    }
    int off = s_cnt[threadIdx.x] - cnt, w = s_blk[threadIdx.x] - blk;      // exclusive
    for (int i = 0; i < per; ++i)
        if (b0 + i < nbr) {
            const int c = a.counts[b0 + i];
            a.cursor[b0 + i] = off;
            for (int q = 0; q < c; q += BRICK_CH) {
                a.work[3 * w + 0] = b0 + i;
                a.work[3 * w + 1] = off + q;
                a.work[3 * w + 2] = off + min(q + BRICK_CH, c);
                ++w;
            }
            off += c;
        }
    if (threadIdx.x == 1023) *a.work_count = s_blk[1023];
}

__global__ void __launch_bounds__(256)
brick_scatter_kernel(BrickBinK a) {
    __shared__ int hist[BRICK_MAX];
    const int nbr = a.nb[0] * a.nb[1] * a.nb[2];
    for (int i = threadIdx.x; i < nbr; i += 256) hist[i] = 0;
    __syncthreads();
    const int p0 = blockIdx.x * BRICK_SORT_PTS;
    int b[BRICK_SORT_PTS / 256], rank[BRICK_SORT_PTS / 256];
#pragma unroll
    for (int u = 0; u < BRICK_SORT_PTS / 256; ++u) {
        const int p = p0 + u * 256 + threadIdx.x;
        b[u] = -1; rank[u] = 0;
        if (p < a.n) {
            b[u] = a.brick_of[p];
            rank[u] = atomicAdd(&hist[b[u]], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbr; i += 256) {
        const int c = hist[i];
        if (c) hist[i] = atomicAdd(a.cursor + i, c);     // count -> first position of this workgroup's run
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BRICK_SORT_PTS / 256; ++u)
        if (b[u] >= 0) a.perm[hist[b[u]] + rank[u]] = p0 + u * 256 + threadIdx.x;
}

struct HexBrickK {
    HexBwdK h;
    int nb[3];
    int tw[2][3];                        // window width in texels per level and spatial axis
    const int* perm; const int* work; const int* work_count;
};

__global__ void __launch_bounds__(BRICK_THREADS)
hexplane_bwd_brick_kernel(HexBrickK k, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) double brick_lds[];
    double* lds = brick_lds;
    const HexBwdK& a = k.h;
    if ((int)blockIdx.x >= *k.work_count) return;
    const int brick = k.work[3 * blockIdx.x], begin = k.work[3 * blockIdx.x + 1], end = k.work[3 * blockIdx.x + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = lane & 15, sub = lane >> 4;
    for (int i = threadIdx.x; i < lds_floats; i += BRICK_THREADS) lds[i] = 0.0;
    // window origin per level and axis, LDS offsets of the windows: per level three spatial planes
    // (ci = 0, 1, 3) as [tw_b][tw_a][16] and, with one shared time stamp, three time planes (ci = 2, 4, 5)
    // as [tw_a][16]
    // Window origins per level and axis as named scalars, not an array: `h ? lo[1][i] : lo[0][i]` is turned into a
    // load through a selected pointer, i.e. a dynamic index, and the array then lives in scratch memory (a memory load
    // per use inside the point loop).
    int lo00, lo01, lo02, lo10, lo11, lo12;
    int base0[6], base1[6];
    auto lo_at = [&](bool upper, int ax) -> int {      // ax is a compile-time value at every call
        const int v0 = ax == 0 ? lo00 : ax == 1 ? lo01 : lo02, v1 = ax == 0 ? lo10 : ax == 1 ? lo11 : lo12;
        return upper ? v1 : v0;
    };
    {
        const int bk[3] = {brick % k.nb[0], (brick / k.nb[0]) % k.nb[1], brick / (k.nb[0] * k.nb[1])};
        int off = 0;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const int v = (bk[ax] * (a.ps.res[l][ax] - 1)) / k.nb[ax];
                if (l == 0) { if (ax == 0) lo00 = v; else if (ax == 1) lo01 = v; else lo02 = v; }
                else { if (ax == 0) lo10 = v; else if (ax == 1) lo11 = v; else lo12 = v; }
            }
#pragma unroll
            for (int ci = 0; ci < 6; ++ci) {
                const int ax = plane_a(ci), bx = plane_b(ci);
                if (l) base1[ci] = off; else base0[ci] = off;
                if (bx < 3) off += k.tw[l][ax] * k.tw[l][bx] * DF;
                else if (a.lds_time) off += k.tw[l][ax] * DF;
            }
        }
    }
    __syncthreads();
    const int n_items = end - begin;     // item = point: 16 lanes, the two levels in turn
    const int stride = (BRICK_THREADS / 64) * 4;
    // the point index and position of the next pass are fetched one pass ahead
    int p_next = 0;
    float xyz_next[3] = {0.f, 0.f, 0.f}, t_next = a.time_scalar;
    {
        const int item = wave * 4 + sub;
        p_next = k.perm[begin + (item < n_items ? item : 0)];
#pragma unroll
        for (int q = 0; q < 3; ++q) xyz_next[q] = a.xyz[3 * (size_t)p_next + q];
        if (a.time) t_next = a.time[p_next];
    }
    for (int ib = wave * 4; ib < n_items; ib += stride) {
        const int item = ib + sub;
        const bool valid = item < n_items;
        const int p = p_next;
        float co[4];
        float sc3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            sc3[q] = 2.0f / (a.aabb[3 + q] - a.aabb[q]);
            co[q] = (xyz_next[q] - a.aabb[q]) * sc3[q] - 1.0f;
        }
        co[3] = t_next;
        {
            const int nitem = item + stride;
            p_next = k.perm[begin + (nitem < n_items ? nitem : 0)];
#pragma unroll
            for (int q = 0; q < 3; ++q) xyz_next[q] = a.xyz[3 * (size_t)p_next + q];
            if (a.time) t_next = a.time[p_next];
        }
        float dco[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {                 // h stays wave-uniform: per-level selects are scalar
        AxisLookup axl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) axl[q] = make_axis(co[q], h ? a.ps.res[1][q] : a.ps.res[0][q]);
        float dfv = valid ? a.ws_df[(size_t)p * 32 + 16 * h + ch] : 0.0f;
        float tex[6][4];
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
            const int ax = plane_a(ci), bx = plane_b(ci);
            const int Wd = h ? a.ps.res[1][ax] : a.ps.res[0][ax];
            const float* __restrict__ pl = (h ? a.ps.pl[1][ci] : a.ps.pl[0][ci]) + ch;
            const int o00 = axl[bx].i0 * Wd + axl[ax].i0;
            const int oy = axl[bx].d * Wd;
            tex[ci][0] = pl[(size_t)o00 * DF];
            tex[ci][1] = pl[(size_t)(o00 + axl[ax].d) * DF];
            tex[ci][2] = pl[(size_t)(o00 + oy) * DF];
            tex[ci][3] = pl[(size_t)(o00 + oy + axl[ax].d) * DF];
        }
        float v[6], ddx[6], ddy[6];
        float feat = 1.0f;
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
            const float fx = axl[plane_a(ci)].f, fy = axl[plane_b(ci)].f;
            const float t00 = tex[ci][0], t01 = tex[ci][1], t10 = tex[ci][2], t11 = tex[ci][3];
            v[ci] = t00 * ((1.0f - fx) * (1.0f - fy)) + t01 * (fx * (1.0f - fy)) + t10 * ((1.0f - fx) * fy) + t11 * (fx * fy);
            ddx[ci] = (t01 - t00) * (1.0f - fy) + (t11 - t10) * fy;
            ddy[ci] = (t10 - t00) * (1.0f - fx) + (t11 - t01) * fx;
            feat *= v[ci];
        }
        if (valid && a.ws_feat) a.ws_feat[(size_t)p * 32 + 16 * h + ch] = feat;
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
            const int ax = plane_a(ci), bx = plane_b(ci);
            const int Wd = h ? a.ps.res[1][ax] : a.ps.res[0][ax];
            const int Hd = h ? a.ps.res[1][bx] : a.ps.res[0][bx];
            float g = dfv;
#pragma unroll
            for (int cj = 0; cj < 6; ++cj)
                if (cj != ci) g *= v[cj];
            // d(texel coordinate)/d(normalised coordinate) = 0.5*(size-1) where strictly inside
            if (axl[ax].in) dco[ax] += g * ddx[ci] * (0.5f * (float)(Wd - 1));
            if (bx < 3 && axl[bx].in) dco[bx] += g * ddy[ci] * (0.5f * (float)(Hd - 1));
            if (valid) {
                const float fx = axl[ax].f, fy = axl[bx].f;
                const int dx = axl[ax].d, dy = axl[bx].d;
                const int twa = h ? k.tw[1][ax] : k.tw[0][ax];
                const int lx = axl[ax].i0 - lo_at(h != 0, ax);
                bool in_lds = lx >= 0 && lx + dx < twa;
                int cell = lx;
                if (bx < 3) {
                    const int twb = h ? k.tw[1][bx] : k.tw[0][bx];
                    const int ly = axl[bx].i0 - lo_at(h != 0, bx);
                    in_lds = in_lds && ly >= 0 && ly + dy < twb;
                    cell += ly * twa;
                } else {
                    in_lds = in_lds && a.lds_time;
                }
                if (in_lds) {
                    double* L = lds + (h ? base1[ci] : base0[ci]) + cell * DF + ch;
                    if (bx < 3) {
                        atomicAdd(L, (double)(g * ((1.0f - fx) * (1.0f - fy))));
                        atomicAdd(L + dx * DF, (double)(g * (fx * (1.0f - fy))));
                        atomicAdd(L + dy * twa * DF, (double)(g * ((1.0f - fx) * fy)));
                        atomicAdd(L + (dy * twa + dx) * DF, (double)(g * (fx * fy)));
                    } else {
                        atomicAdd(L, (double)(g * (1.0f - fx)));
                        atomicAdd(L + dx * DF, (double)(g * fx));
                    }
                } else {
                    float* __restrict__ gp = (h ? a.gpl[1][ci] : a.gpl[0][ci]) + ch;
                    const int o00 = axl[bx].i0 * Wd + axl[ax].i0;
                    const int oy = dy * Wd;
                    atomicAdd(gp + (size_t)o00 * DF, g * ((1.0f - fx) * (1.0f - fy)));
                    atomicAdd(gp + (size_t)(o00 + dx) * DF, g * (fx * (1.0f - fy)));
                    atomicAdd(gp + (size_t)(o00 + oy) * DF, g * ((1.0f - fx) * fy));
                    atomicAdd(gp + (size_t)(o00 + oy + dx) * DF, g * (fx * fy));
                }
            }
        }
        }
        // sum over channels (DPP row); the two levels were added in dco
        const float dx = row_sum15(dco[0] * sc3[0]), dy = row_sum15(dco[1] * sc3[1]), dz = row_sum15(dco[2] * sc3[2]);
        if (valid && ch == 15) {
            a.d_xyz[3 * (size_t)p + 0] = a.g_xyz[3 * (size_t)p + 0] + dx;
            a.d_xyz[3 * (size_t)p + 1] = a.g_xyz[3 * (size_t)p + 1] + dy;
            a.d_xyz[3 * (size_t)p + 2] = a.g_xyz[3 * (size_t)p + 2] + dz;
        }
    }
    __syncthreads();
    // flush the windows
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        float tx; bool tin;
        unnorm_clip(a.time_scalar, a.ps.res[l][3], tx, tin);
        const int ty0 = (int)floorf(tx), ty1 = min(ty0 + 1, a.ps.res[l][3] - 1);
        const double ft = (double)(tx - floorf(tx));
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
            const int ax = plane_a(ci), bx = plane_b(ci);
            if (bx == 3 && !a.lds_time) continue;
            const int twa = k.tw[l][ax], Wd = a.ps.res[l][ax];
            const int rows = bx < 3 ? k.tw[l][bx] : 1;
            const int count = rows * twa * DF;
            const double* L = lds + (l ? base1[ci] : base0[ci]);
            float* gp = a.gpl[l][ci];
            for (int i = threadIdx.x; i < count; i += BRICK_THREADS) {
                const double val = L[i];
                if (val != 0.0) {
                    const int t = i >> 4, c = i & 15;
                    const int ly = t / twa, lx = t - ly * twa;
                    if (bx < 3) {
                        atomicAdd(gp + ((size_t)(lo_at(l != 0, bx) + ly) * Wd + lo_at(l != 0, ax) + lx) * DF + c, (float)val);
                    } else {
                        atomicAdd(gp + ((size_t)ty0 * Wd + lo_at(l != 0, ax) + lx) * DF + c, (float)(val * (1.0 - ft)));
                        atomicAdd(gp + ((size_t)ty1 * Wd + lo_at(l != 0, ax) + lx) * DF + c, (float)(val * ft));
                    }
                }
            }
        }
    }
}
