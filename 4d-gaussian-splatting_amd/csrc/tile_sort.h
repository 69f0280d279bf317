// Stages 3-5 as a segmented sort: the tile lists are built by BINNING the (Gaussian, tile) pairs by tile and then
// sorting every tile's segment by depth inside LDS, instead of sorting all R (tile | depth) keys with a
// least-significant-digit radix sort (radix_sort.hip: five passes over R 12-byte pairs, three launches each).
//
//   1. tile_walk_kernel<false>  each workgroup walks a contiguous chunk of the Gaussians and counts the pairs per
//                               tile in an LDS histogram -> table[workgroup][tile]
//   2. tile_table_scan_kernel   per tile: exclusive prefix over the workgroups, and the tile's total
//   3. tile_walk_kernel<true>   every workgroup scans the totals for itself (first list position of every tile; the
//                               first one also writes them and the ranges), then the same walk again: a pair's slot
//                               is an LDS atomic on the workgroup's cursor of that tile, and
//                               (depth bits << 32 | Gaussian id) goes to binned[slot].  One more workgroup orders the
//                               tile ids by decreasing list length meanwhile.
//   4. tile_sort_kernel         one wave per tile (workgroups of several waves where the lists are long) sorts the
//                               segment in LDS and writes the sorted keys (tile << 32 | depth bits) and values (id); a
//                               second launch of a few large workgroups takes the segments that are too long for
//                               the first
//
// Steps 1 and 2 need neither the list length R nor the R-sized buffers: a caller may run them (phase
// FDGS_BINSORT_COUNT) while R is on its way to the host and the rest afterwards (FDGS_BINSORT_FINISH).
//
// The result is the one the stable radix sort gives: duplicateWithKeys emits the pairs in Gaussian order, so among
// equal (tile, depth) keys the stable sort leaves the ids ascending - which is the order of the 64-bit numbers
// (depth bits << 32 | id).  Within a tile every id occurs once, so the numbers are distinct and the order in which
// step 3 fills a segment (LDS atomics: not reproducible) does not matter.
//
// Traffic: the walks read 32 bytes per Gaussian twice and write 8 bytes per pair; the sort reads 8 and writes 12
// bytes per pair: 28 R against the radix sort's 24 R per pass.
// Included by raster_preprocess.hip (one translation unit with duplicateWithKeys and tile_order_block, built
// without multiply-add contraction: the tile rectangles are recomputed here).
#pragma once

#define TS_WALK_THREADS 1024
#define TS_WALK_WAVES (TS_WALK_THREADS / 64)
#define TS_MAX_TILES FDGS_TILE_SORT_MAX_TILES       // LDS histogram of the walks: 4 bytes per tile
#define TS_TABLE_WORDS FDGS_TILE_SORT_TABLE_WORDS   // table[workgroups][tiles]
#define TS_MAX_WALK_GROUPS 512
#define TS_UNROLL 4

struct ts_plan {
    int groups;          // workgroups of the walks
    int per_group;       // Gaussians per workgroup, a multiple of TS_WALK_THREADS: every wave the same number of 64s
};

static bool ts_make_plan(int n, int tiles, ts_plan* p) {
    if (tiles > TS_MAX_TILES || n <= 0) return false;
    int groups = (int)(TS_TABLE_WORDS / (unsigned)tiles);
    if (groups > TS_MAX_WALK_GROUPS) groups = TS_MAX_WALK_GROUPS;
    if (groups < 1) return false;
    int per = (n + groups - 1) / groups;
    per = (per + TS_WALK_THREADS - 1) / TS_WALK_THREADS * TS_WALK_THREADS;
    p->per_group = per;
    p->groups = (n + per - 1) / per;
    return true;
}

// floor(q / w) for q < 2^24, w >= 1: the float quotient is off by at most one.
__device__ __forceinline__ uint32_t ts_div(uint32_t q, uint32_t w) {
    uint32_t r = (uint32_t)((float)q * __builtin_amdgcn_rcpf((float)w));
    if (r * w > q) --r;
    else if ((r + 1) * w <= q) ++r;
    return r;
}

// Exclusive scan of totals[0, tiles) into slot[] (LDS) by one workgroup of TS_WALK_THREADS threads; optionally also
// to tile_start[] and, as (first, last + 1) or (0, 0) for an empty tile, to ranges[].  wsum: one word per wave.
__device__ __forceinline__ void ts_scan_totals(uint32_t* __restrict__ slot, const uint32_t* __restrict__ totals,
                                               int tiles, uint32_t* __restrict__ wsum,
                                               uint32_t* __restrict__ tile_start, uint32_t* __restrict__ ranges) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (tiles + TS_WALK_THREADS - 1) / TS_WALK_THREADS;      // <= 32 consecutive tiles per thread
    for (int t = threadIdx.x; t < tiles; t += TS_WALK_THREADS) slot[t] = totals[t];
    __syncthreads();
    const int t0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (int k = 0; k < per; ++k)
        if (t0 + k < tiles) sum += slot[t0 + k];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t pos = incl - sum;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
    for (int k = 0; k < per; ++k) {
        const int t = t0 + k;
        if (t < tiles) {
            const uint32_t c = slot[t];
            if (tile_start) {
                tile_start[t] = pos;
                ranges[2 * t] = c ? pos : 0u;
                ranges[2 * t + 1] = c ? pos + c : 0u;
            }
            slot[t] = pos;
            pos += c;
        }
    }
}

// The walk of duplicateWithKeys (raster_preprocess.hip): a wave takes 64 consecutive Gaussians and visits their list
// positions together, lane l the positions first + l, first + l + 64, ...; the owner of a position is found by binary
// search over the wave's 64 scan values in LDS, so a large splat is shared by all lanes instead of holding one.
// SCATTER: one workgroup more than there are chunks: the last one orders the tiles.
template <bool SCATTER>
__global__ void __launch_bounds__(TS_WALK_THREADS)
tile_walk_kernel(int n, int per_group, const float* __restrict__ geom, const uint32_t* __restrict__ offsets, int gx,
                 int gy, uint32_t* __restrict__ table, uint32_t* __restrict__ tile_start,
                 uint64_t* __restrict__ binned, const uint32_t* __restrict__ totals, uint32_t* __restrict__ order,
                 uint32_t* __restrict__ ranges, uint4* __restrict__ desc) {
    extern __shared__ uint32_t ts_lds[];
    const int tiles = gx * gy;
    uint32_t* __restrict__ slot = ts_lds;                                 // [tiles] count, or next free position
    uint32_t* __restrict__ s_end = ts_lds + tiles;                        // [waves][64] each
    uint32_t* __restrict__ s_start = s_end + TS_WALK_WAVES * 64;
    uint32_t* __restrict__ s_x0w = s_start + TS_WALK_WAVES * 64;
    uint32_t* __restrict__ s_y0 = s_x0w + TS_WALK_WAVES * 64;
    uint32_t* __restrict__ s_depth = s_y0 + TS_WALK_WAVES * 64;
    uint32_t* __restrict__ s_wsum = s_depth + TS_WALK_WAVES * 64;         // [waves]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (SCATTER && blockIdx.x == gridDim.x - 1) {
        // first positions in slot[], then the tile ids by decreasing length with (tile, first position, length) each
        ts_scan_totals(slot, totals, tiles, s_wsum, nullptr, nullptr);
        __syncthreads();
        tile_order_block<TS_WALK_THREADS>(nullptr, totals, order, tiles, s_end, s_end + ORDER_CLASSES, slot, desc);
        return;
    }
    // SCATTER: consecutive chunks go to workgroups of one XCD (workgroup j runs on XCD j mod 8).  A tile's segment is
    // filled chunk after chunk, so each part of it is then written from one XCD and its 8-byte stores meet in that
    // XCD's L2 as whole lines; spread over all XCDs, every line is written back in pieces by up to eight L2s.
    // (Only the speed depends on where workgroups run.)
    int chunk = (int)blockIdx.x;
    if (SCATTER) {
        const int G = (int)gridDim.x - 1, x = chunk & 7;
        chunk = x * (G >> 3) + min(x, G & 7) + (chunk >> 3);
    }
    uint32_t* __restrict__ row = table + (size_t)chunk * tiles;

    // the wave's first group of Gaussians is requested before the histogram is set up
    const int begin = chunk * per_group;
    const int end_g = min(n, begin + per_group);
    const int wo = wave * 64;
    int g0 = begin + wo;
    uint32_t end = 0, start = 0;
    float4 g2 = make_float4(0.f, 0.f, 0.f, 0.f), gA = g2;
    if (g0 < end_g) {
        const int i = g0 + lane, ic = i < n ? i : n - 1;
        end = offsets[ic];
        start = i < n ? (i == 0 ? 0u : offsets[i - 1]) : end;
        g2 = reinterpret_cast<const float4*>(geom)[3 * (size_t)ic + 2];
        gA = reinterpret_cast<const float4*>(geom)[3 * (size_t)ic + 0];
    }

    uint32_t rv[8];                                 // SCATTER: this chunk's row of the table (its first 8 x 1024 tiles), requested early
    if (SCATTER) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = (int)threadIdx.x + k * TS_WALK_THREADS;
            rv[k] = t < tiles ? row[t] : 0u;
        }
    }
    if (!SCATTER) {
        for (int t = threadIdx.x; t < tiles; t += TS_WALK_THREADS) slot[t] = 0u;
    } else {
        // slot[t] = first position of tile t (exclusive scan of the totals; the first chunk's workgroup also writes
        // them and the ranges) + the pairs of the chunks before this one
        ts_scan_totals(slot, totals, tiles, s_wsum, chunk == 0 ? tile_start : nullptr, chunk == 0 ? ranges : nullptr);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = (int)threadIdx.x + k * TS_WALK_THREADS;
            if (t < tiles) slot[t] += rv[k];
        }
        for (int t = (int)threadIdx.x + 8 * TS_WALK_THREADS; t < tiles; t += TS_WALK_THREADS) slot[t] += row[t];
    }
    __syncthreads();

    for (; g0 < end_g; g0 += TS_WALK_THREADS) {
        uint32_t x0w = 0, y0v = 0;
        if (end > start) {
            const int radius = fdgs_unpack_radius(__float_as_int(g2.z));
            int x0, y0, x1, y1;
            fdgs_tile_rect(gA.x, gA.y, (float)radius, gx, gy, x0, y0, x1, y1);
            x0w = (uint32_t)x0 | ((uint32_t)(x1 - x0) << 16);
            y0v = (uint32_t)y0;
        }
        __builtin_amdgcn_wave_barrier();            // the previous group's reads come first (a wave's LDS accesses are in order)
        s_end[wo + lane] = end;
        s_start[wo + lane] = start;
        s_x0w[wo + lane] = x0w;
        s_y0[wo + lane] = y0v;
        if (SCATTER) s_depth[wo + lane] = __float_as_uint(g2.y);
        __builtin_amdgcn_wave_barrier();
        const uint32_t first = s_start[wo], last = s_end[wo + 63];
        const int g_now = g0;
        // the next group's loads are in flight during this group's walk
        const int g_next = g0 + TS_WALK_THREADS;
        if (g_next < end_g) {
            const int i = g_next + lane, ic = i < n ? i : n - 1;
            end = offsets[ic];
            start = i < n ? offsets[i - 1] : end;       // i > 0 here
            g2 = reinterpret_cast<const float4*>(geom)[3 * (size_t)ic + 2];
            gA = reinterpret_cast<const float4*>(geom)[3 * (size_t)ic + 0];
        }
        // TS_UNROLL positions per lane and trip: their searches are independent chains of LDS reads that overlap
        for (uint32_t pos0 = first + lane; pos0 < last; pos0 += 64 * TS_UNROLL) {
            uint32_t tile[TS_UNROLL], own[TS_UNROLL];
            bool ok[TS_UNROLL];
#pragma unroll
            for (int u = 0; u < TS_UNROLL; ++u) {
                const uint32_t pos_u = pos0 + 64u * (uint32_t)u;
                ok[u] = pos_u < last;
                const uint32_t pos = ok[u] ? pos_u : pos0;      // a position of the group, so the search stays inside it
                int l = 0;                                      // the first Gaussian of the wave whose end lies beyond pos
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1)
                    if (s_end[wo + l + step - 1] <= pos) l += step;
                const uint32_t q = pos - s_start[wo + l];
                const uint32_t xw = s_x0w[wo + l];
                const uint32_t w = xw >> 16;
                const uint32_t r = ts_div(q, w);
                tile[u] = (s_y0[wo + l] + r) * (uint32_t)gx + (xw & 0xffffu) + (q - r * w);
                own[u] = (uint32_t)l;
            }
            if (SCATTER) {
                uint32_t dst[TS_UNROLL];
#pragma unroll
                for (int u = 0; u < TS_UNROLL; ++u)
                    if (ok[u]) dst[u] = atomicAdd(&slot[tile[u]], 1u);
#pragma unroll
                for (int u = 0; u < TS_UNROLL; ++u)
                    if (ok[u])
                        binned[dst[u]] = ((uint64_t)s_depth[wo + own[u]] << 32) | (uint64_t)((uint32_t)g_now + own[u]);
            } else {
#pragma unroll
                for (int u = 0; u < TS_UNROLL; ++u)
                    if (ok[u]) atomicAdd(&slot[tile[u]], 1u);
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (int t = threadIdx.x; t < tiles; t += TS_WALK_THREADS) row[t] = slot[t];
    }
}

// table[b][t] -> exclusive prefix over b, totals[t].  A workgroup takes 32 consecutive tiles, 32 threads per tile each
// sum one segment of the b axis (at most TS_SCAN_PER loads, all independent: a single thread walking all workgroups
// of its tile would wait for memory once per workgroup).
#define TS_SCAN_THREADS 1024
#define TS_SCAN_PER (TS_MAX_WALK_GROUPS / 32)
__global__ void __launch_bounds__(TS_SCAN_THREADS)
tile_table_scan_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ totals, int tiles, int groups) {
    __shared__ uint32_t seg_sum[32][33];
    const int tl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tl;
    const int per_seg = (groups + 31) / 32;              // <= TS_SCAN_PER
    const int b0 = seg * per_seg;
    uint32_t v[TS_SCAN_PER];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < TS_SCAN_PER; ++k) {
        const int b = b0 + k;
        v[k] = (k < per_seg && b < groups && t < tiles) ? table[(size_t)b * tiles + t] : 0u;
        sum += v[k];
    }
    seg_sum[seg][tl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const uint32_t x = seg_sum[k][tl];
        if (k < seg) run += x;
        total += x;
    }
#pragma unroll
    for (int k = 0; k < TS_SCAN_PER; ++k) {
        const int b = b0 + k;
        if (k < per_seg && b < groups && t < tiles) table[(size_t)b * tiles + t] = run;
        run += v[k];
    }
    if (seg == 0 && t < tiles) totals[t] = total;
}

// ---------------------------------------------------------------------------------------------------------------
// Sorting one segment.
//
// In LDS the segment is sorted as 32-bit words (depth field << lp | position in the segment), P = 2^lp the power of
// two >= n (at least 8), the depth field = the upper 32 - lp bits of (depth bits - the segment's smallest) within the
// segment's depth range: one compare-exchange is a v_min_u32 and a v_max_u32, where the 64-bit numbers need a 64-bit
// compare and four selects.  The words order the entries by depth field; entries that share a depth field (few: the
// field keeps 20-26 bits of the range) are then ranked among themselves by their full 64-bit numbers, so the output
// is the exact order.
//
// The network is the bitonic sorter: merges of span 2^lk, lk = 1 .. lp, each the steps of distance 2^(lk-1) .. 1,
// element i against i ^ distance, ascending where bit lk of i is clear.  A thread holds eight elements per trip
// through the array: the ones that differ in three consecutive index bits (bits 0-2, 3-5, 6-8, 9-11), so the three
// steps of those distances happen in registers and a merge takes one trip per bit triple it spans (16 trips for 512
// elements).  Elements of descending blocks are kept complemented, which makes every comparator "smaller one to the
// lower index"; the complement moves from bit lk to bit lk + 1 at the end of merge lk.
// The array is padded by 8 words per 64 and the low three address bits are xor-ed with bits 3-5 of the index: with
// that, the 64 lanes of each of the eight accesses of a trip hit 64 different banks for every bit triple (a thread
// taking four elements with plain addressing ran 4-way bank conflicts in most steps: 19 us for 5440 tiles).
// ---------------------------------------------------------------------------------------------------------------
#define TS_WORDS(cap) ((cap) + (cap) / 8)
#define TS_TRIPLES 5
__device__ __forceinline__ int ts_pa(int e) { return (e + ((e >> 6) << 3)) ^ ((e >> 3) & 7); }

// One trip: thread T of P >> MB takes the 2^MB elements that differ in index bits [s, s + MB) and runs the steps of
// merge lk whose distance bit lies there; `last` (s == 0): the merge ends here, the complement moves on.
template <int MB>
__device__ __forceinline__ void ts_trip(uint32_t* __restrict__ w, int T, int s, int lk, bool last) {
    const int ibase = ((T >> s) << (s + MB)) | (T & ((1 << s) - 1));
    uint32_t x[1 << MB];
#pragma unroll
    for (int m = 0; m < (1 << MB); ++m) x[m] = w[ts_pa(ibase | (m << s))];
#pragma unroll
    for (int rb = MB - 1; rb >= 0; --rb) {
        if (s + rb <= lk - 1) {
#pragma unroll
            for (int m = 0; m < (1 << MB); ++m) {
                if (!(m & (1 << rb))) {
                    const uint32_t a = x[m], b = x[m | (1 << rb)];
                    x[m] = min(a, b);
                    x[m | (1 << rb)] = max(a, b);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < (1 << MB); ++m) {
        const int i = ibase | (m << s);
        if (last) x[m] ^= 0u - (uint32_t)(((i >> lk) ^ (i >> (lk + 1))) & 1);
        w[ts_pa(i)] = x[m];
    }
}

template <int THREADS>
__device__ __forceinline__ void ts_trip_sync() {
    if (THREADS == 64) {            // one wave: LDS executes its accesses in order, only the compiler must keep it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// The compare-exchanges of a thread's eight registers for the index bits [lo_rb, hi_rb] of its triple, highest first.
__device__ __forceinline__ void ts_ce8(uint32_t (&x)[8], int hi_rb, int lo_rb) {
#pragma unroll
    for (int rb = 2; rb >= 0; --rb) {
        if (rb <= hi_rb && rb >= lo_rb) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (!(m & (1 << rb))) {
                    const uint32_t a = x[m], b = x[m | (1 << rb)];
                    x[m] = min(a, b);
                    x[m | (1 << rb)] = max(a, b);
                }
            }
        }
    }
}

// Word offset of the elements of "virtual thread" tid + THREADS * v from those of thread tid, for the bit triple that
// starts at bit sk: a thread-index bit at or above sk is element-index bit + 3, one below sk is the same element bit;
// all of them are element bits >= 6, where ts_pa is linear (2^p + 2^(p-3) words per bit).
template <int THREADS>
__device__ __forceinline__ int ts_voff(int sk, int v) {
    int tb = 0;
    while ((1 << tb) < THREADS) ++tb;               // constant-folded
    int off = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if ((v >> j) & 1) {
            const int p = tb + j, pe = p >= sk ? p + 3 : p;
            off += (1 << pe) + (1 << (pe - 3));
        }
    }
    return off;
}

// Sorts w[ts_pa(0 .. P-1)] ascending, P = 2^lp >= 8.  The caller synchronises before (the fill) and after.
// With P / 8 <= TS_VMAX * THREADS every thread keeps the same element octets per bit triple in all merges and their
// LDS addresses are computed once (computed per trip they were three quarters of the kernel's vector instructions);
// a thread's second octet lies a fixed number of words behind its first.
// The triples are bits 0-2, 3-5, ... and, at the top, the last three bits lp-3 .. lp-1, which may overlap the triple
// below: that one then leaves the shared bits to the top triple.
#define TS_VMAX 2                                   // element octets per thread and triple on the fast path
template <int THREADS>
__device__ __forceinline__ void ts_sort_words(uint32_t* __restrict__ w, int lp) {
    const int P = 1 << lp;
    if ((P >> 3) <= THREADS * TS_VMAX) {
        const int T = threadIdx.x;
        const int V = (P >> 3) > THREADS ? (P >> 3) / THREADS : 1;       // octets per thread: a power of two
        const bool act = T < (P >> 3);
        const int np = (lp + 2) / 3;                    // bit triples: 1 .. TS_TRIPLES (P <= 2^15)
        uint32_t* ap[TS_TRIPLES][8];
        int sft[TS_TRIPLES];
#pragma unroll
        for (int k = 0; k < TS_TRIPLES; ++k) {
            const int sk = (k >= np - 1) ? lp - 3 : 3 * k;
            sft[k] = sk;
            if (k < np) {                               // uniform: the unused triples cost nothing
                const int ibase = ((T >> sk) << (sk + 3)) | (T & ((1 << sk) - 1));
#pragma unroll
                for (int m = 0; m < 8; ++m) ap[k][m] = w + ts_pa((ibase | (m << sk)) & (P - 1));
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) ap[k][m] = w;
            }
        }
        uint32_t x[8];
        if (act) {                                      // merges 1-3: every aligned run of eight, inside a thread
            for (int v = 0; v < V; ++v) {
                const int d = ts_voff<THREADS>(0, v), Tv = T + THREADS * v;
#pragma unroll
                for (int m = 0; m < 8; ++m) x[m] = ap[0][m][d] ^ (0u - (uint32_t)((m >> 1) & 1));   // bit 1 set: descending
#pragma unroll
                for (int lk = 1; lk <= 3; ++lk) {
                    ts_ce8(x, lk - 1, 0);
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int i = (Tv << 3) | m;
                        x[m] ^= 0u - (uint32_t)(((i >> lk) ^ (i >> (lk + 1))) & 1);
                    }
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) ap[0][m][d] = x[m];
            }
        }
        for (int lk = 4; lk <= lp; ++lk) {
#pragma unroll
            for (int k = TS_TRIPLES - 1; k >= 0; --k) {
                if (k >= np) continue;
                const int sk = sft[k];
                const int hi = (k == np - 1) ? lp - 1 : min(sk + 2, sft[k + 1 < TS_TRIPLES ? k + 1 : TS_TRIPLES - 1] - 1);
                const int top = min(lk - 1, hi);
                if (top < sk) continue;                 // the merge does not reach this triple
                ts_trip_sync<THREADS>();
                if (act) {
                    for (int v = 0; v < V; ++v) {
                        const int d = ts_voff<THREADS>(sk, v), Tv = T + THREADS * v;
#pragma unroll
                        for (int m = 0; m < 8; ++m) x[m] = ap[k][m][d];
                        ts_ce8(x, top - sk, 0);
                        if (k == 0) {                   // the merge ends here: the complement moves from bit lk to lk + 1
                            const uint32_t flip = 0u - (uint32_t)(((Tv >> (lk - 3)) ^ (Tv >> (lk - 2))) & 1);
#pragma unroll
                            for (int m = 0; m < 8; ++m) x[m] ^= flip;
                        }
#pragma unroll
                        for (int m = 0; m < 8; ++m) ap[k][m][d] = x[m];
                    }
                }
            }
        }
        return;
    }
    // more than eight elements per thread and triple: addresses per trip
    for (int T = threadIdx.x; T < (P >> 3); T += THREADS) {
        const int ibase = T << 3;
        uint32_t x[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = w[ts_pa(ibase | m)] ^ (0u - (uint32_t)((m >> 1) & 1));
#pragma unroll
        for (int lk = 1; lk <= 3; ++lk) {
            ts_ce8(x, lk - 1, 0);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int i = ibase | m;
                x[m] ^= 0u - (uint32_t)(((i >> lk) ^ (i >> (lk + 1))) & 1);
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) w[ts_pa(ibase | m)] = x[m];
    }
    for (int lk = 4; lk <= lp; ++lk) {
        for (int ph = (lk - 1) / 3; ph >= 0; --ph) {
            const int s = 3 * ph, mb = min(3, lp - s);
            ts_trip_sync<THREADS>();
            if (mb == 3) {
                for (int T = threadIdx.x; T < (P >> 3); T += THREADS) ts_trip<3>(w, T, s, lk, ph == 0);
            } else if (mb == 2) {
                for (int T = threadIdx.x; T < (P >> 2); T += THREADS) ts_trip<2>(w, T, s, lk, false);
            } else {
                for (int T = threadIdx.x; T < (P >> 1); T += THREADS) ts_trip<1>(w, T, s, lk, false);
            }
        }
    }
}

#define TS_TIE_RUN_MAX 48                           // longest run of equal depth fields ranked pairwise (see tile_sort_kernel)

// The same network on n distinct 64-bit numbers s[0, n) where they lie (in memory: segments longer than any LDS
// segment): the positions n .. P-1 of the padded problem would hold +infinity and are never written, a comparator
// that reaches them is skipped.
template <int THREADS>
__device__ __forceinline__ void ts_bitonic64(uint64_t* s, int n, int P) {
    for (int lk = 1; (1 << lk) <= P; ++lk) {
        const int k = 1 << lk, half = k >> 1;
        for (int t = threadIdx.x; t < (P >> 1); t += THREADS) {
            const int blk = t >> (lk - 1), r = t & (half - 1);
            const int i = (blk << lk) + r, l = (blk << lk) + (k - 1 - r);
            if (l < n) {
                const uint64_t a = s[i], b = s[l];
                if (a > b) { s[i] = b; s[l] = a; }
            }
        }
        __syncthreads();
        for (int j = half >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
                if (l < n) {
                    const uint64_t a = s[i], b = s[l];
                    if (a > b) { s[i] = b; s[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Sorts the segments of the tiles whose length is in (lo, CAP] in LDS (and, with GLOBAL_TAIL, the longer ones in
// place in memory).  Workgroup w takes the entries w, w + gridDim.x, ... of desc = (tile, first position, length),
// which is ordered by decreasing length class, so with STOP_EARLY a workgroup leaves at the first tile of at most
// lo / 2 entries (every later tile is shorter than lo; classes are 1/16 octave wide).  LDS: 4 CAP bytes (the words);
// the entries themselves wait in registers for the segment's depth range and are read again (from L2) for the output.
template <int THREADS, int CAP, bool STOP_EARLY, bool GLOBAL_TAIL>
__global__ void __launch_bounds__(THREADS)
tile_sort_kernel(uint64_t* __restrict__ binned, const uint4* __restrict__ desc, int tiles, int lo,
                 uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    constexpr int ITEMS = CAP / THREADS;
    __shared__ uint32_t word[TS_WORDS(CAP)];
    __shared__ uint32_t s_lo, s_hi;
    __shared__ int s_run;                           // 1 when a run of equal depth fields is longer than TS_TIE_RUN_MAX
    for (int w = blockIdx.x; w < tiles; w += gridDim.x) {
        const uint4 d4 = desc[w];
        const uint32_t tile = d4.x, first = d4.y;
        const int n = (int)d4.z;
        if (tile >= (uint32_t)tiles) return;
        if (STOP_EARLY && n <= lo / 2) return;
        if (n <= lo || (!GLOBAL_TAIL && n > CAP)) continue;
        int lp = 3;
        while ((1 << lp) < n) ++lp;
        const int P = 1 << lp;
        const uint64_t hi = (uint64_t)tile << 32;
        if (n > CAP) {
            // longer than the LDS segment: the 64-bit network on the segment where it lies (the workgroup's own
            // stores are visible to it after the barrier)
            uint64_t* seg = binned + first;
            __syncthreads();
            ts_bitonic64<THREADS>(seg, n, P);
            for (int i = threadIdx.x; i < n; i += THREADS) {
                const uint64_t e = seg[i];
                keys_out[first + i] = hi | (e >> 32);
                vals_out[first + i] = (uint32_t)e;
            }
            continue;
        }
        if (threadIdx.x == 0) { s_lo = 0xffffffffu; s_hi = 0u; s_run = 0; }
        __syncthreads();
        uint32_t dv[ITEMS];
        uint32_t dlo = 0xffffffffu, dhi = 0u;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int i = threadIdx.x + k * THREADS;
            dv[k] = 0u;
            if (k * THREADS < n && i < n) {             // the first test is uniform: short segments skip the rest
                dv[k] = (uint32_t)(binned[first + i] >> 32);
                dlo = min(dlo, dv[k]);
                dhi = max(dhi, dv[k]);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            dlo = min(dlo, (uint32_t)__shfl_xor((int)dlo, d));
            dhi = max(dhi, (uint32_t)__shfl_xor((int)dhi, d));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_lo, dlo); atomicMax(&s_hi, dhi); }
        __syncthreads();
        const uint32_t dmin = s_lo, span = s_hi - dmin;
        const int sb = span == 0 ? 0 : (32 - __builtin_clz(span));             // bits of the depth range
        const int shift = max(0, sb - (32 - lp));
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int i = threadIdx.x + k * THREADS;
            if (k * THREADS < P && i < P)
                word[ts_pa(i)] = i < n ? ((((dv[k] - dmin) >> shift) << lp) | (uint32_t)i) : 0xffffffffu;
        }
        __syncthreads();
        ts_sort_words<THREADS>(word, lp);
        __syncthreads();
        const uint32_t imask = (uint32_t)P - 1u;
        // The exact order among entries of one depth field is restored below by ranking every tied entry inside its
        // run: quadratic in the run length.  A segment with a run of more than TS_TIE_RUN_MAX entries (coincident or
        // duplicated Gaussians, a depth span far narrower than the field) is sorted by the 64-bit network on the
        // full numbers instead, where it lies in memory; the first entry of each run looks that far ahead, no further.
        for (int p = threadIdx.x; p < n; p += THREADS) {
            const uint32_t field = word[ts_pa(p)] >> lp;
            if (p == 0 || (word[ts_pa(p - 1)] >> lp) != field) {
                int b = p + 1;
                while (b < n && b - p <= TS_TIE_RUN_MAX && (word[ts_pa(b)] >> lp) == field) ++b;
                if (b - p > TS_TIE_RUN_MAX) s_run = 1;
            }
        }
        __syncthreads();
        if (s_run) {
            uint64_t* seg = binned + first;
            ts_bitonic64<THREADS>(seg, n, P);
            for (int i = threadIdx.x; i < n; i += THREADS) {
                const uint64_t e = seg[i];
                keys_out[first + i] = hi | (e >> 32);
                vals_out[first + i] = (uint32_t)e;
            }
            __syncthreads();
            continue;
        }
        for (int p = threadIdx.x; p < n; p += THREADS) {
            const uint32_t wd = word[ts_pa(p)], field = wd >> lp;
            const uint64_t e = binned[first + (wd & imask)];
            int pos = p;
            const bool tie_prev = p > 0 && (word[ts_pa(p - 1)] >> lp) == field;
            const bool tie_next = p + 1 < n && (word[ts_pa(p + 1)] >> lp) == field;
            if (tie_prev || tie_next) {
                // entries of one depth field: rank among them by the full numbers
                int a = p, b = p + 1;
                while (a > 0 && (word[ts_pa(a - 1)] >> lp) == field) --a;
                while (b < n && (word[ts_pa(b)] >> lp) == field) ++b;
                int rank = 0;
                for (int q = a; q < b; ++q) rank += binned[first + (word[ts_pa(q)] & imask)] < e ? 1 : 0;
                pos = a + rank;
            }
            keys_out[first + pos] = hi | (e >> 32);
            vals_out[first + pos] = (uint32_t)e;
        }
        __syncthreads();                            // the next tile of this workgroup reuses the LDS array
    }
}

#define TS_CAP_TAIL 8192                            // 36 KB of LDS
static_assert(TS_CAP_TAIL <= (1 << (3 * TS_TRIPLES)), "bit triples of the LDS sort");

// Returns FDGS_OK or an error, or 1 when this path does not apply (more tiles than the LDS histogram holds,
// workspace too small): the caller then runs the radix sort.
int fdgs_tile_sort(const fdgs_binsort_args* a, uint32_t* ws, size_t ws_words, hipStream_t s, int phase) {
    const int gx = fdgs_tiles_x(a->image_width), gy = fdgs_tiles_y(a->image_height);
    const int tiles = gx * gy;
    ts_plan plan;
    if (!ts_make_plan(a->num_points, tiles, &plan)) return 1;
    if (ws_words < (size_t)FDGS_TILE_SORT_WS_WORDS) return 1;
    uint32_t* totals = ws;
    uint32_t* tile_start = ws + TS_MAX_TILES;
    uint32_t* order = a->tile_order ? a->tile_order : ws + 2 * TS_MAX_TILES;
    uint4* desc = reinterpret_cast<uint4*>(ws + 3 * TS_MAX_TILES);   // 16-byte aligned: the workspace is, and so are the areas before
    uint32_t* table = ws + 7 * TS_MAX_TILES;
    uint64_t* binned = a->keys_unsorted;

    const size_t walk_lds = ((size_t)tiles + 5 * TS_WALK_WAVES * 64 + TS_WALK_WAVES) * sizeof(uint32_t);   // >= tiles + 512 + 16 words for the ordering workgroup
    if (walk_lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_walk_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_walk_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds) != hipSuccess)
            return FDGS_ELAUNCH;
    }
    if (phase != FDGS_BINSORT_FINISH) {             // the counting part: needs neither the list length nor the list buffers
        hipLaunchKernelGGL(tile_walk_kernel<false>, dim3(plan.groups), dim3(TS_WALK_THREADS), walk_lds, s,
                           a->num_points, plan.per_group, a->geom, a->point_offsets, gx, gy, table, tile_start,
                           binned, totals, order, a->ranges, desc);
        FDGS_CHECK_LAUNCH();
        hipLaunchKernelGGL(tile_table_scan_kernel, dim3((tiles + 31) / 32), dim3(TS_SCAN_THREADS), 0, s, table,
                           totals, tiles, plan.groups);
        FDGS_CHECK_LAUNCH();
        if (phase == FDGS_BINSORT_COUNT) return FDGS_OK;
    }
    hipLaunchKernelGGL(tile_walk_kernel<true>, dim3(plan.groups + 1), dim3(TS_WALK_THREADS), walk_lds, s,
                       a->num_points, plan.per_group, a->geom, a->point_offsets, gx, gy, table, tile_start, binned,
                       totals, order, a->ranges, desc);
    FDGS_CHECK_LAUNCH();

    // One workgroup per tile sorts segments of up to `cap` entries, cap chosen from the mean list length; a few large
    // workgroups then walk the tiles longest first for the longer segments and leave when the lengths fall below cap.
    const int64_t mean = a->num_rendered / tiles;
    int cap;
    if (mean <= 512) {
        cap = 1024;
        hipLaunchKernelGGL((tile_sort_kernel<64, 1024, false, false>), dim3(tiles), dim3(64), 0, s, binned, desc,
                           tiles, 0, a->keys_sorted, a->vals_sorted);
    } else if (mean <= 1024) {
        cap = 2048;
        hipLaunchKernelGGL((tile_sort_kernel<128, 2048, false, false>), dim3(tiles), dim3(128), 0, s, binned, desc,
                           tiles, 0, a->keys_sorted, a->vals_sorted);
    } else {
        cap = 4096;
        hipLaunchKernelGGL((tile_sort_kernel<256, 4096, false, false>), dim3(tiles), dim3(256), 0, s, binned, desc,
                           tiles, 0, a->keys_sorted, a->vals_sorted);
    }
    FDGS_CHECK_LAUNCH();
    hipLaunchKernelGGL((tile_sort_kernel<1024, TS_CAP_TAIL, true, true>), dim3(tiles < 256 ? tiles : 256), dim3(1024),
                       0, s, binned, desc, tiles, cap, a->keys_sorted, a->vals_sorted);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
