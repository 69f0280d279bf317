// Stage 4: stable LSD radix sort of (uint64 key, uint32 value) pairs for gfx950.
//
// Digits of up to RS_BITS bits, the key bits divided evenly over the passes: the keys of a 1352 x 1014 frame are 13
// tile bits + the bits of the frame's depth range (26-27 for depths between 0.2 and a few hundred, when the caller
// knows the range; 32 otherwise): five passes, six without the range.  (Ten-bit digits need one pass fewer, but the
// 1024-entry tables per 2048-key sub-tile make each pass a third slower: measured 38 us against 29 us a pass.)
// Per pass three launches:
//   1. digit histogram per block (LDS atomics), stored digit-major  hist[d][block]
//   2. one block per digit: exclusive scan of its row + the digit total
//   3. scatter: each block walks its range in sub-tiles of 2048 keys; ranks are
//      computed per wave with 64-bit ballots (peers with the same digit), wave
//      counters live in LDS, so the order of equal digits is the memory order
//      (stable).
// Only bits [0, end_bit) are sorted.  Buffers ping-pong between the caller's
// "unsorted" and "sorted" arrays; the caller arranges the start so that the last
// pass lands in the "sorted" arrays (see fdgs_radix_sort_first_buffer).
#include "fdgs_common.h"

#define RS_THREADS 256
#define RS_IPL 8                                    // items per lane
#define RS_SUBTILE (RS_THREADS * RS_IPL)            // 2048 keys
#define RS_MAX_BLOCKS 1024
#define RS_BITS 8
#define RS_BINS (1 << RS_BITS)
#define RS_DPT ((RS_BINS + RS_THREADS - 1) / RS_THREADS)   // digits a thread owns where threads own digits

struct rs_plan {
    int blocks;
    int64_t per_block;   // multiple of RS_SUBTILE
};

static rs_plan rs_make_plan(int64_t n) {
    rs_plan p;
    int64_t subtiles = (n + RS_SUBTILE - 1) / RS_SUBTILE;
    int64_t blocks = subtiles < RS_MAX_BLOCKS ? subtiles : RS_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    int64_t st_per_block = (subtiles + blocks - 1) / blocks;
    p.per_block = st_per_block * RS_SUBTILE;
    p.blocks = (int)((n + p.per_block - 1) / p.per_block);
    if (p.blocks < 1) p.blocks = 1;
    return p;
}

extern "C" size_t fdgs_raster_sort_workspace_bytes(int64_t num_rendered) {
    (void)num_rendered;
    static_assert(FDGS_RADIX_WS_WORDS == RS_BINS * RS_MAX_BLOCKS + RS_BINS, "radix area of the sort workspace");
    return ((size_t)FDGS_RADIX_WS_WORDS + FDGS_TILE_SORT_WS_WORDS) * sizeof(uint32_t);
}

__global__ void __launch_bounds__(RS_THREADS)
rs_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t per_block, int shift, uint32_t mask,
               uint32_t* __restrict__ hist, int nblocks) {
    __shared__ uint32_t h[RS_BINS];
    for (int d = threadIdx.x; d < RS_BINS; d += RS_THREADS) h[d] = 0;
    __syncthreads();
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    int64_t end = begin + per_block;
    if (end > n) end = n;
    for (int64_t i = begin + threadIdx.x; i < end; i += RS_THREADS) {
        const uint32_t d = (uint32_t)(keys[i] >> shift) & mask;
        atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RS_BINS; d += RS_THREADS) hist[(size_t)d * nblocks + blockIdx.x] = h[d];
}

// One block per digit: exclusive scan of hist[d][0..nblocks) in place, total[d].
__global__ void __launch_bounds__(RS_THREADS)
rs_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals, int nblocks) {
    __shared__ uint32_t wsum[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v[4];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = threadIdx.x * 4 + k;
        v[k] = idx < nblocks ? row[idx] : 0u;
        s += v[k];
    }
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) base += wsum[w];
    uint32_t run = base + incl - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = threadIdx.x * 4 + k;
        if (idx < nblocks) row[idx] = run;
        run += v[k];
    }
    if (threadIdx.x == RS_THREADS - 1) totals[blockIdx.x] = run;
}

__global__ void __launch_bounds__(RS_THREADS)
rs_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                  uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n, int64_t per_block,
                  int shift, uint32_t mask, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals,
                  int nblocks, int digit_bits, int expand_bits, uint32_t expand_add) {
    __shared__ uint32_t running[RS_BINS];          // next global slot per digit for this block
    __shared__ uint32_t wave_cnt[4][RS_BINS];      // per-wave digit counts, then per-wave bases
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // digit base = exclusive scan of the RS_BINS totals (thread t owns digits RS_DPT*t ...), + this block's prefix
    // inside the digit row
    {
        const int d0 = RS_DPT * threadIdx.x;
        uint32_t tt[RS_DPT];
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < RS_DPT; ++k) {
            tt[k] = d0 + k < RS_BINS ? totals[d0 + k] : 0u;
            t += tt[k];
        }
        uint32_t incl = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w < wave) base += wsum[w];
        uint32_t excl = base + incl - t;
#pragma unroll
        for (int k = 0; k < RS_DPT; ++k) {
            if (d0 + k < RS_BINS) running[d0 + k] = excl + hist[(size_t)(d0 + k) * nblocks + blockIdx.x];
            excl += tt[k];
        }
    }

    const int64_t begin = (int64_t)blockIdx.x * per_block;
    int64_t end = begin + per_block;
    if (end > n) end = n;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (int64_t tile = begin; tile < end; tile += RS_SUBTILE) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
            for (int d = threadIdx.x; d < RS_BINS; d += RS_THREADS) wave_cnt[w][d] = 0;
        __syncthreads();

        uint64_t key[RS_IPL];
        uint32_t val[RS_IPL];
        uint32_t rank[RS_IPL];
        const int64_t wbase = tile + (int64_t)wave * (64 * RS_IPL);
#pragma unroll
        for (int k = 0; k < RS_IPL; ++k) {
            const int64_t idx = wbase + k * 64 + lane;
            const bool ok = idx < end;
            key[k] = ok ? keys_in[idx] : ~0ull;
            val[k] = ok ? vals_in[idx] : 0u;
        }
#pragma unroll
        for (int k = 0; k < RS_IPL; ++k) {
            const int64_t idx = wbase + k * 64 + lane;
            const bool ok = idx < end;
            const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
            // lanes of this wave holding the same digit (invalid lanes match nobody valid)
            unsigned long long peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < RS_BITS; ++b) {
                if (b < digit_bits) {
                    const unsigned long long bal = __ballot((d >> b) & 1u);
                    peers &= ((d >> b) & 1u) ? bal : ~bal;
                }
            }
            const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
            uint32_t prev = 0;
            if (ok) prev = wave_cnt[wave][d];
            rank[k] = prev + before;
            if (ok && before == 0) wave_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
        }
        __syncthreads();
        // per digit: turn the four wave counts into wave bases, advance the running slot
        for (int d = threadIdx.x; d < RS_BINS; d += RS_THREADS) {
            uint32_t c0 = wave_cnt[0][d], c1 = wave_cnt[1][d], c2 = wave_cnt[2][d], c3 = wave_cnt[3][d];
            const uint32_t g = running[d];
            wave_cnt[0][d] = g;
            wave_cnt[1][d] = g + c0;
            wave_cnt[2][d] = g + c0 + c1;
            wave_cnt[3][d] = g + c0 + c1 + c2;
            running[d] = g + c0 + c1 + c2 + c3;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RS_IPL; ++k) {
            const int64_t idx = wbase + k * 64 + lane;
            if (idx < end) {
                const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
                const uint32_t dst = wave_cnt[wave][d] + rank[k];
                // last pass of a compressed sort: back to (tile << 32) | depth bits
                const uint64_t ko = expand_bits ? (((key[k] >> expand_bits) << 32) |
                                                   (uint64_t)((uint32_t)(key[k] & ((1ull << expand_bits) - 1ull)) + expand_add))
                                                : key[k];
                keys_out[dst] = ko;
                vals_out[dst] = val[k];
            }
        }
        __syncthreads();
    }
}

// Number of passes (digits of at most RS_BITS bits) needed for bits [0,end_bit).
int fdgs_radix_sort_passes(int end_bit) { return end_bit <= 0 ? 1 : (end_bit + RS_BITS - 1) / RS_BITS; }

// Sorts n pairs.  Pass p reads buffer (p even ? A : B) and writes the other one,
// where A = (keys_a, vals_a).  The result is in B after an odd number of passes
// and in A after an even number; the caller picks A/B accordingly.
// expand_bits != 0: the keys are (tile << expand_bits) | low; the last pass writes (tile << 32) | (low + expand_add).
int fdgs_radix_sort_pingpong(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, int64_t n,
                             int end_bit, void* workspace, size_t workspace_bytes, hipStream_t s, int expand_bits,
                             uint32_t expand_add) {
    if (n <= 0) return FDGS_OK;
    if (workspace_bytes < fdgs_raster_sort_workspace_bytes(n)) return FDGS_EWORKSPACE;
    uint32_t* hist = (uint32_t*)workspace;
    uint32_t* totals = hist + (size_t)RS_BINS * RS_MAX_BLOCKS;
    const rs_plan plan = rs_make_plan(n);
    const int passes = fdgs_radix_sort_passes(end_bit);
    if (end_bit < 1) end_bit = 1;
    int shift = 0;
    for (int p = 0; p < passes; ++p, shift += 0) {
        const int bits = (end_bit - shift + (passes - p) - 1) / (passes - p);      // the remaining bits, evenly
        const uint32_t mask = (1u << bits) - 1u;
        const uint64_t* kin = (p & 1) ? keys_b : keys_a;
        const uint32_t* vin = (p & 1) ? vals_b : vals_a;
        uint64_t* kout = (p & 1) ? keys_a : keys_b;
        uint32_t* vout = (p & 1) ? vals_a : vals_b;
        hipLaunchKernelGGL(rs_hist_kernel, dim3(plan.blocks), dim3(RS_THREADS), 0, s, kin, n, plan.per_block, shift,
                           mask, hist, plan.blocks);
        FDGS_CHECK_LAUNCH();
        hipLaunchKernelGGL(rs_rowscan_kernel, dim3(RS_BINS), dim3(RS_THREADS), 0, s, hist, totals, plan.blocks);
        FDGS_CHECK_LAUNCH();
        const bool last = p == passes - 1;
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(plan.blocks), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n,
                           plan.per_block, shift, mask, hist, totals, plan.blocks, bits, last ? expand_bits : 0,
                           expand_add);
        FDGS_CHECK_LAUNCH();
        shift += bits;
    }
    return FDGS_OK;
}
