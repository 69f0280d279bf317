// Rasterizer stage 6 (front-to-back alpha blending) and stage 7 (its backward).
//
// gfx950 mapping: one 256-thread workgroup per 16x16 tile; each of its four wavefronts owns one 8x8 pixel quadrant
// and walks the tile's depth-sorted list (the forward on its own, with no workgroup barrier; the backward in step
// with the tile's other waves, one barrier per batch, see blend_bwd_kernel).  Inside the wave, each of the four
// 16-lane DPP rows owns one 4x4 pixel block and walks ITS OWN list of survivors: a batch of 64 list entries is loaded
// one per lane, every lane tests its entry's 1/255-alpha box against the four blocks, four 64-bit ballots collect the
// survivors per block, and in every step each row takes the next survivor of its own block.  Splats here cover a
// third of an 8x8 quadrant on average: with one survivor per wave and step two thirds of the lanes computed nothing.
// A survivor's record reaches its row through a wave-private LDS copy of the batch (same-address reads within a
// row; an LDS read is not a vector instruction, and both kernels are bound by vector-instruction issue).  A row
// whose list is exhausted reads a null record (opacity 0: never contributes).  The wave-private LDS areas need no
// barrier: a wave's LDS instructions execute in program order.
//
// The conic is staged pre-scaled, a2 = -log2(e) A, b1 = -log2(e) B, c2 = -log2(e) C, so that
// exp(power) = exp2(dx (a2 dx / 2 + b1 dy) + c2 dy^2 / 2) needs no further constant, and the backward accumulates
// its five geometric sums in that scaling (fdgs_common.h: FDGS_DGEOM_*); preprocess_bwd applies the constants.
#include "fdgs_common.h"

#define BL_SLOTS 65                                // 64 entries of a batch + the null record
#define BL_NULL 64

// 64-bit lane mask of a predicate (__ballot takes an int: the predicate would be turned into 0 / 1 in a vector
// register and compared again)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Turns the four survivor masks of a batch into four compacted lists of batch positions in the wave's LDS area
// (slot k of list g = the k-th set bit of mask g, at byte 4 k + g): the inner loop then finds a row's next survivor
// with one LDS byte read instead of scalar bit scans on four 64-bit masks (the scalar unit is shared by the CU's four
// SIMDs and was the bottleneck with them).  Every slot is first set to the null record, so a row whose list is
// shorter than the longest one reads null records for the rest of the batch's steps without a test.
// Returns the longest list's length.
__device__ __forceinline__ int build_lists(unsigned char* __restrict__ lists, unsigned long long m0,
                                           unsigned long long m1, unsigned long long m2, unsigned long long m3,
                                           int lane) {
    const uint32_t lo0 = (uint32_t)m0, hi0 = (uint32_t)(m0 >> 32), lo1 = (uint32_t)m1, hi1 = (uint32_t)(m1 >> 32);
    const uint32_t lo2 = (uint32_t)m2, hi2 = (uint32_t)(m2 >> 32), lo3 = (uint32_t)m3, hi3 = (uint32_t)(m3 >> 32);
    const uint32_t r0 = __builtin_amdgcn_mbcnt_hi(hi0, __builtin_amdgcn_mbcnt_lo(lo0, 0u));   // set bits below this lane
    const uint32_t r1 = __builtin_amdgcn_mbcnt_hi(hi1, __builtin_amdgcn_mbcnt_lo(lo1, 0u));
    const uint32_t r2 = __builtin_amdgcn_mbcnt_hi(hi2, __builtin_amdgcn_mbcnt_lo(lo2, 0u));
    const uint32_t r3 = __builtin_amdgcn_mbcnt_hi(hi3, __builtin_amdgcn_mbcnt_lo(lo3, 0u));
    reinterpret_cast<uint32_t*>(lists)[lane] = 0x01010101u * (uint32_t)BL_NULL;     // a wave's LDS accesses run in order
    if ((m0 >> lane) & 1ull) lists[4 * r0 + 0] = (unsigned char)lane;
    if ((m1 >> lane) & 1ull) lists[4 * r1 + 1] = (unsigned char)lane;
    if ((m2 >> lane) & 1ull) lists[4 * r2 + 2] = (unsigned char)lane;
    if ((m3 >> lane) & 1ull) lists[4 * r3 + 3] = (unsigned char)lane;
    const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
    return __builtin_amdgcn_readfirstlane(max(max(n0, n1), max(n2, n3)));     // wave-uniform: a scalar loop count
}

// Can the splat's alpha reach 1/255 at some pixel centre of the rectangle [ex0, ex1] x [ey0, ey1] (coordinates
// relative to the splat's centre)?  The exponent E(dx, dy) = (a2 dx^2 + c2 dy^2) / 2 + b1 dx dy (in log2 units,
// a2, c2 < 0) is concave with its maximum 0 at the centre, so over a rectangle that does not contain the centre it
// peaks on an edge facing the centre; on the edge dx = e the best dy is -b1 e / c2 clipped to the edge.
// tau = the exponent at which alpha = 1/255, lowered by a margin that covers the rounding of this test and of
// the per-pixel evaluation.  Anything undecidable (degenerate conic, NaN) counts as reachable.
__device__ __forceinline__ bool exponent_reaches(float ex0, float ex1, float ey0, float ey1, float a2, float b1,
                                                 float c2, float inv_a2, float inv_c2, float tau) {
    const bool in_x = ex0 <= 0.0f && ex1 >= 0.0f, in_y = ey0 <= 0.0f && ey1 >= 0.0f;
    const float e_x = ex0 > 0.0f ? ex0 : ex1, e_y = ey0 > 0.0f ? ey0 : ey1;
    const float ty = fminf(fmaxf(-b1 * e_x * inv_c2, ey0), ey1);
    const float tx = fminf(fmaxf(-b1 * e_y * inv_a2, ex0), ex1);
    const float on_v = 0.5f * a2 * e_x * e_x + ty * (b1 * e_x + 0.5f * c2 * ty);     // best on the edge dx = e_x
    const float on_h = 0.5f * c2 * e_y * e_y + tx * (b1 * e_y + 0.5f * a2 * tx);     // best on the edge dy = e_y
    const float best = in_x ? on_h : (in_y ? on_v : fmaxf(on_v, on_h));
    return (in_x && in_y) || !(best < tau);
}

// Which of the wave's four 4x4 blocks the entry can contribute to; (bx, by) is the quadrant's origin.  First the
// precomputed box of the 1/255-alpha ellipse (fdgs_pack_reach), then the ellipse itself against each block.
__device__ __forceinline__ void reach4(float gx, float gy, uint32_t reach, float a2, float b1, float c2, float o,
                                       float bx, float by, bool valid,
                                       unsigned long long& m0, unsigned long long& m1, unsigned long long& m2,
                                       unsigned long long& m3, uint32_t* bits = nullptr) {
    const float hx = __half2float(__ushort_as_half((unsigned short)(reach & 0xffffu)));
    const float hy = __half2float(__ushort_as_half((unsigned short)(reach >> 16)));
    const float x_hi = gx + hx, x_lo = gx - hx, y_hi = gy + hy, y_lo = gy - hy;
    const bool inx0 = x_hi >= bx && x_lo <= bx + 3.0f, inx1 = x_hi >= bx + 4.0f && x_lo <= bx + 7.0f;
    const bool iny0 = valid && y_hi >= by && y_lo <= by + 3.0f, iny1 = valid && y_hi >= by + 4.0f && y_lo <= by + 7.0f;
    const bool conic_ok = a2 < 0.0f && c2 < 0.0f;
    const float inv_a2 = conic_ok ? __builtin_amdgcn_rcpf(a2) : 0.0f, inv_c2 = conic_ok ? __builtin_amdgcn_rcpf(c2) : 0.0f;
    const float t0 = -__builtin_amdgcn_logf(255.0f * o);            // v_log_f32 is log2
    const float tau = conic_ok ? t0 - (0.001f * fabsf(t0) + 0.001f) : -__builtin_inff();
    const float ex = bx - gx, ey = by - gy;
    const bool r0 = exponent_reaches(ex, ex + 3.0f, ey, ey + 3.0f, a2, b1, c2, inv_a2, inv_c2, tau);
    const bool r1 = exponent_reaches(ex + 4.0f, ex + 7.0f, ey, ey + 3.0f, a2, b1, c2, inv_a2, inv_c2, tau);
    const bool r2 = exponent_reaches(ex, ex + 3.0f, ey + 4.0f, ey + 7.0f, a2, b1, c2, inv_a2, inv_c2, tau);
    const bool r3 = exponent_reaches(ex + 4.0f, ex + 7.0f, ey + 4.0f, ey + 7.0f, a2, b1, c2, inv_a2, inv_c2, tau);
    const bool h0 = inx0 && iny0 && r0, h1 = inx1 && iny0 && r1, h2 = inx0 && iny1 && r2, h3 = inx1 && iny1 && r3;
    m0 = ballot64(h0); m1 = ballot64(h1);
    m2 = ballot64(h2); m3 = ballot64(h3);
    if (bits) *bits = (h0 ? 1u : 0u) | (h1 ? 2u : 0u) | (h2 ? 4u : 0u) | (h3 ? 8u : 0u);
}

#define BL_LOG2E 1.4426950408889634f

__global__ void __launch_bounds__(256)
blend_fwd_kernel(fdgs_blend_fwd_args a, int tiles_x) {
    const int tile = a.tile_order ? (int)a.tile_order[blockIdx.x] : (int)blockIdx.x;
    if ((unsigned)tile >= gridDim.x) return;        // not a tile id: the caller's tile_order is not a permutation
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane >> 4, li = lane & 15;
    const int qx0 = tx * FDGS_TILE + (wave & 1) * 8, qy0 = ty * FDGS_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + 4 * (row & 1) + (li & 3), pyi = qy0 + 4 * (row >> 1) + (li >> 2);
    const int W = a.image_width, H = a.image_height;
    const bool inside = pxi < W && pyi < H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float bx = (float)qx0, by = (float)qy0;

    const uint32_t r_begin = a.ranges[2 * tile], r_end = a.ranges[2 * tile + 1];
    const float4* __restrict__ geom4 = reinterpret_cast<const float4*>(a.geom);

    __shared__ float4 staged[4][BL_SLOTS * 3];              // (x y a2 b1) (c2 o r g) (b depth - -): one address per entry
    float4* __restrict__ my = staged[wave];
    __shared__ unsigned char lists[4][4 * 64];              // per wave: the four rows' survivor lists of a batch
    unsigned char* __restrict__ mylist = lists[wave];
    if (lane == 0) {
        my[BL_NULL * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        my[BL_NULL * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);      // opacity 0
        my[BL_NULL * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, D = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t base = r_begin; base < r_end; base += 64) {
        const unsigned long long live = ballot64(!done);
        if (live == 0ull) break;
        const uint32_t e = base + lane;
        const bool valid = e < r_end;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        if (valid) {
            const uint32_t id = a.point_list[e];
            g0 = geom4[3 * (size_t)id + 0];
            g1 = geom4[3 * (size_t)id + 1];
            g2 = geom4[3 * (size_t)id + 2];
        }
        const float sa2 = -BL_LOG2E * g0.z, sb1 = -BL_LOG2E * g0.w, sc2 = -BL_LOG2E * g1.x;
        my[lane * 3 + 0] = make_float4(g0.x, g0.y, sa2, sb1);
        my[lane * 3 + 1] = make_float4(sc2, g1.y, g1.z, g1.w);
        *reinterpret_cast<float2*>(&my[lane * 3 + 2]) = make_float2(g2.x, g2.y);
        unsigned long long m0, m1, m2, m3;
        if (a.reach_masks) {            // kept for the backward pass (before the rows' saturation is applied)
            uint32_t bits;
            reach4(g0.x, g0.y, __float_as_uint(g2.w), sa2, sb1, sc2, g1.y, bx, by, valid, m0, m1, m2, m3, &bits);
            if (valid) a.reach_masks[4 * (size_t)e + wave] = (uint8_t)bits;
        } else {
            reach4(g0.x, g0.y, __float_as_uint(g2.w), sa2, sb1, sc2, g1.y, bx, by, valid, m0, m1, m2, m3);
        }
        // a row whose sixteen pixels are all saturated takes no more entries
        if ((live & 0xffffull) == 0ull) m0 = 0ull;
        if ((live & 0xffff0000ull) == 0ull) m1 = 0ull;
        if ((live & 0xffff00000000ull) == 0ull) m2 = 0ull;
        if ((live & 0xffff000000000000ull) == 0ull) m3 = 0ull;
        const int steps = build_lists(mylist, m0, m1, m2, m3, lane);
        int last_j = -1;                // batch position of the last entry this pixel took in this batch
        for (int t = 0; t < steps; ++t) {
            const int j = (int)mylist[4 * t + row];
            const float4 c0 = my[j * 3 + 0], c1 = my[j * 3 + 1];
            const float2 c2 = *reinterpret_cast<const float2*>(&my[j * 3 + 2]);
            const float dx = c0.x - pxf, dy = c0.y - pyf;
            const float u = c0.z * dx, w = c1.x * dy, bdx = c0.w * dx;
            const float power = 0.5f * (u * dx + w * dy) + bdx * dy;      // log2(e) x the exponent
            const float alpha = fminf(FDGS_ALPHA_MAX, c1.y * __builtin_amdgcn_exp2f(power));
            // No divergent branch (the two nested ones cost more scalar mask instructions per step than the step has
            // vector instructions): a lane that does not take the entry adds weight 0 and keeps T and `last`.
            {
                const bool ok = !done && power <= 0.0f && alpha >= FDGS_ALPHA_MIN;
                const float test_T = T * (1.0f - alpha);
                const bool stop = ok && test_T < FDGS_T_STOP;
                const bool take = ok && !stop;
                done = done || stop;
                const float wgt = take ? alpha * T : 0.0f;
                Cr += c1.z * wgt;
                Cg += c1.w * wgt;
                Cb += c2.x * wgt;
                D += c2.y * wgt;
                T = take ? test_T : T;
                last_j = take ? j : last_j;
            }
            // (no test for "all pixels saturated" per step: it cost two vector and six scalar instructions in every
            // step to save half a batch's steps once in a wave's life; the test at the head of the batch remains)
        }
        if (last_j >= 0) last = (base - r_begin) + (uint32_t)last_j + 1u;
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
// DPP exchanges inside a 16-lane row: lane ^ 8 (row_ror:8), lane -> 7 - lane inside each 8-lane half
// (row_half_mirror), lane ^ 2 and lane ^ 1 (quad_perm).
template <int CTRL>
__device__ __forceinline__ float dpp_from(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define DPP_ROR8 0x128
#define DPP_HALF_MIRROR 0x141
#define DPP_XOR2 0x4e
#define DPP_XOR1 0xb1

// One step of the transposing butterfly: two values in, one out; lanes where `upper` is false end up with the
// pair sum of x, the others with the pair sum of y (the pair is the lane and its exchange partner).
template <int CTRL>
__device__ __forceinline__ float fold2(float x, float y, bool upper) {
    const float keep = upper ? y : x, give = upper ? x : y;
    return keep + dpp_from<CTRL>(give);
}

// The first two steps of the butterfly for eight values (row_ror:8, then row_half_mirror) without the selects of
// fold2: whether a lane keeps x or y is decided by bit 3 (bit 2) of its position in the row, i.e. by its DPP bank of
// four lanes, so two adds with complementary bank masks write the two halves of one destination register -
// v_add_f32_dpp d, x, x on the banks that keep x, then d, y, y on the others.  Twelve instructions for the six
// folds instead of eighteen.  The s_nop covers the two wait states a DPP read needs after a vector write of the
// same register (inside the block the dependent reads are at least three instructions behind their writes; the
// results are next read by plain vector instructions, which need none).
// All 64 lanes must be active.
__device__ __forceinline__ void butterfly_top(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                              float v7, float& b0, float& b1) {
    float a0, a1, a2, a3;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
}

// The four waves of a tile walk the same batches (from the tile's last contributor down) and add their rows' sums
// into ONE accumulator per batch entry; after a workgroup barrier each wave hands a quarter of the batch's entries to
// memory - one request per (tile, entry, component) instead of one per (quadrant, entry, component): the
// memory-side atomic units are paced by requests.  Two accumulators alternate, so one barrier per batch orders
// everything: a wave adds into accumulator b & 1 only after the barrier of batch b - 1, which every wave reaches
// after it has flushed and cleared its quarter of batch b - 2.  (Measured against four private accumulators
// flushed per wave with no barrier: 0.221 -> 0.205 ms in the training step, bytes written per launch 122 -> 61 MiB;
// the accumulators take 9 KB instead of 18 and the kernel 70 registers instead of 87, so seven workgroups share a
// compute unit instead of five, which more than pays for the waves waiting at the barrier.)
__global__ void __launch_bounds__(256)
blend_bwd_kernel(fdgs_blend_bwd_args a, int tiles_x) {
    const int tile = a.tile_order ? (int)a.tile_order[blockIdx.x] : (int)blockIdx.x;
    if ((unsigned)tile >= gridDim.x) return;        // not a tile id: the caller's tile_order is not a permutation
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane >> 4, li = lane & 15;
    const int qx0 = tx * FDGS_TILE + (wave & 1) * 8, qy0 = ty * FDGS_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + 4 * (row & 1) + (li & 3), pyi = qy0 + 4 * (row >> 1) + (li >> 2);
    const int W = a.image_width, H = a.image_height;
    const bool inside = pxi < W && pyi < H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float bx = (float)qx0, by = (float)qy0;

    const uint32_t r_begin = a.ranges[2 * tile], r_end = a.ranges[2 * tile + 1];
    if (r_begin >= r_end) return;                   // the whole workgroup leaves: no barrier is missed
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
    const float tb = -T_final * (a.bg[0] * dLr + a.bg[1] * dLg + a.bg[2] * dLb);
    float T = T_final;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;

    // last contributor of each row's sixteen pixels, and of the wave
    int row_last = ncon;
    row_last = max(row_last, __builtin_amdgcn_update_dpp(0, row_last, DPP_XOR1, 0xf, 0xf, false));
    row_last = max(row_last, __builtin_amdgcn_update_dpp(0, row_last, DPP_XOR2, 0xf, 0xf, false));
    row_last = max(row_last, __builtin_amdgcn_update_dpp(0, row_last, DPP_HALF_MIRROR, 0xf, 0xf, false));
    row_last = max(row_last, __builtin_amdgcn_update_dpp(0, row_last, DPP_ROR8, 0xf, 0xf, false));
    const int last0 = __builtin_amdgcn_readlane(row_last, 0), last1 = __builtin_amdgcn_readlane(row_last, 16);
    const int last2 = __builtin_amdgcn_readlane(row_last, 32), last3 = __builtin_amdgcn_readlane(row_last, 48);
    const int wave_last = max(max(last0, last1), max(last2, last3));

    __shared__ float4 staged[4][BL_SLOTS * 3];          // (x y a2 b1) (c2 o r g) (b, address of the gradient record lo, hi, -)
    float4* __restrict__ my = staged[wave];
    __shared__ unsigned char lists[4][4 * 64];          // per wave: the four rows' survivor lists of a batch
    unsigned char* __restrict__ mylist = lists[wave];
    if (lane == 0) {
        my[BL_NULL * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        my[BL_NULL * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);      // opacity 0
        my[BL_NULL * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // which of the nine sums this lane hands to memory: the butterfly leaves sum 4 b1 + 2 b2 + b3 (bits of the lane's
    // position in its row) in every lane; even lanes store those eight, lane 1 of the row stores the ninth
    const int comp = (li == 1) ? 8 : (((li >> 1) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 3) & 1));
    const bool stores = (li & 1) == 0 || li == 1;
    const bool up2 = (li & 2) != 0;
    // The rows' sums are collected per batch entry in LDS (float64: LDS float32 atomics are an order of magnitude
    // slower on this part) and handed to memory once per batch, nine consecutive floats per entry: the memory-side
    // atomic units are paced by requests, not bytes, and one request per row and step - four records per
    // instruction - made them the bottleneck.
    __shared__ double sums[2][64 * 9];
    __shared__ int s_last[4];
    for (int q = (int)threadIdx.x; q < 2 * 64 * 9; q += 256) sums[0][q] = 0.0;
    if (lane == 0) s_last[wave] = wave_last;
    __syncthreads();
    const int tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    char* __restrict__ dbase = reinterpret_cast<char*>(a.dL_dgeom);

    int batch = 0;
    for (int hi = tile_last; hi > 0; hi -= 64, ++batch) {      // the same trips for the four waves: barriers inside
        double* __restrict__ mysum = sums[batch & 1];
        const int pos = hi - 1 - lane;        // 0-based position in the tile list, back to front
        const bool valid = pos >= 0;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
        float gb = 0.f;
        uint32_t reach = 0xfc00fc00u;         // (-inf, -inf): reaches nothing
        unsigned long long off = 0ull;
        if (valid) {
            const uint32_t id = a.point_list[r_begin + (uint32_t)pos];
            g0 = geom4[3 * (size_t)id + 0];
            g1 = geom4[3 * (size_t)id + 1];
            gb = a.geom[FDGS_GEOM_FLOATS * (size_t)id + 8];
            if (a.reach_masks) reach = a.reach_masks[4 * (size_t)(r_begin + (uint32_t)pos) + wave];
            else reach = __float_as_uint(a.geom[FDGS_GEOM_FLOATS * (size_t)id + 11]);
            off = (unsigned long long)dbase + (unsigned long long)id * (FDGS_GEOM_FLOATS * sizeof(float));
        }
        const float sa2 = -BL_LOG2E * g0.z, sb1 = -BL_LOG2E * g0.w, sc2 = -BL_LOG2E * g1.x;
        my[lane * 3 + 0] = make_float4(g0.x, g0.y, sa2, sb1);
        my[lane * 3 + 1] = make_float4(sc2, g1.y, g1.z, g1.w);
        my[lane * 3 + 2] = make_float4(gb, __uint_as_float((uint32_t)off), __uint_as_float((uint32_t)(off >> 32)), 0.f);
        unsigned long long m0, m1, m2, m3;
        // entries behind a row's last contributor cannot contribute to it
        if (a.reach_masks) {            // the forward pass's verdicts on this entry (it got at least as far as `last`)
            const uint32_t bits = valid ? reach : 0u;
            m0 = ballot64((bits & 1u) != 0u && pos < last0); m1 = ballot64((bits & 2u) != 0u && pos < last1);
            m2 = ballot64((bits & 4u) != 0u && pos < last2); m3 = ballot64((bits & 8u) != 0u && pos < last3);
        } else {
            reach4(g0.x, g0.y, reach, sa2, sb1, sc2, g1.y, bx, by, valid, m0, m1, m2, m3);
            m0 &= ballot64(pos < last0); m1 &= ballot64(pos < last1);
            m2 &= ballot64(pos < last2); m3 &= ballot64(pos < last3);
        }
        const int steps = build_lists(mylist, m0, m1, m2, m3, lane);
        const int jmin = hi - 1 - ncon;       // entry j of the batch is list position hi - 1 - j: before the pixel's last contributor <=> j > jmin
        for (int t = 0; t < steps; ++t) {
            const int j = (int)mylist[4 * t + row];
            const float4 c0 = my[j * 3 + 0], c1 = my[j * 3 + 1];
            const float o = c1.y;
            const float dx = c0.x - pxf, dy = c0.y - pyf;
            const float u = c0.z * dx, w = c1.x * dy, bdx = c0.w * dx;
            const float power = 0.5f * (u * dx + w * dy) + bdx * dy;      // log2(e) x the exponent
            const float G = __builtin_amdgcn_exp2f(power);
            const float alpha = fminf(FDGS_ALPHA_MAX, o * G);
            const bool contrib = (j > jmin) && power <= 0.0f && alpha >= FDGS_ALPHA_MIN;
            if (ballot64(contrib) == 0ull) continue;
            // No divergent branch: a lane that does not contribute runs the same arithmetic with alpha = 0 and
            // G = 0, which leaves its state unchanged bit for bit (T * rcp(1) = T; the colour behind folds as
            // 0 * c + 1 * acc = acc) and makes all nine of its partial sums exact zeros.
            const float4 c2 = my[j * 3 + 2];
            const float al = contrib ? alpha : 0.0f;
            const float Ge = contrib ? G : 0.0f;
            const float cr = c1.z, cg = c1.w, cb = c2.x;
            const float one_m = 1.0f - al;
            const float inv_one_m = __builtin_amdgcn_rcpf(one_m);       // 1 ulp; alpha <= 0.99
            T = T * inv_one_m;
            const float wgt = al * T;
            // acc = the blended colour behind this entry; the entry itself joins it for the ones in front
            float dL_dalpha = (cr - acc_r) * dLr + (cg - acc_g) * dLg + (cb - acc_b) * dLb;
            acc_r = al * cr + one_m * acc_r;
            acc_g = al * cg + one_m * acc_g;
            acc_b = al * cb + one_m * acc_b;
            const float v_r = wgt * dLr, v_g = wgt * dLg, v_b = wgt * dLb;
            dL_dalpha = dL_dalpha * T + tb * inv_one_m;
            const float v_o = Ge * dL_dalpha;
            const float g = o * v_o;                                    // dL/d(exponent), in the exponent's own scale
            const float gdx = g * dx, gdy = g * dy;
            const float v_mx = g * (u + c0.w * dy);                     // FDGS_DGEOM_* scaling, see fdgs_common.h
            const float v_my = g * (w + bdx);
            const float v_A = gdx * dx, v_B = gdx * dy, v_C = gdy * dy;
            // Eight of the nine partial sums go through a transposing butterfly inside the 16-lane row: each exchange
            // step halves the number of live registers, and after the fourth every lane of the row holds the row's
            // total of one sum (which one: `comp`).  The ninth takes four plain exchange-and-add steps.
            float b0, b1;
            butterfly_top(v_mx, v_my, v_A, v_B, v_C, v_o, v_r, v_g, b0, b1);
            float r = fold2<DPP_XOR2>(b0, b1, up2);
            r += dpp_from<DPP_XOR1>(r);
            float s = v_b + dpp_from<DPP_XOR1>(v_b);
            s += dpp_from<DPP_XOR2>(s);
            s += dpp_from<DPP_HALF_MIRROR>(s);
            s += dpp_from<DPP_ROR8>(s);
            const float out = (li == 1) ? s : r;
            if (stores && out != 0.0f)          // an exhausted or untouched row has nothing to add
                atomicAdd(&mysum[j * 9 + comp], (double)out);
        }
        // hand the batch's sums to memory and leave the LDS slots zero: slot q = 9 entry + component
        __syncthreads();                      // every wave's sums of this batch are in
#pragma unroll
        for (int k = 0; k < 3; ++k) {         // this wave's quarter: entries 16 wave .. 16 wave + 15
            const int q = 144 * wave + 64 * k + lane;
            if (64 * k + lane < 144) {
                const int e = (q * 7282) >> 16;                      // q / 9 for q < 576
                const unsigned long long bits =
                    atomicExch(reinterpret_cast<unsigned long long*>(&mysum[q]), 0ull);
                const float f = (float)__longlong_as_double((long long)bits);
                if (f != 0.0f) {
                    const float4 rec = my[e * 3 + 2];                // every wave staged the same batch
                    const unsigned long long goff = (unsigned long long)__float_as_uint(rec.y) |
                                                    ((unsigned long long)__float_as_uint(rec.z) << 32);
                    atomicAdd(reinterpret_cast<float*>(goff + 4 * (q - 9 * e)), f);
                }
            }
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
