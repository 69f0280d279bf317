/*
 * fdgs.h — C ABI of the MI355X (gfx950) 4D-Gaussian-splatting hot path.
 *
 * One shared library, libfdgs.so.  Plain pointers and sizes only: every pointer
 * below is a DEVICE pointer unless its comment says "host".  The caller owns
 * every buffer (inputs, outputs, workspaces); the library allocates nothing,
 * keeps no state between calls, starts no threads, creates no streams and no
 * events: every entry point is re-entrant.  Every entry point enqueues its
 * kernels on the HIP stream passed in (`stream` is a hipStream_t, spelled
 * void* so that this header needs no HIP include; fdgs_deform_bwd may also use
 * a second stream the caller lends it) and returns without synchronising.  Return value: FDGS_OK (0) or a negative
 * fdgs_status; fdgs_strerror() names it.  No C++ exception crosses the ABI.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference repository):
 *   - the rasterizer entry points together are the body of
 *     diff_gaussian_rasterization.GaussianRasterizer.forward / backward, the
 *     third-party extension called at src/gaussian_renderer/__init__.py:104-113
 *     (settings built at :38-51) and differentiated by loss.backward() at
 *     src/runner/fdgs_trainer.py:140;
 *   - fdgs_deform_fwd / fdgs_deform_bwd are DeformationFields.forward
 *     (src/model/deformation.py:218-243, i.e. HexPlaneField.forward at
 *     src/model/hexplane.py:198-204 + DeformationNet.forward_dynamic at
 *     src/model/deformation.py:111-163) and its autograd backward.
 *
 * Layout invariants: row-major contiguous float32; quaternions (w,x,y,z)
 * (src/utils/general_utils.py:108-111); SH coefficients [N,16,3], coefficient
 * major / channel minor (src/model/gaussian_model.py:86-89); 4x4 matrices in the
 * row-vector form of src/scene/camera.py:48-52 (p_view = [x y z 1] @ M);
 * images planar [3,H,W].
 */
#ifndef FDGS_H
#define FDGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FDGS_OK = 0,
    FDGS_EINVAL = -1,      /* null pointer, negative size, unsupported configuration */
    FDGS_ELAUNCH = -2,     /* a kernel launch was rejected by the runtime            */
    FDGS_EWORKSPACE = -3,  /* workspace smaller than fdgs_*_workspace_bytes() asks   */
    FDGS_ETOOLARGE = -4    /* input larger than a fixed built-in limit of the entry point */
} fdgs_status;

const char* fdgs_strerror(int status);
/* ABI version of the loaded library (bumped on any struct change). */
int fdgs_abi_version(void);

#define FDGS_TILE 16                 /* tile edge in pixels                         */
#define FDGS_GEOM_FLOATS 12          /* floats per Gaussian in the packed record    */

/* Packed per-Gaussian record written by preprocess and gathered by the blend
 * kernels (48 bytes, one or two cache lines per gather):
 *   [0] x_pix  [1] y_pix  [2] conic A  [3] conic B
 *   [4] conic C [5] opacity [6] r [7] g
 *   [8] b      [9] view depth
 *   [10] int32 bits: radius in bits 0-27, SH clamp mask in bits 28-30 (bit 28 + c = channel c clamped)
 *   [11] two fp16 (x low, y high): half-widths of the box in which the splat's alpha can reach 1/255 (the blend
 *        kernels' per-entry cull), -inf = nowhere, +inf = never skip
 */

/* Activations that the reference's render() applies between the deformation field and the rasterizer
 * (src/gaussian_renderer/__init__.py:83-86, callables of src/model/gaussian_model.py:22-26), evaluated
 * inside the preprocess kernels instead: scale = exp(raw), rotation = raw / max(|raw|, 1e-12),
 * opacity = sigmoid(raw).  0 keeps the rasterizer's own interface (activated inputs). */
#define FDGS_ACT_SCALE_EXP 1
#define FDGS_ACT_ROTATION_NORMALIZE 2
#define FDGS_ACT_OPACITY_SIGMOID 4

/* ---- rasterizer forward, stage 1+2: per-Gaussian preprocess + inclusive scan --------------- */
typedef struct {
    int32_t num_points;
    int32_t sh_degree;               /* active degree 0..3                          */
    int32_t image_width, image_height;
    float tanfovx, tanfovy, scale_modifier;
    const float* viewmatrix;         /* [16]                                        */
    const float* projmatrix;         /* [16]                                        */
    const float* campos;             /* [3]                                         */
    const float* means3D;            /* [N,3]                                       */
    const float* scales;             /* [N,3] activated (exp already applied); NULL with cov3D_precomp */
    const float* rotations;          /* [N,4] normalised; NULL with cov3D_precomp   */
    const float* opacities;          /* [N]   activated (sigmoid already applied)   */
    const float* shs;                /* [N,16,3]; may be NULL when colors_precomp is given */
    const float* colors_precomp;     /* [N,3] RGB used as is (no SH evaluation, no clamp), or NULL
                                        (src/gaussian_renderer/__init__.py:89-101)  */
    const float* cov3D_precomp;      /* [N,6] upper triangle of the world-space covariance (scale modifier already
                                        applied), or NULL to build it from scales / rotations (:70-74)       */
    /* outputs */
    float* geom;                     /* [N,12] packed record (see above)            */
    int32_t* radii;                  /* [N]   0 = not visible                       */
    uint32_t* tiles_touched;         /* [N]                                         */
    uint32_t* point_offsets;         /* [N]   inclusive scan of tiles_touched; R = point_offsets[N-1] */
    /* fdgs_raster_scan_workspace_bytes(N) bytes.  On return its first three 32-bit words are: [0] the smallest and
     * [1] the largest depth bit pattern among the Gaussians that enter the tile lists (0xffffffff and 0 when there are
     * none), [2] R.  One 12-byte read gives the caller R and the depth range fdgs_raster_bin_sort can use.        */
    uint32_t* scan_workspace;
    int32_t activations;             /* FDGS_ACT_* flags: scales / rotations / opacities are raw parameters */
} fdgs_preprocess_fwd_args;

size_t fdgs_raster_scan_workspace_bytes(int32_t num_points);
int fdgs_raster_preprocess_fwd(const fdgs_preprocess_fwd_args* a, void* stream);

/* ---- rasterizer forward, stages 3-5: duplicateWithKeys, radix sort, identifyTileRanges ----- */
typedef struct {
    int32_t num_points;
    int32_t image_width, image_height;
    int64_t num_rendered;            /* R, read back by the caller from point_offsets[N-1] */
    const float* geom;               /* [N,12]                                      */
    const uint32_t* point_offsets;   /* [N]                                         */
    /* outputs */
    uint64_t* keys_unsorted;         /* [R] scratch of the sort (both key arrays ping-pong) */
    uint32_t* vals_unsorted;         /* [R] scratch                                 */
    uint64_t* keys_sorted;           /* [R] (tile_id << 32) | bits(depth), ascending */
    uint32_t* vals_sorted;           /* [R] the per-tile, depth-ordered point list  */
    uint32_t* ranges;                /* [tiles,2] (first, last+1); (0,0) when empty */
    void* sort_workspace;            /* fdgs_raster_sort_workspace_bytes(R) bytes   */
    size_t sort_workspace_bytes;
    /* Optional: words [0] and [1] of the preprocess call's scan workspace (bit patterns of the smallest and largest
     * listed depth).  With them the sort runs on (tile << k) | (bits(depth) - min), k = bit width of (max - min):
     * the same order on fewer digits (four passes instead of five at 1352 x 1014 for any depth range within a
     * factor of 2^16).  Both 0: unknown, all 32 depth bits are sorted.  The outputs are the same either way.       */
    uint32_t depth_bits_min, depth_bits_max;
    /* Optional output [tiles]: the tile ids ordered by decreasing list length (to about 6 %: a counting sort on a
     * logarithmic scale of the lengths; the order among tiles of one class is not defined), for the tile_order
     * argument of the blend calls.  Always a permutation of 0 .. tiles-1, also when R is 0.                   */
    uint32_t* tile_order;
    /* 0: the whole call.  The call can also be made in two parts around the caller's wait for num_rendered, so that
     * the GPU has work during that round trip: FDGS_BINSORT_COUNT right behind fdgs_raster_preprocess_fwd counts the
     * (Gaussian, tile) pairs per tile (it reads num_points, the image size, geom, point_offsets and the sort
     * workspace only; num_rendered may still be unknown), FDGS_BINSORT_FINISH, with the complete arguments and the
     * SAME sort workspace, does the rest.  Where the tile-binned sort does not apply (radix sort selected or more than
     * 32,768 tiles) the first part does nothing and the second everything.                                        */
    int32_t phase;
} fdgs_binsort_args;
#define FDGS_BINSORT_ALL 0
#define FDGS_BINSORT_COUNT 1
#define FDGS_BINSORT_FINISH 2

size_t fdgs_raster_sort_workspace_bytes(int64_t num_rendered);
int fdgs_raster_bin_sort(const fdgs_binsort_args* a, void* stream);

/* ---- rasterizer forward, stage 6: front-to-back alpha blending ------------------------------ */
typedef struct {
    int32_t image_width, image_height;
    const uint32_t* ranges;          /* [tiles,2]                                   */
    const uint32_t* point_list;      /* [R] = vals_sorted                           */
    const float* geom;               /* [N,12]                                      */
    const float* bg;                 /* [3]                                         */
    /* outputs */
    float* out_color;                /* [3,H,W]                                     */
    float* out_depth;                /* [1,H,W]  sum depth_i alpha_i T_i            */
    float* final_T;                  /* [H,W]                                       */
    uint32_t* n_contrib;             /* [H,W] 1-based list position of the last contributor */
    const uint32_t* tile_order;      /* NULL, or [tiles]: a permutation of the tile ids; workgroup i takes tile
                                        tile_order[i] (longest lists first shortens the tail of the launch) */
    /* NULL, or [R][4] bytes: for every list entry and each of its tile's four 8x8 quadrants, which of the
     * quadrant's four 4x4 pixel blocks the entry's 1/255-alpha ellipse reaches (bits 0-3).  Written for the
     * entries a quadrant gets to before all its pixels are saturated - a superset of what the backward asks for.
     * fdgs_raster_blend_bwd given the same array reads the bits instead of repeating the test.           */
    uint8_t* reach_masks;
} fdgs_blend_fwd_args;

int fdgs_raster_blend_fwd(const fdgs_blend_fwd_args* a, void* stream);

/* ---- rasterizer backward, stage 7: back-to-front blending gradients ------------------------- */
typedef struct {
    int32_t image_width, image_height;
    const uint32_t* ranges;
    const uint32_t* point_list;
    const float* geom;
    const float* bg;
    const float* final_T;
    const uint32_t* n_contrib;
    const float* dL_dcolor;          /* [3,H,W]                                     */
    /* output: per-Gaussian accumulators, MUST be zero on entry.  [N,12], to be handed to
     * fdgs_raster_preprocess_bwd as they are (slots 0-4 are in the scaling of the blending loop, which stages the
     * conic multiplied by -log2(e); preprocess_bwd applies the constants):
     *   [0] log2(e) dL/dx_pix [1] log2(e) dL/dy_pix [2] -2 dL/dA [3] -dL/dB [4] -2 dL/dC [5] dL/dopacity
     *   [6..8] dL/drgb  [9..11] unused                                            */
    float* dL_dgeom;
    const uint32_t* tile_order;      /* as in fdgs_blend_fwd_args */
    const uint8_t* reach_masks;      /* NULL, or the array fdgs_raster_blend_fwd filled for the same lists */
} fdgs_blend_bwd_args;

int fdgs_raster_blend_bwd(const fdgs_blend_bwd_args* a, void* stream);

/* ---- rasterizer backward, stage 8: per-Gaussian chain rule back to the inputs --------------- */
typedef struct {
    int32_t num_points;
    int32_t sh_degree;
    int32_t image_width, image_height;
    float tanfovx, tanfovy, scale_modifier;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* shs;
    const float* colors_precomp;     /* as in the forward (NULL: SH path)            */
    const float* cov3D_precomp;      /* as in the forward (NULL: scales/rotations)   */
    const float* geom;               /* [N,12] from the forward (radius, clamp mask) */
    const float* dL_dgeom;           /* [N,12] from fdgs_raster_blend_bwd            */
    /* outputs (fully written, no need to zero) */
    float* dL_dmeans3D;              /* [N,3]                                       */
    float* dL_dmeans2D;              /* [N,3]  (0.5*W*dL/dx_pix, 0.5*H*dL/dy_pix, 0) */
    float* dL_dscales;               /* [N,3]  w.r.t. the activated scales          */
    float* dL_drotations;            /* [N,4]  w.r.t. the normalised quaternion     */
    float* dL_dopacities;            /* [N]                                         */
    float* dL_dshs;                  /* [N,16,3]   (SH path)                        */
    float* dL_dcolors;               /* [N,3]      (colors_precomp path)            */
    float* dL_dcov3D;                /* [N,6]      (cov3D_precomp path)             */
    int32_t activations;             /* as in the forward call; the gradients are then w.r.t. the raw tensors */
    int32_t clear_dL_dgeom;          /* non-zero: every record of dL_dgeom that was read is written back as zeros, so a
                                        buffer the caller keeps is ready for the next fdgs_raster_blend_bwd (which
                                        accumulates into it) without a fill in between                          */
} fdgs_preprocess_bwd_args;

int fdgs_raster_preprocess_bwd(const fdgs_preprocess_bwd_args* a, void* stream);

/* ---- HexPlane deformation field ------------------------------------------------------------- */
#define FDGS_DEFORM_MAX_LEVELS 4
#define FDGS_DEFORM_WIDTH 128        /* hidden width W of every Linear (configs/train/dynerf.yaml:86) */
#define FDGS_DEFORM_FEAT 16          /* channels per plane (dynerf.yaml:98)         */
#define FDGS_DEFORM_OUT 59           /* 3 + 3 + 4 + 1 + 48 head outputs             */

/* Both settings take and return fp32 and accumulate in fp32.
 *   F32        v_mfma_f32_32x32x2_f32 (fp32 FMA chain).
 *   SPLIT_BF16 every operand is written exactly as the sum of three bf16 numbers (8+8+8 significant bits) and a
 *              product is the sum of the six bf16 x bf16 MFMA terms of weight >= 2^-16; the dropped terms are below
 *              2^-23 of the product, under one fp32 rounding.  2.7x fewer matrix-core cycles. */
#define FDGS_MATMUL_F32 0
#define FDGS_MATMUL_SPLIT_BF16 1

typedef struct {
    int32_t num_levels;                              /* len(multires)               */
    int32_t res[FDGS_DEFORM_MAX_LEVELS][4];          /* per level: x,y,z,t resolution */
    const float* aabb;                               /* [2,3] row0 = xyz_max, row1 = xyz_min (hexplane.py:134-136,174-180) */
    /* planes[level][ci], ci over (0,1),(0,2),(0,3),(1,2),(1,3),(2,3): [16, res[b], res[a]] */
    const float* planes[FDGS_DEFORM_MAX_LEVELS][6];
    const float* w0; const float* b0;                /* feature_out.0: [128, 16*levels], [128] */
    /* heads in the order pos, scales, rotations, opacity, shs */
    const float* w1[5]; const float* b1[5];          /* <head>_deform.1: [128,128],[128] */
    const float* w2[5]; const float* b2[5];          /* <head>_deform.3: [k,128],[k], k = 3,3,4,1,48 */
    int32_t matmul;                                  /* FDGS_MATMUL_*: how the fp32 matrix products are formed */
} fdgs_deform_params;

typedef struct {                                     /* gradient buffers, same shapes; accumulated INTO (+=) */
    float* planes[FDGS_DEFORM_MAX_LEVELS][6];
    float* w0; float* b0;
    float* w1[5]; float* b1[5];
    float* w2[5]; float* b2[5];
} fdgs_deform_param_grads;

typedef struct {
    int32_t num_points;
    fdgs_deform_params p;
    const float* xyz;                /* [N,3]                                       */
    const float* scales;             /* [N,3]  raw (log)                            */
    const float* rotations;          /* [N,4]  raw                                  */
    const float* opacity;            /* [N,1]  raw (logit)                          */
    const float* shs;                /* [N,16,3]                                    */
    const float* time;               /* [N,1] per-point time, or NULL to use time_scalar */
    float time_scalar;
    /* outputs */
    float* out_xyz; float* out_scales; float* out_rotations; float* out_opacity; float* out_shs;
    /* saved for backward (may be NULL when no gradient is needed):
     *   hidden [N,128], the pre-activation output of feature_out                   */
    float* save_hidden;
    /*   act, 768 bytes per point for ceil(N/32)*32 points: relu(hidden) as split matrix-operand pieces (layout
     *   private to the library); required together with save_hidden when p.matmul == FDGS_MATMUL_SPLIT_BF16 */
    void* save_act;
    /*   mask, 96 bytes per point (12 N 64-bit words): the ReLU decisions of hidden and of the five heads;
     *   required together with save_hidden when p.matmul == FDGS_MATMUL_SPLIT_BF16, unused otherwise */
    void* save_mask;
    /*   yhid, 768 bytes per point for ceil(N/16)*16 + 16 points: relu(hidden) once more as split pieces, in the
     *   points-as-K order of the backward's dW1 products (layout private to the library).  Optional
     *   (FDGS_MATMUL_SPLIT_BF16 only): with it the backward skips its own conversion pass and save_hidden may
     *   be NULL                                                                                        */
    void* save_yhid;
    /*   yfeat, 192 bytes per point for ceil(N/16)*16 + 16 points: the plane features as split pieces in the same
     *   points-as-K order (operand of the first layer's weight gradient).  Optional (FDGS_MATMUL_SPLIT_BF16 only):
     *   with it that product neither waits for the HexPlane backward nor needs the features recomputed  */
    void* save_yfeat;
    void* workspace;                 /* fdgs_deform_fwd_workspace_bytes(&p, N) bytes */
    size_t workspace_bytes;
    /* NULL, or the SH coefficients come as the two tensors GaussianModel keeps (src/model/gaussian_model.py:64-65):
     * `shs` is then _features_dc [N,1,3] and shs_rest is _features_rest [N,15,3]; their concatenation
     * (get_features, :86-89) is never materialised.  out_shs stays one [N,16,3] array.                    */
    const float* shs_rest;
} fdgs_deform_fwd_args;

size_t fdgs_deform_fwd_workspace_bytes(const fdgs_deform_params* p, int32_t num_points);
int fdgs_deform_fwd(const fdgs_deform_fwd_args* a, void* stream);

typedef struct {
    int32_t num_points;
    fdgs_deform_params p;
    const float* xyz;
    const float* time; float time_scalar;
    const float* hidden;             /* [N,128] saved by the forward; may be NULL when yhid is given */
    const void* act;                 /* saved by the forward (SPLIT_BF16 only, else may be NULL) */
    const void* mask;                /* 96 N bytes saved by the forward (SPLIT_BF16 only, else may be NULL) */
    const void* yhid;                /* the forward's save_yhid (SPLIT_BF16 only), or NULL: rebuilt from hidden */
    const void* yfeat;               /* the forward's save_yfeat (SPLIT_BF16 only), or NULL: recomputed */
    const float* planes_channel_last;/* the forward's workspace (channel-last plane copies) if the parameters have
                                        not changed since, else NULL: the copies are rebuilt inside the workspace */
    /* output cotangents; the residual inputs (scales, rotations, opacity, shs)
     * receive these unchanged, which the caller does itself                       */
    const float* g_xyz; const float* g_scales; const float* g_rotations; const float* g_opacity; const float* g_shs;
    /* outputs */
    float* d_xyz;                    /* [N,3] = g_xyz + gradient through the plane coordinates */
    fdgs_deform_param_grads dp;      /* accumulated into (+=); caller zeroes; entries may be NULL to skip */
    void* workspace;                 /* fdgs_deform_bwd_workspace_bytes(&p, N) bytes */
    size_t workspace_bytes;
    /* Optional, caller-owned, all three or none: a second hipStream_t and two hipEvent_t (created with
     * hipEventDisableTiming or not) on the device of `stream`.  With them the HexPlane scatter runs on
     * side_stream beside the weight-gradient products: the call records fork_event on `stream` (twice: before and
     * after its first pass over the points), makes side_stream wait for it, and makes `stream` wait for join_event
     * before it returns, so the caller still sees one stream-ordered operation.  NULL: everything is enqueued on `stream`.  Two calls that may overlap in time
     * (different streams, different threads) need their own triple and their own workspace.                */
    void* side_stream; void* fork_event; void* join_event;
} fdgs_deform_bwd_args;

size_t fdgs_deform_bwd_workspace_bytes(const fdgs_deform_params* p, int32_t num_points);
int fdgs_deform_bwd(const fdgs_deform_bwd_args* a, void* stream);

/* HexPlane features alone (forward only): HexPlaneField.forward / get_density, src/model/hexplane.py:182-204.
 * features [N, 16 * num_levels]; time as in fdgs_deform_fwd_args.  Only p.res, p.aabb and p.planes are read. */
typedef struct {
    int32_t num_points;
    fdgs_deform_params p;
    const float* xyz;                /* [N,3]                                       */
    const float* time;               /* [N,1] or NULL to use time_scalar            */
    float time_scalar;
    float* features;                 /* [N,32]                                      */
    void* workspace;                 /* fdgs_hexplane_features_workspace_bytes(&p, N) bytes */
    size_t workspace_bytes;
} fdgs_hexplane_features_args;

size_t fdgs_hexplane_features_workspace_bytes(const fdgs_deform_params* p, int32_t num_points);
int fdgs_hexplane_features(const fdgs_hexplane_features_args* a, void* stream);

/* ---- L1 image loss: scale * mean |a - b| over n floats (src/model/losses.py:7-8, called at
 * src/runner/fdgs_trainer.py:495) and its gradient w.r.t. a.  Two launches forward (the first can write the
 * gradient for an upstream gradient of 1 as well), one backward. */
typedef struct {
    int64_t n;
    const float* a;                  /* [n] rendered image(s)                       */
    const float* b;                  /* [n] ground truth                            */
    float scale;                     /* multiplies the mean (share of this view in a batch loss) */
    float* loss;                     /* [1]                                         */
    float* partial;                  /* [fdgs_l1_loss_partials(n)] scratch, twice that with psnr */
    float* psnr;                     /* NULL, or [1]: 20 log10(1 / sqrt(mean (a-b)^2)) of the same n elements -
                                        the per-iteration metric of src/model/metrics.py:7-28 (fdgs_trainer.py:131)
                                        from the same pass over the two images */
    float* grad_a;                   /* NULL, or [n]: d(loss)/da = scale / n * sign(a - b), what fdgs_l1_loss_bwd
                                        writes for grad_loss = 1 - for a caller whose loss IS this term (the
                                        training step: one pass over the two images instead of two) */
    const float* add;                /* NULL, or [1]: a device scalar added to the loss value (a term of the training
                                        loss computed elsewhere, e.g. the grid regulariser of fdgs_plane_regulation);
                                        it does not enter grad_a or psnr */
} fdgs_l1_fwd_args;
typedef struct {
    int64_t n;
    const float* a; const float* b;
    float scale;
    const float* grad_loss;          /* [1] device scalar: dL/d(loss)               */
    float* grad_a;                   /* [n] = grad_loss * scale / n * sign(a - b)   */
} fdgs_l1_bwd_args;
size_t fdgs_l1_loss_partials(int64_t n);
int fdgs_l1_loss_fwd(const fdgs_l1_fwd_args* a, void* stream);
int fdgs_l1_loss_bwd(const fdgs_l1_bwd_args* a, void* stream);

/* fdgs_l1_loss_fwd for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): it has no counterpart in the reference.  Same arguments and the same
 * result, bit for bit, as fdgs_l1_loss_fwd, but `partial` is ignored (may be NULL): the per-workgroup partial sums
 * go to a scratch array built into the library, one copy per GPU (128 KiB, held until the process ends).
 * Limit: n <= FDGS_L1_STATIC_MAX_N, larger n returns FDGS_ETOOLARGE.  Calls no allocating routine; the first call
 * on a device looks the array's address up once (hipGetSymbolAddress) and keeps it.  Every call on a GPU shares that
 * GPU's array: calls that may overlap in time on one GPU (two streams, two threads) race and must be serialised by
 * the caller; nothing has to start at zero (every partial that is read was written by the same call).
 * The array used is the CURRENT device's: `stream` must belong to the current device (a stream of another device
 * returns FDGS_EINVAL).  Under graph capture, make one call per device BEFORE the capture begins: the first call's
 * address lookup takes a lock and calls hipGetSymbolAddress, which a capture need not permit.                  */
#define FDGS_L1_STATIC_MAX_PARTIALS 16384
#define FDGS_L1_STATIC_MAX_N ((int64_t)FDGS_L1_STATIC_MAX_PARTIALS * 4096)      /* 67,108,864 floats */
int fdgs_l1_loss_fwd_static(const fdgs_l1_fwd_args* a, void* stream);

/* ---- 8-bit frames for playback: the conversion the reference applies to every rendered frame on the host,
 * to8b = (255 * np.clip(x, 0, 1)).astype(np.uint8) followed by transpose(1, 2, 0) (render.py:47,78), done on the
 * device, so that a frame leaves the GPU as C * H * W bytes instead of four times that in floats.
 * out[(y * W + x) * C + c] = (uint8) trunc(255.0f * min(max(image[c][y][x], 0), 1)): one fp32 multiply and a
 * truncation, bit for bit what numpy computes for finite inputs (a NaN pixel gives 0 here; numpy leaves it
 * undefined).                                                                                                 */
int fdgs_image_to8b(const float* image_chw, uint8_t* out_hwc, int32_t channels, int32_t height, int32_t width,
                    void* stream);

/* ---- densification statistics of one iteration (src/runner/fdgs_trainer.py:181-182,
 * src/model/gaussian_model.py:182-184): for every Gaussian with radii > 0,
 * max_radii2D = max(max_radii2D, radii), xyz_gradient_accum += |viewspace_grad.xy|, denom += 1. */
typedef struct {
    int32_t num_points;
    const int32_t* radii;            /* [N] (max over the views of the batch)       */
    const float* viewspace_grad;     /* [N,3] gradient of the 2-D means (summed over the views) */
    float* max_radii2D;              /* [N] or NULL                                 */
    float* xyz_gradient_accum;       /* [N] (the reference's [N,1])                 */
    float* denom;                    /* [N]                                         */
} fdgs_densify_stats_args;
int fdgs_densify_stats(const fdgs_densify_stats_args* a, void* stream);

/* ---- grid regulariser of the deformation field: DeformationWrapper.compute_regulation
 * (src/model/deformation_wrapper.py:52-92, compute_plane_smoothness of src/utils/plane_utils.py:3-9;
 * added to the loss at src/runner/fdgs_trainer.py:133-135).  Per plane [channels][h][w] (the parameter
 * layout): w_smooth * mean((second difference along h)^2) + w_l1 * mean|1 - t|.  The caller sets
 * w_smooth = plane_tv_weight for the spatial planes (0, 1, 3), time_smoothness_weight for the time
 * planes (2, 4, 5), and w_l1 = l1_time_planes_weight for the time planes, 0 otherwise.
 * loss (optional) receives the sum over the planes; grad[i] (optional per plane) receives
 * grad_scale * d(loss)/d(plane i), added to its contents when accumulate != 0. */
#define FDGS_REG_MAX_PLANES 12
typedef struct {
    int32_t num_planes;
    int32_t channels;
    const float* plane[FDGS_REG_MAX_PLANES];
    int32_t h[FDGS_REG_MAX_PLANES], w[FDGS_REG_MAX_PLANES];
    float w_smooth[FDGS_REG_MAX_PLANES], w_l1[FDGS_REG_MAX_PLANES];
    float* grad[FDGS_REG_MAX_PLANES];
    const float* grad_scale;         /* [1] device scalar, or NULL for 1            */
    int32_t accumulate;
    float* loss;                     /* [1] or NULL                                 */
    float* loss_partial;             /* [fdgs_plane_reg_partials(num_planes)] scratch, with loss */
} fdgs_plane_reg_args;
size_t fdgs_plane_reg_partials(int32_t num_planes);
int fdgs_plane_regulation(const fdgs_plane_reg_args* a, void* stream);

/* ---- initial scales: mean squared distance to the three nearest other points, `simple_knn._C.distCUDA2` at
 * src/model/gaussian_model.py:57 (with fewer than four points the missing neighbours count as +inf, as there).
 * points [N,3] -> mean_dist2 [N]. */
int fdgs_knn_mean_dist2(const float* points, int32_t num_points, float* mean_dist2, void* stream);

/* ---- Adam ------------------------------------------------------------------------------------
 * One step of torch.optim.Adam (no weight decay, no amsgrad) over up to 48 tensors in one launch:
 * the optimizer the reference builds at src/runner/fdgs_trainer.py:262-274 (eps 1e-15, one learning
 * rate per parameter group) and steps at :204.
 *   m += (1-beta1)(g-m);  v = beta2 v + (1-beta2) g^2;
 *   p -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps)      (step: the tensor's own count)  */
#define FDGS_ADAM_MAX_TENSORS 48
typedef struct {
    int32_t num_tensors;
    int32_t step[FDGS_ADAM_MAX_TENSORS];             /* 1-based, per tensor: torch.optim.Adam counts a tensor's steps from
                                                        its first gradient (state[p]["step"]), so tensors that join later
                                                        (the deformation field after the coarse stage) start at 1 */
    double beta1, beta2, eps;                        /* doubles, as Python holds them: 1-beta is formed before rounding to fp32 */
    float lr[FDGS_ADAM_MAX_TENSORS];
    int64_t numel[FDGS_ADAM_MAX_TENSORS];
    float* param[FDGS_ADAM_MAX_TENSORS];
    const float* grad[FDGS_ADAM_MAX_TENSORS];
    float* exp_avg[FDGS_ADAM_MAX_TENSORS];
    float* exp_avg_sq[FDGS_ADAM_MAX_TENSORS];
    /* 0: grad[t] is contiguous like the parameter.  L > 0: the gradient of tensor t is a column slice of a wider
     * row-major array - element e of the tensor is grad[t][(e / L) * grad_row_stride[t] + e % L] (the gradients of
     * _features_dc [N,1,3] and _features_rest [N,15,3] are the slices [:, :3] and [:, 3:] of dL/dSH [N,48]: L = 3
     * or 45, stride 48, and no copy of either is made)                                                    */
    int32_t grad_row_len[FDGS_ADAM_MAX_TENSORS];
    int64_t grad_row_stride[FDGS_ADAM_MAX_TENSORS];
} fdgs_adam_args;

int fdgs_adam_step(const fdgs_adam_args* a, void* stream);

/* ---- densification / pruning surgery ------------------------------------------------------------
 * The reference grows and shrinks the six per-Gaussian parameter tensors, their Adam moments and the
 * per-Gaussian statistics with chains of boolean-mask indexing and torch.cat
 * (src/runner/fdgs_trainer.py:304-366 optimizer surgery, :368-386 prune, :388-424 split, :426-449 clone,
 * :451-475 postfix, :477-492 prune / densify drivers).  Here: classify every Gaussian in one pass, turn the
 * classes into a row map with a stable three-way scan, and move every array through that map in one launch.
 *
 * Row order of the result (the order the reference's clone -> split -> prune sequence produces):
 *   [ survivors (classes KEEP and CLONE), in their old order | copies of the CLONE rows |
 *     first children of the SPLIT rows | second children of the SPLIT rows ]
 * SPLIT parents and DROP rows disappear.                                                              */
#define FDGS_CLS_KEEP 0
#define FDGS_CLS_CLONE 1
#define FDGS_CLS_SPLIT 2
#define FDGS_CLS_DROP 3

/* Classes for one densification (fdgs_trainer.py:487-492 with :426-430 and :392-397):
 * g = xyz_gradient_accum / denom, NaN -> 0;  big = max(exp(scaling)) > size_limit (percent_dense * extent);
 * CLONE if |g| >= grad_threshold and not big; SPLIT if g >= grad_threshold and big; else KEEP.        */
typedef struct {
    int32_t num_points;
    const float* xyz_gradient_accum;   /* [N] */
    const float* denom;                /* [N] */
    const float* scaling;              /* [N,3] log-space */
    float grad_threshold;
    float size_limit;
    uint8_t* cls;                      /* [N] out */
} fdgs_densify_classify_args;
int fdgs_densify_classify(const fdgs_densify_classify_args* a, void* stream);

/* Classes for one pruning pass (fdgs_trainer.py:477-484): DROP if sigmoid(opacity) < min_opacity, or - when
 * max_screen_size > 0 - if max_radii2D > max_screen_size or max(exp(scaling)) > world_limit (0.1 * extent);
 * else KEEP.                                                                                          */
typedef struct {
    int32_t num_points;
    const float* opacity;              /* [N] logits */
    const float* max_radii2D;          /* [N]; may be NULL when max_screen_size <= 0 */
    const float* scaling;              /* [N,3] log-space; may be NULL when max_screen_size <= 0 */
    float min_opacity;
    float max_screen_size;             /* <= 0: the size tests are off (the reference's None) */
    float world_limit;
    uint8_t* cls;                      /* [N] out */
} fdgs_prune_classify_args;
int fdgs_prune_classify(const fdgs_prune_classify_args* a, void* stream);

/* Row map of the result.  src[r] = old row that new row r is taken from, for r < counts[0] <= 2 N.
 * counts (device, 4 x int32): {rows after, survivors, clones, splits}.                              */
typedef struct {
    int32_t num_points;
    const uint8_t* cls;                /* [N] */
    int32_t* src;                      /* [2 N] out */
    int32_t* counts;                   /* [4] out */
    void* workspace; size_t workspace_bytes;   /* fdgs_surgery_workspace_bytes(N) */
} fdgs_surgery_plan_args;
size_t fdgs_surgery_workspace_bytes(int32_t num_points);
int fdgs_surgery_plan(const fdgs_surgery_plan_args* a, void* stream);

/* Moves up to 24 row-major arrays through the map in one launch: out[r] = in[src[r]].  Arrays flagged
 * FDGS_SURGERY_ZERO_NEW (Adam moments: zeros for added points, fdgs_trainer.py:354-355) get zeros in the
 * rows r >= rows_carried.  row_bytes is any positive size; multiples of 4 with 4-byte aligned pointers move
 * as words.                                                                                           */
#define FDGS_SURGERY_MAX_ARRAYS 24
#define FDGS_SURGERY_ZERO_NEW 1
typedef struct {
    int32_t rows_after;                /* counts[0] */
    int32_t rows_carried;              /* counts[1] */
    const int32_t* src;
    int32_t num_arrays;
    const void* in[FDGS_SURGERY_MAX_ARRAYS];
    void* out[FDGS_SURGERY_MAX_ARRAYS];
    int32_t row_bytes[FDGS_SURGERY_MAX_ARRAYS];
    int32_t flags[FDGS_SURGERY_MAX_ARRAYS];
} fdgs_surgery_apply_args;
int fdgs_surgery_apply(const fdgs_surgery_apply_args* a, void* stream);

/* Attributes of the split children (fdgs_trainer.py:402-407 with N = 2): for child j of 2 S (first children,
 * then second children), parent i = src[first_child_row + j]:
 *   xyz_out[row] = R(rotation[i]) (exp(scaling[i]) * noise[j]) + xyz[i],  scaling_out[row] = log(exp(scaling[i]) / 1.6)
 * noise is N(0,1) drawn by the caller ([2 S,3]; torch.normal(mean=0, std) is std * N(0,1)).  xyz / scaling /
 * rotation are the arrays BEFORE the surgery, xyz_out / scaling_out the ones after fdgs_surgery_apply. */
typedef struct {
    int32_t num_children;              /* 2 S */
    int32_t first_child_row;           /* survivors + clones */
    const int32_t* src;
    const float* xyz; const float* scaling; const float* rotation;   /* old arrays */
    const float* noise;                /* [2 S,3] */
    float* xyz_out; float* scaling_out;                               /* new arrays */
} fdgs_split_children_args;
int fdgs_split_children(const fdgs_split_children_args* a, void* stream);

/* Opacity reset (src/model/gaussian_model.py:173-176): logit <- inverse_sigmoid(min(sigmoid(logit), 0.01)),
 * in place; the moments of the opacity tensor are zeroed by the caller (fdgs_trainer.py:304-317).     */
int fdgs_reset_opacity(float* opacity, int32_t num_points, float cap, void* stream);

/* ---- SSIM (src/model/losses.py:13-53; joins the loss when lambda_dssim != 0, fdgs_trainer.py:136-138) ----
 * planes = batch * channels images of height x width.  window: the 2 radius + 1 weights of the 1-D Gaussian
 * (the reference's 2-D window is its outer product), zero padding.  plane_sum[p] = sum over the plane of
 *   ((2 mu1 mu2 + c1)(2 s12 + c2)) / ((mu1^2 + mu2^2 + c1)(s1 + s2 + c2));
 * the caller divides by the pixel count (mean over everything, or per image).  dmaps (optional) receives what
 * the backward needs.  Backward: grad_img1 = plane_scale[p] * d plane_sum[p] / d img1.               */
#define FDGS_SSIM_MAX_RADIUS 7
typedef struct {
    int32_t planes, height, width;
    const float* img1; const float* img2;            /* [planes, height, width] */
    int32_t radius;                                  /* window_size // 2, at most 7 */
    float window[2 * FDGS_SSIM_MAX_RADIUS + 1];
    float c1, c2;                                    /* 0.01^2, 0.03^2 */
    float* plane_sum;                                /* [planes] out */
    float* partial;                                  /* scratch, planes * fdgs_ssim_tiles(height, width) floats */
    float* dmaps;                                    /* NULL, or [3, planes, height, width] out */
} fdgs_ssim_fwd_args;
typedef struct {
    int32_t planes, height, width;
    const float* img1; const float* img2;
    int32_t radius;
    float window[2 * FDGS_SSIM_MAX_RADIUS + 1];
    const float* dmaps;                              /* written by the forward */
    const float* plane_scale;                        /* [planes] */
    float* grad_img1;                                /* [planes, height, width] out */
} fdgs_ssim_bwd_args;
size_t fdgs_ssim_tiles(int32_t height, int32_t width);
int fdgs_ssim_fwd(const fdgs_ssim_fwd_args* a, void* stream);
int fdgs_ssim_bwd(const fdgs_ssim_bwd_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDGS_H */
