/*
 * lightplane_hip.h -- C ABI of liblightplane_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary of the Renderer / Splatter hot path.  Every entry
 * point replaces one Triton kernel launch site of the reference
 * (facebookresearch/lightplane @ 2024-08-07):
 *
 *   lp_renderer_forward   <- LightplaneFunction.forward  fw_kernel[grid](...)
 *                            lightplane/lightplane_renderer.py:505-555
 *                            (kernel: lightplane/triton_src/templates/renderer_fw.py:85-375)
 *   lp_renderer_backward  <- LightplaneFunction.backward bw_kernel[grid](...)
 *                            lightplane/lightplane_renderer.py:657-711
 *                            (kernel: lightplane/triton_src/templates/renderer_bw.py:89-627)
 *   lp_splatter_forward   <- LightplaneSplatterFunction.forward, BOTH launches
 *                            (features :505 and unit weights :507-539 in one march)
 *                            lightplane/lightplane_splatter.py:503-539
 *                            (kernels: templates/splatter_fw.py:71-165, :168-309 with MLP)
 *   lp_splatter_normalize <- weight clamp + divide, lightplane_splatter.py:541,584
 *   lp_splatter_backward  <- LightplaneSplatterFunction.backward bw_kernel[grid](...)
 *                            lightplane/lightplane_splatter.py:608,664
 *                            (kernels: templates/splatter_bw.py:75-180, :183-394 with MLP)
 *   lp_hash_randn         <- int_to_randn_kernel, triton_src/shared/rand_util.py:20-35
 *
 * Conventions
 *   - plain C: raw device pointers + host integers; no torch / C++ types.
 *   - the library never allocates device memory and keeps no state besides a
 *     thread-local error string; the caller owns every buffer.
 *   - all tensors are fp32, contiguous, resident on the device of `stream`.
 *   - accumulation targets (grad_*, splat feature/weight grids) MUST be zeroed by
 *     the caller: kernels accumulate with atomics (reference does the same:
 *     lightplane_renderer.py:470-476, 642-651; lightplane_splatter.py:404-410).
 *   - every function returns 0 on success, a negative LP_E* code on invalid
 *     arguments, or the positive hipError_t of a failed launch; lp_last_error()
 *     describes the last failure of the calling thread.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous.
 *
 * Grid-list layout (reference lightplane/misc_utils.py:25-46): one flat
 * [sum_g B*D_g*H_g*W_g, C] channels-last tensor; grid g begins at row
 * grids[g].row_offset, cell (b,z,y,x) is row ((b*D+z)*H+y)*W+x inside it.
 * Coordinates: x<->W, y<->H, z<->D; a grid with exactly one singleton spatial
 * dim is a plane sampled bilinearly (D==1: xy, H==1: xz, W==1: yz).
 *
 * MLP parameter layout (reference lightplane/mlp_utils.py:390-456): see LpMlp.
 */
#ifndef LIGHTPLANE_HIP_H
#define LIGHTPLANE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LP_VERSION 207 /* 0.2.0: per-grid base pointers (zero-copy grid-lists), fused bg-colour / alpha epilogue,
                           ray-embedding entry points; grad replicas removed
                           0.2.1: segment-parallel backward for small batches (LpRendererArgs.seg_prefix)
                           0.2.2: no struct change; lp_*_kernel_family() report family 3 (layer-looped MFMA kernels: Renderer
                                  decoders of 1-4 layers per MLP / up to 32 colour channels, MLP-Splatter of 2-4 layers and
                                  widths 16 / 32 / 64), 64-channel Splatter walks
                           0.2.3: no struct change; 64-channel Renderer grid-lists on family 3; the Renderer's family 1 is the tuned
                                  default decoder shape only (every other shallow shape reports 3), lp_splatter_kernel_family() no
                                  longer returns 2; lp_version() is NEGATIVE for a library built with -DLP_EXPERIMENTS
                           0.2.4: no struct change; lp_renderer_kernel_family() no longer returns 2 (2/2/2 x 64 decoders run the
                                  layer-looped family's two-block kernels and report 3); family 3 takes up to 256 beyond-far samples
                                  and two-grid decoders of hidden width 64 (heads of at most 2 layers, 16 / 32 grid channels)
                           0.2.5: no struct change; new test hook lp_renderer_backward_relu_dump(); the dX chains of the MFMA
                                  backwards take the gradient operand as two bf16 limbs (-DLP_DX_LIMBS=3 restores
                                  three)
                           0.2.6: LpRendererArgs.arithmetic (LP_ARITH_FP32: every product of the backward fp32-equivalent, selectable
                                  per call); lp_build_info(); lp_renderer_relu_dump_words() and dump twins for the layer-looped
                                  family (the dump of family 1 keeps its five words per sample); LpRendererArgs.march_order
                                  (LP_MARCH_SAMPLES_PER_WAVE: transposed march of the tuned backward for incoherent ray batches)
                           0.2.7: no struct change; LP_SEG_LEN 16 -> 8: LpRendererArgs.seg_prefix holds a record per 8 samples
                                  (lp_renderer_backward_segments() returns ceil(S / 8) for a small batch): batches of up to ~2 000 rays
                                  are dealt to the CUs in 8-sample segments, larger ones in 16-sample segments as before; the MFMA
                                  families take grid-lists of any byte size below 2^31 rows (were: below 4 GB) */

#define LP_MAX_GRIDS 8   /* grids per grid-list                         */
#define LP_MAX_LAYERS 8  /* layers per MLP                              */
#define LP_MAX_WIDTH 128 /* widest layer / grid channel count supported */
/* The forward pass saves the running -log T every LP_NLT_CKPT regular samples (and at the
 * last regular sample) and after EVERY beyond-far sample, so that the backward sweep
 * (far -> near) never reconstructs the transmittance across more than LP_NLT_CKPT
 * subtractions.  Every checkpoint is a float PAIR (hi, lo): -log T is accumulated as an unevaluated sum so
 * that the backward's subtraction of the same products recovers the intermediate values exactly.
 * One more pair per ray closes the list: (index of the last sample the forward marched, low word of
 * the final -log T) -- the backward starts there (see stop_neg_log_t).
 * O(N) memory: 2 * (ceil(S/LP_NLT_CKPT) + S_inf + 1) floats per ray. */
#define LP_NLT_CKPT 32
/* samples per state record of the segment-parallel march of a small batch (LpRendererArgs.seg_prefix); a workgroup marches one or more
 * such blocks (16 before 0.2.7) */
#define LP_SEG_LEN 8

/* error codes (negative; positive values are hipError_t) */
#define LP_OK 0
#define LP_EINVAL (-1)      /* malformed argument (see lp_last_error)          */
#define LP_EUNSUPPORTED (-2) /* shape outside what the kernels are built for    */
#define LP_ENULL (-3)       /* required pointer is NULL                        */

/* kernel selection hints (LpRendererArgs.kernel) */
#define LP_KERNEL_AUTO 0    /* MFMA kernel when the shape allows it, else generic */
#define LP_KERNEL_GENERIC 1 /* force the shape-generic VALU kernel               */
#define LP_KERNEL_MFMA 2    /* force the MFMA kernel (LP_EUNSUPPORTED if n/a)    */

/* arithmetic of the Renderer BACKWARD (LpRendererArgs.arithmetic).  The forward products -- the outputs, the backward's decoder
 * recompute and its ReLU decisions -- are fp32-equivalent in every mode (bf16x3: three exact bf16 limbs per operand, six limb
 * products, fp32 accumulation; the shape-generic kernels: plain fp32 FMAs).
 *   LP_ARITH_DEFAULT  the gradient operand of the dX chains and both operands of the weight gradients as TWO bf16 limbs (16
 *                     significand bits, three of nine limb products) where the kernel family does so -- lp_build_info() names the
 *                     limb counts of the library at hand; results stay inside 1e-4 of the fp32 reference
 *   LP_ARITH_FP32     the reference's arithmetic (triton_src/shared/const.py:9 ALLOW_TF32 = False): three limbs for every operand
 *                     of the dX chains, weight gradients on v_mfma_f32_16x16x4_f32.  The tuned family (kernel family 1) has
 *                     instantiations for it; every other shape runs the shape-generic fp32 kernels (family 0: slow, exact). */
#define LP_ARITH_DEFAULT 0
#define LP_ARITH_FP32 1

/* march order of the Renderer BACKWARD (LpRendererArgs.march_order; same results up to fp32 summation order).  The grid-gradient
 * scatter merges the taps of consecutive lanes that fall into one cell into a single row-contiguous atomic:
 *   LP_MARCH_RAYS_PER_WAVE     a wavefront = 32 consecutive rays at one sample: merges neighbouring rays (image-coherent batches:
 *                              consecutive rays = neighbouring pixels).  Default.
 *   LP_MARCH_SAMPLES_PER_WAVE  a wavefront = 32 consecutive samples of ONE ray (its rays one after the other): merges the samples a
 *                              ray spends in one cell -- for batches of unrelated rays (random training batches; the reference's speed
 *                              benchmark, tests/renderer_speed_benchmark.py:228-246), where every ray is its own run otherwise and
 *                              the backward is bound by the chip's atomic rate.  Tuned family, no beyond-far samples, no early
 *                              termination, >= 32 samples; ignored elsewhere (the rays-per-wavefront kernels run). */
#define LP_MARCH_RAYS_PER_WAVE 0
#define LP_MARCH_SAMPLES_PER_WAVE 1

typedef struct LpGrid {
  int32_t B, D, H, W;  /* batch and spatial extent                         */
  int64_t row_offset;  /* first row of this grid in the tensor that holds it */
  /* Zero-copy grid-lists (reference misc_utils.py:42-45 concatenates a list of grids on every call): a grid may
   * live in its own allocation.  NULL = the grid lives in LpGridList.data (one flat tensor, increasing
   * row_offsets); otherwise the [rows, C] tensor holding this grid, whose row `row_offset` (normally 0) is the
   * grid's first cell.  Gradient buffers mirror this through the *_list fields of the argument structs. */
  const float* data;
} LpGrid;

typedef struct LpGridList {
  const float* data;   /* [rows, channels] fp32; may be NULL when n_grids == 0 or every grid carries its own pointer */
  int32_t n_grids;     /* 0 .. LP_MAX_GRIDS                                 */
  int32_t channels;    /* C                                                  */
  int64_t n_rows;      /* total rows (for bounds checks)                     */
  LpGrid grids[LP_MAX_GRIDS];
} LpGridList;

/* rays: reference lightplane/ray_utils.py:19-57 */
typedef struct LpRays {
  int64_t n_rays;
  const float* directions;  /* [N,3] */
  const float* origins;     /* [N,3] */
  const int32_t* grid_idx;  /* [N]   batch element each ray belongs to */
  const float* near_t;      /* [N]   */
  const float* far_t;       /* [N]   */
  const float* encoding;    /* [N,encoding_dim] (Renderer: colour-MLP input width;
                               Splatter: splatted feature) */
  int32_t encoding_dim;
  int32_t row_length;       /* > 0: the batch is made of image rows in scanline order, row_length consecutive rays each (neighbouring
                               pixels).  A hint (0 = unknown; ABI 0.2.6, the former padding): the Splatter's backward walk then deals
                               2 x 4 PIXEL PATCHES to a wavefront -- four image rows walked column by column, alternating direction --
                               instead of 8 pixels of one row: the rays a wave gathers for are neighbours in both image directions
                               (cfg 3 backward -14 %).  Results are per ray and unchanged.  (The Renderer kernels ignore it: dealt
                               8 x 4 patches they measured SLOWER -- profiles/r06_ray_order.txt.) */
} LpRays;

/* ray-march schedule: reference naive_renderer.py:218-257, ray_util.py:48-58 */
typedef struct LpMarch {
  int32_t num_samples;         /* S  : equispaced in [near, far], both ends included */
  int32_t num_samples_inf;     /* S_inf : extra samples beyond far, linear in disparity */
  int32_t mask_out_of_bounds;  /* zero samples outside [-1,1]^3                      */
  int32_t contract_coords;     /* MeRF contraction (+ x0.5) before sampling          */
  double disparity_at_inf;
} LpMarch;

/* one MLP inside a flat parameter vector: all weights W_0..W_{n-1} ([in,out]
 * row-major, y = x @ W + b) followed by all biases b_0..b_{n-1}, starting at
 * float index `offset`.  dims[0] = input width, dims[l+1] = output width of layer l. */
typedef struct LpMlp {
  int32_t n_layers;                /* 0 .. LP_MAX_LAYERS */
  int32_t dims[LP_MAX_LAYERS + 1];
  int64_t offset;
} LpMlp;

typedef struct LpRendererArgs {
  LpRays rays;            /* encoding_dim == colour MLP input width            */
  LpGridList grid;        /* feature grid-list                                  */
  LpGridList color_grid;  /* n_grids == 0: single-grid mode (trunk MLP used)    */
  const float* scaffold;  /* NULL or [B, D*H*W] occupancy (0/1 floats)          */
  LpGrid scaffold_shape;  /* B,D,H,W of the scaffold (row_offset ignored)       */
  LpMarch march;
  /* decoder: trunk -> {opacity, color}; ReLU after every trunk layer and between
   * head layers; opacity = gain*softplus(raw); color = sigmoid(raw)              */
  const float* mlp_params;
  int64_t n_mlp_params;
  LpMlp trunk, opacity, color;
  int32_t color_chn;      /* real colour channels (<= color.dims[last]; the
                             remaining columns are zero padding, never evaluated) */
  float gain;
  float noise_sigma;      /* > 0: add sigma * hash_randn to the raw opacity      */
  int32_t noise_seed;
  int32_t kernel;         /* LP_KERNEL_*                                         */
  int32_t seg_forward_off; /* 1: with seg_prefix, the forward still marches every ray in one sweep (see seg_prefix) */
  /* forward outputs (written, not accumulated) */
  float* ray_length;      /* [N]                                                 */
  float* neg_log_t;       /* [N]  negative log transmittance after the last sample */
  float* feature;         /* [N, color_chn]                                      */
  float* neg_log_t_ckpt;  /* [N, n_ckpt, 2] written by forward, read by backward; NULL:
                             backward reconstructs from neg_log_t alone (less accurate) */
  /* backward inputs: upstream gradients (NULL = zeros) + neg_log_t from forward */
  const float* grad_ray_length; /* [N]            */
  const float* grad_neg_log_t;  /* [N]            */
  const float* grad_feature;    /* [N, color_chn] */
  /* backward outputs, accumulated with atomics: caller zero-fills. NULL = skip. */
  float* grad_grid;        /* like grid.data        */
  float* grad_color_grid;  /* like color_grid.data  */
  float* grad_mlp_params;  /* [n_mlp_params]        */
  float* grad_encoding;    /* [N, encoding_dim] (written, not accumulated) */
  /* per-grid gradient buffers for grids that carry their own LpGrid.data pointer: entry g (shaped like the tensor
   * grids[g].data points to, same row_offset) receives the gradient of grid g; NULL entries fall back to
   * grad_grid / grad_color_grid. */
  float* grad_grid_list[LP_MAX_GRIDS];
  float* grad_color_grid_list[LP_MAX_GRIDS];
  /* Fused epilogue of the module front-end (reference renderer_module.py:552-561; all optional):
   *   bg_color != NULL : feature[r, c] += T_r * bg_color[c],  T_r = exp(-neg_log_t[r])
   *   alpha    != NULL : alpha[r] = 1 - T_r (alpha_mode 1)  or  log T_r = -neg_log_t[r] (alpha_mode 2)
   * neg_log_t is always written raw (the backward needs it).  The backward takes grad_feature w.r.t. the
   * composited feature and grad_alpha [N] and folds both into the gradient of -log T. */
  const float* bg_color;    /* [color_chn] */
  float* alpha;             /* [N] */
  const float* grad_alpha;  /* [N] (backward) */
  int32_t alpha_mode;       /* 0 none, 1 alpha = 1 - T, 2 log transmittance */
  /* early ray termination (extension; the reference always marches every sample): > 0 = a wavefront stops
   * marching once -log T of all its rays has reached this value (their transmittance is below
   * exp(-stop_neg_log_t)); the backward skips the same samples.  ray_length / feature then miss
   * contributions of at most exp(-stop_neg_log_t) per unit of depth / colour, and neg_log_t is the value
   * reached at the stop (>= stop_neg_log_t) instead of the value after the last sample.  0 = exact. */
  float stop_neg_log_t;
  /* Segment-parallel backward for small batches (extension).  A backward sweep is serial along the ray, so a batch
   * with fewer rays than the GPU has wave slots (65 536 fill an MI355X once) leaves most of the chip idle.  With seg_prefix != NULL the
   * forward also saves, per ray and per block of LP_SEG_LEN regular samples, the state after the block's last sample
   * -- [N, lp_renderer_backward_segments(args), 8] floats: ray_length, feature[0..3], -log T (hi, lo), 0 -- and the
   * backward sweeps every block of a ray in its own workgroup (the part of d loss / d opacity_s that depends on the
   * samples behind the block comes from the saved sums).  grad_encoding is then ACCUMULATED (caller zero-fills).
   * The forward itself is segment-parallel too unless seg_forward_off is set: one workgroup per (128 rays, segment)
   * writes segment-local sums into the records and a combine pass chains them (compositing is associative) -- outputs
   * then differ from the single sweep by rounding (~1e-7 relative).
   * Pass the same pointer to forward and backward, and only when lp_renderer_backward_segments() > 1. */
  float* seg_prefix;
  int32_t arithmetic;     /* LP_ARITH_* (backward only; the forward is fp32-equivalent in every mode) */
  int32_t march_order;    /* LP_MARCH_* (backward only): how (ray, sample) pairs are dealt to the lanes of a wavefront */
} LpRendererArgs;

typedef struct LpSplatterArgs {
  LpRays rays;             /* encoding = splatted feature [N, encoding_dim]       */
  LpMarch march;
  LpGridList out;          /* output grid-list shape; out.data = feature accumulator
                              [rows, C] (zero-filled by caller)                   */
  float* out_feature;      /* == (float*)out.data, writable alias                 */
  float* out_weight;       /* [rows] splat-weight accumulator (zero-filled)       */
  /* MLP-splatter only (mlp.n_layers > 0): MLP(sample(input_grid) + encoding) is
   * splatted instead of the encoding                                            */
  LpGridList input_grid;
  const float* mlp_params;
  int64_t n_mlp_params;
  LpMlp mlp;               /* dims[0] == encoding_dim == input_grid.channels,
                              dims[last] == out.channels                          */
  int32_t kernel;
  int32_t march_order;     /* LP_MARCH_* (plain Splatter forward): LP_MARCH_SAMPLES_PER_WAVE walks 32 consecutive samples of ONE ray per
                              wavefront instead of 32 rays -- for batches of unrelated rays (reference tests/splatter_speed_benchmark.py):
                              the run merge of the atomic walk then works along the ray.  0 = rays per wavefront (image-coherent batches) */
  /* backward */
  const float* grad_out;   /* [rows, C] gradient w.r.t. the NORMALISED output grid */
  const float* weight;     /* [rows] un-clamped splat weights saved by forward     */
  float* grad_encoding;    /* [N, encoding_dim] (written)                          */
  float* grad_input_grid;  /* like input_grid.data (accumulated; MLP-splatter)     */
  float* grad_mlp_params;  /* [n_mlp_params]      (accumulated; MLP-splatter)      */
  float* grad_input_grid_list[LP_MAX_GRIDS]; /* per-grid buffers (see LpGrid.data); NULL entries -> grad_input_grid */
} LpSplatterArgs;

/* Ray-direction embedding of the module front-end, fused into one kernel (reference renderer_module.py:578-601
 * `_get_ray_embedding`: F.normalize -> harmonic embedding (ray_utils.py:181-212) -> Linear):
 *   d = directions / max(|directions|, 1e-12)
 *   emb = [sin(d_c 2^k)]_{c<3,k<n} ++ [sin(d_c 2^k + pi/2)]_{c<3,k<n} ++ d          (3 + 6 n values, index (p*3+c)*n+k)
 *   out[r, e] = bias[e] + sum_i weight[e, i] emb[i]                                  (torch.nn.Linear layout)
 * backward: grad_weight / grad_bias accumulate (atomics; caller zero-fills); directions get no gradient. */
typedef struct LpRayEmbedArgs {
  int64_t n_rays;
  const float* directions;  /* [N,3] */
  int32_t n_harmonics;      /* 0 .. 10 */
  int32_t out_dim;          /* E <= LP_MAX_WIDTH */
  const float* weight;      /* [E, 3 + 6 n] */
  const float* bias;        /* [E] */
  float* out;               /* [N, E] (forward) */
  const float* grad_out;    /* [N, E] (backward) */
  float* grad_weight;       /* [E, 3 + 6 n] */
  float* grad_bias;         /* [E] */
} LpRayEmbedArgs;

int lp_version(void); /* LP_VERSION; negative = built with -DLP_EXPERIMENTS (A/B timing switches), not a product build */
/* What this binary was built from, as one JSON object (static storage): "version", "src_hash" (sha256 over csrc/ *.hip, *.h,
 * build.py and this header, as lightplane_amd/csrc/build.py source_hash() computes it -- compare with the tree), "flags" (global
 * + per-file compiler flags), and per kernel family the limb counts / matrix instructions its backward was compiled with
 * ("tuned_bwd", "loop_bwd_deep", "loop_bwd_shallow", "mlp_splatter_bwd"), "test_hooks" (1 = the DUMP twins are in). */
const char* lp_build_info(void);
const char* lp_last_error(void);
/* sizeof() of the ABI structs as compiled into the library, for binding self-checks:
 * which = 0 LpGrid, 1 LpGridList, 2 LpRays, 3 LpMarch, 4 LpMlp, 5 LpRendererArgs,
 * 6 LpSplatterArgs, 7 LpRayEmbedArgs; anything else returns -1. */
int lp_abi_sizeof(int which);

/* Number of ray segments the backward of these arguments can be split into (see LpRendererArgs.seg_prefix): 1 when the
 * selected kernel has no segmented form, when the march has beyond-far samples or early termination, or when the
 * batch fills the GPU without it; otherwise ceil(num_samples / LP_SEG_LEN).  Depends on shapes only (no launch). */
int lp_renderer_backward_segments(const LpRendererArgs* args);

/* Which kernel family LP_KERNEL_AUTO selects for these arguments (no launch; shapes only):
 *   lp_renderer_kernel_family: 0 shape-generic VALU kernels, 1 tuned bf16x3 MFMA kernels of the default decoder (2/2/2 x 32),
 *                              3 layer-looped bf16x3 MFMA family (1-4 layers per MLP, one hidden width of 16 / 32 and <= 32
 *                              colour channels -- or width 64 / 64 grid channels with at most 2 layers per MLP and <= 4 colour
 *                              channels); 2 (the fp32-MFMA hidden-64 family of 0.1 - 0.2.3) is no longer returned
 *   lp_splatter_kernel_family: 0 shape-generic kernels, 1 run-merged walk (plain Splatter, C in {16,32,64}),
 *                              3 layer-looped bf16x3 MFMA MLP-Splatter (2-4 layers, widths 16 / 32 / 64, Cout 16 / 32);
 *                              2 (the two-layer fp32-MFMA family of 0.1 - 0.2.2) is no longer returned
 * The generic kernels are correctness anchors, one to two orders of magnitude slower. */
int lp_renderer_kernel_family(const LpRendererArgs* args);
int lp_splatter_kernel_family(const LpSplatterArgs* args);

int lp_renderer_forward(const LpRendererArgs* args, void* stream);
int lp_renderer_backward(const LpRendererArgs* args, void* stream);

int lp_splatter_forward(const LpSplatterArgs* args, void* stream);
/* feature[r, :] /= max(weight[r], 1e-5) for r in [0, n_rows) (in place). */
int lp_splatter_normalize(float* feature, const float* weight, int64_t n_rows, int32_t channels,
                          void* stream);
int lp_splatter_backward(const LpSplatterArgs* args, void* stream);

/* <- LightplaneRenderer._get_ray_embedding (lightplane/renderer_module.py:578-601) and its autograd backward */
int lp_ray_embedding_forward(const LpRayEmbedArgs* args, void* stream);
int lp_ray_embedding_backward(const LpRayEmbedArgs* args, void* stream);

/* out[i] = hash_randn(x1[i], x2[i], seed), i < n (test hook for the opacity-noise RNG). */
int lp_hash_randn(const int32_t* x1, const int32_t* x2, float* out, int64_t n, int32_t seed,
                  void* stream);

/* Debug / parity hook: integer corner rows of the Renderer march.  For grid g of
 * args->grid writes rows[(ray*S_tot + step)*K_tot + k] (int64, -1 = corner out of
 * range), K_tot = sum over grids of 8 (voxel) / 4 (plane), grids concatenated in
 * list order.  Used by the tests to prove bit-exact integer indexing vs the oracle. */
int lp_renderer_corner_rows(const LpRendererArgs* args, int64_t* rows, void* stream);

/* Debug / parity hook: lp_renderer_backward() through the DUMP twin of the kernel it would launch (same template, same
 * instruction sequence, stores added), which also writes the ReLU decisions of the backward's decoder recompute:
 * dump[(ray * S_tot + sample) * 5 + {0: trunk layer 1, 1: trunk layer 2 (the trunk output), 2: opacity hidden, 3: colour
 * hidden}] = bit f set when unit f is active, word 4 = 1 (sample contributed), 2 (visited, not contributing: beyond the ray's
 * last marched sample), 0 (never visited).  dump_words must be n_rays * S_tot * lp_renderer_relu_dump_words(args).  Every kernel
 * family has dump twins -- the tuned bf16x3 family (1; four-wave workgroups, default arithmetic), the layer-looped family (3) and the
 * shape-generic kernels (0): LP_EUNSUPPORTED for LP_ARITH_FP32, for the tuned family's eight-wave workgroups (> 64 beyond-far samples)
 * and in a library built without -DLP_TEST_HOOKS (lp_build_info() "test_hooks": 0).  The tests force these decisions onto the
 * fp64 oracle and require every gradient entry within 1e-4 (tests/test_gpu_config_scale.py::test_flips_are_flips). */
int lp_renderer_backward_relu_dump(const LpRendererArgs* args, uint32_t* dump, int64_t dump_words, void* stream);
/* Words per (ray, sample) of that dump for these arguments (shapes only, no launch), or LP_EUNSUPPORTED:
 *   family 1: 5 (above);
 *   family 3 (layer-looped): NB * n_sites + 1, NB = 32-bit words per ReLU site (1 up to 32 units, 2 for hidden width 64 / 64 grid
 *     channels), sites in the reference's evaluation order (naive_renderer.py:328-501) -- single grid-list: trunk layers 1 .. n_t,
 *     opacity hidden layers, colour hidden layers; two-grid decoder: relu(sampled feature), opacity hidden layers, relu(sampled
 *     colour feature), colour hidden layers -- word k * NB + b holds units 32 b .. 32 b + 31 of site k, the last word the
 *     visited flag (1 / 2 / 0 as above);
 *   family 0 (shape-generic, also LP_KERNEL_GENERIC): the same layout with NB = ceil(widest site / 32). */
int lp_renderer_relu_dump_words(const LpRendererArgs* args);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTPLANE_HIP_H */
