/*
 * sahs_nerf.h -- C ABI of the MI355X-native deformable-NeRF volume-rendering hot path.
 *
 * The reference (jematy/SAHS-Deformable-Nerf) has no native boundary: its operator API for this
 * path is three Python seams.  Each entry point below cites the reference
 * interface it replaces; the Python side (sahs-deformable-nerf_amd/{train_utils,models,
 * volume_rendering_utils,nerf_helpers}.py) mirrors those seams one to one and calls down here.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless stated; the library never allocates,
 *    never synchronises and launches only on the given stream (hipStream_t passed as void*),
 *    so a caller may capture any sequence of calls in a hipGraph;
 *  - return 0 on success; non-zero = error, text via sahs_last_error() (thread-local); a call over zero rays (N == 0)
 *    succeeds without launching anything (its buffers may be null);
 *  - "rays" is the reference's packed ray table, row = [ro3, rd3, near, far, ...] with
 *    `ray_stride` floats per row (train_utils.py:255-261 builds it with stride 20);
 *  - precision: SAHS_F32 = exact fp32 (f32 MFMA), SAHS_BF16 = bf16 MFMA operands, fp32 accumulate.
 */
#ifndef SAHS_NERF_H
#define SAHS_NERF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SAHS_ABI_VERSION 1
#define SAHS_F32 0
#define SAHS_BF16 1
#define SAHS_BF16X3 3    /* near-fp32 on the bf16 pipe (AudioFaceModel, split chain only): operands split into bf16 hi + lo, three MFMAs per product,
                          * fp32 accumulate, in the radiance AND (since round 3) the deformation launches; x' = x + tanh(.) and the encodings stay fp32 arithmetic.
                          * SAHS_X3_DEFORM=f32 in the environment keeps the deformation launches on the fp32 kernel (the round-2 form) */
/* precision values 2 and 4 are reserved (A/B kernels of development builds, csrc/sahs_common.hpp; not in the shipped library) */

int sahs_abi_version(void);
const char *sahs_last_error(void);

/* Number of fp32 values in the canonical flat parameter buffer = the model's state_dict
 * (eval_stage_rays.py:299-303), tensors concatenated in state_dict order: 2,775,633. */
long sahs_param_count(void);
/* Size (in floats / 4-byte words) of the packed weight buffer and of the per-frame buffer. */
long sahs_packed_words(int precision);
long sahs_frame_words(void);

/* state_dict -> kernel layout.  Call after every parameter update (eval: once).
 * Replaces nothing in the reference (it keeps nn.Linear weights); feeds sahs_field_forward. */
int sahs_pack_weights(const float *flat_params, void *packed, int precision, void *stream);

/* Per-frame conditioning.  Replaces AudioNet (modules.py:43-73), rot_to_euler /
 * pose_to_euler_trans (models.py:482-504), encode_pose_fn (models.py:203-207) and the
 * per-point .repeat() of both (models.py:518,521), which every point-chunk of the reference
 * recomputes.  audio: (16,29); pose: 3x4 (or 4x4) row-major with row stride pose_ld.
 * frame[0:76] = driving, frame[80:116] = pose encoding, then the folded biases. */
int sahs_fold_conditioning(const float *flat_params, const float *audio, const float *pose, int pose_ld, float *frame,
                           void *stream);

/* get_ray_bundle (nerf_helpers.py:178-233). intrinsics = [fx, fy, cx, cy] (cx, cy relative);
 * c2w device pointer, row stride ld; ro, rd: (H, W, 3). */
int sahs_get_ray_bundle(int H, int W, float fx, float fy, float cx, float cy, const float *c2w, int ld, float *ro, float *rd,
                        void *stream);

/* Uniform draws in [0,1) for rays [ray0, ray0+N) x S samples -> out (N,S), as a pure function of (seed, stream_id, GLOBAL ray
 * index, sample index) (Philox4x32-10): the partition-invariant alternative to the reference's `torch.rand(z_vals.shape)`
 * (train_utils.py:110) and `torch.rand(...)` in sample_pdf_2 (nerf_helpers.py:469), whose values depend on how a frame is
 * chunked or sharded over GPUs.  stream_id separates the draws of one ray (0: t_rand, 1: u). */
int sahs_ray_uniforms(uint64_t seed, int stream_id, long ray0, long N, int S, float *out, void *stream);

/* Coarse depths (train_utils.py:93-113). t_rand (N,S) uniform draws or NULL (perturb off). z: (N,S). */
int sahs_stratified_depths(long N, int S, const float *rays, int ray_stride, int lindisp, const float *t_rand, float *z,
                           void *stream);

/* run_network + model forward (train_utils.py:9-50 -> models.py:514-528) for level 0 (coarse)
 * or 1 (fine), including pts = ro + rd*z (train_utils.py:115,168).  z: (N,S); raw: (N,S,16) =
 * [rgb3, seg12, sigma].  dbg: NULL, or N*S*88 floats receiving [N*S x 56: dx3, w2, first feature of
 * selected hidden layers][N*S x 32: grid features] (test seam). */
int sahs_field_forward(const void *packed, const float *frame, int level, long N, int S, const float *rays, int ray_stride,
                       const float *z, float *raw, float *dbg, int precision, void *stream);

/* volume_render_radiance_field (volume_rendering_utils.py:7-78) with the caller's background
 * overwrite (train_utils.py:135-136,184-185) when bg != NULL.  noise: (N,S) already scaled by
 * radiance_field_noise_std, or NULL.  Outputs rgb (N,15), disp (N), acc (N), weights (N,S), depth (N). */
int sahs_composite_forward(long N, int S, const float *raw, const float *z, const float *rays, int ray_stride,
                           const float *noise, const float *bg, int white_background, float *rgb, float *disp, float *acc,
                           float *weights, float *depth, void *stream);

/* z_vals_mid + sample_pdf_2 + cat + sort (train_utils.py:157-166, nerf_helpers.py:454-497).
 * weights: the full (N,S) composite weights (the [1:-1] slice is taken inside).  u: (N,nf) uniform
 * draws or NULL (det=True).  z_samples (N,nf) and inds (N,nf) int64 may be NULL; z_out: (N,S+nf). */
int sahs_resample(long N, int S, int nf, const float *z, const float *weights, const float *u, float *z_samples,
                  float *z_out, int64_t *inds, void *stream);

/* sample_pdf_2 alone (nerf_helpers.py:454-497): bins (N,nb), weights (N,nb-1), u (N,ns) or NULL (det=True)
 * -> samples (N,ns); inds (N,ns) int64 optional. */
int sahs_sample_pdf(long N, int nb, int ns, const float *bins, const float *weights, const float *u, float *samples,
                    int64_t *inds, void *stream);

/* predict_and_render_radiance (train_utils.py:72-206) for one ray chunk: the six launches above
 * in sequence.  Workspace (caller-owned): z_c (N,Sc), z_f (N,Sc+nf), raw (N,Sc+nf,16),
 * weights (N,Sc+nf).  Random draws in the reference's order (any may be NULL): t_rand (N,Sc),
 * noise_c (N,Sc), u (N,nf), noise_f (N,Sc+nf).  Outputs as the reference's 8-tuple:
 * rgb_c (N,15), disp_c, acc_c, rgb_f (N,15), disp_f, acc_f, w_bg (N) = weights_f[:, -1], depth_f. */
int sahs_render_rays(const void *packed, const float *frame, int precision, long N, const float *rays, int ray_stride, int Sc,
                     int nf, int lindisp, int white_background, const float *bg, const float *t_rand, const float *noise_c,
                     const float *u, const float *noise_f, float *z_c, float *z_f, float *raw, float *weights, float *rgb_c,
                     float *disp_c, float *acc_c, float *rgb_f, float *disp_f, float *acc_f, float *w_bg, float *depth_f,
                     void *stream);

/* ---- training path (BASELINE.json configs[4]: train_stage_rays_auto.py:437-499 calls loss.backward() through the path) ----
 * Gradients follow autograd of the reference graph.  fp32 only.  The backward is layer-wise over saved activations:
 * sahs_field_forward_save = sahs_field_forward that additionally writes sahs_act_words_per_sample() floats per sample (an opaque
 * buffer of N*S*words floats: one dense [N*S x width] array per layer; sahs_field_backward must be given the same P = N*S). */
long sahs_act_words_per_sample(void);
long sahs_field_backward_workspace_words(long P);
int sahs_field_forward_save(const void *packed, const float *frame, int level, long N, int S, const float *rays, int ray_stride,
                            const float *z, float *raw, float *act_out, void *stream);
/* d_raw (P,16) -> grad_flat (+=, canonical flat layout: every Linear weight/bias of warp/hyper/radiance nets and the feature
 * grid) and grad_cond (+=: [0:76] d driving, [80:116] d pose encoding).  P = N*S samples, P <= 4e6 per call. */
int sahs_field_backward(const float *flat_params, const float *frame, int level, long P, const float *act_in, const float *d_raw,
                        float *grad_flat, float *grad_cond, float *workspace, void *stream);
/* Arithmetic of the backward's dense-layer GEMMs (dW = dY^T X, dX = dY W; every model).  SAHS_BF16X3 (default): operands split into bf16
 * hi + lo on their way to the matrix pipe, three bf16 MFMAs per product, fp32 accumulation -- gradients within ~1e-5 of their scale of the
 * f32 form, far inside what two correct fp32 forwards differ by at the (leaky-)ReLU kinks; SAHS_F32: f32 MFMAs, exact
 * products (the A/B reference; SAHS_BWD_GEMM=f32 in the environment selects it at start-up).  precision < 0 queries.  Returns the
 * setting in force, -1 for an unknown value.  Process-wide. */
int sahs_backward_gemm_precision(int precision);
/* backward of sahs_composite_forward: any of d_rgb (N,15), d_disp, d_acc, d_depth, d_wlast (N: gradient of weights[:, -1], the
 * driver's 7th output), d_weights (N,S: gradient of the whole weights output, for the volume_render_radiance_field seam) may
 * be NULL -> d_raw (N,S,16). */
int sahs_composite_backward(long N, int S, const float *raw, const float *z, const float *rays, int ray_stride, const float *noise,
                            const float *bg, int white_background, const float *d_rgb, const float *d_disp, const float *d_acc,
                            const float *d_depth, const float *d_wlast, const float *d_weights, float *d_raw, void *stream);
/* The Stage-I objective of train_stage_rays_auto.py:455-468 over one ray batch, with the loss modules of nerf_helpers.py:14-62
 * (MaskMSELoss, MaskCrossEntropyLoss built with class_weights, :268-271):
 *   loss = sum over the given levels of  mean l2 + 0.02 mean ce + 0.005 sum_{c in 7,8} (masked mean_c l2 + masked mean_c ce)
 * map_coarse / map_fine (N,15) = the rendered [rgb3 | seg12] of a level (either may be NULL), target (N,target_ld >= 3), mask (N,12)
 * one-hot.  stats (64 floats): [0] loss, [1] mean l2 of the last level given (the script's psnr input), [2:14] the new sample_prob
 * (:466-468), [14:26] per-class ray counts (>= 1), [26] N.  One workgroup, fixed summation order. */
#define SAHS_LOSS_STATS_WORDS 64
int sahs_stage1_loss_forward(long N, const float *map_coarse, const float *map_fine, const float *target, int target_ld, const float *mask,
                             const float *class_weights, float *stats, void *stream);
/* sahs_composite_backward for a level whose rendered map enters that objective: the kernel forms d loss / d [rgb3 | seg12] of each ray
 * itself from (loss_map = the level's forward output, target, mask, stats of sahs_stage1_loss_forward, *loss_gscale = d objective /
 * d loss or NULL for 1) and adds it to d_rgb (may be NULL), so the (N,15) gradient never exists in memory. */
int sahs_composite_backward_loss(long N, int S, const float *raw, const float *z, const float *rays, int ray_stride, const float *noise,
                                 const float *bg, int white_background, const float *d_rgb, const float *d_disp, const float *d_acc,
                                 const float *d_depth, const float *d_wlast, const float *loss_map, const float *loss_target, int target_ld,
                                 const float *loss_mask, const float *loss_stats, const float *loss_gscale, float *d_raw, void *stream);
/* grad_cond[0:76] -> AudioNet parameter gradients in grad_flat (+=) and, optionally, grad_audio (16,29) (+=). */
int sahs_conditioning_backward(const float *flat_params, const float *audio, const float *grad_cond, float *grad_flat,
                               float *grad_audio, void *stream);

/* ---- every built architecture behind one family ----
 * The reference builds its field model with getattr(models, cfg.models.mask.type)(cfg) (eval_stage_rays.py:299,
 * train_stage_rays_auto.py:116); the same path serves
 *   SAHS_MODEL_AUDIO           AudioFaceModel, config/audio/<person>.yml (models.py:381-528) -- the functions above;
 *   SAHS_MODEL_NERFACE         NeRFaceModel, config/expression/person_2.yml / person_3.yml (models.py:189-378): warp + hyper
 *                              sheet on, 15-octave encodings, 1-D ambient coordinate, 4-layer trunk fed with the expression;
 *   SAHS_MODEL_NERFACE_STATIC  NeRFaceModel, config/expression/person_1.yml: use_warp False, use_ambient False.
 * flat_params is that model's state_dict in order (2,775,633 / 2,311,140 / 2,066,976 values); `driving` is the (16,29) audio
 * window for the AudioFaceModel and the 76-d expression vector (models.py:368 `driving.repeat`) for the NeRFaceModels;
 * rays, depths and outputs are the same, and sahs_get_ray_bundle, sahs_ray_uniforms, sahs_stratified_depths,
 * sahs_composite_forward, sahs_resample, sahs_sample_pdf are model-independent.  Training is SAHS_F32 for every model; rendering
 * additionally has SAHS_BF16 for all three (for SAHS_MODEL_NERFACE that means MIXED precision: deformation nets with split bf16 operands --
 * hi + lo, three MFMAs per product, as SAHS_BF16X3 -- + plain-bf16 radiance nets, see sahs_model_field_forward_split) and SAHS_BF16X3 for
 * SAHS_MODEL_AUDIO. */
#define SAHS_MODEL_AUDIO 0
#define SAHS_MODEL_NERFACE 1
#define SAHS_MODEL_NERFACE_STATIC 2
long sahs_model_param_count(int model);
long sahs_model_packed_words(int model, int precision);
long sahs_model_frame_words(int model);
int sahs_model_pack_weights(int model, const float *flat_params, void *packed, int precision, void *stream);
int sahs_model_fold_conditioning(int model, const float *flat_params, const float *driving, const float *pose, int pose_ld, float *frame,
                                 void *stream);
int sahs_model_field_forward(int model, const void *packed, const float *frame, int level, long N, int S, const float *rays,
                             int ray_stride, const float *z, float *raw, float *dbg, int precision, void *stream);
int sahs_model_render_rays(int model, const void *packed, const float *frame, int precision, long N, const float *rays, int ray_stride,
                           int Sc, int nf, int lindisp, int white_background, const float *bg, const float *t_rand, const float *noise_c,
                           const float *u, const float *noise_f, float *z_c, float *z_f, float *raw, float *weights, float *rgb_c,
                           float *disp_c, float *acc_c, float *rgb_f, float *disp_f, float *acc_f, float *w_bg, float *depth_f,
                           void *stream);

/* training path of any built model (fp32): the sahs_field_forward_save / sahs_field_backward pair above with a model id.
 * grad_cond: [0:76] d driving (for the NeRFaceModels this IS the gradient of the expression vector; for the AudioFaceModel
 * sahs_conditioning_backward carries it on through AudioNet), [80:116] d pose encoding. */
long sahs_model_act_words_per_sample(int model);
long sahs_model_field_backward_workspace_words(int model, long P);
int sahs_model_field_forward_save(int model, const void *packed, const float *frame, int level, long N, int S, const float *rays,
                                  int ray_stride, const float *z, float *raw, float *act_out, void *stream);
int sahs_model_field_backward(int model, const float *flat_params, const float *frame, int level, long P, const float *act_in,
                              const float *d_raw, float *grad_flat, float *grad_cond, float *workspace, void *stream);

/* Multiply-accumulates per sample evaluation that the field kernel of (model, precision) issues to the matrix pipe: the layer
 * program's zero-padded tiles and k-blocks, without the per-frame constant columns (folded into biases once per frame).  The
 * ALGORITHMIC count the roofline is quoted on is the reference's own (927,872 for AudioFaceModel). */
long sahs_model_executed_macs_per_sample(int model, int precision);
/* the same for a part of the network: 0 all of it, 1 the deformation nets, 2 the radiance net (sahs_model_field_forward_split) */
long sahs_model_executed_macs_part(int model, int precision, int part);

/* The 8-tuple of a ray side by side in one row of SAHS_ROW_COLUMNS floats -- the unit the multi-GPU all-gather of rendered
 * pixels moves and the layout run_one_iter_of_nerf's chunk loop fills in place, so no per-chunk
 * concatenation of eight tensors is needed (train_utils.py:298-319 does `torch.cat` per output). */
#define SAHS_ROW_COLUMNS 36
#define SAHS_ROW_RGB_C 0     /* 15: rgb3 + seg12 of the coarse pass */
#define SAHS_ROW_DISP_C 15
#define SAHS_ROW_ACC_C 16
#define SAHS_ROW_RGB_F 17    /* 15 */
#define SAHS_ROW_DISP_F 32
#define SAHS_ROW_ACC_F 33
#define SAHS_ROW_W_BG 34     /* weights[:, -1] of the fine pass (of the coarse pass when nf == 0) */
#define SAHS_ROW_DEPTH_F 35
/* sahs_composite_forward writing into such rows: the coarse pass (fine_pass 0) fills columns 0..16, the fine pass columns 17..35
 * (incl. w_bg = weights[:, -1] and depth); weights (N,S) stays a dense array (the resampling reads it). */
int sahs_composite_forward_rows(long N, int S, const float *raw, const float *z, const float *rays, int ray_stride, const float *noise,
                                const float *bg, int white_background, float *weights, float *rows, int row_ld, int fine_pass, void *stream);
/* z_vals_mid + sample_pdf_2 + cat + sort as sahs_resample, also returning the merge permutation: src (N,S+nf) int32, position s of the
 * sorted row holds element src[s] of cat(z, z_samples) (train_utils.py:166; equal values keep that order). */
int sahs_resample_merge(long N, int S, int nf, const float *z, const float *weights, const float *u, float *z_samples, float *z_out,
                        int32_t *src, void *stream);

/* The field evaluated in parts.  The deformation nets (WarpFieldMLP, HyperSheetMLP: one instance for both levels, models.py:231-254)
 * map a sample point x to (x', w); the fine pass's depths are sort(cat(coarse depths, new depths)), so the reference's fine pass
 * repeats the deformation of every coarse sample.  mode 0: the whole network for (N,S) depths z, also writing [x'0 x'1 x'2 w0 w1 . . .]
 * of every sample to xw[ray][xw_col0 + s] (xw: (N, xw_row, 8) floats); mode 1: the deformation nets only (raw unused); mode 2: the
 * radiance net of `level` only, for S samples per ray whose (x', w) are xw[ray][src[ray][s]] (z unused).  Same arithmetic on the same
 * operands as sahs_model_field_forward: bit-identical raw.  SAHS_F32 for the models with deformation nets (not SAHS_MODEL_NERFACE_STATIC);
 * SAHS_BF16 for SAHS_MODEL_AUDIO (packed from sahs_pack_weights(..., SAHS_BF16, ...)) and for SAHS_MODEL_NERFACE, where it means MIXED
 * precision: mode 1 runs the deformation nets with split bf16 operands (fp32 kernel under SAHS_X3_DEFORM=f32), mode 2 the bf16 radiance nets
 * (src may then be NULL: sample s of a ray is column s of xw), mode 0 both one after the other (xw_col0 must be 0); packed =
 * sahs_model_pack_weights(SAHS_MODEL_NERFACE, ..., SAHS_BF16, ...) = [bf16 radiance stream | fp32 pack | hi/lo streams].  SAHS_MODEL_NERFACE_STATIC has no deformation nets: its SAHS_BF16 path is sahs_model_field_forward.
 * PRECONDITION (not checked on the device): every src[ray][s] lies in [0, xw_row) -- it indexes xw's row of that ray (a permutation from
 * sahs_resample_merge satisfies it; ops.field_forward_split validates a caller-made one). */
int sahs_model_field_forward_split(int model, const void *packed, const float *frame, int precision, int level, int mode, long N, int S, const float *rays,
                                   int ray_stride, const float *z, float *raw, float *xw, int xw_row, int xw_col0, const int32_t *src,
                                   void *stream);

/* Training through the split evaluation (autograd of the same graph, train_stage_rays_auto.py:493, with the deformation nets
 * evaluated once per depth).  sahs_model_field_forward_split_save = sahs_model_field_forward_split (fp32) that also stores the
 * activations of the layers it runs into act_out: N*S*sahs_model_act_words_part(model, mode) floats (part/mode 0: whole network, 1:
 * deformation nets, 2: radiance nets).  sahs_model_field_backward_split walks `part` of the network backwards over activations saved by
 * the forward of the same part: the seam is the gradient w.r.t. (x', w), (P,8) rows [dx'0 dx'1 dx'2 . dw0 dw1 . .]: part 2 (radiance
 * alone) takes d_raw (P,16) and writes xw_grad_out; part 1 (deformation alone) starts from xw_grad_in; part 3 (everything) takes d_raw
 * and ADDS xw_grad_in (may be NULL) at the seam.  sahs_route_xw_grad scatters the fine pass's (N,Sc+nf,8) seam gradient through the merge
 * permutation src (sahs_resample_merge) into g_coarse (N,Sc,8) -- the xw_grad_in of the coarse pass -- and g_new (N,nf,8). */
long sahs_model_act_words_part(int model, int part);
int sahs_model_field_forward_split_save(int model, const void *packed, const float *frame, int level, int mode, long N, int S, const float *rays,
                                        int ray_stride, const float *z, float *raw, float *xw, int xw_row, int xw_col0, const int32_t *src,
                                        float *act_out, void *stream);
int sahs_model_field_backward_split(int model, const float *flat_params, const float *frame, int level, int part, long P, const float *act_in,
                                    const float *d_raw, const float *xw_grad_in, float *xw_grad_out, float *grad_flat, float *grad_cond,
                                    float *workspace, void *stream);
int sahs_route_xw_grad(long N, int Sc, int nf, const int32_t *src, const float *g_fine, float *g_coarse, float *g_new, void *stream);

/* SAHS_BF16 field kernels: NeRFMLP's LeakyReLU(0.01) (modules.py:252) is by default applied to the packed bf16 bit patterns (slope 0.0095 ..
 * 0.0106 depending on the mantissa: 15 % of a launch faster, PSNR cost measured in bench.py); sahs_bf16_exact_leaky(1) selects the kernel
 * instance that computes max(v, 0.01 v) in fp32 before rounding, as the reference would in bf16 -- e.g. to A/B a real checkpoint.
 * Process-wide; enable < 0 queries; SAHS_BF16_EXACT_LEAKY=1 in the environment selects it at first use.  Returns the state in force. */
int sahs_bf16_exact_leaky(int enable);

/* The fused backward walk (round 4; SAHS_MODEL_AUDIO; in the arithmetic sahs_backward_gemm_precision names: split-bf16 operands,
 * SAHS_BF16X3, or exact fp32 products, SAHS_F32 -- the reference's).  Replaces the ~38 GEMM launches sahs_model_field_backward_split makes
 * per part -- autograd of modules.py:254-295 (NeRFMLP), :371-390 (WarpFieldMLP), :444-462 (HyperSheetMLP) as driven by
 * train_stage_rays_auto.py:437-499 -- by two or three: one sample-major data-gradient chain and the weight gradients over job tables.  The (leaky-)ReLU masks come from SIGN BITS the saving forward writes beside the activations:
 * sahs_model_field_forward_split_save_bits = sahs_model_field_forward_split_save that also fills bits_out, N*S*sahs_model_bits_words_part(
 * model, mode) 32-bit words (mode 0: [deformation planes | radiance planes]).  sahs_model_field_backward_fused takes the same arguments as
 * sahs_model_field_backward_split plus bits_in (the planes of `part`; part 3: both, as written by a mode-0 save); workspace:
 * sahs_model_field_backward_fused_workspace_words(model, part, P) floats (-1: not built for the model).  Same results as the per-layer
 * walk up to summation order (tests/test_gpu_training.py). */
long sahs_model_bits_words_part(int model, int part);
int sahs_model_field_forward_split_save_bits(int model, const void *packed, const float *frame, int level, int mode, long N, int S, const float *rays,
                                             int ray_stride, const float *z, float *raw, float *xw, int xw_row, int xw_col0, const int32_t *src,
                                             float *act_out, uint32_t *bits_out, void *stream);
/* The same buffers written by the split-operand kernels (field_bf16x3.hip; `packed` = the SAHS_BF16X3 pack of the weights): the training
 * forward at three bf16 MFMAs per product instead of fp32 MFMAs.  Modes 1 (deformation nets) and 2 (radiance nets); the saved values are
 * those kernels' own (within a few 1e-6 relative of the fp32 kernel's), the signs are the signs of the values saved. */
int sahs_model_field_forward_split_save_bits_x3(int model, const void *packed, const float *frame, int level, int mode, long N, int S, const float *rays,
                                                int ray_stride, const float *z, float *raw, float *xw, int xw_row, int xw_col0, const int32_t *src,
                                                float *act_out, uint32_t *bits_out, void *stream);
long sahs_model_field_backward_fused_workspace_words(int model, int part, long P);
int sahs_model_field_backward_fused(int model, const float *flat_params, const float *frame, int level, int part, long P, const float *act_in,
                                    const uint32_t *bits_in, const float *d_raw, const float *xw_grad_in, float *xw_grad_out, float *grad_flat,
                                    float *grad_cond, float *workspace, void *stream);

/* sahs_model_render_rays writing rows[r * row_ld + column] instead of eight dense arrays (row_ld >= 36; columns 17..33 are
 * left untouched when nf == 0).  Workspace and draws as sahs_render_rays.  Optional extra workspace xw (N,Sc+nf,8) floats, src
 * (N,Sc+nf) int32, z_new (N,nf) floats: when all three are given (nf > 0; any model with deformation nets, SAHS_F32 or SAHS_BF16 -- required for the mixed-precision
 * SAHS_MODEL_NERFACE + SAHS_BF16, which exists as this chain only) the chain evaluates the deformation nets once per
 * depth (sahs_model_field_forward_split) -- 6 % less matrix work per frame, identical results. */
int sahs_model_render_rays_rows(int model, const void *packed, const float *frame, int precision, long N, const float *rays,
                                int ray_stride, int Sc, int nf, int lindisp, int white_background, const float *bg, const float *t_rand,
                                const float *noise_c, const float *u, const float *noise_f, float *z_c, float *z_f, float *raw,
                                float *weights, float *rows, int row_ld, float *xw, int32_t *src, float *z_new, void *stream);

/* ---- Stage-II refiner building block ----
 * The elementwise core of SPADELayer.forward followed by its SPADEBlock's LeakyReLU (nerf/_init_spade.py:130-139, :262-279):
 *   out = lrelu_slope( InstanceNorm2d(x; eps, biased variance, no affine) * (1 + gamma) + beta )
 * over `planes` = N*C contiguous planes of `hw` = H*W floats each (NCHW); gamma, beta, out have x's shape; slope 1 = no activation;
 * stats: sahs_spade_modulate_workspace_words(planes) floats of workspace (per plane: chunk sums and chunk sums of squares about the mean).  The convolutions that produce gamma / beta stay library
 * calls (MIOpen through PyTorch): sahs-deformable-nerf_amd/spade.py. */
long sahs_spade_modulate_workspace_words(long planes);
int sahs_spade_modulate(long planes, long hw, const float *x, const float *gamma, const float *beta, float eps, float slope, float *out,
                        float *stats, void *stream);

/* ---- launch probe (opt-in measurement aid; the one exception to "never synchronises") ----
 * While armed on the calling thread, every FIELD-kernel launch the library makes (from any entry point above) is bracketed by two HIP
 * events recorded on the launch stream, up to `capacity` launches (further ones are counted as dropped and run unprobed).
 * sahs_probe_read(i) waits for launch i to finish and returns its duration, the number of sample evaluations it covered and
 * kind = model << 16 | level << 12 | part << 8 | precision   (part: 0 whole network, 1 deformation nets, 2 radiance nets; precision = that
 * of the KERNEL launched, e.g. SAHS_F32 for the deformation launches of a mixed-precision model).
 * bench.py times the kernels of run_one_iter_of_nerf's own call chain with it.  Events belong to the device current at arm time. */
int sahs_probe_arm(int capacity);
int sahs_probe_disarm(void);
int sahs_probe_count(void);
int sahs_probe_dropped(void);
int sahs_probe_read(int i, int *kind, long *samples, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* SAHS_NERF_H */
