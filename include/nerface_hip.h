/*
 * libnerface_hip.so -- C ABI of the MI355X (gfx950) implementation of NeRFace's ray-marching hot path.
 *
 * The reference (gafniguy/4D-Facial-Avatars) has no native/FFI layer: its boundary is the Python module
 * `nerf` (nerf/__init__.py:1-9).  Each entry point below replaces the stock-PyTorch op sequence of one
 * reference function; citations are relative to nerface_code/nerf-pytorch/ in the reference tree:
 *   H = nerf/nerf_helpers.py  V = nerf/volume_rendering_utils.py  T = nerf/train_utils.py  M = nerf/models.py
 *
 * Conventions: every pointer is a DEVICE pointer to contiguous row-major fp32 unless stated otherwise;
 * `stream` is a hipStream_t (NULL = default stream); kernels are enqueued asynchronously; the return
 * value is 0 on success or the hipError_t of the failed launch (nf_error_string() renders it), and
 * NF_EINVAL (-22) for an argument the library rejects.  No entry point allocates, frees or synchronises,
 * except nf_paper_pack's one-time upload of its (static) gather table.
 */
#ifndef NERFACE_HIP_H
#define NERFACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nf_stream_t; /* hipStream_t */

#define NF_EINVAL (-22)

/* ---- library ------------------------------------------------------------------------------------ */
int         nf_abi_version(void);            /* bumped on any signature / layout change; now 5        */
const char* nf_error_string(int code);
const char* nf_build_info(void);             /* "gfx950 <compiler> <date>"                            */

/* ---- K1: ray generation  -- replaces get_ray_bundle (H:68-123, meshgrid_xy H:29-41) --------------- */
/* c2w: 3 rows x >=4 floats with row stride `c2w_row_stride`; cx_w = float(W*cx), cy_h = float(H*cy)
 * computed by the caller in double (H:111-114).  ro, rd: (H, W, 3).  Bit-exact with the reference.   */
int nf_ray_bundle(int height, int width, float fx, float fy, float cx_w, float cy_h,
                  const float* c2w, int c2w_row_stride, float* ro, float* rd, nf_stream_t stream);

/* K1 for a training batch -- replaces the full-frame get_ray_bundle plus the index gathers of one iteration
 * (train_transformed_rays.py:302, 325-330).  sel: (n, 2) int64 {row, col}, or with sel_is_flat (n) int64 pixel indices
 * row * width + col; ro, rd: (n, 3), bit-identical to rows of
 * nf_ray_bundle's output; target (n, channels) gathered from image (H, W, channels) and bg_out (n, 3) from bg (H, W, 3),
 * each optional (NULL).  bad_flag: one zeroed int on the device, set to 1 if a selected pixel lies outside the image.   */
int nf_ray_batch(int height, int width, float fx, float fy, float cx_w, float cy_h, const float* c2w, int c2w_row_stride,
                 const int64_t* sel, int sel_is_flat, int64_t n, const float* image, int channels, const float* bg, float* ro,
                 float* rd, float* target, float* bg_out, int* bad_flag, nf_stream_t stream);

/* ---- K0: ray selection of a training iteration -- replaces np.random.choice(H * W, size=n, replace=False, p=probs)
 *      (train_transformed_rays.py:320-322) -----------------------------------------------------------------------------
 * n_select DISTINCT indices in [0, n_items), drawn without replacement with probabilities proportional to weights (>= 0, need not
 * be normalised; items of weight 0 are never chosen).  u: n_items uniform numbers in [0, 1) from the caller's generator.
 * idx_out (n_select) int64, ascending for n_select <= 8192 (else in arbitrary order); entries stay -1, at the end (and word 5 of
 * the workspace is set), if fewer than n_select weights are positive.  workspace: nf_weighted_choice_workspace_bytes() bytes of device memory.                          */
size_t nf_weighted_choice_workspace_bytes(void);
int nf_weighted_choice(const float* weights, const float* u, int64_t n_items, int n_select, int64_t* idx_out, void* workspace,
                       size_t workspace_bytes, nf_stream_t stream);

/* ---- K2: stratified coarse depths -- replaces T:56-76 -------------------------------------------- */
/* z: (n_rays, n_coarse).  t_vals: (n_coarse) = the caller's torch.linspace(0,1,n_coarse) table (T:50-55;
 * passing the table keeps torch's own linspace rounding).  t_rand NULL => perturb off.  Bit-exact.    */
int nf_sample_coarse(int64_t n_rays, int n_coarse, float near_z, float far_z, const float* t_vals,
                     const float* t_rand, float* z, nf_stream_t stream);
/* same with the reference's `lindisp` switch (T:65-66): lindisp != 0 spaces the depths linearly in disparity,
 * z = 1 / (1/near (1 - t) + 1/far t).  Bit-exact.                                                       */
int nf_sample_coarse_ex(int64_t n_rays, int n_coarse, float near_z, float far_z, const float* t_vals,
                        const float* t_rand, int lindisp, float* z, nf_stream_t stream);

/* ---- K3: positional encoder -- replaces positional_encoding (H:195-239) --------------------------- */
/* x: (n_rows, dim) -> out: (n_rows, dim*(include_input + 2*n_freq)), layout [x | sin f0 | cos f0 | ..] */
int nf_posenc(const float* x, int64_t n_rows, int dim, int n_freq, int include_input, float* out,
              nf_stream_t stream);

/* ---- K4: fused MLP -- replaces run_network (T:9-33) + ConditionalBlendshapePaperNeRFModel.forward
 *      (M:236-261) including pts = ro + rd*z (T:78), both positional encodings and all concats ------ */
#define NF_PAPER_NUM_PARAMS 26   /* state_dict order: layers_xyz.{0..5}.{weight,bias}, fc_feat.*, fc_alpha.*,
                                    layers_dir.{0..3}.*, fc_rgb.*   (layers_dir.3 is dead weight, Q3)  */
size_t nf_paper_packed_floats(void);      /* size of the MFMA-fragment-ordered weight image           */
size_t nf_paper_cond_floats(void);        /* size of the per-call bias table                          */
/* Gather the 26 live nn.Parameter storages into the fragment-ordered image (re-run after each
 * optimizer step).  `params` is a HOST array of 26 device pointers.                                   */
int nf_paper_pack(const float* const* params, float* packed, nf_stream_t stream);
/* HOST copy of the gather table behind nf_paper_pack (layout tests without a GPU):
 * out[i] = (state_dict index << 24) | flat element offset, 0xFF000000 = constant zero.                 */
int nf_paper_gather_table(uint32_t* out, size_t n /* must equal nf_paper_packed_floats() */);
/* Fold the per-call constant input columns into bias vectors: expr*1/3 (76) and latent (32) into
 * layers_xyz.0 / layers_xyz.3 (M:239-246), PE4(near), PE4(far) into layers_dir.0 (Quirk Q1, T:14).    */
int nf_paper_condition(const float* packed, const float* expr76, const float* latent32,
                       float near_z, float far_z, float* cond, nf_stream_t stream);
/* raw[(ray*S + s)*4 + {0,1,2,3}] = (rgb_raw, sigma_raw) for points ro + rd*z[ray, s].                 */
/* rd_view: per-ray vector whose z component feeds the "direction" encoding (T:14); NULL = rd.  It differs
 * from rd only on the reference's ray-direction ablation path (T:81-82, Quirk Q7).                     */
int nf_paper_mlp_fwd(const float* packed, const float* cond, const float* ro, const float* rd,
                     const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                     nf_stream_t stream);

/* ---- ConditionalBlendshapePaperNeRFModel.forward on pre-encoded inputs (M:236-261 as run_network calls it, T:20-24):
 * x87 (n_points, 87) = [PE10(xyz) | PE4(dirs)], expr (76), latent (32) -> out (n_points, 4).  Inference; `cond` is scratch
 * of nf_paper_cond_floats() floats.  The hot path never builds x87 (nf_paper_mlp_fwd encodes in registers).            */
int nf_paper_forward_encoded(const float* packed, const float* x87, const float* expr76, const float* latent32,
                             int64_t n_points, float* cond, float* out, nf_stream_t stream);

/* ---- K4, split-bf16 variant (eval): every GEMM as 3 bf16 MFMAs (W_hi x_hi + W_hi x_lo + W_lo x_hi) with f32
 * accumulation -- ~2^-16 relative per layer instead of 2^-24, 3x the throughput of the exact-f32 matrix rate.
 * Same arguments/semantics as nf_paper_mlp_fwd; `cond` is the same table (from the f32 image).  north_star's 1e-4 dB gate
 * (profiles/r06_gate_sensitivity.md): held on whole frames against a uniform-random or a 20 dB target, marginal at a 30 dB
 * target on the x1000 density head (1.3e-4 dB), held everywhere on the x40 head; frames 75 .. 88 / 112 .. 124 dB from exact f32. */
size_t nf_paper_packed_bf16_bytes(void);
int nf_paper_pack_bf16(const float* const* params, void* packed_bf16, nf_stream_t stream);
int nf_paper_mlp_fwd_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                          const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                          nf_stream_t stream);

/* ---- K4, split-fp16 variant ("f16x3", csrc/nf_mlp_f16.hip): same contract as nf_paper_mlp_fwd, every product evaluated as
 * three fp16 MFMAs with f32 accumulation on a per-layer power-of-two-scaled weight stream: fp32-CLASS accuracy (22 operand
 * significand bits; measured error against fp64 at the exact-f32 kernel's level; whole frames AND small ray sets as close to the
 * exact-f32 frame as that frame is to a float64 evaluation, at every target: profiles/r06_gate_sensitivity.md) at the split-bf16
 * kernel's speed.
 * `packed_f16` = nf_paper_packed_f16_bytes() bytes from nf_paper_pack_f16 (stream blocks + per-layer scales).
 * Valid while |activations| < 4094 (fp16 range / 2^4); replaces M:236-261 like nf_paper_mlp_fwd.
 * Range guard: if an activation leaves fp16's range the point's outputs are non-finite and the kernel sets a sticky 32-bit
 * flag at byte nf_paper_f16_flag_offset() of `packed_f16` (cleared by nf_paper_pack_f16); callers poll it per frame.      */
size_t nf_paper_f16_flag_offset(void);
size_t nf_paper_packed_f16_bytes(void);
int nf_paper_pack_f16(const float* const* params, void* stream_out, nf_stream_t stream);
int nf_paper_mlp_fwd_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                         const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream);
/* "f16x2" (round 5, csrc/nf_mlp_f16x2.hip): the same call on the same packed image with TWO fp16 products per weight -- activations enter
 * with their 11-bit `hi` half only, weights keep 22 bits: W_hi x_hi + W_lo x_hi, a third fewer MFMAs.  Inference only.  NOT an fp32-class
 * arithmetic: per-point outputs carry fp16's 2^-12 relative rounding of the activations, whole frames sit 60 .. 75 dB (x1000 density head) /
 * 85 .. 96 dB (x40 head) from the exact-f32 frame.  north_star's 1e-4 dB gate (round 6, profiles/r06_gate_sensitivity.md): held on whole
 * 512 x 512 frames against a uniform-random target (<= 5.6e-5 dB over 21 frames) and, on the x40 head, against targets the render approximates
 * to 20 .. 40 dB; MISSED against such targets on the x1000 head (4e-3 dB at 30 dB) and on ray sets of a few thousand rays.  Use
 * nf_paper_mlp_fwd_f16 where the gate must hold against real images; launch/eval_sharded.py measures it on the sequence being rendered.
 * Range and range guard as nf_paper_mlp_fwd_f16. */
int nf_paper_mlp_fwd_f16x2(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                           const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream);
/* training on the split-fp16 kernels ("f16x3": the three training GEMM kernels -- activation-saving forward, dX chain,
 * weight-gradient GEMMs -- at fp32-class accuracy on the 16-bit matrix pipe).  nf_paper_mlp_fwd_train_f16 fills `saved` like
 * nf_paper_mlp_fwd_train_bf16; nf_paper_mlp_bwd_f16 = nf_paper_mlp_bwd with the chain and the dW GEMMs on fp16 pairs, gradients
 * carried in block floating point (one power-of-two scale per point and layer; workspace as nf_paper_bwd_workspace_floats). */
int nf_paper_mlp_fwd_train_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                               const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream);
size_t nf_paper_packed_bwd_f16_bytes(void);
int nf_paper_pack_bwd_f16(const float* const* params, void* stream_out, nf_stream_t stream);
int nf_paper_mlp_bwd_f16(const float* packed, const void* packed_t_f16, const float* cond, const float* saved, const float* d_raw,
                         int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats, float* grads, nf_stream_t stream);
/* training forward on the split-bf16 kernel: also fills `saved` (f32, same layout as nf_paper_mlp_fwd_train); the
 * backward (nf_paper_mlp_bwd) stays on the exact-f32 kernels.                                                */
int nf_paper_mlp_fwd_train_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                                const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                                float* saved, nf_stream_t stream);

/* ---- K4 training path ---------------------------------------------------------------------------------
 * The reference trains through autograd (train_transformed_rays.py:389); here the forward saves every layer
 * output (`saved`, nf_paper_saved_floats(n_points) floats) and nf_paper_mlp_bwd turns d_raw (n_points,4)
 * into the gradients of all 26 parameters (reference layout, state_dict order, flattened) followed by the
 * 32 latent-code gradients: nf_paper_grad_floats() floats in total.  layers_dir.3 gets zeros (Quirk Q3).   */
size_t nf_paper_saved_floats(int64_t n_points);
int nf_paper_mlp_fwd_train(const float* packed, const float* cond, const float* ro, const float* rd,
                           const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                           float* saved, nf_stream_t stream);
size_t nf_paper_packed_bwd_floats(void);  /* transposed fragment image used by the backward chain        */
int nf_paper_pack_bwd(const float* const* params, float* packed_t, nf_stream_t stream);
size_t nf_paper_grad_floats(void);
size_t nf_paper_bwd_workspace_floats(int64_t n_points);
int nf_paper_mlp_bwd(const float* packed, const float* packed_t, const float* cond, const float* saved,
                     const float* d_raw, int64_t n_rays, int n_samples, float* workspace,
                     size_t workspace_floats, float* grads, nf_stream_t stream);

/* Measurement hook (bench.py's per-kernel training roofline; the reference has no counterpart): one backward in arithmetic
 * `precision` (0 exact f32, 1 split-bf16, 2 split-fp16; packed_t_any = the matching transposed image / stream) with HIP
 * events recorded on `stream` between its stages.  Synchronises the stream; stage_ms[3] (host) = {dX chain,
 * weight-gradient GEMMs, slab reduction + unpack} in milliseconds.                                                  */
int nf_paper_mlp_bwd_stage_ms(const float* packed, const void* packed_t_any, int precision, const float* cond,
                              const float* saved, const float* d_raw, int64_t n_rays, int n_samples, float* workspace,
                              size_t workspace_floats, float* grads, float* stage_ms, nf_stream_t stream);

/* Backward on the split-bf16 kernels: the dX chain and (unless exact_dw != 0) the weight-gradient GEMMs; bias/latent
 * reductions stay f32.  `saved` must have been written by nf_paper_mlp_fwd_train_bf16: the SPLIT TRAINING LAYOUT -- sections
 * n_points rounded up to 32 points long; every hidden layer's output as the weight-gradient kernel's (hi, lo) operand fragments
 * (csrc/nf_mlp_bf16_machinery.inc), positional encoding / dir slots as f32 rows, the ReLU bit masks the chain reads.
 * exact_dw != 0 runs the exact-f32 GEMMs instead; they read f32 rows: saved_f32 = `saved` converted by nf_split_saved_to_f32
 * (else NULL).
 * nf_split_saved_to_f32: model 0 paper / 1 second family; is_f16 = the forward was the split-fp16 one; out = nf_paper_saved_floats /
 * nf_lcode_saved_floats(n_points) floats in the exact-f32 training layout (x = hi + lo: 16 / 22 significand bits).        */
int nf_split_saved_to_f32(int model, const float* saved_split, int64_t n_points, int is_f16, float* out, nf_stream_t stream);
size_t nf_paper_packed_bwd_bf16_bytes(void);
int nf_paper_pack_bwd_bf16(const float* const* params, void* packed_t_bf16, nf_stream_t stream);
int nf_paper_mlp_bwd_bf16(const float* packed, const void* packed_t_bf16, const float* cond, const float* saved,
                          const float* d_raw, int64_t n_rays, int n_samples, float* workspace,
                          size_t workspace_floats, float* grads, int exact_dw, const float* saved_f32, nf_stream_t stream);

/* ---- K5: volume integrator -- replaces volume_render_radiance_field (V:7-75) + cumprod_exclusive
 *      (H:44-65) + the background overwrite of T:95-96 ----------------------------------------------- */
/* bg (n_rays,3) or NULL; noise (n_rays,S) already scaled by noise_std, or NULL.  Outputs: rgb (R,3),
 * disp (R), acc (R), weights (R,S).                                                                   */
int nf_volume_render_fwd(const float* raw, const float* z, const float* rd, const float* noise,
                         const float* bg, int64_t n_rays, int n_samples, int white_background,
                         float* rgb, float* disp, float* acc, float* weights, nf_stream_t stream);
/* d_raw (R,S,4) from d_rgb (R,3) (the trainer's loss only reaches rgb_map, TR:355-387).               */
int nf_volume_render_bwd(const float* raw, const float* z, const float* rd, const float* noise,
                         const float* bg, const float* d_rgb, int64_t n_rays, int n_samples,
                         int white_background, float* d_raw, nf_stream_t stream);

/* ---- second model family: ConditionalBlendshapeLearnableCodeNeRFModel.forward (M:590-636) + run_network (T:9-33), inference.
 * params: HOST array of 16 device pointers in state_dict order (layer1, layers_xyz.0..2, layers_dir.0, fc_alpha, fc_rgb,
 * fc_feat; weight then bias).  Same pack / condition / forward protocol and argument meaning as the paper model.      */
size_t nf_lcode_packed_floats(void);
size_t nf_lcode_cond_floats(void);
int nf_lcode_pack(const float* const* params, float* packed, nf_stream_t stream);
int nf_lcode_condition(const float* packed, const float* expr76, const float* latent32, float near_z, float far_z,
                       float* cond, nf_stream_t stream);
int nf_lcode_mlp_fwd(const float* packed, const float* cond, const float* ro, const float* rd, const float* rd_view,
                     const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream);
/* ConditionalBlendshapeLearnableCodeNeRFModel.forward on PRE-ENCODED inputs (replaces M:590-636 as run_network calls it, T:9-33):
 * x87 (n_points, 87) = [PE10(xyz) | PE4(dirs)] -> out (n_points, 4); `cond` = scratch of nf_lcode_cond_floats() floats.  Inference
 * only (ABI 4); the hot path never materialises x87 and uses nf_lcode_mlp_fwd. */
int nf_lcode_forward_encoded(const float* packed, const float* x87, const float* expr76, const float* latent32,
                             int64_t n_points, float* cond, float* out, nf_stream_t stream);
/* The same inference forward in split-bf16 arithmetic (see nf_paper_mlp_fwd_bf16): stream of
 * nf_lcode_packed_bf16_bytes() bytes packed from the same 16 tensors; `cond` as filled by nf_lcode_condition.          */
size_t nf_lcode_packed_bf16_bytes(void);
int nf_lcode_pack_bf16(const float* const* params, void* packed_bf16, nf_stream_t stream);
int nf_lcode_mlp_fwd_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                          const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                          nf_stream_t stream);
/* split-fp16 ("f16x3") forward of the second model family: contract, valid range and range guard as nf_paper_mlp_fwd_f16 */
size_t nf_lcode_packed_f16_bytes(void);
size_t nf_lcode_f16_flag_offset(void);
int nf_lcode_pack_f16(const float* const* params, void* stream_out, nf_stream_t stream);
int nf_lcode_mlp_fwd_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                         const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream);
int nf_lcode_mlp_fwd_f16x2(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                           const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream);   /* "f16x2": see nf_paper_mlp_fwd_f16x2 */
/* training the second family on the split-fp16 kernels (as nf_paper_mlp_fwd_train_f16 / nf_paper_mlp_bwd_f16) */
int nf_lcode_mlp_fwd_train_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                               const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream);
size_t nf_lcode_packed_bwd_f16_bytes(void);
int nf_lcode_pack_bwd_f16(const float* const* params, void* stream_out, nf_stream_t stream);
int nf_lcode_mlp_bwd_f16(const float* packed, const void* packed_t_f16, const float* cond, const float* saved, const float* d_raw,
                         int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats, float* grads, nf_stream_t stream);

/* Training of the same family, exact f32 (autograd through M:590-636 as the trainer drives it, TR:355-392):
 * forward that also fills `saved` (nf_lcode_saved_floats(n_rays*n_samples) floats), transposed weight image, and the
 * backward: grads = the 16 tensors in the order of nerf.models.LCODE_KEYS (layer1, layers_xyz.0..2, layers_dir.0,
 * fc_alpha, fc_rgb, fc_feat; weight then bias), flattened, followed by d latent (32).                                */
size_t nf_lcode_saved_floats(int64_t n_points);
int nf_lcode_mlp_fwd_train(const float* packed, const float* cond, const float* ro, const float* rd, const float* rd_view,
                           const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream);
size_t nf_lcode_packed_bwd_floats(void);
int nf_lcode_pack_bwd(const float* const* params, float* packed_t, nf_stream_t stream);
size_t nf_lcode_grad_floats(void);
size_t nf_lcode_bwd_workspace_floats(int64_t n_points);
int nf_lcode_mlp_bwd(const float* packed, const float* packed_t, const float* cond, const float* saved, const float* d_raw,
                     int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats, float* grads,
                     nf_stream_t stream);

/* Training of the same family on the split-bf16 kernels (forward, dX chain, weight-gradient GEMMs; reductions f32).   */
int nf_lcode_mlp_fwd_train_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                                const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                                float* saved, nf_stream_t stream);
size_t nf_lcode_packed_bwd_bf16_bytes(void);
int nf_lcode_pack_bwd_bf16(const float* const* params, void* packed_t_bf16, nf_stream_t stream);
int nf_lcode_mlp_bwd_bf16(const float* packed, const void* packed_t_bf16, const float* cond, const float* saved,
                          const float* d_raw, int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats,
                          float* grads, nf_stream_t stream);

/* ---- BASELINE config 1: tiny_nerf.py (reference tiny_nerf.py:12-181) ----------------------------------------------
 * nf_tiny_mlp_fwd = compute_query_points_from_rays' pts = ro + rd*depth (tiny_nerf.py:59-63) + positional_encoding(., 10)
 * + VeryTinyNerfModel.forward (63 -> 128 -> 128 -> 4).  params: HOST array of 6 device pointers
 * (layer1.weight, layer1.bias, layer2.*, layer3.*).  depth: (n_rays, n_samples) if depth_per_ray else (n_samples).
 * nf_render_volume_density = render_volume_density (tiny_nerf.py:68-107) -> rgb (R,3), depth_map (R), acc (R).     */
size_t nf_tiny_packed_floats(void);
int nf_tiny_pack(const float* const* params, float* packed, nf_stream_t stream);
int nf_tiny_mlp_fwd(const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray,
                    int64_t n_rays, int n_samples, float* raw, nf_stream_t stream);
int nf_render_volume_density(const float* raw, const float* depth, int64_t n_rays, int n_samples, float* rgb,
                             float* depth_map, float* acc, nf_stream_t stream);

/* ---- training the tiny path: autograd of run_one_iter_of_tinynerf (tiny_nerf.py:111-159) under the trainer's rgb loss
 *      (tiny_nerf.py:291-302), exact f32.  nf_tiny_mlp_fwd_train also writes the activations the backward needs (`saved`,
 *      nf_tiny_saved_floats(n_points) floats); nf_render_volume_density_bwd turns d loss / d rgb (R,3) into d_raw (R,S,4);
 *      nf_tiny_mlp_bwd turns d_raw into the six parameter gradients, concatenated in state_dict order
 *      [layer1.weight (128,63) | layer1.bias | layer2.weight | layer2.bias | layer3.weight (4,128) | layer3.bias]
 *      (nf_tiny_grad_floats() floats).  packed_t = nf_tiny_pack_bwd (transposed fragment image of layer2 / layer3).       */
size_t nf_tiny_saved_floats(int64_t n_points);
int nf_tiny_mlp_fwd_train(const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray,
                          int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream);
int nf_render_volume_density_bwd(const float* raw, const float* depth, const float* d_rgb, int64_t n_rays, int n_samples,
                                 float* d_raw, nf_stream_t stream);
size_t nf_tiny_packed_bwd_floats(void);
int nf_tiny_pack_bwd(const float* const* params, float* packed_t, nf_stream_t stream);
size_t nf_tiny_grad_floats(void);
size_t nf_tiny_bwd_workspace_floats(int64_t n_points);
int nf_tiny_mlp_bwd(const float* packed_t, const float* saved, const float* d_raw, int64_t n_rays, int n_samples,
                    float* workspace, size_t workspace_floats, float* grads, nf_stream_t stream);

/* ---- BASELINE config 1 read literally ("4-layer MLP"): the tiny path with the reference's FlexibleNeRFModel
 *      (nerf/models.py:351-422) constructed as FlexibleNeRFModel(num_layers = L, hidden_size = 128, num_encoding_fn_xyz = 10,
 *      include_input_xyz = True, use_viewdirs = False), L = 2 .. 5 (ABI 5):
 *        PE(63) -> layer1 (Linear 128, no activation, models.py:402) -> (L - 1) x [layers_xyz.k: Linear 128 + ReLU, models.py:403-410]
 *        -> fc_out (Linear 4, models.py:422).
 *      The entry points mirror nf_tiny_* with num_layers in front (other values: NF_EINVAL / size 0).  params: HOST array of
 *      4 + 2 (L - 1) device pointers in state_dict order (layer1.weight, layer1.bias, layers_xyz.0.weight, layers_xyz.0.bias, ...,
 *      fc_out.weight, fc_out.bias); nf_flex_mlp_bwd returns the gradients concatenated in the same order
 *      (nf_flex_grad_floats(L) floats).  Compositing: nf_render_volume_density(_bwd), as for the tiny path.                    */
size_t nf_flex_packed_floats(int num_layers);
int nf_flex_pack(int num_layers, const float* const* params, float* packed, nf_stream_t stream);
int nf_flex_mlp_fwd(int num_layers, const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray,
                    int64_t n_rays, int n_samples, float* raw, nf_stream_t stream);
size_t nf_flex_saved_floats(int num_layers, int64_t n_points);
int nf_flex_mlp_fwd_train(int num_layers, const float* packed, const float* ro, const float* rd, const float* depth,     /* < 2^22 points */
                          int depth_per_ray, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream);
size_t nf_flex_packed_bwd_floats(int num_layers);
int nf_flex_pack_bwd(int num_layers, const float* const* params, float* packed_t, nf_stream_t stream);
size_t nf_flex_grad_floats(int num_layers);
size_t nf_flex_bwd_workspace_floats(int num_layers, int64_t n_points);
int nf_flex_mlp_bwd(int num_layers, const float* packed_t, const float* saved, const float* d_raw, int64_t n_rays, int n_samples,
                    float* workspace, size_t workspace_floats, float* grads, nf_stream_t stream);

/* ---- eval post-processing on the device -- replaces cast_to_image (eval_transformed_rays.py:184-190) and
 *      torch_normal_map(depthmap, focal, weights, clean=True) (eval_transformed_rays.py:84-119) ------------------------
 * rgb (H,W,3) -> rgb_u8 (H,W,3) = uint8(clamp(x,0,1)*255);  depthmap (H,W) [+ weights (H,W)] -> normals_u8 (H-1,W-1,3).
 * Either output may be NULL.  cx_w = cx*W, cy_h = cy*H as in nf_ray_bundle.                                              */
int nf_eval_postprocess(const float* rgb, const float* depthmap, const float* weights, int height, int width, float fx,
                        float fy, float cx_w, float cy_h, uint8_t* rgb_u8, uint8_t* normals_u8, nf_stream_t stream);

/* ---- K6: inverse-CDF sampler -- replaces sample_pdf_2 (H:344-387) --------------------------------- */
/* bins (R,n_bins), weights (R,n_bins-1); u: row r at u + r*u_row_stride, n_out values
 * (u_row_stride = n_out for torch.rand draws, 0 to broadcast the det-mode linspace(0,1,n_out) table). */
int nf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_row_stride,
                  int64_t n_rays, int n_bins, int n_out, float* samples, nf_stream_t stream);
/* The CDF table is bit-identical to torch-CPU's (float row sum in ATen's 8-lane / 4-way-ILP order, sequential double
 * cumsum rounded per element), so the searchsorted indices equal the reference's.  The _ex form also returns them:
 * inds (R,n_out) int32 = torch.searchsorted(cdf, u, right=True) (H:368), cdf (R,n_bins); either may be NULL.           */
int nf_sample_pdf_ex(const float* bins, const float* weights, const float* u, int64_t u_row_stride, int64_t n_rays,
                     int n_bins, int n_out, float* samples, int* inds, float* cdf, nf_stream_t stream);

/* ---- K6+K7 fused: hierarchical resampling -- replaces T:116-126 (z_mid, sample_pdf on w[1:-1],
 *      sort(cat(z, z_samples))) ---------------------------------------------------------------------- */
int nf_resample_merge(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_row_stride,
                      int64_t n_rays, int n_coarse, int n_fine, float* z_samples /* (R,n_fine) or NULL */,
                      float* z_fine /* (R,n_coarse+n_fine) */, nf_stream_t stream);

/* ---- whole per-chunk inference pipeline -- replaces predict_and_render_radiance (T:36-162) in eval: coarse depths,
 *      coarse MLP, integrator, resampling, fine MLP, integrator, 7-tuple (T:162), paper model.  One call, caller-provided
 *      workspace (nf_render_rays_workspace_floats), kernels enqueued on `stream`.  packed_bf16_* non-NULL selects the
 *      split-bf16 MLP kernel for that network.  n_fine = 0: coarse only (fine outputs untouched, w_last from the coarse pass).
 *      t_vals: linspace(0,1,n_coarse) table; t_rand / noise_* NULL = off; u as in nf_sample_pdf.                         */
size_t nf_render_rays_workspace_floats(int64_t n_rays, int n_coarse, int n_fine);
int nf_render_rays_fwd(const float* packed_coarse, const void* packed_bf16_coarse, const float* packed_fine,
                       const void* packed_bf16_fine, const float* expr76, const float* latent32, const float* ro,
                       const float* rd, const float* rd_view, const float* bg, const float* t_vals, const float* t_rand,
                       const float* u, int64_t u_row_stride, const float* noise_coarse, const float* noise_fine,
                       int64_t n_rays, int n_coarse, int n_fine, float near_z, float far_z, int white_background,
                       float* workspace, size_t workspace_floats, float* rgb_coarse, float* disp_coarse, float* acc_coarse,
                       float* rgb_fine, float* disp_fine, float* acc_fine, float* w_last, nf_stream_t stream);

/* The same pipeline with the split-fp16 MLP kernels: packed_f16_* = streams of nf_paper_pack_f16 (NULL: that network runs exact f32). */
int nf_render_rays_fwd_f16(const float* packed_coarse, const void* packed_f16_coarse, const float* packed_fine,
                           const void* packed_f16_fine, const float* expr76, const float* latent32, const float* ro,
                           const float* rd, const float* rd_view, const float* bg, const float* t_vals, const float* t_rand,
                           const float* u, int64_t u_row_stride, const float* noise_coarse, const float* noise_fine,
                           int64_t n_rays, int n_coarse, int n_fine, float near_z, float far_z, int white_background,
                           float* workspace, size_t workspace_floats, float* rgb_coarse, float* disp_coarse, float* acc_coarse,
                           float* rgb_fine, float* disp_fine, float* acc_fine, float* w_last, nf_stream_t stream);

/* ... and with two fp16 products per weight ("f16x2", nf_paper_mlp_fwd_f16x2): the same streams, the same arguments. */
int nf_render_rays_fwd_f16x2(const float* packed_coarse, const void* packed_f16_coarse, const float* packed_fine,
                             const void* packed_f16_fine, const float* expr76, const float* latent32, const float* ro,
                             const float* rd, const float* rd_view, const float* bg, const float* t_vals, const float* t_rand,
                             const float* u, int64_t u_row_stride, const float* noise_coarse, const float* noise_fine,
                             int64_t n_rays, int n_coarse, int n_fine, float near_z, float far_z, int white_background,
                             float* workspace, size_t workspace_floats, float* rgb_coarse, float* disp_coarse, float* acc_coarse,
                             float* rgb_fine, float* disp_fine, float* acc_fine, float* w_last, nf_stream_t stream);

/* ---- optimizer step of the trainer -- replaces torch.optim.Adam.step() over [coarse model, fine model, latent codes]
 *      (train_transformed_rays.py:193-199, 391-392) for all tensors in one launch.  Host arrays of n_tensors device pointers
 *      (contiguous f32, numel[i] elements each); step = 1-based count of this update; torch's arithmetic (no weight decay,
 *      no amsgrad): m += (1-b1)(g-m); v = b2 v + (1-b2) g g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).            */
int nf_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int64_t* numel, int n_tensors, float lr, float beta1, float beta2, float eps, int64_t step,
                 nf_stream_t stream);

/* ---- the trainer's loss and its gradients -- replaces TR:355-387 (coarse = mse_loss(rgb_coarse, target); fine = mse_loss(rgb_fine,
 *      target); code = code_weight * torch.norm(latent); loss = coarse + fine + code_scale * code) and the backward of those nodes
 *      (~20 torch launches) by two launches.  n_elems = numel of the colour maps (contiguous f32, same shape as target); rgb_fine and
 *      latent may be NULL.  out7 = {loss, coarse mse, fine mse, code loss, coarse + fine, -10 log10(coarse + fine), ||latent||}.
 *      Backward: grad_out = device scalar d(loss); d_rgb = ((2 / n)(rgb - target)) * grad_out (ATen's mse_loss_backward),
 *      d_latent = latent * (grad_out * code_scale * code_weight / ||latent||), 0 at a zero norm (ATen's norm_backward).            */
int nf_train_loss_fwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n_elems, const float* latent,
                      int n_latent, float code_weight, float code_scale, float* out7, nf_stream_t stream);
int nf_train_loss_bwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n_elems, const float* latent,
                      int n_latent, float code_weight, float code_scale, const float* out7, const float* grad_out,
                      float* d_rgb_coarse, float* d_rgb_fine, float* d_latent, nf_stream_t stream);

/* ---- K7: per-ray ascending sort -- replaces torch.sort(...)[0] at T:126 --------------------------- */
int nf_sort_rows(const float* in, int64_t n_rows, int n_cols, float* out, nf_stream_t stream);

/* ---- host-only self-tests (no device needed): consistency of the weight-gradient job tables -- every slab entry written
 * exactly once per slice, tile ids inside their bundle.  0 = consistent, negative = which check failed.                 */
int nf_selftest_dw_tables_f32(void);
int nf_selftest_dw_tables_lcode_f32(void);
int nf_selftest_dw_tables_bf16(void);
int nf_selftest_dw_tables_tiny(void);
int nf_selftest_dw_tables_flex(int num_layers);
/* gather tables of the four split-bf16 weight streams (host code; out == NULL: number of entries): one code per element of
 * the hi blocks, tensor id << 24 | element offset, 0xFF000000 = zero padding                                            */
long nf_paper_stream_table_bf16(uint32_t* out, size_t n_entries);
long nf_paper_stream_table_bwd_bf16(uint32_t* out, size_t n_entries);
long nf_lcode_stream_table_bf16(uint32_t* out, size_t n_entries);
long nf_lcode_stream_table_bwd_bf16(uint32_t* out, size_t n_entries);

#ifdef __cplusplus
}
#endif
#endif /* NERFACE_HIP_H */
