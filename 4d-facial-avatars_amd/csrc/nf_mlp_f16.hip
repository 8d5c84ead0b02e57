// K4, split-fp16 variant ("f16x3"): the fused paper-MLP forward on the 16-bit matrix pipe at fp32-CLASS accuracy.
//
// Same kernel body, weight-stream layout, LDS ring and register-resident activations as the split-bf16 kernel
// (nf_mlp_bf16.hip / nf_mlp_bf16_kernel.inc), with fp16 operand pairs instead of bf16 pairs:
//        x = x_hi + x_lo,  x_hi = fp16(x), x_lo = fp16(x - x_hi)          (22 significand bits kept; bf16 pairs keep 16)
//        W.x ~= W_hi.x_hi + W_hi.x_lo + W_lo.x_hi                          (dropped term ~ 2^-22), f32 accumulation
// i.e. three v_mfma_f32_32x32x16_f16 per product at the bf16 rate, with a per-product error of a few 2^-23 -- the order of
// the rounding an f32 accumulation commits anyway -- instead of 2^-16.  What bf16 gave for free, f32's exponent range, is
// restored explicitly: every layer's weights are scaled by a power of two chosen at pack time (nf_pack.h), activations
// travel scaled by 2^NFB_ACT_SHIFT, biases are scaled to match when the accumulators are initialised and the scale is
// taken out again (exact power-of-two multiplies) when a layer's outputs become the next layer's operands.
// Valid range: |activations| * 2^NFB_ACT_SHIFT < 65504 (fp16 max; NeRFace activations are O(1..10)); weights: any f32.
#include <vector>
#include <mutex>

#define NFB_F16 1
#ifndef NFB_TILE_GROUP
#define NFB_TILE_GROUP 4          // A fragments of 4 output tiles at a time: 508 VGPRs, no spills (8 at a time: 42 spilled registers)
#endif
#ifndef NFB_ACT_SHIFT
#define NFB_ACT_SHIFT 4
#endif
#include "nf_mlp_bf16_common.h"
#include "nf_pack.h"

// the gather table of the stream is the split-bf16 one (same K order, same blocks): defined in nf_mlp_bf16.hip
void nf_build_table_bf16_shared(std::vector<uint32_t>& t);

static NfPackTable g_paper_table_h;

extern "C" size_t nf_paper_packed_f16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2 + NF_F16_TAIL_BYTES; }

extern "C" int nf_paper_pack_f16(const float* const* params, void* stream_out, nf_stream_t stream) {
    NfLayerPairs<nfb::NL> lp;
    for (int l = 0; l <= nfb::NL; ++l) lp.off[l] = nfb::pair_off(l);
    return nf_pack_split_f16<NF_PAPER_NUM_PARAMS, 7, nfb::NL>(g_paper_table_h, nf_build_table_bf16_shared, params, stream_out,
                                                              nfb::N_PAIRS * 512, lp, (float)(1 << NFB_ACT_SHIFT), stream);
}

#define NFB_SAVE 0
#define NFB_KERNEL_NAME k_paper_mlp_fwd_f16
#include "nf_mlp_bf16_kernel.inc"

extern "C" size_t nf_paper_f16_flag_offset(void) { return (size_t)nfb::STREAM_BF16 * 2 + 4 * NF_F16_FLAG_WORD; }

// defined in nf_mlp_f16_train.hip (separate translation unit)
int nfh_launch_train(const char* wstream, const float* cond, const float* ro, const float* rd, const float* rd_view, const float* z,
                     int64_t n_points, int n_samples, float* raw, float* saved, unsigned grid, nf_stream_t stream);

// Training forward on the split-fp16 kernel: also fills `saved` (TRUE layer outputs, the layout nf_paper_mlp_bwd* read) and the
// ReLU bit masks the split backward chain reads.
extern "C" int nf_paper_mlp_fwd_train_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                          const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_f16 || !cond || !ro || !rd || !z || !raw || !saved || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    if (((n_points + 31) & ~(int64_t)31) >= ((int64_t)1 << 22)) return NF_EINVAL;   // 32-bit byte offsets into a (32-padded) saved section
    return nfh_launch_train(reinterpret_cast<const char*>(packed_f16), cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, saved,
                            (unsigned)grid, stream);
}

extern "C" int nf_paper_mlp_fwd_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                    const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_f16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_paper_mlp_fwd_f16, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), reinterpret_cast<const char*>(packed_f16),
                       cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, (float*)nullptr);
    NF_RETURN_LAUNCH();
}
