// K4, "f16x2": the split-fp16 forward with TWO products per weight instead of three -- inference only.  It keeps the north_star gate
// (|PSNR(ours, target) - PSNR(reference, target)| <= 1e-4 dB) on whole frames against a uniform-random target; against a target the
// render approximates to 30 dB it misses it on the x1000 density head (4e-3 dB) and keeps it on the x40 head: profiles/r06_gate_sensitivity.md.
//
//        x ~= x_hi = fp16(x)                        (activations enter with 11 significand bits; the `lo` half is never formed)
//        W.x ~= W_hi.x_hi + W_lo.x_hi               (weights keep both halves: 22 bits), f32 accumulation
//
// Same kernel body, weight stream (nf_paper_pack_f16: the packed image of "f16x3" is used as it is), LDS ring, scales and range guard
// as nf_mlp_f16.hip; the W_hi.x_lo MFMAs and the conversions that feed them are not issued: 1988 instead of 2982 MFMAs per 32 points.
// Why THIS term (profiles/r05_split_products.md, nine variants measured on MI355X): rounding an ACTIVATION is an error that differs
// from point to point and averages out over a ray and over the image (whole 512 x 512 frames: |dPSNR| 2e-6 .. 5.6e-5 dB
// over 21 frames -- median 9e-6 on the bench scene's x1000 density head, 5e-6 on the x40 head; self-PSNR 60 .. 115 dB), rounding a WEIGHT is the same error at every point --
// a bias: without W_lo the same frames move by 1e-4 .. 2e-4 dB and fail the gate.
#include <vector>
#include <mutex>

#define NFB_F16 1
#define NFB_PRODUCTS 5            // bit 0: W_lo x_hi, bit 2: W_hi x_hi
#ifndef NFB_TILE_GROUP
#define NFB_TILE_GROUP 4
#endif
#ifndef NFB_ACT_SHIFT
#define NFB_ACT_SHIFT 4
#endif
#include "nf_mlp_bf16_common.h"
#include "nf_pack.h"

#define NFB_SAVE 0
#define NFB_KERNEL_NAME k_paper_mlp_fwd_f16x2
#include "nf_mlp_bf16_kernel.inc"

extern "C" int nf_paper_mlp_fwd_f16x2(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                      const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_f16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_paper_mlp_fwd_f16x2, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), reinterpret_cast<const char*>(packed_f16),
                       cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, (float*)nullptr);
    NF_RETURN_LAUNCH();
}
