// "f16x2" forward of the second model family (ConditionalBlendshapeLearnableCodeNeRFModel, M:529-636): the kernel body of
// nf_mlp_lcode_bf16_kernel.inc on fp16 operands with two products per weight (W_hi.x_hi + W_lo.x_hi) -- see nf_mlp_f16x2.hip.
// Inference only; reads the packed image of nf_lcode_pack_f16.
#include <vector>
#include <mutex>

#define NFB_F16 1
#define NFB_PRODUCTS 5
#ifndef NFB_TILE_GROUP
#define NFB_TILE_GROUP 4
#endif
#ifndef NFB_ACT_SHIFT
#define NFB_ACT_SHIFT 4
#endif
#include "nf_mlp_lcode_bf16_common.h"
#include "nf_pack.h"

#define NFB_SAVE 0
#define NFB_KERNEL_NAME k_lcode_mlp_fwd_f16x2
#include "nf_mlp_lcode_bf16_kernel.inc"

extern "C" int nf_lcode_mlp_fwd_f16x2(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                      const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_f16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_lcode_mlp_fwd_f16x2, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), reinterpret_cast<const char*>(packed_f16),
                       cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, (float*)nullptr);
    NF_RETURN_LAUNCH();
}
