// Training (activation-saving) instantiation of the split-bf16 fused forward of the second model family; its own translation
// unit so that it cannot perturb the code generation of the inference kernel (nf_mlp_lcode_bf16.hip).
#ifndef NFB_TILE_GROUP
#define NFB_TILE_GROUP 4          // two accumulator sets (deferred saves): A fragments of 4 output tiles at a time, no spills (8: 11 spilled registers)
#endif
#include "nf_mlp_lcode_bf16_common.h"

#define NFB_SAVE 1
#define NFB_KERNEL_NAME k_lcode_mlp_fwd_bf16_train
#include "nf_mlp_lcode_bf16_kernel.inc"

// Training forward on the split-bf16 kernel: also fills `saved` (nf_lcode_saved_floats(n_points) floats: the f32 sections of the
// exact-f32 training forward plus the ReLU bit masks nf_lcode_mlp_bwd_bf16 reads).
extern "C" int nf_lcode_mlp_fwd_train_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                                           const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                                           float* saved, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_bf16 || !cond || !ro || !rd || !z || !raw || !saved || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    if (((n_points + 31) & ~(int64_t)31) >= ((int64_t)1 << 22)) return NF_EINVAL;   // 32-bit byte offsets into a (32-padded) saved section
    hipLaunchKernelGGL(k_lcode_mlp_fwd_bf16_train, dim3((unsigned)grid), dim3(256), 0, nf_s(stream),
                       reinterpret_cast<const char*>(packed_bf16), cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, saved);
    NF_RETURN_LAUNCH();
}
