// Training (activation-saving) instantiation of the split-fp16 forward of the second model family (cf. nf_mlp_f16_train.hip).
#define NFB_F16 1
#ifndef NFB_TILE_GROUP
#define NFB_TILE_GROUP 2          // A fragments of 2 output tiles at a time: with the transposing side job of the saves, 4 spill 15 registers
#endif
#ifndef NFB_ACT_SHIFT
#define NFB_ACT_SHIFT 4
#endif
#include "nf_mlp_lcode_bf16_common.h"
#include "nf_pack.h"

#define NFB_SAVE 1
#define NFB_KERNEL_NAME k_lcode_mlp_fwd_f16_train
#include "nf_mlp_lcode_bf16_kernel.inc"

int nfh_lcode_launch_train(const char* wstream, const float* cond, const float* ro, const float* rd, const float* rd_view, const float* z,
                           int64_t n_points, int n_samples, float* raw, float* saved, unsigned grid, nf_stream_t stream) {
    hipLaunchKernelGGL(k_lcode_mlp_fwd_f16_train, dim3(grid), dim3(256), 0, nf_s(stream), wstream, cond, ro, rd, rd_view, z, n_points,
                       n_samples, raw, saved);
    NF_RETURN_LAUNCH();
}
