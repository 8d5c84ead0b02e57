// Training (activation-saving) instantiation of the split-bf16 fused forward; its own translation unit so that it cannot
// perturb the code generation of the inference kernel (nf_mlp_bf16.hip).
#include "nf_mlp_bf16_common.h"

#define NFB_SAVE 1
#define NFB_KERNEL_NAME k_paper_mlp_fwd_bf16_train
#include "nf_mlp_bf16_kernel.inc"

int nfb_launch_train(const char* wstream, const float* cond, const float* ro, const float* rd, const float* rd_view, const float* z,
                     int64_t n_points, int n_samples, float* raw, float* saved, unsigned grid, nf_stream_t stream) {
    hipLaunchKernelGGL(k_paper_mlp_fwd_bf16_train, dim3(grid), dim3(256), 0, nf_s(stream), wstream, cond, ro, rd, rd_view, z, n_points,
                       n_samples, raw, saved);
    NF_RETURN_LAUNCH();
}
