// Split-fp16 ("f16x3") forward of the second model family (ConditionalBlendshapeLearnableCodeNeRFModel, M:529-636): the
// kernel body of nf_mlp_lcode_bf16_kernel.inc on fp16 operand pairs with per-layer power-of-two weight scales -- fp32-class
// accuracy at the split-bf16 speed; see nf_mlp_f16.hip for the scheme, the valid range and the range guard.
#include <vector>
#include <mutex>

#define NFB_F16 1
#ifndef NFB_TILE_GROUP
#define NFB_TILE_GROUP 4
#endif
#ifndef NFB_ACT_SHIFT
#define NFB_ACT_SHIFT 4
#endif
#include "nf_mlp_lcode_bf16_common.h"
#include "nf_pack.h"

void nf_lcode_table_bf16_shared(std::vector<uint32_t>& t);          // nf_mlp_lcode_bf16.hip: same K order, same blocks

static NfPackTable g_lcode_table_h;

extern "C" size_t nf_lcode_packed_f16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2 + NF_F16_TAIL_BYTES; }
extern "C" size_t nf_lcode_f16_flag_offset(void) { return (size_t)nfb::STREAM_BF16 * 2 + 4 * NF_F16_FLAG_WORD; }

extern "C" int nf_lcode_pack_f16(const float* const* params, void* stream_out, nf_stream_t stream) {
    NfLayerPairs<nfb::NL> lp;
    for (int l = 0; l <= nfb::NL; ++l) lp.off[l] = nfb::pair_off(l);
    return nf_pack_split_f16<nlc::NPARAMS, 11, nfb::NL>(g_lcode_table_h, nf_lcode_table_bf16_shared, params, stream_out, nfb::N_PAIRS * 512, lp,
                                              (float)(1 << NFB_ACT_SHIFT), stream);
}

#define NFB_SAVE 0
#define NFB_KERNEL_NAME k_lcode_mlp_fwd_f16
#include "nf_mlp_lcode_bf16_kernel.inc"

// defined in nf_mlp_lcode_f16_train.hip (separate translation unit)
int nfh_lcode_launch_train(const char* wstream, const float* cond, const float* ro, const float* rd, const float* rd_view, const float* z,
                           int64_t n_points, int n_samples, float* raw, float* saved, unsigned grid, nf_stream_t stream);

extern "C" int nf_lcode_mlp_fwd_train_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                          const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_f16 || !cond || !ro || !rd || !z || !raw || !saved || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    if (((n_points + 31) & ~(int64_t)31) >= ((int64_t)1 << 22)) return NF_EINVAL;   // 32-bit byte offsets into a (32-padded) saved section
    return nfh_lcode_launch_train(reinterpret_cast<const char*>(packed_f16), cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw,
                                  saved, (unsigned)grid, stream);
}

extern "C" int nf_lcode_mlp_fwd_f16(const void* packed_f16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                    const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_f16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_lcode_mlp_fwd_f16, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), reinterpret_cast<const char*>(packed_f16),
                       cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, (float*)nullptr);
    NF_RETURN_LAUNCH();
}
