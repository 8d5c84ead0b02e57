// Second model family (ConditionalBlendshapeLearnableCodeNeRFModel, reference nerf/models.py:529-636), inference forward on
// the bf16 matrix pipe at fp32-class accuracy: the split-bf16 scheme and the machinery of nf_mlp_bf16.hip (3 bf16 MFMAs per
// product, K order chosen so that a layer's D registers are the next layer's B operands, weight stream L2 -> LDS through a
// 4-deep LDS-DMA ring) instantiated for this family's layer table:
//   0 layer1 (PE 4 k-steps -> 256, no activation)   1..3 layers_xyz.0..2 (ReLU)   4 fc_alpha (reads x: 256 -> 1)
//   5 fc_feat (ReLU)   6 layers_dir.0 ([feat | dir k-step | pad] -> 128, ReLU)   7 fc_rgb (128 -> 3)
// Same per-call bias table (`cond`, nf_lcode_condition) and packed-weight source tensors as the exact-f32 kernel
// (nf_mlp_lcode.hip).  Training instantiation: nf_mlp_lcode_bf16_train.hip; backward chain: nf_mlp_lcode_bf16_bwd.hip.
#include <vector>
#include <mutex>
#include "nf_mlp_lcode_bf16_common.h"
#include "nf_pack.h"

// =================================================================================================
// pack: fp32 parameters (nerf.models.LCODE_KEYS order) -> (hi, lo) bf16 fragment stream
// =================================================================================================

static void nf_lcode_table_bf16(std::vector<uint32_t>& t) {
    using namespace nfb;
    const uint32_t Z = 0xFF000000u;
    t.assign((size_t)N_PAIRS * 512, Z);
    auto code = [](int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); };
    for (int l = 0; l < NL; ++l)
        for (int s = 0; s < KS[l]; ++s)
            for (int nt = 0; nt < NO[l]; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int h = lane >> 5, i = lane & 31, n = 32 * nt + i;
                        uint32_t c = Z;
                        switch (l) {
                            case 0: { const int col = pe_col(s, h, j); if (col >= 0) c = code(0, n, col, 171); } break;
                            case 1: c = code(2, n, hid_feature(s, h, j), 256); break;
                            case 2: c = code(4, n, hid_feature(s, h, j), 256); break;
                            case 3: c = code(6, n, hid_feature(s, h, j), 256); break;
                            case 4: if (n == 0) c = code(10, 0, hid_feature(s, h, j), 256); break;
                            case 5: c = code(14, n, hid_feature(s, h, j), 256); break;
                            case 6:
                                if (s < 16) c = code(8, n, hid_feature(s, h, j), 280);
                                else if (s == 16) { const int col = dir_col(h, j); if (col >= 0) c = code(8, n, col, 280); }
                                break;
                            case 7: if (n < 3) c = code(12, n, hid_feature(s, h, j), 128); break;
                        }
                        t[((size_t)(pair_off(l) + s * NO[l] + nt)) * 512 + lane * 8 + j] = c;
                    }
}

void nf_lcode_table_bf16_shared(std::vector<uint32_t>& t) { nf_lcode_table_bf16(t); }        // also the split-fp16 stream's table

static NfPackTable g_lcode_table_b;

extern "C" size_t nf_lcode_packed_bf16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2; }

extern "C" int nf_lcode_pack_bf16(const float* const* params, void* stream_out, nf_stream_t stream) {
    return nf_pack_split_bf16<nlc::NPARAMS, 6>(g_lcode_table_b, nf_lcode_table_bf16, params, stream_out, nfb::N_PAIRS * 512, stream);
}

#define NFB_SAVE 0
#define NFB_KERNEL_NAME k_lcode_mlp_fwd_bf16
#include "nf_mlp_lcode_bf16_kernel.inc"

// cond must be the padded table nf_lcode_condition fills (nf_lcode_cond_floats() floats >= 10 KiB)
extern "C" int nf_lcode_mlp_fwd_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                     const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_bf16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_lcode_mlp_fwd_bf16, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), reinterpret_cast<const char*>(packed_bf16),
                       cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, (float*)nullptr);
    NF_RETURN_LAUNCH();
}

// host-only: the gather table of this stream (one 32-bit code per bf16 element of the hi blocks: tensor id << 24 | element
// offset, 0xFF000000 = zero) for tests/test_host.py; out == NULL returns the number of entries.  Forward stream of the second model family.
extern "C" long nf_lcode_stream_table_bf16(uint32_t* out, size_t n_entries) {
    std::vector<uint32_t> t;
    nf_lcode_table_bf16(t);
    if (!out) return (long)t.size();
    if (n_entries != t.size()) return -1;
    for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
    return (long)t.size();
}
