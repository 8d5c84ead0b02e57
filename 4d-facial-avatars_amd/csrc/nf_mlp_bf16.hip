// K4, split-bf16 variant: the fused paper-MLP forward on the bf16 matrix pipe at fp32-class accuracy.
//
// CDNA4 has no TF32-like mode: exact-f32 MFMA runs at 1/16 of the bf16 rate.  This kernel evaluates every GEMM
// as THREE bf16 MFMAs with f32 accumulation,
//        x = x_hi + x_lo,  x_hi = bf16(x), x_lo = bf16(x - x_hi)          (16 significand bits kept)
//        W.x ~= W_hi.x_hi + W_hi.x_lo + W_lo.x_hi                          (dropped term ~ 2^-16)
// products of bf16 pairs are exact in f32, so the only deviations from the f32 kernel are the 2^-17 truncation of
// the operands and the dropped lo*lo term: relative error ~2^-16 per layer instead of 2^-24, which keeps the
// end-to-end |dPSNR| two orders of magnitude inside the 1e-4 dB gate (tests/test_gpu_bf16.py) while tripling
// throughput.  Same structure as nf_mlp.hip otherwise (see nf_mlp_layout.h), adapted to
// v_mfma_f32_32x32x16_bf16 (D[32 out-features x 32 points] += W[32 x 16] act[16 x 32]):
//   * lane (h = l>>5, c = l&31): A operand = W[n0 + c][8 k-slots of half h], B operand = act[point c][same slots],
//     D regs r = 0..15 = out-feature n0 + (r&3) + 8(r>>2) + 4h of point c;
//   * K order is chosen so that the 8 slots lane half h feeds to k-step s are exactly the features that half h
//     holds in D regs 8u..8u+7 of tile nt (s = 2 nt + u): a layer's accumulators become the next layer's B
//     operands by a lane-local f32 -> (bf16 hi, bf16 lo) split -- activations never leave registers;
//   * a wave owns 32 points for the whole network; the 4 waves of a workgroup share the weight stream, staged
//     L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) two k-steps at a time into a
//     double buffer, one workgroup barrier per stage; LDS reads are lane-linear ds_read_b128 (conflict-free).
#include <vector>
#include <mutex>

#include "nf_mlp_bf16_common.h"
#include "nf_pack.h"

// =================================================================================================
// pack: fp32 parameters -> (hi, lo) bf16 fragment stream
// =================================================================================================
static const uint32_t NF_ZERO_B = 0xFF000000u;

static void nf_build_table_bf16(std::vector<uint32_t>& t) {
    using namespace nfb;
    t.assign((size_t)N_PAIRS * 512, NF_ZERO_B);
    // per layer: tensor id of the weight, its column count, and the slot -> column rule
    auto code = [](int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); };
    for (int l = 0; l < NL; ++l) {
        for (int s = 0; s < KS[l]; ++s)
            for (int nt = 0; nt < NO[l]; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int h = lane >> 5, i = lane & 31, n = 32 * nt + i;
                        uint32_t c = NF_ZERO_B;
                        switch (l) {
                            case 0: { const int col = pe_col(s, h, j); if (col >= 0) c = code(0, n, col, 171); } break;
                            case 1: c = code(2, n, hid_feature(s, h, j), 256); break;
                            case 2: c = code(4, n, hid_feature(s, h, j), 256); break;
                            case 3: {
                                if (s < 4) { const int col = pe_col(s, h, j); if (col >= 0) c = code(6, n, col, 427); }
                                else c = code(6, n, 171 + hid_feature(s - 4, h, j), 427);
                            } break;
                            case 4: c = code(8, n, hid_feature(s, h, j), 256); break;
                            case 5: c = code(10, n, hid_feature(s, h, j), 256); break;
                            case 6: c = code(12, n, hid_feature(s, h, j), 256); break;
                            case 7: {   // layers_dir.0 (+ fc_alpha as row 128); k-steps 0..15 feat, 16 dir slots, 17 zero
                                if (n < 128) {
                                    if (s < 16) c = code(16, n, hid_feature(s, h, j), 280);
                                    else if (s == 16) { const int col = dir_col(h, j); if (col >= 0) c = code(16, n, col, 280); }
                                } else if (n == 128 && s < 16) c = code(14, 0, hid_feature(s, h, j), 256);
                            } break;
                            case 8: c = code(18, n, hid_feature(s, h, j), 128); break;
                            case 9: c = code(20, n, hid_feature(s, h, j), 128); break;
                            case 10: if (n < 3) c = code(24, n, hid_feature(s, h, j), 128); break;
                        }
                        t[((size_t)(pair_off(l) + s * NO[l] + nt)) * 512 + lane * 8 + j] = c;
                    }
    }
}

void nf_build_table_bf16_shared(std::vector<uint32_t>& t) { nf_build_table_bf16(t); }      // also the split-fp16 stream's table

static NfPackTable g_paper_table_b;

extern "C" size_t nf_paper_packed_bf16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2; }

extern "C" int nf_paper_pack_bf16(const float* const* params, void* stream_out, nf_stream_t stream) {
    return nf_pack_split_bf16<NF_PAPER_NUM_PARAMS, 2>(g_paper_table_b, nf_build_table_bf16, params, stream_out, nfb::N_PAIRS * 512, stream);
}

#define NFB_SAVE 0
#define NFB_KERNEL_NAME k_paper_mlp_fwd_bf16
#include "nf_mlp_bf16_kernel.inc"

// defined in nf_mlp_bf16_train.hip (separate translation unit, see nf_mlp_bf16_kernel.inc)
int nfb_launch_train(const char* wstream, const float* cond, const float* ro, const float* rd, const float* rd_view, const float* z,
                     int64_t n_points, int n_samples, float* raw, float* saved, unsigned grid, nf_stream_t stream);

static int nfb_launch(const void* packed_bf16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                      const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed_bf16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    // the save path addresses a section with 32-bit byte offsets (up to 1 KiB per padded point): larger launches would wrap the
    // buffer descriptor to an empty range and drop every save without an error
    if (saved && nfb_pad32(n_points) >= ((int64_t)1 << 22)) return NF_EINVAL;
    if (saved) return nfb_launch_train(reinterpret_cast<const char*>(packed_bf16), cond, ro, rd, rd_view ? rd_view : rd, z, n_points,
                                       n_samples, raw, saved, (unsigned)grid, stream);
    hipLaunchKernelGGL(k_paper_mlp_fwd_bf16, dim3((unsigned)grid), dim3(256), 0, nf_s(stream),
                       reinterpret_cast<const char*>(packed_bf16), cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw,
                       (float*)nullptr);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_paper_mlp_fwd_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                                     const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                                     nf_stream_t stream) {
    return nfb_launch(packed_bf16, cond, ro, rd, rd_view, z, n_rays, n_samples, raw, nullptr, stream);
}

// Training forward on the split-bf16 kernel: also fills `saved` (nf_paper_saved_floats(n_points) floats, f32, the layout
// nf_paper_mlp_bwd reads) plus the ReLU bit masks nf_paper_mlp_bwd_bf16 reads (S_MASK).
extern "C" int nf_paper_mlp_fwd_train_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                                           const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                                           float* saved, nf_stream_t stream) {
    if (!saved) return NF_EINVAL;
    return nfb_launch(packed_bf16, cond, ro, rd, rd_view, z, n_rays, n_samples, raw, saved, stream);
}

// host-only: the gather table of this stream (one 32-bit code per bf16 element of the hi blocks: tensor id << 24 | element
// offset, 0xFF000000 = zero) for tests/test_host.py; out == NULL returns the number of entries.  Forward stream of the paper model.
extern "C" long nf_paper_stream_table_bf16(uint32_t* out, size_t n_entries) {
    std::vector<uint32_t> t;
    nf_build_table_bf16(t);
    if (!out) return (long)t.size();
    if (n_entries != t.size()) return -1;
    for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
    return (long)t.size();
}
