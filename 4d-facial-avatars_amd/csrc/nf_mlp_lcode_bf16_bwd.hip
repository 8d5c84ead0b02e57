// B1 of the second model family on the bf16 matrix pipe: the backward chain of nf_mlp_lcode_bwd.hip as split-bf16 (3 bf16
// MFMAs per product, f32 accumulate), the mirror image of nf_mlp_lcode_bf16_kernel.inc on a TRANSPOSED (hi, lo) weight stream;
// same structure as nf_mlp_bf16_bwd.hip (paper model).  ReLU gates come from the bit masks the split-bf16 training forward
// wrote (nlc::S_MASK); every dZ is written to HBM in f32 for the weight-gradient GEMMs.
//   0 fc_rgb^T (3 -> 128)   1 layers_dir.0[:, :256]^T (128 -> 256)   2 [fc_feat ; fc_alpha]^T (256 + 1 -> 256)
//   3 layers_xyz.2^T   4 layers_xyz.1^T   5 layers_xyz.0^T (-> dZ of layer1, which has no activation)
#include <vector>
#include <mutex>
#include "nf_common.h"
#include "nf_mlp_lcode_layout.h"

// NFB_F16 = 1 (nf_mlp_lcode_f16_bwd.hip includes this file): the same chain on fp16 operand pairs, see nf_mlp_bf16_bwd.hip
#ifndef NFB_F16
#define NFB_F16 0
#endif
#if NFB_F16
typedef _Float16 nfb_elt;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
#define NFB_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define NFB_BWD_NAME(x) x##_f16
#ifndef NFB_TILE_GROUP
#define NFB_TILE_GROUP 4
#endif
#else
typedef __bf16 nfb_elt;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define NFB_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define NFB_BWD_NAME(x) x##_bf16
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace nfb {
constexpr int NL = 6;
constexpr int KS[NL] = {2, 8, 18, 16, 16, 16};
constexpr int NO[NL] = {4, 8, 8, 8, 8, 8};
constexpr int pair_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += KS[i] * NO[i]; return o; }
constexpr int N_PAIRS = pair_off(NL);
constexpr int STREAM_BF16 = N_PAIRS * 2 * 512;
__host__ __device__ constexpr int hid_feature(int s, int h, int j) { return 16 * s + 4 * h + (j & 3) + 8 * (j >> 2); }
}  // namespace nfb

#include "nf_mlp_bf16_machinery.inc"
#include "nf_pack.h"

// =================================================================================================
// transposed (hi, lo) stream: block (s, nt), lane (h', i), j  ->  W[row = reduction feature(s, h', j)][col = 32 nt + i]
// =================================================================================================

static void nf_lcode_table_bf16_t(std::vector<uint32_t>& t) {
    using namespace nfb;
    const uint32_t Z = 0xFF000000u;
    t.assign((size_t)N_PAIRS * 512, Z);
    auto code = [](int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); };
    for (int l = 0; l < NL; ++l)
        for (int s = 0; s < KS[l]; ++s)
            for (int nt = 0; nt < NO[l]; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int h = lane >> 5, i = lane & 31, col = 32 * nt + i, row = hid_feature(s, h, j);
                        uint32_t c = Z;
                        switch (l) {
                            case 0: if (s == 0 && h == 0 && j < 3) c = code(12, j, col, 128); break;      // slots 0..2 carry d r, d g, d b
                            case 1: if (row < 128) c = code(8, row, col, 280); break;
                            case 2:                                                                      // k-step 16, slot (0, 0): d sigma
                                if (s < 16) c = code(14, row, col, 256);
                                else if (s == 16 && h == 0 && j == 0) c = code(10, 0, col, 256);
                                break;
                            case 3: c = code(6, row, col, 256); break;
                            case 4: c = code(4, row, col, 256); break;
                            case 5: c = code(2, row, col, 256); break;
                        }
                        t[((size_t)(pair_off(l) + s * NO[l] + nt)) * 512 + lane * 8 + j] = c;
                    }
}

static NfPackTable g_lcode_table_bt;

#if NFB_F16
extern "C" size_t nf_lcode_packed_bwd_f16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2 + NF_F16_TAIL_BYTES; }

extern "C" int nf_lcode_pack_bwd_f16(const float* const* params, void* stream_out, nf_stream_t stream) {
    NfLayerPairs<nfb::NL> lp;
    for (int l = 0; l <= nfb::NL; ++l) lp.off[l] = nfb::pair_off(l);
    return nf_pack_split_f16<nlc::NPARAMS, 17, nfb::NL>(g_lcode_table_bt, nf_lcode_table_bf16_t, params, stream_out, nfb::N_PAIRS * 512, lp,
                                                        1.0f, stream);
}
#else
extern "C" size_t nf_lcode_packed_bwd_bf16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2; }

extern "C" int nf_lcode_pack_bwd_bf16(const float* const* params, void* stream_out, nf_stream_t stream) {
    return nf_pack_split_bf16<nlc::NPARAMS, 7>(g_lcode_table_bt, nf_lcode_table_bf16_t, params, stream_out, nfb::N_PAIRS * 512, stream);
}
#endif

// =================================================================================================
// B1 kernel
// =================================================================================================
// acc[nt] reg r keeps its value where bit (16 nt + r) of the lane's 128-bit mask is set
template <int NO>
__device__ __forceinline__ void nfb_lc_apply_mask(f32x16 (&acc)[8], const u32x4& m) {
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned bit = (m[nt >> 1] >> (16 * (nt & 1) + r)) & 1u;
            acc[nt][r] = bit ? acc[nt][r] : 0.0f;
        }
}

template <int NO>
__device__ __forceinline__ void nfb_zero_tiles(f32x16 (&acc)[8]) {
#pragma unroll
    for (int nt = 0; nt < NO; ++nt) nfb_zero(acc[nt]);
}

__global__ void __launch_bounds__(256, 1)
NFB_BWD_NAME(k_lcode_mlp_bwd_chain)(const char* __restrict__ wstream, const float* __restrict__ saved, const float* __restrict__ d_raw,
                                    int64_t n_points, float* __restrict__ dz, float* __restrict__ gscale) {
    using namespace nlc;
    __shared__ __attribute__((aligned(16))) char lds[NFB_LDS_BYTES];
    NfbCtx cx;
    cx.lane = threadIdx.x & 63;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cx.lds = lds;
    nfb_ctx_stream(cx, wstream, (unsigned)nfb::STREAM_BF16 * 2u);
    const int h = cx.lane >> 5, c = cx.lane & 31;
    const int64_t p_tile = ((int64_t)blockIdx.x * 4 + cx.wave) * 32;                 // first point of this wave's tile
    const int64_t p_raw = p_tile + c;
    const int64_t p = p_raw < n_points ? p_raw : n_points - 1;
    const bool live = p_raw < n_points;
    const int64_t n = n_points;
    // every point runs the chain with its own sign, so that the rounding bias of the 16-bit MFMA accumulation alternates over
    // points and cancels in the sums over points (see k_paper_mlp_bwd_chain in nf_mlp_bf16_bwd.hip)
    const float sgn = (c & 1) ? -1.0f : 1.0f;
#if NFB_F16
    const float* __restrict__ wscale = reinterpret_cast<const float*>(wstream + (size_t)nfb::STREAM_BF16 * 2);
    float wsc[nfb::NL];
#pragma unroll
    for (int i = 0; i < nfb::NL; ++i) wsc[i] = wscale[nfb::NL + i];
#pragma unroll
    for (int i = 0; i < nfb::NL; ++i) asm volatile("" : "+s"(wsc[i]));
    unsigned* lmax = reinterpret_cast<unsigned*>(gscale);
    const unsigned seen = cx.lane < 16 ? __atomic_load_n(lmax + cx.lane, __ATOMIC_RELAXED) : 0u;   // possibly stale: only saves atomics
#define INV(L_) wsc[L_]
#else
    const float G = sgn, invG = sgn;
#define INV(L_) 1.0f
#endif

    // ordinary loads first (d_raw, the five ReLU bit masks), then the ring
    const f32x4 d = reinterpret_cast<const f32x4*>(d_raw)[p];
    u32x4 mask[5];
    const u32x4* mbase = reinterpret_cast<const u32x4*>(saved + (int64_t)S_MASK * nfb_pad32(n));   // (split training layout: sections are n rounded up to 32 points long)
#pragma unroll
    for (int l = 0; l < 5; ++l) mask[l] = mbase[((int64_t)l * n + p) * 2 + h];
    nfb_issue_w<nfb::stage_nblk(0)>(cx, nfb::stage_blk0(0), 0);
    nfb_issue_w<nfb::stage_nblk(1)>(cx, nfb::stage_blk0(1), NFB_STAGE_BYTES);
    nfb_issue_w<nfb::stage_nblk(2)>(cx, nfb::stage_blk0(2), 2 * NFB_STAGE_BYTES);

    bf16x8 bh[20], bl[20], th[20], tl[20];
#if NFB_F16
    // per-point gradient scales (block floating point, nf_mlp_bf16_machinery.inc)
    float invG;
    float lm[NFB_GS_DRAW + 1];                                          // this lane's max |gradient| per section (slots: machinery.inc)
#pragma unroll
    for (int i = 0; i <= NFB_GS_DRAW; ++i) lm[i] = 0.0f;
    lm[NFB_GS_DRAW] = live ? fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))) : 0.0f;
    float G = sgn * nfb_pow2_scale(fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)), invG);  // the rgb gradient of this point
    invG *= sgn;
#endif
    {
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (h == 0 && live) { x[0] = d.x * G; x[1] = d.y * G; x[2] = d.z * G; }
        nfb_split(x, th[0], tl[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { th[1][j] = (nfb_elt)0.f; tl[1][j] = (nfb_elt)0.f; }
    }
    nfb_wait_vm<nfb::inflight_after(-1)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // two accumulator sets (deferred saves, nf_mlp_bf16_machinery.inc): dZ of layer l leaves from inside the K loop of layer l+1
    f32x16 accA[8], accB[8];
    // EXTRA_: a gradient that joins the next layer's operands (d sigma), so that the point's scale covers it
#if NFB_F16
#define NFB_LC_BWD_RESCALE(L_, acc_, NO_, EXTRA_)                                                        \
    do {                                                                                                 \
        lm[L_] = nfb_pair_max(live ? nfb_lane_absmax<NO_>(acc_) : 0.0f);                                 \
        G = sgn * nfb_pow2_scale(fmaxf(lm[L_], EXTRA_), invG);                                           \
        invG *= sgn;                                                                                     \
    } while (0)
#else
#define NFB_LC_BWD_RESCALE(L_, acc_, NO_, EXTRA_) (void)0
#endif
#define NFB_LC_BWD_FINISH(L_, acc_, NO_, MASK_, EXTRA_)                                                  \
    do {                                                                                                 \
        if ((MASK_) >= 0) nfb_lc_apply_mask<NO_>(acc_, mask[(MASK_) >= 0 ? (MASK_) : 0]);                \
        nfb_scale<NO_>(acc_, INV(L_) * invG);                      /* true gradients for dz: stored by the next layer's K loop */ \
        NFB_LC_BWD_RESCALE(L_, acc_, NO_, EXTRA_);                                                       \
        nfb_to_operands<NO_, false>(acc_, bh, bl, 0, G);                                                 \
    } while (0)
#define NFB_LC_BWD_RUN(L_, acc_, oh_, ol_, NOP_, prev_, PZSEC_)                                          \
    do {                                                                                                 \
        const NfbSaveTarget tg_ = nfb_save_target(dz + (int64_t)(PZSEC_) * n, 32 * (NOP_), p_tile, n, cx.lane); \
        NFB_LAYER_SAVING(L_, acc_, oh_, ol_, NOP_, prev_, tg_);                                          \
    } while (0)
    // mask indices: layers_xyz.0..2 -> 0..2, fc_feat -> 3, layers_dir.0 -> 4
    nfb_zero_tiles<4>(accA);
    NFB_LAYER(0, accA, th, tl);
    NFB_LC_BWD_FINISH(0, accA, 4, 4, 0.0f);                            // dZ_DIR
    nfb_zero_tiles<8>(accB);
    NFB_LC_BWD_RUN(1, accB, bh, bl, 4, accA, Z_DIR);
    NFB_LC_BWD_FINISH(1, accB, 8, 3, fabsf(d.w));                      // dZ_FEAT
    // d x2 = dZ_feat . fc_feat.weight + d sigma * fc_alpha.weight (fc_alpha reads x), gated by layers_xyz.2's ReLU
#pragma unroll
    for (int s = 0; s < 16; ++s) { th[s] = bh[s]; tl[s] = bl[s]; }
    {                                                                  // d sigma k-step, at the scale of dZ_feat's operands
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (h == 0 && live) x[0] = d.w * G;
        nfb_split(x, th[16], tl[16]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { th[17][j] = (nfb_elt)0.f; tl[17][j] = (nfb_elt)0.f; }
    nfb_zero_tiles<8>(accA);
    NFB_LC_BWD_RUN(2, accA, th, tl, 8, accB, Z_FEAT);
    NFB_LC_BWD_FINISH(2, accA, 8, 2, 0.0f);                            // dZ_X2
    nfb_zero_tiles<8>(accB);
    NFB_LC_BWD_RUN(3, accB, bh, bl, 8, accA, Z_X2);
    NFB_LC_BWD_FINISH(3, accB, 8, 1, 0.0f);                            // dZ_X1
    nfb_zero_tiles<8>(accA);
    NFB_LC_BWD_RUN(4, accA, bh, bl, 8, accB, Z_X1);
    NFB_LC_BWD_FINISH(4, accA, 8, 0, 0.0f);                            // dZ_X0
    nfb_zero_tiles<8>(accB);
    NFB_LC_BWD_RUN(5, accB, bh, bl, 8, accA, Z_X0);                    // layer1 has no activation: dZ = d(out)
    nfb_scale<8>(accB, INV(5) * invG);
    nfb_save_now<8>(accB, nfb_save_target(dz + (int64_t)Z_L1 * n, 256, p_tile, n, cx.lane), cx.lds + NFB_XPOSE_OFF + cx.wave * NFB_XPOSE_BYTES,
                    cx.lane);                                          // the last dZ has no K loop behind it
    NFB_LC_BWD_RESCALE(5, accB, 8, 0.0f);                              // only for max |dZ_L1| (the weight-gradient kernel's scale)
#if NFB_F16
    nfb_flush_layer_max<NFB_GS_DRAW + 1>(lm, lmax, seen, cx.lane);
#endif
#undef NFB_LC_BWD_FINISH
#undef NFB_LC_BWD_RUN
#undef NFB_LC_BWD_RESCALE
#undef INV
}

int NFB_BWD_NAME(nfb_lcode_launch_bwd_chain)(const void* packed_t, const float* saved, const float* d_raw, int64_t n_points, float* dz,
                                             float* gscale, nf_stream_t stream) {
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(NFB_BWD_NAME(k_lcode_mlp_bwd_chain), dim3((unsigned)grid), dim3(256), 0, nf_s(stream),
                       reinterpret_cast<const char*>(packed_t), saved, d_raw, n_points, dz, gscale);
    NF_RETURN_LAUNCH();
}

#if !NFB_F16
// host-only: the gather table of this stream (one 32-bit code per bf16 element of the hi blocks: tensor id << 24 | element
// offset, 0xFF000000 = zero) for tests/test_host.py; out == NULL returns the number of entries.  Transposed stream of the second model family.
extern "C" long nf_lcode_stream_table_bwd_bf16(uint32_t* out, size_t n_entries) {
    std::vector<uint32_t> t;
    nf_lcode_table_bf16_t(t);
    if (!out) return (long)t.size();
    if (n_entries != t.size()) return -1;
    for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
    return (long)t.size();
}
#endif
