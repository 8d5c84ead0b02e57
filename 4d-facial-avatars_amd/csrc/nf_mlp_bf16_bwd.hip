// B1 on the bf16 matrix pipe: the backward chain dZ_l = (dZ_{l+1} . W_{l+1}) * [X_l > 0] of the paper MLP as split-bf16
// (3 bf16 MFMAs per product, f32 accumulate), the mirror image of nf_mlp_bf16_kernel.inc: gradients w.r.t. activations stay
// in registers from layer to layer, the TRANSPOSED weight stream (hi, lo fragment blocks) is staged L2 -> LDS through the same
// 4-deep ring.  ReLU masks come as bit masks written by the split-bf16 training forward (one 16-byte load per layer, all
// issued before the ring starts, so that no ordinary load sits in the pipelined part); every dZ_l is written to HBM in f32 for
// the weight-gradient GEMMs (split-bf16: nf_mlp_bf16_dw.hip; exact f32: nf_mlp_dw.h).
#include <vector>
#include <mutex>
#include "nf_common.h"
#include "nf_mlp_layout.h"

// NFB_F16 = 1 (nf_mlp_f16_bwd.hip includes this file): the same chain on fp16 operand pairs -- transposed weight stream with
// per-layer power-of-two scales (nf_pack.h), gradients carried times a per-POINT power of two chosen layer by layer from the
// point's largest gradient (block floating point, nf_mlp_bf16_machinery.inc) so that they sit at the top of fp16's exponent range.
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
// backward layers in execution order: reduction k-steps (over the forward layer's OUTPUT features) and 32-row output tiles
//   0 fc_rgb^T (3 -> 128)   1 layers_dir.2^T   2 layers_dir.1^T   3 [layers_dir.0[:, :256] ; fc_alpha]^T (128 + 1 -> 256)
//   4 fc_feat^T   5 layers_xyz.5^T   6 .4^T   7 .3[:, 171:]^T   8 .2^T   9 .1^T
constexpr int NL = 10;
constexpr int KS[NL] = {2, 8, 8, 10, 16, 16, 16, 16, 16, 16};
constexpr int NO[NL] = {4, 4, 4, 8, 8, 8, 8, 8, 8, 8};
constexpr int pair_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += KS[i] * NO[i]; return o; }
constexpr int N_PAIRS = pair_off(NL);
constexpr int STREAM_BF16 = N_PAIRS * 2 * 512;
__host__ __device__ constexpr int hid_feature(int s, int h, int j) { return 16 * s + 4 * h + (j & 3) + 8 * (j >> 2); }
}  // namespace nfb

#include "nf_mlp_bf16_machinery.inc"
#include "nf_pack.h"

// =================================================================================================
// transposed (hi, lo) stream: block (s, nt), lane (h', i), j  ->  W[row = reduction feature(s, h', j)][col0 + 32 nt + i]
// =================================================================================================
static const uint32_t NF_ZERO_BT = 0xFF000000u;

static void nf_build_table_bf16_t(std::vector<uint32_t>& t) {
    using namespace nfb;
    t.assign((size_t)N_PAIRS * 512, NF_ZERO_BT);
    auto code = [](int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); };
    // per layer: (tensor, rows valid, ncols, col0)
    const int tensor[NL] = {24, 20, 18, 16, 12, 10, 8, 6, 4, 2};
    const int nrows[NL] = {3, 128, 128, 128, 256, 256, 256, 256, 256, 256};
    const int ncols[NL] = {128, 128, 128, 280, 256, 256, 256, 427, 256, 256};
    const int col0[NL] = {0, 0, 0, 0, 0, 0, 0, 171, 0, 0};
    for (int l = 0; l < NL; ++l)
        for (int s = 0; s < KS[l]; ++s)
            for (int nt = 0; nt < NO[l]; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int h = lane >> 5, i = lane & 31, col = col0[l] + 32 * nt + i;
                        uint32_t c = NF_ZERO_BT;
                        if (l == 0) {                                   // slots (s = 0, h = 0, j = 0..2) carry d r, d g, d b
                            if (s == 0 && h == 0 && j < 3) c = code(24, j, col, 128);
                        } else if (l == 3) {                            // k-steps 0..7: dZ of layers_dir.0; k-step 8, slot (0, 0): d sigma
                            if (s < 8) c = code(16, hid_feature(s, h, j), col, 280);
                            else if (s == 8 && h == 0 && j == 0) c = code(14, 0, col, 256);
                        } else {
                            const int row = hid_feature(s, h, j);
                            if (row < nrows[l]) c = code(tensor[l], row, col, ncols[l]);
                        }
                        t[((size_t)(pair_off(l) + s * NO[l] + nt)) * 512 + lane * 8 + j] = c;
                    }
}

static NfPackTable g_paper_table_bt;

#if NFB_F16
extern "C" size_t nf_paper_packed_bwd_f16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2 + NF_F16_TAIL_BYTES; }

extern "C" int nf_paper_pack_bwd_f16(const float* const* params, void* stream_out, nf_stream_t stream) {
    NfLayerPairs<nfb::NL> lp;
    for (int l = 0; l <= nfb::NL; ++l) lp.off[l] = nfb::pair_off(l);
    return nf_pack_split_f16<NF_PAPER_NUM_PARAMS, 13, nfb::NL>(g_paper_table_bt, nf_build_table_bf16_t, params, stream_out,
                                                               nfb::N_PAIRS * 512, lp, 1.0f, stream);
}
#else
extern "C" size_t nf_paper_packed_bwd_bf16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2; }

extern "C" int nf_paper_pack_bwd_bf16(const float* const* params, void* stream_out, nf_stream_t stream) {
    return nf_pack_split_bf16<NF_PAPER_NUM_PARAMS, 3>(g_paper_table_bt, nf_build_table_bf16_t, params, stream_out, nfb::N_PAIRS * 512, stream);
}
#endif

// =================================================================================================
// B1 kernel
// =================================================================================================
// acc[nt] reg r keeps its value where bit (16 nt + r) of the lane's 128-bit mask is set
template <int NO>
__device__ __forceinline__ void nfb_apply_mask(f32x16 (&acc)[8], const u32x4& m) {
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
NFB_BWD_NAME(k_paper_mlp_bwd_chain)(const char* __restrict__ wstream, const float* __restrict__ saved, const float* __restrict__ d_raw,
                                    int64_t n_points, float* __restrict__ dz, float* __restrict__ gscale) {
    using namespace nfl;
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
    // The 16-bit-input MFMA accumulates with a small rounding bias toward -inf (about -2^-26 of the addends, measured: profiles/
    // r02_experiments.md section 8) that survives in sums over points where the gradients themselves cancel (bias gradients:
    // sum |dz| / |sum dz| = 300 ... 12000).  Every point therefore runs the chain with its own sign: odd points carry -gradient
    // in the operands (the sign is taken back when dz is written), so the bias alternates over points and cancels like noise.
    const float sgn = (c & 1) ? -1.0f : 1.0f;
#if NFB_F16
    // operands carry gradient * g, g the point's own (signed) power-of-two scale (block floating point, machinery.inc);
    // INV(l) = 1 / s_W(l), so INV(l) / g turns an accumulator into the true pre-activation gradient that is written to `dz`.
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

    // ordinary loads first (d_raw, the nine ReLU bit masks), then the ring
    const f32x4 d = reinterpret_cast<const f32x4*>(d_raw)[p];
    u32x4 mask[9];
    const u32x4* mbase = reinterpret_cast<const u32x4*>(saved + (int64_t)S_MASK * nfb_pad32(n));   // (split training layout: sections are n rounded up to 32 points long)
#pragma unroll
    for (int l = 0; l < 9; ++l) mask[l] = mbase[((int64_t)l * n + p) * 2 + h];
    nfb_issue_w<nfb::stage_nblk(0)>(cx, nfb::stage_blk0(0), 0);
    nfb_issue_w<nfb::stage_nblk(1)>(cx, nfb::stage_blk0(1), NFB_STAGE_BYTES);
    nfb_issue_w<nfb::stage_nblk(2)>(cx, nfb::stage_blk0(2), 2 * NFB_STAGE_BYTES);

    bf16x8 bh[20], bl[20], th[20], tl[20];
#if NFB_F16
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
#define NFB_BWD_RESCALE(L_, acc_, NO_, EXTRA_)                                                           \
    do {                                                                                                 \
        lm[L_] = nfb_pair_max(live ? nfb_lane_absmax<NO_>(acc_) : 0.0f);                                 \
        G = sgn * nfb_pow2_scale(fmaxf(lm[L_], EXTRA_), invG);                                           \
        invG *= sgn;                                                                                     \
    } while (0)
#else
#define NFB_BWD_RESCALE(L_, acc_, NO_, EXTRA_) (void)0
#endif
    // layer epilogue: ReLU mask, true gradients (they stay in acc_ until the NEXT layer's K loop has stored them), next operands
#define NFB_BWD_FINISH(L_, acc_, NO_, MASK_, EXTRA_)                                                     \
    do {                                                                                                 \
        if ((MASK_) >= 0) nfb_apply_mask<NO_>(acc_, mask[(MASK_) >= 0 ? (MASK_) : 0]);                   \
        nfb_scale<NO_>(acc_, INV(L_) * invG);                      /* true gradients for dz */            \
        NFB_BWD_RESCALE(L_, acc_, NO_, EXTRA_);                                                          \
        nfb_to_operands<NO_, false>(acc_, bh, bl, 0, G);                                                 \
    } while (0)
#define NFB_BWD_RUN(L_, acc_, oh_, ol_, NOP_, prev_, PZSEC_)                                             \
    do {                                                                                                 \
        const NfbSaveTarget tg_ = nfb_save_target(dz + (int64_t)(PZSEC_) * n, 32 * (NOP_), p_tile, n, cx.lane); \
        NFB_LAYER_SAVING(L_, acc_, oh_, ol_, NOP_, prev_, tg_);                                          \
    } while (0)
    // mask indices: h0..h5 -> 0..5, layers_dir.0..2 outputs -> 6..8
    nfb_zero_tiles<4>(accA);
    NFB_LAYER(0, accA, th, tl);
    NFB_BWD_FINISH(0, accA, 4, 8, 0.0f);                               // dZ_D2
    nfb_zero_tiles<4>(accB);
    NFB_BWD_RUN(1, accB, bh, bl, 4, accA, Z_D2);
    NFB_BWD_FINISH(1, accB, 4, 7, 0.0f);                               // dZ_D1
    nfb_zero_tiles<4>(accA);
    NFB_BWD_RUN(2, accA, bh, bl, 4, accB, Z_D1);
    NFB_BWD_FINISH(2, accA, 4, 6, fabsf(d.w));                         // dZ_D0
    // d feat = dZ_D0 . layers_dir.0[:, :256] + d sigma * fc_alpha.weight (no activation on feat)
#pragma unroll
    for (int s = 0; s < 8; ++s) { th[s] = bh[s]; tl[s] = bl[s]; }
    {                                                                  // d sigma k-step, at the scale of dZ_D0's operands
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (h == 0 && live) x[0] = d.w * G;
        nfb_split(x, th[8], tl[8]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { th[9][j] = (nfb_elt)0.f; tl[9][j] = (nfb_elt)0.f; }
    nfb_zero_tiles<8>(accB);
    NFB_BWD_RUN(3, accB, th, tl, 4, accA, Z_D0);
    NFB_BWD_FINISH(3, accB, 8, -1, 0.0f);                              // dZ_FEAT
    nfb_zero_tiles<8>(accA);
    NFB_BWD_RUN(4, accA, bh, bl, 8, accB, Z_FEAT);
    NFB_BWD_FINISH(4, accA, 8, 5, 0.0f);                               // dZ_L5
    nfb_zero_tiles<8>(accB);
    NFB_BWD_RUN(5, accB, bh, bl, 8, accA, Z_L5);
    NFB_BWD_FINISH(5, accB, 8, 4, 0.0f);                               // dZ_L4
    nfb_zero_tiles<8>(accA);
    NFB_BWD_RUN(6, accA, bh, bl, 8, accB, Z_L4);
    NFB_BWD_FINISH(6, accA, 8, 3, 0.0f);                               // dZ_L3
    nfb_zero_tiles<8>(accB);
    NFB_BWD_RUN(7, accB, bh, bl, 8, accA, Z_L3);
    NFB_BWD_FINISH(7, accB, 8, 2, 0.0f);                               // dZ_L2
    nfb_zero_tiles<8>(accA);
    NFB_BWD_RUN(8, accA, bh, bl, 8, accB, Z_L2);
    NFB_BWD_FINISH(8, accA, 8, 1, 0.0f);                               // dZ_L1
    nfb_zero_tiles<8>(accB);
    NFB_BWD_RUN(9, accB, bh, bl, 8, accA, Z_L1);
    nfb_apply_mask<8>(accB, mask[0]);
    nfb_scale<8>(accB, INV(9) * invG);
    nfb_save_now<8>(accB, nfb_save_target(dz + (int64_t)Z_L0 * n, 256, p_tile, n, cx.lane), cx.lds + NFB_XPOSE_OFF + cx.wave * NFB_XPOSE_BYTES,
                    cx.lane);                                          // the last dZ has no K loop behind it
    NFB_BWD_RESCALE(9, accB, 8, 0.0f);                                 // only for max |dZ_L0| (the weight-gradient kernel's scale)
#if NFB_F16
    nfb_flush_layer_max<NFB_GS_DRAW + 1>(lm, lmax, seen, cx.lane);
#endif
#undef NFB_BWD_FINISH
#undef NFB_BWD_RUN
#undef NFB_BWD_RESCALE
#undef INV
}

// gscale: 16 zeroed words on the device that receive max |dz| per layer (fp16 instantiation; ignored by the bf16 one)
int NFB_BWD_NAME(nfb_launch_bwd_chain)(const void* packed_t, const float* saved, const float* d_raw, int64_t n_points, float* dz,
                                       float* gscale, nf_stream_t stream) {
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(NFB_BWD_NAME(k_paper_mlp_bwd_chain), dim3((unsigned)grid), dim3(256), 0, nf_s(stream),
                       reinterpret_cast<const char*>(packed_t), saved, d_raw, n_points, dz, gscale);
    NF_RETURN_LAUNCH();
}

#if !NFB_F16
// host-only: the gather table of this stream (one 32-bit code per bf16 element of the hi blocks: tensor id << 24 | element
// offset, 0xFF000000 = zero) for tests/test_host.py; out == NULL returns the number of entries.  Transposed (backward-chain) stream of the paper model.
extern "C" long nf_paper_stream_table_bwd_bf16(uint32_t* out, size_t n_entries) {
    std::vector<uint32_t> t;
    nf_build_table_bf16_t(t);
    if (!out) return (long)t.size();
    if (n_entries != t.size()) return -1;
    for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
    return (long)t.size();
}
#endif
