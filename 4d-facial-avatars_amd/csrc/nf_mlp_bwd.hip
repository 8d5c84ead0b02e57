// K4 backward: gradients of the fused paper MLP w.r.t. all of its parameters and the latent code
// (reference: autograd through nerf/models.py:236-261; what is learnable is listed in SURVEY §8 A12).
//
// Three stages, all on exact-f32 MFMA (v_mfma_f32_16x16x4_f32):
//   B1  k_paper_mlp_bwd_chain   per 32-point wave tile, the forward kernel run in reverse on a transposed
//                               fragment image: dZ_l = (dZ_{l+1} . W_{l+1}) * [X_l > 0]; masks come from the
//                               activations the training forward saved; every dZ_l is written to HBM.
//   B2  k_dw_gemm (nf_mlp_dw.h)  dW_l = dZ_l^T . X_{l-1} as wave-level 128x128 output tiles with the point
//                               dimension (hundreds of thousands) split into slices; bias grads (column sums
//                               of dZ) fall out of the A fragments.  Deterministic: partial slabs per slice.
//   B3  k_paper_grad_reduce /   sum the slabs over slices, then scatter into the 26 reference-layout tensors:
//       k_paper_grad_unpack     PE slot order -> reference columns, folded conditioning columns as outer
//                               products (db (x) [expr/3 | latent]), d latent = W[:,139:171]^T db.
#include <cstdlib>
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"
#include "nf_mlp_stream.h"
#include "nf_mlp_dw.h"
#include "nf_pack.h"


// =================================================================================================
// transposed pack
// =================================================================================================
static const uint32_t NF_ZERO_CODE_T = 0xFF000000u;
static inline uint32_t nf_code_t(int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); }

// block (ni, no), lane (g, i), r  ->  W[row = 16 ni + 4 g + r][col0 + 16 no + i]
static void nf_fill_layer_t(std::vector<uint32_t>& t, int off, int nk, int no_tiles, int tensor, int n_rows, int n_cols, int col0) {
    for (int ni = 0; ni < nk; ++ni)
        for (int no = 0; no < no_tiles; ++no)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int g = lane >> 4, i = lane & 15;
                    const int row = 16 * ni + 4 * g + r, col = col0 + 16 * no + i;
                    t[(size_t)off + ((size_t)(ni * no_tiles + no) * 64 + lane) * 4 + r] =
                        row < n_rows ? nf_code_t(tensor, row, col, n_cols) : NF_ZERO_CODE_T;
                }
}

static void nf_build_gather_table_t(std::vector<uint32_t>& t) {
    using namespace nfl;
    t.assign(PACKED_T_FLOATS, NF_ZERO_CODE_T);
    nf_fill_layer_t(t, OFFT_RGB, 1, 8, 24, 3, 128, 0);            // fc_rgb.weight (3,128)
    nf_fill_layer_t(t, OFFT_D2, 8, 8, 20, 128, 128, 0);           // layers_dir.2
    nf_fill_layer_t(t, OFFT_D1, 8, 8, 18, 128, 128, 0);           // layers_dir.1
    nf_fill_layer_t(t, OFFT_D0, 8, 16, 16, 128, 280, 0);          // layers_dir.0[:, :256]
    nf_fill_layer_t(t, OFFT_D0 + 8 * 16 * FRAG, 1, 16, 14, 1, 256, 0);   // chunk 8: slot 0 = fc_alpha.weight (1,256)
    nf_fill_layer_t(t, OFFT_FEAT, 16, 16, 12, 256, 256, 0);       // fc_feat
    nf_fill_layer_t(t, OFFT_L5, 16, 16, 10, 256, 256, 0);
    nf_fill_layer_t(t, OFFT_L4, 16, 16, 8, 256, 256, 0);
    nf_fill_layer_t(t, OFFT_L3, 16, 16, 6, 256, 427, 171);        // layers_xyz.3[:, 171:427]
    nf_fill_layer_t(t, OFFT_L2, 16, 16, 4, 256, 256, 0);
    nf_fill_layer_t(t, OFFT_L1, 16, 16, 2, 256, 256, 0);
}

static NfPackTable g_paper_table_t;

extern "C" size_t nf_paper_packed_bwd_floats(void) { return (size_t)nfl::PACKED_T_FLOATS; }

extern "C" int nf_paper_pack_bwd(const float* const* params, float* packed_t, nf_stream_t stream) {
    return nf_pack_f32<NF_PAPER_NUM_PARAMS, 1>(g_paper_table_t, nf_build_gather_table_t, params, packed_t, (int)nfl::PACKED_T_FLOATS, stream);
}

// =================================================================================================
// B1: backward chain
// =================================================================================================
// The chain runs on the forward kernels' streamed K loops (nf_mlp_stream.h: nf_seg_lds, nf_tail_dz): C = 0 is the C operand of a layer's
// first MFMAs, the copy of the slab to `dz` rides in the loops (whole lines through a range-checked descriptor), the layer boundary sits
// under the last chunk's MFMAs with the ReLU mask applied on the way to the slab, and a layer's two mask words are fetched when its loop
// starts.  Round 3's block epilogue (k_paper_mlp_bwd_chain_masks: 1.87 ms per 262144 points) against this form: 1.78-1.80 ms, every dZ
// section and every gradient bit-identical (profiles/r04_experiments.md section 8).
template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_paper_mlp_bwd_chain_masks(const float* __restrict__ packed_t, const float* __restrict__ saved, const float* __restrict__ d_raw,
                             int64_t n_points, float* __restrict__ dz) {
    using namespace nfl;
    static_assert(NT == 2, "the copy schedule below is written for 32-point slabs");
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    const int64_t n = n_points;
    const NfW Wi = nf_w_image(packed_t, PACKED_T_FLOATS);
    auto sec = [&](int zs, int width) { return nf_slab_copy(dz, zs, width, p0, n); };
    auto masks = [&](int l, uint2 (&m)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            m[t] = p0 + 16 * t < n ? *nf_mask_ptr(const_cast<float*>(saved), n, l, (p0 >> 4) + t, lane) : make_uint2(0u, 0u);
    };
    f32x4 frag_rgb[NT][1], frag_sig[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int64_t p = p0 + 16 * t + c;
        f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p < n && g == 0) d = reinterpret_cast<const f32x4*>(d_raw)[p];
        frag_rgb[t][0] = (f32x4){d.x, d.y, d.z, 0.f};
        frag_sig[t] = (f32x4){d.w, 0.f, 0.f, 0.f};
    }
    f32x4 acc[NT][16];
    NfStream<NT> st;
    f32x4 bj[NT];
    uint64_t unused64[NT];
    uint2 m[NT];
#pragma unroll
    for (int no = 0; no < 16; ++no) st.bias[no] = (f32x4){0.f, 0.f, 0.f, 0.f};      // C = 0: the C operand of every layer's first MFMAs
    // d(layers_dir.2 out) = d rgb . fc_rgb.weight, masked by layers_dir.2's ReLU: one register chunk, the round-3 form
    masks(8, m);
    nf_zero_acc<NT, 8>(acc);
    nf_mma_from_regs<NT, 8, 1>(acc, reinterpret_cast<const f32x4*>(packed_t) + OFFT_RGB / 4, frag_rgb, lane);
    nf_apply_mask<NT, 8>(acc, m);
    nf_store_act<NT, 8, false>(acc, act4, lane);
    nf_load_w16<8>(st.wa, Wi, OFFT_D2 / 4, lane);
    nf_read_b<NT>(st.b0, act4, lane, 0);
    // one layer from the slab: the slab = dZ section ZSEC_ (W4_ float4 per row) is copied out and consumed; the output is masked by
    // ReLU layer MASKL_ (-1: none) under the last chunk and becomes the next slab
#define NF_CHAIN_LAYER(OFF_, NO_, NCH_, W4_, ZSEC_, MASKL_, OFF_NEXT_, NO_NEXT_)                                       \
    do {                                                                                                             \
        NfCopyH<W4_, 4, false> cs{act4, sec(ZSEC_, 4 * (W4_)), lane, (NCH_) / 2, {}};                                  \
        cs.prime();                                                                                                  \
        if ((MASKL_) >= 0) masks((MASKL_) >= 0 ? (MASKL_) : 0, m);                                                   \
        nf_seg_lds<NT, NO_, true, false, false>(acc, st, Wi, OFF_, NCH_, act4, lane, cs, unused64);                  \
        nf_pending_b<NT, false>(bj, st);                                                                             \
        nf_tail_dz<NT, NO_, NO_NEXT_, ((MASKL_) >= 0)>(acc, st.wb, bj, st, Wi, OFF_NEXT_, act4, lane, m);            \
    } while (0)
    NF_CHAIN_LAYER(OFFT_D2 / 4, 8, 8, 32, Z_D2, 7, OFFT_D1 / 4, 8);
    NF_CHAIN_LAYER(OFFT_D1 / 4, 8, 8, 32, Z_D1, 6, OFFT_D0 / 4, 16);
    // d feat = dZ_D0 . layers_dir.0.weight[:, :256] + d sigma * fc_alpha.weight   (no activation on feat): 8 slab chunks + one register chunk
    {
        f32x4 wd[16];
        nf_load_w16<16>(wd, Wi, OFFT_D0 / 4 + 8 * 16 * 64, lane);          // the d-sigma chunk's weights, a layer ahead
        NfCopyH<32, 4, false> cs{act4, sec(Z_D0, 128), lane, 4, {}};
        cs.prime();
        nf_seg_lds<NT, 16, true, false, false>(acc, st, Wi, OFFT_D0 / 4, 8, act4, lane, cs, unused64);
        nf_pending_b<NT, false>(bj, st);
        nf_chunk<NT, 16, false>(acc, st.wb, bj, st.bias);
        nf_tail_dz<NT, 16, 16, false>(acc, wd, frag_sig, st, Wi, OFFT_FEAT / 4, act4, lane, m);
    }
    NF_CHAIN_LAYER(OFFT_FEAT / 4, 16, 16, 64, Z_FEAT, 5, OFFT_L5 / 4, 16);
    NF_CHAIN_LAYER(OFFT_L5 / 4, 16, 16, 64, Z_L5, 4, OFFT_L4 / 4, 16);
    NF_CHAIN_LAYER(OFFT_L4 / 4, 16, 16, 64, Z_L4, 3, OFFT_L3 / 4, 16);
    NF_CHAIN_LAYER(OFFT_L3 / 4, 16, 16, 64, Z_L3, 2, OFFT_L2 / 4, 16);       // hidden columns of the skip layer only
    NF_CHAIN_LAYER(OFFT_L2 / 4, 16, 16, 64, Z_L2, 1, OFFT_L1 / 4, 16);
    NF_CHAIN_LAYER(OFFT_L1 / 4, 16, 16, 64, Z_L1, 0, 0, 0);
#undef NF_CHAIN_LAYER
    {   // the last section has no K loop behind it
        const NfSlabCopy cp = sec(Z_L0, 256);
#pragma unroll 4
        for (int k = 0; k < 16 * NT; ++k) nf_copy_rows<64>(act4, cp, k, lane);
    }
}

// =================================================================================================
// B2: weight-gradient GEMMs (generic kernel in nf_mlp_dw.h); the paper model's job table
// =================================================================================================
// The 36 products (128 x 128 each) as groups of four that share operand panels (k_dw_gemm_lds, nf_mlp_dw.h)
#define NF_DW_GROUPS 9

static void nf_build_dw_groups(NfDwGroup* gr) {
    using namespace nfl;
    int n = 0;
    const NfDwPanel off{-1, 0, 0, 0, 0};
    auto fresh = [&]() -> NfDwGroup& {
        NfDwGroup& g = gr[n++];
        for (auto& p : g.panel) p = off;
        g.share = 2;
        g.n_slices = g.pts_per_slice = 0;
        return g;
    };
    auto job = [](const NfDwGroup& g, int a, int b, int out_off, int ldo, int cs) {
        return NfDwWaveJob{a, b, g.panel[a].valid, g.panel[b].valid, out_off, ldo, cs};
    };
    // a 256 x 256 layer: panels {dZ lo, dZ hi, X lo, X hi}, waves = the 2 x 2 blocks of the product
    auto layer256 = [&](int zsec, int bsec, int gout, int cs) {
        NfDwGroup& g = fresh();
        for (int h = 0; h < 2; ++h) {
            g.panel[h] = NfDwPanel{0, zsec, 256, 128 * h, 128};
            g.panel[2 + h] = NfDwPanel{2, bsec, 256, 128 * h, 128};
        }
        for (int nb = 0; nb < 2; ++nb)
            for (int kb = 0; kb < 2; ++kb)
                g.wave[2 * nb + kb] = job(g, nb, 2 + kb, gout + 128 * nb * 256 + 128 * kb, 256, (kb == 0 && cs >= 0) ? cs + 128 * nb : -1);
    };
    layer256(Z_L1, S_H0, G_L1, CS_L0 + 256);
    layer256(Z_L2, S_H1, G_L2, CS_L0 + 512);
    layer256(Z_L3, S_H2, G_L3B, -1);
    layer256(Z_L4, S_H3, G_L4, CS_L0 + 1024);
    layer256(Z_L5, S_H4, G_L5, CS_L0 + 1280);
    layer256(Z_FEAT, S_H5, G_FEAT, CS_L0 + 1536);
    {   // the four products against the positional encoding: (dZ_L0 | dZ_L3) x PE
        NfDwGroup& g = fresh();
        for (int h = 0; h < 2; ++h) {
            g.panel[h] = NfDwPanel{0, Z_L0, 256, 128 * h, 128};
            g.panel[2 + h] = NfDwPanel{0, Z_L3, 256, 128 * h, 128};
        }
        g.panel[4] = NfDwPanel{2, S_PE, 64, 0, 64};
        g.share = 1;                                                 // second halves idle: half the MFMAs per point
        for (int nb = 0; nb < 2; ++nb) {
            g.wave[nb] = job(g, nb, 4, G_L0 + 128 * nb * 64, 64, CS_L0 + 128 * nb);
            g.wave[2 + nb] = job(g, 2 + nb, 4, G_L3A + 128 * nb * 64, 64, CS_L0 + 768 + 128 * nb);
        }
    }
    {   // dZ_D0 x (feat | dir slots), dZ_D1 x d0
        NfDwGroup& g = fresh();
        g.panel[0] = NfDwPanel{0, Z_D0, 128, 0, 128};
        g.panel[1] = NfDwPanel{2, S_FEAT, 256, 0, 128};
        g.panel[2] = NfDwPanel{2, S_FEAT, 256, 128, 128};
        g.panel[3] = NfDwPanel{2, S_DIRF, 16, 0, 16};
        g.panel[4] = NfDwPanel{0, Z_D1, 128, 0, 128};
        g.panel[5] = NfDwPanel{2, S_D0, 128, 0, 128};
        g.wave[0] = job(g, 0, 1, G_D0A, 256, CS_D0);
        g.wave[1] = job(g, 0, 2, G_D0A + 128, 256, -1);
        g.wave[2] = job(g, 0, 3, G_D0B, 16, -1);
        g.wave[3] = job(g, 4, 5, G_D1, 128, CS_D0 + 128);
    }
    {   // dZ_D2 x d1, d_raw x (d2 | feat): rows 0..2 = fc_rgb.weight, row 3 (d sigma) = fc_alpha.weight, cs = the 4 output-bias gradients
        NfDwGroup& g = fresh();
        g.panel[0] = NfDwPanel{0, Z_D2, 128, 0, 128};
        g.panel[1] = NfDwPanel{2, S_D1, 128, 0, 128};
        g.panel[2] = NfDwPanel{1, 0, 4, 0, 4};
        g.panel[3] = NfDwPanel{2, S_D2, 128, 0, 128};
        g.panel[4] = NfDwPanel{2, S_FEAT, 256, 0, 128};
        g.panel[5] = NfDwPanel{2, S_FEAT, 256, 128, 128};
        g.wave[0] = job(g, 0, 1, G_D2, 128, CS_D0 + 256);
        g.wave[1] = job(g, 2, 3, G_RGB, 128, CS_RGB);
        g.wave[2] = job(g, 2, 4, G_ALPHA, 256, -1);
        g.wave[3] = job(g, 2, 5, G_ALPHA + 128, 256, -1);
    }
    // n == NF_DW_GROUPS by construction
}

// =================================================================================================
// B3: reduce over slices + scatter to the reference parameter layout
// =================================================================================================
// off[t]: first flat element of tensor t; blk[t]: first workgroup of tensor t (a workgroup handles 256 elements of ONE tensor, so
// the tensor id -- and with it the switch below -- is uniform: no per-element search, no divergence); tensor 26 = d latent
struct NfGradOffsets { int off[NF_PAPER_NUM_PARAMS + 2]; int blk[NF_PAPER_NUM_PARAMS + 2]; };

__device__ __forceinline__ int nf_pe_col_to_slot(int col) { return nfl::pe_col_to_slot(col); }

__global__ void __launch_bounds__(256) k_paper_grad_unpack(const float* __restrict__ sum, const float* __restrict__ packed,
                                                           const float* __restrict__ cond, NfGradOffsets offs,
                                                           float* __restrict__ grads) {
    using namespace nfl;
    const float* cvec = cond + B_CVEC;
    const float* dvec = cond + B_DVEC;
    int t = 0;
    while ((int)blockIdx.x >= offs.blk[t + 1]) ++t;                 // uniform
    const int local = ((int)blockIdx.x - offs.blk[t]) * 256 + (int)threadIdx.x;
    if (t == NF_PAPER_NUM_PARAMS) {
        // d latent_j = sum_n W0[n][139+j] db0[n] + W3[n][139+j] db3[n]: the one workgroup of this "tensor" used to run 32 threads through
        // 256 dependent trips -- the longest path of the launch.  All 256 threads: thread (q, j) sums n = 32 q .. 32 q + 31, the eight
        // partial sums of a j are added in a fixed order (deterministic; the association differs from a single running sum).
        __shared__ float part[8][32];
        const int j = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
        const float* w0 = packed + OFF_WC0 + 76 + j;
        const float* w3 = packed + OFF_WC3 + 76 + j;
        float v = 0.f;
#pragma unroll 8
        for (int n = 32 * q; n < 32 * q + 32; ++n) v += w0[n * NCOND] * sum[CS_L0 + n] + w3[n * NCOND] * sum[CS_L0 + 768 + n];
        part[q][j] = v;
        __syncthreads();
        if (threadIdx.x < 32) {
            float r = part[0][j];
#pragma unroll
            for (int k = 1; k < 8; ++k) r += part[k][j];
            grads[offs.off[t] + j] = r;
        }
        return;
    }
    if (local >= offs.off[t + 1] - offs.off[t]) return;
    const int e = offs.off[t] + local;
    {
        float v = 0.f;

        switch (t) {
            case 0: {  // layers_xyz.0.weight [256][171]
                const int n = local / 171, col = local - 171 * n;
                v = col < 63 ? sum[G_L0 + n * 64 + nf_pe_col_to_slot(col)] : sum[CS_L0 + n] * cvec[col - 63];
            } break;
            case 1: v = sum[CS_L0 + local]; break;
            case 2: v = sum[G_L1 + local]; break;
            case 3: v = sum[CS_L0 + 256 + local]; break;
            case 4: v = sum[G_L2 + local]; break;
            case 5: v = sum[CS_L0 + 512 + local]; break;
            case 6: {  // layers_xyz.3.weight [256][427] = [pe 63 | cond 108 | hidden 256]
                const int n = local / 427, col = local - 427 * n;
                v = col < 63 ? sum[G_L3A + n * 64 + nf_pe_col_to_slot(col)]
                             : (col < 171 ? sum[CS_L0 + 768 + n] * cvec[col - 63] : sum[G_L3B + n * 256 + (col - 171)]);
            } break;
            case 7: v = sum[CS_L0 + 768 + local]; break;
            case 8: v = sum[G_L4 + local]; break;
            case 9: v = sum[CS_L0 + 1024 + local]; break;
            case 10: v = sum[G_L5 + local]; break;
            case 11: v = sum[CS_L0 + 1280 + local]; break;
            case 12: v = sum[G_FEAT + local]; break;
            case 13: v = sum[CS_L0 + 1536 + local]; break;
            case 14: v = sum[G_ALPHA + 3 * 256 + local]; break;   // fc_alpha.weight [1][256] = row 3 (d sigma) of d_raw^T feat
            case 15: v = sum[CS_RGB + 3]; break;               // fc_alpha.bias
            case 16: {  // layers_dir.0.weight [128][280] = [feat 256 | PE4(rd_z, near, far) 24]
                const int n = local / 280, col = local - 280 * n;
                if (col < 256) v = sum[G_D0A + n * 256 + col];
                else {
                    const int q = col - 256, f = q / 6, rem = q - 6 * f, sc = rem / 3, comp = rem - 3 * sc;
                    v = comp == 0 ? sum[G_D0B + n * 16 + 4 * f + sc] : sum[CS_D0 + n] * dvec[4 * f + 2 * sc + (comp - 1)];
                }
            } break;
            case 17: v = sum[CS_D0 + local]; break;
            case 18: v = sum[G_D1 + local]; break;
            case 19: v = sum[CS_D0 + 128 + local]; break;
            case 20: v = sum[G_D2 + local]; break;
            case 21: v = sum[CS_D0 + 256 + local]; break;
            case 22: case 23: v = 0.f; break;                   // layers_dir.3: never used (Quirk Q3)
            case 24: v = sum[G_RGB + local]; break;             // fc_rgb.weight [3][128]
            case 25: v = sum[CS_RGB + local]; break;
        }
        grads[e] = v;
    }
}

static const int NF_PARAM_NUMEL[NF_PAPER_NUM_PARAMS] = {
    256 * 171, 256, 65536, 256, 65536, 256, 256 * 427, 256, 65536, 256, 65536, 256,   // layers_xyz.0..5
    65536, 256, 256, 1,                                                              // fc_feat, fc_alpha
    128 * 280, 128, 16384, 128, 16384, 128, 16384, 128,                              // layers_dir.0..3
    384, 3};                                                                         // fc_rgb

extern "C" size_t nf_paper_grad_floats(void) { return (size_t)nfl::GRAD_FLOATS; }

// defined in nf_mlp_bf16_dw.hip
void nfb_dw_plan(int model, int64_t n_points, int64_t* pts_per_slice, int* n_slices);
int nfb_launch_dw_gemm_bf16(int model, const float* dz, const float* d_raw, const float* saved, int64_t n_points, int64_t pts_per_slice,
                            int n_slices, float* slabs, const float* gscale, nf_stream_t stream);
int nfb_launch_dw_gemm_f16(int model, const float* dz, const float* d_raw, const float* saved, int64_t n_points, int64_t pts_per_slice,
                           int n_slices, float* slabs, const float* gscale, nf_stream_t stream);

extern "C" size_t nf_paper_bwd_workspace_floats(int64_t n_points) {
    int64_t pps; int ns, ns_b;
    nfb_dw_plan(0, n_points, &pps, &ns);
    NfDwGroup groups[NF_DW_GROUPS];
    nf_build_dw_groups(groups);
    ns_b = nf_dw_plan_groups(groups, NF_DW_GROUPS, n_points);
    if (ns_b > ns) ns = ns_b;
    return (size_t)nfl::DZ_PER_POINT * (size_t)n_points + (size_t)(ns + 1) * nfl::SLAB_FLOATS + 16;      // + max |gradient| per dz section (fp16 kernels)
}


// defined in nf_mlp_bf16_bwd.hip / nf_mlp_f16_bwd.hip
int nfb_launch_bwd_chain_bf16(const void* packed_t, const float* saved, const float* d_raw, int64_t n_points, float* dz,
                              float* gscale, nf_stream_t stream);
int nfb_launch_bwd_chain_f16(const void* packed_t, const float* saved, const float* d_raw, int64_t n_points, float* dz,
                             float* gscale, nf_stream_t stream);

// packed_t (exact f32 chain) | packed_t_bf16 (split-bf16 chain) | packed_t_f16 (split-fp16 chain + dW): exactly one non-NULL
static int nf_bwd_impl(const float* packed, const float* packed_t, const void* packed_t_bf16, const void* packed_t_f16, bool split_dw,
                       const float* cond, const float* saved, const float* d_raw, int64_t n_rays, int n_samples, float* workspace,
                       size_t workspace_floats, float* grads, nf_stream_t stream, float* stage_ms = nullptr, const float* saved_f32 = nullptr) {
    // saved_f32: split chain + exact-f32 weight-gradient GEMMs only -- the split forward's activations converted to the exact-f32
    // layout (nf_split_saved_to_f32); the chain reads its bit masks from `saved`, the GEMMs their operands from `saved_f32`
    using namespace nfl;
    if (!packed || (!packed_t && !packed_t_bf16 && !packed_t_f16) || !cond || !saved || !d_raw || !workspace || !grads || n_rays <= 0 ||
        n_samples <= 0)
        return NF_EINVAL;
    if (packed_t_f16) split_dw = true;
    if ((packed_t_bf16 || packed_t_f16) && !split_dw && !saved_f32) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (workspace_floats < nf_paper_bwd_workspace_floats(n_points)) return NF_EINVAL;
    if (((n_points + 31) & ~(int64_t)31) >= ((int64_t)1 << 22)) return NF_EINVAL;   // as the training forward: 32-bit byte offsets into a (32-padded) section
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return NF_EINVAL;
    int64_t pps; int ns;
    NfDwGroupSet gset;
    if (split_dw) nfb_dw_plan(0, n_points, &pps, &ns);
    else {
        nf_build_dw_groups(gset.g);
        for (int k = 0; k <= NF_DW_MAX_GROUPS; ++k) gset.first_block[k] = 0x7fffffff;
        ns = nf_dw_plan_groups(gset.g, NF_DW_GROUPS, n_points, gset.first_block);
        pps = 0;
    }
    float* dz = workspace;
    float* slabs = workspace + (size_t)DZ_PER_POINT * n_points;
    float* sum = slabs + (size_t)ns * SLAB_FLOATS;
    float* gscale = workspace + nf_paper_bwd_workspace_floats(n_points) - 16;
    hipStream_t s = nf_s(stream);
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    NfReduceAlt alt;
    alt.n_slices = 0;
    for (int q = 0; q < NF_REDUCE_ALT_MAX; ++q) alt.lo4[q] = alt.hi4[q] = 0;
    // the slabs are fully written by the GEMM kernel -- except, in the shared-panel plan, by the group that runs fewer slices: the
    // reduction is told which regions end early instead of a 71 MB zero-fill per call
    if (!(!split_dw && nf_dw_reduce_alt(gset.g, NF_DW_GROUPS, ns, &alt))) {
        alt.n_slices = 0;
        e = hipMemsetAsync(slabs, 0, (size_t)ns * SLAB_FLOATS * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};          // stage_ms: dX chain | weight-gradient GEMMs | reduce + unpack
    auto mark = [&](int k) {
        if (stage_ms && hipEventCreate(&ev[k]) == hipSuccess) (void)hipEventRecord(ev[k], s);
    };
    if (packed_t_f16) {
        e = hipMemsetAsync(gscale, 0, 16 * sizeof(float), s);           // max |gradient| per section, filled by the chain
        if (e != hipSuccess) return (int)e;
        mark(0);
        const int rc2 = nfb_launch_bwd_chain_f16(packed_t_f16, saved, d_raw, n_points, dz, gscale, stream);
        if (rc2) return rc2;
    } else if (packed_t_bf16) {
        mark(0);
        const int rc2 = nfb_launch_bwd_chain_bf16(packed_t_bf16, saved, d_raw, n_points, dz, nullptr, stream);
        if (rc2) return rc2;
    } else {
        mark(0);
        hipLaunchKernelGGL((k_paper_mlp_bwd_chain_masks<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, s, packed_t, saved, d_raw,
                           n_points, dz);
    }
    mark(1);
    if (split_dw) {
        const int rc3 = packed_t_f16 ? nfb_launch_dw_gemm_f16(0, dz, d_raw, saved, n_points, pps, ns, slabs, gscale, stream)
                                     : nfb_launch_dw_gemm_bf16(0, dz, d_raw, saved, n_points, pps, ns, slabs, nullptr, stream);
        if (rc3) return rc3;
    } else {
        hipLaunchKernelGGL((k_dw_gemm_lds<0>), dim3(gset.first_block[NF_DW_GROUPS]), dim3(64 * NF_DW_WAVES), 0, s, gset, (int)SLAB_FLOATS, dz, d_raw,
                           saved_f32 ? saved_f32 : saved, n_points, slabs);
    }
    mark(2);
    hipLaunchKernelGGL((k_grad_reduce<0>), dim3(512), dim3(256), 0, s, slabs, ns, (int)SLAB_FLOATS, sum, alt);
    NfGradOffsets offs;
    offs.off[0] = offs.blk[0] = 0;
    for (int i = 0; i <= NF_PAPER_NUM_PARAMS; ++i) {                  // 26 tensors, then the 32 latent-code gradients
        const int numel = i < NF_PAPER_NUM_PARAMS ? NF_PARAM_NUMEL[i] : 32;
        offs.off[i + 1] = offs.off[i] + numel;
        offs.blk[i + 1] = offs.blk[i] + (numel + 255) / 256;
    }
    hipLaunchKernelGGL(k_paper_grad_unpack, dim3(offs.blk[NF_PAPER_NUM_PARAMS + 1]), dim3(256), 0, s, sum, packed, cond, offs, grads);
    if (stage_ms) {
        mark(3);
        e = hipStreamSynchronize(s);
        for (int k = 0; k < 3; ++k) {
            stage_ms[k] = -1.0f;
            if (e == hipSuccess && ev[k] && ev[k + 1]) (void)hipEventElapsedTime(&stage_ms[k], ev[k], ev[k + 1]);
        }
        for (auto& x : ev)
            if (x) (void)hipEventDestroy(x);
        if (e != hipSuccess) return (int)e;
    }
    NF_RETURN_LAUNCH();
}

// Measurement hook (bench.py's per-kernel training roofline): one backward in arithmetic `precision` (0 exact f32, 1 split-bf16,
// 2 split-fp16; packed_t_any = the matching transposed image / stream) with HIP events recorded on `stream` between its stages;
// synchronises the stream and returns stage_ms[3] = {dX chain, weight-gradient GEMMs, slab reduction + unpack} in milliseconds.
extern "C" int nf_paper_mlp_bwd_stage_ms(const float* packed, const void* packed_t_any, int precision, const float* cond, const float* saved,
                                         const float* d_raw, int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats,
                                         float* grads, float* stage_ms, nf_stream_t stream) {
    if (!packed_t_any || !stage_ms || precision < 0 || precision > 2) return NF_EINVAL;
    return nf_bwd_impl(packed, precision == 0 ? (const float*)packed_t_any : nullptr, precision == 1 ? packed_t_any : nullptr,
                       precision == 2 ? packed_t_any : nullptr, precision != 0, cond, saved, d_raw, n_rays, n_samples, workspace,
                       workspace_floats, grads, stream, stage_ms);
}

extern "C" int nf_paper_mlp_bwd(const float* packed, const float* packed_t, const float* cond, const float* saved,
                                const float* d_raw, int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats,
                                float* grads, nf_stream_t stream) {
    if (!packed_t) return NF_EINVAL;
    return nf_bwd_impl(packed, packed_t, nullptr, nullptr, false, cond, saved, d_raw, n_rays, n_samples, workspace, workspace_floats, grads,
                       stream);
}

// Same, with the dX chain (nf_mlp_bf16_bwd.hip) and, unless exact_dw, the weight-gradient GEMMs (nf_mlp_bf16_dw.hip) on the
// split-bf16 kernels.  `saved` must come from nf_paper_mlp_fwd_train_bf16 (it carries the ReLU bit masks the chain reads).
// saved_f32 (exact_dw only, else NULL): `saved` converted by nf_split_saved_to_f32 -- the exact-f32 GEMMs read f32 rows.
extern "C" int nf_paper_mlp_bwd_bf16(const float* packed, const void* packed_t_bf16, const float* cond, const float* saved,
                                     const float* d_raw, int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats,
                                     float* grads, int exact_dw, const float* saved_f32, nf_stream_t stream) {
    if (!packed_t_bf16 || (exact_dw && !saved_f32)) return NF_EINVAL;
    return nf_bwd_impl(packed, nullptr, packed_t_bf16, nullptr, exact_dw == 0, cond, saved, d_raw, n_rays, n_samples, workspace,
                       workspace_floats, grads, stream, nullptr, exact_dw ? saved_f32 : nullptr);
}

// Same on fp16 operand pairs ("f16x3": fp32-class accuracy at the split-bf16 speed): dX chain (nf_mlp_f16_bwd.hip) and weight-
// gradient GEMMs (nf_mlp_f16_dw.hip), gradients in block floating point (a power-of-two scale per point and layer in the chain, per dZ section in the GEMMs).  `saved` must
// come from nf_paper_mlp_fwd_train_f16; packed_t_f16 from nf_paper_pack_bwd_f16.
extern "C" int nf_paper_mlp_bwd_f16(const float* packed, const void* packed_t_f16, const float* cond, const float* saved,
                                    const float* d_raw, int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats,
                                    float* grads, nf_stream_t stream) {
    if (!packed_t_f16) return NF_EINVAL;
    return nf_bwd_impl(packed, nullptr, nullptr, packed_t_f16, true, cond, saved, d_raw, n_rays, n_samples, workspace, workspace_floats,
                       grads, stream);
}

// host-only self-test of the exact-f32 group table (tests/test_host.py)
extern "C" int nf_selftest_dw_tables_f32(void) {
    const long paper = 2L * 256 * 64 + 6L * 65536 + 128L * 272 + 2L * 128 * 128 + 4L * 128 + 4L * 256 + 7 * 256 + 3 * 128 + 4;
    NfDwGroup groups[NF_DW_GROUPS];
    nf_build_dw_groups(groups);
    int rc = nf_check_dw_groups(groups, NF_DW_GROUPS, nfl::SLAB_FLOATS, paper);
    if (rc) return rc;
    // the plan at the training sizes: one workgroup per CU at most, and the reduction can describe the short groups
    for (int64_t n : {(int64_t)131072, (int64_t)262144, (int64_t)259969, (int64_t)512}) {
        int first[NF_DW_MAX_GROUPS + 1];
        const int most = nf_dw_plan_groups(groups, NF_DW_GROUPS, n, first);
        NfReduceAlt alt;
        if (first[NF_DW_GROUPS] > 256 || most < 1 || !nf_dw_reduce_alt(groups, NF_DW_GROUPS, most, &alt)) return -200;
        for (int i = 0; i < NF_DW_GROUPS; ++i)
            if ((int64_t)groups[i].n_slices * groups[i].pts_per_slice < n || (groups[i].pts_per_slice & 15)) return -201;
    }
    return 0;
}
