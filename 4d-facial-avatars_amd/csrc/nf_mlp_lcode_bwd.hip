// Backward of the second model family (ConditionalBlendshapeLearnableCodeNeRFModel, reference nerf/models.py:529-636;
// autograd through M:590-636 as the trainer drives it, TR:355-392): gradients w.r.t. its 16 parameter tensors and the
// latent code, exact f32, same three stages as the paper model (nf_mlp_bwd.hip):
//   B1 k_lcode_mlp_bwd_chain   dZ_dir -> dZ_feat -> dZ_x2 (+ d sigma * fc_alpha.weight: fc_alpha reads x) -> dZ_x1 -> dZ_x0
//                              -> dZ_layer1 (layer1 has no activation), masks from the saved post-ReLU activations
//   B2 k_dw_gemm<1>            dW = dZ^T . X over point slices (24 wave jobs), bias grads as column sums
//   B3 k_grad_reduce<1>, k_lcode_grad_unpack   slabs -> the 16 reference-layout tensors (PE slot order -> columns, folded
//                              expression / latent / (near, far) columns as outer products) + d latent = W1[:,139:171]^T db1
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"
#include "nf_mlp_stream.h"
#include "nf_mlp_lcode_layout.h"
#include "nf_mlp_dw.h"
#include "nf_pack.h"


// =================================================================================================
// transposed pack: block (ni, no), lane (g, i), r -> W[row = 16 ni + 4 g + r][col = 16 no + i]
// =================================================================================================
static void nf_lcode_table_t(std::vector<uint32_t>& t) {
    using namespace nlc;
    const uint32_t Z = 0xFF000000u;
    t.assign(PACKED_T, Z);
    auto fill = [&](int off, int nk, int no_tiles, int tensor, int n_rows, int n_cols) {
        for (int ni = 0; ni < nk; ++ni)
            for (int no = 0; no < no_tiles; ++no)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int g = lane >> 4, i = lane & 15, row = 16 * ni + 4 * g + r, col = 16 * no + i;
                        if (row < n_rows) t[(size_t)off + ((size_t)(ni * no_tiles + no) * 64 + lane) * 4 + r] = ((uint32_t)tensor << 24) | (uint32_t)(row * n_cols + col);
                    }
    };
    fill(OFFT_RGB, 1, 8, 12, 3, 128);                       // fc_rgb.weight (3, 128)
    fill(OFFT_DIR, 8, 16, 8, 128, 280);                     // layers_dir.0.weight[:, :256]
    fill(OFFT_FEAT, 16, 16, 14, 256, 256);                  // fc_feat.weight
    fill(OFFT_FEAT + 16 * 16 * FRAG, 1, 16, 10, 1, 256);    // chunk 16, slot 0: fc_alpha.weight (1, 256)
    fill(OFFT_X2, 16, 16, 6, 256, 256);
    fill(OFFT_X1, 16, 16, 4, 256, 256);
    fill(OFFT_X0, 16, 16, 2, 256, 256);
}

static NfPackTable g_lcode_table_t;

extern "C" size_t nf_lcode_packed_bwd_floats(void) { return (size_t)nlc::PACKED_T; }

extern "C" int nf_lcode_pack_bwd(const float* const* params, float* packed_t, nf_stream_t stream) {
    return nf_pack_f32<nlc::NPARAMS, 5>(g_lcode_table_t, nf_lcode_table_t, params, packed_t, (int)nlc::PACKED_T, stream);
}

// =================================================================================================
// B1: backward chain
// =================================================================================================
template <int NT, int NO>
__device__ __forceinline__ void nf_lc_zero_acc(f32x4 (&acc)[NT][16]) {
#pragma unroll
    for (int no = 0; no < NO; ++no)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][no] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// Layer-streamed like the paper model's chain (nf_mlp_bwd.hip: k_paper_mlp_bwd_chain_masks; nf_mlp_stream.h: nf_seg_lds, nf_tail_dz):
// C = 0 as the C operand of a layer's first MFMAs, the slab copied to `dz` from inside the K loops, the masked layer boundary under the
// last chunk, a layer's two mask words fetched when its loop starts.
template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_lcode_mlp_bwd_chain(const float* __restrict__ packed_t, const float* __restrict__ saved, const float* __restrict__ d_raw,
                      int64_t n_points, float* __restrict__ dz) {
    using namespace nlc;
    static_assert(NT == 2, "the copy schedule below is written for 32-point slabs");
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    const int64_t n = n_points;
    const NfW Wi = nf_w_image(packed_t, PACKED_T);
    auto sec = [&](int zs, int width) { return nf_slab_copy(dz, zs, width, p0, n); };
    auto masks = [&](int l, uint2 (&m)[NT]) {           // layers_xyz.0..2 -> 0..2, fc_feat -> 3, layers_dir.0 -> 4
#pragma unroll
        for (int t = 0; t < NT; ++t)
            m[t] = p0 + 16 * t < n ? *nf_mask_ptr<S_MASK>(const_cast<float*>(saved), n, l, (p0 >> 4) + t, lane) : make_uint2(0u, 0u);
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
    for (int no = 0; no < 16; ++no) st.bias[no] = (f32x4){0.f, 0.f, 0.f, 0.f};      // C = 0
    // d(layers_dir.0 out) = d rgb . fc_rgb.weight, masked by its ReLU: one register chunk, the round-3 form
    masks(4, m);
    nf_lc_zero_acc<NT, 8>(acc);
    nf_mma_from_regs<NT, 8, 1>(acc, reinterpret_cast<const f32x4*>(packed_t) + OFFT_RGB / 4, frag_rgb, lane);
    nf_apply_mask<NT, 8>(acc, m);
    nf_store_act<NT, 8, false>(acc, act4, lane);
    nf_load_w16<16>(st.wa, Wi, OFFT_DIR / 4, lane);
    nf_read_b<NT>(st.b0, act4, lane, 0);
#define NF_LC_CHAIN_LAYER(OFF_, NCH_, W4_, ZSEC_, MASKL_, OFF_NEXT_, NO_NEXT_)                                         \
    do {                                                                                                             \
        NfCopyH<W4_, 4, false> cs{act4, sec(ZSEC_, 4 * (W4_)), lane, (NCH_) / 2, {}};                                  \
        cs.prime();                                                                                                  \
        if ((MASKL_) >= 0) masks((MASKL_) >= 0 ? (MASKL_) : 0, m);                                                   \
        nf_seg_lds<NT, 16, true, false, false>(acc, st, Wi, OFF_, NCH_, act4, lane, cs, unused64);                   \
        nf_pending_b<NT, false>(bj, st);                                                                             \
        nf_tail_dz<NT, 16, NO_NEXT_, ((MASKL_) >= 0)>(acc, st.wb, bj, st, Wi, OFF_NEXT_, act4, lane, m);             \
    } while (0)
    // d feat = dZ_dir . layers_dir.0.weight[:, :256], masked by relu(fc_feat)
    NF_LC_CHAIN_LAYER(OFFT_DIR / 4, 8, 32, Z_DIR, 3, OFFT_FEAT / 4, 16);
    // d x2 = dZ_feat . fc_feat.weight + d sigma * fc_alpha.weight, masked by layers_xyz.2's ReLU: 16 slab chunks + one register chunk
    {
        f32x4 wd[16];
        nf_load_w16<16>(wd, Wi, OFFT_FEAT / 4 + 16 * 16 * 64, lane);
        NfCopyH<64, 4, false> cs{act4, sec(Z_FEAT, 256), lane, 8, {}};
        cs.prime();
        masks(2, m);
        nf_seg_lds<NT, 16, true, false, false>(acc, st, Wi, OFFT_FEAT / 4, 16, act4, lane, cs, unused64);
        nf_pending_b<NT, false>(bj, st);
        nf_chunk<NT, 16, false>(acc, st.wb, bj, st.bias);
        nf_tail_dz<NT, 16, 16, true>(acc, wd, frag_sig, st, Wi, OFFT_X2 / 4, act4, lane, m);
    }
    NF_LC_CHAIN_LAYER(OFFT_X2 / 4, 16, 64, Z_X2, 1, OFFT_X1 / 4, 16);
    NF_LC_CHAIN_LAYER(OFFT_X1 / 4, 16, 64, Z_X1, 0, OFFT_X0 / 4, 16);
    NF_LC_CHAIN_LAYER(OFFT_X0 / 4, 16, 64, Z_X0, -1, 0, 0);              // d(layer1 out): layer1 has no activation (M:609)
#undef NF_LC_CHAIN_LAYER
    {   // the last section has no K loop behind it
        const NfSlabCopy cp = sec(Z_L1, 256);
#pragma unroll 4
        for (int k = 0; k < 16 * NT; ++k) nf_copy_rows<64>(act4, cp, k, lane);
    }
}

// =================================================================================================
// B2 job table
// =================================================================================================
// the 24 products (128 x 128 each) as groups of four that share operand panels (k_dw_gemm_lds, nf_mlp_dw.h)
#define NF_LC_DW_GROUPS 6
static void nf_lcode_build_dw_groups(NfDwGroup* gr) {
    using namespace nlc;
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
    auto layer256 = [&](int zsec, int bsec, int gout, int cs) {
        NfDwGroup& g = fresh();
        for (int h = 0; h < 2; ++h) {
            g.panel[h] = NfDwPanel{0, zsec, 256, 128 * h, 128};
            g.panel[2 + h] = NfDwPanel{2, bsec, 256, 128 * h, 128};
        }
        for (int nb = 0; nb < 2; ++nb)
            for (int kb = 0; kb < 2; ++kb)
                g.wave[2 * nb + kb] = job(g, nb, 2 + kb, gout + 128 * nb * 256 + 128 * kb, 256, kb == 0 ? cs + 128 * nb : -1);
    };
    layer256(Z_X0, S_L1, G_X0, CS_L1 + 256);
    layer256(Z_X1, S_X0, G_X1, CS_L1 + 512);
    layer256(Z_X2, S_X1, G_X2, CS_L1 + 768);
    layer256(Z_FEAT, S_X2, G_FEAT, CS_L1 + 1024);
    {   // dZ_L1 x PE (two row blocks), dZ_dir x (dir slots | feat columns 0..127)
        NfDwGroup& g = fresh();
        g.panel[0] = NfDwPanel{0, Z_L1, 256, 0, 128};
        g.panel[1] = NfDwPanel{0, Z_L1, 256, 128, 128};
        g.panel[2] = NfDwPanel{2, S_PE, 64, 0, 64};
        g.panel[3] = NfDwPanel{0, Z_DIR, 128, 0, 128};
        g.panel[4] = NfDwPanel{2, S_DIRF, 16, 0, 16};
        g.panel[5] = NfDwPanel{2, S_FEAT, 256, 0, 128};
        g.wave[0] = job(g, 0, 2, G_L1, 64, CS_L1);
        g.wave[1] = job(g, 1, 2, G_L1 + 128 * 64, 64, CS_L1 + 128);
        g.wave[2] = job(g, 3, 4, G_DIRB, 16, -1);
        g.wave[3] = job(g, 3, 5, G_DIRA, 256, CS_DIR);
    }
    {   // dZ_dir x feat columns 128..255; d_raw x (dir-layer output | x2): rows 0..2 = fc_rgb.weight, row 3 (d sigma) = fc_alpha.weight
        NfDwGroup& g = fresh();
        g.panel[0] = NfDwPanel{0, Z_DIR, 128, 0, 128};
        g.panel[1] = NfDwPanel{2, S_FEAT, 256, 128, 128};
        g.panel[2] = NfDwPanel{1, 0, 4, 0, 4};
        g.panel[3] = NfDwPanel{2, S_DIR, 128, 0, 128};
        g.panel[4] = NfDwPanel{2, S_X2, 256, 0, 128};
        g.panel[5] = NfDwPanel{2, S_X2, 256, 128, 128};
        g.wave[0] = job(g, 0, 1, G_DIRA + 128, 256, -1);
        g.wave[1] = job(g, 2, 3, G_RGB, 128, CS_RGB);
        g.wave[2] = job(g, 2, 4, G_ALPHA, 256, -1);
        g.wave[3] = job(g, 2, 5, G_ALPHA + 128, 256, -1);
    }
    // n == NF_LC_DW_GROUPS by construction
}

// =================================================================================================
// B3, second half: scatter to the reference parameter layout (order = nerf.models.LCODE_KEYS)
// =================================================================================================
struct NfLcodeGradOffsets { int off[nlc::NPARAMS + 1]; };

__global__ void __launch_bounds__(256) k_lcode_grad_unpack(const float* __restrict__ sum, const float* __restrict__ packed,
                                                           const float* __restrict__ cond, NfLcodeGradOffsets offs,
                                                           float* __restrict__ grads) {
    using namespace nlc;
    const float* cvec = cond + B_CVEC;
    const float* dvec = cond + B_DVEC;
    if (blockIdx.x == gridDim.x - 1) {
        // d latent_j = sum_n layer1.weight[n][139 + j] * d b1[n], the last workgroup's job (one thread per j through 256 dependent trips was
        // the longest path of the launch): thread (q, j) sums n = 32 q .. 32 q + 31, the eight partial sums are added in a fixed order
        __shared__ float part[8][32];
        const int j = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
        const float* w1 = packed + OFF_WC1 + 76 + j;
        float v = 0.f;
#pragma unroll 8
        for (int n = 32 * q; n < 32 * q + 32; ++n) v += w1[n * 108] * sum[CS_L1 + n];
        part[q][j] = v;
        __syncthreads();
        if (threadIdx.x < 32) {
            float r = part[0][j];
#pragma unroll
            for (int k = 1; k < 8; ++k) r += part[k][j];
            grads[GRAD_PARAM_FLOATS + j] = r;
        }
        return;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < GRAD_PARAM_FLOATS; e += (gridDim.x - 1) * blockDim.x) {
        float v = 0.f;
        int t = 0;
        while (e >= offs.off[t + 1]) ++t;
        const int local = e - offs.off[t];
        switch (t) {
            case 0: {  // layer1.weight [256][171] = [pe 63 | expr/3 76 | latent 32]
                const int n = local / 171, col = local - 171 * n;
                v = col < 63 ? sum[G_L1 + n * 64 + nfl::pe_col_to_slot(col)] : sum[CS_L1 + n] * cvec[col - 63];
            } break;
            case 1: v = sum[CS_L1 + local]; break;
            case 2: v = sum[G_X0 + local]; break;
            case 3: v = sum[CS_L1 + 256 + local]; break;
            case 4: v = sum[G_X1 + local]; break;
            case 5: v = sum[CS_L1 + 512 + local]; break;
            case 6: v = sum[G_X2 + local]; break;
            case 7: v = sum[CS_L1 + 768 + local]; break;
            case 8: {  // layers_dir.0.weight [128][280] = [feat 256 | PE4(rd_z, near, far) 24]
                const int n = local / 280, col = local - 280 * n;
                if (col < 256) v = sum[G_DIRA + n * 256 + col];
                else {
                    const int q = col - 256, f = q / 6, rem = q - 6 * f, sc = rem / 3, comp = rem - 3 * sc;
                    v = comp == 0 ? sum[G_DIRB + n * 16 + 4 * f + sc] : sum[CS_DIR + n] * dvec[4 * f + 2 * sc + (comp - 1)];
                }
            } break;
            case 9: v = sum[CS_DIR + local]; break;
            case 10: v = sum[G_ALPHA + 3 * 256 + local]; break;   // fc_alpha.weight [1][256] = row 3 (d sigma) of d_raw^T x
            case 11: v = sum[CS_RGB + 3]; break;
            case 12: v = sum[G_RGB + local]; break;               // fc_rgb.weight [3][128]
            case 13: v = sum[CS_RGB + local]; break;
            case 14: v = sum[G_FEAT + local]; break;
            case 15: v = sum[CS_L1 + 1024 + local]; break;
        }
        grads[e] = v;
    }
}

static const int NF_LC_PARAM_NUMEL[nlc::NPARAMS] = {256 * 171, 256, 65536, 256, 65536, 256, 65536, 256,   // layer1, layers_xyz.0..2
                                                     128 * 280, 128, 256, 1, 384, 3, 65536, 256};          // layers_dir.0, fc_alpha, fc_rgb, fc_feat

extern "C" size_t nf_lcode_grad_floats(void) { return (size_t)nlc::GRAD_FLOATS; }

// defined in nf_mlp_bf16_dw.hip / nf_mlp_lcode_bf16_bwd.hip
void nfb_dw_plan(int model, int64_t n_points, int64_t* pts_per_slice, int* n_slices);
int nfb_launch_dw_gemm_bf16(int model, const float* dz, const float* d_raw, const float* saved, int64_t n_points, int64_t pts_per_slice,
                            int n_slices, float* slabs, const float* gscale, nf_stream_t stream);
int nfb_launch_dw_gemm_f16(int model, const float* dz, const float* d_raw, const float* saved, int64_t n_points, int64_t pts_per_slice,
                           int n_slices, float* slabs, const float* gscale, nf_stream_t stream);
int nfb_lcode_launch_bwd_chain_bf16(const void* packed_t, const float* saved, const float* d_raw, int64_t n_points, float* dz,
                                    float* gscale, nf_stream_t stream);
int nfb_lcode_launch_bwd_chain_f16(const void* packed_t, const float* saved, const float* d_raw, int64_t n_points, float* dz,
                                   float* gscale, nf_stream_t stream);

extern "C" size_t nf_lcode_bwd_workspace_floats(int64_t n_points) {
    int64_t pps; int ns, ns_b;
    nfb_dw_plan(1, n_points, &pps, &ns);
    NfDwGroup groups[NF_LC_DW_GROUPS];
    nf_lcode_build_dw_groups(groups);
    ns_b = nf_dw_plan_groups(groups, NF_LC_DW_GROUPS, n_points);
    if (ns_b > ns) ns = ns_b;
    return (size_t)nlc::DZ_PER_POINT * (size_t)n_points + (size_t)(ns + 1) * nlc::SLAB_FLOATS + 16;      // + max |gradient| per dz section (fp16 kernels)
}


// grads: nf_lcode_grad_floats() floats = the 16 tensors in nerf.models.LCODE_KEYS order, flattened, then d latent (32)
// packed_t (exact f32) | packed_t_bf16 (split-bf16) | packed_t_f16 (split-fp16): exactly one non-NULL
static int nf_lcode_bwd_impl(const float* packed, const float* packed_t, const void* packed_t_bf16, const void* packed_t_f16,
                             const float* cond, const float* saved, const float* d_raw, int64_t n_rays, int n_samples, float* workspace,
                             size_t workspace_floats, float* grads, nf_stream_t stream) {
    using namespace nlc;
    if (!packed || (!packed_t && !packed_t_bf16 && !packed_t_f16) || !cond || !saved || !d_raw || !workspace || !grads || n_rays <= 0 ||
        n_samples <= 0)
        return NF_EINVAL;
    const bool split = packed_t_bf16 != nullptr || packed_t_f16 != nullptr;
    const int64_t n_points = n_rays * n_samples;
    if (workspace_floats < nf_lcode_bwd_workspace_floats(n_points)) return NF_EINVAL;
    if (((n_points + 31) & ~(int64_t)31) >= ((int64_t)1 << 22)) return NF_EINVAL;   // 32-bit byte offsets into a (32-padded) dZ / saved section
    int64_t pps; int ns;
    NfDwGroupSet gset;
    NfReduceAlt alt;
    alt.n_slices = 0;
    for (int q = 0; q < NF_REDUCE_ALT_MAX; ++q) alt.lo4[q] = alt.hi4[q] = 0;
    bool zero_fill = true;
    if (split) nfb_dw_plan(1, n_points, &pps, &ns);
    else {
        nf_lcode_build_dw_groups(gset.g);
        for (int k = 0; k <= NF_DW_MAX_GROUPS; ++k) gset.first_block[k] = 0x7fffffff;
        ns = nf_dw_plan_groups(gset.g, NF_LC_DW_GROUPS, n_points, gset.first_block);
        pps = 0;
        zero_fill = !nf_dw_reduce_alt(gset.g, NF_LC_DW_GROUPS, ns, &alt);   // every group writes every slab: nothing to clear
        if (zero_fill) alt.n_slices = 0;
    }
    float* dz = workspace;
    float* slabs = workspace + (size_t)DZ_PER_POINT * n_points;
    float* sum = slabs + (size_t)ns * SLAB_FLOATS;
    float* gscale = workspace + nf_lcode_bwd_workspace_floats(n_points) - 16;
    hipStream_t s = nf_s(stream);
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipError_t e = hipSuccess;
    if (zero_fill) {
        e = hipMemsetAsync(slabs, 0, (size_t)ns * SLAB_FLOATS * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    if (packed_t_f16) {
        e = hipMemsetAsync(gscale, 0, 16 * sizeof(float), s);           // max |gradient| per section, filled by the chain
        if (e != hipSuccess) return (int)e;
        int rc = nfb_lcode_launch_bwd_chain_f16(packed_t_f16, saved, d_raw, n_points, dz, gscale, stream);
        if (rc) return rc;
        rc = nfb_launch_dw_gemm_f16(1, dz, d_raw, saved, n_points, pps, ns, slabs, gscale, stream);
        if (rc) return rc;
    } else if (split) {
        int rc = nfb_lcode_launch_bwd_chain_bf16(packed_t_bf16, saved, d_raw, n_points, dz, nullptr, stream);
        if (rc) return rc;
        rc = nfb_launch_dw_gemm_bf16(1, dz, d_raw, saved, n_points, pps, ns, slabs, nullptr, stream);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL((k_lcode_mlp_bwd_chain<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, s, packed_t, saved, d_raw, n_points, dz);
        hipLaunchKernelGGL((k_dw_gemm_lds<1>), dim3(gset.first_block[NF_LC_DW_GROUPS]), dim3(64 * NF_DW_WAVES), 0, s, gset, (int)SLAB_FLOATS, dz,
                           d_raw, saved, n_points, slabs);
    }
    hipLaunchKernelGGL((k_grad_reduce<1>), dim3(512), dim3(256), 0, s, slabs, ns, (int)SLAB_FLOATS, sum, alt);
    NfLcodeGradOffsets offs;
    offs.off[0] = 0;
    for (int i = 0; i < NPARAMS; ++i) offs.off[i + 1] = offs.off[i] + NF_LC_PARAM_NUMEL[i];
    hipLaunchKernelGGL(k_lcode_grad_unpack, dim3(1024 + 1), dim3(256), 0, s, sum, packed, cond, offs, grads);      // + 1: the d-latent workgroup
    NF_RETURN_LAUNCH();
}

extern "C" int nf_lcode_mlp_bwd(const float* packed, const float* packed_t, const float* cond, const float* saved, const float* d_raw,
                                int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats, float* grads,
                                nf_stream_t stream) {
    if (!packed_t) return NF_EINVAL;
    return nf_lcode_bwd_impl(packed, packed_t, nullptr, nullptr, cond, saved, d_raw, n_rays, n_samples, workspace, workspace_floats, grads,
                             stream);
}

// Same on the split-bf16 kernels (dX chain nf_mlp_lcode_bf16_bwd.hip, weight-gradient GEMMs nf_mlp_bf16_dw.hip); `saved` must come
// from nf_lcode_mlp_fwd_train_bf16 (it carries the ReLU bit masks the chain reads).
extern "C" int nf_lcode_mlp_bwd_bf16(const float* packed, const void* packed_t_bf16, const float* cond, const float* saved,
                                     const float* d_raw, int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats,
                                     float* grads, nf_stream_t stream) {
    if (!packed_t_bf16) return NF_EINVAL;
    return nf_lcode_bwd_impl(packed, nullptr, packed_t_bf16, nullptr, cond, saved, d_raw, n_rays, n_samples, workspace, workspace_floats,
                             grads, stream);
}

// Same on fp16 operand pairs ("f16x3"): `saved` from nf_lcode_mlp_fwd_train_f16, packed_t_f16 from nf_lcode_pack_bwd_f16.
extern "C" int nf_lcode_mlp_bwd_f16(const float* packed, const void* packed_t_f16, const float* cond, const float* saved,
                                    const float* d_raw, int64_t n_rays, int n_samples, float* workspace, size_t workspace_floats,
                                    float* grads, nf_stream_t stream) {
    if (!packed_t_f16) return NF_EINVAL;
    return nf_lcode_bwd_impl(packed, nullptr, nullptr, packed_t_f16, cond, saved, d_raw, n_rays, n_samples, workspace, workspace_floats,
                             grads, stream);
}

// host-only self-test of this family's exact-f32 group table (tests/test_host.py)
extern "C" int nf_selftest_dw_tables_lcode_f32(void) {
    NfDwGroup groups[NF_LC_DW_GROUPS];
    nf_lcode_build_dw_groups(groups);
    const long lcode = 256L * 64 + 4L * 65536 + 128L * 272 + 4L * 128 + 4L * 256 + 5 * 256 + 128 + 4;
    int rc = nf_check_dw_groups(groups, NF_LC_DW_GROUPS, nlc::SLAB_FLOATS, lcode);
    if (rc) return rc;
    for (int64_t n : {(int64_t)131072, (int64_t)262144, (int64_t)259969, (int64_t)512}) {
        int first[NF_DW_MAX_GROUPS + 1];
        const int most = nf_dw_plan_groups(groups, NF_LC_DW_GROUPS, n, first);
        NfReduceAlt alt;
        if (first[NF_LC_DW_GROUPS] > 256 || most < 1 || !nf_dw_reduce_alt(groups, NF_LC_DW_GROUPS, most, &alt) || alt.n_slices != 0) return -200;
    }
    return 0;
}
