// K4: fused forward of ConditionalBlendshapePaperNeRFModel (reference nerf/models.py:189-261) together
// with run_network's input assembly (reference nerf/train_utils.py:9-33,78): pts = ro + rd*z, both
// positional encodings, the [xyz | expr | latent] / skip / [feat | dirs] concatenations and all 11
// live Linear layers -- one launch, nothing but (ro, rd, z) read and 16 B/point written.
//
// Structure (gfx950, exact-f32 MFMA v_mfma_f32_16x16x4_f32; see nf_mlp_layout.h for the fragment maths):
//   * one wavefront owns 16*NT consecutive points for the whole network; waves never communicate, so
//     the kernel has no barrier at all (block = 4 independent waves, one per SIMD, 512-VGPR budget);
//   * the layer output of a wave (NT x 16 tiles x 4 regs) is accumulated in registers, written once to
//     the wave's private LDS slab as 16-byte fragments (XOR-swizzled, conflict-free) and read back as
//     the next layer's B fragments -- one ds_read_b128 per 64 MFMAs;
//   * weights stream from L2 (2.2 MB image, shared by every workgroup) as perfectly coalesced 1-KiB
//     wave loads, one K-chunk ahead of the MFMAs that consume them (register double buffer);
//   * the 63-wide positional encoding lives in registers in B-fragment order (one sin / cos pair per two
//     slots) and is consumed twice (layers_xyz.0 and the skip input of layers_xyz.3);
//   * the per-call constant input columns (expression, latent code, PE(near), PE(far)) never enter
//     the GEMMs: nf_paper_condition folds them into bias vectors (the `cond` table).
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"
#include "nf_mlp_stream.h"
#include "nf_pack.h"


// =================================================================================================
// pack: gather the 26 nn.Parameter storages into the fragment-ordered image
// =================================================================================================

static const uint32_t NF_ZERO_CODE = 0xFF000000u;
static inline uint32_t nf_code(int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); }

// state_dict ids (NF_PAPER_NUM_PARAMS order)
enum { ID_XYZ0_W = 0, ID_XYZ0_B = 1, ID_FEAT_W = 12, ID_FEAT_B = 13, ID_ALPHA_W = 14, ID_ALPHA_B = 15, ID_DIR0_W = 16,
       ID_DIR0_B = 17, ID_RGB_W = 24, ID_RGB_B = 25 };

// Fill one MFMA layer section.  col_of(slot) -> reference column of `tensor` (or -1 = zero);
// rows >= n_out are zero, except that `alpha_row` (if >= 0) is served from fc_alpha.weight.
template <class ColFn>
static void nf_fill_layer(std::vector<uint32_t>& t, int off, int nk, int no_tiles, int tensor, int n_out, int n_cols,
                          ColFn col_of, int alpha_row = -1) {
    for (int ni = 0; ni < nk; ++ni)
        for (int no = 0; no < no_tiles; ++no)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int g = lane >> 4, i = lane & 15;
                    const int n = 16 * no + i, slot = 16 * ni + 4 * g + r;
                    uint32_t code = NF_ZERO_CODE;
                    if (n < n_out) {
                        const int col = col_of(slot);
                        if (col >= 0) code = nf_code(tensor, n, col, n_cols);
                    } else if (n == alpha_row) {
                        if (slot < 256) code = nf_code(ID_ALPHA_W, 0, slot, 256);   // fc_alpha reads feat only
                    }
                    t[(size_t)off + ((size_t)(ni * no_tiles + no) * 64 + lane) * 4 + r] = code;
                }
}

static void nf_build_gather_table(std::vector<uint32_t>& t) {
    using namespace nfl;
    t.assign(PACKED_FLOATS, NF_ZERO_CODE);
    auto ident = [](int slot) { return slot; };
    nf_fill_layer(t, OFF_L0, 4, 16, 0, 256, 171, [](int s) { return pe_slot_to_col(s); });
    nf_fill_layer(t, OFF_L1, 16, 16, 2, 256, 256, ident);
    nf_fill_layer(t, OFF_L2, 16, 16, 4, 256, 256, ident);
    nf_fill_layer(t, OFF_L3, 20, 16, 6, 256, 427, [](int s) { return s < 64 ? pe_slot_to_col(s) : 171 + (s - 64); });
    nf_fill_layer(t, OFF_L4, 16, 16, 8, 256, 256, ident);
    nf_fill_layer(t, OFF_L5, 16, 16, 10, 256, 256, ident);
    nf_fill_layer(t, OFF_FEAT, 16, 16, ID_FEAT_W, 256, 256, ident);
    // layers_dir.0: slots 0..255 = feat; chunk 16: lane group g holds (sin, cos)(rd_z * 2^g) at r = 0, 1
    nf_fill_layer(t, OFF_D0, 17, 9, ID_DIR0_W, 128, 280,
                  [](int s) {
                      if (s < 256) return s;
                      const int g = ((s - 256) >> 2) & 3, r = (s - 256) & 3;
                      return r < 2 ? 256 + 6 * g + 3 * r : -1;
                  },
                  /*alpha_row=*/128);
    nf_fill_layer(t, OFF_D0E, 18, 9, ID_DIR0_W, 128, 280, [](int s) { return s < 280 ? s : -1; }, /*alpha_row=*/128);
    nf_fill_layer(t, OFF_D1, 8, 8, 18, 128, 128, ident);
    nf_fill_layer(t, OFF_D2, 8, 8, 20, 128, 128, ident);
    nf_fill_layer(t, OFF_RGB, 8, 1, ID_RGB_W, 3, 128, ident);
    for (int n = 0; n < 256; ++n)
        for (int k = 0; k < NCOND; ++k) {
            t[OFF_WC0 + n * NCOND + k] = nf_code(0, n, 63 + k, 171);
            t[OFF_WC3 + n * NCOND + k] = nf_code(6, n, 63 + k, 427);
        }
    for (int n = 0; n < 128; ++n)
        for (int f = 0; f < 4; ++f)
            for (int sc = 0; sc < 2; ++sc)
                for (int comp = 1; comp < 3; ++comp)
                    t[OFF_WCD + n * 16 + 4 * f + 2 * sc + (comp - 1)] = nf_code(ID_DIR0_W, n, 256 + 6 * f + 3 * sc + comp, 280);
    const int bias_ids[7] = {1, 3, 5, 7, 9, 11, ID_FEAT_B};
    for (int l = 0; l < 7; ++l)
        for (int n = 0; n < 256; ++n) t[OFF_BIAS + 256 * l + n] = nf_code(bias_ids[l], 0, n, 256);
    for (int n = 0; n < 128; ++n) {
        t[OFF_BIAS + B_D0 + n] = nf_code(ID_DIR0_B, 0, n, 128);
        t[OFF_BIAS + B_D1 + n] = nf_code(19, 0, n, 128);
        t[OFF_BIAS + B_D2 + n] = nf_code(21, 0, n, 128);
    }
    t[OFF_BIAS + B_D0 + 128] = nf_code(ID_ALPHA_B, 0, 0, 1);
    for (int n = 0; n < 3; ++n) t[OFF_BIAS + B_RGB + n] = nf_code(ID_RGB_B, 0, n, 3);
}

static NfPackTable g_paper_table;

extern "C" size_t nf_paper_packed_floats(void) { return (size_t)nfl::PACKED_FLOATS; }
// 2332 floats are written; the buffer is padded to 10 KiB so that the bf16 kernel can DMA it into LDS in whole KiB blocks
extern "C" size_t nf_paper_cond_floats(void) { return 2560; }

// Host-side copy of the gather table (for layout tests without a GPU): code = tensor_id << 24 | offset.
extern "C" int nf_paper_gather_table(uint32_t* out, size_t n) {
    if (!out || n != (size_t)nfl::PACKED_FLOATS) return NF_EINVAL;
    std::vector<uint32_t> host;
    nf_build_gather_table(host);
    for (size_t i = 0; i < n; ++i) out[i] = host[i];
    return 0;
}

extern "C" int nf_paper_pack(const float* const* params, float* packed, nf_stream_t stream) {
    return nf_pack_f32<NF_PAPER_NUM_PARAMS, 0>(g_paper_table, nf_build_gather_table, params, packed, (int)nfl::PACKED_FLOATS, stream);
}

// =================================================================================================
// condition: per-call bias table
//   cond[L0]  = b0  + W0[:, 63:139] (expr*1/3) + W0[:, 139:171] latent          (models.py:239-242)
//   cond[L3]  = b3  + W3[:, 63:171] [expr/3 ; latent]                             (models.py:246)
//   cond[D0]  = bd0 + Wd0[:, 256 + 6f + 3sc + {1,2}] . {sin,cos}({near,far} 2^f)  (Quirk Q1)
//   every other entry is the plain bias.
// =================================================================================================
__global__ void __launch_bounds__(256) k_paper_condition(const float* __restrict__ packed, const float* __restrict__ expr,
                                                         const float* __restrict__ latent, float near_z, float far_z,
                                                         float* __restrict__ cond) {
    using namespace nfl;
    __shared__ float cvec[NCOND];
    __shared__ float dvec[16];
    const int tid = threadIdx.x;
    if (tid < 76) cvec[tid] = nf_div(nf_mul(expr[tid], 1.0f), 3.0f);        // (expr * 1) / 3, a true division
    else if (tid < NCOND) cvec[tid] = latent[tid - 76];
    if (tid >= 128 && tid < 144) {
        const int k = tid - 128, f = k >> 2, sc = (k >> 1) & 1, comp = k & 1;
        const float a = nf_mul(comp ? far_z : near_z, exp2f((float)f));
        dvec[k] = sc ? cosf(a) : sinf(a);
    }
    __syncthreads();
    const float* bias = packed + OFF_BIAS;
    for (int i = blockIdx.x * blockDim.x + tid; i < COND_FLOATS; i += gridDim.x * blockDim.x) {
        if (i >= B_CVEC) { cond[i] = i < B_DVEC ? cvec[i - B_CVEC] : dvec[i - B_DVEC]; continue; }
        float v = bias[i];
        if (i < B_L1 || (i >= B_L3 && i < B_L4)) {
            const int n = i < B_L1 ? i : i - B_L3;
            const float* w = packed + (i < B_L1 ? OFF_WC0 : OFF_WC3) + n * NCOND;
            float s = 0.0f;
            for (int k = 0; k < NCOND; ++k) s = fmaf(w[k], cvec[k], s);
            v += s;
        } else if (i >= B_D0 && i < B_D0 + 128) {
            const float* w = packed + OFF_WCD + (i - B_D0) * 16;
            float s = 0.0f;
            for (int k = 0; k < 16; ++k) s = fmaf(w[k], dvec[k], s);
            v += s;
        }
        cond[i] = v;
    }
}

extern "C" int nf_paper_condition(const float* packed, const float* expr76, const float* latent32, float near_z, float far_z,
                                  float* cond, nf_stream_t stream) {
    if (!packed || !expr76 || !latent32 || !cond) return NF_EINVAL;
    hipLaunchKernelGGL(k_paper_condition, dim3((nfl::COND_FLOATS + 255) / 256), dim3(256), 0, nf_s(stream), packed, expr76,
                       latent32, near_z, far_z, cond);
    NF_RETURN_LAUNCH();
}

// =================================================================================================
// forward
// =================================================================================================
struct NfPointIn {
    float z, ox, oy, oz, dx, dy, dz, dv;
};
template <int NT>
__device__ __forceinline__ void nf_load_point_in(NfPointIn (&in)[NT], int64_t p0, int c, int64_t n_points, int S, const float* __restrict__ ro,
                                                 const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
        const int64_t ray = p / S;
        in[t].z = z[p];
        in[t].dx = rd[ray * 3 + 0]; in[t].dy = rd[ray * 3 + 1]; in[t].dz = rd[ray * 3 + 2];
        in[t].ox = ro[ray * 3 + 0]; in[t].oy = ro[ray * 3 + 1]; in[t].oz = ro[ray * 3 + 2];
        in[t].dv = rd_view[ray * 3 + 2];
    }
}

template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_paper_mlp_fwd(const float* __restrict__ packed, const float* __restrict__ cond_, const float* __restrict__ ro,
                const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z,
                int64_t n_points, int S, float* __restrict__ raw) {
    using namespace nfl;
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    typedef __attribute__((address_space(1))) f32x4 nf_gf32x4;
    nf_gf32x4* raw_v = (nf_gf32x4*)raw;               // the output pointer and the grid stride live in VGPRs (see the epilogue)
    asm volatile("" : "+v"(raw_v));
    int stride_v = (int)gridDim.x;
    asm volatile("" : "+v"(stride_v));
    // persistent form: one workgroup per CU walks the point blocks with the grid's stride (no workgroup dispatch between blocks)
#pragma unroll 1
    for (int64_t blk = blockIdx.x;; blk += __builtin_amdgcn_readfirstlane(stride_v)) {
    const int64_t p0 = (blk * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) break;                        // wave-uniform; no barriers anywhere below
    int opaque0 = 0;                                  // per-block opaque zero: the weight / bias loads must stay where the layers issue them
    asm volatile("" : "+s"(opaque0));
    const f32x4* W = reinterpret_cast<const f32x4*>(packed) + opaque0;
    const float* cond = cond_ + opaque0;

    // ---- inputs: pts = ro + rd*z (T:78), PE fragments, dir fragment -----------------------------
    f32x4 pe[NT][4];
    f32x4 dirf[NT][1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
        const int64_t ray = p / S;
        const float zz = z[p];
        const float dx = rd[ray * 3 + 0], dy = rd[ray * 3 + 1], dz = rd[ray * 3 + 2];
        const float px = nf_add(ro[ray * 3 + 0], nf_mul(dx, zz));
        const float py = nf_add(ro[ray * 3 + 1], nf_mul(dy, zz));
        const float pz = nf_add(ro[ray * 3 + 2], nf_mul(dz, zz));
        nf_encode_point(px, py, pz, g, pe[t]);
        float s, cs;
        nf_sincos(nf_mul(rd_view[ray * 3 + 2], (float)(1 << g)), &s, &cs);   // Quirk Q1: "direction" = (rd_z, near, far)
        dirf[t][0] = (f32x4){s, cs, 0.0f, 0.0f};
    }

    f32x4 acc[NT][16];
    // Layer-streamed form (nf_mlp_stream.h): every layer ends in nf_tail, which writes the raw accumulators to the slab tile by
    // tile under the last chunk's MFMAs and fetches the next layer's bias, first weights and first B fragment; every layer starts
    // with the bias as the C operand of its first MFMAs; the ReLU is applied where the slab is read.
    NfStream<NT> st;
    const NfW Wi = nf_w_image(reinterpret_cast<const float*>(W), PACKED_FLOATS), Ci = nf_w_image(cond, COND_FLOATS);
    f32x4 bj[NT];
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
    // ---- layers_xyz.0 : PE(64 slots) -> 256 ------------------------------------------------------------
    nf_load_bias<16>(st.bias, Ci, B_L0, lane);
    {
        f32x4 w[16];
        nf_load_w16<16>(w, Wi, OFF_L0 / 4, lane);
        NF_PE_B(0); nf_chunk<NT, 16, true>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L0 / 4 + 1 * 16 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L0 / 4 + 2 * 16 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L0 / 4 + 3 * 16 * 64, lane);
        NF_PE_B(3); nf_tail<NT, 16, 16, 16, 1, NF_ONEWAIT>(acc, w, bj, st, Wi, OFF_L1 / 4, Ci, B_L1, act4, lane);
    }
    // ---- layers_xyz.1, .2 (ReLU of the previous layer on read) ------------------------------------------
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_L1 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 16, 1, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_L2 / 4, Ci, B_L2, act4, lane);
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_L2 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 16, 0, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_L3 / 4, Ci, B_L3, act4, lane);
    // ---- layers_xyz.3 : [PE | h] -> 256 (skip connection, M:246) ----------------------------------------
    NF_PE_B(0); nf_chunk<NT, 16, true>(acc, st.wa, bj, st.bias);
    nf_load_w16<16>(st.wa, Wi, OFF_L3 / 4 + 4 * 16 * 64, lane);            // the first slab chunk, three chunks ahead
    nf_read_b<NT>(st.b0, act4, lane, 0);
    {
        f32x4 w[16];
        nf_load_w16<16>(w, Wi, OFF_L3 / 4 + 1 * 16 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L3 / 4 + 2 * 16 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L3 / 4 + 3 * 16 * 64, lane);
        NF_PE_B(3); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
    }
    nf_seg_lds<NT, 16, false, true>(acc, st, Wi, OFF_L3 / 4 + 4 * 16 * 64, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 16, 1, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_L4 / 4, Ci, B_L4, act4, lane);
    // ---- layers_xyz.4, .5 --------------------------------------------------------------------------------
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_L4 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 16, 1, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_L5 / 4, Ci, B_L5, act4, lane);
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_L5 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 16, 1, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_FEAT / 4, Ci, B_FEAT, act4, lane);
    // ---- fc_feat (no activation, M:250: layers_dir.0 reads it as stored) -----------------------------------
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_FEAT / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 9, 1, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_D0 / 4, Ci, B_D0, act4, lane);
    // ---- layers_dir.0 : [feat | dir slots] -> 128; tile 8 row 0 = fc_alpha(feat) (Q2) -----------------------
    float sigma_raw[NT];
    {
        f32x4 wd[16];
        nf_load_w16<9>(wd, Wi, OFF_D0 / 4 + 16 * 9 * 64, lane);             // the dir-slot chunk's weights, a layer ahead
        nf_seg_lds<NT, 9, true, false>(acc, st, Wi, OFF_D0 / 4, 16, act4, lane);
        nf_pending_b<NT, false>(bj, st);
        nf_chunk<NT, 9, false>(acc, st.wb, bj, st.bias);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][0];
        nf_tail<NT, 9, 8, 8, 1, NF_ONEWAIT>(acc, wd, bj, st, Wi, OFF_D1 / 4, Ci, B_D1, act4, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) sigma_raw[t] = acc[t][8].x;
    }
    // ---- layers_dir.1, .2 -----------------------------------------------------------------------------------
    nf_seg_lds<NT, 8, true, true>(acc, st, Wi, OFF_D1 / 4, 8, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 8, 8, 8, 1, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_D2 / 4, Ci, B_D2, act4, lane);
    nf_seg_lds<NT, 8, true, true>(acc, st, Wi, OFF_D2 / 4, 8, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 8, 8, 1, 1, NF_ONEWAIT>(acc, st.wb, bj, st, Wi, OFF_RGB / 4, Ci, B_RGB, act4, lane);
#undef NF_PE_B
    // ---- fc_rgb -------------------------------------------------------------------------------------------
    nf_seg_lds<NT, 1, true, true>(acc, st, Wi, OFF_RGB / 4, 8, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_chunk<NT, 1, false>(acc, st.wb, bj, st.bias);
    // (the store predicate, the output pointer and the grid stride are re-derived from opaque VGPR copies: as loop invariants of the
    // persistent block loop they were the last scalar values the allocator could not keep -- 106 SGPRs are in use -- and went through
    // v_writelane / v_readlane)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    if ((lane_o >> 4) == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t p = p0 + 16 * t + c;
            if (p < n_points)
                raw_v[p] = (f32x4){acc[t][0].x, acc[t][0].y, acc[t][0].z, sigma_raw[t]};
        }
    }
    }
}

// Training forward (exact f32): the same arithmetic as k_paper_mlp_fwd, plus everything the backward needs in `saved`
// (layout nfl::S_*): every layer output as row-major [n][width] matrices for the weight-gradient GEMMs -- copied out of
// the wave's LDS slab as whole 128-byte lines from inside the NEXT layer's K loop (nf_mma_from_lds_copy) -- and the ReLU
// bit masks the dX chain applies (nf_relu_with_mask; 8 bytes per lane, tile and layer).
template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_paper_mlp_fwd_save(const float* __restrict__ packed, const float* __restrict__ cond, const float* __restrict__ ro,
                     const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z,
                     int64_t n_points, int S, float* __restrict__ raw, float* __restrict__ saved) {
    using namespace nfl;
    static_assert(NT == 2, "the copy schedule below is written for 32-point slabs");
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;                       // wave-uniform; no barriers anywhere below
    f32x4* act4 = lds + wave * (16 * NT * 64);
    const f32x4* W = reinterpret_cast<const f32x4*>(packed);
    const int64_t n = n_points;
    auto sec = [&](int s, int width) { return nf_slab_copy(saved, s, width, p0, n); };

    // ---- inputs: pts = ro + rd*z (T:78), PE fragments, dir fragment -----------------------------
    f32x4 pe[NT][4];
    f32x4 dirf[NT][1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
        const int64_t ray = p / S;
        const float zz = z[p];
        const float dx = rd[ray * 3 + 0], dy = rd[ray * 3 + 1], dz = rd[ray * 3 + 2];
        const float px = nf_add(ro[ray * 3 + 0], nf_mul(dx, zz));
        const float py = nf_add(ro[ray * 3 + 1], nf_mul(dy, zz));
        const float pz = nf_add(ro[ray * 3 + 2], nf_mul(dz, zz));
        nf_encode_point(px, py, pz, g, pe[t]);
        float s, cs;
        nf_sincos(nf_mul(rd_view[ray * 3 + 2], (float)(1 << g)), &s, &cs);   // Quirk Q1: "direction" = (rd_z, near, far)
        dirf[t][0] = (f32x4){s, cs, 0.0f, 0.0f};
        // dir slots: 64 B per point, the 16 points of a tile are one contiguous KiB
        if (p0 + 16 * t + c < n_points) *reinterpret_cast<f32x4*>(saved + S_DIRF * n_points + p * 16 + 4 * g) = dirf[t][0];
#pragma unroll
        for (int j = 0; j < 4; ++j) act4[nf_act_idx4(16 * t + c, 4 * j + g)] = pe[t][j];       // PE slots 16 j + 4 g .. + 3
    }
    {   // PE rows (64 slots = 256 B per point): four rows per instruction, before layers_xyz.0's output takes the slab
        const NfSlabCopy cp = sec(S_PE, 64);
#pragma unroll
        for (int k = 0; k < 16 * NT / 4; ++k) nf_copy_rows<16>(act4, cp, k, lane);
    }

    // Layer-streamed like the inference kernel (nf_mlp_stream.h): bias as the C operand, the layer boundary under the last chunk's
    // MFMAs (raw accumulators to the slab), the next layer prefetched.  Everything the backward needs is produced where the slab is
    // READ, inside the K loops: the loop that consumes a layer's output applies the ReLU to its B fragments, collects their [x > 0]
    // bits (nf_mask_bits -- a lane reads back exactly the elements it wrote) and carries the copy of the slab to `saved`, ReLU
    // applied on the way out (NfCopyH: four whole-line stores per step of two chunks).
    f32x4 acc[NT][16];
    uint64_t m64[NT];
    NfStream<NT> st;
    f32x4 bj[NT];
    const NfW Wi = nf_w_image(packed, PACKED_FLOATS), Ci = nf_w_image(cond, COND_FLOATS);
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
    // the last chunk's fragment (+ its mask bits), then the finished mask words of layer MASKL_ (the layer whose output was just consumed)
#define NF_PENDING(RELU_, MASKL_, NCH_)                                                              \
    do {                                                                                            \
        nf_pending_b<NT, RELU_>(bj, st);                                                            \
        if ((MASKL_) >= 0) {                                                                        \
            nf_mask_bits<NT>(m64, st.bp, (NCH_) - 2);                                               \
            nf_mask_bits<NT>(m64, bj, (NCH_) - 1);                                                  \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                          \
                if (p0 + 16 * t < n)                                                                \
                    *nf_mask_ptr(saved, n, MASKL_, (p0 >> 4) + t, lane) = make_uint2((uint32_t)m64[t], (uint32_t)(m64[t] >> 32)); \
        }                                                                                           \
    } while (0)
    // one 256-wide layer from the slab: the slab = section SEC_ (ReLU layer MASKL_, or -1: as stored) is copied out and consumed
#define NF_LAYER256(OFF_, FIRST_, SEC_, MASKL_, OFF_NEXT_, B_NEXT_, NO_NEXT_, NEXT_B_)                                    \
    do {                                                                                                                   \
        NfCopyH<64, 4, ((MASKL_) >= 0)> cs{act4, sec(SEC_, 256), lane, 8, {}};                                             \
        cs.prime();                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) m64[t] = 0;                                                         \
        nf_seg_lds<NT, 16, FIRST_, ((MASKL_) >= 0), ((MASKL_) >= 0)>(acc, st, Wi, OFF_, 16, act4, lane, cs, m64);          \
        NF_PENDING(((MASKL_) >= 0), MASKL_, 16);                                                                           \
        nf_tail<NT, 16, 16, NO_NEXT_, NEXT_B_>(acc, st.wb, bj, st, Wi, OFF_NEXT_, Ci, B_NEXT_, act4, lane);               \
    } while (0)
    // ---- layers_xyz.0 : PE(64 slots) -> 256 ------------------------------------------------------------
    nf_load_bias<16>(st.bias, Ci, B_L0, lane);
    {
        f32x4 w[16];
        nf_load_w16<16>(w, Wi, OFF_L0 / 4, lane);
        NF_PE_B(0); nf_chunk<NT, 16, true>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L0 / 4 + 1 * 16 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L0 / 4 + 2 * 16 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L0 / 4 + 3 * 16 * 64, lane);
        NF_PE_B(3); nf_tail<NT, 16, 16, 16, 1>(acc, w, bj, st, Wi, OFF_L1 / 4, Ci, B_L1, act4, lane);
    }
    // ---- layers_xyz.1, .2 (each K loop also streams the layer output it consumes to `saved`) -------------
    NF_LAYER256(OFF_L1 / 4, true, S_H0, 0, OFF_L2 / 4, B_L2, 16, 1);
    NF_LAYER256(OFF_L2 / 4, true, S_H1, 1, OFF_L3 / 4, B_L3, 16, 0);
    // ---- layers_xyz.3 : [PE | h] -> 256 (skip connection, M:246) ------------------------------------
    NF_PE_B(0); nf_chunk<NT, 16, true>(acc, st.wa, bj, st.bias);
    nf_load_w16<16>(st.wa, Wi, OFF_L3 / 4 + 4 * 16 * 64, lane);            // the first slab chunk, three chunks ahead
    nf_read_b<NT>(st.b0, act4, lane, 0);
    {
        f32x4 w[16];
        nf_load_w16<16>(w, Wi, OFF_L3 / 4 + 1 * 16 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L3 / 4 + 2 * 16 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L3 / 4 + 3 * 16 * 64, lane);
        NF_PE_B(3); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
    }
    NF_LAYER256(OFF_L3 / 4 + 4 * 16 * 64, false, S_H2, 2, OFF_L4 / 4, B_L4, 16, 1);
    // ---- layers_xyz.4, .5, fc_feat (no activation, M:250) -----------------------------------------
    NF_LAYER256(OFF_L4 / 4, true, S_H3, 3, OFF_L5 / 4, B_L5, 16, 1);
    NF_LAYER256(OFF_L5 / 4, true, S_H4, 4, OFF_FEAT / 4, B_FEAT, 16, 1);
    NF_LAYER256(OFF_FEAT / 4, true, S_H5, 5, OFF_D0 / 4, B_D0, 9, 1);
    // ---- layers_dir.0 : [feat | dir slots] -> 128; tile 8 row 0 = fc_alpha(feat) (Q2) -------------------
    float sigma_raw[NT];
    {
        f32x4 wd[16];
        nf_load_w16<9>(wd, Wi, OFF_D0 / 4 + 16 * 9 * 64, lane);             // the dir-slot chunk's weights, a layer ahead
        NfCopyH<64, 4, false> cs{act4, sec(S_FEAT, 256), lane, 8, {}};
        cs.prime();
        nf_seg_lds<NT, 9, true, false, false>(acc, st, Wi, OFF_D0 / 4, 16, act4, lane, cs, m64);
        NF_PENDING(false, -1, 16);
        nf_chunk<NT, 9, false>(acc, st.wb, bj, st.bias);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][0];
        nf_tail<NT, 9, 8, 8, 1>(acc, wd, bj, st, Wi, OFF_D1 / 4, Ci, B_D1, act4, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) sigma_raw[t] = acc[t][8].x;
    }
    // ---- layers_dir.1, .2 (128-wide rows: two per copy instruction) -------------------------------------
    {
        NfCopyH<32, 4, true> cs{act4, sec(S_D0, 128), lane, 4, {}};
        cs.prime();
#pragma unroll
        for (int t = 0; t < NT; ++t) m64[t] = 0;
        nf_seg_lds<NT, 8, true, true, true>(acc, st, Wi, OFF_D1 / 4, 8, act4, lane, cs, m64);
        NF_PENDING(true, 6, 8);
        nf_tail<NT, 8, 8, 8, 1>(acc, st.wb, bj, st, Wi, OFF_D2 / 4, Ci, B_D2, act4, lane);
    }
    {
        NfCopyH<32, 4, true> cs{act4, sec(S_D1, 128), lane, 4, {}};
        cs.prime();
#pragma unroll
        for (int t = 0; t < NT; ++t) m64[t] = 0;
        nf_seg_lds<NT, 8, true, true, true>(acc, st, Wi, OFF_D2 / 4, 8, act4, lane, cs, m64);
        NF_PENDING(true, 7, 8);
        nf_tail<NT, 8, 8, 1, 1>(acc, st.wb, bj, st, Wi, OFF_RGB / 4, Ci, B_RGB, act4, lane);
    }
    // ---- fc_rgb -------------------------------------------------------------------------------------------
    {
        NfCopyH<32, 4, true> cs{act4, sec(S_D2, 128), lane, 4, {}};
        cs.prime();
#pragma unroll
        for (int t = 0; t < NT; ++t) m64[t] = 0;
        nf_seg_lds<NT, 1, true, true, true>(acc, st, Wi, OFF_RGB / 4, 8, act4, lane, cs, m64);
        NF_PENDING(true, 8, 8);
        nf_chunk<NT, 1, false>(acc, st.wb, bj, st.bias);
    }
#undef NF_LAYER256
#undef NF_PENDING
#undef NF_PE_B
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t p = p0 + 16 * t + c;
            if (p < n_points)
                reinterpret_cast<f32x4*>(raw)[p] = (f32x4){acc[t][0].x, acc[t][0].y, acc[t][0].z, sigma_raw[t]};
        }
    }
}

#ifndef NF_F32_RR
#define NF_F32_RR 0
#endif
#if NF_F32_RR
#include "nf_mlp_rr.h"
#endif

static int nf_launch_fwd(const float* packed, const float* cond, const float* ro, const float* rd, const float* rd_view,
                         const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    if (saved && n_points >= ((int64_t)1 << 22)) return NF_EINVAL;       // the save path addresses a section with 32-bit byte offsets (1 KiB per point)
    if (saved)
        hipLaunchKernelGGL((k_paper_mlp_fwd_save<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed,
                           cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, saved);
    else
#if NF_F32_RR
        hipLaunchKernelGGL((k_paper_mlp_fwd_rr<NT>), dim3((unsigned)(grid < nf_cu_count() ? grid : nf_cu_count())), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed,
                           cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw);
#else
        hipLaunchKernelGGL((k_paper_mlp_fwd<NT>), dim3((unsigned)(grid < nf_cu_count() ? grid : nf_cu_count())), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed,
                           cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw);
#endif
    NF_RETURN_LAUNCH();
}

extern "C" int nf_paper_mlp_fwd(const float* packed, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    return nf_launch_fwd(packed, cond, ro, rd, rd_view, z, n_rays, n_samples, raw, nullptr, stream);
}

// + one point tile of mask words: the exact-f32 masks are kept per 16-point tile, ceil(n / 16) of them per layer
// (sized for the split training layout too: its sections are n_points rounded up to 32 points long, nf_mlp_bf16_machinery.inc)
extern "C" size_t nf_paper_saved_floats(int64_t n_points) { return (size_t)nfl::SAVED_PER_POINT * (size_t)((n_points + 31) & ~(int64_t)31) + 9 * 128; }

extern "C" int nf_paper_mlp_fwd_train(const float* packed, const float* cond, const float* ro, const float* rd,
                                      const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                                      float* saved, nf_stream_t stream) {
    if (!saved) return NF_EINVAL;
    return nf_launch_fwd(packed, cond, ro, rd, rd_view, z, n_rays, n_samples, raw, saved, stream);
}
