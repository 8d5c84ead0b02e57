// ConditionalBlendshapePaperNeRFModel.forward on PRE-ENCODED inputs (reference nerf/models.py:236-261 as called by
// run_network, nerf/train_utils.py:20-24): x (P, 87) = [PE10(xyz) (63) | PE4(dirs) (24)] -> (P, 4).  Inference only; the
// hot path (run_one_iter_of_nerf) never materialises x and uses nf_paper_mlp_fwd instead.  Own translation unit on purpose.
#include "nf_mlp_dev.h"
#include "nf_mlp_stream.h"

// bias table without the direction fold: the 24 direction columns arrive with x
__global__ void __launch_bounds__(256) k_paper_condition_encoded(const float* __restrict__ packed, const float* __restrict__ expr,
                                                                 const float* __restrict__ latent, float* __restrict__ cond) {
    using namespace nfl;
    __shared__ float cvec[NCOND];
    const int tid = threadIdx.x;
    if (tid < 76) cvec[tid] = nf_div(nf_mul(expr[tid], 1.0f), 3.0f);
    else if (tid < NCOND) cvec[tid] = latent[tid - 76];
    __syncthreads();
    const float* bias = packed + OFF_BIAS;
    for (int i = blockIdx.x * blockDim.x + tid; i < COND_FLOATS; i += gridDim.x * blockDim.x) {
        if (i >= B_CVEC) { cond[i] = i < B_DVEC ? cvec[i - B_CVEC] : 0.0f; continue; }
        float v = bias[i];
        if (i < B_L1 || (i >= B_L3 && i < B_L4)) {
            const int n = i < B_L1 ? i : i - B_L3;
            const float* w = packed + (i < B_L1 ? OFF_WC0 : OFF_WC3) + n * NCOND;
            float s = 0.0f;
            for (int k = 0; k < NCOND; ++k) s = fmaf(w[k], cvec[k], s);
            v += s;
        }
        cond[i] = v;
    }
}

template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_paper_mlp_fwd_encoded(const float* __restrict__ packed, const float* __restrict__ cond, const float* __restrict__ x87,
                        int64_t n_points, float* __restrict__ out) {
    using namespace nfl;
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    f32x4 pe[NT][4];
    f32x4 dirf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
        const float* row = x87 + p * 87;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = pe_slot_to_col(16 * j + 4 * g + r);
                v[r] = col >= 0 ? row[col] : 0.0f;
            }
            pe[t][j] = (f32x4){v[0], v[1], v[2], v[3]};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 16 * j + 4 * g + r;
                v[r] = s < 24 ? row[63 + s] : 0.0f;
            }
            dirf[t][j] = (f32x4){v[0], v[1], v[2], v[3]};
        }
    }
    // Layer-streamed form, as k_paper_mlp_fwd (nf_mlp_stream.h): raw accumulators to the slab under the last K chunk, the bias as the C
    // operand of a layer's first MFMAs, the ReLU where the slab is read.  Differences: the inputs came from x87 above, and layers_dir.0
    // takes its 24 direction columns as two register chunks (weights OFF_D0E: 16 feature chunks, then 2 direction chunks of 9 tiles).
    f32x4 acc[NT][16];
    NfStream<NT> st;
    const NfW Wi = nf_w_image(packed, PACKED_FLOATS), Ci = nf_w_image(cond, COND_FLOATS);
    f32x4 bj[NT];
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
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
#define NF_ENC_LAYER256(OFF_, OFF_NEXT_, B_NEXT_, NO_NEXT_, NEXT_B_)                                                        \
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, (OFF_) / 4, 16, act4, lane);                                                \
    nf_pending_b<NT, true>(bj, st);                                                                                         \
    nf_tail<NT, 16, 16, NO_NEXT_, NEXT_B_>(acc, st.wb, bj, st, Wi, (OFF_NEXT_) / 4, Ci, B_NEXT_, act4, lane)
    NF_ENC_LAYER256(OFF_L1, OFF_L2, B_L2, 16, 1);
    NF_ENC_LAYER256(OFF_L2, OFF_L3, B_L3, 16, 0);
    // layers_xyz.3 : [PE | h] -> 256 (skip connection, M:246)
    NF_PE_B(0); nf_chunk<NT, 16, true>(acc, st.wa, bj, st.bias);
    nf_load_w16<16>(st.wa, Wi, OFF_L3 / 4 + 4 * 16 * 64, lane);
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
    nf_tail<NT, 16, 16, 16, 1>(acc, st.wb, bj, st, Wi, OFF_L4 / 4, Ci, B_L4, act4, lane);
    NF_ENC_LAYER256(OFF_L4, OFF_L5, B_L5, 16, 1);
    NF_ENC_LAYER256(OFF_L5, OFF_FEAT, B_FEAT, 16, 1);
    NF_ENC_LAYER256(OFF_FEAT, OFF_D0E, B_D0, 9, 1);             // fc_feat: no activation (M:250), layers_dir.0 reads it as stored
#undef NF_ENC_LAYER256
#undef NF_PE_B
    // layers_dir.0 : [feat | 24 direction columns] -> 128; tile 8 row 0 = fc_alpha(feat) (Q2)
    float sigma_raw[NT];
    {
        f32x4 wd[16];
        nf_load_w16<9>(wd, Wi, OFF_D0E / 4 + 16 * 9 * 64, lane);
        nf_seg_lds<NT, 9, true, false>(acc, st, Wi, OFF_D0E / 4, 16, act4, lane);
        nf_pending_b<NT, false>(bj, st);
        nf_chunk<NT, 9, false>(acc, st.wb, bj, st.bias);
        nf_load_w16<9>(st.wb, Wi, OFF_D0E / 4 + 17 * 9 * 64, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][0];
        nf_chunk<NT, 9, false>(acc, wd, bj, st.bias);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][1];
        nf_tail<NT, 9, 8, 8, 1>(acc, st.wb, bj, st, Wi, OFF_D1 / 4, Ci, B_D1, act4, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) sigma_raw[t] = acc[t][8].x;
    }
    nf_seg_lds<NT, 8, true, true>(acc, st, Wi, OFF_D1 / 4, 8, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 8, 8, 8, 1>(acc, st.wb, bj, st, Wi, OFF_D2 / 4, Ci, B_D2, act4, lane);
    nf_seg_lds<NT, 8, true, true>(acc, st, Wi, OFF_D2 / 4, 8, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 8, 8, 1, 1>(acc, st.wb, bj, st, Wi, OFF_RGB / 4, Ci, B_RGB, act4, lane);
    nf_seg_lds<NT, 1, true, true>(acc, st, Wi, OFF_RGB / 4, 8, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_chunk<NT, 1, false>(acc, st.wb, bj, st.bias);
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t p = p0 + 16 * t + c;
            if (p < n_points) reinterpret_cast<f32x4*>(out)[p] = (f32x4){acc[t][0].x, acc[t][0].y, acc[t][0].z, sigma_raw[t]};
        }
    }
}

// x87: (n_points, 87) pre-encoded inputs; cond: scratch of nf_paper_cond_floats() floats; out: (n_points, 4).
extern "C" int nf_paper_forward_encoded(const float* packed, const float* x87, const float* expr76, const float* latent32,
                                        int64_t n_points, float* cond, float* out, nf_stream_t stream) {
    if (n_points == 0) return 0;                           // nothing to do (empty tensors have NULL data pointers)
    if (!packed || !x87 || !expr76 || !latent32 || !cond || !out || n_points < 0) return NF_EINVAL;
    hipLaunchKernelGGL(k_paper_condition_encoded, dim3((nfl::COND_FLOATS + 255) / 256), dim3(256), 0, nf_s(stream), packed, expr76,
                       latent32, cond);
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL((k_paper_mlp_fwd_encoded<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, cond, x87,
                       n_points, out);
    NF_RETURN_LAUNCH();
}
