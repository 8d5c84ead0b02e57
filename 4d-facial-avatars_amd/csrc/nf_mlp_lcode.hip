// Second model family: ConditionalBlendshapeLearnableCodeNeRFModel (reference nerf/models.py:529-636) as every NeRFace
// config instantiates it (num_layers 4, hidden 256, no skip, 10/4 encoding functions, include_input_dir False):
//   x = layer1([PE(63) | expr/3 (76) | latent (32)])          (no activation)
//   3 x relu(layers_xyz.i(x));  feat = relu(fc_feat(x));  sigma = fc_alpha(x)   (reads x, not feat)
//   relu(layers_dir.0([feat | PE4(dir) (24)])) -> rgb = fc_rgb
// Inference forward, exact-f32 MFMA, same building blocks / folding rules as the paper model (nf_mlp.hip).
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"
#include "nf_mlp_stream.h"

#include "nf_mlp_lcode_layout.h"
#include "nf_pack.h"



static void nf_lcode_table(std::vector<uint32_t>& t) {
    using namespace nlc;
    const uint32_t Z = 0xFF000000u;
    t.assign(PACKED, Z);
    auto code = [](int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); };
    auto fill = [&](int off, int nk, int no_tiles, int tensor, int n_out, int n_cols, auto col_of) {
        for (int ni = 0; ni < nk; ++ni)
            for (int no = 0; no < no_tiles; ++no)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int g = lane >> 4, i = lane & 15, n = 16 * no + i, col = col_of(16 * ni + 4 * g + r);
                        if (n < n_out && col >= 0) t[(size_t)off + ((size_t)(ni * no_tiles + no) * 64 + lane) * 4 + r] = code(tensor, n, col, n_cols);
                    }
    };
    auto ident = [](int s) { return s; };
    fill(OFF_L1, 4, 16, 0, 256, 171, [](int s) { return nfl::pe_slot_to_col(s); });
    fill(OFF_X0, 16, 16, 2, 256, 256, ident);
    fill(OFF_X1, 16, 16, 4, 256, 256, ident);
    fill(OFF_X2, 16, 16, 6, 256, 256, ident);
    fill(OFF_ALPHA, 16, 1, 10, 1, 256, ident);
    fill(OFF_FEAT, 16, 16, 14, 256, 256, ident);
    fill(OFF_DIR, 17, 8, 8, 128, 280, [](int s) {
        if (s < 256) return s;
        const int g = ((s - 256) >> 2) & 3, r = (s - 256) & 3;
        return r < 2 ? 256 + 6 * g + 3 * r : -1;
    });
    fill(OFF_RGB, 8, 1, 12, 3, 128, ident);
    fill(OFF_DIRE, 18, 8, 8, 128, 280, [](int s) { return s < 280 ? s : -1; });
    for (int n = 0; n < 256; ++n)
        for (int k = 0; k < 108; ++k) t[OFF_WC1 + n * 108 + k] = code(0, n, 63 + k, 171);
    for (int n = 0; n < 128; ++n)
        for (int f = 0; f < 4; ++f)
            for (int sc = 0; sc < 2; ++sc)
                for (int comp = 1; comp < 3; ++comp) t[OFF_WCD + n * 16 + 4 * f + 2 * sc + (comp - 1)] = code(8, n, 256 + 6 * f + 3 * sc + comp, 280);
    const int b256[5] = {1, 3, 5, 7, 15};          // layer1, layers_xyz.0..2, fc_feat biases
    for (int l = 0; l < 5; ++l)
        for (int n = 0; n < 256; ++n) t[OFF_BIAS + 256 * l + n] = code(b256[l], 0, n, 256);
    t[OFF_BIAS + B_ALPHA] = code(11, 0, 0, 1);
    for (int n = 0; n < 128; ++n) t[OFF_BIAS + B_DIR + n] = code(9, 0, n, 128);
    for (int n = 0; n < 3; ++n) t[OFF_BIAS + B_RGB + n] = code(13, 0, n, 3);
}

static NfPackTable g_lcode_table;

extern "C" size_t nf_lcode_packed_floats(void) { return (size_t)nlc::PACKED; }
// padded to 10 KiB: the split-bf16 kernel stages the table into LDS with ten 1-KiB DMA pieces
extern "C" size_t nf_lcode_cond_floats(void) { return 2560; }

extern "C" int nf_lcode_pack(const float* const* params, float* packed, nf_stream_t stream) {
    return nf_pack_f32<nlc::NPARAMS, 4>(g_lcode_table, nf_lcode_table, params, packed, (int)nlc::PACKED, stream);
}

__global__ void __launch_bounds__(256) k_lcode_condition(const float* __restrict__ packed, const float* __restrict__ expr,
                                                         const float* __restrict__ latent, float near_z, float far_z,
                                                         float* __restrict__ cond) {
    using namespace nlc;
    __shared__ float cvec[108];
    __shared__ float dvec[16];
    const int tid = threadIdx.x;
    if (tid < 76) cvec[tid] = nf_div(nf_mul(expr[tid], 1.0f), 3.0f);
    else if (tid < 108) cvec[tid] = latent[tid - 76];
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
        if (i < B_X0) {
            const float* w = packed + OFF_WC1 + i * 108;
            float s = 0.0f;
            for (int k = 0; k < 108; ++k) s = fmaf(w[k], cvec[k], s);
            v += s;
        } else if (i >= B_DIR && i < B_DIR + 128) {
            const float* w = packed + OFF_WCD + (i - B_DIR) * 16;
            float s = 0.0f;
            for (int k = 0; k < 16; ++k) s = fmaf(w[k], dvec[k], s);
            v += s;
        }
        cond[i] = v;
    }
}

extern "C" int nf_lcode_condition(const float* packed, const float* expr76, const float* latent32, float near_z, float far_z,
                                  float* cond, nf_stream_t stream) {
    if (!packed || !expr76 || !latent32 || !cond) return NF_EINVAL;
    hipLaunchKernelGGL(k_lcode_condition, dim3((nlc::COND_FLOATS + 255) / 256), dim3(256), 0, nf_s(stream), packed, expr76, latent32,
                       near_z, far_z, cond);
    NF_RETURN_LAUNCH();
}

template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_lcode_mlp_fwd(const float* __restrict__ packed, const float* __restrict__ cond, const float* __restrict__ ro,
                const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z, int64_t n_points, int S,
                float* __restrict__ raw) {
    using namespace nlc;
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    f32x4 pe[NT][4];
    f32x4 dirf[NT][1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
        const int64_t ray = p / S;
        const float zz = z[p];
        const float px = nf_add(ro[ray * 3 + 0], nf_mul(rd[ray * 3 + 0], zz));
        const float py = nf_add(ro[ray * 3 + 1], nf_mul(rd[ray * 3 + 1], zz));
        const float pz = nf_add(ro[ray * 3 + 2], nf_mul(rd[ray * 3 + 2], zz));
        nf_encode_point(px, py, pz, g, pe[t]);
        float s, cs;
        nf_sincos(nf_mul(rd_view[ray * 3 + 2], (float)(1 << g)), &s, &cs);
        dirf[t][0] = (f32x4){s, cs, 0.0f, 0.0f};
    }
    f32x4 acc[NT][16];
    // Layer-streamed form (nf_mlp_stream.h; the paper model's k_paper_mlp_fwd is the template).  layers_xyz.2's output feeds fc_alpha AND
    // fc_feat: fc_alpha's tail stores nothing (NO_ST = 0), so fc_feat reads the same slab again.
    NfStream<NT> st;
    f32x4 bj[NT];
    float sigma_raw[NT];
    const NfW Wi = nf_w_image(packed, PACKED), Ci = nf_w_image(cond, COND_FLOATS);
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
    nf_load_bias<16>(st.bias, Ci, B_L1, lane);                           // layer1: no activation (M:609)
    {
        f32x4 w[16];
        nf_load_w16<16>(w, Wi, OFF_L1 / 4, lane);
        NF_PE_B(0); nf_chunk<NT, 16, true>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 1 * 16 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 2 * 16 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 3 * 16 * 64, lane);
        NF_PE_B(3); nf_tail<NT, 16, 16, 16, 1>(acc, w, bj, st, Wi, OFF_X0 / 4, Ci, B_X0, act4, lane);
    }
#undef NF_PE_B
    nf_seg_lds<NT, 16, true, false>(acc, st, Wi, OFF_X0 / 4, 16, act4, lane);       // reads layer1's output as stored
    nf_pending_b<NT, false>(bj, st);
    nf_tail<NT, 16, 16, 16, 1>(acc, st.wb, bj, st, Wi, OFF_X1 / 4, Ci, B_X1, act4, lane);
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_X1 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 16, 1>(acc, st.wb, bj, st, Wi, OFF_X2 / 4, Ci, B_X2, act4, lane);
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_X2 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 1, 1>(acc, st.wb, bj, st, Wi, OFF_ALPHA / 4, Ci, B_ALPHA, act4, lane);
    nf_seg_lds<NT, 1, true, true>(acc, st, Wi, OFF_ALPHA / 4, 16, act4, lane);      // fc_alpha(x)
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 1, 0, 16, 1>(acc, st.wb, bj, st, Wi, OFF_FEAT / 4, Ci, B_FEAT, act4, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) sigma_raw[t] = acc[t][0].x;
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_FEAT / 4, 16, act4, lane);      // feat = relu(fc_feat(x)): the ReLU is the reader's
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 8, 1>(acc, st.wb, bj, st, Wi, OFF_DIR / 4, Ci, B_DIR, act4, lane);
    {
        f32x4 wd[16];
        nf_load_w16<8>(wd, Wi, OFF_DIR / 4 + 16 * 8 * 64, lane);                    // the dir-slot chunk's weights, a layer ahead
        nf_seg_lds<NT, 8, true, true>(acc, st, Wi, OFF_DIR / 4, 16, act4, lane);    // relu(layers_dir.0([feat | dir]))
        nf_pending_b<NT, true>(bj, st);
        nf_chunk<NT, 8, false>(acc, st.wb, bj, st.bias);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][0];
        nf_tail<NT, 8, 8, 1, 1>(acc, wd, bj, st, Wi, OFF_RGB / 4, Ci, B_RGB, act4, lane);
    }
    nf_seg_lds<NT, 1, true, true>(acc, st, Wi, OFF_RGB / 4, 8, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_chunk<NT, 1, false>(acc, st.wb, bj, st.bias);
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t p = p0 + 16 * t + c;
            if (p < n_points) reinterpret_cast<f32x4*>(raw)[p] = (f32x4){acc[t][0].x, acc[t][0].y, acc[t][0].z, sigma_raw[t]};
        }
    }
}

// ConditionalBlendshapeLearnableCodeNeRFModel.forward on PRE-ENCODED inputs (reference nerf/models.py:590-636 as called by run_network,
// nerf/train_utils.py:9-33): x (P, 87) = [PE10(xyz) (63) | PE4(dirs) (24)] -> (P, 4).  Inference only; the hot path
// (run_one_iter_of_nerf) never materialises x and uses nf_lcode_mlp_fwd.  Same body as k_lcode_mlp_fwd, inputs from x87, layers_dir.0
// with its 24 direction columns as two register chunks (weights OFF_DIRE), bias table without the direction fold.
__global__ void __launch_bounds__(256) k_lcode_condition_encoded(const float* __restrict__ packed, const float* __restrict__ expr,
                                                                 const float* __restrict__ latent, float* __restrict__ cond) {
    using namespace nlc;
    __shared__ float cvec[108];
    const int tid = threadIdx.x;
    if (tid < 76) cvec[tid] = nf_div(nf_mul(expr[tid], 1.0f), 3.0f);
    else if (tid < 108) cvec[tid] = latent[tid - 76];
    __syncthreads();
    const float* bias = packed + OFF_BIAS;
    for (int i = blockIdx.x * blockDim.x + tid; i < COND_FLOATS; i += gridDim.x * blockDim.x) {
        if (i >= B_CVEC) { cond[i] = i < B_DVEC ? cvec[i - B_CVEC] : 0.0f; continue; }
        float v = bias[i];
        if (i < B_X0) {
            const float* w = packed + OFF_WC1 + i * 108;
            float s = 0.0f;
            for (int k = 0; k < 108; ++k) s = fmaf(w[k], cvec[k], s);
            v += s;
        }
        cond[i] = v;
    }
}

template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_lcode_mlp_fwd_encoded(const float* __restrict__ packed, const float* __restrict__ cond, const float* __restrict__ x87, int64_t n_points,
                        float* __restrict__ out) {
    using namespace nlc;
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
                const int col = nfl::pe_slot_to_col(16 * j + 4 * g + r);
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
    f32x4 acc[NT][16];
    NfStream<NT> st;
    f32x4 bj[NT];
    float sigma_raw[NT];
    const NfW Wi = nf_w_image(packed, PACKED), Ci = nf_w_image(cond, COND_FLOATS);
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
    nf_load_bias<16>(st.bias, Ci, B_L1, lane);                           // layer1: no activation (M:609)
    {
        f32x4 w[16];
        nf_load_w16<16>(w, Wi, OFF_L1 / 4, lane);
        NF_PE_B(0); nf_chunk<NT, 16, true>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 1 * 16 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 2 * 16 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 3 * 16 * 64, lane);
        NF_PE_B(3); nf_tail<NT, 16, 16, 16, 1>(acc, w, bj, st, Wi, OFF_X0 / 4, Ci, B_X0, act4, lane);
    }
#undef NF_PE_B
    nf_seg_lds<NT, 16, true, false>(acc, st, Wi, OFF_X0 / 4, 16, act4, lane);       // reads layer1's output as stored
    nf_pending_b<NT, false>(bj, st);
    nf_tail<NT, 16, 16, 16, 1>(acc, st.wb, bj, st, Wi, OFF_X1 / 4, Ci, B_X1, act4, lane);
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_X1 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 16, 1>(acc, st.wb, bj, st, Wi, OFF_X2 / 4, Ci, B_X2, act4, lane);
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_X2 / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 1, 1>(acc, st.wb, bj, st, Wi, OFF_ALPHA / 4, Ci, B_ALPHA, act4, lane);
    nf_seg_lds<NT, 1, true, true>(acc, st, Wi, OFF_ALPHA / 4, 16, act4, lane);      // fc_alpha(x)
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 1, 0, 16, 1>(acc, st.wb, bj, st, Wi, OFF_FEAT / 4, Ci, B_FEAT, act4, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) sigma_raw[t] = acc[t][0].x;
    nf_seg_lds<NT, 16, true, true>(acc, st, Wi, OFF_FEAT / 4, 16, act4, lane);      // feat = relu(fc_feat(x)): the ReLU is the reader's
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 16, 16, 8, 1>(acc, st.wb, bj, st, Wi, OFF_DIRE / 4, Ci, B_DIR, act4, lane);
    {
        f32x4 wd[16];
        nf_load_w16<8>(wd, Wi, OFF_DIRE / 4 + 16 * 8 * 64, lane);                   // the first direction chunk's weights, a layer ahead
        nf_seg_lds<NT, 8, true, true>(acc, st, Wi, OFF_DIRE / 4, 16, act4, lane);   // relu(layers_dir.0([feat | dir]))
        nf_pending_b<NT, true>(bj, st);
        nf_chunk<NT, 8, false>(acc, st.wb, bj, st.bias);
        nf_load_w16<8>(st.wb, Wi, OFF_DIRE / 4 + 17 * 8 * 64, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][0];
        nf_chunk<NT, 8, false>(acc, wd, bj, st.bias);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][1];
        nf_tail<NT, 8, 8, 1, 1>(acc, st.wb, bj, st, Wi, OFF_RGB / 4, Ci, B_RGB, act4, lane);
    }
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

// x87: (n_points, 87) pre-encoded inputs; cond: scratch of nf_lcode_cond_floats() floats; out: (n_points, 4).
extern "C" int nf_lcode_forward_encoded(const float* packed, const float* x87, const float* expr76, const float* latent32,
                                        int64_t n_points, float* cond, float* out, nf_stream_t stream) {
    if (n_points == 0) return 0;                           // nothing to do (empty tensors have NULL data pointers)
    if (!packed || !x87 || !expr76 || !latent32 || !cond || !out || n_points < 0) return NF_EINVAL;
    hipLaunchKernelGGL(k_lcode_condition_encoded, dim3((nlc::COND_FLOATS + 255) / 256), dim3(256), 0, nf_s(stream), packed, expr76,
                       latent32, cond);
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL((k_lcode_mlp_fwd_encoded<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, cond, x87,
                       n_points, out);
    NF_RETURN_LAUNCH();
}

// Training forward (exact f32): the same arithmetic plus `saved` (layout nlc::S_*) -- every layer output as whole rows out of the
// wave's LDS slab from inside the next layer's K loop, ReLU bit masks beside them (nf_mlp_dev.h, nf_mlp_stream.h; the paper model's
// k_paper_mlp_fwd_save is the template).
template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_lcode_mlp_fwd_save(const float* __restrict__ packed, const float* __restrict__ cond, const float* __restrict__ ro,
                     const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z, int64_t n_points, int S,
                     float* __restrict__ raw, float* __restrict__ saved) {
    using namespace nlc;
    static_assert(NT == 2, "the copy schedule below is written for 32-point slabs");
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    const int64_t n = n_points;
    auto sec = [&](int s_, int width) { return nf_slab_copy(saved, s_, width, p0, n); };
    f32x4 pe[NT][4];
    f32x4 dirf[NT][1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
        const int64_t ray = p / S;
        const float zz = z[p];
        const float px = nf_add(ro[ray * 3 + 0], nf_mul(rd[ray * 3 + 0], zz));
        const float py = nf_add(ro[ray * 3 + 1], nf_mul(rd[ray * 3 + 1], zz));
        const float pz = nf_add(ro[ray * 3 + 2], nf_mul(rd[ray * 3 + 2], zz));
        nf_encode_point(px, py, pz, g, pe[t]);
        float s, cs;
        nf_sincos(nf_mul(rd_view[ray * 3 + 2], (float)(1 << g)), &s, &cs);
        dirf[t][0] = (f32x4){s, cs, 0.0f, 0.0f};
        if (p0 + 16 * t + c < n_points) *reinterpret_cast<f32x4*>(saved + S_DIRF * n_points + p * 16 + 4 * g) = dirf[t][0];
#pragma unroll
        for (int j = 0; j < 4; ++j) act4[nf_act_idx4(16 * t + c, 4 * j + g)] = pe[t][j];
    }
    {
        const NfSlabCopy cp = sec(S_PE, 64);
#pragma unroll
        for (int k = 0; k < 16 * NT / 4; ++k) nf_copy_rows<16>(act4, cp, k, lane);
    }
    // Layer-streamed like k_paper_mlp_fwd_save (nf_mlp.hip) and this family's inference kernel above: bias as the C operand, the layer
    // boundary under the last chunk's MFMAs (raw accumulators to the slab), the next layer prefetched; what the backward needs is produced
    // where the slab is READ -- the loop that consumes a layer's output applies the ReLU to its B fragments, collects their [x > 0] bits
    // and carries the copy of the slab to `saved`.  layers_xyz.2's output is read twice (fc_alpha, fc_feat): copied and masked under fc_feat.
    f32x4 acc[NT][16];
    uint64_t m64[NT];
    NfStream<NT> st;
    f32x4 bj[NT];
    float sigma_raw[NT];
    const NfW Wi = nf_w_image(packed, PACKED), Ci = nf_w_image(cond, COND_FLOATS);
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
    // the last chunk's fragment (+ its mask bits), then the finished mask words of ReLU layer MASKL_ (the layer whose output was just consumed)
#define NF_LC_PENDING(RELU_, MASKL_, NCH_)                                                          \
    do {                                                                                            \
        nf_pending_b<NT, RELU_>(bj, st);                                                            \
        if ((MASKL_) >= 0) {                                                                        \
            nf_mask_bits<NT>(m64, st.bp, (NCH_) - 2);                                               \
            nf_mask_bits<NT>(m64, bj, (NCH_) - 1);                                                  \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                          \
                if (p0 + 16 * t < n)                                                                \
                    *nf_mask_ptr<S_MASK>(saved, n, (MASKL_) >= 0 ? (MASKL_) : 0, (p0 >> 4) + t, lane) = make_uint2((uint32_t)m64[t], (uint32_t)(m64[t] >> 32)); \
        }                                                                                           \
    } while (0)
    // one 256-wide layer from the slab: the slab = section SEC_ (ReLU layer MASKL_, or -1: as stored) is copied out and consumed
#define NF_LC_LAYER256(OFF_, SEC_, MASKL_, OFF_NEXT_, B_NEXT_, NO_NEXT_)                                                   \
    do {                                                                                                                   \
        NfCopyH<64, 4, ((MASKL_) >= 0)> cs{act4, sec(SEC_, 256), lane, 8, {}};                                             \
        cs.prime();                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) m64[t] = 0;                                                         \
        nf_seg_lds<NT, 16, true, ((MASKL_) >= 0), ((MASKL_) >= 0)>(acc, st, Wi, OFF_, 16, act4, lane, cs, m64);            \
        NF_LC_PENDING(((MASKL_) >= 0), MASKL_, 16);                                                                        \
        nf_tail<NT, 16, 16, NO_NEXT_, 1>(acc, st.wb, bj, st, Wi, OFF_NEXT_, Ci, B_NEXT_, act4, lane);                     \
    } while (0)
    // ---- layer1 : PE(64 slots) -> 256, no activation (M:609) ---------------------------------------------------------
    nf_load_bias<16>(st.bias, Ci, B_L1, lane);
    {
        f32x4 w[16];
        nf_load_w16<16>(w, Wi, OFF_L1 / 4, lane);
        NF_PE_B(0); nf_chunk<NT, 16, true>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 1 * 16 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 2 * 16 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 16, false>(acc, w, bj, st.bias);
        nf_load_w16<16>(w, Wi, OFF_L1 / 4 + 3 * 16 * 64, lane);
        NF_PE_B(3); nf_tail<NT, 16, 16, 16, 1>(acc, w, bj, st, Wi, OFF_X0 / 4, Ci, B_X0, act4, lane);
    }
    // ---- layers_xyz.0 (reads layer1's output as stored), .1, .2 ---------------------------------------------------------
    NF_LC_LAYER256(OFF_X0 / 4, S_L1, -1, OFF_X1 / 4, B_X1, 16);
    NF_LC_LAYER256(OFF_X1 / 4, S_X0, 0, OFF_X2 / 4, B_X2, 16);
    NF_LC_LAYER256(OFF_X2 / 4, S_X1, 1, OFF_ALPHA / 4, B_ALPHA, 1);
    // ---- fc_alpha(x2): one tile, stores nothing (the slab stays for fc_feat), copies nothing -------------------------------
    nf_seg_lds<NT, 1, true, true>(acc, st, Wi, OFF_ALPHA / 4, 16, act4, lane);
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 1, 0, 16, 1>(acc, st.wb, bj, st, Wi, OFF_FEAT / 4, Ci, B_FEAT, act4, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) sigma_raw[t] = acc[t][0].x;
    // ---- feat = relu(fc_feat(x2)): x2 is copied and its mask collected here ---------------------------------------------
    NF_LC_LAYER256(OFF_FEAT / 4, S_X2, 2, OFF_DIR / 4, B_DIR, 8);
    // ---- layers_dir.0 : [feat | dir slots] -> 128 ---------------------------------------------------------------------------
    {
        f32x4 wd[16];
        nf_load_w16<8>(wd, Wi, OFF_DIR / 4 + 16 * 8 * 64, lane);             // the dir-slot chunk's weights, a layer ahead
        NfCopyH<64, 4, true> cs{act4, sec(S_FEAT, 256), lane, 8, {}};
        cs.prime();
#pragma unroll
        for (int t = 0; t < NT; ++t) m64[t] = 0;
        nf_seg_lds<NT, 8, true, true, true>(acc, st, Wi, OFF_DIR / 4, 16, act4, lane, cs, m64);
        NF_LC_PENDING(true, 3, 16);
        nf_chunk<NT, 8, false>(acc, st.wb, bj, st.bias);
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = dirf[t][0];
        nf_tail<NT, 8, 8, 1, 1>(acc, wd, bj, st, Wi, OFF_RGB / 4, Ci, B_RGB, act4, lane);
    }
    // ---- fc_rgb (128-wide rows: two per copy instruction) ---------------------------------------------------------------------
    {
        NfCopyH<32, 4, true> cs{act4, sec(S_DIR, 128), lane, 4, {}};
        cs.prime();
#pragma unroll
        for (int t = 0; t < NT; ++t) m64[t] = 0;
        nf_seg_lds<NT, 1, true, true, true>(acc, st, Wi, OFF_RGB / 4, 8, act4, lane, cs, m64);
        NF_LC_PENDING(true, 4, 8);
        nf_chunk<NT, 1, false>(acc, st.wb, bj, st.bias);
    }
#undef NF_LC_LAYER256
#undef NF_LC_PENDING
#undef NF_PE_B
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t p = p0 + 16 * t + c;
            if (p < n_points) reinterpret_cast<f32x4*>(raw)[p] = (f32x4){acc[t][0].x, acc[t][0].y, acc[t][0].z, sigma_raw[t]};
        }
    }
}

static int nf_lcode_launch_fwd(const float* packed, const float* cond, const float* ro, const float* rd, const float* rd_view,
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
        hipLaunchKernelGGL((k_lcode_mlp_fwd_save<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, cond, ro,
                           rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw, saved);
    else
        hipLaunchKernelGGL((k_lcode_mlp_fwd<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, cond, ro,
                           rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_lcode_mlp_fwd(const float* packed, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    return nf_lcode_launch_fwd(packed, cond, ro, rd, rd_view, z, n_rays, n_samples, raw, nullptr, stream);
}

// + one point tile of mask words (the exact-f32 masks are kept per 16-point tile)
// (sized for the split training layout too: its sections are n_points rounded up to 32 points long, nf_mlp_bf16_machinery.inc)
extern "C" size_t nf_lcode_saved_floats(int64_t n_points) { return (size_t)nlc::SAVED_PER_POINT * (size_t)((n_points + 31) & ~(int64_t)31) + 5 * 128; }

// Training forward: also fills `saved` (nf_lcode_saved_floats(n_points) floats), which nf_lcode_mlp_bwd reads.
extern "C" int nf_lcode_mlp_fwd_train(const float* packed, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                      const float* z, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (!saved) return NF_EINVAL;
    return nf_lcode_launch_fwd(packed, cond, ro, rd, rd_view, z, n_rays, n_samples, raw, saved, stream);
}
