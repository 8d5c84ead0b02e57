// BASELINE config 1 read literally ("tiny_nerf 64x64 image, 32 coarse samples, 4-layer MLP"): the tiny path of nf_tiny.hip with the
// reference's FlexibleNeRFModel (nerf/models.py:351-422) in place of VeryTinyNerfModel -- constructed as
//   FlexibleNeRFModel(num_layers = L, hidden_size = 128, num_encoding_fn_xyz = 10, include_input_xyz = True, use_viewdirs = False)
// with L = 2 .. 5 (below 6 layers the skip connection of models.py:373 / 404-409 never fires):
//   pts = ro + rd * depth -> PE(63) -> layer1 (Linear 128, NO activation: models.py:402) -> (L - 1) x [Linear 128 + ReLU] (layers_xyz)
//   -> fc_out (Linear 4).
// Same building blocks as the other exact-f32 kernels (nf_mlp_dev.h / nf_mlp_stream.h: v_mfma_f32_16x16x4_f32, wave = 32 points, no
// barrier, layer boundaries under the MFMAs); NH = L - 1 is a template parameter, the entry points take num_layers.
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"
#include "nf_mlp_stream.h"
#include "nf_pack.h"
#include "nf_mlp_dw.h"

#define NF_FLEX_MIN_LAYERS 2
#define NF_FLEX_MAX_LAYERS 5

// Offsets in floats.  Layers: 0 = layer1 (63 -> 128, linear), 1 .. NH = layers_xyz[0 .. NH - 1] (128 -> 128, ReLU), NH + 1 = fc_out.
template <int NH>
struct NfFlex {
    static constexpr int FRAG = 256;
    static constexpr int N_PARAMS = 4 + 2 * NH;                                       // state_dict order: weight, bias per layer
    // forward image
    static constexpr int OFF_1 = 0;                                                   // layer1: 4 PE chunks x 8 tiles
    static constexpr int off_h(int k) { return 4 * 8 * FRAG + k * 8 * 8 * FRAG; }     // layers_xyz[k]: 8 chunks x 8 tiles
    static constexpr int OFF_O = off_h(NH);                                           // fc_out: 8 chunks x 1 tile (rows 0..3)
    static constexpr int OFF_B = OFF_O + 8 * FRAG;                                    // biases: 128 per layer | 16
    static constexpr int bias(int l) { return OFF_B + 128 * l; }
    static constexpr int PACKED = OFF_B + 128 * (NH + 1) + 16;
    // saved activations, floats per point (section X of n points = [n][width] row-major at X * n)
    static constexpr int T_PE = 0;                                                    // 64, PE slot order (nfl::pe_slot_to_col)
    static constexpr int t_h(int k) { return 64 + 128 * k; }                          // k = 0: layer1's output as it is; k >= 1: post-ReLU
    static constexpr int SAVED_PER_POINT = 64 + 128 * (NH + 1);
    // backward: transposed image, dZ sections, slab of one point slice, gradient vector
    static constexpr int OFFT_O = 0;                                                  // fc_out^T: 1 chunk x 8 tiles
    static constexpr int offt_h(int k) { return 8 * FRAG + k * 8 * 8 * FRAG; }        // layers_xyz[k]^T
    static constexpr int PACKED_T = offt_h(NH);
    static constexpr int tz(int k) { return 128 * k; }                                // dZ of layer k's output (k = 0: layer1)
    static constexpr int DZ_PER_POINT = 128 * (NH + 1);
    static constexpr int g_wx(int k) { return k * (128 * 128 + 128); }                // layers_xyz[k].weight | .bias: the order of the gradient vector
    static constexpr int cs_x(int k) { return g_wx(k) + 128 * 128; }
    static constexpr int G_W1 = g_wx(NH);                                             // layer1.weight in slot order (128 x 64)
    static constexpr int CS_1 = G_W1 + 128 * 64;
    static constexpr int G_WO = CS_1 + 128;
    static constexpr int CS_O = G_WO + 4 * 128;
    static constexpr int SLAB = CS_O + 16;
    static constexpr int GRAD_FLOATS = 128 * 63 + 128 + NH * (128 * 128 + 128) + 4 * 128 + 4;
    static constexpr int N_JOBS = NH + 2;
};

#define NF_FLEX_DISPATCH(num_layers, ...)                   \
    switch (num_layers) {                                   \
        case 2: { constexpr int NH = 1; __VA_ARGS__; }      \
        case 3: { constexpr int NH = 2; __VA_ARGS__; }      \
        case 4: { constexpr int NH = 3; __VA_ARGS__; }      \
        case 5: { constexpr int NH = 4; __VA_ARGS__; }      \
        default: break;                                     \
    }

template <int NH>
static void nf_flex_table(std::vector<uint32_t>& t) {
    using L = NfFlex<NH>;
    t.assign(L::PACKED, 0xFF000000u);
    auto fill = [&](int off, int nk, int no_tiles, int tensor, int n_out, int n_cols, bool pe) {
        for (int ni = 0; ni < nk; ++ni)
            for (int no = 0; no < no_tiles; ++no)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int g = lane >> 4, i = lane & 15, n = 16 * no + i, slot = 16 * ni + 4 * g + r;
                        const int col = pe ? nfl::pe_slot_to_col(slot) : slot;
                        if (n < n_out && col >= 0)
                            t[(size_t)off + ((size_t)(ni * no_tiles + no) * 64 + lane) * 4 + r] = ((uint32_t)tensor << 24) | (uint32_t)(n * n_cols + col);
                    }
    };
    fill(L::OFF_1, 4, 8, 0, 128, 63, true);
    for (int k = 0; k < NH; ++k) fill(L::off_h(k), 8, 8, 2 + 2 * k, 128, 128, false);
    fill(L::OFF_O, 8, 1, 2 + 2 * NH, 4, 128, false);
    for (int l = 0; l <= NH; ++l)
        for (int n = 0; n < 128; ++n) t[L::bias(l) + n] = ((uint32_t)(1 + 2 * l) << 24) | (uint32_t)n;
    for (int n = 0; n < 4; ++n) t[L::bias(NH + 1) + n] = ((uint32_t)(3 + 2 * NH) << 24) | (uint32_t)n;
}

// block (ni, no), lane (g, i), r  ->  W[row = 16 ni + 4 g + r][16 no + i]   (A operand of the transposed product)
template <int NH>
static void nf_flex_table_t(std::vector<uint32_t>& t) {
    using L = NfFlex<NH>;
    t.assign(L::PACKED_T, 0xFF000000u);
    auto fill = [&](int off, int nk, int no_tiles, int tensor, int n_rows, int n_cols) {
        for (int ni = 0; ni < nk; ++ni)
            for (int no = 0; no < no_tiles; ++no)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int g = lane >> 4, i = lane & 15, row = 16 * ni + 4 * g + r, col = 16 * no + i;
                        if (row < n_rows) t[(size_t)off + ((size_t)(ni * no_tiles + no) * 64 + lane) * 4 + r] = ((uint32_t)tensor << 24) | (uint32_t)(row * n_cols + col);
                    }
    };
    fill(L::OFFT_O, 1, 8, 2 + 2 * NH, 4, 128);                                        // fc_out.weight (4, 128)
    for (int k = 0; k < NH; ++k) fill(L::offt_h(k), 8, 8, 2 + 2 * k, 128, 128);       // layers_xyz[k].weight (128, 128)
}

static NfPackTable g_flex_table[NF_FLEX_MAX_LAYERS], g_flex_table_t[NF_FLEX_MAX_LAYERS];

extern "C" size_t nf_flex_packed_floats(int num_layers) {
    NF_FLEX_DISPATCH(num_layers, return (size_t)NfFlex<NH>::PACKED)
    return 0;
}

extern "C" int nf_flex_pack(int num_layers, const float* const* params, float* packed, nf_stream_t stream) {
    NF_FLEX_DISPATCH(num_layers, return (nf_pack_f32<NfFlex<NH>::N_PARAMS, 10 + NH>(g_flex_table[NH], nf_flex_table<NH>, params, packed,
                                                                                     (int)NfFlex<NH>::PACKED, stream)))
    return NF_EINVAL;
}

// hidden layer K (layers_xyz[K]) and the boundary behind it; entry: its first weight chunk, bias and first B fragment are in `st`
template <int NT, int NH, bool SAVE, int K>
__device__ __forceinline__ void nf_flex_hidden(f32x4 (&acc)[NT][16], NfStream<NT>& st, const NfW& Wi, f32x4* act4, int lane,
                                               float* __restrict__ saved, int64_t p0, int64_t n_points) {
    using L = NfFlex<NH>;
    constexpr bool RELU_IN = K > 0;                              // layer1 has no activation (models.py:402): its output enters as it is
    f32x4 bj[NT];
    uint64_t m64[NT];                                            // not collected (the backward reads [h > 0] off the saved rows)
    if constexpr (SAVE) {
        NfCopyH<32, 4, RELU_IN> cs{act4, nf_slab_copy(saved, L::t_h(K), 128, p0, n_points), lane, 4, {}};
        cs.prime();
        nf_seg_lds<NT, 8, true, RELU_IN, false>(acc, st, Wi, L::off_h(K) / 4, 8, act4, lane, cs, m64);
    } else {
        nf_seg_lds<NT, 8, true, RELU_IN>(acc, st, Wi, L::off_h(K) / 4, 8, act4, lane);
    }
    nf_pending_b<NT, RELU_IN>(bj, st);
    if constexpr (K + 1 < NH) {
        nf_tail<NT, 8, 8, 8, 1>(acc, st.wb, bj, st, Wi, L::off_h(K + 1) / 4, Wi, L::bias(K + 2), act4, lane);
        nf_flex_hidden<NT, NH, SAVE, K + 1>(acc, st, Wi, act4, lane, saved, p0, n_points);
    } else {
        nf_tail<NT, 8, 8, 1, 1>(acc, st.wb, bj, st, Wi, L::OFF_O / 4, Wi, L::bias(NH + 1), act4, lane);
    }
}

template <int NT, int NH, bool SAVE>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_flex_mlp_fwd(const float* __restrict__ packed, const float* __restrict__ ro, const float* __restrict__ rd,
               const float* __restrict__ depth, int64_t n_points, int S, int depth_per_ray, float* __restrict__ raw,
               float* __restrict__ saved) {
    using L = NfFlex<NH>;
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    f32x4 pe[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
        const int64_t ray = p / S;
        const float zz = depth_per_ray ? depth[p] : depth[p - ray * S];      // (R, S) jittered depths or one shared (S) table
        const float px = nf_add(ro[ray * 3 + 0], nf_mul(rd[ray * 3 + 0], zz));
        const float py = nf_add(ro[ray * 3 + 1], nf_mul(rd[ray * 3 + 1], zz));
        const float pz = nf_add(ro[ray * 3 + 2], nf_mul(rd[ray * 3 + 2], zz));
        nf_encode_point(px, py, pz, g, pe[t]);
        if (SAVE && p0 + 16 * t + c < n_points) {                            // PE in slot order: chunk j, slots 16 j + 4 g .. +3
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(saved + (int64_t)L::T_PE * n_points + p * 64 + 16 * j + 4 * g) = pe[t][j];
        }
    }
    f32x4 acc[NT][16];
    NfStream<NT> st;
    f32x4 bj[NT];
    const NfW Wi = nf_w_image(packed, L::PACKED);
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
    nf_load_bias<8>(st.bias, Wi, L::bias(0), lane);
    {
        f32x4 w[16];
        nf_load_w16<8>(w, Wi, L::OFF_1 / 4, lane);
        NF_PE_B(0); nf_chunk<NT, 8, true>(acc, w, bj, st.bias);
        nf_load_w16<8>(w, Wi, L::OFF_1 / 4 + 1 * 8 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 8, false>(acc, w, bj, st.bias);
        nf_load_w16<8>(w, Wi, L::OFF_1 / 4 + 2 * 8 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 8, false>(acc, w, bj, st.bias);
        nf_load_w16<8>(w, Wi, L::OFF_1 / 4 + 3 * 8 * 64, lane);
        NF_PE_B(3); nf_tail<NT, 8, 8, 8, 1>(acc, w, bj, st, Wi, L::off_h(0) / 4, Wi, L::bias(1), act4, lane);
    }
#undef NF_PE_B
    nf_flex_hidden<NT, NH, SAVE, 0>(acc, st, Wi, act4, lane, saved, p0, n_points);
    // fc_out on the last hidden layer's output (ReLU where the slab is read)
    if constexpr (SAVE) {
        uint64_t m64[NT];
        NfCopyH<32, 4, true> cs{act4, nf_slab_copy(saved, L::t_h(NH), 128, p0, n_points), lane, 4, {}};
        cs.prime();
        nf_seg_lds<NT, 1, true, true, false>(acc, st, Wi, L::OFF_O / 4, 8, act4, lane, cs, m64);
    } else {
        nf_seg_lds<NT, 1, true, true>(acc, st, Wi, L::OFF_O / 4, 8, act4, lane);
    }
    nf_pending_b<NT, true>(bj, st);
    nf_chunk<NT, 1, false>(acc, st.wb, bj, st.bias);
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t p = p0 + 16 * t + c;
            if (p < n_points) reinterpret_cast<f32x4*>(raw)[p] = acc[t][0];
        }
    }
}

// depth: (n_rays, n_samples) when depth_per_ray != 0, else one (n_samples) table shared by all rays.
template <int NH>
static int nf_flex_fwd_impl(const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray, int64_t n_rays,
                            int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (n_rays == 0 && n_samples > 0) return 0;            // nothing to do (empty tensors have NULL data pointers)
    if (!packed || !ro || !rd || !depth || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    if (saved && n_points >= ((int64_t)1 << 22)) return NF_EINVAL;       // the save path addresses a 128-wide section with 32-bit byte offsets (512 B per point)
    if (saved)
        hipLaunchKernelGGL((k_flex_mlp_fwd<NT, NH, true>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, ro, rd, depth,
                           n_points, n_samples, depth_per_ray, raw, saved);
    else
        hipLaunchKernelGGL((k_flex_mlp_fwd<NT, NH, false>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, ro, rd, depth,
                           n_points, n_samples, depth_per_ray, raw, (float*)nullptr);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_flex_mlp_fwd(int num_layers, const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray,
                               int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    NF_FLEX_DISPATCH(num_layers, return nf_flex_fwd_impl<NH>(packed, ro, rd, depth, depth_per_ray, n_rays, n_samples, raw, nullptr, stream))
    return NF_EINVAL;
}

// Training forward: also writes PE, layer1's output and every hidden layer's post-ReLU output (NfFlex::t_h) for nf_flex_mlp_bwd.
extern "C" size_t nf_flex_saved_floats(int num_layers, int64_t n_points) {
    NF_FLEX_DISPATCH(num_layers, return (size_t)NfFlex<NH>::SAVED_PER_POINT * (size_t)(n_points > 0 ? n_points : 0))
    return 0;
}

extern "C" int nf_flex_mlp_fwd_train(int num_layers, const float* packed, const float* ro, const float* rd, const float* depth,
                                     int depth_per_ray, int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (!saved) return NF_EINVAL;
    NF_FLEX_DISPATCH(num_layers, return nf_flex_fwd_impl<NH>(packed, ro, rd, depth, depth_per_ray, n_rays, n_samples, raw, saved, stream))
    return NF_EINVAL;
}


// =================================================================================================
// Backward (autograd of FlexibleNeRFModel.forward models.py:396-422 behind the tiny path's compositing): exact f32.
//   h_0 = layer1(PE) (linear), h_k = relu(layers_xyz[k-1](h_{k-1})), out = fc_out(h_NH)
//   chain   dH_NH = fc_out^T d_raw; for k = NH .. 1: dZ_k = dH_k * [h_k > 0], dH_{k-1} = layers_xyz[k-1]^T dZ_k; dZ_0 = dH_0
//   dW      d fc_out = d_raw^T h_NH, d layers_xyz[k-1] = dZ_k^T h_{k-1}, d layer1 = dZ_0^T PE, db = column sums   (k_dw_gemm<3>)
//   reduce  deterministic per-slice slabs -> sum -> reference-layout tensors (PE slot order -> columns)
// =================================================================================================
extern "C" size_t nf_flex_packed_bwd_floats(int num_layers) {
    NF_FLEX_DISPATCH(num_layers, return (size_t)NfFlex<NH>::PACKED_T)
    return 0;
}

extern "C" int nf_flex_pack_bwd(int num_layers, const float* const* params, float* packed_t, nf_stream_t stream) {
    NF_FLEX_DISPATCH(num_layers, return (nf_pack_f32<NfFlex<NH>::N_PARAMS, 20 + NH>(g_flex_table_t[NH], nf_flex_table_t<NH>, params, packed_t,
                                                                                     (int)NfFlex<NH>::PACKED_T, stream)))
    return NF_EINVAL;
}

template <int NT, int NH>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_flex_bwd_chain(const float* __restrict__ packed_t, const float* __restrict__ saved, const float* __restrict__ d_raw, int64_t n_points,
                 float* __restrict__ dz) {
    using L = NfFlex<NH>;
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    const f32x4* WT = reinterpret_cast<const f32x4*>(packed_t);
    f32x4 frag[NT][1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int64_t p = p0 + 16 * t + c;
        frag[t][0] = (p < n_points && g == 0) ? reinterpret_cast<const f32x4*>(d_raw)[p] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 acc[NT][16];
    nf_zero_acc<NT, 8>(acc);
    nf_mma_from_regs<NT, 8, 1>(acc, WT + L::OFFT_O / 4, frag, lane);
#pragma unroll
    for (int k = NH; k >= 1; --k) {
        nf_mask_by_saved<NT, 8>(acc, saved + (int64_t)L::t_h(k) * n_points, 128, p0, n_points, lane);
        nf_store_global<NT, 8>(acc, dz + (int64_t)L::tz(k) * n_points, 128, p0, n_points, lane);
        nf_store_act<NT, 8, false>(acc, act4, lane);
        nf_zero_acc<NT, 8>(acc);
        nf_mma_from_lds<NT, 8>(acc, WT + L::offt_h(k - 1) / 4, 8, act4, lane);
    }
    nf_store_global<NT, 8>(acc, dz + (int64_t)L::tz(0) * n_points, 128, p0, n_points, lane);         // layer1 is linear: dZ_0 = dH_0
}

template <int NH>
static void nf_flex_dw_jobs(NfDwJob* j) {
    using L = NfFlex<NH>;
    //                a_kind a_sec   lda a_col0 n_valid  b_sec       ldb b_col0 k_valid  out_off   ldo  cs_off
    for (int k = 1; k <= NH; ++k)
        j[k - 1] = {0, L::tz(k), 128, 0, 128, L::t_h(k - 1), 128, 0, 128, L::g_wx(k - 1), 128, L::cs_x(k - 1)};   // d layers_xyz[k-1] = dZ_k^T h_{k-1}
    j[NH] = {0, L::tz(0), 128, 0, 128, L::T_PE, 64, 0, 64, L::G_W1, 64, L::CS_1};                                // d layer1 = dZ_0^T PE (slot order)
    j[NH + 1] = {1, 0, 4, 0, 4, L::t_h(NH), 128, 0, 128, L::G_WO, 128, L::CS_O};                                 // d fc_out = d_raw^T h_NH
}

static NfDwJobTable g_flex_jobs[NF_FLEX_MAX_LAYERS];

// sum (slab layout) -> [layer1.weight (128,63) | layer1.bias | layers_xyz.k.weight | layers_xyz.k.bias ... | fc_out.weight (4,128) | fc_out.bias]
template <int NH>
__global__ void __launch_bounds__(256) k_flex_grad_unpack(const float* __restrict__ sum, float* __restrict__ grads) {
    using L = NfFlex<NH>;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < L::GRAD_FLOATS; e += gridDim.x * blockDim.x) {
        int local = e;
        float v;
        if (local < 128 * 63) { const int n = local / 63, col = local - 63 * n; v = sum[L::G_W1 + n * 64 + nfl::pe_col_to_slot(col)]; }
        else if ((local -= 128 * 63) < 128) v = sum[L::CS_1 + local];
        else if ((local -= 128) < L::G_W1) v = sum[local];                    // the hidden layers sit in the slab in the vector's own order
        else if ((local -= L::G_W1) < 512) v = sum[L::G_WO + local];
        else v = sum[L::CS_O + (local - 512)];
        grads[e] = v;
    }
}

extern "C" size_t nf_flex_grad_floats(int num_layers) {
    NF_FLEX_DISPATCH(num_layers, return (size_t)NfFlex<NH>::GRAD_FLOATS)
    return 0;
}

// point slices of the weight-gradient kernel: as the tiny path's (nf_tiny.hip) -- one slice per CU, at least 256 points
static inline void nf_flex_bwd_plan(int64_t n_points, int64_t* pts_per_slice, int* n_slices) {
    int64_t pps = (n_points + 255) / 256;
    pps = (pps + 15) / 16 * 16;
    if (pps < 256) pps = 256;
    *pts_per_slice = pps;
    *n_slices = (int)((n_points + pps - 1) / pps);
}

extern "C" size_t nf_flex_bwd_workspace_floats(int num_layers, int64_t n_points) {
    if (n_points <= 0) return 0;
    int64_t pps; int ns;
    nf_flex_bwd_plan(n_points, &pps, &ns);
    NF_FLEX_DISPATCH(num_layers, return (size_t)NfFlex<NH>::DZ_PER_POINT * (size_t)n_points + (size_t)(ns + 1) * NfFlex<NH>::SLAB)
    return 0;
}

template <int NH>
static int nf_flex_bwd_impl(const float* packed_t, const float* saved, const float* d_raw, int64_t n_rays, int n_samples, float* workspace,
                            size_t workspace_floats, float* grads, nf_stream_t stream) {
    using L = NfFlex<NH>;
    if (!packed_t || !saved || !d_raw || !workspace || !grads || n_rays <= 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (workspace_floats < nf_flex_bwd_workspace_floats(NH + 1, n_points)) return NF_EINVAL;
    const NfDwJob* jobs = nullptr;
    const int rcj = g_flex_jobs[NH].get(L::N_JOBS, nf_flex_dw_jobs<NH>, &jobs);
    if (rcj) return rcj;
    int64_t pps; int ns;
    nf_flex_bwd_plan(n_points, &pps, &ns);
    float* dz = workspace;
    float* slabs = workspace + (size_t)L::DZ_PER_POINT * n_points;
    float* sum = slabs + (size_t)ns * L::SLAB;
    hipStream_t s = nf_s(stream);
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipError_t e = hipMemsetAsync(slabs, 0, (size_t)ns * L::SLAB * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_flex_bwd_chain<NT, NH>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, s, packed_t, saved, d_raw, n_points, dz);
    hipLaunchKernelGGL((k_dw_gemm<3>), dim3((L::N_JOBS + 3) / 4, ns), dim3(256), 0, s, jobs, (int)L::N_JOBS, (int)L::SLAB, dz, d_raw, saved,
                       n_points, pps, slabs);
    hipLaunchKernelGGL((k_grad_reduce<3>), dim3(64), dim3(256), 0, s, slabs, ns, (int)L::SLAB, sum, NfReduceAlt{});
    hipLaunchKernelGGL((k_flex_grad_unpack<NH>), dim3(64), dim3(256), 0, s, sum, grads);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_flex_mlp_bwd(int num_layers, const float* packed_t, const float* saved, const float* d_raw, int64_t n_rays, int n_samples,
                               float* workspace, size_t workspace_floats, float* grads, nf_stream_t stream) {
    NF_FLEX_DISPATCH(num_layers, return nf_flex_bwd_impl<NH>(packed_t, saved, d_raw, n_rays, n_samples, workspace, workspace_floats, grads, stream))
    return NF_EINVAL;
}

// host-only self-test of the job table (tests/test_host.py): every slab entry written exactly once
extern "C" int nf_selftest_dw_tables_flex(int num_layers) {
    NF_FLEX_DISPATCH(num_layers, {
        NfDwJob jobs[NfFlex<NH>::N_JOBS];
        nf_flex_dw_jobs<NH>(jobs);
        return nf_check_dw_jobs(jobs, NfFlex<NH>::N_JOBS, NfFlex<NH>::SLAB, (long)NH * (128L * 128 + 128) + 128L * 64 + 128 + 4L * 128 + 4);
    })
    return NF_EINVAL;
}
