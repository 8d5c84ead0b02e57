// ConditionalBlendshapePaperNeRFModel.forward on PRE-ENCODED inputs (reference nerf/models.py:236-261 as called by
// run_network, nerf/train_utils.py:20-24): x (P, 87) = [PE10(xyz) (63) | PE4(dirs) (24)] -> (P, 4).  Inference only; the
// hot path (run_one_iter_of_nerf) never materialises x and uses nf_paper_mlp_fwd instead.  Own translation unit on purpose.
#include "nf_mlp_dev.h"

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
    const f32x4* W = reinterpret_cast<const f32x4*>(packed);
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
    f32x4 acc[NT][16];
    nf_init_acc<NT, 16>(acc, cond + B_L0, lane);
    nf_mma_from_regs<NT, 16, 4>(acc, W + OFF_L0 / 4, pe, lane);
    nf_store_act<NT, 16, true>(acc, act4, lane);
    nf_init_acc<NT, 16>(acc, cond + B_L1, lane);
    nf_mma_from_lds<NT, 16>(acc, W + OFF_L1 / 4, 16, act4, lane);
    nf_store_act<NT, 16, true>(acc, act4, lane);
    nf_init_acc<NT, 16>(acc, cond + B_L2, lane);
    nf_mma_from_lds<NT, 16>(acc, W + OFF_L2 / 4, 16, act4, lane);
    nf_store_act<NT, 16, true>(acc, act4, lane);
    nf_init_acc<NT, 16>(acc, cond + B_L3, lane);
    nf_mma_from_regs<NT, 16, 4>(acc, W + OFF_L3 / 4, pe, lane);
    nf_mma_from_lds<NT, 16>(acc, W + OFF_L3 / 4 + 4 * 16 * 64, 16, act4, lane);
    nf_store_act<NT, 16, true>(acc, act4, lane);
    nf_init_acc<NT, 16>(acc, cond + B_L4, lane);
    nf_mma_from_lds<NT, 16>(acc, W + OFF_L4 / 4, 16, act4, lane);
    nf_store_act<NT, 16, true>(acc, act4, lane);
    nf_init_acc<NT, 16>(acc, cond + B_L5, lane);
    nf_mma_from_lds<NT, 16>(acc, W + OFF_L5 / 4, 16, act4, lane);
    nf_store_act<NT, 16, true>(acc, act4, lane);
    nf_init_acc<NT, 16>(acc, cond + B_FEAT, lane);
    nf_mma_from_lds<NT, 16>(acc, W + OFF_FEAT / 4, 16, act4, lane);
    nf_store_act<NT, 16, false>(acc, act4, lane);
    nf_init_acc<NT, 9>(acc, cond + B_D0, lane);
    nf_mma_from_lds<NT, 9>(acc, W + OFF_D0E / 4, 16, act4, lane);
    nf_mma_from_regs<NT, 9, 2>(acc, W + OFF_D0E / 4 + 16 * 9 * 64, dirf, lane);
    float sigma_raw[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sigma_raw[t] = acc[t][8].x;
    nf_store_act<NT, 8, true>(acc, act4, lane);
    nf_init_acc<NT, 8>(acc, cond + B_D1, lane);
    nf_mma_from_lds<NT, 8>(acc, W + OFF_D1 / 4, 8, act4, lane);
    nf_store_act<NT, 8, true>(acc, act4, lane);
    nf_init_acc<NT, 8>(acc, cond + B_D2, lane);
    nf_mma_from_lds<NT, 8>(acc, W + OFF_D2 / 4, 8, act4, lane);
    nf_store_act<NT, 8, true>(acc, act4, lane);
    nf_init_acc<NT, 1>(acc, cond + B_RGB, lane);
    nf_mma_from_lds<NT, 1>(acc, W + OFF_RGB / 4, 8, act4, lane);
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
    if (!packed || !x87 || !expr76 || !latent32 || !cond || !out || n_points < 0) return NF_EINVAL;
    if (n_points == 0) return 0;
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
