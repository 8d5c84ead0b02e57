// BASELINE config 1: tiny_nerf.py's VeryTinyNerfModel (reference tiny_nerf.py:162-181) fused with its input assembly
// (compute_query_points_from_rays tiny_nerf.py:59-63 + positional_encoding(., 10), tiny_nerf.py:140):
//   pts = ro + rd * depth -> PE(63) -> Linear 128 + ReLU -> Linear 128 + ReLU -> Linear 4.
// A third instantiation of the building blocks of nf_mlp_dev.h (exact-f32 MFMA, wave = 32 points, no barriers).
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"
#include "nf_mlp_stream.h"
#include "nf_pack.h"

namespace nft {
constexpr int FRAG = 256;
constexpr int OFF_1 = 0;                         // layer1: 4 PE chunks x 8 tiles
constexpr int OFF_2 = OFF_1 + 4 * 8 * FRAG;      // layer2: 8 chunks x 8 tiles
constexpr int OFF_3 = OFF_2 + 8 * 8 * FRAG;      // layer3: 8 chunks x 1 tile (rows 0..3)
constexpr int OFF_B = OFF_3 + 8 * 1 * FRAG;      // biases: 128 | 128 | 16
constexpr int PACKED = OFF_B + 128 + 128 + 16;
// training: activations saved by the forward, floats per point (section X of n points = [n][width] row-major at X * n)
constexpr int T_PE = 0;                          // 64, PE slot order (nfl::pe_slot_to_col)
constexpr int T_H1 = 64, T_H2 = 192;             // 128 each, post-ReLU
constexpr int SAVED_PER_POINT = 320;
}  // namespace nft


static void nf_tiny_table(std::vector<uint32_t>& t) {
    using namespace nft;
    const uint32_t Z = 0xFF000000u;
    t.assign(PACKED, Z);
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
    fill(OFF_1, 4, 8, 0, 128, 63, true);
    fill(OFF_2, 8, 8, 2, 128, 128, false);
    fill(OFF_3, 8, 1, 4, 4, 128, false);
    for (int n = 0; n < 128; ++n) { t[OFF_B + n] = (1u << 24) | n; t[OFF_B + 128 + n] = (3u << 24) | n; }
    for (int n = 0; n < 4; ++n) t[OFF_B + 256 + n] = (5u << 24) | n;
}

static NfPackTable g_tiny_table;

extern "C" size_t nf_tiny_packed_floats(void) { return (size_t)nft::PACKED; }

extern "C" int nf_tiny_pack(const float* const* params, float* packed, nf_stream_t stream) {
    return nf_pack_f32<6, 8>(g_tiny_table, nf_tiny_table, params, packed, (int)nft::PACKED, stream);
}

template <int NT, bool SAVE>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_tiny_mlp_fwd(const float* __restrict__ packed, const float* __restrict__ ro, const float* __restrict__ rd,
               const float* __restrict__ depth, int64_t n_points, int S, int depth_per_ray, float* __restrict__ raw,
               float* __restrict__ saved) {
    using namespace nft;
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
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(saved + (int64_t)T_PE * n_points + p * 64 + 16 * j + 4 * g) = pe[t][j];
        }
    }
    // Layer-streamed like the paper model's kernels (nf_mlp_stream.h): the bias is the C operand of a layer's first MFMAs, the raw
    // accumulators go to the slab under the last K chunk, the ReLU is applied where the slab is read; SAVE: the post-ReLU copy of the slab
    // to `saved` rides in the loop that consumes it (NfCopyH: 128-wide rows, two per instruction).  The biases live in the weight image.
    f32x4 acc[NT][16];
    NfStream<NT> st;
    f32x4 bj[NT];
    uint64_t m64[NT];                                            // not collected here (the backward reads [h > 0] off the saved rows)
    const NfW Wi = nf_w_image(packed, PACKED);
#define NF_PE_B(J_) do { _Pragma("unroll") for (int t = 0; t < NT; ++t) bj[t] = pe[t][J_]; } while (0)
    nf_load_bias<8>(st.bias, Wi, OFF_B, lane);
    {
        f32x4 w[16];
        nf_load_w16<8>(w, Wi, OFF_1 / 4, lane);
        NF_PE_B(0); nf_chunk<NT, 8, true>(acc, w, bj, st.bias);
        nf_load_w16<8>(w, Wi, OFF_1 / 4 + 1 * 8 * 64, lane);
        NF_PE_B(1); nf_chunk<NT, 8, false>(acc, w, bj, st.bias);
        nf_load_w16<8>(w, Wi, OFF_1 / 4 + 2 * 8 * 64, lane);
        NF_PE_B(2); nf_chunk<NT, 8, false>(acc, w, bj, st.bias);
        nf_load_w16<8>(w, Wi, OFF_1 / 4 + 3 * 8 * 64, lane);
        NF_PE_B(3); nf_tail<NT, 8, 8, 8, 1>(acc, w, bj, st, Wi, OFF_2 / 4, Wi, OFF_B + 128, act4, lane);
    }
#undef NF_PE_B
    if constexpr (SAVE) {
        NfCopyH<32, 4, true> cs{act4, nf_slab_copy(saved, T_H1, 128, p0, n_points), lane, 4, {}};
        cs.prime();
        nf_seg_lds<NT, 8, true, true, false>(acc, st, Wi, OFF_2 / 4, 8, act4, lane, cs, m64);
    } else {
        nf_seg_lds<NT, 8, true, true>(acc, st, Wi, OFF_2 / 4, 8, act4, lane);
    }
    nf_pending_b<NT, true>(bj, st);
    nf_tail<NT, 8, 8, 1, 1>(acc, st.wb, bj, st, Wi, OFF_3 / 4, Wi, OFF_B + 256, act4, lane);
    if constexpr (SAVE) {
        NfCopyH<32, 4, true> cs{act4, nf_slab_copy(saved, T_H2, 128, p0, n_points), lane, 4, {}};
        cs.prime();
        nf_seg_lds<NT, 1, true, true, false>(acc, st, Wi, OFF_3 / 4, 8, act4, lane, cs, m64);
    } else {
        nf_seg_lds<NT, 1, true, true>(acc, st, Wi, OFF_3 / 4, 8, act4, lane);
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
static int nf_tiny_fwd_impl(const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray, int64_t n_rays,
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
        hipLaunchKernelGGL((k_tiny_mlp_fwd<NT, true>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, ro, rd, depth,
                           n_points, n_samples, depth_per_ray, raw, saved);
    else
        hipLaunchKernelGGL((k_tiny_mlp_fwd<NT, false>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, ro, rd, depth,
                           n_points, n_samples, depth_per_ray, raw, (float*)nullptr);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_tiny_mlp_fwd(const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray,
                               int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    return nf_tiny_fwd_impl(packed, ro, rd, depth, depth_per_ray, n_rays, n_samples, raw, nullptr, stream);
}

// Training forward: also writes PE, h1, h2 (nft::T_*; nf_tiny_saved_floats(n_points) floats) for nf_tiny_mlp_bwd.
extern "C" size_t nf_tiny_saved_floats(int64_t n_points) { return (size_t)nft::SAVED_PER_POINT * (size_t)(n_points > 0 ? n_points : 0); }

extern "C" int nf_tiny_mlp_fwd_train(const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray,
                                     int64_t n_rays, int n_samples, float* raw, float* saved, nf_stream_t stream) {
    if (!saved) return NF_EINVAL;
    return nf_tiny_fwd_impl(packed, ro, rd, depth, depth_per_ray, n_rays, n_samples, raw, saved, stream);
}


// =================================================================================================
// Backward of the tiny path (autograd of tiny_nerf.py:111-159 + the trainer's rgb MSE, tiny_nerf.py:291-302): exact f32.
//   chain   dZ3 = d_raw; dZ2 = (W3^T dZ3) * [h2 > 0]; dZ1 = (W2^T dZ2) * [h1 > 0]        (k_tiny_bwd_chain, wave = 32 points)
//   dW      dW3 = dZ3^T h2, dW2 = dZ2^T h1, dW1 = dZ1^T PE, db = column sums of dZ       (k_dw_gemm<2>: three jobs, nf_mlp_dw.h)
//   reduce  deterministic per-slice slabs -> sum -> reference-layout tensors (PE slot order -> columns)
// =================================================================================================
#include "nf_mlp_dw.h"

namespace nft {
constexpr int OFFT_3 = 0;                         // W3^T: 1 chunk (slots 0..3 = d r, d g, d b, d sigma) x 8 tiles of 16 h2 features
constexpr int OFFT_2 = OFFT_3 + 1 * 8 * FRAG;     // W2^T: 8 chunks x 8 tiles
constexpr int PACKED_T = OFFT_2 + 8 * 8 * FRAG;
constexpr int TZ_2 = 0, TZ_1 = 128, DZ_PER_POINT = 256;
constexpr int G_W2 = 0, CS_2 = 16384, G_W1 = 16512, CS_1 = G_W1 + 128 * 64, G_W3 = CS_1 + 128, CS_3 = G_W3 + 512;
constexpr int SLAB = CS_3 + 16;                   // 25360 floats (multiple of 4)
constexpr int GRAD_FLOATS = 128 * 63 + 128 + 128 * 128 + 128 + 4 * 128 + 4;
constexpr int N_JOBS = 3;
}  // namespace nft

// block (ni, no), lane (g, i), r  ->  W[row = 16 ni + 4 g + r][16 no + i]   (A operand of the transposed product)
static void nf_tiny_table_t(std::vector<uint32_t>& t) {
    using namespace nft;
    t.assign(PACKED_T, 0xFF000000u);
    auto fill = [&](int off, int nk, int no_tiles, int tensor, int n_rows, int n_cols) {
        for (int ni = 0; ni < nk; ++ni)
            for (int no = 0; no < no_tiles; ++no)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int g = lane >> 4, i = lane & 15, row = 16 * ni + 4 * g + r, col = 16 * no + i;
                        if (row < n_rows) t[(size_t)off + ((size_t)(ni * no_tiles + no) * 64 + lane) * 4 + r] = ((uint32_t)tensor << 24) | (uint32_t)(row * n_cols + col);
                    }
    };
    fill(OFFT_3, 1, 8, 4, 4, 128);                // layer3.weight (4, 128)
    fill(OFFT_2, 8, 8, 2, 128, 128);              // layer2.weight (128, 128)
}

static NfPackTable g_tiny_table_t;

extern "C" size_t nf_tiny_packed_bwd_floats(void) { return (size_t)nft::PACKED_T; }

extern "C" int nf_tiny_pack_bwd(const float* const* params, float* packed_t, nf_stream_t stream) {
    return nf_pack_f32<6, 9>(g_tiny_table_t, nf_tiny_table_t, params, packed_t, (int)nft::PACKED_T, stream);
}

template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_tiny_bwd_chain(const float* __restrict__ packed_t, const float* __restrict__ saved, const float* __restrict__ d_raw, int64_t n_points,
                 float* __restrict__ dz) {
    using namespace nft;
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
    nf_mma_from_regs<NT, 8, 1>(acc, WT + OFFT_3 / 4, frag, lane);
    nf_mask_by_saved<NT, 8>(acc, saved + (int64_t)T_H2 * n_points, 128, p0, n_points, lane);
    nf_store_global<NT, 8>(acc, dz + (int64_t)TZ_2 * n_points, 128, p0, n_points, lane);
    nf_store_act<NT, 8, false>(acc, act4, lane);
    nf_zero_acc<NT, 8>(acc);
    nf_mma_from_lds<NT, 8>(acc, WT + OFFT_2 / 4, 8, act4, lane);
    nf_mask_by_saved<NT, 8>(acc, saved + (int64_t)T_H1 * n_points, 128, p0, n_points, lane);
    nf_store_global<NT, 8>(acc, dz + (int64_t)TZ_1 * n_points, 128, p0, n_points, lane);
}

static void nf_tiny_dw_jobs(NfDwJob* j) {
    using namespace nft;
    //      a_kind a_sec lda a_col0 n_valid  b_sec  ldb b_col0 k_valid  out_off ldo  cs_off
    j[0] = {0, TZ_2, 128, 0, 128, T_H1, 128, 0, 128, G_W2, 128, CS_2};      // dW2 = dZ2^T h1, db2
    j[1] = {0, TZ_1, 128, 0, 128, T_PE, 64, 0, 64, G_W1, 64, CS_1};         // dW1 = dZ1^T PE (slot order), db1
    j[2] = {1, 0, 4, 0, 4, T_H2, 128, 0, 128, G_W3, 128, CS_3};             // dW3 = d_raw^T h2, db3
}

static NfDwJobTable g_tiny_jobs;

// sum (slab layout) -> [layer1.weight (128,63) | layer1.bias | layer2.weight | layer2.bias | layer3.weight (4,128) | layer3.bias]
__global__ void __launch_bounds__(256) k_tiny_grad_unpack(const float* __restrict__ sum, float* __restrict__ grads) {
    using namespace nft;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < GRAD_FLOATS; e += gridDim.x * blockDim.x) {
        int local = e;
        float v;
        if (local < 128 * 63) { const int n = local / 63, col = local - 63 * n; v = sum[G_W1 + n * 64 + nfl::pe_col_to_slot(col)]; }
        else if ((local -= 128 * 63) < 128) v = sum[CS_1 + local];
        else if ((local -= 128) < 128 * 128) v = sum[G_W2 + local];
        else if ((local -= 128 * 128) < 128) v = sum[CS_2 + local];
        else if ((local -= 128) < 512) v = sum[G_W3 + local];
        else v = sum[CS_3 + (local - 512)];
        grads[e] = v;
    }
}

extern "C" size_t nf_tiny_grad_floats(void) { return (size_t)nft::GRAD_FLOATS; }

// point slices of the weight-gradient kernel: the three jobs of the tiny model fit ONE workgroup (a wave per job), so the slice
// count is the grid -- one workgroup per CU (256 slices of >= 256 points) instead of the 28 slices the big models' plan gives
// (28 workgroups on 256 CUs: 1.38 ms for a 64 x 64 x 32 image, most of a training iteration)
static inline void nf_tiny_bwd_plan(int64_t n_points, int64_t* pts_per_slice, int* n_slices) {
    int64_t pps = (n_points + 255) / 256;
    pps = (pps + 15) / 16 * 16;
    if (pps < 256) pps = 256;
    *pts_per_slice = pps;
    *n_slices = (int)((n_points + pps - 1) / pps);
}

extern "C" size_t nf_tiny_bwd_workspace_floats(int64_t n_points) {
    int64_t pps; int ns;
    nf_tiny_bwd_plan(n_points, &pps, &ns);
    return (size_t)nft::DZ_PER_POINT * (size_t)n_points + (size_t)(ns + 1) * nft::SLAB;
}

extern "C" int nf_tiny_mlp_bwd(const float* packed_t, const float* saved, const float* d_raw, int64_t n_rays, int n_samples,
                               float* workspace, size_t workspace_floats, float* grads, nf_stream_t stream) {
    using namespace nft;
    if (!packed_t || !saved || !d_raw || !workspace || !grads || n_rays <= 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (workspace_floats < nf_tiny_bwd_workspace_floats(n_points)) return NF_EINVAL;
    const NfDwJob* jobs = nullptr;
    const int rcj = g_tiny_jobs.get(N_JOBS, nf_tiny_dw_jobs, &jobs);
    if (rcj) return rcj;
    int64_t pps; int ns;
    nf_tiny_bwd_plan(n_points, &pps, &ns);
    float* dz = workspace;
    float* slabs = workspace + (size_t)DZ_PER_POINT * n_points;
    float* sum = slabs + (size_t)ns * SLAB;
    hipStream_t s = nf_s(stream);
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipError_t e = hipMemsetAsync(slabs, 0, (size_t)ns * SLAB * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_tiny_bwd_chain<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, s, packed_t, saved, d_raw, n_points, dz);
    hipLaunchKernelGGL((k_dw_gemm<2>), dim3((N_JOBS + 3) / 4, ns), dim3(256), 0, s, jobs, (int)N_JOBS, (int)SLAB, dz, d_raw, saved, n_points,
                       pps, slabs);
    hipLaunchKernelGGL((k_grad_reduce<2>), dim3(64), dim3(256), 0, s, slabs, ns, (int)SLAB, sum, NfReduceAlt{});
    hipLaunchKernelGGL(k_tiny_grad_unpack, dim3(64), dim3(256), 0, s, sum, grads);
    NF_RETURN_LAUNCH();
}

// host-only self-test of the tiny job table (tests/test_host.py)
extern "C" int nf_selftest_dw_tables_tiny(void) {
    NfDwJob jobs[nft::N_JOBS];
    nf_tiny_dw_jobs(jobs);
    return nf_check_dw_jobs(jobs, nft::N_JOBS, nft::SLAB, 128L * 128 + 128 + 128L * 64 + 128 + 4L * 128 + 4);
}
