// BASELINE config 1: tiny_nerf.py's VeryTinyNerfModel (reference tiny_nerf.py:162-181) fused with its input assembly
// (compute_query_points_from_rays tiny_nerf.py:59-63 + positional_encoding(., 10), tiny_nerf.py:140):
//   pts = ro + rd * depth -> PE(63) -> Linear 128 + ReLU -> Linear 128 + ReLU -> Linear 4.
// A third instantiation of the building blocks of nf_mlp_dev.h (exact-f32 MFMA, wave = 32 points, no barriers).
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"
#include "nf_pack.h"

namespace nft {
constexpr int FRAG = 256;
constexpr int OFF_1 = 0;                         // layer1: 4 PE chunks x 8 tiles
constexpr int OFF_2 = OFF_1 + 4 * 8 * FRAG;      // layer2: 8 chunks x 8 tiles
constexpr int OFF_3 = OFF_2 + 8 * 8 * FRAG;      // layer3: 8 chunks x 1 tile (rows 0..3)
constexpr int OFF_B = OFF_3 + 8 * 1 * FRAG;      // biases: 128 | 128 | 16
constexpr int PACKED = OFF_B + 128 + 128 + 16;
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

template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_tiny_mlp_fwd(const float* __restrict__ packed, const float* __restrict__ ro, const float* __restrict__ rd,
               const float* __restrict__ depth, int64_t n_points, int S, int depth_per_ray, float* __restrict__ raw) {
    using namespace nft;
    __shared__ __attribute__((aligned(16))) f32x4 lds[NF_MLP_WAVES * 16 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * NF_MLP_WAVES + wave) * (16 * NT);
    if (p0 >= n_points) return;
    f32x4* act4 = lds + wave * (16 * NT * 64);
    const f32x4* W = reinterpret_cast<const f32x4*>(packed);
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
    }
    f32x4 acc[NT][16];
    nf_init_acc<NT, 8>(acc, packed + OFF_B, lane);
    nf_mma_from_regs<NT, 8, 4>(acc, W + OFF_1 / 4, pe, lane);
    nf_relu_inplace<NT, 8>(acc);
    nf_store_act<NT, 8, false>(acc, act4, lane);
    nf_init_acc<NT, 8>(acc, packed + OFF_B + 128, lane);
    nf_mma_from_lds<NT, 8>(acc, W + OFF_2 / 4, 8, act4, lane);
    nf_relu_inplace<NT, 8>(acc);
    nf_store_act<NT, 8, false>(acc, act4, lane);
    nf_init_acc<NT, 1>(acc, packed + OFF_B + 256, lane);
    nf_mma_from_lds<NT, 1>(acc, W + OFF_3 / 4, 8, act4, lane);
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t p = p0 + 16 * t + c;
            if (p < n_points) reinterpret_cast<f32x4*>(raw)[p] = acc[t][0];
        }
    }
}

// depth: (n_rays, n_samples) when depth_per_ray != 0, else one (n_samples) table shared by all rays.
extern "C" int nf_tiny_mlp_fwd(const float* packed, const float* ro, const float* rd, const float* depth, int depth_per_ray,
                               int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    if (!packed || !ro || !rd || !depth || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    constexpr int NT = NF_MLP_NT;
    const int64_t per_block = (int64_t)NF_MLP_WAVES * 16 * NT;
    const int64_t grid = (n_points + per_block - 1) / per_block;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL((k_tiny_mlp_fwd<NT>), dim3((unsigned)grid), dim3(64 * NF_MLP_WAVES), 0, nf_s(stream), packed, ro, rd, depth,
                       n_points, n_samples, depth_per_ray, raw);
    NF_RETURN_LAUNCH();
}
