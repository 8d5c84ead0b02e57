// Second model family (ConditionalBlendshapeLearnableCodeNeRFModel, reference nerf/models.py:529-636), inference forward on
// the bf16 matrix pipe at fp32-class accuracy: the split-bf16 scheme and the machinery of nf_mlp_bf16.hip (3 bf16 MFMAs per
// product, K order chosen so that a layer's D registers are the next layer's B operands, weight stream L2 -> LDS through a
// 4-deep LDS-DMA ring) instantiated for this family's layer table:
//   0 layer1 (PE 4 k-steps -> 256, no activation)   1..3 layers_xyz.0..2 (ReLU)   4 fc_alpha (reads x: 256 -> 1)
//   5 fc_feat (ReLU)   6 layers_dir.0 ([feat | dir k-step | pad] -> 128, ReLU)   7 fc_rgb (128 -> 3)
// Same per-call bias table (`cond`, nf_lcode_condition) and packed-weight source tensors as the exact-f32 kernel
// (nf_mlp_lcode.hip); the training path of this family stays exact f32.
#include <vector>
#include <mutex>
#include "nf_common.h"
#include "nf_mlp_lcode_layout.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace nfb {
constexpr int NL = 8;
constexpr int KS[NL] = {4, 16, 16, 16, 16, 16, 20, 8};
constexpr int NO[NL] = {8, 8, 8, 8, 1, 8, 4, 1};
constexpr int pair_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += KS[i] * NO[i]; return o; }
constexpr int N_PAIRS = pair_off(NL);
constexpr int STREAM_BF16 = N_PAIRS * 2 * 512;
// slot (s, h, j) of a hidden input -> feature index (D register order of the producing layer); PE / dir slots as in
// nf_mlp_bf16_common.h (the kernels share the prologue)
__host__ __device__ constexpr int hid_feature(int s, int h, int j) { return 16 * s + 4 * h + (j & 3) + 8 * (j >> 2); }
__host__ __device__ constexpr int pe_col(int s, int h, int j) {
    const int p = 16 * h + 4 * s + (j >> 1), sc = j & 1;
    if (p < 30) return 3 + 6 * (p / 3) + 3 * sc + (p % 3);
    if (p == 30) return sc;
    return sc == 0 ? 2 : -1;
}
__host__ __device__ constexpr int dir_col(int h, int j) { return j < 4 ? 256 + 6 * (2 * h + (j >> 1)) + 3 * (j & 1) : -1; }
}  // namespace nfb

#include "nf_mlp_bf16_machinery.inc"

// =================================================================================================
// pack: fp32 parameters (nerf.models.LCODE_KEYS order) -> (hi, lo) bf16 fragment stream
// =================================================================================================
struct NfLcodePtrsH { const float* p[nlc::NPARAMS]; };

static void nf_lcode_table_bf16(std::vector<uint32_t>& t) {
    using namespace nfb;
    const uint32_t Z = 0xFF000000u;
    t.assign((size_t)N_PAIRS * 512, Z);
    auto code = [](int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); };
    for (int l = 0; l < NL; ++l)
        for (int s = 0; s < KS[l]; ++s)
            for (int nt = 0; nt < NO[l]; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int h = lane >> 5, i = lane & 31, n = 32 * nt + i;
                        uint32_t c = Z;
                        switch (l) {
                            case 0: { const int col = pe_col(s, h, j); if (col >= 0) c = code(0, n, col, 171); } break;
                            case 1: c = code(2, n, hid_feature(s, h, j), 256); break;
                            case 2: c = code(4, n, hid_feature(s, h, j), 256); break;
                            case 3: c = code(6, n, hid_feature(s, h, j), 256); break;
                            case 4: if (n == 0) c = code(10, 0, hid_feature(s, h, j), 256); break;
                            case 5: c = code(14, n, hid_feature(s, h, j), 256); break;
                            case 6:
                                if (s < 16) c = code(8, n, hid_feature(s, h, j), 280);
                                else if (s == 16) { const int col = dir_col(h, j); if (col >= 0) c = code(8, n, col, 280); }
                                break;
                            case 7: if (n < 3) c = code(12, n, hid_feature(s, h, j), 128); break;
                        }
                        t[((size_t)(pair_off(l) + s * NO[l] + nt)) * 512 + lane * 8 + j] = c;
                    }
}

__global__ void __launch_bounds__(256) k_lcode_pack_bf16(NfLcodePtrsH ptrs, const uint32_t* __restrict__ table, __bf16* __restrict__ stream,
                                                         int n_entries) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const uint32_t code = table[e], id = code >> 24;
        const float w = id == 0xFFu ? 0.0f : ptrs.p[id][code & 0xFFFFFFu];
        const __bf16 hi = (__bf16)w;
        const __bf16 lo = (__bf16)(w - (float)hi);
        const int pair = e >> 9, within = e & 511;
        stream[(size_t)(2 * pair) * 512 + within] = hi;
        stream[(size_t)(2 * pair + 1) * 512 + within] = lo;
    }
}

static std::mutex g_lcode_b_mutex;
static uint32_t* g_lcode_b_table[64] = {nullptr};

extern "C" size_t nf_lcode_packed_bf16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2; }

extern "C" int nf_lcode_pack_bf16(const float* const* params, void* stream_out, nf_stream_t stream) {
    if (!params || !stream_out) return NF_EINVAL;
    NfLcodePtrsH ptrs;
    for (int i = 0; i < nlc::NPARAMS; ++i) { if (!params[i]) return NF_EINVAL; ptrs.p[i] = params[i]; }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return NF_EINVAL;
    {
        std::lock_guard<std::mutex> lock(g_lcode_b_mutex);
        if (!g_lcode_b_table[dev]) {
            std::vector<uint32_t> host;
            nf_lcode_table_bf16(host);
            uint32_t* d = nullptr;
            e = hipMalloc(&d, host.size() * sizeof(uint32_t));
            if (e != hipSuccess) return (int)e;
            e = hipMemcpy(d, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(d); return (int)e; }
            g_lcode_b_table[dev] = d;
        }
    }
    hipLaunchKernelGGL(k_lcode_pack_bf16, dim3(1024), dim3(256), 0, nf_s(stream), ptrs, g_lcode_b_table[dev],
                       reinterpret_cast<__bf16*>(stream_out), nfb::N_PAIRS * 512);
    NF_RETURN_LAUNCH();
}

// =================================================================================================
// kernel
// =================================================================================================
__global__ void __launch_bounds__(256, 1)
k_lcode_mlp_fwd_bf16(const char* __restrict__ wstream, const float* __restrict__ cond, const float* __restrict__ ro,
                     const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z, int64_t n_points, int S,
                     float* __restrict__ raw) {
    using namespace nlc;
    __shared__ __attribute__((aligned(16))) char lds[NFB_LDS_BYTES];
    NfbCtx cx;
    cx.lane = threadIdx.x & 63;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cx.lds = lds;
    cx.gsrc = wstream + cx.lane * 16;
    const int h = cx.lane >> 5, c = cx.lane & 31;
    const int64_t p_raw = ((int64_t)blockIdx.x * 4 + cx.wave) * 32 + c;
    const int64_t p = p_raw < n_points ? p_raw : n_points - 1;       // clamp: every wave must reach every barrier
    const float* bias = reinterpret_cast<const float*>(lds + NFB_NBUF * NFB_STAGE_BYTES);

    // prologue DMA: bias table (10 KiB of the padded cond buffer), stages 0..2
    nfb_issue<NFB_BIAS_BLOCKS>(cx, reinterpret_cast<const char*>(cond) + cx.lane * 16, 0, NFB_NBUF * NFB_STAGE_BYTES);
    nfb_issue<nfb::stage_nblk(0)>(cx, cx.gsrc, nfb::stage_blk0(0), 0);
    nfb_issue<nfb::stage_nblk(1)>(cx, cx.gsrc, nfb::stage_blk0(1), NFB_STAGE_BYTES);
    nfb_issue<nfb::stage_nblk(2)>(cx, cx.gsrc, nfb::stage_blk0(2), 2 * NFB_STAGE_BYTES);

    bf16x8 bh[20], bl[20];                                            // [0..4): PE k-steps
    bf16x8 dh, dl;                                                    // dir k-step
    {
        const int64_t ray = p / S;
        const float zz = z[p];
        const float px = nf_add(ro[ray * 3 + 0], nf_mul(rd[ray * 3 + 0], zz));
        const float py = nf_add(ro[ray * 3 + 1], nf_mul(rd[ray * 3 + 1], zz));
        const float pz = nf_add(ro[ray * 3 + 2], nf_mul(rd[ray * 3 + 2], zz));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float x[8];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int pr = 16 * h + 4 * s + jj;                   // pair index (nfb::pe_col)
                const int freq = pr / 3, comp = pr - 3 * freq;
                const float v = comp == 0 ? px : (comp == 1 ? py : pz);
                float sn, cs;
                sincosf(nf_mul(v, (float)(1 << (freq < 10 ? freq : 0))), &sn, &cs);
                x[2 * jj] = sn;
                x[2 * jj + 1] = cs;
            }
            if (s == 3 && h == 1) { x[4] = px; x[5] = py; x[6] = pz; x[7] = 0.f; }
            nfb_split(x, bh[s], bl[s]);
        }
        const float dzv = rd_view[ray * 3 + 2];                       // Quirk Q1: "direction" = (rd_z, near, far)
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float sn, cs;
            sincosf(nf_mul(dzv, (float)(1 << (2 * h + jj))), &sn, &cs);
            x[2 * jj] = sn;
            x[2 * jj + 1] = cs;
        }
        nfb_split(x, dh, dl);
    }
    nfb_wait_vm<nfb::inflight_after(-1)>();                            // bias + stage 0 landed (later stages may be in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x16 acc[8];
    bf16x8 th[20], tl[20];
    // ---- layer1: PE -> 256, no activation (M:609) -----------------------------------------------------------------------
    nfb_init_bias<8>(acc, bias + B_L1, h);
    NFB_LAYER(0, acc, bh, bl);
    nfb_to_operands<8, false>(acc, bh, bl, 4);                        // hidden operands live in slots 4..19
#define NFB_LC_HIDDEN(L_, BIAS_)                                                     \
    do {                                                                             \
        nfb_init_bias<8>(acc, bias + (BIAS_), h);                                    \
        _Pragma("unroll") for (int s = 0; s < 16; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; } \
        NFB_LAYER(L_, acc, th, tl);                                                  \
        nfb_to_operands<8, true>(acc, bh, bl, 4);                                    \
    } while (0)
    NFB_LC_HIDDEN(1, B_X0);
    NFB_LC_HIDDEN(2, B_X1);
    NFB_LC_HIDDEN(3, B_X2);
    // ---- fc_alpha(x) (one tile, row 0), then feat = relu(fc_feat(x)) on the same operands -----------------------------------
#pragma unroll
    for (int s = 0; s < 16; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; }
    nfb_zero(acc[0]);
    if (h == 0) acc[0][0] = bias[B_ALPHA];
    NFB_LAYER(4, acc, th, tl);
    const float sigma_raw = acc[0][0];
    nfb_init_bias<8>(acc, bias + B_FEAT, h);
    NFB_LAYER(5, acc, th, tl);
    nfb_to_operands<8, true>(acc, bh, bl, 4);
#undef NFB_LC_HIDDEN
    // ---- layers_dir.0: 16 feat k-steps + dir k-step + 3 zero k-steps -> 128, ReLU -----------------------------------------
    nfb_init_bias<4>(acc, bias + B_DIR, h);
#pragma unroll
    for (int s = 0; s < 16; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; }
    th[16] = dh; tl[16] = dl;
#pragma unroll
    for (int s = 17; s < 20; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { th[s][j] = (__bf16)0.f; tl[s][j] = (__bf16)0.f; }
    NFB_LAYER(6, acc, th, tl);
    nfb_to_operands<4, true>(acc, bh, bl, 4);
    // ---- fc_rgb ---------------------------------------------------------------------------------------------------------
    nfb_zero(acc[0]);
#pragma unroll
    for (int s = 0; s < 8; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; }
    NFB_LAYER(7, acc, th, tl);
    if (h == 0 && p_raw < n_points) {
        const f32x4 o = {acc[0][0] + bias[B_RGB + 0], acc[0][1] + bias[B_RGB + 1], acc[0][2] + bias[B_RGB + 2], sigma_raw};
        reinterpret_cast<f32x4*>(raw)[p_raw] = o;
    }
}

// cond must be the padded table nf_lcode_condition fills (nf_lcode_cond_floats() floats >= 10 KiB)
extern "C" int nf_lcode_mlp_fwd_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd, const float* rd_view,
                                     const float* z, int64_t n_rays, int n_samples, float* raw, nf_stream_t stream) {
    if (!packed_bf16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_lcode_mlp_fwd_bf16, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), reinterpret_cast<const char*>(packed_bf16),
                       cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw);
    NF_RETURN_LAUNCH();
}
