// K4, split-bf16 variant: the fused paper-MLP forward on the bf16 matrix pipe at fp32-class accuracy.
//
// CDNA4 has no TF32-like mode: exact-f32 MFMA runs at 1/16 of the bf16 rate.  This kernel evaluates every GEMM
// as THREE bf16 MFMAs with f32 accumulation,
//        x = x_hi + x_lo,  x_hi = bf16(x), x_lo = bf16(x - x_hi)          (16 significand bits kept)
//        W.x ~= W_hi.x_hi + W_hi.x_lo + W_lo.x_hi                          (dropped term ~ 2^-16)
// products of bf16 pairs are exact in f32, so the only deviations from the f32 kernel are the 2^-17 truncation of
// the operands and the dropped lo*lo term: relative error ~2^-16 per layer instead of 2^-24, which keeps the
// end-to-end |dPSNR| two orders of magnitude inside the 1e-4 dB gate (tests/test_gpu_bf16.py) while tripling
// throughput.  Same structure as nf_mlp.hip otherwise (see nf_mlp_layout.h), adapted to
// v_mfma_f32_32x32x16_bf16 (D[32 out-features x 32 points] += W[32 x 16] act[16 x 32]):
//   * lane (h = l>>5, c = l&31): A operand = W[n0 + c][8 k-slots of half h], B operand = act[point c][same slots],
//     D regs r = 0..15 = out-feature n0 + (r&3) + 8(r>>2) + 4h of point c;
//   * K order is chosen so that the 8 slots lane half h feeds to k-step s are exactly the features that half h
//     holds in D regs 8u..8u+7 of tile nt (s = 2 nt + u): a layer's accumulators become the next layer's B
//     operands by a lane-local f32 -> (bf16 hi, bf16 lo) split -- activations never leave registers;
//   * a wave owns 32 points for the whole network; the 4 waves of a workgroup share the weight stream, staged
//     L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) two k-steps at a time into a
//     double buffer, one workgroup barrier per stage; LDS reads are lane-linear ds_read_b128 (conflict-free).
#include <vector>
#include <mutex>
#include "nf_common.h"
#include "nf_mlp_layout.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace nfb {
// layer table of the bf16 stream: k-steps (16 slots each, always even) and 32-row output tiles
constexpr int NL = 11;
constexpr int KS[NL] = {4, 16, 16, 20, 16, 16, 16, 20, 8, 8, 8};   // multiples of the stage depth (4 k-steps)
constexpr int NO[NL] = {8, 8, 8, 8, 8, 8, 8, 5, 4, 4, 1};
constexpr int pair_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += KS[i] * NO[i]; return o; }
constexpr int N_PAIRS = pair_off(NL);                 // (hi, lo) 1-KiB block pairs
constexpr int STREAM_BF16 = N_PAIRS * 2 * 512;        // bf16 elements
// slot (s, h, j) of a hidden input -> feature index (D register order of the producing layer)
__host__ __device__ constexpr int hid_feature(int s, int h, int j) { return 16 * s + 4 * h + (j & 3) + 8 * (j >> 2); }
// PE slots: 4 k-steps; lane half h, step s, j: pair p = 16 h + 4 s + (j >> 1); sc = j & 1.
// p < 30: (freq, comp) = (p / 3, p % 3); p = 30: raw x, raw y; p = 31: raw z, zero pad.
__host__ __device__ constexpr int pe_col(int s, int h, int j) {
    const int p = 16 * h + 4 * s + (j >> 1), sc = j & 1;
    if (p < 30) return 3 + 6 * (p / 3) + 3 * sc + (p % 3);
    if (p == 30) return sc;            // x, y
    return sc == 0 ? 2 : -1;           // z, pad
}
// dir slots (one k-step): half h, j < 4: freq = 2 h + (j >> 1), sc = j & 1 -> layers_dir.0 column 256 + 6 f + 3 sc
__host__ __device__ constexpr int dir_col(int h, int j) { return j < 4 ? 256 + 6 * (2 * h + (j >> 1)) + 3 * (j & 1) : -1; }
}  // namespace nfb

// =================================================================================================
// pack: fp32 parameters -> (hi, lo) bf16 fragment stream
// =================================================================================================
struct NfParamPtrsB { const float* p[NF_PAPER_NUM_PARAMS]; };
static const uint32_t NF_ZERO_B = 0xFF000000u;

static void nf_build_table_bf16(std::vector<uint32_t>& t) {
    using namespace nfb;
    t.assign((size_t)N_PAIRS * 512, NF_ZERO_B);
    // per layer: tensor id of the weight, its column count, and the slot -> column rule
    auto code = [](int tensor, int row, int col, int ncols) { return ((uint32_t)tensor << 24) | (uint32_t)(row * ncols + col); };
    for (int l = 0; l < NL; ++l) {
        for (int s = 0; s < KS[l]; ++s)
            for (int nt = 0; nt < NO[l]; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int h = lane >> 5, i = lane & 31, n = 32 * nt + i;
                        uint32_t c = NF_ZERO_B;
                        switch (l) {
                            case 0: { const int col = pe_col(s, h, j); if (col >= 0) c = code(0, n, col, 171); } break;
                            case 1: c = code(2, n, hid_feature(s, h, j), 256); break;
                            case 2: c = code(4, n, hid_feature(s, h, j), 256); break;
                            case 3: {
                                if (s < 4) { const int col = pe_col(s, h, j); if (col >= 0) c = code(6, n, col, 427); }
                                else c = code(6, n, 171 + hid_feature(s - 4, h, j), 427);
                            } break;
                            case 4: c = code(8, n, hid_feature(s, h, j), 256); break;
                            case 5: c = code(10, n, hid_feature(s, h, j), 256); break;
                            case 6: c = code(12, n, hid_feature(s, h, j), 256); break;
                            case 7: {   // layers_dir.0 (+ fc_alpha as row 128); k-steps 0..15 feat, 16 dir slots, 17..19 zero
                                if (n < 128) {
                                    if (s < 16) c = code(16, n, hid_feature(s, h, j), 280);
                                    else if (s == 16) { const int col = dir_col(h, j); if (col >= 0) c = code(16, n, col, 280); }
                                } else if (n == 128 && s < 16) c = code(14, 0, hid_feature(s, h, j), 256);
                            } break;
                            case 8: c = code(18, n, hid_feature(s, h, j), 128); break;
                            case 9: c = code(20, n, hid_feature(s, h, j), 128); break;
                            case 10: if (n < 3) c = code(24, n, hid_feature(s, h, j), 128); break;
                        }
                        t[((size_t)(pair_off(l) + s * NO[l] + nt)) * 512 + lane * 8 + j] = c;
                    }
    }
}

__global__ void __launch_bounds__(256) k_paper_pack_bf16(NfParamPtrsB ptrs, const uint32_t* __restrict__ table,
                                                         __bf16* __restrict__ stream, int n_entries) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const uint32_t code = table[e];
        const uint32_t id = code >> 24;
        const float w = id == 0xFFu ? 0.0f : ptrs.p[id][code & 0xFFFFFFu];
        const __bf16 hi = (__bf16)w;
        const __bf16 lo = (__bf16)(w - (float)hi);
        const int pair = e >> 9, within = e & 511;
        stream[(size_t)(2 * pair) * 512 + within] = hi;
        stream[(size_t)(2 * pair + 1) * 512 + within] = lo;
    }
}

static std::mutex g_table_b_mutex;
static uint32_t* g_table_b_dev[64] = {nullptr};

extern "C" size_t nf_paper_packed_bf16_bytes(void) { return (size_t)nfb::STREAM_BF16 * 2; }

extern "C" int nf_paper_pack_bf16(const float* const* params, void* stream_out, nf_stream_t stream) {
    if (!params || !stream_out) return NF_EINVAL;
    NfParamPtrsB ptrs;
    for (int i = 0; i < NF_PAPER_NUM_PARAMS; ++i) {
        if (!params[i]) return NF_EINVAL;
        ptrs.p[i] = params[i];
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return NF_EINVAL;
    {
        std::lock_guard<std::mutex> lock(g_table_b_mutex);
        if (!g_table_b_dev[dev]) {
            std::vector<uint32_t> host;
            nf_build_table_bf16(host);
            uint32_t* d = nullptr;
            e = hipMalloc(&d, host.size() * sizeof(uint32_t));
            if (e != hipSuccess) return (int)e;
            e = hipMemcpy(d, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(d); return (int)e; }
            g_table_b_dev[dev] = d;
        }
    }
    hipLaunchKernelGGL(k_paper_pack_bf16, dim3(1024), dim3(256), 0, nf_s(stream), ptrs, g_table_b_dev[dev],
                       reinterpret_cast<__bf16*>(stream_out), nfb::N_PAIRS * 512);
    NF_RETURN_LAUNCH();
}

// =================================================================================================
// forward
// =================================================================================================
// Weight stream -> LDS pipeline.  A stage = NFB_KPS k-steps of one layer (all of its output tiles, hi and lo):
// <= 32 KiB.  Three LDS buffers form a ring; the DMA of stage g+2 is issued while stage g is consumed, and the
// end-of-stage wait is a COUNTED s_waitcnt vmcnt(n) that leaves exactly that newest group in flight across the
// (raw) workgroup barrier.  Per-call biases are DMA'd once into LDS so that no ordinary global load (whose
// compiler-inserted vmcnt(0) would drain the pipeline) remains in the layer loop.  All LDS lives in ONE array.
#define NFB_KPS 2                                   // k-steps per stage
#define NFB_STAGE_BYTES (NFB_KPS * 8 * 2 * 1024)    // largest stage: 8 tiles x (hi, lo) x 1 KiB per k-step = 32 KiB
#define NFB_NBUF 4                                  // ring depth; stage g + NFB_NBUF - 1 is prefetched while stage g is consumed
#define NFB_LA (NFB_NBUF - 1)
#define NFB_BIAS_BLOCKS 10                          // cond table (2332 f32) padded to 10 KiB
#define NFB_LDS_BYTES (NFB_NBUF * NFB_STAGE_BYTES + NFB_BIAS_BLOCKS * 1024)

namespace nfb {
// global stage table (compile time): stage g -> (first stream block, number of blocks)
constexpr int stages_of(int l) { return KS[l] / NFB_KPS; }
constexpr int stage0_of(int l) { int o = 0; for (int i = 0; i < l; ++i) o += stages_of(i); return o; }
constexpr int N_STAGES = stage0_of(NL);
constexpr int layer_of_stage(int g) { int l = 0; while (l < NL && g >= stage0_of(l + 1)) ++l; return l; }
constexpr int stage_nblk(int g) { return (g < 0 || g >= N_STAGES) ? 0 : 2 * NFB_KPS * NO[layer_of_stage(g)]; }
// DMA pieces of this wave that may stay in flight when stage g ends: those of stages g+2 .. g+NFB_LA
constexpr int inflight_after(int g) { int n = 0; for (int k = 2; k <= NFB_NBUF - 1; ++k) n += stage_nblk(g + k) / 4; return n; }
constexpr int stage_blk0(int g) {
    if (g < 0 || g >= N_STAGES) return 0;
    const int l = layer_of_stage(g);
    return 2 * pair_off(l) + (g - stage0_of(l)) * 2 * NFB_KPS * NO[l];
}
}  // namespace nfb

struct NfbCtx {
    const char* gsrc;        // this lane's source pointer into the weight stream (stream base + lane * 16)
    char* lds;               // workgroup LDS base
    int lane, wave;
};

template <int N> __device__ __forceinline__ void nfb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA of NBLK 1-KiB blocks, stream block blk0.. -> LDS byte offset lds_off (block b is fetched by wave b % 4).
template <int NBLK>
__device__ __forceinline__ void nfb_issue(const NfbCtx& cx, const char* gsrc, int blk0, int lds_off) {
#pragma unroll
    for (int q = 0; q < (NBLK + 3) / 4; ++q) {
        const int b = cx.wave + 4 * q;
        if (NBLK % 4 == 0 || b < NBLK)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)(blk0 + b) * 1024),
                                             (__attribute__((address_space(3))) void*)(cx.lds + lds_off + b * 1024), 16, 0, 0);
    }
}

// Stage G (compile time): consume stage G out of ring buffer G % 3 while prefetching stage G+2, then leave only that
// newest DMA group in flight (counted vmcnt) and cross the workgroup barrier.
// The four waves issue their DMA bursts at DIFFERENT points of the stage (wave w after a quarter w of its MFMAs): the
// CU has one texture-address unit, and four simultaneous bursts right behind the barrier make every wave wait for all
// 4 x NQ pieces to be accepted; staggered, a wave only waits for its own.
template <int G, int NO>
__device__ __forceinline__ void nfb_stage(const NfbCtx& cx, f32x16 (&acc)[8], const bf16x8 (&bh)[20], const bf16x8 (&bl)[20], int s0) {
    constexpr int nb2 = nfb::stage_nblk(G + NFB_LA);
    constexpr int blk2 = nfb::stage_blk0(G + NFB_LA);
    constexpr int off2 = ((G + NFB_LA) % NFB_NBUF) * NFB_STAGE_BYTES;
    constexpr int NM = NFB_KPS * 3 * NO;                      // MFMAs in this stage
    const char* base = cx.lds + (G % NFB_NBUF) * NFB_STAGE_BYTES + cx.lane * 16;
    int m = 0;                                                // MFMA counter (compile time after unrolling)
#pragma unroll
    for (int u = 0; u < NFB_KPS; ++u) {
        bf16x8 ah[NO], al[NO];
#pragma unroll
        for (int nt = 0; nt < NO; ++nt) {
            ah[nt] = *reinterpret_cast<const bf16x8*>(base + ((u * NO + nt) * 2 + 0) * 1024);
            al[nt] = *reinterpret_cast<const bf16x8*>(base + ((u * NO + nt) * 2 + 1) * 1024);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int nt = 0; nt < NO; ++nt) {
                if (nb2 > 0) {
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (m == (w * NM) / 4 && cx.wave == w)
                            nfb_issue<(nb2 > 0 ? nb2 : 4)>(cx, cx.gsrc, blk2, off2);
                }
                const bf16x8& a = t == 0 ? al[nt] : ah[nt];
                const bf16x8& b = t == 1 ? bl[s0 + u] : bh[s0 + u];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);
                ++m;
            }
    }
    nfb_wait_vm<nfb::inflight_after(G)>();                    // stage G+1 has landed for this wave; later stages may still fly
    __builtin_amdgcn_s_barrier();                             // ... and for every wave; buffer G % 3 is free again
    asm volatile("" ::: "memory");
}

template <int G0, int NST, int NO, int I = 0>
__device__ __forceinline__ void nfb_layer(const NfbCtx& cx, f32x16 (&acc)[8], const bf16x8 (&bh)[20], const bf16x8 (&bl)[20]) {
    if constexpr (I < NST) {
        nfb_stage<G0 + I, NO>(cx, acc, bh, bl, NFB_KPS * I);
        nfb_layer<G0, NST, NO, I + 1>(cx, acc, bh, bl);
    }
}
#define NFB_LAYER(L_, acc_, bh_, bl_) nfb_layer<nfb::stage0_of(L_), nfb::stages_of(L_), nfb::NO[L_]>(cx, acc_, bh_, bl_)

__device__ __forceinline__ void nfb_split(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

// accumulators (NO tiles) -> B operands of the next layer: tile nt, regs 8u..8u+7 -> k-step 2 nt + u
template <int NO, bool RELU>
__device__ __forceinline__ void nfb_to_operands(const f32x16 (&acc)[8], bf16x8 (&bh)[20], bf16x8 (&bl)[20], int s_off) {
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = acc[nt][8 * u + j];
                x[j] = RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v;
            }
            nfb_split(x, bh[s_off + 2 * nt + u], bl[s_off + 2 * nt + u]);
        }
}

// acc[nt] reg r <- bias[32 nt + (r&3) + 8 (r>>2) + 4 h]   (bias table in LDS)
template <int NO>
__device__ __forceinline__ void nfb_init_bias(f32x16 (&acc)[8], const float* bias, int h) {
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * nt + 8 * q + 4 * h);
            acc[nt][4 * q + 0] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
        }
}

__device__ __forceinline__ void nfb_zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

__global__ void __launch_bounds__(256, 1)
k_paper_mlp_fwd_bf16(const char* __restrict__ wstream, const float* __restrict__ cond, const float* __restrict__ ro,
                     const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z,
                     int64_t n_points, int S, float* __restrict__ raw) {
    using namespace nfl;
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

    // prologue DMA: bias table, stage 0, stage 1
    nfb_issue<NFB_BIAS_BLOCKS>(cx, reinterpret_cast<const char*>(cond) + cx.lane * 16, 0, NFB_NBUF * NFB_STAGE_BYTES);
    nfb_issue<nfb::stage_nblk(0)>(cx, cx.gsrc, nfb::stage_blk0(0), 0);
    nfb_issue<nfb::stage_nblk(1)>(cx, cx.gsrc, nfb::stage_blk0(1), NFB_STAGE_BYTES);
    if (NFB_LA > 2) nfb_issue<nfb::stage_nblk(2)>(cx, cx.gsrc, nfb::stage_blk0(2), 2 * NFB_STAGE_BYTES);

    // ---- inputs ---------------------------------------------------------------------------------------
    bf16x8 bh[20], bl[20];                                            // [0..4): PE k-steps (kept for the skip layer)
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
        const float dzv = rd_view[ray * 3 + 2];
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
    // ---- layers_xyz.0 : PE -> 256 --------------------------------------------------------------------------
    nfb_init_bias<8>(acc, bias + B_L0, h);
    NFB_LAYER(0, acc, bh, bl);
    nfb_to_operands<8, true>(acc, bh, bl, 4);                          // hidden operands live in slots 4..19, PE stays in 0..3
#define NFB_HIDDEN_LAYER(L_, BIAS_, RELU_)                                           \
    do {                                                                             \
        nfb_init_bias<8>(acc, bias + (BIAS_), h);                                    \
        _Pragma("unroll") for (int s = 0; s < 16; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; } \
        NFB_LAYER(L_, acc, th, tl);                                                  \
        nfb_to_operands<8, RELU_>(acc, bh, bl, 4);                                   \
    } while (0)
    NFB_HIDDEN_LAYER(1, B_L1, true);
    NFB_HIDDEN_LAYER(2, B_L2, true);
    // ---- layers_xyz.3 : [PE | h] (20 k-steps: operands 0..19 as they sit) ------------------------------------------
    nfb_init_bias<8>(acc, bias + B_L3, h);
    NFB_LAYER(3, acc, bh, bl);
    nfb_to_operands<8, true>(acc, bh, bl, 4);
    NFB_HIDDEN_LAYER(4, B_L4, true);
    NFB_HIDDEN_LAYER(5, B_L5, true);
    NFB_HIDDEN_LAYER(6, B_FEAT, false);
#undef NFB_HIDDEN_LAYER
    // ---- layers_dir.0 (+ fc_alpha as tile 4): 16 feat k-steps + dir k-step + 3 zero k-steps ------------------------------
    nfb_init_bias<4>(acc, bias + B_D0, h);
    nfb_zero(acc[4]);
    if (h == 0) acc[4][0] = bias[B_D0 + 128];
#pragma unroll
    for (int s = 0; s < 16; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; }
    th[16] = dh; tl[16] = dl;
#pragma unroll
    for (int s = 17; s < 20; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { th[s][j] = (__bf16)0.f; tl[s][j] = (__bf16)0.f; }
    NFB_LAYER(7, acc, th, tl);
    const float sigma_raw = acc[4][0];
    nfb_to_operands<4, true>(acc, bh, bl, 4);
    // ---- layers_dir.1, .2 ----------------------------------------------------------------------------------------------
    nfb_init_bias<4>(acc, bias + B_D1, h);
#pragma unroll
    for (int s = 0; s < 8; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; }
    NFB_LAYER(8, acc, th, tl);
    nfb_to_operands<4, true>(acc, bh, bl, 4);
    nfb_init_bias<4>(acc, bias + B_D2, h);
#pragma unroll
    for (int s = 0; s < 8; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; }
    NFB_LAYER(9, acc, th, tl);
    nfb_to_operands<4, true>(acc, bh, bl, 4);
    // ---- fc_rgb -----------------------------------------------------------------------------------------------------------
    nfb_zero(acc[0]);
#pragma unroll
    for (int s = 0; s < 8; ++s) { th[s] = bh[4 + s]; tl[s] = bl[4 + s]; }
    NFB_LAYER(10, acc, th, tl);
    if (h == 0 && p_raw < n_points) {
        const f32x4 o = {acc[0][0] + bias[B_RGB + 0], acc[0][1] + bias[B_RGB + 1], acc[0][2] + bias[B_RGB + 2], sigma_raw};
        reinterpret_cast<f32x4*>(raw)[p_raw] = o;
    }
}

extern "C" int nf_paper_mlp_fwd_bf16(const void* packed_bf16, const float* cond, const float* ro, const float* rd,
                                     const float* rd_view, const float* z, int64_t n_rays, int n_samples, float* raw,
                                     nf_stream_t stream) {
    if (!packed_bf16 || !cond || !ro || !rd || !z || !raw || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t n_points = n_rays * n_samples;
    if (n_points == 0) return 0;
    const int64_t grid = (n_points + 127) / 128;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_paper_mlp_fwd_bf16, dim3((unsigned)grid), dim3(256), 0, nf_s(stream),
                       reinterpret_cast<const char*>(packed_bf16), cond, ro, rd, rd_view ? rd_view : rd, z, n_points, n_samples, raw);
    NF_RETURN_LAUNCH();
}
