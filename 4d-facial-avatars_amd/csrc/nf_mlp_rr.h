// Register-resident form of the exact-f32 inference kernel (k_paper_mlp_fwd_rr, build switch NF_F32_RR): activations never touch LDS.
//
// nf_mlp_layout.h chose the K order so that the D registers of output tile `no` ARE the B operands of the next layer's K chunk `no`
// (lane-local, register r = k-step r).  k_paper_mlp_fwd still sends them through the wave's LDS slab, because one accumulator set is
// overwritten in place by the next layer.  Here a wave keeps TWO accumulator sets (2 x 128 registers at NT = 2) and layers ping-pong
// between them: layer l reads set A as B operands (integer ReLU on the way, 4 NT v_max per 128 MFMAs) while it accumulates into set B.
// What pays for the second set: the bias is no longer a register array that rides as the C operand of a layer's first MFMAs -- a
// layer's bias fragments are LOADED INTO the accumulator registers of the set it will accumulate into, tile by tile, as the layer before
// it retires those tiles (tile ni of the input set is dead once chunk ni's B fragment has been taken).  No ds_write / ds_read, no LDS
// latency at layer boundaries, no slab; same products in the same order with the bias as the first addend: results are bit-identical
// to k_paper_mlp_fwd.
#pragma once
#include "nf_mlp_stream.h"

// timing ablations (WRONG results): NF_RR_NOLOAD = no weight / bias loads inside the layers (stale registers), NF_RR_NORELU = fragments taken
// without the ReLU, NF_RR_NOHEAD = no positional encoding (constants instead)
#ifndef NF_RR_VALU_MODE
#define NF_RR_VALU_MODE 0            // where the next fragment's ReLU sits in a chunk: 0 = behind the first tile, 1 / 2 = halves / quarters behind the first tiles, 4 = free
#endif
#ifndef NF_RR_PAIR
#define NF_RR_PAIR 0                 // 1: MFMAs of two output tiles interleaved (four independent accumulators in flight)
#endif
#ifndef NF_RR_NOLOAD
#define NF_RR_NOLOAD 0
#endif
#ifndef NF_RR_NORELU
#define NF_RR_NORELU 0
#endif
#ifndef NF_RR_NOHEAD
#define NF_RR_NOHEAD 0
#endif

template <int NT, bool RELU>
__device__ __forceinline__ void nf_rr_b(f32x4 (&b)[NT], const f32x4 (&in)[NT][16], int ni) {
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = (RELU && !NF_RR_NORELU) ? nf_relu_i(in[t][ni]) : in[t][ni];
}

// The lane-dependent parts of the load addresses: lane * 16 (weight fragments) and (lane >> 4) * 16 (bias fragments), re-derived through an
// opaque copy at the top of every point block.  As invariants of the persistent block loop the compiler hoisted `g * 16 + no * 64` and
// `lane * 16 + k * 1024` for every k and no -- two dozen address registers live across the whole body, which the 256 + 128 registers of the
// accumulator sets and the weight double buffer leave no room for (58 spilled registers); defined inside the block, the constant parts fold
// into the instructions' immediate offsets.
struct NfRrLane { int w16, g16; };

// bias fragment of output tile `no` (floats [off + 16 no + 4 g, + 4) of the per-call table) into both point tiles' accumulators
template <int NT>
__device__ __forceinline__ void nf_rr_bias(f32x4 (&acc)[NT][16], int no, const NfW& C, unsigned off, const NfRrLane& L) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
        acc[t][no] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(C.rsrc, L.g16 + no * 64, (int)(off * 4u), 0));
}

// ---- weights: a ring of three 8-tile groups (96 registers) instead of two 16-tile buffers (128) ----------------------------------------
// A chunk of NO output tiles occupies ng = ceil(NO / 8) consecutive groups from ring position POS.  The weights of the NEXT chunk's tile
// `no` are requested right behind this chunk's MFMAs of tile `no` and land in group (POS + ng + no / 8) % 3: for no < 8 that group is
// free, for no >= 8 (256-wide layers: ng = 2) it is the group this chunk's tiles 0..7 have just finished with.  Same prefetch distance
// as the double buffer (one chunk = 4096 cycles), 32 registers fewer -- the margin the second accumulator set needs.
struct NfRrRing { f32x4 g[3][8]; };

template <int POS, int NO>
__device__ __forceinline__ constexpr int nf_rr_next_pos() { return (POS + (NO + 7) / 8) % 3; }

// weight fragment `no` of the chunk at f32x4 offset off4 (NO_SEC tiles per chunk) into the ring slot of a chunk that starts at position POS
template <int POS>
__device__ __forceinline__ void nf_rr_w1(NfRrRing& R, int no, const NfW& W, unsigned off4, const NfRrLane& L) {
    R.g[(POS + no / 8) % 3][no % 8] =
        __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W.rsrc, L.w16 + (no & 3) * 1024, (int)(off4 * 16u) + (no >> 2) * 4096, 0));
}

// One K chunk: MFMAs of the chunk at ring position POS (NO tiles, fragment b) into out, tile by tile, each followed by the request for the
// same tile of the chunk after it (NO_N tiles at wn4; NO_N <= NO) -- and, behind the first NB tiles, one bias fragment of the layer that will
// accumulate into `dead` (tiles nb0 .. nb0 + NB - 1).  NVALU: vector instructions to place behind the first tile (the next fragment's ReLU).
template <int NT, int NO, int NO_N, int POS, int NB, int NVALU>
__device__ __forceinline__ void nf_rr_step(f32x4 (&out)[NT][16], NfRrRing& R, const f32x4 (&b)[NT], const NfW& W, unsigned wn4,
                                           f32x4 (&dead)[NT][16], int nb0, const NfW& C, unsigned bias_off, const NfRrLane& L) {
    static_assert(NO_N <= NO, "the next chunk's tiles ride behind this chunk's");
    constexpr int PN = nf_rr_next_pos<POS, NO>();
#if NF_RR_PAIR
    // tiles in pairs: four independent accumulators between two MFMAs on the same one (distance 4 instead of 2)
#pragma unroll
    for (int no = 0; no < NO; no += 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (no + q < NO)
                        out[t][no + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.g[(POS + (no + q) / 8) % 3][(no + q) % 8][r], b[t][r], out[t][no + q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!NF_RR_NOLOAD && no + q < NO_N) nf_rr_w1<PN>(R, no + q, W, wn4, L);
            if (!NF_RR_NOLOAD && no + q < NB) nf_rr_bias<NT>(dead, nb0 + no + q, C, bias_off, L);
        }
    }
#else
#pragma unroll
    for (int no = 0; no < NO; ++no) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                out[t][no] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.g[(POS + no / 8) % 3][no % 8][r], b[t][r], out[t][no], 0, 0, 0);
        if (!NF_RR_NOLOAD && no < NO_N) nf_rr_w1<PN>(R, no, W, wn4, L);
        if (!NF_RR_NOLOAD && no < NB) nf_rr_bias<NT>(dead, nb0 + no, C, bias_off, L);
    }
#endif
#pragma unroll
    for (int no = 0; no < NO; ++no) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
        // the next fragment's ReLU (v_accvgpr_read + v_max per element) two instructions per tile: 16 of them in a row behind one MFMA
        // cost 2.0 % of the kernel (ablation NF_RR_NORELU), an MFMA's 32 cycles cover about six
#if NF_RR_VALU_MODE == 0
        if (no == 0 && NVALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
#elif NF_RR_VALU_MODE == 1
        if (no < 2 && NVALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, NVALU / 2, 0);
#elif NF_RR_VALU_MODE == 2
        if (no < 4 && NVALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, NVALU / 4, 0);
#elif NF_RR_VALU_MODE == 3
        if (no >= 1 && no < 5 && NVALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, NVALU / 4, 0);
#endif
        if (!NF_RR_NOLOAD) NF_SGB_N(0x020, (no < NO_N ? 1 : 0) + (no < NB ? NT : 0));
    }
    __builtin_amdgcn_sched_barrier(0);
}

// A layer whose K chunks 0 .. NCH-1 are the tiles of the input set `in` (template recursion over the chunk index NI: every register index,
// ring position and scheduling count is a compile-time constant).  Entry: chunk 0's weights at ring position POS, chunk 0's fragment in b (as
// the MFMAs take it), out[.][0 .. NO) holds this layer's bias.  While it runs: weights and the B fragment one chunk ahead; the NEXT layer
// accumulates into `in`, so `in`'s tiles 0 .. NB_NEXT-1 take that layer's bias as they die (tile ni is dead once chunk ni's fragment has been
// taken).  The last chunk requests the chunk at `wnext` (NO_NEXT tiles): the next layer's first chunk, or this layer's register-fed chunk
// (layers_dir.0's dir slots).  Exit: that chunk at position (POS + NCH * ng) % 3.
template <int NT, int NO, int NCH, bool RELU_IN, int NO_NEXT, int NB_NEXT, int POS, int NI = 0>
__device__ __forceinline__ void nf_rr_layer(f32x4 (&out)[NT][16], f32x4 (&in)[NT][16], NfRrRing& R, f32x4 (&b)[NT], const NfW& W, unsigned wsec,
                                            unsigned wnext, const NfW& C, unsigned bias_next, const NfRrLane& L) {
    static_assert(NB_NEXT <= NCH, "a tile takes the next bias when its chunk retires it");
    if constexpr (NI < NCH) {
        constexpr bool last = NI + 1 == NCH;
        f32x4 bn[NT];
        if constexpr (!last) nf_rr_b<NT, RELU_IN>(bn, in, NI + 1);
        nf_rr_step<NT, NO, (last ? NO_NEXT : NO), POS, (NI < NB_NEXT ? 1 : 0), ((!last && RELU_IN) ? 8 * NT : 0)>(
            out, R, b, W, last ? wnext : wsec + (unsigned)(NI + 1) * NO * 64, in, NI, C, bias_next, L);
        if constexpr (!last) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = bn[t];
        }
        nf_rr_layer<NT, NO, NCH, RELU_IN, NO_NEXT, NB_NEXT, nf_rr_next_pos<POS, NO>(), NI + 1>(out, in, R, b, W, wsec, wnext, C, bias_next, L);
    }
}

// K chunks whose fragments are register arrays (PE slots, dir slots): chunks 0 .. NCH-1 of the section at wsec; the chunk after the last one
// is at wnext (NO_NEXT tiles).  PER > 0: an idle accumulator set `idle` takes PER bias tiles of the layer that will accumulate into it per chunk
template <int NT, int NO, int NCH, int NO_NEXT, int POS, int PER, int J = 0>
__device__ __forceinline__ void nf_rr_regs(f32x4 (&out)[NT][16], const f32x4 (&src)[NT][4], NfRrRing& R, const NfW& W, unsigned wsec,
                                           unsigned wnext, f32x4 (&idle)[NT][16], const NfW& C, unsigned bias_idle, const NfRrLane& L) {
    if constexpr (J < NCH) {
        constexpr bool last = J + 1 == NCH;
        f32x4 bj[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bj[t] = src[t][J];
        nf_rr_step<NT, NO, (last ? NO_NEXT : NO), POS, PER, 0>(out, R, bj, W, last ? wnext : wsec + (unsigned)(J + 1) * NO * 64, idle, J * PER, C,
                                                               bias_idle, L);
        nf_rr_regs<NT, NO, NCH, NO_NEXT, nf_rr_next_pos<POS, NO>(), PER, J + 1>(out, src, R, W, wsec, wnext, idle, C, bias_idle, L);
    }
}

// ring position after NCH chunks of NO tiles from POS
template <int POS, int NO, int NCH>
__device__ __forceinline__ constexpr int nf_rr_after() { return (POS + NCH * ((NO + 7) / 8)) % 3; }

template <int NT>
__global__ void __launch_bounds__(64 * NF_MLP_WAVES, 1)
k_paper_mlp_fwd_rr(const float* __restrict__ packed, const float* __restrict__ cond_, const float* __restrict__ ro,
                   const float* __restrict__ rd, const float* __restrict__ rd_view, const float* __restrict__ z,
                   int64_t n_points, int S, float* __restrict__ raw) {
    using namespace nfl;
    // the only LDS use: a lane parks its own PE fragments (32 registers) between layers_xyz.0 and the skip input of layers_xyz.3, and its
    // dir fragment until layers_dir.0 -- lane-private slots, no cross-lane traffic
    __shared__ __attribute__((aligned(16))) f32x4 park[NF_MLP_WAVES * (4 * NT + NT) * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    f32x4* mypark = park + wave * ((4 * NT + NT) * 64) + lane;
    typedef __attribute__((address_space(1))) f32x4 nf_gf32x4;
    nf_gf32x4* raw_v = (nf_gf32x4*)raw;
    asm volatile("" : "+v"(raw_v));
    int stride_v = (int)gridDim.x;
    asm volatile("" : "+v"(stride_v));
#pragma unroll 1
    for (int64_t blk = blockIdx.x;; blk += __builtin_amdgcn_readfirstlane(stride_v)) {
        const int64_t p0 = (blk * NF_MLP_WAVES + wave) * (16 * NT);
        if (p0 >= n_points) break;                        // wave-uniform; no barriers, no LDS
        int opaque0 = 0;                                  // per-block opaque zero: the weight / bias loads stay where the layers issue them
        asm volatile("" : "+s"(opaque0));
        const NfW Wi = nf_w_image(packed + opaque0, PACKED_FLOATS), Ci = nf_w_image(cond_ + opaque0, COND_FLOATS);
        NfRrLane L{lane * 16, (lane >> 4) * 16};
        asm volatile("" : "+v"(L.w16), "+v"(L.g16));
        f32x4 X[NT][16], Y[NT][16];                       // the two accumulator sets
        NfRrRing R;
        // ---- inputs: pts = ro + rd*z (T:78), PE fragments, dir fragment -----------------------------
        f32x4 pe[NT][4];
        f32x4 dirf[NT][4];
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
#if NF_RR_NOHEAD
#pragma unroll
            for (int j = 0; j < 4; ++j) pe[t][j] = (f32x4){px, py, pz, 0.5f};
            dirf[t][0] = (f32x4){rd_view[ray * 3 + 2], 1.0f, 0.0f, 0.0f};
#else
            nf_encode_point(px, py, pz, g, pe[t]);
            float s, cs;
            nf_sincos(nf_mul(rd_view[ray * 3 + 2], (float)(1 << g)), &s, &cs);   // Quirk Q1: "direction" = (rd_z, near, far)
            dirf[t][0] = (f32x4){s, cs, 0.0f, 0.0f};
#endif
        }
        // the first layer's bias (into its accumulators) and first weight chunk
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int no = 0; no < 16; ++no) nf_rr_bias<NT>(X, no, Ci, B_L0, L);
#pragma unroll
        for (int no = 0; no < 16; ++no) nf_rr_w1<0>(R, no, Wi, OFF_L0 / 4, L);
        f32x4 b[NT];
        constexpr int P_L1 = nf_rr_after<0, 16, 4>(), P_L2 = nf_rr_after<P_L1, 16, 16>(), P_L3 = nf_rr_after<P_L2, 16, 16>(),
                      P_L3B = nf_rr_after<P_L3, 16, 4>(), P_L4 = nf_rr_after<P_L3B, 16, 16>(), P_L5 = nf_rr_after<P_L4, 16, 16>(),
                      P_FEAT = nf_rr_after<P_L5, 16, 16>(), P_D0 = nf_rr_after<P_FEAT, 16, 16>(), P_D0B = nf_rr_after<P_D0, 9, 16>(),
                      P_D1 = nf_rr_after<P_D0B, 9, 1>(), P_D2 = nf_rr_after<P_D1, 8, 8>(), P_RGB = nf_rr_after<P_D2, 8, 8>();
        // layers_xyz.0: PE -> X (layers_xyz.1's bias into the idle Y)
        nf_rr_regs<NT, 16, 4, 16, 0, 4>(X, pe, R, Wi, OFF_L0 / 4, OFF_L1 / 4, Y, Ci, B_L1, L);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mypark[(4 * t + j) * 64] = pe[t][j];
            mypark[(4 * NT + t) * 64] = dirf[t][0];
        }
        // layers_xyz.1: X -> Y (layers_xyz.2's bias into X); layers_xyz.2: Y -> X (layers_xyz.3's bias into Y)
        nf_rr_b<NT, true>(b, X, 0);
        nf_rr_layer<NT, 16, 16, true, 16, 16, P_L1>(Y, X, R, b, Wi, OFF_L1 / 4, OFF_L2 / 4, Ci, B_L2, L);
        nf_rr_b<NT, true>(b, Y, 0);
        nf_rr_layer<NT, 16, 16, true, 16, 16, P_L2>(X, Y, R, b, Wi, OFF_L2 / 4, OFF_L3 / 4, Ci, B_L3, L);
        // layers_xyz.3: [PE | X] -> Y (skip connection, M:246); layers_xyz.4's bias into X
        f32x4 pe3[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) pe3[t][j] = mypark[(4 * t + j) * 64];
        nf_rr_regs<NT, 16, 4, 16, P_L3, 0>(Y, pe3, R, Wi, OFF_L3 / 4, OFF_L3 / 4 + 4 * 16 * 64, Y, Ci, 0u, L);
        nf_rr_b<NT, true>(b, X, 0);
        nf_rr_layer<NT, 16, 16, true, 16, 16, P_L3B>(Y, X, R, b, Wi, OFF_L3 / 4 + 4 * 16 * 64, OFF_L4 / 4, Ci, B_L4, L);
        // layers_xyz.4: Y -> X; layers_xyz.5: X -> Y; fc_feat: Y -> X (layers_dir.0's 9 bias tiles into Y)
        nf_rr_b<NT, true>(b, Y, 0);
        nf_rr_layer<NT, 16, 16, true, 16, 16, P_L4>(X, Y, R, b, Wi, OFF_L4 / 4, OFF_L5 / 4, Ci, B_L5, L);
        nf_rr_b<NT, true>(b, X, 0);
        nf_rr_layer<NT, 16, 16, true, 16, 16, P_L5>(Y, X, R, b, Wi, OFF_L5 / 4, OFF_FEAT / 4, Ci, B_FEAT, L);
        nf_rr_b<NT, true>(b, Y, 0);
        nf_rr_layer<NT, 16, 16, true, 9, 9, P_FEAT>(X, Y, R, b, Wi, OFF_FEAT / 4, OFF_D0 / 4, Ci, B_D0, L);
        // layers_dir.0: [feat (no activation, M:250) | dir slots] -> Y tiles 0..7, tile 8 row 0 = fc_alpha(feat) (Q2); layers_dir.1's bias into X
        nf_rr_b<NT, false>(b, X, 0);
        nf_rr_layer<NT, 9, 16, false, 9, 8, P_D0>(Y, X, R, b, Wi, OFF_D0 / 4, OFF_D0 / 4 + 16 * 9 * 64, Ci, B_D1, L);
        f32x4 dir0[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) dir0[t][0] = mypark[(4 * NT + t) * 64];
        nf_rr_regs<NT, 9, 1, 8, P_D0B, 0>(Y, dir0, R, Wi, OFF_D0 / 4 + 16 * 9 * 64, OFF_D1 / 4, Y, Ci, 0u, L);
        float sigma_raw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) sigma_raw[t] = Y[t][8].x;
        // layers_dir.1: Y -> X; layers_dir.2: X -> Y; fc_rgb: Y -> X tile 0
        nf_rr_b<NT, true>(b, Y, 0);
        nf_rr_layer<NT, 8, 8, true, 8, 8, P_D1>(X, Y, R, b, Wi, OFF_D1 / 4, OFF_D2 / 4, Ci, B_D2, L);
        nf_rr_b<NT, true>(b, X, 0);
        nf_rr_layer<NT, 8, 8, true, 1, 1, P_D2>(Y, X, R, b, Wi, OFF_D2 / 4, OFF_RGB / 4, Ci, B_RGB, L);
        nf_rr_b<NT, true>(b, Y, 0);
        nf_rr_layer<NT, 1, 8, true, 0, 0, P_RGB>(X, Y, R, b, Wi, OFF_RGB / 4, OFF_RGB / 4, Ci, B_RGB, L);
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        if ((lane_o >> 4) == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int64_t p = p0 + 16 * t + c;
                if (p < n_points) raw_v[p] = (f32x4){X[t][0].x, X[t][0].y, X[t][0].z, sigma_raw[t]};
            }
        }
    }
}
