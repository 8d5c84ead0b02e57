// Layer-streamed K loops of the exact-f32 INFERENCE kernels: the layer boundary runs under the MFMAs of the layers it joins.
//
// The round-2 kernel finished a layer with one block of non-matrix work per wave: 128 v_accvgpr_read + 256 v_max (the float
// ReLU is canonicalise + max) + 32 ds_write_b128, then 16 bias loads whose L2 latency nothing covered, 64 v_accvgpr_mov to seed
// the second point tile's accumulators and the first weight / B-fragment fetches of the next layer -- about 3000 cycles in
// which the matrix pipe of the SIMD idles (one wave per SIMD), 4.6 % of a 256 x 256 layer's 65536 MFMA cycles.  Here:
//   * accumulators go to the LDS slab RAW, straight from the accumulator registers (ds_write_b128 takes AGPRs), tile by tile
//     while the layer's last K chunk is still running: tile `no` is written NF_TAIL_LAG tiles after its last MFMA.  The slab
//     is dead by then -- the last chunk's B fragments are already in registers;
//   * the ReLU moves to the consumer: a lane reads back exactly the 16-byte fragments it wrote (D-register order = B-fragment
//     order, nf_mlp_layout.h), so max(x, 0) on the fragment after the ds_read is the same values -- 4 NT integer v_max per
//     128 MFMAs inside the K loop (as integers x > 0 iff the float is: no canonicalise, -0 -> +0 like v_max_f32);
//   * the bias is the C operand of the layer's first MFMAs (D = A B + bias): no accumulator initialisation at all.  The
//     bias fragments and the next layer's first weight chunk are requested at the top of the last chunk, one chunk = 4096
//     cycles before their use, and the next layer's first B fragment is read as soon as tile 0 is in the slab.
// Same products, same accumulation order, same values as the round-2 kernel: the outputs are bit-identical.
#pragma once
#include "nf_mlp_dev.h"

#ifndef NF_TAIL_LAG
#define NF_TAIL_LAG 2
#endif

template <int NT>
struct NfStream {
    f32x4 wa[16], wb[16];     // weight fragments of two consecutive K chunks
    f32x4 bias[16];           // the layer's bias fragments
    f32x4 b0[NT], b1[NT];     // B fragments of the same two chunks
};

__device__ __forceinline__ f32x4 nf_relu_i(f32x4 v) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int xi = __float_as_int(v[r]);
        o[r] = __int_as_float(xi > 0 ? xi : 0);                                  // v_max_i32
    }
    return o;
}

#ifndef NF_FWD_WBUF
#define NF_FWD_WBUF 1
#endif

// the packed weight image as the K loops address it: f32x4 index `off4` of a chunk (wave-uniform) + fragment no * 64 + lane
struct NfW {
    const f32x4* p;
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ NfW nf_w_image(const float* packed, int n_floats) {
    NfW w;
    w.p = reinterpret_cast<const f32x4*>(packed);
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), (short)0, n_floats * 4, 0x00020000);
    return w;
}

template <int NO>
__device__ __forceinline__ void nf_load_w16(f32x4 (&w)[16], const NfW& W, unsigned off4, int lane) {
#pragma unroll
    for (int no = 0; no < NO; ++no) {
#if NF_FWD_WBUF
        // buffer form: one VGPR (lane * 16 + a 12-bit immediate) and a scalar offset per load -- no 64-bit vector address arithmetic
        w[no] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W.rsrc, lane * 16 + (no & 3) * 1024, (int)(off4 * 16u) + (no >> 2) * 4096, 0));
#else
        w[no] = W.p[off4 + no * 64 + lane];
#endif
    }
}

// bias fragments of a layer: floats [off + 16 no + 4 g, + 4) of the per-call bias table `cond`
template <int NO>
__device__ __forceinline__ void nf_load_bias(f32x4 (&bias)[16], const NfW& C, unsigned off, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int no = 0; no < NO; ++no) {
#if NF_FWD_WBUF
        bias[no] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(C.rsrc, g * 16 + no * 64, (int)(off * 4u), 0));
#else
        bias[no] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(C.p) + off + 16 * no + 4 * g);
#endif
    }
}

template <int NT, bool RELU>
__device__ __forceinline__ void nf_read_b(f32x4 (&b)[NT], const f32x4* act4, int lane, int ni) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 v = act4[nf_act_idx4(16 * t + c, 4 * ni + g)];
        b[t] = RELU ? nf_relu_i(v) : v;
    }
}

// one K chunk; FIRST: the layer's first chunk, whose first MFMA per tile takes the bias as its C operand
template <int NT, int NO, bool FIRST>
__device__ __forceinline__ void nf_chunk(f32x4 (&acc)[NT][16], const f32x4 (&w)[16], const f32x4 (&b)[NT], const f32x4 (&bias)[16]) {
#pragma unroll
    for (int no = 0; no < NO; ++no)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t][no] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[no][r], b[t][r], (FIRST && r == 0) ? bias[no] : acc[t][no], 0, 0, 0);
}

#define NF_SGB_N(mask, n)                                                              \
    do {                                                                               \
        if ((n) == 1) __builtin_amdgcn_sched_group_barrier(mask, 1, 0);                \
        else if ((n) == 2) __builtin_amdgcn_sched_group_barrier(mask, 2, 0);           \
        else if ((n) == 3) __builtin_amdgcn_sched_group_barrier(mask, 3, 0);           \
        else if ((n) == 4) __builtin_amdgcn_sched_group_barrier(mask, 4, 0);           \
        else if ((n) == 8) __builtin_amdgcn_sched_group_barrier(mask, 8, 0);           \
    } while (0)

// timing ablations (results invalid): no weight loads / no slab reads inside the K loops
#ifndef NF_ABL_NOLOAD
#define NF_ABL_NOLOAD 0
#endif
#ifndef NF_ABL_NOLDS
#define NF_ABL_NOLDS 0
#endif
#ifndef NF_FWD_LOOP_PIPE
#define NF_FWD_LOOP_PIPE 1
#endif

// one half-iteration of the explicit pipeline: the MFMAs of a chunk (w, ReLU(raw)) with the fetches of the chunk after it
// (weights -> wn, raw fragment -> rawn) placed among them: the LDS reads first, one weight load per output tile
template <int NT, int NO, bool FIRST, bool RELU_IN>
__device__ __forceinline__ void nf_half(f32x4 (&acc)[NT][16], const f32x4 (&w)[16], const f32x4 (&raw)[NT], f32x4 (&wn)[16], f32x4 (&rawn)[NT],
                                        const f32x4 (&bias)[16], const NfW& W, unsigned wsrc, const f32x4* act4, int lane, int ni_next) {
    f32x4 b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = RELU_IN ? nf_relu_i(raw[t]) : raw[t];
    nf_read_b<NT, false>(rawn, act4, lane, ni_next);
    nf_load_w16<NO>(wn, W, wsrc, lane);
    nf_chunk<NT, NO, FIRST>(acc, w, b, bias);
    __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
#if NF_FWD_LOOP_PIPE == 2
    // burst form: all fetches of the next chunk at the top of this one (a full chunk ahead), then the MFMAs back to back
    __builtin_amdgcn_sched_group_barrier(0x020, NO, 0);
    if (RELU_IN) __builtin_amdgcn_sched_group_barrier(0x002, 4 * NT, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT * NO, 0);
#else
    if (RELU_IN) __builtin_amdgcn_sched_group_barrier(0x002, 4 * NT, 0);
#pragma unroll
    for (int no = 0; no < NO; ++no) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// The first nch - 1 of nch slab chunks of a layer (nch even).  Entry: st.wa / st.b0 hold chunk 0 (the fragment as stored: the ReLU,
// if any, is applied where it is consumed).  Exit: chunk nch - 1 is pending in st.wb / st.b1 -- the caller runs it with nf_tail
// (the layer ends there) or nf_chunk, through nf_pending_b.
template <int NT, int NO, bool FIRST, bool RELU_IN>
__device__ __forceinline__ void nf_seg_lds(f32x4 (&acc)[NT][16], NfStream<NT>& st, const NfW& W, unsigned wsec, int nch, const f32x4* act4,
                                           int lane) {
#if NF_FWD_LOOP_PIPE
    __builtin_amdgcn_sched_barrier(0);
    nf_half<NT, NO, FIRST, RELU_IN>(acc, st.wa, st.b0, st.wb, st.b1, st.bias, W, wsec + NO * 64, act4, lane, 1);
#pragma unroll 1
    for (int ni = 1; ni < nch - 1; ni += 2) {
        nf_half<NT, NO, false, RELU_IN>(acc, st.wb, st.b1, st.wa, st.b0, st.bias, W, wsec + (ni + 1) * NO * 64, act4, lane, ni + 1);
        nf_half<NT, NO, false, RELU_IN>(acc, st.wa, st.b0, st.wb, st.b1, st.bias, W, wsec + (ni + 2) * NO * 64, act4, lane, ni + 2);
    }
#else
    f32x4 b[NT];
    nf_load_w16<NO>(st.wb, W, wsec + NO * 64, lane);
    nf_read_b<NT, false>(st.b1, act4, lane, 1);
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = RELU_IN ? nf_relu_i(st.b0[t]) : st.b0[t];
    nf_chunk<NT, NO, FIRST>(acc, st.wa, b, st.bias);
#pragma unroll 1
    for (int ni = 1; ni < nch - 1; ni += 2) {
        if (!NF_ABL_NOLOAD) nf_load_w16<NO>(st.wa, W, wsec + (ni + 1) * NO * 64, lane);
        if (!NF_ABL_NOLDS) nf_read_b<NT, false>(st.b0, act4, lane, ni + 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = RELU_IN ? nf_relu_i(st.b1[t]) : st.b1[t];
        nf_chunk<NT, NO, false>(acc, st.wb, b, st.bias);
        if (!NF_ABL_NOLOAD) nf_load_w16<NO>(st.wb, W, wsec + (ni + 2) * NO * 64, lane);
        if (!NF_ABL_NOLDS) nf_read_b<NT, false>(st.b1, act4, lane, ni + 2);
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = RELU_IN ? nf_relu_i(st.b0[t]) : st.b0[t];
        nf_chunk<NT, NO, false>(acc, st.wa, b, st.bias);
    }
#endif
}

// the pending chunk's fragment as the MFMAs take it
template <int NT, bool RELU_IN>
__device__ __forceinline__ void nf_pending_b(f32x4 (&b)[NT], const NfStream<NT>& st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = RELU_IN ? nf_relu_i(st.b1[t]) : st.b1[t];
}


// The layer's last K chunk (weights w, fragments b) and the layer boundary under it.  NO tiles are computed, the first NO_ST of
// them go to the slab.  The next layer has NO_NEXT output tiles, weights at wnext, bias at bias_next; NEXT_B says how it starts:
// 0 = not from the slab (register chunks), 1 = from the slab.  Leaves the next layer's
// first weight chunk in st.wa, its bias in st.bias and (NEXT_B != 0) its first B fragment, as stored, in st.b0.
template <int NT, int NO, int NO_ST, int NO_NEXT, int NEXT_B>
__device__ __forceinline__ void nf_tail(f32x4 (&acc)[NT][16], const f32x4 (&w)[16], const f32x4 (&b)[NT], NfStream<NT>& st,
                                        const NfW& W, unsigned wnext, const NfW& C, unsigned bias_next, f32x4* act4, int lane) {
    constexpr int LAG = NF_TAIL_LAG;
    constexpr int NL = 2 * NO_NEXT;                                    // prefetch loads, spread over the NO tile steps
    const int g = lane >> 4, c = lane & 15;
    __builtin_amdgcn_sched_barrier(0);
    nf_load_bias<NO_NEXT>(st.bias, C, bias_next, lane);
    nf_load_w16<NO_NEXT>(st.wa, W, wnext, lane);
    f32x4 braw[NT];
#pragma unroll
    for (int no = 0; no < NO + LAG; ++no) {
        if (no < NO) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][no] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[no][r], b[t][r], acc[t][no], 0, 0, 0);
        }
        if (no >= LAG && no - LAG < NO_ST) {
#pragma unroll
            for (int t = 0; t < NT; ++t) act4[nf_act_idx4(16 * t + c, 4 * (no - LAG) + g)] = acc[t][no - LAG];
        }
        if (no == LAG && NEXT_B != 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) braw[t] = act4[nf_act_idx4(16 * t + c, g)];
        }
    }
#pragma unroll
    for (int no = 0; no < NO + LAG; ++no) {
        if (no < NO) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
            NF_SGB_N(0x020, (no + 1) * NL / NO - no * NL / NO);
        }
        if (no >= LAG && no - LAG < NO_ST) __builtin_amdgcn_sched_group_barrier(0x200, NT, 0);
        if (no == LAG && NEXT_B != 0) __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT_B != 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) st.b0[t] = braw[t];
    }
}
