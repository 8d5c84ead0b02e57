// Layer-streamed K loops of the exact-f32 forward kernels (inference: k_paper_mlp_fwd; training: k_paper_mlp_fwd_save, with the copy to
// `saved`, the ReLU and the mask bits riding in the loops): the layer boundary runs under the MFMAs of the layers it joins.
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
// NF_ONEWAIT (inference kernel only; experiment): one `s_waitcnt vmcnt(0)` at the top of a K chunk instead of one counted wait per output
// tile.  With the next chunk's 16 weight loads spread one per tile over the whole chunk, every tile's first MFMA needs its own
// `s_waitcnt vmcnt(15)` -- 16 extra issue slots per 128 MFMAs in a one-wave-per-SIMD stream where a non-MFMA instruction costs ~5 cycles
// (profiles/r06_f32_rr.md).  Here the loads go out two per tile behind the FIRST half of the chunk's tiles, so that they are at least half
// a chunk (2048 cycles) old at the next chunk's top, where a single wait covers them all.
#ifndef NF_ONEWAIT
#define NF_ONEWAIT 0
#endif
#define NF_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)          // vmcnt(0), expcnt / lgkmcnt unconstrained (gfx9 encoding)

template <int NT>
struct NfStream {
    f32x4 wa[16], wb[16];     // weight fragments of two consecutive K chunks
    f32x4 bias[16];           // the layer's bias fragments
    f32x4 b0[NT], b1[NT];     // B fragments of the same two chunks
    f32x4 bp[NT];             // training forward: the fragment the previous half-iteration consumed, until its mask bits are taken
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

// the packed weight image (or the per-call bias table) as the K loops address it: a buffer descriptor, so that the per-lane part of
// an address is ONE VGPR (lane * 16 + a multiple of 1 KiB) and the chunk offset is scalar -- against a 64-bit vector address plus a
// v_add_co / v_addc pair per four loads in the global form, which a one-wave-per-SIMD MFMA loop does not get for free
// (91.6 -> 86.7 ms per fine launch on this change alone, profiles/r03_mlp_f32_stream.md)
struct NfW {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ NfW nf_w_image(const float* base, int n_floats) {
    NfW w;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, n_floats * 4, 0x00020000);
    return w;
}

// weight fragments of one K chunk: f32x4 index `off4` of the chunk (wave-uniform) + fragment no * 64 + lane
template <int NO>
__device__ __forceinline__ void nf_load_w16(f32x4 (&w)[16], const NfW& W, unsigned off4, int lane) {
#pragma unroll
    for (int no = 0; no < NO; ++no)
        w[no] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W.rsrc, lane * 16 + (no & 3) * 1024, (int)(off4 * 16u) + (no >> 2) * 4096, 0));
}

// bias fragments of a layer: floats [off + 16 no + 4 g, + 4) of the per-call bias table `cond`
template <int NO>
__device__ __forceinline__ void nf_load_bias(f32x4 (&bias)[16], const NfW& C, unsigned off, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int no = 0; no < NO; ++no)
        bias[no] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(C.rsrc, g * 16 + no * 64, (int)(off * 4u), 0));
}

template <int NT>
__device__ __forceinline__ void nf_read_b(f32x4 (&b)[NT], const f32x4* act4, int lane, int ni) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = act4[nf_act_idx4(16 * t + c, 4 * ni + g)];
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
        else if ((n) == 5) __builtin_amdgcn_sched_group_barrier(mask, 5, 0);           \
        else if ((n) == 6) __builtin_amdgcn_sched_group_barrier(mask, 6, 0);           \
    } while (0)

// What else rides in a K loop.  The inference kernel: nothing.  The training forward: the deferred copy of the slab the loop reads
// (the previous layer's output) to `saved`, PER instructions per step -- the head chunk is step 0, loop iteration j is step j; the
// rows of step j + 1 are read from LDS during step j and stored under the first chunk of step j + 1 (NfSlabCopy, nf_mlp_dev.h:
// whole 128-byte lines, rows past n_points dropped by the descriptor's range check).  (nch / 2) * PER must equal the
// 16 NT * W4 / 64 instructions the slab takes.  The slab holds the accumulators as they were (see nf_tail): RELU applies the
// activation on the way out, so that `saved` has what the weight-gradient GEMMs read.
struct NfNoCopy {
    static constexpr int N = 0;
    __device__ __forceinline__ void stores(int) const {}
    __device__ __forceinline__ void reads(int) {}
};

template <int W4, int PER, bool RELU>
struct NfCopyH {
    const f32x4* act4;
    NfSlabCopy cp;
    int lane, n_step;
    f32x4 cv[PER];
    static constexpr int N = PER;
    __device__ __forceinline__ void prime() {
#pragma unroll
        for (int k = 0; k < PER; ++k) cv[k] = nf_copy_read<W4>(act4, k, lane);
    }
    __device__ __forceinline__ void stores(int step) const {
#pragma unroll
        for (int k = 0; k < PER; ++k) nf_copy_write<W4>(RELU ? nf_relu_i(cv[k]) : cv[k], cp, step * PER + k, lane);
    }
    __device__ __forceinline__ void reads(int step) {                    // the rows of step + 1; past the last step: re-read (unused)
        const int nx = step + 1 < n_step ? step + 1 : step;
#pragma unroll
        for (int k = 0; k < PER; ++k) cv[k] = nf_copy_read<W4>(act4, nx * PER + k, lane);
    }
};

// [x > 0] bits of a consumed fragment into the lane's 64-bit mask word of the layer that produced it: bit 4 ni + r of lane (g, c)
// <-> feature 16 ni + 4 g + r of point 16 t + c (the order nf_relu_with_mask defined and the dX chain reads).  `b` is the fragment
// after the ReLU: as integers, min(b, 1) is the bit.
template <int NT>
__device__ __forceinline__ void nf_mask_bits(uint64_t (&m64)[NT], const f32x4 (&b)[NT], int ni) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        uint32_t nib = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t y = (uint32_t)__float_as_int(b[t][r]);
            nib |= (y < 1u ? y : 1u) << r;                                  // v_min_u32, v_lshl_or_b32
        }
        m64[t] |= (uint64_t)nib << (4 * ni);
    }
}

// one half-iteration of the pipeline: the MFMAs of a chunk (w, fragment raw -- ReLU applied here if RELU_IN) with the fetches of
// the chunk after it (weights -> wn, fragment -> rawn) placed among them: the LDS reads first, one weight load per output tile;
// ST / RD: the copy's stores / LDS reads of `step` ride along; MASK: the consumed fragment's ReLU bits are collected in m64
template <int NT, int NO, bool FIRST, bool RELU_IN, bool MASK, bool ST, bool RD, class Side>
__device__ __forceinline__ void nf_half(f32x4 (&acc)[NT][16], const f32x4 (&w)[16], const f32x4 (&raw)[NT], f32x4 (&wn)[16], f32x4 (&rawn)[NT],
                                        const f32x4 (&bias)[16], const NfW& W, unsigned wsrc, const f32x4* act4, int lane, int ni_next, Side& side,
                                        int step, uint64_t (&m64)[NT], f32x4 (&bp)[NT]) {
    constexpr int NS = ST ? Side::N : 0, NR = RD ? Side::N : 0;
    constexpr bool onewait = NF_ONEWAIT != 0 && Side::N == 0 && !MASK && NO >= 2;
    if (onewait) NF_WAIT_VM0();                   // this chunk's weights (and, FIRST, the bias): requested at least half a chunk ago
    f32x4 b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = RELU_IN ? nf_relu_i(raw[t]) : raw[t];
    // mask bits: all taken in the half-iterations that carry the copy's stores (their vector work interleaves with the first MFMAs
    // behind the LDS reads); the other half only parks its fragment -- with its own LDS reads and their addresses in front, its bit
    // arithmetic ended up ahead of everything, in front of an idle matrix pipe
    if (MASK && ST) {
        if (!RD) nf_mask_bits<NT>(m64, bp, ni_next - 2);            // (the head chunk, ST and RD, has no predecessor)
        nf_mask_bits<NT>(m64, b, ni_next - 1);
    }
    if (MASK && !ST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bp[t] = b[t];
    }
    nf_read_b<NT>(rawn, act4, lane, ni_next);
    const Side rows = side;                       // (the rows this half stores: a head chunk also reads the next ones over them)
    if (RD) side.reads(step);                     // (source order = the order the groups below ask for: LDS reads, loads, stores)
    nf_load_w16<NO>(wn, W, wsrc, lane);
    if (ST) rows.stores(step);
    nf_chunk<NT, NO, FIRST>(acc, w, b, bias);
    NF_SGB_N(0x100, NT + NR);
    if (RELU_IN) __builtin_amdgcn_sched_group_barrier(0x002, 4 * NT, 0);
#pragma unroll
    for (int no = 0; no < NO; ++no) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
        if (onewait) {
            constexpr int H = (NO + 1) / 2;       // loads behind the first H tiles
            if (no < H) NF_SGB_N(0x020, (no + 1) * NO / H - no * NO / H);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        NF_SGB_N(0x040, (no + 1) * NS / NO - no * NS / NO);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The first nch - 1 of nch slab chunks of a layer (nch even).  Entry: st.wa / st.b0 hold chunk 0 (the fragment as stored).  Exit:
// chunk nch - 1 is pending in st.wb / st.b1 -- the caller runs it with nf_tail (the layer ends there) or nf_chunk, through nf_pending_b.
template <int NT, int NO, bool FIRST, bool RELU_IN, bool MASK, class Side>
__device__ __forceinline__ void nf_seg_lds(f32x4 (&acc)[NT][16], NfStream<NT>& st, const NfW& W, unsigned wsec, int nch, const f32x4* act4,
                                           int lane, Side& side, uint64_t (&m64)[NT]) {
    if (MASK) {                                   // (the first loop iteration finds no parked fragment: all zero = no bits)
#pragma unroll
        for (int t = 0; t < NT; ++t) st.bp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    nf_half<NT, NO, FIRST, RELU_IN, MASK, true, true>(acc, st.wa, st.b0, st.wb, st.b1, st.bias, W, wsec + NO * 64, act4, lane, 1, side, 0, m64, st.bp);
#pragma unroll 1
    for (int ni = 1; ni < nch - 1; ni += 2) {
        const int step = (ni + 1) >> 1;
        nf_half<NT, NO, false, RELU_IN, MASK, true, false>(acc, st.wb, st.b1, st.wa, st.b0, st.bias, W, wsec + (ni + 1) * NO * 64, act4, lane, ni + 1, side,
                                                           step, m64, st.bp);
        nf_half<NT, NO, false, RELU_IN, MASK, false, true>(acc, st.wa, st.b0, st.wb, st.b1, st.bias, W, wsec + (ni + 2) * NO * 64, act4, lane, ni + 2, side,
                                                           step, m64, st.bp);
    }
}
template <int NT, int NO, bool FIRST, bool RELU_IN>
__device__ __forceinline__ void nf_seg_lds(f32x4 (&acc)[NT][16], NfStream<NT>& st, const NfW& W, unsigned wsec, int nch, const f32x4* act4,
                                           int lane) {
    NfNoCopy none;
    uint64_t unused[NT];
    nf_seg_lds<NT, NO, FIRST, RELU_IN, false>(acc, st, W, wsec, nch, act4, lane, none, unused);
}

// the pending chunk's fragment as the MFMAs take it
template <int NT, bool RELU_IN>
__device__ __forceinline__ void nf_pending_b(f32x4 (&b)[NT], const NfStream<NT>& st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = RELU_IN ? nf_relu_i(st.b1[t]) : st.b1[t];
}

// The layer's last K chunk (weights w, fragments b) and the layer boundary under it.  NO tiles are computed, the first NO_ST of
// them go to the slab.  The next layer has NO_NEXT output tiles, weights at wnext, bias at bias_next; NEXT_B: 1 = it starts from
// the slab, 0 = from register chunks.  Leaves the next layer's first weight chunk in st.wa, its bias in st.bias and (NEXT_B) its
// first B fragment, as stored, in st.b0.
// The accumulators go to the slab as they are, straight from the accumulator registers: whoever reads the slab applies the ReLU.
template <int NT, int NO, int NO_ST, int NO_NEXT, int NEXT_B, int ONEWAIT = 0>
__device__ __forceinline__ void nf_tail(f32x4 (&acc)[NT][16], const f32x4 (&w)[16], const f32x4 (&b)[NT], NfStream<NT>& st,
                                        const NfW& W, unsigned wnext, const NfW& C, unsigned bias_next, f32x4* act4, int lane) {
    constexpr int LAG = NF_TAIL_LAG;
    constexpr int NL = 2 * NO_NEXT;                                    // prefetch loads, spread over the NO tile steps
    const int g = lane >> 4, c = lane & 15;
    __builtin_amdgcn_sched_barrier(0);
    if (ONEWAIT) NF_WAIT_VM0();
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
            if (ONEWAIT) {                            // all prefetch loads behind the first half of the tiles (see NF_ONEWAIT)
                constexpr int H = (NO + 1) / 2;
                if (no < H) NF_SGB_N(0x020, (no + 1) * NL / H - no * NL / H);
            } else {
                NF_SGB_N(0x020, (no + 1) * NL / NO - no * NL / NO);
            }
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

// ---- the backward chains' layer boundary (round 4) ----------------------------------------------------------------------------
// Round 3's dX chains ended every layer with a block of non-matrix work per wave (128 v_accvgpr_read, the mask arithmetic, 32 ds_write,
// 128 v_accvgpr_write to zero the accumulators, the first loads of the next layer) during which the matrix pipe of the SIMD idles.
// nf_tail_dz is nf_tail for them: the last K chunk with every tile written to the slab two tiles behind its last MFMA -- MASKED on the
// way (dZ needs writer-side masks: the row-wise copy of the slab to `dz` is done by other lanes than the ones that own the mask bits) --
// and the next layer's first weight chunk and B fragment requested underneath.  The chains pass zeros as st.bias: C = 0 is the C operand
// of a layer's first MFMAs (no accumulator initialisation).
template <int NT, int NO, int NO_NEXT, bool MASKED>
__device__ __forceinline__ void nf_tail_dz(f32x4 (&acc)[NT][16], const f32x4 (&w)[16], const f32x4 (&b)[NT], NfStream<NT>& st, const NfW& W,
                                           unsigned wnext, f32x4* act4, int lane, const uint2 (&m)[NT]) {
    constexpr int LAG = NF_TAIL_LAG;                                   // (lags 1, 3, 4 and no VALU group hint: all within noise, profiles/r04_experiments.md section 8)
    constexpr int NL = NO_NEXT;                                        // prefetch loads (weights only), spread over the NO tile steps
    const int g = lane >> 4, c = lane & 15;
    __builtin_amdgcn_sched_barrier(0);
    if (NO_NEXT > 0) nf_load_w16<(NO_NEXT > 0 ? NO_NEXT : 1)>(st.wa, W, wnext, lane);
    f32x4 braw[NT];
#pragma unroll
    for (int no = 0; no < NO + LAG; ++no) {
        if (no < NO) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][no] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[no][r], b[t][r], acc[t][no], 0, 0, 0);
        }
        if (no >= LAG) {
            const int q = no - LAG;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[t][q];
                if (MASKED) {
                    const uint32_t wd = q < 8 ? m[t].x : m[t].y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int keep = ((int)(wd << (31 - (4 * (q & 7) + r)))) >> 31;      // v_bfe_i32: 0 or -1
                        v[r] = __int_as_float(__float_as_int(v[r]) & keep);
                    }
                }
                act4[nf_act_idx4(16 * t + c, 4 * q + g)] = v;
            }
        }
        if (no == LAG && NO_NEXT > 0) {
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
        if (no >= LAG) {
            __builtin_amdgcn_sched_group_barrier(0x002, (MASKED ? 12 : 4) * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, NT, 0);
        }
        if (no == LAG && NO_NEXT > 0) __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (NO_NEXT > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) st.b0[t] = braw[t];
    }
}

