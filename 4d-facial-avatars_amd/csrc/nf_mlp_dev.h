// Device-side building blocks shared by the forward and backward kernels of the fused paper MLP.
#pragma once
#include "nf_common.h"
#include "nf_mlp_layout.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NF_MLP_WAVES 4      // independent waves per workgroup (one per SIMD)
#define NF_MLP_NT 2         // 16-point MFMA column tiles per wave -> 32 points per wave, 128 per workgroup

// Wave-private activation slab: [16*NT points][256 features]; the 16-byte fragment (feature/4 = q) of
// point row p is stored at float4 index p*64 + (q ^ (p & 15)): ds_read_b128 / ds_write_b128 lane
// groups then touch 16 distinct 16-byte bank slots (conflict-free, see nf_mlp_layout.h).
__device__ __forceinline__ int nf_act_idx4(int prow, int q) { return prow * 64 + (q ^ (prow & 15)); }

template <int NT, int NO>
__device__ __forceinline__ void nf_load_w(f32x4 (&w)[NO], const f32x4* __restrict__ src, int lane) {
#pragma unroll
    for (int no = 0; no < NO; ++no) w[no] = src[no * 64 + lane];
}

template <int NT, int NO>
__device__ __forceinline__ void nf_mma_chunk(f32x4 (&acc)[NT][16], const f32x4 (&w)[NO], const f32x4 (&b)[NT]) {
#pragma unroll
    for (int no = 0; no < NO; ++no)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t][no] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[no][r], b[t][r], acc[t][no], 0, 0, 0);
}

// K chunks whose B fragments come from registers (PE / dir slots); NCH is small and fully unrolled.
template <int NT, int NO, int NCH>
__device__ __forceinline__ void nf_mma_from_regs(f32x4 (&acc)[NT][16], const f32x4* __restrict__ wsec, const f32x4 (&breg)[NT][NCH],
                                                 int lane) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        f32x4 w[NO];
        nf_load_w<NT, NO>(w, wsec + (size_t)j * NO * 64, lane);
        f32x4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = breg[t][j];
        nf_mma_chunk<NT, NO>(acc, w, b);
    }
}

// K chunks whose B fragments come from the wave's LDS slab -- the TRAINING kernels' loop.  Explicit software pipeline, one
// half-iteration = one chunk:
//   half 1:  MFMAs of chunk ni   (weights wa, fragment b0)  |  loads of chunk ni + 1 -> wb,  ds_read of its fragment -> b1
//   half 2:  MFMAs of chunk ni+1 (weights wb, fragment b1)  |  loads of chunk ni + 2 -> wa,  ds_read of its fragment -> b0
// Every weight fragment is requested one chunk (128 MFMAs, 4096 cycles) before its MFMAs and every B fragment one chunk before
// its use; the sched_group_barrier sequence asks for one load per 8 MFMAs (one output tile), with the slab copy's stores and LDS
// reads (`Side`) at fixed places among them.  The loads past the last chunk re-read it (harmless): no branch, one wait schedule.
// nch must be even.
struct NfNoSide {
    __device__ __forceinline__ void half1(int) const {}
    __device__ __forceinline__ void half2(int) const {}
    static constexpr int N_STORE = 0, N_READ = 0;
};

// Weight fragments of a K chunk through a buffer descriptor over the layer's section: the per-lane part of the address is ONE VGPR
// (lane * 16 + a multiple of 1 KiB, four values kept in registers) and the chunk offset is scalar -- against a 64-bit vector address
// plus a v_add_co / v_addc pair per four loads in the global form.  In a one-wave-per-SIMD MFMA loop that vector arithmetic and the
// wider VMEM issue are not free: the f32 inference kernel went from 91.6 to 86.7 ms per fine launch on this change alone
// (profiles/r03_mlp_f32_stream.md).
typedef unsigned nf_u32x4_ __attribute__((ext_vector_type(4)));
template <int NO>
__device__ __forceinline__ void nf_load_w_buf(f32x4 (&w)[NO], __amdgpu_buffer_rsrc_t rsrc, int chunk, int lane) {
#pragma unroll
    for (int no = 0; no < NO; ++no)
        w[no] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + (no & 3) * 1024, chunk * (NO * 1024) + (no >> 2) * 4096, 0));
}

template <int NT, int NO, class Side>
__device__ __forceinline__ void nf_mma_from_lds_side(f32x4 (&acc)[NT][16], const f32x4* __restrict__ wsec, int nch, const f32x4* act4,
                                                     int lane, Side& side) {
    const int g = lane >> 4, c = lane & 15;
    f32x4 wa[NO], wb[NO], b0[NT], b1[NT];
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(wsec), (short)0, nch * (NO * 1024), 0x00020000);
#define NF_LOADW_(dst, chunk) nf_load_w_buf<NO>(dst, wr, chunk, lane)
    NF_LOADW_(wa, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) b0[t] = act4[nf_act_idx4(16 * t + c, g)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int ni = 0; ni < nch; ni += 2) {
        // ---- half 1
#pragma unroll
        for (int t = 0; t < NT; ++t) b1[t] = act4[nf_act_idx4(16 * t + c, 4 * (ni + 1) + g)];
        NF_LOADW_(wb, ni + 1);
        nf_mma_chunk<NT, NO>(acc, wa, b0);
        side.half1(ni >> 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);                              // the next fragment's LDS reads first
#pragma unroll
        for (int no = 0; no < NO; ++no) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);                      // one output tile: 4 NT MFMAs
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           // one weight load of the next chunk
            if (Side::N_STORE > 0 && no % (NO / Side::N_STORE > 0 ? NO / Side::N_STORE : 1) == 0 && no / (NO / Side::N_STORE > 0 ? NO / Side::N_STORE : 1) < Side::N_STORE)
                __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                       // (training: one store of the slab copy)
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- half 2
        const int nx = ni + 2 < nch ? ni + 2 : ni;
#pragma unroll
        for (int t = 0; t < NT; ++t) b0[t] = act4[nf_act_idx4(16 * t + c, 4 * nx + g)];
        side.half2(ni >> 1);
        NF_LOADW_(wa, nx);
        nf_mma_chunk<NT, NO>(acc, wb, b1);
        __builtin_amdgcn_sched_group_barrier(0x100, NT + Side::N_READ, 0);
#pragma unroll
        for (int no = 0; no < NO; ++no) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef NF_LOADW_
}

// The plain loop, weights register double-buffered one chunk ahead, scheduled by the compiler: the K loop of the inference kernels that
// are not layer-streamed yet (second model family, tiny_nerf, the pre-encoded entry).  The paper model's exact-f32 inference and
// training-forward kernels use nf_mlp_stream.h (layer boundaries under the MFMAs, pipelined loop on buffer loads: 94.1 -> 85.9 ms per fine
// launch, profiles/r03_mlp_f32_stream.md); on global loads this loop and the explicit pipeline above measured the same (94.03 vs 94.74 ms,
// profiles/r03_mlp_f32_pmc.md).  nch must be even.
template <int NT, int NO>
__device__ __forceinline__ void nf_mma_from_lds(f32x4 (&acc)[NT][16], const f32x4* __restrict__ wsec, int nch, const f32x4* act4, int lane) {
    const int g = lane >> 4, c = lane & 15;
    f32x4 wa[NO], wb[NO];
    nf_load_w<NT, NO>(wa, wsec, lane);
#pragma unroll 1
    for (int ni = 0; ni < nch; ni += 2) {
        nf_load_w<NT, NO>(wb, wsec + (size_t)(ni + 1) * NO * 64, lane);
        f32x4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = act4[nf_act_idx4(16 * t + c, 4 * ni + g)];
        nf_mma_chunk<NT, NO>(acc, wa, b);
        if (ni + 2 < nch) nf_load_w<NT, NO>(wa, wsec + (size_t)(ni + 2) * NO * 64, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = act4[nf_act_idx4(16 * t + c, 4 * (ni + 1) + g)];
        nf_mma_chunk<NT, NO>(acc, wb, b);
    }
}

template <int NT, int NO>
__device__ __forceinline__ void nf_init_acc(f32x4 (&acc)[NT][16], const float* __restrict__ bias, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int no = 0; no < NO; ++no) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 16 * no + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][no] = b;
    }
}

template <int NT, int NO, bool RELU>
__device__ __forceinline__ void nf_store_act(const f32x4 (&acc)[NT][16], f32x4* act4, int lane) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int no = 0; no < NO; ++no)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = acc[t][no];
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            act4[nf_act_idx4(16 * t + c, 4 * no + g)] = v;
        }
}

// Write a layer's output tiles to a row-major [n_points][width] global matrix (training: saved activations
// or pre-activation gradients).  Lane (g, c) owns 4 consecutive features of point c: one 16-byte store.
template <int NT, int NO>
__device__ __forceinline__ void nf_store_global(const f32x4 (&v)[NT][16], float* __restrict__ sec, int width, int64_t p0,
                                                int64_t n_points, int lane) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int64_t p = p0 + 16 * t + c;
        if (p < n_points) {
#pragma unroll
            for (int no = 0; no < NO; ++no) *reinterpret_cast<f32x4*>(sec + p * width + 16 * no + 4 * g) = v[t][no];
        }
    }
}

template <int NT, int NO>
__device__ __forceinline__ void nf_relu_inplace(f32x4 (&acc)[NT][16]) {
#pragma unroll
    for (int no = 0; no < NO; ++no)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = acc[t][no];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            acc[t][no] = v;
        }
}

// Positional encoding of one point in B-fragment order (see nfl::pe_slot_pair).  The pair index of slot (j, h) is base(g) + 2 j + h with
// base = 8 g (24 for g = 3), its component (base + 2 j + h) % 3: the lane-dependent part is only base % 3 = {0, 2, 1, 0}[g], so the
// coordinates are rotated by that ONCE per point (two lane-invariant masks) and every slot picks by a compile-time index.  (Selecting
// px / py / pz per slot made 32 loop-invariant lane masks, which the persistent inference kernel kept alive across its block loop:
// 38 spilled SGPRs, 72 v_readlane per block.)
__device__ __forceinline__ void nf_encode_point(float px, float py, float pz, int g, f32x4 (&pe)[4]) {
    const int rot = g == 1 ? 2 : (g == 2 ? 1 : 0);
    const float q[3] = {rot == 0 ? px : (rot == 1 ? py : pz), rot == 0 ? py : (rot == 1 ? pz : px), rot == 0 ? pz : (rot == 1 ? px : py)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pidx = (g < 3 ? g * 8 : 24) + j * 2 + h;
            const int freq = pidx / 3;
            const float x = q[(j * 2 + h) % 3];                    // component (rot + 2 j + h) % 3
            float s, cs;
            nf_sincos(nf_mul(x, (float)(1 << freq)), &s, &cs);
            v[2 * h] = s;
            v[2 * h + 1] = cs;
        }
        if (j == 3 && g == 3) { v[0] = px; v[1] = py; v[2] = pz; v[3] = 0.0f; }
        pe[j] = (f32x4){v[0], v[1], v[2], v[3]};
    }
}

// ---- backward-chain helpers (nf_mlp_bwd.hip, nf_mlp_lcode_bwd.hip, nf_tiny_bwd.hip) --------------------------------
template <int NT, int NO>
__device__ __forceinline__ void nf_zero_acc(f32x4 (&acc)[NT][16]) {
#pragma unroll
    for (int no = 0; no < NO; ++no)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][no] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// acc *= [X > 0] with X read from the saved activations ([n_points][width] row-major)
template <int NT, int NO>
__device__ __forceinline__ void nf_mask_by_saved(f32x4 (&acc)[NT][16], const float* __restrict__ sec, int width, int64_t p0,
                                                 int64_t n_points, int lane) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t p = p0 + 16 * t + c;
        if (p >= n_points) p = n_points - 1;
#pragma unroll
        for (int no = 0; no < NO; ++no) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(sec + p * width + 16 * no + 4 * g);
            f32x4 v = acc[t][no];
            v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f; v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
            acc[t][no] = v;
        }
    }
}


// =====================================================================================================================
// Training kernels of the paper model (exact f32): line-wide saves and ReLU bit masks
// =====================================================================================================================
// After nf_store_act the wave's LDS slab holds a layer's [16 NT points][width] outputs, and a point's row is contiguous in
// the row-major global matrix the weight-gradient GEMMs read.  So 64 lanes x 16 B copy one 256-wide row (1 KiB), two 128-wide
// rows or four 64-wide rows per instruction: every global store covers whole 128-byte lines (the register-direct
// nf_store_global writes 16 rows x 64 B per instruction, which the memory system takes at a third of that rate).  The copies
// are issued from inside the NEXT layer's K loop -- a few rows per iteration, under its MFMAs -- instead of as one burst at
// the layer boundary, where all 1024 waves of the chip used to queue their 32 KiB behind one another while the matrix
// pipes idled (one in-order wave per SIMD: nothing else can run).
// Stores go through a buffer descriptor that covers exactly the section: rows of points past n (the last wave's partial
// tile) fall outside it and are dropped by the hardware's range check -- no predicate, no branch in the K loop, so the LDS
// reads of the copy can be scheduled ahead of their stores like any other load.
typedef unsigned nf_u32x4 __attribute__((ext_vector_type(4)));
struct NfSlabCopy {
    __amdgpu_buffer_rsrc_t rsrc;      // [n][W] row-major section
    unsigned row0_b;                  // byte offset of the slab's first row
};
// section `sec` ([n][width] floats at base + sec * n) as a copy target for the slab that starts at point p0
__device__ __forceinline__ NfSlabCopy nf_slab_copy(float* base, int sec, int width, int64_t p0, int64_t n) {
    NfSlabCopy c;
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc(base + (int64_t)sec * n, (short)0, (int)((unsigned)n * (unsigned)(4 * width)), 0x00020000);
    c.row0_b = (unsigned)p0 * (unsigned)(4 * width);
    return c;
}

// copy instruction `inst` of a slab whose rows are W4 float4 wide (64 / W4 rows per instruction): LDS read and global store halves
template <int W4>
__device__ __forceinline__ f32x4 nf_copy_read(const f32x4* act4, int inst, int lane) {
    constexpr int RPI = 64 / W4;
    const int p = inst * RPI + lane / W4, q = lane % W4;
    return act4[nf_act_idx4(p, q)];
}
template <int W4>
__device__ __forceinline__ void nf_copy_write(const f32x4 v, const NfSlabCopy& cp, int inst, int lane) {
    constexpr int RPI = 64 / W4;
    const int p = inst * RPI + lane / W4, q = lane % W4;
    // aux 2 = nt (non-temporal): the 9 KB per point stream to HBM and are not read back before the weight-gradient kernel; as ordinary
    // write-back stores they pushed the 2 MB weight image out of L2 and delayed the in-order vmcnt of the weight loads behind them
    // (forward 2.224 -> 2.184 ms, chain 1.972 -> 1.940 ms per 262144-point launch, profiles/r03_experiments.md)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nf_u32x4, v), cp.rsrc, (int)(cp.row0_b + (unsigned)(p * W4 + q) * 16u), 0, 2);
}
template <int W4>
__device__ __forceinline__ void nf_copy_rows(const f32x4* act4, const NfSlabCopy& cp, int inst, int lane) {
    nf_copy_write<W4>(nf_copy_read<W4>(act4, inst, lane), cp, inst, lane);
}

// nf_mma_from_lds + the deferred copy of the slab it reads (the previous layer's output, W4 float4 per row): PER copy
// instructions per loop iteration; (nch / 2) * PER must equal the 16 NT * W4 / 64 instructions the slab takes.  The rows of
// iteration k are read from LDS during iteration k - 1 (PER staging registers) and stored under the first chunk's MFMAs.
template <int W4, int PER>
struct NfCopySide {
    const f32x4* act4;
    NfSlabCopy cp;
    int lane, n_it;
    f32x4 cv[PER];
    static constexpr int N_STORE = PER, N_READ = PER;
    __device__ __forceinline__ void prime() {
#pragma unroll
        for (int k = 0; k < PER; ++k) cv[k] = nf_copy_read<W4>(act4, k, lane);
    }
    __device__ __forceinline__ void half1(int it) const {
#pragma unroll
        for (int k = 0; k < PER; ++k) nf_copy_write<W4>(cv[k], cp, it * PER + k, lane);
    }
    __device__ __forceinline__ void half2(int it) {                      // past the last iteration: re-read its rows (unused)
        const int nx = it + 1 < n_it ? it + 1 : it;
#pragma unroll
        for (int k = 0; k < PER; ++k) cv[k] = nf_copy_read<W4>(act4, nx * PER + k, lane);
    }
};

template <int NT, int NO, int W4, int PER>
__device__ __forceinline__ void nf_mma_from_lds_copy(f32x4 (&acc)[NT][16], const f32x4* __restrict__ wsec, int nch, const f32x4* act4,
                                                     int lane, const NfSlabCopy& cp) {
    NfCopySide<W4, PER> side{act4, cp, lane, nch >> 1, {}};
    side.prime();
    nf_mma_from_lds_side<NT, NO>(acc, wsec, nch, act4, lane, side);
}

// ReLU bit masks of the exact-f32 kernels: section S_MASK of `saved`, [ReLU layers: 9 paper / 5 second family][ceil(n / 16) point tiles][64 lanes][2 dwords];
// bit 4 no + r of lane (g, c) <-> feature 16 no + 4 g + r of point 16 tile + c (the D-register order of the f32 MFMA tiles), i.e.
// exactly what the lane holds: one 8-byte store per lane and tile in the forward, one 8-byte load in the backward chain.
// (The split-bf16/fp16 kernels keep their own bit order in the same section; a `saved` buffer goes back to the family that wrote it.)
template <int S_MASK_SEC = nfl::S_MASK>       // section offset of the model family's mask words (floats per point)
__device__ __forceinline__ uint2* nf_mask_ptr(float* saved, int64_t n, int layer, int64_t tile, int lane) {
    const int64_t n_tiles = (n + 15) >> 4;
    return reinterpret_cast<uint2*>(saved + (int64_t)S_MASK_SEC * n) + ((int64_t)layer * n_tiles + tile) * 64 + lane;
}

// acc = max(acc, 0); returns the mask bits of the lane ([x > 0], exact also for +-0: the int view of a float is > 0 iff the float is)
template <int NT, int NO>
__device__ __forceinline__ void nf_relu_with_mask(f32x4 (&acc)[NT][16], uint2 (&m)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        uint32_t w[2] = {0u, 0u};
#pragma unroll
        for (int no = 0; no < NO; ++no) {
            f32x4 v = acc[t][no];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xi = __float_as_int(v[r]);
                const int bit = xi > 1 ? 1 : (xi < 0 ? 0 : xi);                   // v_med3_i32(xi, 0, 1)
                w[no >> 3] |= (uint32_t)bit << (4 * (no & 7) + r);
                v[r] = fmaxf(v[r], 0.f);
            }
            acc[t][no] = v;
        }
        m[t] = make_uint2(w[0], w[1]);
    }
}

// acc *= mask bits
template <int NT, int NO>
__device__ __forceinline__ void nf_apply_mask(f32x4 (&acc)[NT][16], const uint2 (&m)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int no = 0; no < NO; ++no) {
            const uint32_t w = no < 8 ? m[t].x : m[t].y;
            f32x4 v = acc[t][no];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int keep = ((int)(w << (31 - (4 * (no & 7) + r)))) >> 31;     // v_bfe_i32: 0 or -1
                v[r] = __int_as_float(__float_as_int(v[r]) & keep);
            }
            acc[t][no] = v;
        }
}
