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

// K chunks whose B fragments come from the wave's LDS slab; weights are register double-buffered one
// chunk ahead.  nch must be even.
template <int NT, int NO>
__device__ __forceinline__ void nf_mma_from_lds(f32x4 (&acc)[NT][16], const f32x4* __restrict__ wsec, int nch,
                                                const f32x4* act4, int lane) {
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

// Positional encoding of one point in B-fragment order (see nfl::pe_slot_pair).
__device__ __forceinline__ void nf_encode_point(float px, float py, float pz, int g, f32x4 (&pe)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pidx = (g < 3 ? g * 8 : 24) + j * 2 + h;
            const int freq = pidx / 3, comp = pidx - 3 * freq;
            const float x = comp == 0 ? px : (comp == 1 ? py : pz);
            float s, cs;
            sincosf(nf_mul(x, (float)(1 << freq)), &s, &cs);
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

