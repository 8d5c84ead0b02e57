// Layer table of the split-bf16 kernels of the second model family (forward stream), shared by the inference and the
// training translation units (nf_mlp_lcode_bf16.hip, nf_mlp_lcode_bf16_train.hip).
#pragma once
#include "nf_common.h"
#include "nf_mlp_lcode_layout.h"

// element type of the split operands: bf16 pairs, or (NFB_F16 = 1, nf_mlp_lcode_f16.hip) fp16 pairs -- see nf_mlp_bf16_common.h
#ifndef NFB_F16
#define NFB_F16 0
#endif
#if NFB_F16
typedef _Float16 nfb_elt;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
#define NFB_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 nfb_elt;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define NFB_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace nfb {
constexpr int NL = 8;
constexpr int KS[NL] = {4, 16, 16, 16, 16, 16, 18, 8};   // layers_dir.0: 16 feat k-steps + the dir k-step + ONE zero k-step (a stage is 2 k-steps; round 5: 3)
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

