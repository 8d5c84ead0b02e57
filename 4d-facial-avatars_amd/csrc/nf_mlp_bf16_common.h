// Shared declarations of the split-bf16 kernels (stream layout tables).
#pragma once
#include "nf_common.h"
#include "nf_mlp_layout.h"

// Element type of the split operands.  NFB_F16 = 0: bf16 pairs (x = hi + lo keeps 16 significand bits, f32's exponent range);
// NFB_F16 = 1: fp16 pairs (22 significand bits -- fp32-class accuracy at the same three MFMAs per product; the narrow fp16
// exponent range is handled by a per-layer power-of-two weight scale chosen at pack time, see nf_mlp_f16.hip).
#ifndef NFB_F16
#define NFB_F16 0
#endif
#if NFB_F16
typedef _Float16 nfb_elt;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));      // (the name is historical: 8 split-operand elements of type nfb_elt)
#define NFB_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 nfb_elt;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define NFB_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace nfb {
// layer table of the bf16 stream: k-steps (16 slots each, always even) and 32-row output tiles
constexpr int NL = 11;
constexpr int KS[NL] = {4, 16, 16, 20, 16, 16, 16, 18, 8, 8, 8};   // multiples of the stage depth (2 k-steps); layers_dir.0: 16 feat + dir + ONE zero k-step (round 5: 3)
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

