// Layouts of the second model family (ConditionalBlendshapeLearnableCodeNeRFModel, reference nerf/models.py:529-636):
// packed weight image, per-call bias table, and -- training -- saved activations, pre-activation gradients, transposed
// image and gradient slab.  Same conventions as nf_mlp_layout.h (fragment order, PE slot order, section-major buffers).
#pragma once
#include "nf_mlp_layout.h"

namespace nlc {
constexpr int FRAG = 256;
constexpr int OFF_L1 = 0;                              // 4 PE chunks x 16 tiles
constexpr int OFF_X0 = OFF_L1 + 4 * 16 * FRAG;         // 16 x 16 each
constexpr int OFF_X1 = OFF_X0 + 16 * 16 * FRAG;
constexpr int OFF_X2 = OFF_X1 + 16 * 16 * FRAG;
constexpr int OFF_ALPHA = OFF_X2 + 16 * 16 * FRAG;     // 16 chunks x 1 tile (row 0)
constexpr int OFF_FEAT = OFF_ALPHA + 16 * 1 * FRAG;
constexpr int OFF_DIR = OFF_FEAT + 16 * 16 * FRAG;     // 16 feat chunks + 1 dir chunk, 8 tiles
constexpr int OFF_RGB = OFF_DIR + 17 * 8 * FRAG;       // 8 chunks x 1 tile (rows 0..2)
constexpr int OFF_WC1 = OFF_RGB + 8 * 1 * FRAG;        // [256][108] layer1.weight[:, 63:171]
constexpr int OFF_WCD = OFF_WC1 + 256 * 108;           // [128][16]  layers_dir.0.weight[:, 256+6f+3sc+{1,2}]
constexpr int OFF_BIAS = OFF_WCD + 128 * 16;
constexpr int B_L1 = 0, B_X0 = 256, B_X1 = 512, B_X2 = 768, B_FEAT = 1024, B_ALPHA = 1280, B_DIR = 1296, B_RGB = 1424;
constexpr int BIAS_FLOATS = 1440;
constexpr int B_CVEC = BIAS_FLOATS, B_DVEC = B_CVEC + 108, COND_FLOATS = B_DVEC + 16;
// layers_dir.0 for PRE-ENCODED inputs (model.forward(x87, ...), nf_lcode_forward_encoded): 16 feat chunks + 2 chunks holding the 24
// reference direction columns 256..279 in reference order (slot 256 + s <-> column 256 + s, s < 24), 8 tiles (cf. nfl::OFF_D0E)
constexpr int OFF_DIRE = OFF_BIAS + BIAS_FLOATS;
constexpr int PACKED = OFF_DIRE + 18 * 8 * FRAG;
constexpr int NPARAMS = 16;   // layer1, layers_xyz.0..2, layers_dir.0, fc_alpha, fc_rgb, fc_feat (weight, bias each)

// ---- training: activations saved by the forward, floats per point (section X of an n-point buffer starts at X * n) ----
constexpr int S_PE = 0;                                   // 64, PE slot order
constexpr int S_L1 = 64;                                  // 256, layer1 output (no activation)
constexpr int S_X0 = 320, S_X1 = 576, S_X2 = 832;         // 256 each, post-ReLU
constexpr int S_FEAT = 1088;                              // 256, relu(fc_feat)
constexpr int S_DIR = 1344;                               // 128, relu(layers_dir.0)
constexpr int S_DIRF = 1472;                              // 16, dir slot order (sin, cos, 0, 0)(rd_z 2^g)
constexpr int S_MASK = 1488;                              // split-bf16 training forward only: ReLU bit masks of layers_xyz.0..2, fc_feat,
                                                          // layers_dir.0 x [n points] x [2 lane halves] x 4 dwords (as nfl::S_MASK)
constexpr int SAVED_PER_POINT = 1488 + 5 * 8;
// ---- pre-activation gradients written by the backward chain ------------------------------------------------------
constexpr int Z_L1 = 0, Z_X0 = 256, Z_X1 = 512, Z_X2 = 768, Z_FEAT = 1024, Z_DIR = 1280;
constexpr int DZ_PER_POINT = 1408;
// ---- transposed image for the backward chain: block (ni, no), lane (g, i), r -> W[16 ni + 4 g + r][16 no + i] ----
constexpr int OFFT_RGB = 0;                               // 1 chunk (3 rows) x 8 tiles
constexpr int OFFT_DIR = OFFT_RGB + 1 * 8 * FRAG;         // layers_dir.0[:, :256]: 8 chunks x 16 tiles
constexpr int OFFT_FEAT = OFFT_DIR + 8 * 16 * FRAG;       // fc_feat: 16 x 16, then chunk 16: slot 0 = fc_alpha.weight
constexpr int OFFT_X2 = OFFT_FEAT + 17 * 16 * FRAG;
constexpr int OFFT_X1 = OFFT_X2 + 16 * 16 * FRAG;
constexpr int OFFT_X0 = OFFT_X1 + 16 * 16 * FRAG;
constexpr int PACKED_T = OFFT_X0 + 16 * 16 * FRAG;
// ---- gradient slab (per point slice), then the flat gradient vector in hip_param_list order + d latent -------------
constexpr int G_L1 = 0;                                   // [256][64]  PE slot order
constexpr int G_X0 = G_L1 + 256 * 64;                     // [256][256]
constexpr int G_X1 = G_X0 + 65536;
constexpr int G_X2 = G_X1 + 65536;
constexpr int G_FEAT = G_X2 + 65536;
constexpr int G_DIRA = G_FEAT + 65536;                    // [128][256]
constexpr int G_DIRB = G_DIRA + 128 * 256;                // [128][16]  dir slot order
constexpr int G_RGB = G_DIRB + 128 * 16;                  // [16][128]  rows 0..2 = fc_rgb.weight grad
constexpr int G_ALPHA = G_RGB + 16 * 128;                 // [16][256]  row 3 (d sigma) = fc_alpha.weight grad
constexpr int CS_L1 = G_ALPHA + 16 * 256;                 // column sums of dZ: layer1, layers_xyz.0..2, fc_feat (5 x 256)
constexpr int CS_DIR = CS_L1 + 5 * 256;                   // 128
constexpr int CS_RGB = CS_DIR + 128;                      // 16: [d b_r, d b_g, d b_b, d b_alpha, 0...]
constexpr int SLAB_FLOATS = CS_RGB + 16;
constexpr int GRAD_PARAM_FLOATS = 256 * 171 + 256 + 3 * (65536 + 256) + 128 * 280 + 128 + 256 + 1 + 384 + 3 + 65536 + 256;
constexpr int GRAD_FLOATS = GRAD_PARAM_FLOATS + 32;
}  // namespace nlc
