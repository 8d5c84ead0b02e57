// Layout of the fragment-ordered weight image ("packed") and of the per-call bias table ("cond")
// shared by the pack / condition / forward / backward kernels of the fused paper MLP.
//
// MFMA: v_mfma_f32_16x16x4_f32, D[16 out-features x 16 points] += A[16 x 4] * B[4 x 16]
//   lane l = (g = l>>4, i = l&15):  A operand = W[n0+i][k_g],  B operand = act[point i][k_g],
//   D regs r=0..3 = out-feature n0 + 4g + r of point i.
// The K dimension is consumed in 16-wide chunks; inside chunk `ni`, MFMA step r (0..3) takes from lane
// group g the K slot 16*ni + 4*g + r.  So a lane's four A (or B) values of one chunk are 4 consecutive
// K slots = one 16-byte load, and -- because D's row index is 4g+r too -- the output registers of one
// layer already ARE the B operands of the next layer (slot n == feature n): activations only ever move
// as whole 16-byte fragments, never element-wise.
//
// A fragment block (ni, no) = 64 lanes x 4 floats = 1 KiB, contiguous; a layer section is
// [ni][no][lane][4], so one K-chunk of a layer is NO consecutive KiB (perfectly coalesced wave loads).
#pragma once

namespace nfl {

// ---- MFMA layers -------------------------------------------------------------------------------
//                       K chunks                        N tiles
// L0  (layers_xyz.0)    4  (PE slots)                   16
// L1,L2,L4,L5,FEAT      16 (hidden)                     16
// L3  (layers_xyz.3)    4 (PE) + 16 (hidden)            16
// D0  (layers_dir.0)    16 (feat) + 1 (dir slots)       8 + 1 (tile 8, row 0 = fc_alpha)
// D1,D2                 8                               8
// RGB (fc_rgb)          8                               1 (rows 0..2)
constexpr int FRAG = 256;                       // floats per (ni, no) fragment block
constexpr int OFF_L0 = 0;
constexpr int OFF_L1 = OFF_L0 + 4 * 16 * FRAG;
constexpr int OFF_L2 = OFF_L1 + 16 * 16 * FRAG;
constexpr int OFF_L3 = OFF_L2 + 16 * 16 * FRAG;
constexpr int OFF_L4 = OFF_L3 + 20 * 16 * FRAG;
constexpr int OFF_L5 = OFF_L4 + 16 * 16 * FRAG;
constexpr int OFF_FEAT = OFF_L5 + 16 * 16 * FRAG;
constexpr int OFF_D0 = OFF_FEAT + 16 * 16 * FRAG;
constexpr int OFF_D1 = OFF_D0 + 17 * 9 * FRAG;
constexpr int OFF_D2 = OFF_D1 + 8 * 8 * FRAG;
constexpr int OFF_RGB = OFF_D2 + 8 * 8 * FRAG;
constexpr int FRAG_END = OFF_RGB + 8 * 1 * FRAG;

// ---- conditioning matrices (row-major) -----------------------------------------------------------
constexpr int NCOND = 108;                      // 76 expression + 32 latent columns
constexpr int OFF_WC0 = FRAG_END;               // [256][108] = layers_xyz.0.weight[:, 63:171]
constexpr int OFF_WC3 = OFF_WC0 + 256 * NCOND;  // [256][108] = layers_xyz.3.weight[:, 63:171]
constexpr int OFF_WCD = OFF_WC3 + 256 * NCOND;  // [128][16]  = layers_dir.0.weight[:, 256+6f+3sc+{1,2}], col = 4f+2sc+(comp-1)
constexpr int OFF_BIAS = OFF_WCD + 128 * 16;    // un-folded bias table, same layout as `cond`
// ---- bias table / cond layout ----------------------------------------------------------------------
constexpr int B_L0 = 0, B_L1 = 256, B_L2 = 512, B_L3 = 768, B_L4 = 1024, B_L5 = 1280, B_FEAT = 1536;
constexpr int B_D0 = 1792;                      // 128 + 16 (alpha tile: [fc_alpha.bias, 0 x 15])
constexpr int B_D1 = B_D0 + 144, B_D2 = B_D1 + 128, B_RGB = B_D2 + 128;   // rgb tile: [b_r, b_g, b_b, 0 x 13]
constexpr int BIAS_FLOATS = B_RGB + 16;         // 2208 (the bias table proper)
constexpr int B_CVEC = BIAS_FLOATS;             // cond only: [expr*1/3 (76) | latent (32)] as the kernels used it
constexpr int B_DVEC = B_CVEC + NCOND;          // cond only: PE4 of (near, far): index 4f + 2sc + (0: near, 1: far)
constexpr int COND_FLOATS = B_DVEC + 16;        // 2332
// layers_dir.0 for PRE-ENCODED inputs (model.forward(x87, ...), nf_mlp_encoded.hip): 16 feat chunks + 2 chunks holding the
// 24 reference direction columns 256..279 in reference order (slot 256 + s <-> column 256 + s, s < 24), 9 tiles as OFF_D0.
constexpr int OFF_D0E = OFF_BIAS + BIAS_FLOATS;
constexpr int PACKED_FLOATS = OFF_D0E + 18 * 9 * FRAG;

// ---- training: activations saved by the forward, floats per point ---------------------------------------
// Section X of a buffer for n points starts at X * n and is an [n][width] row-major matrix.
constexpr int S_PE = 0;                          // 64, PE slot order (see pe_slot_to_col)
constexpr int S_H0 = 64, S_H1 = 320, S_H2 = 576, S_H3 = 832, S_H4 = 1088, S_H5 = 1344;   // 256 each, post-ReLU
constexpr int S_FEAT = 1600;                     // 256, fc_feat output
constexpr int S_D0 = 1856, S_D1 = 1984, S_D2 = 2112;                                     // 128 each, post-ReLU
constexpr int S_DIRF = 2240;                     // 16, dir slot order: (sin, cos, 0, 0)(rd_z 2^g), g = 0..3
constexpr int S_MASK = 2256;                     // split-bf16 training forward only: ReLU bit masks, 9 layers (h0..h5, layers_dir.0..2)
                                                 // x [n points] x [2 lane halves] x 4 dwords; bit 16 nt + r of half h <-> feature
                                                 // 32 nt + (r&3) + 8 (r>>2) + 4 h  (the D-register order of the bf16 kernels)
constexpr int SAVED_PER_POINT = 2256 + 9 * 8;
// ---- training: pre-activation gradients written by the backward chain, floats per point ------------------
constexpr int Z_L0 = 0, Z_L1 = 256, Z_L2 = 512, Z_L3 = 768, Z_L4 = 1024, Z_L5 = 1280, Z_FEAT = 1536;
constexpr int Z_D0 = 1792, Z_D1 = 1920, Z_D2 = 2048;
constexpr int DZ_PER_POINT = 2176;

// ---- PE slot permutation ---------------------------------------------------------------------------
// PE slot s = 16*j + 4*g + r (chunk j, lane group g, step r).  Lane groups 0..2 hold 8 (freq, comp)
// pairs per point, as (sin, cos) register pairs, so that ONE sincosf feeds two slots; group 3 holds
// pairs 24..29, the raw xyz (chunk 3, r = 0..2) and the zero pad (chunk 3, r = 3).
//   pair index pidx -> freq = pidx / 3, comp = pidx % 3;  reference column = 3 + 6*freq + 3*sc + comp
//   (reference layout [x y z | sin f0 xyz | cos f0 xyz | sin f1 xyz | ...], nerf_helpers.py:231-239).
__host__ __device__ inline int pe_slot_pair(int j, int g, int r) {           // -1: not a sin/cos slot
    if (g < 3) return g * 8 + j * 2 + (r >> 1);
    return j < 3 ? 24 + j * 2 + (r >> 1) : -1;
}
__host__ __device__ inline int pe_slot_to_col(int slot) {                     // -1: zero pad
    const int j = slot >> 4, g = (slot >> 2) & 3, r = slot & 3;
    const int pidx = pe_slot_pair(j, g, r);
    if (pidx >= 0) return 3 + 6 * (pidx / 3) + 3 * (r & 1) + (pidx % 3);
    return r < 3 ? r : -1;
}

// inverse of pe_slot_to_col: reference PE column (0..62) -> slot
__host__ __device__ inline int pe_col_to_slot(int col) {
    if (col < 3) return 16 * 3 + 4 * 3 + col;                       // raw xyz: chunk 3, group 3, r = col
    const int q = col - 3, freq = q / 6, rem = q - 6 * freq, sc = rem / 3, comp = rem - 3 * sc;
    const int pidx = 3 * freq + comp;
    int g, j, h;
    if (pidx < 24) { g = pidx >> 3; j = (pidx & 7) >> 1; h = pidx & 1; }
    else { g = 3; j = (pidx - 24) >> 1; h = (pidx - 24) & 1; }
    return 16 * j + 4 * g + 2 * h + sc;
}

}  // namespace nfl

namespace nfl {
// ---- transposed fragment image for the backward chain dX = dZ . W ("packed_t") ------------------------
// Section layout [ni][no][lane][4] as in the forward image, but a block (ni, no) now holds
// W[16*ni + 4*g + r][col0 + 16*no + i]: the reduction runs over the layer's OUTPUT features.
//                       reduction chunks                 output tiles
// T_RGB  fc_rgb         1 (slots 0..2 = d rgb)           8   -> d(layers_dir.2 out)
// T_D2, T_D1            8                                8
// T_D0   layers_dir.0   8 + 1 (slot 0 = d sigma, fc_alpha.weight)   16  -> d feat   (columns 0..255 only)
// T_FEAT, T_L5, T_L4, T_L3 (columns 171..426), T_L2, T_L1          16 x 16
constexpr int OFFT_RGB = 0;
constexpr int OFFT_D2 = OFFT_RGB + 1 * 8 * FRAG;
constexpr int OFFT_D1 = OFFT_D2 + 8 * 8 * FRAG;
constexpr int OFFT_D0 = OFFT_D1 + 8 * 8 * FRAG;
constexpr int OFFT_FEAT = OFFT_D0 + 9 * 16 * FRAG;
constexpr int OFFT_L5 = OFFT_FEAT + 16 * 16 * FRAG;
constexpr int OFFT_L4 = OFFT_L5 + 16 * 16 * FRAG;
constexpr int OFFT_L3 = OFFT_L4 + 16 * 16 * FRAG;
constexpr int OFFT_L2 = OFFT_L3 + 16 * 16 * FRAG;
constexpr int OFFT_L1 = OFFT_L2 + 16 * 16 * FRAG;
constexpr int PACKED_T_FLOATS = OFFT_L1 + 16 * 16 * FRAG;

// ---- per-slice partial-gradient slab written by the weight-gradient GEMMs --------------------------------
constexpr int G_L0 = 0;                          // [256][64]   (PE slot order)
constexpr int G_L1 = G_L0 + 256 * 64;            // [256][256]
constexpr int G_L2 = G_L1 + 65536;
constexpr int G_L3A = G_L2 + 65536;              // [256][64]   (PE slot order)
constexpr int G_L3B = G_L3A + 256 * 64;          // [256][256]  (hidden part, reference columns 171..426)
constexpr int G_L4 = G_L3B + 65536;
constexpr int G_L5 = G_L4 + 65536;
constexpr int G_FEAT = G_L5 + 65536;
constexpr int G_D0A = G_FEAT + 65536;            // [128][256]
constexpr int G_D0B = G_D0A + 128 * 256;         // [128][16]   (dir slot order)
constexpr int G_D1 = G_D0B + 128 * 16;           // [128][128]
constexpr int G_D2 = G_D1 + 128 * 128;
constexpr int G_RGB = G_D2 + 128 * 128;          // [16][128]   rows 0..2 = fc_rgb.weight grad
constexpr int G_ALPHA = G_RGB + 16 * 128;        // [16][256]   row 3 (the d sigma column of d_raw) = fc_alpha.weight grad
constexpr int CS_L0 = G_ALPHA + 16 * 256;        // column sums of dZ = bias grads: 7 x 256
constexpr int CS_D0 = CS_L0 + 7 * 256;           // 3 x 128
constexpr int CS_RGB = CS_D0 + 3 * 128;          // 16: [d b_r, d b_g, d b_b, d b_alpha, 0...]
constexpr int SLAB_FLOATS = CS_RGB + 16;

constexpr int GRAD_PARAM_FLOATS = 568708;        // the 26 tensors, state_dict order, flattened
constexpr int GRAD_FLOATS = GRAD_PARAM_FLOATS + 32;   // + d latent
}  // namespace nfl
