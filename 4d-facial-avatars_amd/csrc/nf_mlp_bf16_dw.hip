// B2, split-bf16 variant: the weight-gradient GEMMs  dW = dZ^T . X  on the bf16 matrix pipe at fp32-class accuracy.
//
// Same inputs and outputs as the exact-f32 k_dw_gemm (nf_mlp_dw.h): dZ sections [n_points][width] written by the backward
// chain, d_raw [n_points][4], the activations saved by the training forward, and per-slice 594k-float slabs of partial
// dW / column sums that B3 reduces.  What changes is the arithmetic and the data movement:
//   * every operand value x is split into bf16 (hi, lo) and each 32x32 output tile takes three
//     v_mfma_f32_32x32x16_bf16 per 16 points (hi.hi + hi.lo + lo.hi, f32 accumulate) instead of sixteen f32 MFMAs;
//   * at that rate the kernel is HBM-bound (every dZ / activation byte has to be read once: 17.7 KB per point), so a
//     WORKGROUP (16 waves) owns a whole 256 x 256 product -- or a bundle of smaller ones with a similar byte count --
//     and reads each operand ONCE per point; each wave keeps a 64 x 64 block of the output in 64 accumulators
//     (4 waves per SIMD, 128 registers each);
//   * software pipeline over 16-point stages (one MFMA k-step), one workgroup barrier per stage:
//       L(i+2)  each wave loads the raw f32 values of "its" 32-feature operand tile straight into registers, already
//               transposed: lane (h, c) reads feature c of points 8 h .. 8 h + 7 (8 dword loads, two full 128-byte
//               lines per wave instruction); two stages (64 KB per CU) stay in flight, which is what it takes to
//               keep HBM busy -- tails and narrow sections are zero-filled here,
//       C(i)    the tile is split ONCE into the MFMA fragment pair (hi, lo) and written to LDS lane-linearly
//               (the MFMA contraction index is the point); bias gradients (column sums of dZ) are accumulated here
//               in f32,
//       M(i-1)  2 + 2 fragment pairs per wave (conflict-free ds_read_b128), 12 MFMAs, branch-free.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

// NFB_F16 = 1 (nf_mlp_f16_dw.hip includes this file): the same kernel on fp16 operand pairs.  Saved activations are converted
// times 2^4 and every gradient section times its own power of two, from max |dz| of the section that the backward chain leaves in
// `gscale` (float bits; slot = NfbDwSeg::gs), so that
// both sit inside fp16's exponent range; the accumulators are multiplied by 1 / (16 G) before they go to the slab.
#include "nf_mlp_bf16_common.h"
#include "nf_mlp_lcode_layout.h"
#if NFB_F16
#define NFB_DW_NAME(x) x##_f16
#else
#define NFB_DW_NAME(x) x##_bf16
#endif

struct NfbDwSeg {
    int kind;      // 0: dz section, 1: d_raw, 2: saved section in f32 rows (PE, dir slots), 3: saved section as (hi, lo) FRAGMENT STREAM
    int sec;       // section offset (floats per point)
    int width;     // floats per point
    int gs;        // kinds 0, 1: slot of the section's max |gradient| in the chain's table (fp16 instantiation)
};
struct NfbDwTile {   // one 32-feature operand tile of a stage
    int seg, f0;     // segment, first feature
    int cs_off;      // >= 0: slab offset of the column sums of these features (bias gradients)
};
struct NfbDwProd {   // one wave: (na <= 2 row tiles of dZ) x (nb <= 2 column tiles of activations), consecutive tile ids
    int na, nb, a_tile, b_tile;
    int a_valid, b_valid;           // valid rows / columns (narrow sections)
    int out_off, ldo;               // slab offset of (row 0, column 0), row stride
};
#define NFB_DW_WAVES 16
#define NFB_DW_MAX_TILES NFB_DW_WAVES                    // one operand tile per wave and stage
struct NfbDwJob {
    int nseg, ntile;
    NfbDwSeg seg[4];
    NfbDwTile tile[NFB_DW_MAX_TILES];
    NfbDwProd prod[NFB_DW_WAVES];
};

#define NFB_DW_JOBS 11                                   // paper model
#define NFB_DW_JOBS_LCODE 8                              // second model family
#define NFB_DW_PTS 16                                    // points per stage = one MFMA k-step
#define NFB_DW_NSET 3                                    // register sets of raw tiles: NSET - 1 stages of loads in flight
#define NFB_DW_CVT_U4 (NFB_DW_MAX_TILES * 128)           // 16-byte units: per tile 64 lanes x (hi, lo)
#define NFB_DW_RING 4                                    // fragment buffers: stage i lives in buffer i % 4 (a streamed tile is DMA'd two stages ahead)
// K-slot j of lane half h of a 16-point stage <-> point 4 h + (j & 3) + 8 (j >> 2): the order in which the forward's transposing MFMA
// leaves the points of a fragment block (NfbStreamSide).  Both operands of a product use it, so the sum over points is unchanged.
__host__ __device__ constexpr int nfb_dw_point_of_slot(int h, int j) { return 4 * h + (j & 3) + 8 * (j >> 2); }
static __constant__ NfbDwJob c_dwb_jobs[NFB_DW_JOBS];
static __constant__ NfbDwJob c_dwb_jobs_lcode[NFB_DW_JOBS_LCODE];

// job-table builder helpers (host)
struct NfbDwBuilder {
    NfbDwJob* jobs;
    int (*slot_of)(int zsec);          // dz section -> the chain's layer index (slot of max |dz|)
    int nj = 0;
    int first_tile[4];
    NfbDwJob& new_job() {
        NfbDwJob& j = jobs[nj++];
        j.nseg = j.ntile = 0;
        for (auto& s : j.seg) s = NfbDwSeg{0, 0, 0, -1};
        for (auto& t : j.tile) t = NfbDwTile{0, 0, -1};
        for (auto& p : j.prod) p = NfbDwProd{0, 0, 0, 0, 0, 0, 0, 0};   // idle wave: multiplies tiles 0, 1 and stores nothing
        return j;
    }
    // segment + its tiles; cs >= 0: slab offset of the column sums of the section
    int add_seg(NfbDwJob& j, int kind, int sec, int width, int cs) {
        // the split training forwards write every hidden layer's output (the 128- and 256-wide sections) as this kernel's operand
        // fragments (nf_mlp_bf16_machinery.inc: NfbStreamSide); the positional encoding and the dir slots stay f32 rows
        if (kind == 2 && width >= 128) kind = 3;
        j.seg[j.nseg] = NfbDwSeg{kind, sec, width, kind >= 2 ? -1 : (kind == 1 ? 10 : slot_of(sec))};
        first_tile[j.nseg] = j.ntile;
        for (int f0 = 0; f0 < width; f0 += 32) j.tile[j.ntile++] = NfbDwTile{j.nseg, f0, cs >= 0 ? cs + f0 : -1};
        return j.nseg++;
    }
    // rows [a0, a0 + a_valid) of segment sa  x  columns [b0, b0 + b_valid) of segment sb   (a0, b0 multiples of 32)
    NfbDwProd prod(int sa, int a0, int a_valid, int sb, int b0, int b_valid, int out_off, int ldo) const {
        return NfbDwProd{(a_valid + 31) / 32, (b_valid + 31) / 32, first_tile[sa] + a0 / 32, first_tile[sb] + b0 / 32,
                         a_valid, b_valid, out_off, ldo};
    }
    // a 256 x 256 layer: 4 x 4 blocks of 64 x 64
    void layer256(int zsec, int bsec, int gout, int cs) {
        NfbDwJob& j = new_job();
        const int sa = add_seg(j, 0, zsec, 256, cs), sb = add_seg(j, 2, bsec, 256, -1);
        for (int w = 0; w < 16; ++w) {
            const int ag = w >> 2, bg = w & 3;
            j.prod[w] = prod(sa, 64 * ag, 64, sb, 64 * bg, 64, gout + 64 * ag * 256 + 64 * bg, 256);
        }
    }
    // a 256-row dZ section against the 64 positional-encoding slots
    void pe256(int zsec, int pesec, int gout, int cs) {
        NfbDwJob& j = new_job();
        const int sz = add_seg(j, 0, zsec, 256, cs), sp = add_seg(j, 2, pesec, 64, -1);
        for (int w = 0; w < 8; ++w) j.prod[w] = prod(sz, 32 * w, 32, sp, 0, 64, gout + 32 * w * 64, 64);
    }
};

// second model family (layouts: nf_mlp_lcode_layout.h)
static void nfb_build_dw_jobs_lcode(NfbDwJob* jobs) {
    using namespace nlc;
    NfbDwBuilder b{jobs, [](int z) { return z == Z_DIR ? 0 : z == Z_FEAT ? 1 : z == Z_X2 ? 2 : z == Z_X1 ? 3 : z == Z_X0 ? 4 : 5; }};
    b.layer256(Z_X0, S_L1, G_X0, CS_L1 + 256);
    b.layer256(Z_X1, S_X0, G_X1, CS_L1 + 512);
    b.layer256(Z_X2, S_X1, G_X2, CS_L1 + 768);
    b.layer256(Z_FEAT, S_X2, G_FEAT, CS_L1 + 1024);
    b.pe256(Z_L1, S_PE, G_L1, CS_L1);
    {   // dZ_dir x feat
        NfbDwJob& j = b.new_job();
        const int sz = b.add_seg(j, 0, Z_DIR, 128, CS_DIR), sf = b.add_seg(j, 2, S_FEAT, 256, -1);
        for (int w = 0; w < 8; ++w) {
            const int ag = w >> 2, bg = w & 3;
            j.prod[w] = b.prod(sz, 64 * ag, 64, sf, 64 * bg, 64, G_DIRA + 64 * ag * 256 + 64 * bg, 256);
        }
    }
    {   // d_raw x x2: row 3 (d sigma) = fc_alpha.weight gradient (fc_alpha reads x)
        NfbDwJob& j = b.new_job();
        const int sr = b.add_seg(j, 1, 0, 4, -1), sx = b.add_seg(j, 2, S_X2, 256, -1);
        for (int bg = 0; bg < 4; ++bg) j.prod[bg] = b.prod(sr, 0, 4, sx, 64 * bg, 64, G_ALPHA + 64 * bg, 256);
    }
    {   // dZ_dir x dir features, d_raw x dir-layer output (fc_rgb.weight; the 4 output-bias gradients = column sums of d_raw)
        NfbDwJob& j = b.new_job();
        const int sz = b.add_seg(j, 0, Z_DIR, 128, -1), sd = b.add_seg(j, 2, S_DIRF, 16, -1), sr = b.add_seg(j, 1, 0, 4, CS_RGB);
        const int s2 = b.add_seg(j, 2, S_DIR, 128, -1);
        for (int ag = 0; ag < 2; ++ag) j.prod[ag] = b.prod(sz, 64 * ag, 64, sd, 0, 16, G_DIRB + 64 * ag * 16, 16);
        for (int bg = 0; bg < 2; ++bg) j.prod[2 + bg] = b.prod(sr, 0, 4, s2, 64 * bg, 64, G_RGB + 64 * bg, 128);
    }
    // b.nj == NFB_DW_JOBS_LCODE by construction
}

static void nfb_build_dw_jobs(NfbDwJob* jobs) {
    using namespace nfl;
    // dz section -> the chain's layer index (slot of max |dz|)
    NfbDwBuilder b{jobs, [](int z) {
                       const int order[10] = {Z_D2, Z_D1, Z_D0, Z_FEAT, Z_L5, Z_L4, Z_L3, Z_L2, Z_L1, Z_L0};
                       for (int i = 0; i < 10; ++i)
                           if (order[i] == z) return i;
                       return -1;
                   }};
    b.layer256(Z_L1, S_H0, G_L1, CS_L0 + 256);
    b.layer256(Z_L2, S_H1, G_L2, CS_L0 + 512);
    b.layer256(Z_L3, S_H2, G_L3B, -1);
    b.layer256(Z_L4, S_H3, G_L4, CS_L0 + 1024);
    b.layer256(Z_L5, S_H4, G_L5, CS_L0 + 1280);
    b.layer256(Z_FEAT, S_H5, G_FEAT, CS_L0 + 1536);
    for (int q = 0; q < 2; ++q) {   // the two products against the positional encoding: dZ_L0 x PE, dZ_L3 x PE
        NfbDwJob& j = b.new_job();
        const int sz = b.add_seg(j, 0, q ? Z_L3 : Z_L0, 256, q ? CS_L0 + 768 : CS_L0), sp = b.add_seg(j, 2, S_PE, 64, -1);
        for (int w = 0; w < 8; ++w) j.prod[w] = b.prod(sz, 32 * w, 32, sp, 0, 64, (q ? G_L3A : G_L0) + 32 * w * 64, 64);
    }
    {   // (dZ_D0 | d_raw) x feat
        NfbDwJob& j = b.new_job();
        const int sz = b.add_seg(j, 0, Z_D0, 128, CS_D0), sf = b.add_seg(j, 2, S_FEAT, 256, -1), sr = b.add_seg(j, 1, 0, 4, -1);
        for (int w = 0; w < 8; ++w) {
            const int ag = w >> 2, bg = w & 3;
            j.prod[w] = b.prod(sz, 64 * ag, 64, sf, 64 * bg, 64, G_D0A + 64 * ag * 256 + 64 * bg, 256);
        }
        for (int bg = 0; bg < 4; ++bg)     // row 3 (d sigma) = fc_alpha.weight gradient
            j.prod[8 + bg] = b.prod(sr, 0, 4, sf, 64 * bg, 64, G_ALPHA + 64 * bg, 256);
    }
    {   // dZ_D1 x d0 and dZ_D2 x d1
        NfbDwJob& j = b.new_job();
        const int z1 = b.add_seg(j, 0, Z_D1, 128, CS_D0 + 128), a0 = b.add_seg(j, 2, S_D0, 128, -1);
        const int z2 = b.add_seg(j, 0, Z_D2, 128, CS_D0 + 256), a1 = b.add_seg(j, 2, S_D1, 128, -1);
        for (int w = 0; w < 4; ++w) {
            const int ag = w >> 1, bg = w & 1;
            j.prod[w] = b.prod(z1, 64 * ag, 64, a0, 64 * bg, 64, G_D1 + 64 * ag * 128 + 64 * bg, 128);
            j.prod[4 + w] = b.prod(z2, 64 * ag, 64, a1, 64 * bg, 64, G_D2 + 64 * ag * 128 + 64 * bg, 128);
        }
    }
    {   // dZ_D0 x dir features, d_raw x d2 (fc_rgb.weight; the 4 output-bias gradients are the column sums of d_raw)
        NfbDwJob& j = b.new_job();
        const int sz = b.add_seg(j, 0, Z_D0, 128, -1), sd = b.add_seg(j, 2, S_DIRF, 16, -1), sr = b.add_seg(j, 1, 0, 4, CS_RGB);
        const int s2 = b.add_seg(j, 2, S_D2, 128, -1);
        for (int ag = 0; ag < 2; ++ag) j.prod[ag] = b.prod(sz, 64 * ag, 64, sd, 0, 16, G_D0B + 64 * ag * 16, 16);
        for (int bg = 0; bg < 2; ++bg) j.prod[2 + bg] = b.prod(sr, 0, 4, s2, 64 * bg, 64, G_RGB + 64 * bg, 128);
    }
    // b.nj == NFB_DW_JOBS by construction
}

__device__ __forceinline__ void nfb_dw_split(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const nfb_elt hh = (nfb_elt)x[j];
        hi[j] = hh;
        lo[j] = (nfb_elt)(x[j] - (float)hh);
    }
}

#if NFB_F16
// 2^(13 - k) for a section whose largest |gradient| (float bits) lies in [2^k, 2^(k+1)); exponents clamp to [-60, 60]
__device__ __forceinline__ float nfb_dw_pow2_scale(unsigned max_bits, float& inv) {
    int se = 267 - (int)(max_bits >> 23);
    se = se < 67 ? 67 : (se > 187 ? 187 : se);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
    return __uint_as_float((unsigned)se << 23);
}
// (hi, lo) fp16 split of x * s (s a power of two) on the mixed-precision FMA instructions: v_fma_mixlo/hi_f16 multiply and
// round to fp16 in one step, v_fma_mix_f32 forms the exact residual x * s - hi straight from the packed halves, one packed
// convert rounds it -- 2.5 instructions per element instead of 5 (multiply, convert, convert back, subtract, convert); the
// result is bit-identical (tools/micro probe), and this kernel sits on the edge of being conversion-bound.
__device__ __forceinline__ void nfb_dw_split_scaled(const float (&x)[8], float s, bf16x8& hi, bf16x8& lo) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned h;
        float r0, r1;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x[2 * q]), "v"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x[2 * q + 1]), "v"(s));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x[2 * q]), "v"(s), "v"(h));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x[2 * q + 1]), "v"(s), "v"(h));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lw[q]) : "v"(r0), "v"(r1));
        hw[q] = h;
    }
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(bf16x8, (u32x4_t){hw[0], hw[1], hw[2], hw[3]});
    lo = __builtin_bit_cast(bf16x8, (u32x4_t){lw[0], lw[1], lw[2], lw[3]});
}
#endif

// L step for one f32-row tile: lane (h, c) <- feature c of points p0 + nfb_dw_point_of_slot(h, 0..7) (zero beyond the slice / the section width)
// L step: raw buffer loads.  voff = this lane's byte offset of (point 4 h of the stage, its feature) within the slice's rows
// of the section, or 0x80000000 for lanes past the section width; the 8 points are reached through the scalar offset
// j * stride.  The hardware range check sees only voff (never the scalar offset), so the descriptor of the pipelined loop
// covers exactly the WHOLE stages of the slice: they need no test, and stages past them (the pipeline overruns by a few)
// read 0 through the check.  The partial stage the last slice can end with is handled after the loop (nfb_dw_load_tail).
__device__ __forceinline__ void nfb_dw_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned stride_b, float (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)(nfb_dw_point_of_slot(0, j) * stride_b), 0));
}
__device__ __forceinline__ void nfb_dw_load_tail(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned stride_b, int n_ok, int h,
                                                 float (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(nfb_dw_point_of_slot(h, j) < n_ok ? voff : 0x80000000u),
                                                                            (int)(nfb_dw_point_of_slot(0, j) * stride_b), 0));
}

// MODEL selects the job table: 0 paper model, 1 second model family
template <int MODEL>
__global__ void __launch_bounds__(64 * NFB_DW_WAVES, 1)
NFB_DW_NAME(k_paper_dw_gemm)(const float* __restrict__ dz, const float* __restrict__ d_raw, const float* __restrict__ saved,
                             int64_t n_points, int64_t pts_per_slice, float* __restrict__ slabs, int slab_floats,
                             const float* __restrict__ gscale) {
    __shared__ __attribute__((aligned(16))) uint4 lds_cvt[NFB_DW_RING * NFB_DW_CVT_U4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const NfbDwJob& job = MODEL ? c_dwb_jobs_lcode[blockIdx.x] : c_dwb_jobs[blockIdx.x];
    const NfbDwProd pr = job.prod[wave];
    const int slice = blockIdx.y;
    const int64_t p_begin = (int64_t)slice * pts_per_slice;
    int64_t p_end = p_begin + pts_per_slice;
    if (p_end > n_points) p_end = n_points;
    const int n_stages = p_end > p_begin ? (int)((p_end - p_begin) / NFB_DW_PTS) : 0;                // whole stages
    const int n_tail = p_end > p_begin ? (int)(p_end - p_begin) - n_stages * NFB_DW_PTS : 0;         // points of the partial one

    // the tile this wave loads + converts every stage
    const bool t_on = wave < job.ntile;
    const NfbDwTile tl = job.tile[t_on ? wave : 0];
    const NfbDwSeg tsg = job.seg[tl.seg];
    const bool t_fok = tl.f0 + c < tsg.width;
    const unsigned t_stride_b = 4u * (unsigned)tsg.width;
    // sections of the split training forward's `saved` are n_pad = n_points rounded up to 32 points long
    const int64_t n_pad = (n_points + 31) & ~(int64_t)31;
    const bool t_stream = __builtin_amdgcn_readfirstlane((int)(t_on && tsg.kind == 3)) != 0;
    const float* t_sec = tsg.kind == 1 ? d_raw : (tsg.kind == 0 ? dz + (int64_t)tsg.sec * n_points : saved + (int64_t)tsg.sec * n_pad);
    const float* t_g = t_sec + p_begin * tsg.width;
    const __amdgpu_buffer_rsrc_t t_rsrc = t_stream
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t_sec), (short)0, (int)((unsigned)n_pad * t_stride_b), 0x00020000)   // the whole stream
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t_g), (short)0, (int)(n_stages * NFB_DW_PTS * t_stride_b), 0x00020000);
    const unsigned t_voff = (t_on && t_fok) ? 4u * (unsigned)(tl.f0 + c) + (unsigned)(4 * h) * t_stride_b : 0x80000000u;
    // streamed tile: block pair of stage i = ((p_begin / 16 + i) * tiles of the section + tile) * 2 KiB, hi then lo, lane-linear
    const unsigned s_blk0 = (unsigned)(p_begin >> 4) * (unsigned)(tsg.width >> 5) + (unsigned)(tl.f0 >> 5);
    const unsigned s_step = (unsigned)(tsg.width >> 5);
    // Odd slices accumulate MINUS their gradient (the gradient operand is negated on the way into the MFMA, the slab entry on the
    // way out): the 16-bit-input MFMA accumulation is biased toward -inf by a fraction of an ulp per step, 700+ steps deep here,
    // and alternating the sign over slices lets that bias cancel in the slab reduction instead of adding up.
    const float slice_sgn = (slice & 1) ? -1.0f : 1.0f;
#if !NFB_F16
    const unsigned t_flip = (tsg.kind != 2 && (slice & 1)) ? 0x80000000u : 0u;     // sign bit of the raw floats (after their column sum)
#endif
#if NFB_F16
    // activations x 2^4, gradient sections x the power of two that lifts their largest entry into [2^13, 2^14)
    const unsigned* gbits = reinterpret_cast<const unsigned*>(gscale);
    float unused_inv;
    const float t_scale = __uint_as_float(__builtin_amdgcn_readfirstlane(                      // wave-uniform: keep it in an SGPR
        __float_as_uint(tsg.kind == 2 ? 16.0f : slice_sgn * nfb_dw_pow2_scale(gbits[tsg.gs & 15], unused_inv))));
    const int out_gs = __builtin_amdgcn_readfirstlane(job.seg[job.tile[pr.a_tile].seg].gs & 15);   // this wave's rows are gradient tiles
#endif
    float t_cs = 0.f;
    float xs[NFB_DW_NSET][8];
    auto load = [&](int i, float (&x)[8]) { nfb_dw_load(t_rsrc, t_voff + (unsigned)i * NFB_DW_PTS * t_stride_b, t_stride_b, x); };
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // One stage: M(i-1) out of fragment buffer (i-1) % 4; an f32-row tile: C(i) into buffer i % 4, L(i+2) into registers; a streamed
    // tile: LDS-DMA of stage i+2's block pair into buffer (i+2) % 4 (free: last read by M(i-2), a barrier ago).  Always the full 2 x 2
    // block of MFMAs (narrower products simply do not store the surplus tiles).  The three MFMA groups (hi.hi, hi.lo, lo.hi; four
    // independent accumulators each) are interleaved with the vector work of C and the loads of L so that one hides under the
    // other; sched_barrier keeps the compiler from re-clustering them.  The barrier is raw: __syncthreads() would drain the DMA.
    auto dma = [&](int i) {                                 // this wave's block pair of stage i -> its tile's place in buffer i % 4
        uint4* dst = lds_cvt + (i & (NFB_DW_RING - 1)) * NFB_DW_CVT_U4 + wave * 128;
        const unsigned so = (s_blk0 + (unsigned)i * s_step) * 2048u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)(lane * 16), (int)so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (__attribute__((address_space(3))) void*)(dst + 64), 16, (int)(lane * 16), (int)(so + 1024u), 0, 0);
    };
    auto dma_into0 = [&](int i) {                           // the partial last stage: into buffer 0, outside the ring discipline
        uint4* dst = lds_cvt + wave * 128;
        const unsigned so = (s_blk0 + (unsigned)i * s_step) * 2048u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)(lane * 16), (int)so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (__attribute__((address_space(3))) void*)(dst + 64), 16, (int)(lane * 16), (int)(so + 1024u), 0, 0);
    };
    auto stage = [&](auto stream_tag, int i, const float (&xc)[8], float (&xn)[8]) {
        constexpr bool STREAM = decltype(stream_tag)::value;
        const uint4* rd = lds_cvt + ((i + NFB_DW_RING - 1) & (NFB_DW_RING - 1)) * NFB_DW_CVT_U4 + lane;
        uint4* wr = lds_cvt + (i & (NFB_DW_RING - 1)) * NFB_DW_CVT_U4 + wave * 128 + lane;
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = __builtin_bit_cast(bf16x8, rd[(pr.a_tile + t) * 128]);
            bh[t] = __builtin_bit_cast(bf16x8, rd[(pr.b_tile + t) * 128]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!STREAM) {
            bf16x8 hi, lo;
#if NFB_F16
            t_cs += ((xc[0] + xc[1]) + (xc[2] + xc[3])) + ((xc[4] + xc[5]) + (xc[6] + xc[7]));
            nfb_dw_split_scaled(xc, t_scale, hi, lo);
#else
            t_cs += ((xc[0] + xc[1]) + (xc[2] + xc[3])) + ((xc[4] + xc[5]) + (xc[6] + xc[7]));
            float xf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = __uint_as_float(__float_as_uint(xc[j]) ^ t_flip);
            nfb_dw_split(xf, hi, lo);
#endif
            wr[0] = __builtin_bit_cast(uint4, hi);
            wr[64] = __builtin_bit_cast(uint4, lo);
        } else {
            dma(i + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) bl[t] = __builtin_bit_cast(bf16x8, rd[(pr.b_tile + t) * 128 + 64]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = NFB_MFMA(ah[t], bh[u], acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = NFB_MFMA(ah[t], bl[u], acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) al[t] = __builtin_bit_cast(bf16x8, rd[(pr.a_tile + t) * 128 + 64]);
        if constexpr (!STREAM) load(i + NFB_DW_NSET - 1, xn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = NFB_MFMA(al[t], bh[u], acc[t][u], 0, 0, 0);
        // this wave's fragment writes of stage i are done (f32-row tile) / its block pair of stage i has landed, the two later ones may fly
        if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // the first M step (i = 0) reads fragment buffer 3 before anything was written into it
    for (int e = threadIdx.x; e < NFB_DW_CVT_U4; e += 64 * NFB_DW_WAVES) lds_cvt[(NFB_DW_RING - 1) * NFB_DW_CVT_U4 + e] = make_uint4(0u, 0u, 0u, 0u);
    if (t_stream) {
        dma(0);
        dma(1);
        __syncthreads();                                    // (drains the two DMAs: once, before the pipelined loop)
        for (int i0 = 0; i0 <= n_stages; i0 += NFB_DW_NSET) {
#pragma unroll
            for (int q = 0; q < NFB_DW_NSET; ++q) stage(std::true_type{}, i0 + q, xs[q], xs[(q + NFB_DW_NSET - 1) % NFB_DW_NSET]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < NFB_DW_NSET - 1; ++q) load(q, xs[q]);
        __syncthreads();
        for (int i0 = 0; i0 <= n_stages; i0 += NFB_DW_NSET) {
#pragma unroll
            for (int q = 0; q < NFB_DW_NSET; ++q) stage(std::false_type{}, i0 + q, xs[q], xs[(q + NFB_DW_NSET - 1) % NFB_DW_NSET]);
        }
    }
    if (n_tail > 0) {   // the partial stage (last slice only), not pipelined
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (overrun DMAs of the loop target the ring: let them land before buffer 0 is reused)
        __syncthreads();
        if (t_stream) {
            dma_into0(n_stages);
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(t_g), (short)0, (int)((p_end - p_begin) * (int64_t)t_stride_b), 0x00020000);
            float xt[8];
            nfb_dw_load_tail(rs, t_voff + (unsigned)n_stages * NFB_DW_PTS * t_stride_b, t_stride_b, n_tail, h, xt);
            bf16x8 hi, lo;
#if NFB_F16
            t_cs += ((xt[0] + xt[1]) + (xt[2] + xt[3])) + ((xt[4] + xt[5]) + (xt[6] + xt[7]));
            nfb_dw_split_scaled(xt, t_scale, hi, lo);
#else
            t_cs += ((xt[0] + xt[1]) + (xt[2] + xt[3])) + ((xt[4] + xt[5]) + (xt[6] + xt[7]));
#pragma unroll
            for (int j = 0; j < 8; ++j) xt[j] = __uint_as_float(__float_as_uint(xt[j]) ^ t_flip);
            nfb_dw_split(xt, hi, lo);
#endif
            lds_cvt[wave * 128 + lane] = __builtin_bit_cast(uint4, hi);
            lds_cvt[wave * 128 + lane + 64] = __builtin_bit_cast(uint4, lo);
        }
        __syncthreads();                                    // (its fence also waits for the tail DMA)
        const uint4* rd = lds_cvt + lane;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, rd[(pr.a_tile + t) * 128]), al = __builtin_bit_cast(bf16x8, rd[(pr.a_tile + t) * 128 + 64]);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, rd[(pr.b_tile + u) * 128]), bl = __builtin_bit_cast(bf16x8, rd[(pr.b_tile + u) * 128 + 64]);
                acc[t][u] = NFB_MFMA(ah, bh, acc[t][u], 0, 0, 0);
                acc[t][u] = NFB_MFMA(ah, bl, acc[t][u], 0, 0, 0);
                acc[t][u] = NFB_MFMA(al, bh, acc[t][u], 0, 0, 0);
            }
    }

    // D of tile (t, u): lane (h, c), reg r -> row 32 t + (r & 3) + 8 (r >> 2) + 4 h, column 32 u + c
    float* slab = slabs + (int64_t)slice * slab_floats;
#if NFB_F16
    float out_scale;
    (void)nfb_dw_pow2_scale(gbits[out_gs], out_scale);
    out_scale *= slice_sgn * (1.0f / 16.0f);
#else
    const float out_scale = slice_sgn;
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= pr.a_valid) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (32 * u + c < pr.b_valid) slab[pr.out_off + row * pr.ldo + 32 * u + c] = acc[t][u][r] * out_scale;
        }
    }
    if (t_on && tl.cs_off >= 0) {
        const float v = t_cs + __shfl_xor(t_cs, 32, 64);
        // (lane id re-derived from mbcnt: the threadIdx-derived copy would have to live -- in the fp16 instantiation: in scratch -- across the point loop)
        if (__lane_id() < 32 && t_fok) slab[tl.cs_off + c] = v;
    }
}

static std::mutex g_dwb_mutex;
static bool g_dwb_ready[64] = {false};

// called by nf_paper_mlp_bwd_bf16 / _f16 (nf_mlp_bwd.hip) and the lcode counterparts; slabs: n_slices x slab floats of the model;
// gscale: max |gradient| per section (float bits) as left by the backward chain (fp16 instantiation only)
int NFB_DW_NAME(nfb_launch_dw_gemm)(int model, const float* dz, const float* d_raw, const float* saved, int64_t n_points,
                                    int64_t pts_per_slice, int n_slices, float* slabs, const float* gscale, nf_stream_t stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return NF_EINVAL;
    {
        std::lock_guard<std::mutex> lock(g_dwb_mutex);
        if (!g_dwb_ready[dev]) {                      // job tables -> constant memory of this device (marked ready only on success)
            static NfbDwJob jobs[NFB_DW_JOBS], jobs_l[NFB_DW_JOBS_LCODE];
            nfb_build_dw_jobs(jobs);
            nfb_build_dw_jobs_lcode(jobs_l);
            e = hipMemcpyToSymbol(HIP_SYMBOL(c_dwb_jobs), jobs, sizeof(jobs));
            if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_dwb_jobs_lcode), jobs_l, sizeof(jobs_l));
            if (e != hipSuccess) return (int)e;
            g_dwb_ready[dev] = true;
        }
    }
    if (model == 0)
        hipLaunchKernelGGL((NFB_DW_NAME(k_paper_dw_gemm)<0>), dim3(NFB_DW_JOBS, (unsigned)n_slices), dim3(64 * NFB_DW_WAVES), 0, nf_s(stream),
                           dz, d_raw, saved, n_points, pts_per_slice, slabs, (int)nfl::SLAB_FLOATS, gscale);
    else
        hipLaunchKernelGGL((NFB_DW_NAME(k_paper_dw_gemm)<1>), dim3(NFB_DW_JOBS_LCODE, (unsigned)n_slices), dim3(64 * NFB_DW_WAVES), 0,
                           nf_s(stream), dz, d_raw, saved, n_points, pts_per_slice, slabs, (int)nlc::SLAB_FLOATS, gscale);
    NF_RETURN_LAUNCH();
}

#if !NFB_F16

// =================================================================================================
// split training layout -> the exact-f32 training layout (tests, and the exact-f32 dW GEMMs run on a split forward's activations:
// nf_*_mlp_bwd_bf16(..., exact_dw = 1)).  Stream sections: x = hi + lo (fp16 instantiation: / 2^4); f32-row sections and the bit
// masks are copied from their n_pad-based places to the n-based places of nf_mlp_layout.h / nf_mlp_lcode_layout.h.
// =================================================================================================
__global__ void __launch_bounds__(256) k_unsplit_section(const unsigned short* __restrict__ stream, int64_t n_points, int width, int is_f16,
                                                         float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_points * width) return;
    const int64_t p = e / width;
    const int f = (int)(e - p * width), r = (int)(p & 15), h = (r >> 2) & 1, j = (r & 3) + 4 * (r >> 3), nt = f >> 5, c = f & 31;
    const int64_t blk = ((p >> 4) * (width >> 5) + nt) * 2;                      // hi block; lo = + 1
    const int idx = (h * 32 + c) * 8 + j;
    const unsigned short bh = stream[blk * 512 + idx], bl = stream[(blk + 1) * 512 + idx];
    float v;
    if (is_f16) v = ((float)__builtin_bit_cast(_Float16, bh) + (float)__builtin_bit_cast(_Float16, bl)) * (1.0f / 16.0f);
    else v = __uint_as_float((unsigned)bh << 16) + __uint_as_float((unsigned)bl << 16);
    out[e] = v;
}

// model 0: paper, 1: second family.  out: nf_paper_saved_floats / nf_lcode_saved_floats(n_points) floats in the exact-f32 layout.
extern "C" int nf_split_saved_to_f32(int model, const float* saved_split, int64_t n_points, int is_f16, float* out, nf_stream_t stream) {
    if (!saved_split || !out || n_points <= 0 || (model != 0 && model != 1)) return NF_EINVAL;
    const int64_t n = n_points, n_pad = (n + 31) & ~(int64_t)31;
    struct Sec { int off, width, stream; };
    static const Sec paper[] = {{nfl::S_PE, 64, 0}, {nfl::S_H0, 256, 1}, {nfl::S_H1, 256, 1}, {nfl::S_H2, 256, 1}, {nfl::S_H3, 256, 1},
                                {nfl::S_H4, 256, 1}, {nfl::S_H5, 256, 1}, {nfl::S_FEAT, 256, 1}, {nfl::S_D0, 128, 1}, {nfl::S_D1, 128, 1},
                                {nfl::S_D2, 128, 1}, {nfl::S_DIRF, 16, 0}, {nfl::S_MASK, 9 * 8, 0}};
    static const Sec lcode[] = {{nlc::S_PE, 64, 0}, {nlc::S_L1, 256, 1}, {nlc::S_X0, 256, 1}, {nlc::S_X1, 256, 1}, {nlc::S_X2, 256, 1},
                                {nlc::S_FEAT, 256, 1}, {nlc::S_DIR, 128, 1}, {nlc::S_DIRF, 16, 0}, {nlc::S_MASK, 5 * 8, 0}};
    const Sec* secs = model ? lcode : paper;
    const int n_secs = model ? (int)(sizeof(lcode) / sizeof(Sec)) : (int)(sizeof(paper) / sizeof(Sec));
    for (int k = 0; k < n_secs; ++k) {
        const Sec& sc = secs[k];
        const float* src = saved_split + (int64_t)sc.off * n_pad;
        float* dst = out + (int64_t)sc.off * n;
        if (sc.stream) {
            const int64_t total = n * sc.width, grid = (total + 255) / 256;
            if (grid > 0x7fffffff) return NF_EINVAL;
            hipLaunchKernelGGL(k_unsplit_section, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), reinterpret_cast<const unsigned short*>(src), n,
                               sc.width, is_f16, dst);
        } else {
            const hipError_t e = hipMemcpyAsync(dst, src, (size_t)n * sc.width * sizeof(float), hipMemcpyDeviceToDevice, nf_s(stream));
            if (e != hipSuccess) return (int)e;
        }
    }
    NF_RETURN_LAUNCH();
}

// slices for the split-bf16 dW kernel: one workgroup per CU (paper: 11 bundles x 23 slices = 253 workgroups, second family:
// 8 x 32 = 256); two rounds (46 slices) measured 2 % slower end to end (twice the slab traffic), three rounds slower still
void nfb_dw_plan(int model, int64_t n_points, int64_t* pts_per_slice, int* n_slices) {
    int target = model == 0 ? 23 : 32;
    if (const char* ev = getenv("NERFACE_DW_SLICES")) { const int v = atoi(ev); if (v >= 1 && v <= 512) target = v; }   // tuning knob
    int64_t pps = (n_points + target - 1) / target;
    pps = (pps + 15) / 16 * 16;
    if (pps < 256) pps = 256;
    *pts_per_slice = pps;
    *n_slices = (int)((n_points + pps - 1) / pps);
}

// =================================================================================================
// host-only self-test of the job tables (no device needed; tests/test_host.py): every slab entry the unpack kernels read
// must be written by exactly one (bundle, wave) per slice, nothing may be written twice, and every wave's tile ids must
// stay inside its bundle.  Returns 0, or a negative code that says what failed.
// =================================================================================================
static int nfb_check_tables(const NfbDwJob* jobs, int n_jobs, int slab_floats, long expected_entries) {
    std::vector<unsigned char> hits((size_t)slab_floats, 0);
    long total = 0;
    for (int jb = 0; jb < n_jobs; ++jb) {
        const NfbDwJob& j = jobs[jb];
        if (j.nseg < 1 || j.nseg > 4 || j.ntile < 1 || j.ntile > NFB_DW_MAX_TILES) return -1;
        for (int t = 0; t < j.ntile; ++t) {
            const NfbDwTile& tl = j.tile[t];
            if (tl.seg < 0 || tl.seg >= j.nseg || tl.f0 < 0 || tl.f0 >= j.seg[tl.seg].width) return -2;
            if (tl.cs_off >= 0)
                for (int c = 0; c < 32 && tl.f0 + c < j.seg[tl.seg].width; ++c) {
                    if (tl.cs_off + c >= slab_floats) return -3;
                    if (hits[tl.cs_off + c]++) return -4;
                    ++total;
                }
        }
        for (int w = 0; w < NFB_DW_WAVES; ++w) {
            const NfbDwProd& p = j.prod[w];
            if (p.a_tile < 0 || p.b_tile < 0 || p.a_tile + 2 > NFB_DW_MAX_TILES || p.b_tile + 2 > NFB_DW_MAX_TILES) return -5;
            if (p.a_valid > 0 && (p.a_tile + p.na > j.ntile || p.b_tile + p.nb > j.ntile || p.na > 2 || p.nb > 2)) return -6;
            for (int r = 0; r < p.a_valid; ++r)
                for (int c = 0; c < p.b_valid; ++c) {
                    const long e = (long)p.out_off + (long)r * p.ldo + c;
                    if (e < 0 || e >= slab_floats) return -7;
                    if (hits[e]++) return -8;
                    ++total;
                }
        }
    }
    return total == expected_entries ? 0 : -9;
}

extern "C" int nf_selftest_dw_tables_bf16(void) {
    static NfbDwJob jobs[NFB_DW_JOBS], jobs_l[NFB_DW_JOBS_LCODE];
    nfb_build_dw_jobs(jobs);
    nfb_build_dw_jobs_lcode(jobs_l);
    // paper: dW of layers_xyz.0 / .3 (PE part) 2 x 256 x 64, six 256 x 256, layers_dir.0 128 x (256 + 16), two 128 x 128,
    // fc_rgb / fc_alpha as 4 rows of d_raw each, column sums 7 x 256 + 3 x 128 + 4
    const long paper = 2L * 256 * 64 + 6L * 65536 + 128L * 272 + 2L * 128 * 128 + 4L * 128 + 4L * 256 + 7 * 256 + 3 * 128 + 4;
    int rc = nfb_check_tables(jobs, NFB_DW_JOBS, nfl::SLAB_FLOATS, paper);
    if (rc) return rc;
    // second family: layer1 (PE part) 256 x 64, four 256 x 256, layers_dir.0 128 x (256 + 16), 4 x 128, 4 x 256, sums 5 x 256 + 128 + 4
    const long lcode = 256L * 64 + 4L * 65536 + 128L * 272 + 4L * 128 + 4L * 256 + 5 * 256 + 128 + 4;
    rc = nfb_check_tables(jobs_l, NFB_DW_JOBS_LCODE, nlc::SLAB_FLOATS, lcode);
    return rc ? rc - 100 : 0;
}
#endif
