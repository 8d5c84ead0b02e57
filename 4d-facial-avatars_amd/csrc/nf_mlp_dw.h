// B2 (exact f32) and the slab reduction of B3, shared by the model families (nf_mlp_bwd.hip, nf_mlp_lcode_bwd.hip).
#pragma once
#include <vector>
#include <mutex>
#include "nf_mlp_dev.h"

// =================================================================================================
// B2: weight-gradient GEMMs.   One WAVE = one job: a 128 x 128 tile of  dW = A^T B  over one point slice.
//   A = dZ (or d_raw) [points][lda], B = saved activations [points][ldb].
//   MFMA: D[n][k] += A[n][pt] * B[pt][k]; step r of a 16-point chunk takes from lane group g the point
//   chunk + 4 r + g; lane (g, i) therefore issues dword loads of 16 consecutive floats per row (64 B).
// =================================================================================================
struct NfDwJob {
    int a_kind;      // 0: dz section, 1: d_raw
    int a_sec;       // section offset (floats per point) within dz
    int lda, a_col0, n_valid;
    int b_sec, ldb, b_col0, k_valid;
    int out_off, ldo;
    int cs_off;      // >= 0: also write column sums of A (bias grads) for this n-block
};

// MODEL only separates the instantiations of the model families (each lives in its own translation unit).
template <int MODEL>
__global__ void __launch_bounds__(256, 1)
k_dw_gemm(const NfDwJob* __restrict__ jobs, int n_jobs, int slab_floats, const float* __restrict__ dz, const float* __restrict__ d_raw,
          const float* __restrict__ saved, int64_t n_points, int64_t pts_per_slice, float* __restrict__ slabs) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int jid = blockIdx.x * 4 + wave;
    const int slice = blockIdx.y;
    if (jid >= n_jobs) return;
    const NfDwJob job = jobs[jid];
    const int64_t p_begin = (int64_t)slice * pts_per_slice;
    int64_t p_end = p_begin + pts_per_slice;
    if (p_end > n_points) p_end = n_points;
    const float* A = (job.a_kind ? d_raw : dz + (int64_t)job.a_sec * n_points) + job.a_col0;
    const float* B = saved + (int64_t)job.b_sec * n_points + job.b_col0;
    float* out = slabs + (int64_t)slice * slab_floats + job.out_off;
    const int lda = job.lda, ldb = job.ldb;

    // Row/column order inside the 128 x 128 tile is free, so it is chosen for 16-byte operand loads: lane (g, i) reads
    // 4 consecutive features 64 sb + 4 i .. +3 of ONE point; component t of that float4 is the lane's operand for MFMA tile
    // (sb, t), whose 16 rows are therefore the features 64 sb + 4 i' + t.  One load feeds four tiles.
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { a_ok[q] = 64 * q + 4 * i < job.n_valid; b_ok[q] = 64 * q + 4 * i < job.k_valid; }

    f32x4 acc[8][8];                        // [4 sb + t][4 sk + t']
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) acc[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 cs[2];
    cs[0] = cs[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 a[4][2], b[4][2], an[4][2], bn[4][2];     // [step r][sub-block]
    auto load_chunk = [&](int64_t p, f32x4 (&aa)[4][2], f32x4 (&bb)[4][2]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = p + 4 * r + g;
            const bool rv = row < p_end;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                aa[r][q] = (rv && a_ok[q]) ? *reinterpret_cast<const f32x4*>(A + row * lda + 64 * q + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
                bb[r][q] = (rv && b_ok[q]) ? *reinterpret_cast<const f32x4*>(B + row * ldb + 64 * q + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    if (p_begin < p_end) load_chunk(p_begin, a, b);
    for (int64_t p = p_begin; p < p_end; p += 16) {
        const bool more = p + 16 < p_end;
        if (more) load_chunk(p + 16, an, bn);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cs[0] += a[r][0];
            cs[1] += a[r][1];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][nt >> 2][nt & 3], b[r][kt >> 2][kt & 3], acc[nt][kt], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q) { a[r][q] = an[r][q]; b[r][q] = bn[r][q]; }
        }
    }
    // D of tile (nt = 4 sb + t, kt = 4 sk + t'): lane (g, c = i), reg r' -> row n = 64 sb + 4 (4 g + r') + t,
    // column k = 64 sk + 4 c + t'.  For fixed (nt, r', sk) a lane holds 4 consecutive k: one 16-byte store.
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nrow = 64 * (nt >> 2) + 4 * (4 * g + r) + (nt & 3);
            if (nrow < job.n_valid) {
#pragma unroll
                for (int sk = 0; sk < 2; ++sk)
                    if (b_ok[sk])
                        *reinterpret_cast<f32x4*>(out + (int64_t)nrow * job.ldo + 64 * sk + 4 * i) =
                            (f32x4){acc[nt][4 * sk + 0][r], acc[nt][4 * sk + 1][r], acc[nt][4 * sk + 2][r], acc[nt][4 * sk + 3][r]};
            }
        }
    if (job.cs_off >= 0) {
        float* cso = slabs + (int64_t)slice * slab_floats + job.cs_off;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 v = cs[q];
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] += __shfl_xor(v[t], 16, 64); v[t] += __shfl_xor(v[t], 32, 64); }
            if (g == 0 && a_ok[q]) *reinterpret_cast<f32x4*>(cso + 64 * q + 4 * i) = v;
        }
    }
}


// =================================================================================================
// B2, shared-panel form (the paper model's exact-f32 training path).
// Same arithmetic, slices and per-element accumulation order as k_dw_gemm -- results are bit-identical -- but a WORKGROUP is
// a GROUP of four 128 x 128 products that read common operand panels (the 2 x 2 blocks of one 256 x 256 product; the four
// products against the positional encoding; ...):
//   * a panel = 16 points x 128 columns of dZ / d_raw / saved activations = 8 KiB; the group's panels of a 16-point chunk are
//     fetched ONCE per workgroup, by LDS-DMA (global_load_lds_dwordx4: two full rows = 1 KiB per wave instruction, no registers
//     in between), into a three-stage ring, three chunks ahead of the MFMAs.  k_dw_gemm fetched every panel once per JOB (twice
//     per 256 x 256 layer: 35 KB per point against 17.7 KB of distinct data, 3.2 TB/s) through 64 staging registers;
//   * eight waves, two per SIMD: wave 2 j + h owns the 128 x 64 half h of product j (128 accumulator registers, 48 operand
//     registers).  Two in-order waves on a SIMD cover each other's LDS reads, DMA issue and waits with MFMAs, and the
//     compiler has register room to interleave the loads the way the sched_group_barrier sequence asks for; a wave whose half
//     lies past a narrow panel's columns (PE: 64, dir slots: 16) issues no MFMAs and leaves the matrix pipe to its partner;
//   * one raw s_barrier per chunk; the wait before it is a counted vmcnt that leaves the newest chunks' DMA in flight.
// =================================================================================================
#define NF_DW_MAXP 6                                  // operand panels per group
#define NF_DW_STAGES 3
#define NF_DW_PANEL_BYTES (16 * 128 * 4)
#define NF_DW_STAGE_BYTES (NF_DW_MAXP * NF_DW_PANEL_BYTES)
#define NF_DW_WAVES 8
struct NfDwPanel {
    int kind;        // 0: dz section, 1: d_raw, 2: saved section; -1: unused slot
    int sec;         // section offset (floats per point)
    int ld;          // floats per point row
    int col0;        // first column of the panel
    int valid;       // valid columns (<= 128, multiple of 4)
};
struct NfDwWaveJob {     // one 128 x 128 product = two waves
    int a, b;        // panel slots of the A (gradient) and B (activation) operand
    int n_valid, k_valid;
    int out_off, ldo;
    int cs_off;      // >= 0: also write column sums of A (bias grads)
};
struct NfDwGroup {
    NfDwPanel panel[NF_DW_MAXP];
    NfDwWaveJob wave[4];
    int share;       // relative cost of the group in half-units: 2 = four full products, 1 = products whose second half is idle (PE)
    int n_slices;    // filled by nf_dw_plan_groups for the launch at hand
    int pts_per_slice;
};

#define NF_DW_MAX_GROUPS 9
struct NfDwGroupSet {                                         // passed by value: the slice plan changes with the launch size
    NfDwGroup g[NF_DW_MAX_GROUPS];
    int first_block[NF_DW_MAX_GROUPS + 1];                    // 1-D grid: blocks first_block[i] .. first_block[i + 1] - 1 are the slices of group i
};

// Slices per group: one workgroup per CU and all of them busy for the same time -- a group's slice count is proportional to its
// cost, and the grid holds exactly the (group, slice) pairs that exist, at most n_cu of them: workgroups are dealt to the eight XCDs
// round-robin, so a 2-D grid padded with empty blocks put 33 live workgroups on some XCDs' 32 CUs and doubled the kernel time.
// Returns the largest slice count (= number of slabs the reduction sums).
static inline int nf_dw_plan_groups(NfDwGroup* g, int n_groups, int64_t n_points, int* first_block = nullptr, int n_cu = 0) {
    if (n_cu <= 0) n_cu = (int)nf_cu_count();                       // the device the caller is about to launch on
    int shares = 0;
    for (int i = 0; i < n_groups; ++i) shares += g[i].share;
    int unit = (2 * n_cu) / shares;                                  // slices of a share-2 group
    if (unit < 1) unit = 1;
    int most = 1;
    for (int i = 0; i < n_groups; ++i) {
        int ns = unit * g[i].share / 2;
        if (ns < 1) ns = 1;
        int64_t pps = (n_points + ns - 1) / ns;
        pps = (pps + 15) / 16 * 16;
        if (pps < 1024) pps = 1024;
        g[i].pts_per_slice = (int)pps;
        g[i].n_slices = (int)((n_points + pps - 1) / pps);
        if (g[i].n_slices > most) most = g[i].n_slices;
    }
    if (first_block) {
        first_block[0] = 0;
        for (int i = 0; i < n_groups; ++i) first_block[i + 1] = first_block[i] + g[i].n_slices;
    }
    return most;
}

#define NF_REDUCE_ALT_MAX 8
struct NfReduceAlt {
    int lo4[NF_REDUCE_ALT_MAX], hi4[NF_REDUCE_ALT_MAX];
    int n_slices;        // 0: no such regions
};
// Slab regions of the groups that run fewer slices than `most` (their products fill only the first n_slices slabs).  Requires such a
// group's outputs to be whole rows (ldo == k_valid) -- true for the products against the positional encoding.  Returns false if the
// plan cannot be expressed (then the caller zero-fills the slabs instead).
static inline bool nf_dw_reduce_alt(const NfDwGroup* g, int n_groups, int most, NfReduceAlt* alt) {
    int n = 0;
    alt->n_slices = 0;
    for (int q = 0; q < NF_REDUCE_ALT_MAX; ++q) alt->lo4[q] = alt->hi4[q] = 0;
    for (int i = 0; i < n_groups; ++i) {
        if (g[i].n_slices == most) continue;
        if (alt->n_slices && alt->n_slices != g[i].n_slices) return false;
        alt->n_slices = g[i].n_slices;
        for (int w = 0; w < 4; ++w) {
            const NfDwWaveJob& j = g[i].wave[w];
            if (j.ldo != j.k_valid || (j.out_off & 3) || ((j.n_valid * j.ldo) & 3)) return false;
            if (n + 2 > NF_REDUCE_ALT_MAX) return false;
            alt->lo4[n] = j.out_off >> 2;
            alt->hi4[n++] = (j.out_off + j.n_valid * j.ldo) >> 2;
            if (j.cs_off >= 0) {
                if ((j.cs_off & 3) || (j.n_valid & 3)) return false;
                alt->lo4[n] = j.cs_off >> 2;
                alt->hi4[n++] = (j.cs_off + j.n_valid) >> 2;
            }
        }
    }
    return true;
}

__device__ __attribute__((aligned(16))) static const float nf_dw_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <int N> __device__ __forceinline__ void nf_dw_wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int MODEL>
__global__ void __launch_bounds__(64 * NF_DW_WAVES, 1)
k_dw_gemm_lds(NfDwGroupSet gs, int slab_floats, const float* __restrict__ dz, const float* __restrict__ d_raw,
              const float* __restrict__ saved, int64_t n_points, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) char lds[NF_DW_STAGES * NF_DW_STAGE_BYTES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    int gi = 0;
#pragma unroll
    for (int k = 1; k < NF_DW_MAX_GROUPS; ++k) gi += (int)blockIdx.x >= gs.first_block[k] ? 1 : 0;     // (first_block is non-decreasing)
    const NfDwGroup& grp = gs.g[gi];
    const int slice = (int)blockIdx.x - gs.first_block[gi];
    const int64_t pts_per_slice = grp.pts_per_slice;
    // waves w and w + 4 share a SIMD (waves are dealt round-robin over the four SIMDs): the two halves of one product, so a
    // product with an idle second half costs its SIMD half the MFMAs instead of leaving another SIMD empty
    const NfDwWaveJob job = grp.wave[wave & 3];
    const int half = wave >> 2;                       // B columns 64 half .. + 63 of the product
    const bool active = 64 * half < job.k_valid;      // wave-uniform: an idle wave only moves data
    const bool want_cs = job.cs_off >= 0 && half == 0;
    const int64_t p_begin = (int64_t)slice * pts_per_slice;
    int64_t p_end = p_begin + pts_per_slice;
    if (p_end > n_points) p_end = n_points;
    const int n_rows = (int)(p_end > p_begin ? p_end - p_begin : 0);
    const int n_full = n_rows >> 4;                   // whole 16-point chunks (block-uniform): the pipelined loop
    float* out = slabs + (int64_t)slice * slab_floats + job.out_off;

    // ---- DMA: piece (panel s, j) moves rows 2 j, 2 j + 1 of the chunk; wave w issues piece j = w of every panel.
    // Every lane keeps its 6 source pointers and advances them by one chunk per issue.  Lanes past a narrow panel's `valid`
    // columns re-read its last valid piece: those columns only feed output tiles nobody stores.
    const int dma_row = lane >> 5, dma_col = (lane & 31) * 4;
#ifndef NF_DW_BUFFER_DMA
#define NF_DW_BUFFER_DMA 1
#endif
#if NF_DW_BUFFER_DMA
    // Buffer form of the DMA: per panel one descriptor based at the slice's first row, the lane's part of the address in ONE VGPR
    // that never changes, and the chunk offset in an SGPR advanced by scalar adds -- no 64-bit vector address per panel, no
    // v_add_co / v_addc per issue.
    __amdgpu_buffer_rsrc_t rs[NF_DW_MAXP];
    int voff[NF_DW_MAXP], soff[NF_DW_MAXP], step_b[NF_DW_MAXP];
#pragma unroll
    for (int s = 0; s < NF_DW_MAXP; ++s) {
        const NfDwPanel pn = grp.panel[s];
        const bool on = pn.kind >= 0;
        const float* base = pn.kind == 1 ? d_raw : ((pn.kind == 0 ? dz : saved) + (int64_t)pn.sec * n_points);
        const int ld = on ? pn.ld : 0;
        const int col = on ? (dma_col < pn.valid ? dma_col : pn.valid - 4) : 0;
        step_b[s] = 64 * ld;
        soff[s] = 0;
        voff[s] = on ? ((2 * wave + dma_row) * ld + pn.col0 + col) * 4 : 0;
        rs[s] = __builtin_amdgcn_make_buffer_rsrc(on ? const_cast<float*>(base + p_begin * ld) : const_cast<float*>(nf_dw_zero16), (short)0, on ? -1 : 16,
                                                  0x00020000);
    }
    auto issue_piece = [&](int s, int stage, bool advance) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[s], (__attribute__((address_space(3))) void*)(lds + stage * NF_DW_STAGE_BYTES + wave * 1024 + s * NF_DW_PANEL_BYTES),
                                                 16, voff[s], soff[s], 0, 0);
        soff[s] += advance ? step_b[s] : 0;
    };
#else
    const char* src[NF_DW_MAXP];               // source of the NEXT chunk to issue
    int step_b[NF_DW_MAXP];                    // bytes per chunk (wave-uniform)
#pragma unroll
    for (int s = 0; s < NF_DW_MAXP; ++s) {
        const NfDwPanel pn = grp.panel[s];
        const bool on = pn.kind >= 0;
        const float* base = pn.kind == 1 ? d_raw : ((pn.kind == 0 ? dz : saved) + (int64_t)pn.sec * n_points);
        const int ld = on ? pn.ld : 0;
        const int col = on ? (dma_col < pn.valid ? dma_col : pn.valid - 4) : 0;
        step_b[s] = 64 * ld;
        src[s] = on ? reinterpret_cast<const char*>(base + (p_begin + 2 * wave + dma_row) * ld + pn.col0 + col)
                    : reinterpret_cast<const char*>(nf_dw_zero16);
    }
    // issue this wave's piece of panel s of the next chunk into `stage`; with advance = false the pointer stays (dummy issues
    // past the last chunk re-read it: harmless, and the counted waits stay uniform)
    auto issue_piece = [&](int s, int stage, bool advance) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[s],
                                         (__attribute__((address_space(3))) void*)(lds + stage * NF_DW_STAGE_BYTES + wave * 1024 + s * NF_DW_PANEL_BYTES),
                                         16, 0, 0);
        src[s] += advance ? step_b[s] : 0;
    };
#endif
    auto issue = [&](int stage, bool advance) {
#pragma unroll
        for (int s = 0; s < NF_DW_MAXP; ++s) issue_piece(s, stage, advance);
    };
    constexpr int NDMA = NF_DW_MAXP;           // per wave and chunk

    // ---- operands: lane (g, i) reads 4 consecutive features 4 i .. + 3 (+ 64 q) of point 4 r + g; component t of that float4 is
    // the lane's operand of MFMA tile (q, t), whose 16 rows are the features 64 q + 4 i' + t (as in k_dw_gemm).
    const int lane_off = g * 512 + i * 16;
    f32x4 a0[4][2], b0[4];
    auto read_step = [&](int stage, int r) {
        const char* sa = lds + stage * NF_DW_STAGE_BYTES + job.a * NF_DW_PANEL_BYTES + lane_off;
        const char* sb = lds + stage * NF_DW_STAGE_BYTES + job.b * NF_DW_PANEL_BYTES + half * 256 + lane_off;
        a0[r][0] = *reinterpret_cast<const f32x4*>(sa + r * 2048);
        a0[r][1] = *reinterpret_cast<const f32x4*>(sa + r * 2048 + 256);
        b0[r] = *reinterpret_cast<const f32x4*>(sb + r * 2048);
    };

    f32x4 acc[8][4];                        // [4 sb + t][t']: rows 64 sb + 4 (4 g + r') + t, columns 64 half + 4 c + t'
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 cs[2];
    cs[0] = cs[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma_step = [&](int r) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r][nt >> 2][nt & 3], b0[r][kt], acc[nt][kt], 0, 0, 0);
    };

    // ---- pipeline over the whole chunks, three LDS stages.  At the top of iteration c: the registers hold chunk c, chunk c + 1
    // has landed in LDS, chunk c + 2 is in flight, and the stage of chunk c is free (every wave's reads of it completed before the
    // barrier).  Under its 128 MFMAs the iteration issues the DMA of chunk c + 3 into that free stage and, as soon as the 32 MFMAs
    // of a 4-point step have issued, refills the step's 3 operand registers from chunk c + 1.
    if (n_full > 0) {
        issue(0, n_full > 1);
        issue(1, n_full > 2);
        issue(2, n_full > 3);
        nf_dw_wait_vm_lgkm0<2 * NDMA>();                             // my pieces of chunk 0 have landed
        __builtin_amdgcn_s_barrier();                                // ... and everybody's
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) read_step(0, r);                 // (idle halves too: keeps the prologue uniform)
        nf_dw_wait_vm_lgkm0<NDMA>();                                 // chunk 1 landed; chunk 0 is in my registers
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int st = 0;                                                  // stage of the chunk in the registers
#pragma unroll 1
        for (int c = 0; c < n_full; ++c) {
            const int s1 = st == NF_DW_STAGES - 1 ? 0 : st + 1;
            const bool adv = c + 4 < n_full;
            if (active) {
                __builtin_amdgcn_sched_barrier(0);
                // Four steps, fenced from one another (sched_barrier) so that the requested order is local and exact: 16 MFMAs,
                // a DMA piece of chunk c + 3 (-> the stage chunk c left), 16 MFMAs, a second piece, then the refill of the step's
                // 3 operand registers from chunk c + 1 -- issued right behind the MFMAs that read them.  The LAST step's refill
                // would sit right in front of the barrier (its LDS round trip in the path of all eight waves): it is read early, into
                // three staging registers behind step 0, and moved over after step 3's MFMAs.
#ifndef NF_DW_NO_STAGGER
                // The two waves of a SIMD (the halves of one product) run this stream side by side, and a DMA piece holds its wave's
                // issue for ~100 cycles: half 1 starts every chunk 8 MFMA slots late (its partner has the pipe to itself meanwhile).
#ifndef NF_DW_STAGGER_SLEEP
#define NF_DW_STAGGER_SLEEP 4                                        // x 64 cycles
#endif
                if (half) __builtin_amdgcn_s_sleep(NF_DW_STAGGER_SLEEP);
#endif
                if (want_cs) {   // column sums of this chunk's A operands (bias gradients), before the refills overwrite them
#pragma unroll
                    for (int r = 0; r < 4; ++r) { cs[0] += a0[r][0]; cs[1] += a0[r][1]; }
                }
                f32x4 a3s[2], b3s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mma_step(r);
                    if (r < 3) { issue_piece(2 * r, st, adv); issue_piece(2 * r + 1, st, adv); }
                    if (r < 3) read_step(s1, r);                     // (after the last chunk: reads a stage nobody uses)
                    if (r == 0) {
                        const char* sa = lds + s1 * NF_DW_STAGE_BYTES + job.a * NF_DW_PANEL_BYTES + lane_off + 3 * 2048;
                        const char* sb = lds + s1 * NF_DW_STAGE_BYTES + job.b * NF_DW_PANEL_BYTES + half * 256 + lane_off + 3 * 2048;
                        a3s[0] = *reinterpret_cast<const f32x4*>(sa);
                        a3s[1] = *reinterpret_cast<const f32x4*>(sa + 256);
                        b3s = *reinterpret_cast<const f32x4*>(sb);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                    if (r < 3) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                    if (r < 3) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    if (r == 0) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                    else if (r < 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                a0[3][0] = a3s[0];
                a0[3][1] = a3s[1];
                b0[3] = b3s;
            } else {
                issue(st, adv);                                      // an idle half only moves data
            }
            nf_dw_wait_vm_lgkm0<NDMA>();                             // chunk c + 2 landed (c + 3 may still fly); my refills are in
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            st = s1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no DMA may outlive the workgroup's LDS
    }
    bool a_ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) a_ok[q] = 64 * q + 4 * i < job.n_valid;
    const bool b_ok = 64 * half + 4 * i < job.k_valid;
    if (n_rows & 15) {   // the partial chunk the last slice can end with: straight from memory, rows past the end read 0
        const NfDwPanel pa = grp.panel[job.a], pb = grp.panel[job.b];
        const float* A = (pa.kind == 1 ? d_raw : dz + (int64_t)pa.sec * n_points) + pa.col0;
        const float* B = saved + (int64_t)pb.sec * n_points + pb.col0 + 64 * half;
        const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = p_begin + 16 * n_full + 4 * r + g;
            const bool rv = row < p_end;
#pragma unroll
            for (int q = 0; q < 2; ++q) a0[r][q] = (rv && a_ok[q]) ? *reinterpret_cast<const f32x4*>(A + row * pa.ld + 64 * q + 4 * i) : zero4;
            b0[r] = (rv && b_ok) ? *reinterpret_cast<const f32x4*>(B + row * pb.ld + 4 * i) : zero4;
        }
        if (want_cs) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { cs[0] += a0[r][0]; cs[1] += a0[r][1]; }
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mma_step(r);
        }
    }

    // D of tile (nt = 4 sb + t, t'): lane (g, c = i), reg r' -> row n = 64 sb + 4 (4 g + r') + t, column k = 64 half + 4 c + t'.
    // For fixed (nt, r') a lane holds 4 consecutive k: one 16-byte store.
    if (active) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nrow = 64 * (nt >> 2) + 4 * (4 * g + r) + (nt & 3);
                if (nrow < job.n_valid && b_ok)
                    *reinterpret_cast<f32x4*>(out + (int64_t)nrow * job.ldo + 64 * half + 4 * i) =
                        (f32x4){acc[nt][0][r], acc[nt][1][r], acc[nt][2][r], acc[nt][3][r]};
            }
    }
    if (want_cs) {
        float* cso = slabs + (int64_t)slice * slab_floats + job.cs_off;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 v = cs[q];
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] += __shfl_xor(v[t], 16, 64); v[t] += __shfl_xor(v[t], 32, 64); }
            if (g == 0 && a_ok[q]) *reinterpret_cast<f32x4*>(cso + 64 * q + 4 * i) = v;
        }
    }
}

// B3, first half: sum the per-slice slabs in a fixed order (deterministic).  16 bytes per thread, four independent partial
// sums (slices k = 0, 1, 2, 3 mod 4) so that the loads of consecutive slices overlap; slab_floats is a multiple of 4.
// `alt`: slab regions (float4 units) that only the first alt.n_slices slabs hold -- the products of a group that runs fewer,
// longer slices (nf_dw_plan_groups) -- so that the slabs need no zero-fill (71 MB per backward call) before the GEMM kernel.
template <int MODEL>
__global__ void __launch_bounds__(256) k_grad_reduce(const float* __restrict__ slabs, int n_slices, int slab_floats, float* __restrict__ sum,
                                                     NfReduceAlt alt) {
    const int n4 = slab_floats >> 2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += gridDim.x * blockDim.x) {
        const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + e;
        int ns = n_slices;
        if (alt.n_slices > 0) {
#pragma unroll
            for (int q = 0; q < NF_REDUCE_ALT_MAX; ++q)
                if (e >= alt.lo4[q] && e < alt.hi4[q]) ns = alt.n_slices;
        }
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 4 <= ns; k += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += src[(int64_t)(k + q) * n4];
        }
        // tail (k is a multiple of 4 here): slabs k, k + 1, k + 2 go to a[0], a[1], a[2] -- the order a[k & 3] gave, without indexing the
        // accumulator array by a run-time value (the compiler turned that into exec-mask sequences: 29 spilled SGPRs)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (k + q < ns) a[q] += src[(int64_t)(k + q) * n4];
        reinterpret_cast<f32x4*>(sum)[e] = (a[0] + a[1]) + (a[2] + a[3]);
    }
}

// point slices of the exact-f32 dW kernel
static inline void nf_bwd_plan(int64_t n_points, int64_t* pts_per_slice, int* n_slices) {
    int64_t pps = (n_points + 27) / 28;
    pps = (pps + 15) / 16 * 16;
    if (pps < 1024) pps = 1024;
    *pts_per_slice = pps;
    *n_slices = (int)((n_points + pps - 1) / pps);
}

// per-device copy of a job table
struct NfDwJobTable {
    std::mutex mutex;
    NfDwJob* dev[64] = {nullptr};
    template <class Build>
    int get(int n_jobs, Build build, const NfDwJob** out) {
        int d = 0;
        hipError_t e = hipGetDevice(&d);
        if (e != hipSuccess) return (int)e;
        if (d < 0 || d >= 64) return NF_EINVAL;
        std::lock_guard<std::mutex> lock(mutex);
        if (!dev[d]) {
            std::vector<NfDwJob> host(n_jobs);
            build(host.data());
            NfDwJob* p = nullptr;
            e = hipMalloc(&p, host.size() * sizeof(NfDwJob));
            if (e != hipSuccess) return (int)e;
            e = hipMemcpy(p, host.data(), host.size() * sizeof(NfDwJob), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
            dev[d] = p;
        }
        *out = dev[d];
        return 0;
    }
};

// host-only self-test of a shared-panel group table: as nf_check_dw_jobs, plus every wave's tile shape must equal its panels'
static inline int nf_check_dw_groups(const NfDwGroup* groups, int n_groups, int slab_floats, long expected_entries) {
    std::vector<unsigned char> hits((size_t)slab_floats, 0);
    long total = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
        const NfDwGroup& gr = groups[gi];
        if (gr.share < 1 || gr.share > 2) return -14;
        for (int s = 0; s < NF_DW_MAXP; ++s) {
            const NfDwPanel& pn = gr.panel[s];
            if (pn.kind < 0) continue;
            if (pn.kind > 2 || pn.valid < 4 || pn.valid > 128 || (pn.valid & 3) || (pn.col0 & 3) || (pn.ld & 3) || pn.col0 + pn.valid > pn.ld) return -10;
        }
        for (int w = 0; w < 4; ++w) {
            const NfDwWaveJob& j = gr.wave[w];
            if (j.a < 0 || j.a >= NF_DW_MAXP || j.b < 0 || j.b >= NF_DW_MAXP) return -11;
            const NfDwPanel &pa = gr.panel[j.a], &pb = gr.panel[j.b];
            if (pa.kind < 0 || pa.kind > 1 || pb.kind != 2) return -12;
            if (j.n_valid != pa.valid || j.k_valid != pb.valid) return -13;
            for (int r = 0; r < j.n_valid; ++r)
                for (int c = 0; c < j.k_valid; ++c) {
                    const long e = (long)j.out_off + (long)r * j.ldo + c;
                    if (e < 0 || e >= slab_floats) return -2;
                    if (hits[e]++) return -3;
                    ++total;
                }
            if (j.cs_off >= 0)
                for (int r = 0; r < j.n_valid; ++r) {
                    if (j.cs_off + r >= slab_floats) return -4;
                    if (hits[j.cs_off + r]++) return -5;
                    ++total;
                }
        }
    }
    return total == expected_entries ? 0 : -6;
}

// host-only self-test of an exact-f32 job table: no slab entry written twice, `expected_entries` entries written in total
static inline int nf_check_dw_jobs(const NfDwJob* jobs, int n_jobs, int slab_floats, long expected_entries) {
    std::vector<unsigned char> hits((size_t)slab_floats, 0);
    long total = 0;
    for (int jb = 0; jb < n_jobs; ++jb) {
        const NfDwJob& j = jobs[jb];
        if (j.n_valid < 1 || j.n_valid > 128 || j.k_valid < 1 || j.k_valid > 128) return -1;
        for (int r = 0; r < j.n_valid; ++r)
            for (int c = 0; c < j.k_valid; ++c) {
                const long e = (long)j.out_off + (long)r * j.ldo + c;
                if (e < 0 || e >= slab_floats) return -2;
                if (hits[e]++) return -3;
                ++total;
            }
        if (j.cs_off >= 0)
            for (int r = 0; r < j.n_valid; ++r) {
                if (j.cs_off + r >= slab_floats) return -4;
                if (hits[j.cs_off + r]++) return -5;
                ++total;
            }
    }
    return total == expected_entries ? 0 : -6;
}
