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


// B3, first half: sum the per-slice slabs in a fixed order (deterministic).  16 bytes per thread, four independent partial
// sums (slices k = 0, 1, 2, 3 mod 4) so that the loads of consecutive slices overlap; slab_floats is a multiple of 4.
template <int MODEL>
__global__ void __launch_bounds__(256) k_grad_reduce(const float* __restrict__ slabs, int n_slices, int slab_floats, float* __restrict__ sum) {
    const int n4 = slab_floats >> 2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += gridDim.x * blockDim.x) {
        const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + e;
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 4 <= n_slices; k += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += src[(int64_t)(k + q) * n4];
        }
        for (; k < n_slices; ++k) a[k & 3] += src[(int64_t)k * n4];
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
