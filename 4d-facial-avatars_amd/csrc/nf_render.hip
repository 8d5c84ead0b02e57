// K5 volume integrator (fwd + bwd), K6 inverse-CDF sampler, K7 per-ray sort, and the fused
// hierarchical-resampling kernel (z_mid -> sample_pdf -> sort(cat)).
//
// Mapping: ONE 64-lane wavefront per ray, samples strided over lanes (sample s lives in lane s%64,
// chunk s/64) so every global access of a wave is a contiguous, coalesced run.  The transmittance
// cumprod, the CDF cumsum and the reductions are wavefront scans / butterflies on cross-lane shuffles
// (no LDS, no block barrier); only the CDF search table and the bitonic sort use wave-private LDS.
// These kernels are HBM-bound (K5: 20 B/sample read + 4 B/sample written).
#include "nf_common.h"

#define NF_RAYS_PER_BLOCK 4                   // 4 waves = 256 threads per block, one ray each
#define NF_MAX_CHUNKS 16                      // <= 1024 samples per ray
#define NF_MAX_BINS 512
#define NF_MAX_SORT 1024

__device__ __forceinline__ int nf_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v) {       // inclusive
    const int l = nf_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v) {       // inclusive
    const int l = nf_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o, 64); if (l >= o) v *= t; }
    return v;
}

// ---------------------------------------------------------------------------------------------
// Per-sample quantities of volume_render_radiance_field (reference volume_rendering_utils.py:19-55)
// ---------------------------------------------------------------------------------------------
struct NfSample {
    float c[3];     // colour entering the sum (sigmoid(raw) or the background for the last sample)
    float alpha;    // 1 - exp(-sigma*dist)
    float dist;
    float pre;      // raw_sigma + noise (ReLU argument)
    bool  is_bg;
};

__device__ __forceinline__ float nf_sigmoid(float x) { return nf_div(1.0f, nf_add(1.0f, expf(-x))); }

// mode bits: 1 = scale the sample spacing by |rd| (V:26), 2 = add 1e-6 to the last density (V:52-53), 4 = report the depth
// map sum(w z) instead of the disparity.  NeRFace = 3; tiny_nerf's render_volume_density (tiny_nerf.py:68-107) = 4.
#define NF_VR_NERFACE 3
#define NF_VR_TINY 4

__device__ __forceinline__ NfSample nf_load_sample(const float4* __restrict__ raw_row, const float* __restrict__ z_row,
                                                   const float* __restrict__ noise_row, const float* __restrict__ bg_ray,
                                                   float rd_norm, int s, int S, int mode = NF_VR_NERFACE) {
    NfSample o;
    const float4 r = raw_row[s];
    const bool last = (s == S - 1);
    o.is_bg = last && bg_ray != nullptr;
    if (o.is_bg) { o.c[0] = bg_ray[0]; o.c[1] = bg_ray[1]; o.c[2] = bg_ray[2]; }
    else { o.c[0] = nf_sigmoid(r.x); o.c[1] = nf_sigmoid(r.y); o.c[2] = nf_sigmoid(r.z); }
    const float d = last ? 1e10f : nf_sub(z_row[s + 1], z_row[s]);
    o.dist = (mode & 1) ? nf_mul(d, rd_norm) : d;
    o.pre = noise_row ? nf_add(r.w, noise_row[s]) : r.w;
    float sigma = fmaxf(o.pre, 0.0f);
    if (last && (mode & 2)) sigma = nf_add(sigma, 1e-6f);         // V:52-53
    o.alpha = nf_sub(1.0f, expf(-nf_mul(sigma, o.dist)));
    return o;
}

__device__ __forceinline__ float nf_rd_norm(const float* __restrict__ rd_ray) {
    const float x = rd_ray[0], y = rd_ray[1], zc = rd_ray[2];
    return sqrtf(nf_add(nf_add(nf_mul(x, x), nf_mul(y, y)), nf_mul(zc, zc)));
}

// ---------------------------------------------------------------------------------------------
// K5 forward
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_volume_render_fwd(const float* __restrict__ raw, const float* __restrict__ z,
                                                           const float* __restrict__ rd, const float* __restrict__ noise,
                                                           const float* __restrict__ bg, int64_t n_rays, int S,
                                                           int white_bg, float* __restrict__ rgb, float* __restrict__ disp,
                                                           float* __restrict__ acc, float* __restrict__ weights, int mode) {
    const int lane = nf_lane();
    const int64_t ray = (int64_t)blockIdx.x * NF_RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float4* raw_row = reinterpret_cast<const float4*>(raw) + ray * S;
    const float* z_row = z + ray * S;
    const float* noise_row = noise ? noise + ray * S : nullptr;
    const float* bg_ray = bg ? bg + ray * 3 : nullptr;
    const float norm = (mode & 1) ? nf_rd_norm(rd + ray * 3) : 1.0f;
    float carry = 1.0f;                        // running exclusive transmittance at the chunk start
    float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f, a_w = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool on = s < S;
        float alpha = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, zz = 0.f;
        if (on) {
            const NfSample q = nf_load_sample(raw_row, z_row, noise_row, bg_ray, norm, s, S, mode);
            alpha = q.alpha; c0 = q.c[0]; c1 = q.c[1]; c2 = q.c[2]; zz = z_row[s];
        }
        const float b = on ? nf_add(nf_sub(1.0f, alpha), 1e-10f) : 1.0f;
        const float incl = wave_scan_mul(b);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T = nf_mul(carry, excl);
        const float w = nf_mul(alpha, T);
        if (on && weights) weights[ray * S + s] = w;
        a_r += w * c0; a_g += w * c1; a_b += w * c2; a_d += w * zz; a_w += w;
        carry = nf_mul(carry, __shfl(incl, 63, 64));
    }
    a_r = wave_sum(a_r); a_g = wave_sum(a_g); a_b = wave_sum(a_b); a_d = wave_sum(a_d); a_w = wave_sum(a_w);
    if (lane == 0) {
        if (white_bg) { const float k = nf_sub(1.0f, a_w); a_r += k; a_g += k; a_b += k; }
        rgb[ray * 3 + 0] = a_r; rgb[ray * 3 + 1] = a_g; rgb[ray * 3 + 2] = a_b;
        acc[ray] = a_w;
        disp[ray] = (mode & 4) ? a_d : nf_div(1.0f, fmaxf(1e-10f, nf_div(a_d, a_w)));
    }
}

extern "C" int nf_volume_render_fwd(const float* raw, const float* z, const float* rd, const float* noise, const float* bg,
                                    int64_t n_rays, int n_samples, int white_background, float* rgb, float* disp,
                                    float* acc, float* weights, nf_stream_t stream) {
    if (n_rays == 0) return 0;                       // nothing to do (empty tensors have NULL data pointers)
    if (!raw || !z || !rd || !rgb || !disp || !acc || !weights || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t grid = (n_rays + NF_RAYS_PER_BLOCK - 1) / NF_RAYS_PER_BLOCK;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_volume_render_fwd, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), raw, z, rd, noise, bg, n_rays,
                       n_samples, white_background, rgb, disp, acc, weights, NF_VR_NERFACE);
    NF_RETURN_LAUNCH();
}

// tiny_nerf's render_volume_density (reference tiny_nerf.py:68-107): no background sample, no +1e-6, spacing not scaled by
// |rd|; returns (rgb_map, depth_map, acc_map).  depth: (n_rays, n_samples).
extern "C" int nf_render_volume_density(const float* raw, const float* depth, int64_t n_rays, int n_samples, float* rgb,
                                        float* depth_map, float* acc, nf_stream_t stream) {
    if (n_rays == 0) return 0;                       // nothing to do (empty tensors have NULL data pointers)
    if (!raw || !depth || !rgb || !depth_map || !acc || n_rays < 0 || n_samples <= 0) return NF_EINVAL;
    const int64_t grid = (n_rays + NF_RAYS_PER_BLOCK - 1) / NF_RAYS_PER_BLOCK;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_volume_render_fwd, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), raw, depth, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, n_rays, n_samples, 0, rgb, depth_map, acc, (float*)nullptr,
                       NF_VR_TINY);
    NF_RETURN_LAUNCH();
}

// ---------------------------------------------------------------------------------------------
// K5 backward: d_raw from d_rgb.   With b_j = 1-alpha_j+1e-10, T_i = prod_{j<i} b_j, w_i = alpha_i T_i:
//   dL/dw_i     = <d_rgb, c_i>  (+ white-bg term: -sum(d_rgb))
//   dL/dalpha_i = dw_i T_i - (sum_{k>i} dw_k w_k) / b_i
//   dL/dsigma_i = dL/dalpha_i * dist_i * (1-alpha_i);   dL/dpre_i = [pre_i > 0] dL/dsigma_i
//   dL/draw_rgb = w_i d_rgb c(1-c)   (zero for the background sample, whose colour is a constant)
// Two passes over the ray's chunks; the forward quantities are recomputed, not stored.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__global__ void __launch_bounds__(256) k_volume_render_bwd(const float* __restrict__ raw, const float* __restrict__ z,
                                                           const float* __restrict__ rd, const float* __restrict__ noise,
                                                           const float* __restrict__ bg, const float* __restrict__ d_rgb,
                                                           int64_t n_rays, int S, int white_bg, float* __restrict__ d_raw, int mode) {
    const int lane = nf_lane();
    const int64_t ray = (int64_t)blockIdx.x * NF_RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float4* raw_row = reinterpret_cast<const float4*>(raw) + ray * S;
    const float* z_row = z + ray * S;
    const float* noise_row = noise ? noise + ray * S : nullptr;
    const float* bg_ray = bg ? bg + ray * 3 : nullptr;
    const float norm = (mode & 1) ? nf_rd_norm(rd + ray * 3) : 1.0f;
    const float g0 = d_rgb[ray * 3 + 0], g1 = d_rgb[ray * 3 + 1], g2 = d_rgb[ray * 3 + 2];
    const float gw_white = white_bg ? -(g0 + g1 + g2) : 0.0f;

    float T[NCH], w[NCH], dw[NCH], alpha[NCH], dist[NCH], pre[NCH], c[NCH][3];
    bool isbg[NCH];
    float carry = 1.0f, total = 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int s = k * 64 + lane;
        const bool on = s < S;
        alpha[k] = 0.f; dist[k] = 0.f; pre[k] = 0.f; c[k][0] = c[k][1] = c[k][2] = 0.f; isbg[k] = false;
        if (on) {
            const NfSample q = nf_load_sample(raw_row, z_row, noise_row, bg_ray, norm, s, S, mode);
            alpha[k] = q.alpha; dist[k] = q.dist; pre[k] = q.pre; isbg[k] = q.is_bg;
            c[k][0] = q.c[0]; c[k][1] = q.c[1]; c[k][2] = q.c[2];
        }
        const float b = on ? nf_add(nf_sub(1.0f, alpha[k]), 1e-10f) : 1.0f;
        const float incl = wave_scan_mul(b);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        T[k] = carry * excl;
        w[k] = alpha[k] * T[k];
        dw[k] = on ? (g0 * c[k][0] + g1 * c[k][1] + g2 * c[k][2] + gw_white) : 0.0f;
        carry *= __shfl(incl, 63, 64);
        total += wave_sum(dw[k] * w[k]);
    }
    float prefix = 0.0f;                       // sum over all earlier chunks of dw*w
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int s = k * 64 + lane;
        const float v = dw[k] * w[k];
        const float incl = wave_scan_add(v);
        const float suffix = total - (prefix + incl);      // sum_{j>s} dw_j w_j
        prefix += __shfl(incl, 63, 64);
        if (s < S) {
            const float b = (1.0f - alpha[k]) + 1e-10f;
            const float d_alpha = dw[k] * T[k] - suffix / b;
            const float d_sigma = d_alpha * dist[k] * (1.0f - alpha[k]);
            float4 o;
            if (isbg[k]) { o.x = o.y = o.z = 0.0f; }
            else {
                o.x = w[k] * g0 * c[k][0] * (1.0f - c[k][0]);
                o.y = w[k] * g1 * c[k][1] * (1.0f - c[k][1]);
                o.z = w[k] * g2 * c[k][2] * (1.0f - c[k][2]);
            }
            o.w = pre[k] > 0.0f ? d_sigma : 0.0f;
            reinterpret_cast<float4*>(d_raw)[ray * S + s] = o;
        }
    }
}

static int nf_volume_render_bwd_impl(const float* raw, const float* z, const float* rd, const float* noise, const float* bg,
                                     const float* d_rgb, int64_t n_rays, int n_samples, int white_background, float* d_raw,
                                     int mode, nf_stream_t stream) {
    if (n_rays == 0) return 0;                       // nothing to do (empty tensors have NULL data pointers)
    if (!raw || !z || (!rd && (mode & 1)) || !d_rgb || !d_raw || n_rays < 0 || n_samples <= 0 || n_samples > 64 * NF_MAX_CHUNKS) return NF_EINVAL;
    const int64_t grid = (n_rays + NF_RAYS_PER_BLOCK - 1) / NF_RAYS_PER_BLOCK;
    if (grid > 0x7fffffff) return NF_EINVAL;
    const int nch = (n_samples + 63) / 64;
#define NF_BWD(N)                                                                                                         \
    hipLaunchKernelGGL(k_volume_render_bwd<N>, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), raw, z, rd, noise, bg,   \
                       d_rgb, n_rays, n_samples, white_background, d_raw, mode)
    if (nch == 1) NF_BWD(1); else if (nch == 2) NF_BWD(2); else if (nch == 3) NF_BWD(3); else if (nch == 4) NF_BWD(4);
    else if (nch <= 8) NF_BWD(8); else NF_BWD(16);
#undef NF_BWD
    NF_RETURN_LAUNCH();
}

extern "C" int nf_volume_render_bwd(const float* raw, const float* z, const float* rd, const float* noise, const float* bg,
                                    const float* d_rgb, int64_t n_rays, int n_samples, int white_background, float* d_raw,
                                    nf_stream_t stream) {
    return nf_volume_render_bwd_impl(raw, z, rd, noise, bg, d_rgb, n_rays, n_samples, white_background, d_raw, NF_VR_NERFACE, stream);
}

// Backward of tiny_nerf's render_volume_density (tiny_nerf.py:68-107) w.r.t. the radiance field, for the rgb output (the only
// one the tiny trainer's loss reads, tiny_nerf.py:291-302): d_rgb (n_rays, 3) -> d_raw (n_rays, n_samples, 4).
extern "C" int nf_render_volume_density_bwd(const float* raw, const float* depth, const float* d_rgb, int64_t n_rays, int n_samples,
                                            float* d_raw, nf_stream_t stream) {
    return nf_volume_render_bwd_impl(raw, depth, nullptr, nullptr, nullptr, d_rgb, n_rays, n_samples, 0, d_raw, NF_VR_TINY, stream);
}

// ---------------------------------------------------------------------------------------------
// K6: sample_pdf_2 (reference nerf_helpers.py:344-387), wave-private LDS tables.
//   cdf[0] = 0, cdf[i] = cumsum((w+1e-5)/sum(w+1e-5))[i-1];  idx = #{cdf <= u} (searchsorted right=True)
// `lds_cdf`/`lds_bins` are this wave's tables (n_bins entries each); returns via callback-free loop.
// ---------------------------------------------------------------------------------------------
// The table is BIT-IDENTICAL to what torch's CPU kernels give the reference (H:349-353), so that searchsorted lands in the
// same bin for every u (tests/test_gpu_kernels.py::test_sample_pdf_bit_exact):
//   * torch.sum over a contiguous float row (ATen SumKernel, vectorized_inner_sum): 8-lane vector partial sums with 4-way
//     ILP -- P[k][l] accumulates x[(4i+k)*8+l], leftover whole vectors go to P[0], P[0] += P[1..3], then a scalar
//     accumulator takes the tail elements and finally the 8 lanes of P[0], all in float; rows shorter than one vector use
//     the same scheme on scalars (row_sum);
//   * torch.cumsum (cumsum_cpu_kernel): one sequential accumulator in DOUBLE, rounded to float per element.
// Every wave of the block must call this (it contains block barriers); `on` = this wave has a ray.
__device__ __forceinline__ void nf_build_cdf(const float* __restrict__ w_row, int n_w, float* lds_cdf, bool on) {
    const int lane = nf_lane();
    float* x = lds_cdf + 1;                                     // x[i] = w[i] + 1e-5, later pdf[i], later cdf[i+1]
    if (on) for (int i = lane; i < n_w; i += 64) x[i] = nf_add(w_row[i], 1e-5f);
    __syncthreads();
    float sum = 0.0f;
    if (on) {
        if (n_w < 8) {
            if (lane == 0) {
                float p[4] = {0.f, 0.f, 0.f, 0.f};
                const int q = n_w >> 2;
                for (int i = 0; i < q; ++i)
                    for (int k = 0; k < 4; ++k) p[k] = nf_add(p[k], x[4 * i + k]);
                for (int i = q * 4; i < n_w; ++i) p[0] = nf_add(p[0], x[i]);
                sum = nf_add(nf_add(nf_add(p[0], p[1]), p[2]), p[3]);
            }
        } else {
            const int V = n_w >> 3, q = V >> 2, k = (lane >> 3) & 3, l = lane & 7;
            float P = 0.0f;                                     // lanes 0..31: P[k][l]
            for (int i = 0; i < q; ++i) P = nf_add(P, x[((4 * i + k) << 3) + l]);
            if (k == 0) for (int i = q * 4; i < V; ++i) P = nf_add(P, x[(i << 3) + l]);
            const float p1 = __shfl(P, 8 + l, 64), p2 = __shfl(P, 16 + l, 64), p3 = __shfl(P, 24 + l, 64);
            P = nf_add(nf_add(nf_add(P, p1), p2), p3);          // meaningful in lanes 0..7
            float fin = 0.0f;
            for (int i = V << 3; i < n_w; ++i) fin = nf_add(fin, x[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) fin = nf_add(fin, __shfl(P, j, 64));
            sum = fin;
        }
        sum = __shfl(sum, 0, 64);
    }
    __syncthreads();
    if (on) for (int i = lane; i < n_w; i += 64) x[i] = nf_div(x[i], sum);
    __syncthreads();
    if (on && lane == 0) {
        lds_cdf[0] = 0.0f;
        double acc = 0.0;
        for (int i = 0; i < n_w; ++i) { acc += (double)x[i]; x[i] = (float)acc; }
    }
}

__device__ __forceinline__ float nf_invert_cdf(const float* lds_cdf, const float* lds_bins, int n_bins, float u,
                                               int* idx_out = nullptr) {
    int lo = 0, hi = n_bins;                   // first index with cdf > u  (== count of cdf <= u)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (lds_cdf[mid] <= u) lo = mid + 1; else hi = mid; }
    if (idx_out) *idx_out = lo;                // torch.searchsorted(cdf, u, right=True), H:368
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < n_bins - 1 ? lo : n_bins - 1;
    const float cb = lds_cdf[below], ca = lds_cdf[above];
    const float bb = lds_bins[below], ba = lds_bins[above];
    float den = nf_sub(ca, cb);
    if (den < 1e-5f) den = 1.0f;
    const float t = nf_div(nf_sub(u, cb), den);
    return nf_add(bb, nf_mul(t, nf_sub(ba, bb)));
}

__global__ void __launch_bounds__(256) k_sample_pdf(const float* __restrict__ bins, const float* __restrict__ weights,
                                                    const float* __restrict__ u, int64_t u_stride, int64_t n_rays,
                                                    int n_bins, int n_out, float* __restrict__ samples,
                                                    int* __restrict__ inds_out, float* __restrict__ cdf_out) {
    __shared__ float lds[NF_RAYS_PER_BLOCK][2 * NF_MAX_BINS];
    const int lane = nf_lane(), wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * NF_RAYS_PER_BLOCK + wv;
    float* cdf = lds[wv];
    float* lb = lds[wv] + NF_MAX_BINS;
    const bool on = ray < n_rays;
    nf_build_cdf(weights + ray * (n_bins - 1), n_bins - 1, cdf, on);
    if (on) for (int i = lane; i < n_bins; i += 64) lb[i] = bins[ray * n_bins + i];
    __syncthreads();
    if (on) {
        for (int j = lane; j < n_out; j += 64) {
            int idx;
            samples[ray * n_out + j] = nf_invert_cdf(cdf, lb, n_bins, u[ray * u_stride + j], &idx);
            if (inds_out) inds_out[ray * n_out + j] = idx;
        }
        if (cdf_out) for (int i = lane; i < n_bins; i += 64) cdf_out[ray * n_bins + i] = cdf[i];
    }
}

extern "C" int nf_sample_pdf_ex(const float* bins, const float* weights, const float* u, int64_t u_row_stride, int64_t n_rays,
                                int n_bins, int n_out, float* samples, int* inds, float* cdf, nf_stream_t stream) {
    if (n_rays == 0) return 0;                       // nothing to do (empty tensors have NULL data pointers)
    if (!bins || !weights || !u || !samples || n_rays < 0 || n_bins < 2 || n_bins > NF_MAX_BINS || n_out <= 0) return NF_EINVAL;
    const int64_t grid = (n_rays + NF_RAYS_PER_BLOCK - 1) / NF_RAYS_PER_BLOCK;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_sample_pdf, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), bins, weights, u, u_row_stride, n_rays,
                       n_bins, n_out, samples, inds, cdf);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_row_stride, int64_t n_rays,
                             int n_bins, int n_out, float* samples, nf_stream_t stream) {
    return nf_sample_pdf_ex(bins, weights, u, u_row_stride, n_rays, n_bins, n_out, samples, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------
// K7: ascending bitonic sort of one row per wave in wave-private LDS (n padded to a power of two
// with +inf).  Values only (the reference discards torch.sort's indices, T:126).
// ---------------------------------------------------------------------------------------------
// The order of torch.sort: ascending, NaNs after +inf.  A plain `a > b` is no order once a NaN is present (every comparison with it is
// false): a compare-exchange network then leaves the row unsorted AND lets the +inf padding of the power-of-two tail migrate below n,
// pushing NaNs out of the part that is written back.  nf_sort_gt is a total order: finite / inf values by value (ties and -0 / +0
// exactly as `a > b` treats them), then NaNs (all equal), then the padding entries (NF_SORT_PAD: a NaN payload no arithmetic produces).
#define NF_SORT_PAD __int_as_float(0x7fffffff)
__device__ __forceinline__ int nf_sort_class(float x) {
    const int b = __float_as_int(x);
    return (b & 0x7fffffff) > 0x7f800000 ? (b == 0x7fffffff ? 2 : 1) : 0;
}
__device__ __forceinline__ bool nf_sort_gt(float a, float b) { return a > b || nf_sort_class(a) > nf_sort_class(b); }

__device__ __forceinline__ void nf_bitonic_sort(float* buf, int n_pow2) {
    const int lane = nf_lane();
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n_pow2 >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // lower index of the pair
                const int p = i | j;
                const bool up = (i & k) == 0;
                const float a = buf[i], b = buf[p];
                if (nf_sort_gt(a, b) == up) { buf[i] = b; buf[p] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int nf_next_pow2(int n) { int p = 2; while (p < n) p <<= 1; return p; }

__global__ void __launch_bounds__(256) k_sort_rows(const float* __restrict__ in, int64_t n_rows, int n_cols,
                                                   float* __restrict__ out) {
    __shared__ float lds[NF_RAYS_PER_BLOCK][NF_MAX_SORT];
    const int lane = nf_lane(), wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * NF_RAYS_PER_BLOCK + wv;
    const bool on = row < n_rows;
    const int np2 = nf_next_pow2(n_cols);
    for (int i = lane; i < np2; i += 64) lds[wv][i] = (on && i < n_cols) ? in[row * n_cols + i] : NF_SORT_PAD;
    __syncthreads();
    nf_bitonic_sort(lds[wv], np2);
    if (on) for (int i = lane; i < n_cols; i += 64) out[row * n_cols + i] = lds[wv][i];
}

extern "C" int nf_sort_rows(const float* in, int64_t n_rows, int n_cols, float* out, nf_stream_t stream) {
    if (n_rows == 0) return 0;                       // nothing to do (empty tensors have NULL data pointers)
    if (!in || !out || n_rows < 0 || n_cols <= 0 || n_cols > NF_MAX_SORT) return NF_EINVAL;
    const int64_t grid = (n_rows + NF_RAYS_PER_BLOCK - 1) / NF_RAYS_PER_BLOCK;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), in, n_rows, n_cols, out);
    NF_RETURN_LAUNCH();
}

// ---------------------------------------------------------------------------------------------
// K6+K7 fused (reference train_utils.py:116-126): bins = 0.5*(z[1:]+z[:-1]), weights = w[1:-1],
// z_samples = sample_pdf(...), z_fine = sort(cat(z, z_samples)).  One wave per ray; everything stays
// in wave-private LDS between the stages: HBM traffic is 8*Nc read + 4*(Nc+Nf) (+4*Nf) written per ray.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_resample_merge(const float* __restrict__ zc, const float* __restrict__ wc,
                                                        const float* __restrict__ u, int64_t u_stride, int64_t n_rays,
                                                        int nc, int nf, float* __restrict__ z_samples,
                                                        float* __restrict__ z_fine) {
    __shared__ float lds[NF_RAYS_PER_BLOCK][2 * NF_MAX_BINS + NF_MAX_SORT];
    const int lane = nf_lane(), wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * NF_RAYS_PER_BLOCK + wv;
    const bool on = ray < n_rays;
    float* cdf = lds[wv];
    float* lb = lds[wv] + NF_MAX_BINS;
    float* srt = lds[wv] + 2 * NF_MAX_BINS;
    const int n_bins = nc - 1, nt = nc + nf, np2 = nf_next_pow2(nt);
    nf_build_cdf(wc + ray * nc + 1, nc - 2, cdf, on);
    if (on) {
        for (int i = lane; i < nc; i += 64) {
            const float zi = zc[ray * nc + i];
            srt[i] = zi;
            if (i < n_bins) lb[i] = nf_mul(0.5f, nf_add(zc[ray * nc + i + 1], zi));
        }
    }
    for (int i = nt + lane; i < np2; i += 64) srt[i] = NF_SORT_PAD;
    __syncthreads();
    if (on)
        for (int j = lane; j < nf; j += 64) {
            const float v = nf_invert_cdf(cdf, lb, n_bins, u[ray * u_stride + j]);
            srt[nc + j] = v;
            if (z_samples) z_samples[ray * nf + j] = v;
        }
    __syncthreads();
    nf_bitonic_sort(srt, np2);
    if (on) for (int i = lane; i < nt; i += 64) z_fine[ray * nt + i] = srt[i];
}

// ---------------------------------------------------------------------------------------------
// K6+K7, the shipped sizes (Nc <= 128, Nf <= 128: 64 + 128 in eval, 64 + 64 in training): the same results as
// k_resample_merge from a wave that never waits for the block, never runs one lane alone and keeps its cross-lane traffic on
// the vector unit (DPP / v_permlane*_swap) instead of the LDS crossbar.  The kernel is VALU-issue-bound (64 rays per SIMD, a
// few hundred instructions each), so every stage is written for instruction count:
//   * the torch-exact CDF table (nf_build_cdf's arithmetic) without block barriers -- the tables are wave-private and one
//     wave's LDS instructions execute in order -- and with the sequential DOUBLE cumsum (62 dependent adds on one lane)
//     replaced by a DPP wave prefix sum in double WHERE THAT IS EXACT: if every pdf entry is 0 or has 2^-22 <= |p| < 2, every
//     entry is a multiple of 2^-45 and every sum of up to 126 of them is a multiple of 2^-45 below 2^8, i.e. representable
//     in double -- no addition rounds, so the association order cannot matter and the scan equals torch's sequential loop
//     bit for bit.  (pdf = (w + 1e-5) / sum >= 9.9e-6 for weights in [0, 1]: the guard holds for every ray of the hot path;
//     any other input takes the one-lane loop.)
//   * searchsorted(right) as a branch-free descent over power-of-two steps (the count of table entries <= u);
//   * the Nf samples are sorted in registers: a bitonic network over 2 values per lane, ONE v_min per compare-exchange (sign-domain
//     trick, see nf_sort128_regs): exchanges at lane distance 1, 2, 8 are a DPP-fused v_min, at distance 4 two banked ones, and at
//     distance 16 / 32 / 64 IN-LANE after v_permlane16_swap / v_permlane32_swap have moved the element-index bit concerned into the
//     register index (the element an entry holds is tracked through the layout changes; nothing is moved back).  Samples that come
//     out of the inverse CDF ascending already (deterministic abscissae) skip the network;
//   * the merge with the coarse depths (ascending: checked, else the general kernel's full sort runs) is by counting: a
//     sample from bin b lies between the bin's mid-points, i.e. z[b] <= s <= z[b+2], so its rank among the depths is b + 1 or
//     b + 2 -- three LDS reads decide and verify (any lane that cannot verify sends the wave to the binary search); an LDS
//     histogram of those ranks, prefix-summed (DPP), gives every depth its output slot i + #{samples < z[i]}; the sorted
//     samples fill the remaining slots in order, which each OUTPUT slot works out for itself from a ballot of the depth slots
//     (no scatter, the row leaves coalesced).  The sorted multiset is what torch.sort returns (T:126 keeps the values only).
// ---------------------------------------------------------------------------------------------
#define NF_RS_MAXC 128
#define NF_RS_MAXF 128
// floats per wave: cdf[128] + 128 (the slot flags, 256 entries, take both once the table is dead) | A[128] B[128] | x[128] / hist[132]
#define NF_RS_FLOATS (128 + 128 + 256 + 132)
#define NF_DPP_ROW_SHL(n) (0x100 + (n))                     // lane i <- lane i + n of its row of 16
#define NF_DPP_ROW_SHR(n) (0x110 + (n))                     // lane i <- lane i - n
#define NF_DPP_BCAST15 0x142                                // lane 15 of a row -> every lane of the next row
#define NF_DPP_BCAST31 0x143                                // lane 31 -> every lane of rows 2, 3
__device__ __forceinline__ void nf_wave_sync() {            // order one wave's LDS accesses for the compiler (the hardware runs them in order)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int CTRL, int ROW_MASK, int BANK_MASK, bool ZERO>
__device__ __forceinline__ float nf_dpp_f(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, ZERO));
}
template <int CTRL, int ROW_MASK, int BANK_MASK, bool ZERO>
__device__ __forceinline__ int nf_dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, ZERO); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double nf_dpp_d0(double v) {     // the permuted value where a source lane exists and the row is enabled, else 0.0
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// inclusive prefix sums over the 64 lanes (Kogge-Stone inside rows of 16, then the row totals broadcast forward)
__device__ __forceinline__ double nf_scan64_f64(double a) {
    a += nf_dpp_d0<NF_DPP_ROW_SHR(1), 0xF>(a);
    a += nf_dpp_d0<NF_DPP_ROW_SHR(2), 0xF>(a);
    a += nf_dpp_d0<NF_DPP_ROW_SHR(4), 0xF>(a);
    a += nf_dpp_d0<NF_DPP_ROW_SHR(8), 0xF>(a);
    a += nf_dpp_d0<NF_DPP_BCAST15, 0xA>(a);
    a += nf_dpp_d0<NF_DPP_BCAST31, 0xC>(a);
    return a;
}
__device__ __forceinline__ int nf_scan64_i32(int a) {
    a += nf_dpp_i<NF_DPP_ROW_SHR(1), 0xF, 0xF, true>(0, a);
    a += nf_dpp_i<NF_DPP_ROW_SHR(2), 0xF, 0xF, true>(0, a);
    a += nf_dpp_i<NF_DPP_ROW_SHR(4), 0xF, 0xF, true>(0, a);
    a += nf_dpp_i<NF_DPP_ROW_SHR(8), 0xF, 0xF, true>(0, a);
    a += nf_dpp_i<NF_DPP_BCAST15, 0xA, 0xF, false>(0, a);
    a += nf_dpp_i<NF_DPP_BCAST31, 0xC, 0xF, false>(0, a);
    return a;
}
__device__ __forceinline__ float nf_readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// wave-private version of nf_build_cdf (same arithmetic, see there); 1 <= n_w <= NF_RS_MAXC - 2; every lane of the wave calls it;
// x: scratch of n_w floats.  Leaves cdf[0 .. n_w] in lds_cdf, padded with +inf to 128 entries.
__device__ __forceinline__ void nf_build_cdf_wave(const float* __restrict__ w_row, int n_w, float* lds_cdf, float* x, int lane) {
    float xr[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane + 64 * r;
        xr[r] = 0.0f;
        if (i < n_w) { xr[r] = nf_add(w_row[i], 1e-5f); x[i] = xr[r]; }
    }
    nf_wave_sync();
    float sum;
    if (n_w < 8) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        const int q = n_w >> 2;
        for (int i = 0; i < q; ++i)
            for (int k = 0; k < 4; ++k) p[k] = nf_add(p[k], x[4 * i + k]);
        for (int i = q * 4; i < n_w; ++i) p[0] = nf_add(p[0], x[i]);
        sum = nf_add(nf_add(nf_add(p[0], p[1]), p[2]), p[3]);                 // (every lane computes the same value)
    } else {
        const int V = n_w >> 3, q = V >> 2;
        const float* xl = x + (lane & 31);                  // lanes 0..31: P[k][l] with 8 k + l = lane
        float P = 0.0f;
        for (int i = 0; i < q; ++i) P = nf_add(P, xl[32 * i]);
        const float* x8 = x + (lane & 7);
        for (int i = q * 4; i < V; ++i) { const float t = nf_add(P, x8[8 * i]); P = lane < 8 ? t : P; }      // leftover whole vectors -> P[0]
        // P[0] += P[1], += P[2], += P[3] (lanes 0..7): lane + 8 by a row shift; row 1 (lanes 16..31) brought to row 0 by one swap
        const float p1 = nf_dpp_f<NF_DPP_ROW_SHL(8), 0xF, 0xF, true>(0.0f, P);
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(P), __float_as_uint(P), false, false);
        const float R1 = __uint_as_float(sw[1]);            // rows {1, 1, 3, 3} of P
        const float p3 = nf_dpp_f<NF_DPP_ROW_SHL(8), 0xF, 0xF, true>(0.0f, R1);
        P = nf_add(nf_add(nf_add(P, p1), R1), p3);          // meaningful in lanes 0..7
        float fin = 0.0f;
        for (int i = V << 3; i < n_w; ++i) fin = nf_add(fin, x[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) fin = nf_add(fin, nf_readlane_f(P, j));
        sum = fin;
    }
    // pdf, and the exactness guard of the parallel cumsum (0, or 2^-22 <= |p| < 2; NaN fails)
    float v[2];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane + 64 * r;
        v[r] = i < n_w ? nf_div(xr[r], sum) : 0.0f;
        const unsigned m = __float_as_uint(v[r]) & 0x7fffffffu;
        ok = ok && (m == 0u || (m - 0x34800000u) < (0x40000000u - 0x34800000u));
    }
    if (__all(ok)) {
        const double a0 = nf_scan64_f64((double)v[0]);
        float c1 = INFINITY;                                // entries past the table: +inf (the inversion's descent never looks at a count)
        if (n_w > 64) {
            const unsigned long long b = __double_as_longlong(a0);
            const double carry = __longlong_as_double(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) |
                                                      (unsigned)__builtin_amdgcn_readlane((int)b, 63));
            const double a1 = nf_scan64_f64((double)v[1]) + carry;
            if (lane + 64 < n_w) c1 = (float)a1;
        }
        lds_cdf[1 + lane] = lane < n_w ? (float)a0 : INFINITY;
        if (lane < 63) lds_cdf[65 + lane] = c1;
        if (lane == 0) lds_cdf[0] = 0.0f;
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int i = lane + 64 * r; if (i < n_w) x[i] = v[r]; }
        nf_wave_sync();
        if (lane == 0) {
            lds_cdf[0] = 0.0f;
            double acc = 0.0;
            for (int i = 0; i < n_w; ++i) { acc += (double)x[i]; lds_cdf[1 + i] = (float)acc; }
        }
        for (int i = n_w + 1 + lane; i < 128; i += 64) lds_cdf[i] = INFINITY;
    }
    nf_wave_sync();
}

// ---- register bitonic sort of 128 values, v[r] = element 64 r + lane on entry ------------------------------------------------------
// lanes that take the minimum in a compare-exchange at lane-bit JB inside blocks whose direction is lane-bit UB (-1: ascending)
constexpr unsigned long long nf_takemin_mask(int jb, int ub) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const int lower = ((l >> jb) & 1) == 0, up = ub < 0 ? 1 : (((l >> ub) & 1) == 0);
        if (lower == up) m |= 1ull << l;
    }
    return m;
}
constexpr unsigned long long nf_bitclear_mask(int b) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (((l >> b) & 1) == 0) m |= 1ull << l;
    return m;
}
// The network runs in a SIGN DOMAIN so that a compare-exchange is ONE instruction: a lane that is to keep the minimum of its pair holds
// its value as it is, a lane that is to keep the maximum holds it NEGATED; then w <- min(w, -partner(w)) is right for both (the plain
// lane gets min(x, y), the negated one min(-y, -x) = -max(x, y); DPP takes the negation as a source modifier).  Between two levels the
// lanes whose role changes flip sign (one v_cndmask with a negated source and a constant lane mask).  Which lanes are plain is compile-time
// bookkeeping (NfDom), also across the v_permlane swaps, which move values -- and their signs -- between the two registers.
typedef unsigned long long nf_u64;
struct NfDom { nf_u64 s0, s1; };                            // lanes of register 0 / 1 that hold plain values
struct NfSortStep { int kind, jb, ub; };                    // kind 0: DPP level at lane bit jb; 1: between the registers; 2: swap16; 3: swap32
constexpr NfSortStep NF_SORT_STEPS[] = {
    // element bits b0..b3 = lane bits 0..3 throughout; (register, lane bit 4, lane bit 5) = (b6, b4, b5) on entry
    {0, 0, 1},                                              // k = 2 (direction = lane bit 1)
    {0, 1, 2}, {0, 0, 2},                                   // k = 4
    {0, 2, 3}, {0, 1, 3}, {0, 0, 3},                        // k = 8
    {0, 3, 4}, {0, 2, 4}, {0, 1, 4}, {0, 0, 4},             // k = 16 (direction b4 = lane bit 4)
    {2, 0, 0}, {1, 0, 5},                                   // k = 32: swap16 -> (b4, b6, b5); direction b5 = lane bit 5
    {0, 3, 5}, {0, 2, 5}, {0, 1, 5}, {0, 0, 5},
    {3, 0, 0}, {1, 0, 4},                                   // k = 64 (direction b6 = lane bit 4): swap32 -> (b5, b6, b4)
    {3, 0, 0}, {1, 0, 4},                                   //          swap32 -> (b4, b6, b5)
    {0, 3, 4}, {0, 2, 4}, {0, 1, 4}, {0, 0, 4},
    {2, 0, 0}, {1, 0, -1},                                  // k = 128 (ascending): swap16 -> (b6, b4, b5)
    {3, 0, 0}, {1, 0, -1},                                  //          swap32 -> (b5, b4, b6)
    {2, 0, 0}, {1, 0, -1},                                  //          swap16 -> (b4, b5, b6)
    {0, 3, -1}, {0, 2, -1}, {0, 1, -1}, {0, 0, -1},
};
constexpr int NF_SORT_NSTEPS = (int)(sizeof(NF_SORT_STEPS) / sizeof(NF_SORT_STEPS[0]));
constexpr NfDom nf_dom_swapped(NfDom d, int sh) {           // v_permlane{16,32}_swap: (register, lane bit) trade places
    NfDom o{0, 0};
    for (int l = 0; l < 64; ++l) {
        const bool odd = (l & sh) != 0;
        const nf_u64 b0 = odd ? (d.s1 >> (l - sh)) & 1 : (d.s0 >> l) & 1;
        const nf_u64 b1 = odd ? (d.s1 >> l) & 1 : (d.s0 >> (l + sh)) & 1;
        o.s0 |= b0 << l;
        o.s1 |= b1 << l;
    }
    return o;
}
constexpr NfDom nf_dom_wanted(NfSortStep st, NfDom cur) {   // the domain step `st` needs on entry (= leaves behind)
    if (st.kind == 0) { const nf_u64 m = nf_takemin_mask(st.jb, st.ub); return NfDom{m, m}; }
    if (st.kind == 1) { const nf_u64 up = st.ub < 0 ? ~0ull : nf_bitclear_mask(st.ub); return NfDom{up, ~up}; }
    return cur;
}
constexpr NfDom nf_dom_before(int n) {                      // what the registers hold before step n
    NfDom d{~0ull, ~0ull};
    for (int i = 0; i < n; ++i) {
        d = nf_dom_wanted(NF_SORT_STEPS[i], d);
        if (NF_SORT_STEPS[i].kind == 2) d = nf_dom_swapped(d, 16);
        if (NF_SORT_STEPS[i].kind == 3) d = nf_dom_swapped(d, 32);
    }
    return d;
}
template <nf_u64 F>
__device__ __forceinline__ void nf_sd_flip(float& w) {
    if constexpr (F != 0) asm("v_cndmask_b32_e64 %0, %1, -%1, %2" : "=v"(w) : "v"(w), "s"(F));
}
// (the s_nop covers the VALU-write -> DPP-read hazard, which the assembler does not see inside an asm statement)
template <int JB>
__device__ __forceinline__ void nf_sd_min_dpp(float& w) {
    float t;
    if constexpr (JB == 0) asm("s_nop 1\n\tv_min_f32_dpp %0, -%1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(w));
    else if constexpr (JB == 1) asm("s_nop 1\n\tv_min_f32_dpp %0, -%1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(w));
    else if constexpr (JB == 2)                             // banks 0, 2 of a row: partner at lane + 4; banks 1, 3: at lane - 4
        asm("s_nop 1\n\tv_min_f32_dpp %0, -%1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_min_f32_dpp %0, -%1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(t) : "v"(w));
    else asm("s_nop 1\n\tv_min_f32_dpp %0, -%1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(w));
    w = t;
}
template <int I>
__device__ __forceinline__ void nf_sort_run(float (&v)[2]) {
    if constexpr (I < NF_SORT_NSTEPS) {
        constexpr NfSortStep st = NF_SORT_STEPS[I];
        constexpr NfDom cur = nf_dom_before(I), want = nf_dom_wanted(st, cur);
        nf_sd_flip<cur.s0 ^ want.s0>(v[0]);
        nf_sd_flip<cur.s1 ^ want.s1>(v[1]);
        if constexpr (st.kind == 0) { nf_sd_min_dpp<st.jb>(v[0]); nf_sd_min_dpp<st.jb>(v[1]); }
        else if constexpr (st.kind == 1) {
            float n0, n1;
            asm("s_nop 1\n\tv_min_f32_e64 %0, %2, -%3\n\tv_min_f32_e64 %1, %3, -%2" : "=&v"(n0), "=&v"(n1) : "v"(v[0]), "v"(v[1]));
            v[0] = n0; v[1] = n1;
        } else if constexpr (st.kind == 2) asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[1]));
        else asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[1]));
        nf_sort_run<I + 1>(v);
    } else {
        constexpr NfDom cur = nf_dom_before(NF_SORT_NSTEPS);
        nf_sd_flip<~cur.s0>(v[0]);                          // back to plain values
        nf_sd_flip<~cur.s1>(v[1]);
    }
}
// Ascending bitonic sort of 128 values, v[r] = element 64 r + lane on entry.  Exchanges at lane distance 1, 2, 8 are one DPP v_min, at
// distance 4 two banked ones; at distance 16 / 32 / 64 the element-index bit concerned is first moved into the register index
// (v_permlane16_swap / v_permlane32_swap; nothing is moved back).  On exit register r of lane l holds sorted element
// ((l >> 5) << 6) | (((l >> 4) & 1) << 5) | (r << 4) | (l & 15).
__device__ __forceinline__ void nf_sort128_regs(float (&v)[2]) { nf_sort_run<0>(v); }
__device__ __forceinline__ int nf_sort128_index(int r, int lane) { return ((lane >> 5) << 6) | (((lane >> 4) & 1) << 5) | (r << 4) | (lane & 15); }

// NC_FIXED / NF_FIXED: the sample counts when they are the shipped ones (64 + 128 in eval, 64 + 64 in training: loop bounds, the second
// register's guards and the > 64 paths fold away; 58 -> 51 us per 65536 rays for the first alone), 0: taken from the arguments
template <int NC_FIXED, int NF_FIXED>
__global__ void __launch_bounds__(256) k_resample_merge_small(const float* __restrict__ zc, const float* __restrict__ wc,
                                                              const float* __restrict__ u, int64_t u_stride, int64_t n_rays,
                                                              int nc_arg, int nf_arg, float* __restrict__ z_samples,
                                                              float* __restrict__ z_fine) {
    const int nc = NC_FIXED ? NC_FIXED : nc_arg, nf = NF_FIXED ? NF_FIXED : nf_arg;
    __shared__ float lds[NF_RAYS_PER_BLOCK][NF_RS_FLOATS];
    const int lane = nf_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ray = (int64_t)blockIdx.x * NF_RAYS_PER_BLOCK + wv;
    if (ray >= n_rays) return;                              // wave-uniform; no block barrier below
    float* cdf = lds[wv];
    int* flags = reinterpret_cast<int*>(cdf);               // 256 entries: the table (dead by then) and the 128 floats behind it
    float* A = cdf + 256;                                    // coarse depths (nc)
    float* B = A + 128;                                     // sorted samples (nf)
    float* xs = B + 128;
    int* hist = reinterpret_cast<int*>(xs);                 // (the pdf scratch is dead by then)
    const int n_bins = nc - 1, nt = nc + nf;
    const float* zrow = zc + ray * nc;
    const float* urow = u + ray * u_stride;
    // every global read of the row is requested before the table is built (their latency runs under it)
    float zr[2], ur[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane + 64 * r;
        zr[r] = i < nc ? zrow[i] : 0.0f;
        ur[r] = i < nf ? urow[i] : 0.0f;
    }
    nf_build_cdf_wave(wc + ray * nc + 1, nc - 2, cdf, xs, lane);
#pragma unroll
    for (int r = 0; r < 2; ++r) { const int i = lane + 64 * r; if (i < nc) A[i] = zr[r]; }
    for (int i = lane; i <= nc; i += 64) hist[i] = 0;
    nf_wave_sync();
    bool sorted = true;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane + 64 * r;
        if (i < n_bins) sorted = sorted && zr[r] <= A[i + 1];
    }
    // ---- inverse CDF (H:368-387): count = #{cdf <= u} by descent, then the interpolation of nf_invert_cdf.  The bin mid-points are formed
    // from the three depths around the bin, which the rank test below needs anyway (0.5 (z[i+1] + z[i]), T:116, as the general kernel) ------
    float v[2], a0[2], a1[2], a2[2];
    int below[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {                           // (lanes past nf run along on u = 0: no branches, the two descents interleave)
        const int j = lane + 64 * r;
        const float uu = ur[r];
        // the table is padded with +inf to 128 entries: no bound checks inside; u = +inf (counts the padding) is clamped after
        int pos = 0;
        if (n_bins >= 64) pos = cdf[63] <= uu ? 64 : 0;
#pragma unroll
        for (int st = 32; st > 0; st >>= 1) pos += cdf[pos + st - 1] <= uu ? st : 0;
        pos = pos < n_bins ? pos : n_bins;
        const int lo = pos;                                 // torch.searchsorted(cdf, u, right=True)
        const int bl = lo - 1 > 0 ? lo - 1 : 0;
        const int ab = lo < n_bins - 1 ? lo : n_bins - 1;   // bl + 1, or bl at either end of the table
        const float cb = cdf[bl], ca = cdf[ab];
        a0[r] = A[bl]; a1[r] = A[bl + 1]; a2[r] = A[bl + 2 < nc ? bl + 2 : nc - 1];
        const float bb = nf_mul(0.5f, nf_add(a1[r], a0[r]));
        const float ba = ab == bl ? bb : nf_mul(0.5f, nf_add(a2[r], a1[r]));
        float den = nf_sub(ca, cb);
        if (den < 1e-5f) den = 1.0f;
        const float t = nf_div(nf_sub(uu, cb), den);
        const float val = nf_add(bb, nf_mul(t, nf_sub(ba, bb)));
        v[r] = j < nf ? val : INFINITY;
        below[r] = bl;
        if (z_samples && j < nf) z_samples[ray * nf + j] = val;
    }
    // NaN samples (NaN weights / u, a diverged model) must survive as NaNs: the v_min-only register network below returns the non-NaN
    // operand of a compare-exchange, so such rows take the compare-swap path too (a total order with NaNs last, like torch.sort)
    const bool has_nan = __any((v[0] != v[0]) || (v[1] != v[1]));
    if (has_nan || !__all(sorted)) {                        // coarse depths not ascending (no caller on the hot path produces such a row):
        float* out = A;                                     // sort the concatenation like k_resample_merge does, wave-private (A | B = 256 floats;
                                                            // the depths are in place, the samples follow them directly)
        const int np2 = nf_next_pow2(nt);
        nf_wave_sync();
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int j = lane + 64 * r; if (j < nf) out[nc + j] = v[r]; }
        for (int i = nt + lane; i < np2; i += 64) out[i] = NF_SORT_PAD;
        nf_wave_sync();
        for (int k = 2; k <= np2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (np2 >> 1); t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), q = i | j;
                    const bool up = (i & k) == 0;
                    const float a = out[i], b = out[q];
                    if (nf_sort_gt(a, b) == up) { out[i] = b; out[q] = a; }
                }
                nf_wave_sync();
            }
        for (int i = lane; i < nt; i += 64) z_fine[ray * nt + i] = out[i];
        return;
    }
    // ---- rank of every sample among the depths, #{A <= s}: bin b = below -> b + 1 or b + 2, verified -------------------------------
    int rank[2];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int g = below[r] + 1;                         // 1 <= g <= nc - 1; a0, a1, a2 = A[g - 1], A[g], A[min(g + 1, nc - 1)]
        rank[r] = g + (a1[r] <= v[r] ? 1 : 0);
        ok = ok && (lane + 64 * r >= nf || (a0[r] <= v[r] && (g + 1 >= nc || !(a2[r] <= v[r]))));
    }
    if (!__all(ok)) {                                       // (degenerate spacing, NaN: the binary search of the round-4 kernel)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float b = v[r];
            int lo = 0, hi = nc;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (A[mid] <= b) lo = mid + 1; else hi = mid; }
            rank[r] = lo;
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (lane + 64 * r < nf) __hip_atomic_fetch_add(&hist[rank[r]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    // ---- the samples in ascending order -> B -------------------------------------------------------------------------------------
    {
        const float nx0 = __shfl_down(v[0], 1, 64), nx1 = __shfl_down(v[1], 1, 64), first1 = nf_readlane_f(v[1], 0);
        const bool asc = v[0] <= (lane < 63 ? nx0 : first1) && (lane == 63 || v[1] <= nx1);
        if (__all(asc)) {
#pragma unroll
            for (int r = 0; r < 2; ++r) B[lane + 64 * r] = v[r];
        } else {
            nf_sort128_regs(v);
#pragma unroll
            for (int r = 0; r < 2; ++r) B[nf_sort128_index(r, lane)] = v[r];
        }
    }
    // ---- slots of the depths: i + #{samples < z[i]} = i + #{samples of rank <= i} ---------------------------------------------------
    nf_wave_sync();                                         // (the flags alias the table as another type: keep the table's reads above this line)
    for (int i = lane; i < nt; i += 64) flags[i] = 0;
    nf_wave_sync();
    {
        const int h0 = lane < nc ? hist[lane] : 0;
        const int p0 = nf_scan64_i32(h0);
        if (lane < nc) flags[lane + p0] = 1;
        if (nc > 64) {
            const int carry = __builtin_amdgcn_readlane(p0, 63);
            const int h1 = lane + 64 < nc ? hist[lane + 64] : 0;
            const int p1 = nf_scan64_i32(h1) + carry;
            if (lane + 64 < nc) flags[lane + 64 + p1] = 1;
        }
    }
    nf_wave_sync();
    // ---- every output slot fetches its value: the c-th depth, or the (slot - c)-th sample, c = depth slots before it ------------------
    int before = 0;
    auto round = [&](int q) {
        const int s = 64 * q + lane;
        const bool dep = s < nt && flags[s] != 0;           // (flags: 256 entries, zeroed up to nt)
        const unsigned long long m = __ballot(dep);
        const int c = before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const float val = A[dep ? c : 128 + ((s - c) & 127)];
        if (s < nt) z_fine[ray * nt + s] = val;
        before += __popcll(m);
    };
    round(0); round(1);                                     // (independent up to the running count: their LDS reads overlap)
    if (nt > 128) { round(2); round(3); }
}

extern "C" int nf_resample_merge(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_row_stride,
                                 int64_t n_rays, int n_coarse, int n_fine, float* z_samples, float* z_fine,
                                 nf_stream_t stream) {
    if (n_rays == 0) return 0;                       // nothing to do (empty tensors have NULL data pointers)
    if (!z_coarse || !w_coarse || !u || !z_fine || n_rays < 0 || n_coarse < 3 || n_coarse - 1 > NF_MAX_BINS || n_fine <= 0 ||
        n_coarse + n_fine > NF_MAX_SORT)
        return NF_EINVAL;
    const int64_t grid = (n_rays + NF_RAYS_PER_BLOCK - 1) / NF_RAYS_PER_BLOCK;
    if (grid > 0x7fffffff) return NF_EINVAL;
    if (n_coarse <= NF_RS_MAXC && n_fine <= NF_RS_MAXF) {
#define NF_RS_LAUNCH(NC_, NF_)                                                                                                                  \
    hipLaunchKernelGGL((k_resample_merge_small<NC_, NF_>), dim3((unsigned)grid), dim3(256), 0, nf_s(stream), z_coarse, w_coarse, u, u_row_stride, \
                       n_rays, n_coarse, n_fine, z_samples, z_fine)
        if (n_coarse == 64 && n_fine == 128) NF_RS_LAUNCH(64, 128);
        else if (n_coarse == 64 && n_fine == 64) NF_RS_LAUNCH(64, 64);
        else NF_RS_LAUNCH(0, 0);
#undef NF_RS_LAUNCH
        NF_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL(k_resample_merge, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), z_coarse, w_coarse, u, u_row_stride,
                       n_rays, n_coarse, n_fine, z_samples, z_fine);
    NF_RETURN_LAUNCH();
}
