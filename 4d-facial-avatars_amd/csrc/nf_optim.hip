// Adam step of the trainer (reference: torch.optim.Adam over [coarse model, fine model, latent codes], train_transformed_rays.py:193-199,
// stepped at TR:391-392) for ALL parameter tensors of a step in ONE launch.
//
// torch's own multi-tensor paths cost more than the arithmetic here: the fused optimizer runs two multi_tensor_apply kernels of ~55 us
// each over the 54 tensors of this model pair (one of them only to add 1 to 54 step counters), 1.1 % of a 10 ms training iteration for
// 28 MB of state traffic.  Here a workgroup handles 1024 consecutive elements of ONE tensor (tensor id from a block table in the
// kernel arguments: uniform, no search per element), 16 bytes per lane and access, parameters / gradients / both moments streamed once.
// HBM-bound: 28 bytes per element (p, g, m, v read; p, m, v written), ~8 us for the 2.3 M elements of the paper model pair.
//
// Arithmetic = torch's (torch/optim/adam.py, _multi_tensor_adam / fused_adam_utils.cuh, no weight decay, no amsgrad, minimise), in f32:
//   m <- m + (1 - beta1) (g - m);  v <- beta2 v + (1 - beta2) g g;  p <- p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with the bias corrections bc1 = 1 - beta1^t, bc2 = 1 - beta2^t formed on the host in double, as torch does.
#include "nf_common.h"

#define NF_ADAM_MAX_TENSORS 64
#define NF_ADAM_ELEMS_PER_BLOCK 1024
struct NfAdamArgs {
    float* p[NF_ADAM_MAX_TENSORS];
    const float* g[NF_ADAM_MAX_TENSORS];
    float* m[NF_ADAM_MAX_TENSORS];
    float* v[NF_ADAM_MAX_TENSORS];
    int numel[NF_ADAM_MAX_TENSORS];
    int blk[NF_ADAM_MAX_TENSORS + 1];          // first workgroup of tensor t
};

typedef float nf_f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_adam_step(NfAdamArgs a, int n_tensors, float w1, float beta2, float w2, float step_size,
                                                   float bc2_sqrt, float eps) {
    int t = 0;
    while (t + 1 < n_tensors && (int)blockIdx.x >= a.blk[t + 1]) ++t;          // uniform
    const int base = ((int)blockIdx.x - a.blk[t]) * NF_ADAM_ELEMS_PER_BLOCK + (int)threadIdx.x * 4;
    const int n = a.numel[t];
    if (base >= n) return;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const bool vec = base + 4 <= n && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (vec) {
        nf_f32x4 pp = *reinterpret_cast<nf_f32x4*>(p + base), gg = *reinterpret_cast<const nf_f32x4*>(g + base);
        nf_f32x4 mm = *reinterpret_cast<nf_f32x4*>(m + base), vv = *reinterpret_cast<nf_f32x4*>(v + base);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = pp[k], mk = mm[k], vk = vv[k];
            // sqrt(v) / sqrt(bc2) as a true division, like torch's _foreach_div_ by the scalar
            mk = mk + w1 * (gg[k] - mk);
            vk = beta2 * vk + w2 * gg[k] * gg[k];
            const float denom = sqrtf(vk) / bc2_sqrt + eps;
            pk = pk - step_size * (mk / denom);
            pp[k] = pk; mm[k] = mk; vv[k] = vk;
        }
        *reinterpret_cast<nf_f32x4*>(p + base) = pp;
        *reinterpret_cast<nf_f32x4*>(m + base) = mm;
        *reinterpret_cast<nf_f32x4*>(v + base) = vv;
    } else {
        for (int k = 0; k < 4 && base + k < n; ++k) {
            float pk = p[base + k], mk = m[base + k], vk = v[base + k];
            const float gk = g[base + k];
            mk = mk + w1 * (gk - mk);
            vk = beta2 * vk + w2 * gk * gk;
            const float denom = sqrtf(vk) / bc2_sqrt + eps;
            pk = pk - step_size * (mk / denom);
            p[base + k] = pk; m[base + k] = mk; v[base + k] = vk;
        }
    }
}

// params / grads / exp_avg / exp_avg_sq: host arrays of n_tensors device pointers (contiguous f32 tensors of numel[i] elements each);
// step = the 1-based step count of this update (the same for every tensor, as when torch steps them together).
extern "C" int nf_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                            const int64_t* numel, int n_tensors, float lr, float beta1, float beta2, float eps, int64_t step,
                            nf_stream_t stream) {
    if (n_tensors == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || n_tensors < 0 || step < 1) return NF_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
    for (int t0 = 0; t0 < n_tensors; t0 += NF_ADAM_MAX_TENSORS) {
        const int nt = n_tensors - t0 < NF_ADAM_MAX_TENSORS ? n_tensors - t0 : NF_ADAM_MAX_TENSORS;
        NfAdamArgs a;
        a.blk[0] = 0;
        for (int i = 0; i < NF_ADAM_MAX_TENSORS; ++i) {
            const bool on = i < nt;
            if (on && (!params[t0 + i] || !grads[t0 + i] || !exp_avg[t0 + i] || !exp_avg_sq[t0 + i] || numel[t0 + i] < 0 ||
                       numel[t0 + i] > 0x7fffffff))
                return NF_EINVAL;
            a.p[i] = on ? params[t0 + i] : nullptr;
            a.g[i] = on ? grads[t0 + i] : nullptr;
            a.m[i] = on ? exp_avg[t0 + i] : nullptr;
            a.v[i] = on ? exp_avg_sq[t0 + i] : nullptr;
            a.numel[i] = on ? (int)numel[t0 + i] : 0;
            a.blk[i + 1] = a.blk[i] + (a.numel[i] + NF_ADAM_ELEMS_PER_BLOCK - 1) / NF_ADAM_ELEMS_PER_BLOCK;
        }
        if (a.blk[nt] == 0) continue;
        hipLaunchKernelGGL(k_adam_step, dim3(a.blk[nt]), dim3(256), 0, nf_s(stream), a, nt, w1, beta2, w2, step_size, bc2_sqrt, eps);
    }
    NF_RETURN_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The trainer's loss (train_transformed_rays.py:355-387) and its gradients in two launches instead of the ~20 torch launches of
//   coarse = mse_loss(rgb_coarse, target); fine = mse_loss(rgb_fine, target); code = 0.0005 * torch.norm(latent);
//   loss = coarse + fine + 10 * code; loss.backward()
// (two element-wise kernels and a reduction per mse, the norm, the scalar arithmetic, the fills and element-wise kernels of their
// backward nodes: ~100 us of 4-5 us launches per iteration, 1-2.5 % of a training iteration of the split arithmetics).  The tensors
// are tiny (2048 x 3 colours, 32 latent values): one workgroup.
//   forward:  out[0] = loss, out[1] = coarse mse, out[2] = fine mse (0 without a fine map), out[3] = code_weight * ||latent||,
//             out[4] = coarse + fine, out[5] = -10 log10(max(out[4], 1e-20)) (the PSNR the trainer logs), out[6] = ||latent||
//   backward: d_rgb = ((2 / n) (rgb - target)) * go   (ATen's mse_loss_backward order),
//             d_latent = latent * ((go * code_scale * code_weight) / ||latent||), 0 where the norm is 0 (ATen's norm_backward)
// Sums are accumulated in double (deterministic: fixed assignment of elements to lanes, fixed reduction tree).
// ---------------------------------------------------------------------------------------------------------------------------------
#define NF_LOSS_THREADS 1024
__device__ __forceinline__ double nf_block_sum_f64(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long b = __double_as_longlong(v);
        const unsigned lo = __shfl_xor((unsigned)b, o, 64), hi = __shfl_xor((unsigned)(b >> 32), o, 64);
        v += __longlong_as_double(((unsigned long long)hi << 32) | lo);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < NF_LOSS_THREADS / 64; ++i) t += sh[i];
    return t;
}

__global__ void __launch_bounds__(NF_LOSS_THREADS) k_train_loss_fwd(const float* __restrict__ rgb_c, const float* __restrict__ rgb_f,
                                                                    const float* __restrict__ target, int64_t n,
                                                                    const float* __restrict__ latent, int n_latent, float code_weight,
                                                                    float code_scale, float* __restrict__ out) {
    __shared__ double sh[NF_LOSS_THREADS / 64];
    double sc = 0.0, sf = 0.0, sl = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += NF_LOSS_THREADS) {
        const float t = target[i];
        const float dc = rgb_c[i] - t;
        sc += (double)(dc * dc);
        if (rgb_f) { const float df = rgb_f[i] - t; sf += (double)(df * df); }
    }
    if (latent) for (int i = threadIdx.x; i < n_latent; i += NF_LOSS_THREADS) sl += (double)(latent[i] * latent[i]);
    sc = nf_block_sum_f64(sc, sh);
    sf = nf_block_sum_f64(sf, sh);
    sl = nf_block_sum_f64(sl, sh);
    if (threadIdx.x == 0) {
        const float coarse = (float)(sc / (double)n), fine = rgb_f ? (float)(sf / (double)n) : 0.0f;
        const float nrm = (float)sqrt(sl), code = nrm * code_weight;
        const float mse = rgb_f ? coarse + fine : coarse;
        out[0] = latent ? mse + code_scale * code : mse;
        out[1] = coarse; out[2] = fine; out[3] = code; out[4] = mse;
        out[5] = -10.0f * log10f(fmaxf(mse, 1e-20f));
        out[6] = nrm;
    }
}

__global__ void __launch_bounds__(256) k_train_loss_bwd(const float* __restrict__ rgb_c, const float* __restrict__ rgb_f,
                                                        const float* __restrict__ target, int64_t n, const float* __restrict__ latent,
                                                        int n_latent, float code_weight, float code_scale, const float* __restrict__ out,
                                                        const float* __restrict__ grad_out, float* __restrict__ d_rgb_c,
                                                        float* __restrict__ d_rgb_f, float* __restrict__ d_latent) {
    const float go = grad_out[0];
    const float norm2n = (float)(2.0 / (double)n);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float t = target[i];
        d_rgb_c[i] = (norm2n * (rgb_c[i] - t)) * go;
        if (rgb_f) d_rgb_f[i] = (norm2n * (rgb_f[i] - t)) * go;
    }
    if (latent && i < n_latent) {
        const float nrm = out[6];
        const float g = ((go * code_scale) * code_weight) / nrm;
        d_latent[i] = nrm == 0.0f ? 0.0f : latent[i] * g;
    }
}

extern "C" int nf_train_loss_fwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n_elems, const float* latent,
                                 int n_latent, float code_weight, float code_scale, float* out7, nf_stream_t stream) {
    if (!rgb_coarse || !target || !out7 || n_elems <= 0 || n_latent < 0 || (latent && n_latent == 0)) return NF_EINVAL;
    hipLaunchKernelGGL(k_train_loss_fwd, dim3(1), dim3(NF_LOSS_THREADS), 0, nf_s(stream), rgb_coarse, rgb_fine, target, n_elems, latent,
                       n_latent, code_weight, code_scale, out7);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_train_loss_bwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n_elems, const float* latent,
                                 int n_latent, float code_weight, float code_scale, const float* out7, const float* grad_out,
                                 float* d_rgb_coarse, float* d_rgb_fine, float* d_latent, nf_stream_t stream) {
    if (!rgb_coarse || !target || !out7 || !grad_out || !d_rgb_coarse || n_elems <= 0 || (rgb_fine && !d_rgb_fine) || (latent && !d_latent))
        return NF_EINVAL;
    const int64_t work = n_elems > n_latent ? n_elems : n_latent;
    const int64_t grid = (work + 255) / 256;
    if (grid > 0x7fffffff) return NF_EINVAL;
    hipLaunchKernelGGL(k_train_loss_bwd, dim3((unsigned)grid), dim3(256), 0, nf_s(stream), rgb_coarse, rgb_fine, target, n_elems, latent,
                       n_latent, code_weight, code_scale, out7, grad_out, d_rgb_coarse, d_rgb_fine, d_latent);
    NF_RETURN_LAUNCH();
}
