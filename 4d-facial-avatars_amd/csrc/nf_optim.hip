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
