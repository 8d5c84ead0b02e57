// Weight packing shared by the model families and arithmetics: a gather table (one 32-bit code per packed element:
// tensor id << 24 | element offset, 0xFF000000 = zero) is built on the host once per device, uploaded, and a one-pass
// kernel gathers the live nn.Parameter storages into the MFMA fragment image -- f32 as is, or split into (hi, lo) bf16
// 1-KiB block pairs for the split-bf16 kernels.  Runs again whenever a parameter's version counter moves (nerf/ops.py).
#pragma once
#include <vector>
#include <mutex>
#include "nf_common.h"

template <int N>
struct NfPackPtrs { const float* p[N]; };

// TAG only keeps the instantiations of different translation units apart
template <int N, int TAG>
__global__ void __launch_bounds__(256) k_pack_f32(NfPackPtrs<N> ptrs, const uint32_t* __restrict__ table, float* __restrict__ packed, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t code = table[i], id = code >> 24;
        packed[i] = id == 0xFFu ? 0.0f : ptrs.p[id][code & 0xFFFFFFu];
    }
}

// entry e = (pair e >> 9, position e & 511): hi goes to block 2 * pair, lo = bf16(w - hi) to block 2 * pair + 1
template <int N, int TAG>
__global__ void __launch_bounds__(256) k_pack_split_bf16(NfPackPtrs<N> ptrs, const uint32_t* __restrict__ table, __bf16* __restrict__ stream,
                                                         int n_entries) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const uint32_t code = table[e], id = code >> 24;
        const float w = id == 0xFFu ? 0.0f : ptrs.p[id][code & 0xFFFFFFu];
        const __bf16 hi = (__bf16)w;
        const __bf16 lo = (__bf16)(w - (float)hi);
        const int pair = e >> 9, within = e & 511;
        stream[(size_t)(2 * pair) * 512 + within] = hi;
        stream[(size_t)(2 * pair + 1) * 512 + within] = lo;
    }
}

// per-device copy of a gather table, built and uploaded on first use
struct NfPackTable {
    std::mutex mutex;
    uint32_t* dev[64] = {nullptr};
    template <class Build>
    int get(Build build, uint32_t** out) {
        int d = 0;
        hipError_t e = hipGetDevice(&d);
        if (e != hipSuccess) return (int)e;
        if (d < 0 || d >= 64) return NF_EINVAL;
        std::lock_guard<std::mutex> lock(mutex);
        if (!dev[d]) {
            std::vector<uint32_t> host;
            build(host);
            uint32_t* p = nullptr;
            e = hipMalloc(&p, host.size() * sizeof(uint32_t));
            if (e != hipSuccess) return (int)e;
            e = hipMemcpy(p, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
            dev[d] = p;
        }
        *out = dev[d];
        return 0;
    }
};

template <int N>
static inline int nf_pack_ptrs(const float* const* params, NfPackPtrs<N>& ptrs) {
    if (!params) return NF_EINVAL;
    for (int i = 0; i < N; ++i) {
        if (!params[i]) return NF_EINVAL;
        ptrs.p[i] = params[i];
    }
    return 0;
}

template <int N, int TAG, class Build>
static inline int nf_pack_f32(NfPackTable& cache, Build build, const float* const* params, float* packed, int n, nf_stream_t stream) {
    NfPackPtrs<N> ptrs;
    if (!packed || nf_pack_ptrs<N>(params, ptrs)) return NF_EINVAL;
    uint32_t* table = nullptr;
    const int rc = cache.get(build, &table);
    if (rc) return rc;
    hipLaunchKernelGGL((k_pack_f32<N, TAG>), dim3(1024), dim3(256), 0, nf_s(stream), ptrs, table, packed, n);
    NF_RETURN_LAUNCH();
}

template <int N, int TAG, class Build>
static inline int nf_pack_split_bf16(NfPackTable& cache, Build build, const float* const* params, void* stream_out, int n_entries,
                                     nf_stream_t stream) {
    NfPackPtrs<N> ptrs;
    if (!stream_out || nf_pack_ptrs<N>(params, ptrs)) return NF_EINVAL;
    uint32_t* table = nullptr;
    const int rc = cache.get(build, &table);
    if (rc) return rc;
    hipLaunchKernelGGL((k_pack_split_bf16<N, TAG>), dim3(1024), dim3(256), 0, nf_s(stream), ptrs, table, reinterpret_cast<__bf16*>(stream_out),
                       n_entries);
    NF_RETURN_LAUNCH();
}

// ---- fp16 split streams (nf_mlp_f16.hip): x = hi + lo in fp16 keeps 22 significand bits, but fp16's exponent range is
// narrow, so every layer's weights are multiplied by a power of two 2^e chosen from the layer's largest |w| such that the
// scaled maximum lies in [2^13, 2^14): the `lo` parts (<= 2^-11 of the value) then stay normal fp16 numbers for every
// weight down to 2^-17 of the layer's largest.  Tail of the stream buffer (NF_F16_TAIL_BYTES behind the blocks):
// [NL] bias scales s_W * act_scale | [NL] inverse weight scales 1 / s_W | [NL] scratch (|w| maxima as uint bits) | ... |
// dword NF_F16_FLAG_WORD: range-guard flag set by the forward kernel.
#define NF_F16_TAIL_BYTES 256
#define NF_F16_FLAG_WORD 48                  // dword index in the tail: sticky "non-finite output" flag of the forward kernel
template <int NL> struct NfLayerPairs { int off[NL + 1]; };

// A block owns a CONTIGUOUS range of entries (entries are ordered by layer, so a block sees one or two layers), reduces into
// eleven LDS words and issues at most NL global atomics: a million global atomics on eleven addresses took 5.5 ms.
template <int N, int TAG, int NL>
__global__ void __launch_bounds__(256) k_stream_absmax(NfPackPtrs<N> ptrs, const uint32_t* __restrict__ table, int n_entries,
                                                       NfLayerPairs<NL> lp, unsigned* __restrict__ amax_bits) {
    __shared__ unsigned smax[NL];
    if (threadIdx.x < NL) smax[threadIdx.x] = 0u;
    __syncthreads();
    const int span = (n_entries + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * span, e1 = e0 + span < n_entries ? e0 + span : n_entries;
    // a thread walks its entries in ascending order, so their layer index never decreases: the running maximum of the current layer stays
    // in a register and goes to LDS when the layer changes and at the end (one or two LDS atomics per thread instead of one per entry)
    int cur_l = 0;
    unsigned cur = 0u;
    auto take = [&](int e, float w) {
        const int pair = e >> 9;
        int l = cur_l;
        while (l + 1 < NL && pair >= lp.off[l + 1]) ++l;
        if (l != cur_l) {
            if (cur != 0u) atomicMax(&smax[cur_l], cur);
            cur_l = l;
            cur = 0u;
        }
        w = fabsf(w);
        if (w > 0.0f && w < INFINITY && __float_as_uint(w) > cur) cur = __float_as_uint(w);
    };
    // four entries per trip: the four table reads, then the four gathered weights, are in flight together (two dependent L2 round trips per
    // trip instead of per entry); 256 workgroups: the kernel's cost was its ~2000 global atomics on eleven addresses (5 ns each), not the reads
    int e = e0 + threadIdx.x;
    for (; e + 3 * (int)blockDim.x < e1; e += 4 * blockDim.x) {
        uint32_t code[4];
        float w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) code[q] = table[e + q * blockDim.x];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (code[q] >> 24) == 0xFFu ? 0.0f : ptrs.p[code[q] >> 24][code[q] & 0xFFFFFFu];
#pragma unroll
        for (int q = 0; q < 4; ++q) take(e + q * blockDim.x, w[q]);
    }
    for (; e < e1; e += blockDim.x) {
        const uint32_t code = table[e], id = code >> 24;
        if (id != 0xFFu) take(e, ptrs.p[id][code & 0xFFFFFFu]);
    }
    if (cur != 0u) atomicMax(&smax[cur_l], cur);
    __syncthreads();
    if (threadIdx.x < NL && smax[threadIdx.x] != 0u) atomicMax(amax_bits + threadIdx.x, smax[threadIdx.x]);
}

__device__ __forceinline__ float nf_f16_layer_scale(unsigned amax_bits) {
    const float amax = __uint_as_float(amax_bits);
    if (!(amax > 0.0f)) return 1.0f;
    int k;
    (void)frexpf(amax, &k);                         // amax = m 2^k, m in [0.5, 1)
    int e = 14 - k;
    e = e < -24 ? -24 : (e > 40 ? 40 : e);
    return ldexpf(1.0f, e);
}

template <int N, int TAG, int NL>
__global__ void __launch_bounds__(256) k_pack_split_f16(NfPackPtrs<N> ptrs, const uint32_t* __restrict__ table, _Float16* __restrict__ stream,
                                                        int n_entries, NfLayerPairs<NL> lp, float* __restrict__ tail, float act_scale) {
    const unsigned* amax_bits = reinterpret_cast<const unsigned*>(tail + 2 * NL);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const uint32_t code = table[e], id = code >> 24;
        const int pair = e >> 9, within = e & 511;
        int l = 0;
        while (l + 1 < NL && pair >= lp.off[l + 1]) ++l;
        const float sc = nf_f16_layer_scale(amax_bits[l]);
        const float w = id == 0xFFu ? 0.0f : ptrs.p[id][code & 0xFFFFFFu] * sc;
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        stream[(size_t)(2 * pair) * 512 + within] = hi;
        stream[(size_t)(2 * pair + 1) * 512 + within] = lo;
        if (e < NL) {
            const float s = nf_f16_layer_scale(amax_bits[e]);
            tail[e] = s * act_scale;
            tail[NL + e] = 1.0f / s;
        }
    }
}

template <int N, int TAG, int NL, class Build>
static inline int nf_pack_split_f16(NfPackTable& cache, Build build, const float* const* params, void* stream_out, int n_entries,
                                    const NfLayerPairs<NL>& lp, float act_scale, nf_stream_t stream) {
    NfPackPtrs<N> ptrs;
    if (!stream_out || nf_pack_ptrs<N>(params, ptrs)) return NF_EINVAL;
    uint32_t* table = nullptr;
    const int rc = cache.get(build, &table);
    if (rc) return rc;
    float* tail = reinterpret_cast<float*>(reinterpret_cast<char*>(stream_out) + (size_t)n_entries * 4);   // 2 blocks x 2 bytes per entry
    hipError_t e = hipMemsetAsync(tail, 0, NF_F16_TAIL_BYTES, nf_s(stream));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_stream_absmax<N, TAG, NL>), dim3(256), dim3(256), 0, nf_s(stream), ptrs, table, n_entries, lp,
                       reinterpret_cast<unsigned*>(tail + 2 * NL));
    hipLaunchKernelGGL((k_pack_split_f16<N, TAG, NL>), dim3(1024), dim3(256), 0, nf_s(stream), ptrs, table,
                       reinterpret_cast<_Float16*>(stream_out), n_entries, lp, tail, act_scale);
    NF_RETURN_LAUNCH();
}

