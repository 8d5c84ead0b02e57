// K0: ray selection of a training iteration -- np.random.choice(H * W, size=n, replace=False, p=probs) of the reference trainer
// (train_transformed_rays.py:320-322, a 262144-way weighted draw on the host every iteration) on the device.
//
// Weighted sampling without replacement = the n smallest of the keys e_i / w_i with e_i ~ Exp(1) (Efraimidis & Spirakis 2006:
// successive draws proportional to the remaining weights, which is what np.random.choice(replace=False, p=...) does).  The
// caller supplies uniform numbers u_i in [0, 1) (torch.rand on the device: the draw stays under torch's generator and seeds);
// e_i = -log(1 - u_i).  The n smallest keys are found by an exact three-level radix select on the 32 key bits (12 + 12 + 8;
// positive floats order like their bit patterns): each level is one launch that histograms the digit of the keys that match the
// prefix found so far (LDS histogram per workgroup, merged with global atomics), and the LAST workgroup to finish scans the
// 4096 bins, extends the prefix and clears the histogram for the next level.  A fourth launch emits the indices of every key below
// the threshold (slots come from an atomic counter) and collects the items whose key EQUALS it; a one-workgroup kernel then takes
// as many of those ties as are still missing -- the ones with the LOWEST indices, not the first to arrive (torch.rand has 24
// random bits: on a 512 x 512 frame the threshold key is shared in ~1.5 % of the draws) -- and a one-workgroup bitonic sort puts the
// batch in ascending order, so that a seed reproduces the batch element for element (the order of the rays decides the
// summation order of the gradients downstream).  HBM-bound: 4 passes over 8 bytes per item (2 MB per pass for a
// 512 x 512 frame, L2-resident).
#include "nf_common.h"

#define NF_CHOICE_BINS 4096
#define NF_CHOICE_MAX_TIES 1024   // tie candidates kept for the deterministic tie-break (more than that: the first to arrive)
struct NfChoiceState {            // workspace header (uint32 words): zeroed by the host-side memset before the first level
    unsigned done;                // workgroups that have flushed their histogram (reset by the scanning workgroup)
    unsigned prefix;              // key bits fixed so far
    unsigned remaining;           // how many keys of the current prefix class are still wanted
    unsigned emitted;             // select pass: slots handed out to keys below the threshold
    unsigned ties;                // select pass: keys equal to the threshold seen so far
    unsigned short_of;            // != 0: fewer than n items with a positive weight (np.random.choice raises ValueError there)
    unsigned pad[2];
};

__device__ __forceinline__ unsigned nf_choice_key(const float* __restrict__ w, const float* __restrict__ u, int64_t i) {
    const float wi = w[i];
    if (!(wi > 0.0f)) return 0x7f800000u;                         // never chosen (+inf); NaN weights too
    const float e = -logf(1.0f - u[i]);                            // u in [0, 1) -> e in [0, 16.7]
    const float k = e / wi;
    const unsigned b = __float_as_uint(k);
    return b > 0x7f7fffffu ? 0x7f7fffffu : b;                      // finite: an overflowing quotient still beats weight 0
}

// LEVEL 0: bits 31..20, LEVEL 1: bits 19..8, LEVEL 2: bits 7..0
template <int LEVEL>
__global__ void __launch_bounds__(256) k_choice_level(const float* __restrict__ w, const float* __restrict__ u, int64_t n_items, int n_select,
                                                      NfChoiceState* __restrict__ st, unsigned* __restrict__ hist) {
    constexpr int SHIFT = LEVEL == 0 ? 20 : (LEVEL == 1 ? 8 : 0);
    constexpr unsigned DIGITS = LEVEL == 2 ? 256u : 4096u;
    constexpr unsigned HIGH_MASK = LEVEL == 0 ? 0u : (LEVEL == 1 ? 0xfff00000u : 0xffffff00u);
    __shared__ unsigned lh[NF_CHOICE_BINS];
    __shared__ unsigned s_last;
    for (int b = threadIdx.x; b < (int)DIGITS; b += 256) lh[b] = 0;
    __syncthreads();
    const unsigned prefix = LEVEL == 0 ? 0u : st->prefix;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
        const unsigned k = nf_choice_key(w, u, i);
        if (LEVEL == 0 || (k & HIGH_MASK) == prefix) atomicAdd(&lh[(k >> SHIFT) & (DIGITS - 1)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < (int)DIGITS; b += 256)
        if (lh[b]) atomicAdd(&hist[b], lh[b]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&st->done, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // ---- last workgroup: find the digit in which the cumulative count reaches `remaining` -------------------------------------
    __threadfence();
    const unsigned want = LEVEL == 0 ? (unsigned)n_select : st->remaining;
    constexpr int PER = (int)DIGITS / 256;                          // 16 bins per thread (1 at the last level)
    unsigned mine[PER], sum = 0;
    for (int q = 0; q < PER; ++q) {
        mine[q] = __atomic_load_n(&hist[threadIdx.x * PER + q], __ATOMIC_RELAXED);
        sum += mine[q];
    }
    lh[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {                                         // 256-entry exclusive scan, serial: a few hundred cycles, once per level
        unsigned run = 0;
        for (int t = 0; t < 256; ++t) { const unsigned v = lh[t]; lh[t] = run; run += v; }
        lh[256] = run;
    }
    __syncthreads();
    const unsigned below = lh[threadIdx.x], total = lh[256];
    if (LEVEL == 0 && threadIdx.x == 0) {
        // keys of weight-0 items sit in the +inf digit (0x7f8): they count in `total` but may not be chosen
        const unsigned inf_items = __atomic_load_n(&hist[0x7f8], __ATOMIC_RELAXED);
        if (total - inf_items < (unsigned)n_select) st->short_of = 1u;
    }
    if (below < want && want <= below + sum) {                      // exactly one thread: the crossing lies in its bins
        unsigned run = below;
        for (int q = 0; q < PER; ++q) {
            if (want <= run + mine[q]) {
                st->prefix = prefix | ((unsigned)(threadIdx.x * PER + q) << SHIFT);
                st->remaining = want - run;                         // still wanted among the keys with this digit
                break;
            }
            run += mine[q];
        }
    }
    if (total < want && threadIdx.x == 0) {                         // fewer candidates than wanted: take them all (flagged above)
        st->prefix = prefix | ((DIGITS - 1) << SHIFT);
        st->remaining = 0xffffffffu;
    }
    __syncthreads();
    for (int q = 0; q < PER; ++q) hist[threadIdx.x * PER + q] = 0;  // clean for the next level
    if (threadIdx.x == 0) st->done = 0;
}

__global__ void __launch_bounds__(256) k_choice_select(const float* __restrict__ w, const float* __restrict__ u, int64_t n_items, int n_select,
                                                       NfChoiceState* __restrict__ st, int64_t* __restrict__ tie_buf, int64_t* __restrict__ idx_out) {
    const unsigned thr = st->prefix;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
        const unsigned k = nf_choice_key(w, u, i);
        if (k == 0x7f800000u) continue;                             // weight 0
        if (k < thr) {
            const unsigned slot = atomicAdd(&st->emitted, 1u);
            if (slot < (unsigned)n_select) idx_out[slot] = i;
        } else if (k == thr) {
            const unsigned t = atomicAdd(&st->ties, 1u);
            if (t < NF_CHOICE_MAX_TIES) tie_buf[t] = i;
        }
    }
}

// the `remaining` lowest-indexed ties fill the slots behind the keys below the threshold (rank by counting: the list is short)
__global__ void __launch_bounds__(256) k_choice_ties(int n_select, const NfChoiceState* __restrict__ st, const int64_t* __restrict__ tie_buf,
                                                     int64_t* __restrict__ idx_out) {
    const unsigned n_t = st->ties < NF_CHOICE_MAX_TIES ? st->ties : NF_CHOICE_MAX_TIES;
    const unsigned wanted = st->remaining, base = st->emitted;
    for (unsigned a = threadIdx.x; a < n_t; a += 256) {
        const int64_t mine = tie_buf[a];
        unsigned rank = 0;
        for (unsigned b = 0; b < n_t; ++b) rank += tie_buf[b] < mine ? 1u : 0u;
        if (rank < wanted && base + rank < (unsigned)n_select) idx_out[base + rank] = mine;
    }
}

// ascending order, unsigned compare (the -1 fillers of a short draw go last); n <= NF_CHOICE_SORT_MAX, one workgroup
#define NF_CHOICE_SORT_MAX 8192
__global__ void __launch_bounds__(1024) k_choice_sort(int64_t* __restrict__ idx, int n) {
    __shared__ uint64_t v[NF_CHOICE_SORT_MAX];
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = threadIdx.x; i < m; i += 1024) v[i] = i < n ? (uint64_t)idx[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const uint64_t a = v[i], b = v[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { v[i] = b; v[p] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 1024) idx[i] = (int64_t)v[i];
}

extern "C" size_t nf_weighted_choice_workspace_bytes(void) {
    return sizeof(NfChoiceState) + NF_CHOICE_BINS * sizeof(unsigned) + NF_CHOICE_MAX_TIES * sizeof(int64_t);
}

extern "C" int nf_weighted_choice(const float* weights, const float* u, int64_t n_items, int n_select, int64_t* idx_out, void* workspace,
                                  size_t workspace_bytes, nf_stream_t stream) {
    if (n_select == 0) return 0;
    if (!weights || !u || !idx_out || !workspace || n_items <= 0 || n_select < 0 || n_select > n_items ||
        workspace_bytes < nf_weighted_choice_workspace_bytes())
        return NF_EINVAL;
    hipStream_t s = nf_s(stream);
    hipError_t e = hipMemsetAsync(workspace, 0, nf_weighted_choice_workspace_bytes(), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(idx_out, 0xff, (size_t)n_select * sizeof(int64_t), s);       // -1 where fewer than n items have weight > 0
    if (e != hipSuccess) return (int)e;
    NfChoiceState* st = reinterpret_cast<NfChoiceState*>(workspace);
    unsigned* hist = reinterpret_cast<unsigned*>(st + 1);
    int64_t* tie_buf = reinterpret_cast<int64_t*>(hist + NF_CHOICE_BINS);
    const int64_t want = (n_items + 255) / 256;
    const int grid = (int)(want < 256 ? want : 256);
    hipLaunchKernelGGL(k_choice_level<0>, dim3(grid), dim3(256), 0, s, weights, u, n_items, n_select, st, hist);
    hipLaunchKernelGGL(k_choice_level<1>, dim3(grid), dim3(256), 0, s, weights, u, n_items, n_select, st, hist);
    hipLaunchKernelGGL(k_choice_level<2>, dim3(grid), dim3(256), 0, s, weights, u, n_items, n_select, st, hist);
    hipLaunchKernelGGL(k_choice_select, dim3(grid), dim3(256), 0, s, weights, u, n_items, n_select, st, tie_buf, idx_out);
    hipLaunchKernelGGL(k_choice_ties, dim3(1), dim3(256), 0, s, n_select, st, tie_buf, idx_out);
    if (n_select <= NF_CHOICE_SORT_MAX)                              // larger draws stay in slot order (the Python wrapper sorts them)
        hipLaunchKernelGGL(k_choice_sort, dim3(1), dim3(1024), 0, s, idx_out, n_select);
    NF_RETURN_LAUNCH();
}
