// K0: ray selection of a training iteration -- np.random.choice(H * W, size=n, replace=False, p=probs) of the reference trainer
// (train_transformed_rays.py:320-322, a 262144-way weighted draw on the host every iteration) on the device.
//
// Weighted sampling without replacement = the n smallest of the keys e_i / w_i with e_i ~ Exp(1) (Efraimidis & Spirakis 2006:
// successive draws proportional to the remaining weights, which is what np.random.choice(replace=False, p=...) does).  The
// caller supplies uniform numbers u_i in [0, 1) (torch.rand on the device: the draw stays under torch's generator and seeds);
// e_i = -log(1 - u_i).  The n smallest keys are found by an exact three-level radix select on the 32 key bits (12 + 12 + 8;
// positive floats order like their bit patterns): each level is one launch that histograms the digit of the keys that match the
// prefix found so far (LDS histogram per workgroup, merged with global atomics), and the LAST workgroup to finish scans the
// 4096 bins, extends the prefix and clears the histogram for the next level.  Then an ORDERED compaction in two launches: every
// workgroup counts, in its contiguous range of items, the keys below the threshold and the keys equal to it; the second launch
// turns the counts of the workgroups before it into its output offset and writes its items in index order -- every key below the
// threshold and, of the keys equal to it, the first `remaining` in index order (torch.rand has 24 random bits: on a 512 x 512 frame
// the threshold key is shared in ~1.5 % of the draws).  The batch comes out ascending without a sort and is a function of the
// seed alone (the order of the rays decides the summation order of the gradients downstream).  64 workgroups: the level passes
// merge their histograms with global atomics on the few bins the keys populate, and 256 workgroups queued 256-deep on each of them
// (25 us per level, more than the 2 MB the pass reads would take 8 times over).  HBM-bound in principle: 5 passes over 8 bytes per
// item (2 MB per pass for a 512 x 512 frame, L2-resident); ~30 us per draw.
#include "nf_common.h"

#define NF_CHOICE_BINS 4096
#define NF_CHOICE_MAX_BLOCKS 64        // level passes (global histogram atomics: few workgroups)
#define NF_CHOICE_MAX_CBLOCKS 1024     // compaction passes (no atomics: one 256-item tile per workgroup on a 512 x 512 frame)
struct NfChoiceState {            // workspace header (uint32 words): zeroed by the host-side memset before the first level
    unsigned done;                // workgroups that have flushed their histogram (reset by the scanning workgroup)
    unsigned prefix;              // key bits fixed so far
    unsigned remaining;           // how many keys of the current prefix class are still wanted
    unsigned emitted;             // (unused)
    unsigned ties;                // (unused)
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
    // four items per thread and trip: 64 workgroups leave one wave per SIMD, so the loads' latency has to be covered by the thread itself
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 1024) {
        unsigned k[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j = i + (int64_t)q * gridDim.x * 256;
            k[q] = j < n_items ? nf_choice_key(w, u, j) : 0xffffffffu;       // (past the end: matches no prefix, no level-0 digit below)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k[q] != 0xffffffffu && (LEVEL == 0 || (k[q] & HIGH_MASK) == prefix)) atomicAdd(&lh[(k[q] >> SHIFT) & (DIGITS - 1)], 1u);
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

// contiguous item range of a workgroup (the same in both compaction kernels)
__device__ __forceinline__ void nf_choice_range(int64_t n_items, int64_t& i0, int64_t& i1) {
    const int64_t span = ((n_items + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    i0 = (int64_t)blockIdx.x * span;
    i1 = i0 + span < n_items ? i0 + span : n_items;
}

// per workgroup: how many keys of its range lie below the threshold / equal it
__global__ void __launch_bounds__(256) k_choice_count(const float* __restrict__ w, const float* __restrict__ u, int64_t n_items,
                                                      const NfChoiceState* __restrict__ st, unsigned* __restrict__ counts) {
    __shared__ unsigned s_lt, s_eq;
    if (threadIdx.x == 0) s_lt = s_eq = 0u;
    __syncthreads();
    const unsigned thr = st->prefix;
    int64_t i0, i1;
    nf_choice_range(n_items, i0, i1);
    unsigned lt = 0, eq = 0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) {
        unsigned k[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) k[q] = i + 256 * q < i1 ? nf_choice_key(w, u, i + 256 * q) : 0x7f800000u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (k[q] == 0x7f800000u) continue;                      // weight 0 (or past the range)
            lt += k[q] < thr ? 1u : 0u;
            eq += k[q] == thr ? 1u : 0u;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o, 64); eq += __shfl_xor(eq, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_lt, lt); atomicAdd(&s_eq, eq); }
    __syncthreads();
    if (threadIdx.x == 0) { counts[2 * blockIdx.x] = s_lt; counts[2 * blockIdx.x + 1] = s_eq; }
}

// per workgroup: offsets from the counts of the workgroups before it, then its items in index order
__global__ void __launch_bounds__(256) k_choice_emit(const float* __restrict__ w, const float* __restrict__ u, int64_t n_items, int n_select,
                                                     const NfChoiceState* __restrict__ st, const unsigned* __restrict__ counts,
                                                     int64_t* __restrict__ idx_out) {
    __shared__ unsigned s_wave[2][4], s_base[2];
    const unsigned thr = st->prefix, ties_wanted = st->remaining;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) {                                         // keys below / equal to the threshold in the workgroups before this one
        unsigned lt = 0, eq = 0;
        for (int b = lane; b < (int)blockIdx.x; b += 64) { lt += counts[2 * b]; eq += counts[2 * b + 1]; }
        for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o, 64); eq += __shfl_xor(eq, o, 64); }
        if (lane == 0) { s_base[0] = lt; s_base[1] = eq; }
    }
    __syncthreads();
    unsigned lt_before = s_base[0], eq_before = s_base[1];
    int64_t i0, i1;
    nf_choice_range(n_items, i0, i1);
    for (int64_t t4 = i0; t4 < i1; t4 += 1024) {                    // keys of four 256-item tiles at a time (latency), tiles in order
      unsigned k4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
          const int64_t j = t4 + 256 * q + threadIdx.x;
          k4[q] = j < i1 ? nf_choice_key(w, u, j) : 0x7f800000u;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t t0 = t4 + 256 * q;
        if (t0 >= i1) break;                                        // (uniform)
        const int64_t i = t0 + threadIdx.x;
        const unsigned k = k4[q];
        const bool lt = k < thr && k != 0x7f800000u, eq = k == thr && k != 0x7f800000u;
        const unsigned long long m_lt = __ballot(lt), m_eq = __ballot(eq);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { s_wave[0][wave] = (unsigned)__popcll(m_lt); s_wave[1][wave] = (unsigned)__popcll(m_eq); }
        __syncthreads();
        unsigned lt_w = 0, eq_w = 0, lt_all = 0, eq_all = 0;
        for (int q = 0; q < 4; ++q) {
            if (q < wave) { lt_w += s_wave[0][q]; eq_w += s_wave[1][q]; }
            lt_all += s_wave[0][q];
            eq_all += s_wave[1][q];
        }
        const unsigned my_lt = lt_before + lt_w + (unsigned)__popcll(m_lt & below);       // taken keys below the threshold with a smaller index
        const unsigned my_eq = eq_before + eq_w + (unsigned)__popcll(m_eq & below);       // ties with a smaller index
        const unsigned ties_taken_before = my_eq < ties_wanted ? my_eq : ties_wanted;
        if (lt || (eq && my_eq < ties_wanted)) {
            const unsigned slot = my_lt + ties_taken_before;
            if (slot < (unsigned)n_select) idx_out[slot] = i;
        }
        lt_before += lt_all;
        eq_before += eq_all;
        __syncthreads();
      }
    }
}

extern "C" size_t nf_weighted_choice_workspace_bytes(void) {
    return sizeof(NfChoiceState) + NF_CHOICE_BINS * sizeof(unsigned) + 2 * NF_CHOICE_MAX_CBLOCKS * sizeof(unsigned);
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
    unsigned* counts = hist + NF_CHOICE_BINS;
    const int64_t want = (n_items + 255) / 256;
    const int grid = (int)(want < NF_CHOICE_MAX_BLOCKS ? want : NF_CHOICE_MAX_BLOCKS);
    hipLaunchKernelGGL(k_choice_level<0>, dim3(grid), dim3(256), 0, s, weights, u, n_items, n_select, st, hist);
    hipLaunchKernelGGL(k_choice_level<1>, dim3(grid), dim3(256), 0, s, weights, u, n_items, n_select, st, hist);
    hipLaunchKernelGGL(k_choice_level<2>, dim3(grid), dim3(256), 0, s, weights, u, n_items, n_select, st, hist);
    const int cgrid = (int)(want < NF_CHOICE_MAX_CBLOCKS ? want : NF_CHOICE_MAX_CBLOCKS);
    hipLaunchKernelGGL(k_choice_count, dim3(cgrid), dim3(256), 0, s, weights, u, n_items, st, counts);
    hipLaunchKernelGGL(k_choice_emit, dim3(cgrid), dim3(256), 0, s, weights, u, n_items, n_select, st, counts, idx_out);
    NF_RETURN_LAUNCH();
}
