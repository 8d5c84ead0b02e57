// Store-pattern probe: what THIS box's memory system does with the write streams of the training kernels, with nothing else
// in the way (no MFMA, no LDS traffic, no reads).  A sequential 1 GiB fill reads ~6.9 TB/s on every box of the pool and does
// not see the condition that made the driver's boxes of rounds 3 and 4 take 2.9x the cycles in the store-heavy kernels; this
// probe drives the kernels' REAL pattern: 256 persistent workgroups x 4 waves (one workgroup per CU), wave w takes the 32-point
// tiles w, w + 1024, ...; per tile and per section (= one layer's saved plane, 256 MiB apart at 262144 points) it issues 32
// store instructions of 1 KiB:
//   stream : lane-linear 1 KiB blocks, 32 KiB contiguous per (tile, section)        (k_paper_mlp_fwd_*_train, nfb_stream_target)
//   rows   : 8 rows x 128 B per instruction, row pitch 1 KiB                        (k_paper_mlp_bwd_chain_*: dZ, nfb_save_target)
//   seq    : the same grid writing one contiguous span grid-stride                  (control: what a fill does)
// each with non-temporal (the kernels' policy) and default stores, into a freshly hipMalloc'ed span (first touch reported
// separately) of 9 sections = 2.4 GB.  Prints one JSON object.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/store_bw tools/micro/store_bw.hip && tools/micro/store_bw [points]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { STREAM = 0, ROWS = 1, SEQ = 2 };

template <int PATTERN, int NT>
__global__ void __launch_bounds__(256, 1) k_store(char* out, long n_tiles, int n_sec, long sec_bytes) {
    __shared__ char pad[100 * 1024];                                  // one workgroup per CU, like the MLP kernels
    if (threadIdx.x == 1000) pad[0] = 1;
    const int lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (long t = gw; t < n_tiles; t += nw)
        for (int s = 0; s < n_sec; ++s) {
            char* base = PATTERN == SEQ ? out + ((long)s * n_tiles + t) * 32768 : out + (long)s * sec_bytes + t * 32768;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                char* p;
                if (PATTERN == ROWS)   // instruction (tile f = k / 4, row group g = k % 4): rows 8 g + lane / 8, 16-byte chunk lane % 8 of feature tile f
                    p = base + (long)(8 * (k & 3) + (lane >> 3)) * 1024 + (k >> 2) * 128 + (lane & 7) * 16;
                else
                    p = base + k * 1024 + lane * 16;
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = v;
            }
        }
}

typedef void (*kern_t)(char*, long, int, long);
static double run(kern_t k, char* out, long n_tiles, int n_sec, long sec_bytes, int reps, double* first) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double best = 0.0;
    const double gb = (double)n_tiles * n_sec * 32768 / 1e9;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, n_tiles, n_sec, sec_bytes);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double g = gb / ms * 1e3;
        if (r == 0 && first) *first = g;
        if (g > best) best = g;
    }
    return best;
}

int main(int argc, char** argv) {
    const long points = argc > 1 ? atol(argv[1]) : 262144;
    const long n_tiles = points / 32;
    const int n_sec = 9;
    const long sec_bytes = points * 1024;                              // 256 features x 4 bytes per point
    char* out;
    if (hipMalloc(&out, (size_t)sec_bytes * n_sec) != hipSuccess) { printf("{\"error\": \"alloc failed\"}\n"); return 1; }
    double first = 0.0;
    const double stream_nt = run(k_store<STREAM, 1>, out, n_tiles, n_sec, sec_bytes, 4, &first);
    const double stream_wb = run(k_store<STREAM, 0>, out, n_tiles, n_sec, sec_bytes, 3, nullptr);
    const double rows_nt = run(k_store<ROWS, 1>, out, n_tiles, n_sec, sec_bytes, 3, nullptr);
    const double rows_wb = run(k_store<ROWS, 0>, out, n_tiles, n_sec, sec_bytes, 3, nullptr);
    const double seq_nt = run(k_store<SEQ, 1>, out, n_tiles, n_sec, sec_bytes, 3, nullptr);
    const double seq_wb = run(k_store<SEQ, 0>, out, n_tiles, n_sec, sec_bytes, 3, nullptr);
    printf("{\"points\": %ld, \"span_gb\": %.3f, \"first_touch_stream_nt_gbs\": %.1f, \"stream_nt_gbs\": %.1f, \"stream_gbs\": %.1f, "
           "\"rows_nt_gbs\": %.1f, \"rows_gbs\": %.1f, \"seq_nt_gbs\": %.1f, \"seq_gbs\": %.1f}\n",
           points, (double)sec_bytes * n_sec / 1e9, first, stream_nt, stream_wb, rows_nt, rows_wb, seq_nt, seq_wb);
    hipFree(out);
    return 0;
}
