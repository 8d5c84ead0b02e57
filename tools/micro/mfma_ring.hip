// Synthetic model of the split-bf16 MLP kernels' inner structure, to compare wave shapes before rewriting the real kernel:
//   MODE 0: 4 waves x 32 points, v_mfma_f32_32x32x16_bf16, 8 output tiles of 32 (the shipped structure)
//   MODE 1: 8 waves x 16 points, v_mfma_f32_16x16x32_bf16, 16 output tiles of 16 (two waves per SIMD)
// Per stage (32 K values x 256 outputs x (hi, lo) = 32 KiB of weights): LDS-DMA into a 4-deep ring, every wave reads the
// whole stage from LDS and issues 3 MFMAs per (tile, k-step); one barrier per stage.  Prints achieved bf16 TFLOP/s.
// STORES = 1 models the TRAINING kernels: after every 8 stages (one 256-wide layer) each wave writes its f32 accumulator
// tiles to global memory with 16-byte stores (1 KiB per wave instruction): 32 store instructions per layer for a 32-point
// wave, 16 for a 16-point wave -- the question being whether a second wave per SIMD hides the store-issue stalls.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define STAGE_BYTES 32768
#define NBUF 4

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// EPI = 1 models the layer epilogue of the real kernels: after every 8 stages each accumulator element goes through ReLU and a
// (hi, lo) bf16 split (5 VALU instructions per element: 128 elements per lane for a 32-point wave, 64 for a 16-point wave), and
// the next layer's B operands depend on the result -- the question being whether the second wave of a SIMD hides that VALU time
// behind its partner's MFMAs (the waves cross a barrier every stage, so their epilogues tend to coincide).
__device__ __forceinline__ void epi_elt(float v, unsigned& fh, unsigned& fl) {
    const float x = __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff());
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    fh ^= (unsigned)__builtin_bit_cast(unsigned short, h);
    fl ^= (unsigned)__builtin_bit_cast(unsigned short, l);
}
template <int MODE, int STORES, int EPI = 0, int SKEW = 0>
__global__ void __launch_bounds__(MODE ? 512 : 256, 1) k_ring(const char* __restrict__ w, int n_stages, float* __restrict__ out,
                                                              float* __restrict__ sink) {
    constexpr int NW = MODE ? 8 : 4;
    constexpr int PIECES = 32 / NW;                 // 1-KiB DMA pieces per wave and stage
    __shared__ __attribute__((aligned(16))) char lds[NBUF * STAGE_BYTES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* g = w + lane * 16;
    auto issue = [&](int st, int buf) {
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int b = wave + NW * q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((size_t)(st % 64) * 32 + b) * 1024),
                                             (__attribute__((address_space(3))) void*)(lds + buf * STAGE_BYTES + b * 1024), 16, 0, 0);
        }
    };
    bf16x8 bh[2], bl[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[0][j] = (__bf16)(0.001f * lane); bl[0][j] = (__bf16)1e-5f; bh[1][j] = (__bf16)0.5f; bl[1][j] = (__bf16)2e-5f; }
    issue(0, 0); issue(1, 1);
    if (SKEW) wait_vm<PIECES>();
    else { issue(2, 2); wait_vm<2 * PIECES>(); }
    __builtin_amdgcn_s_barrier();
    if constexpr (MODE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int st = 0; st < n_stages; ++st) {
            const char* base = lds + (st % NBUF) * STAGE_BYTES + lane * 16;
            issue(st + 3, (st + 3) % NBUF);
#pragma unroll
            for (int u = 0; u < 2; ++u) {                      // two k-steps of 16
                bf16x8 ah[8], al[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    ah[t] = *reinterpret_cast<const bf16x8*>(base + ((u * 8 + t) * 2 + 0) * 1024);
                    al[t] = *reinterpret_cast<const bf16x8*>(base + ((u * 8 + t) * 2 + 1) * 1024);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k == 0 ? al[t] : ah[t], k == 1 ? bl[u] : bh[u], acc[t], 0, 0, 0);
            }
            if (EPI && (st & 7) == 7) {                        // a layer ends: epilogue over 8 tiles x 16 elements per lane
                unsigned fh = 0, fl = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { epi_elt(acc[t][r], fh, fl); acc[t][r] = 0.f; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {                   // next layer's operands depend on the epilogue
                    bh[0][j] = __builtin_bit_cast(__bf16, (unsigned short)((fh >> (j & 1)) & 0x3f80u));
                    bl[0][j] = __builtin_bit_cast(__bf16, (unsigned short)((fl >> (j & 1)) & 0x3f80u));
                }
            }
            if (STORES && (st & 7) == 7) {                     // a layer ends: 8 tiles x 4 stores of 16 B per lane
                // STORES 1: row-major [point][256 features] (a wave instruction touches 64 lines, 16 B each);
                // STORES 2: wave-linear blocks (a wave instruction writes 1 KiB contiguous = 8 full lines)
                // STORES 3: row-major again, but a wave instruction covers 8 points x one full 128-byte line each (lane = point%8 x 16-byte
                //           chunk; the address pattern a store through an LDS transpose would have -- data content is irrelevant here)
                float* dst = STORES == 1 ? sink + ((size_t)(blockIdx.x * 4 + wave) * 32 + (lane & 31)) * 256 + 4 * (lane >> 5)
                           : STORES == 2 ? sink + (size_t)(blockIdx.x * 4 + wave) * 32 * 256 + 4 * lane
                                         : sink + ((size_t)(blockIdx.x * 4 + wave) * 32 + (lane >> 3)) * 256 + 4 * (lane & 7);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(dst + (STORES == 1 ? 32 * t + 8 * q : STORES == 2 ? 256 * (4 * t + q) : 32 * t + 8 * 256 * q)) =
                            (f32x4){acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                wait_vm<2 * PIECES + 32>();
            } else
            wait_vm<2 * PIECES>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // SKEW = 1: waves 4..7 (the second wave of every SIMD) run ONE STAGE behind waves 0..3, so that a wave's epilogue overlaps
        // its partner's MFMAs.  The barrier counts global steps; at step t the leaders consume stage t, the laggers stage t - 1, and
        // everybody issues its DMA pieces of stage t + 2 (ring of 4: t - 1, t, t + 1 landing, t + 2 in flight).
        const int lag = (SKEW && wave >= 4) ? 1 : 0;
        constexpr int LA = SKEW ? 2 : 3;
        for (int step = 0; step < n_stages + (SKEW ? 1 : 0); ++step) {
            const int st = step - lag;                         // this wave's stage
            if (step < n_stages) issue(step + LA, (step + LA) % NBUF);
            if (st >= 0 && st < n_stages) {
            const char* base = lds + (st % NBUF) * STAGE_BYTES + lane * 16;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                      // one k-step of 32, in two half-sets of 8 tiles (register pressure)
                bf16x8 ah[8], al[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    ah[t] = *reinterpret_cast<const bf16x8*>(base + ((q * 8 + t) * 2 + 0) * 1024);
                    al[t] = *reinterpret_cast<const bf16x8*>(base + ((q * 8 + t) * 2 + 1) * 1024);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        acc[q * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k == 0 ? al[t] : ah[t], k == 1 ? bl[0] : bh[0], acc[q * 8 + t], 0, 0, 0);
            }
            if (EPI && (st & 7) == 7) {                        // a layer ends: epilogue over 16 tiles x 4 elements per lane
                unsigned fh = 0, fl = 0;
#pragma unroll
                for (int t = 0; t < 16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { epi_elt(acc[t][r], fh, fl); acc[t][r] = 0.f; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bh[0][j] = __builtin_bit_cast(__bf16, (unsigned short)((fh >> (j & 1)) & 0x3f80u));
                    bl[0][j] = __builtin_bit_cast(__bf16, (unsigned short)((fl >> (j & 1)) & 0x3f80u));
                }
            }
            if (STORES && (st & 7) == 7) {                     // a layer ends: 16 tiles x 1 store of 16 B per lane
                float* dst = sink + ((size_t)(blockIdx.x * 8 + wave) * 16 + (lane & 15)) * 256 + 4 * (lane >> 4);
#pragma unroll
                for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(dst + 16 * t) = acc[t];
            }
            }
            if (STORES) wait_vm<0>(); else if (SKEW) wait_vm<PIECES>(); else wait_vm<2 * PIECES>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

int main(int argc, char** argv) {
    const int n_stages = argc > 1 ? atoi(argv[1]) : 4000, grid = argc > 2 ? atoi(argv[2]) : 256;
    char* w; float* out;
    hipMalloc(&w, 64 * STAGE_BYTES + 65536); hipMemset(w, 0, 64 * STAGE_BYTES + 65536);
    hipMalloc(&out, grid * 512 * sizeof(float));
    float* sink; hipMalloc(&sink, (size_t)grid * 128 * 256 * sizeof(float));       // one 128-point x 256-feature tile per workgroup
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode)                       // EPI: epilogue VALU work after every layer, no stores
        for (int rep = 0; rep < 3; ++rep) {                    // mode 2: 16-point waves, second wave of a SIMD one stage behind; mode 3: same, no epilogue
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((k_ring<0, 0, 1>), dim3(grid), dim3(256), 0, 0, w, n_stages, out, sink);
            else if (mode == 1) hipLaunchKernelGGL((k_ring<1, 0, 1>), dim3(grid), dim3(512), 0, 0, w, n_stages, out, sink);
            else if (mode == 2) hipLaunchKernelGGL((k_ring<1, 0, 1, 1>), dim3(grid), dim3(512), 0, 0, w, n_stages, out, sink);
            else hipLaunchKernelGGL((k_ring<1, 0, 0, 1>), dim3(grid), dim3(512), 0, 0, w, n_stages, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * n_stages * 128.0 * 256 * 32 * 2 * 3;
            printf("epilogue mode %d rep %d: %.3f ms  %.1f TFLOP/s issued bf16 (%.2f of 2500)  err=%d\n", mode, rep, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500, (int)hipGetLastError());
        }
    for (int stores = 0; stores < 1; ++stores)
    for (int mode = 0; mode < (stores >= 2 ? 1 : 2); ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0 && !stores) hipLaunchKernelGGL((k_ring<0, 0>), dim3(grid), dim3(256), 0, 0, w, n_stages, out, sink);
            else if (mode == 0 && stores == 1) hipLaunchKernelGGL((k_ring<0, 1>), dim3(grid), dim3(256), 0, 0, w, n_stages, out, sink);
            else if (mode == 0 && stores == 2) hipLaunchKernelGGL((k_ring<0, 2>), dim3(grid), dim3(256), 0, 0, w, n_stages, out, sink);
            else if (mode == 0) hipLaunchKernelGGL((k_ring<0, 3>), dim3(grid), dim3(256), 0, 0, w, n_stages, out, sink);
            else if (!stores) hipLaunchKernelGGL((k_ring<1, 0>), dim3(grid), dim3(512), 0, 0, w, n_stages, out, sink);
            else hipLaunchKernelGGL((k_ring<1, 1>), dim3(grid), dim3(512), 0, 0, w, n_stages, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per stage and workgroup: 128 points x 256 outputs x 32 K x 2 flop x 3 products
            const double flop = (double)grid * n_stages * 128.0 * 256 * 32 * 2 * 3;
            printf("stores %d mode %d rep %d: %.3f ms  %.1f TFLOP/s issued bf16 (%.2f of 2500)  err=%d\n", stores, mode, rep, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500, (int)hipGetLastError());
        }
    }
    return 0;
}
