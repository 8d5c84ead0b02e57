// Follow-up to mfma_ring.hip: can two 16-point waves per SIMD hide the layer epilogue if the per-stage workgroup BARRIER is
// replaced by counters in LDS (a wave signals "my DMA pieces of stage g landed" / "I am done reading stage g" and only waits for
// what it needs), so that the partner wave may run up to SKEW stages ahead?  gfx950 has no split arrive/wait barrier; this
// emulates one with ds_add / ds_read polling.
//   8 waves x 16 points, v_mfma_f32_16x16x32_bf16; stage = 16 KiB (8 output tiles x one k-step of 32 x (hi, lo)); ring of NBUF = 8
//   stages (128 KiB), LA stages of DMA in flight; epilogue model as in mfma_ring.hip after every 16 stages (one 256 x 256 layer).
//   MODE 0: full s_barrier per stage (reference point);  MODE 1: counters.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_flags tools/micro/mfma_flags.hip && ./mfma_flags 4000 1024
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define STAGE_BYTES 16384
#define NBUF 8
#define PIECES 2                                    // 16 KiB / 8 waves / 1 KiB

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void epi_elt(float v, unsigned& fh, unsigned& fl) {
    const float x = __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff());
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    fh ^= (unsigned)__builtin_bit_cast(unsigned short, h);
    fl ^= (unsigned)__builtin_bit_cast(unsigned short, l);
}

template <int MODE, int EPI, int LA>
__global__ void __launch_bounds__(512, 1) k_flags(const char* __restrict__ w, int n_stages, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char lds[NBUF * STAGE_BYTES];
    __shared__ unsigned landed[NBUF], freed[NBUF];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < NBUF) { landed[threadIdx.x] = 0; freed[threadIdx.x] = 0; }
    __syncthreads();
    const char* g = w + lane * 16;
    auto issue = [&](int st) {
        const int buf = st % NBUF;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int b = wave + 8 * q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((size_t)(st % 128) * 16 + b) * 1024),
                                             (__attribute__((address_space(3))) void*)(lds + buf * STAGE_BYTES + b * 1024), 16, 0, 0);
        }
    };
    auto signal = [&](unsigned* c) { if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto wait_ge = [&](unsigned* c, unsigned target) {
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    };
    bf16x8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[j] = (__bf16)(0.001f * lane); bl[j] = (__bf16)1e-5f; }
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < LA; ++s) issue(s);
    for (int st = 0; st < n_stages; ++st) {
        const int buf = st % NBUF;
        const unsigned gen = (unsigned)(st / NBUF);
        if (st + LA < n_stages) {
            if (MODE == 1 && st + LA >= NBUF) wait_ge(&freed[(st + LA) % NBUF], 8u * (unsigned)((st + LA) / NBUF));   // buffer free again?
            issue(st + LA);
            wait_vm<LA * PIECES>();                            // my pieces of stage st have landed
        } else
            wait_vm<0>();
        if (MODE == 1) {
            asm volatile("" ::: "memory");
            signal(&landed[buf]);
            wait_ge(&landed[buf], 8u * (gen + 1u));            // everybody's pieces of stage st
        } else
            __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* base = lds + buf * STAGE_BYTES + lane * 16;
        const int half = st & 1;                               // stage = one half of the output tiles of a k-step
        bf16x8 ah[8], al[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            ah[t] = *reinterpret_cast<const bf16x8*>(base + (t * 2 + 0) * 1024);
            al[t] = *reinterpret_cast<const bf16x8*>(base + (t * 2 + 1) * 1024);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (half) acc[8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k == 0 ? al[t] : ah[t], k == 1 ? bl : bh, acc[8 + t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k == 0 ? al[t] : ah[t], k == 1 ? bl : bh, acc[t], 0, 0, 0);
            }
        if (MODE == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // my reads of this buffer are done
            signal(&freed[buf]);
        } else if (NBUF - LA < 2) {
            __builtin_amdgcn_s_barrier();                      // (not needed for LA <= NBUF - 2: the next barrier covers the reuse)
        }
        if (EPI && (st & 15) == 15) {                          // a layer ends
            unsigned fh = 0, fl = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { epi_elt(acc[t][r], fh, fl); acc[t][r] = 0.f; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bh[j] = __builtin_bit_cast(__bf16, (unsigned short)((fh >> (j & 1)) & 0x3f80u));
                bl[j] = __builtin_bit_cast(__bf16, (unsigned short)((fl >> (j & 1)) & 0x3f80u));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int EPI, int LA>
static void run(const char* name, const char* w, int n_stages, int grid, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_flags<MODE, EPI, LA>), dim3(grid), dim3(512), 0, 0, w, n_stages, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)grid * n_stages * 128.0 * 128 * 32 * 2 * 3;     // per stage: 128 points x 128 outputs x 32 K, 3 products
    printf("%-44s %8.3f ms  %.2f of the bf16 peak  err=%d\n", name, best, flop / best / 1e9 / 2500, (int)hipGetLastError());
}

int main(int argc, char** argv) {
    const int n_stages = argc > 1 ? atoi(argv[1]) : 4000, grid = argc > 2 ? atoi(argv[2]) : 1024;
    char* w; float* out;
    hipMalloc(&w, 128 * STAGE_BYTES + 65536); hipMemset(w, 0, 128 * STAGE_BYTES + 65536);
    hipMalloc(&out, (size_t)grid * 512 * sizeof(float));
    run<0, 0, 4>("barrier,  no epilogue, 4 stages in flight", w, n_stages, grid, out);
    run<0, 1, 4>("barrier,  epilogue,    4 stages in flight", w, n_stages, grid, out);
    run<1, 0, 4>("counters, no epilogue, 4 in flight (skew <= 3)", w, n_stages, grid, out);
    run<1, 1, 4>("counters, epilogue,    4 in flight (skew <= 3)", w, n_stages, grid, out);
    run<1, 1, 3>("counters, epilogue,    3 in flight (skew <= 3)", w, n_stages, grid, out);
    run<1, 1, 2>("counters, epilogue,    2 in flight (skew <= 2)", w, n_stages, grid, out);
    return 0;
}
