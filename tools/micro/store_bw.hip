// How fast can ONE CU (4 waves, one per SIMD -- the occupancy of the fused MLP kernels) push f32 tiles to HBM, and does the rate
// depend on how many CUs store at the same time?  Each wave writes `rounds` x [8 points][32 floats] full 128-byte lines per
// instruction (the pattern nfb_save_tiles produces) to fresh addresses.  Variants: plain / nontemporal stores.
//   hipcc --offload-arch=gfx950 -O2 -o store_bw tools/micro/store_bw.hip && ./store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void __launch_bounds__(256, 1) k_store(float* out, int rounds, size_t stride_block) {
    __shared__ char pad[100 * 1024];                                  // one workgroup per CU, like the MLP kernels
    if (threadIdx.x == 1000) pad[0] = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = out + (size_t)blockIdx.x * stride_block + (size_t)wave * (stride_block / 4);
    f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (int r = 0; r < rounds; ++r) {
        f32x4* p = reinterpret_cast<f32x4*>(base + (size_t)r * 256) + lane;   // 1 KiB per wave instruction, contiguous
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}
int main() {
    const size_t per_block = 64ull << 20;                             // 64 MiB per block: 16 MiB per wave
    float* out;
    if (hipMalloc(&out, per_block * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = (int)(per_block / 4 / 1024);
    for (int nt = 0; nt < 2; ++nt)
        for (int blocks : {256, 128, 64, 32, 8}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (nt) hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, out, rounds, per_block / 4);
                else hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, out, rounds, per_block / 4);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            const double gb = (double)blocks * per_block / 1e9;
            printf("%s stores, %3d CUs busy: %7.1f GB/s total, %6.1f GB/s per CU\n", nt ? "nontemporal" : "plain      ", blocks, gb / ms * 1e3, gb / ms * 1e3 / blocks);
        }
    return 0;
}
