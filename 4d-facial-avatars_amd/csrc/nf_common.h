// Shared helpers for the gfx950 kernels of libnerface_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nerface_hip.h"
#include "nf_sincos.h"

#define NF_WAVE 64

// Launch check: kernels are asynchronous, so this only reports launch-time failures.
#define NF_RETURN_LAUNCH()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        return e__ == hipSuccess ? 0 : (int)e__;   \
    } while (0)

static inline hipStream_t nf_s(nf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units of the CURRENT device (cached per device index: a process may drive several GPUs).  Persistent grids and the
// weight-gradient slice plan are sized from it; without a device (host-only size queries in the CPU tests) it is gfx950's 256.
static inline int64_t nf_cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}


// IEEE single ops that must not be contracted into FMAs (bit parity with the reference's
// separate mul / add tensor ops).  The library is also built with -ffp-contract=off.
__device__ __forceinline__ float nf_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float nf_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float nf_sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float nf_div(float a, float b) { return __fdiv_rn(a, b); }
