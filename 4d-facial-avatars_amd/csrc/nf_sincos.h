// Branch-free sin / cos pair for the in-kernel positional encoding (H:195-239) -- one instruction stream for all 64 lanes, nothing the
// compiler has to keep lane masks for.  (OCML's sincosf carries divergent slow-path branches for huge arguments: in the fused MLP kernels
// they cost 38 spilled SGPRs and a block of code that cannot be scheduled under the MFMAs.)
//   k = rint(x * 2/pi);  r = x - k * pi/2 with pi/2 in three fp32 pieces (fused multiply-adds: every product is exact);
//   a second, tiny step (k2 in {-1, 0, 1}) because the fp32 product x * 2/pi is itself rounded -- k can be one off once |x| > 2^19;
//   minimax polynomials on |r| <= pi/4 (the cephes fp32 coefficients), quadrant by integer bit operations.
// Accuracy against the correctly rounded value (tests/test_host.py compiles THIS header for the host and sweeps it; IEEE operations
// only, so the device computes the same bits): |error| <= 1.0e-7 for |x| <= 2^20, <= 1.5e-7 up to 2^23 (an encoding argument of
// 2^9 * coordinate: |coordinate| <= 16384), degrading gracefully above (2e-6 at 2.7e8) -- SURVEY 8(d)(i) asks 2e-6 of the encoding.
// The stand-alone encoder (k_posenc) and the per-call folded columns keep sinf / cosf.
#pragma once
#if defined(__HIPCC__)
#define NF_SC_FN __device__ __forceinline__
#define NF_SC_FMA(a, b, c) __builtin_fmaf(a, b, c)
#define NF_SC_RINT(a) __builtin_rintf(a)
#define NF_SC_BITS(f) __builtin_bit_cast(unsigned, f)
#define NF_SC_FLOAT(u) __builtin_bit_cast(float, u)
#else
#include <math.h>
#include <string.h>
#define NF_SC_FN static inline
#define NF_SC_FMA(a, b, c) fmaf(a, b, c)
#define NF_SC_RINT(a) rintf(a)
static inline unsigned nf_sc_bits_(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float nf_sc_float_(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define NF_SC_BITS(f) nf_sc_bits_(f)
#define NF_SC_FLOAT(u) nf_sc_float_(u)
#endif

NF_SC_FN void nf_sincos(float x, float* s, float* c) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float P1 = -1.57079625129699707031e+00f, P2 = -7.54978941586159635335e-08f, P3 = -5.39030252995776476554e-15f;   // -(pi/2), three pieces
    const float k = NF_SC_RINT(x * TWO_OVER_PI);
    float r = NF_SC_FMA(k, P1, x);
    r = NF_SC_FMA(k, P2, r);
    r = NF_SC_FMA(k, P3, r);
    const float k2 = NF_SC_RINT(r * TWO_OVER_PI);
    r = NF_SC_FMA(k2, P1, r);
    r = NF_SC_FMA(k2, P2, r);
    const float r2 = r * r;
    float ps = NF_SC_FMA(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = NF_SC_FMA(r2, ps, -1.6666654611e-1f);
    const float sn = NF_SC_FMA(r * r2, ps, r);
    float pc = NF_SC_FMA(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = NF_SC_FMA(r2, pc, 4.166664568298827e-2f);
    const float cs = NF_SC_FMA(r2 * r2, pc, NF_SC_FMA(r2, -0.5f, 1.0f));
    const int q = (int)k + (int)k2;                                   // quadrant: sin -> {s, c, -s, -c}[q & 3], cos -> {c, -s, -c, s}[q & 3]
    const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    *s = NF_SC_FLOAT(NF_SC_BITS(a) ^ ((unsigned)(q & 2) << 30));
    *c = NF_SC_FLOAT(NF_SC_BITS(b) ^ ((unsigned)((q + 1) & 2) << 30));
}
