// Library-level entry points of libnerface_hip.so.
#include "nf_common.h"

extern "C" int nf_abi_version(void) { return 1; }

extern "C" const char* nf_error_string(int code) {
    if (code == 0) return "ok";
    if (code == NF_EINVAL) return "nerface_hip: invalid argument";
    return hipGetErrorString((hipError_t)code);
}

extern "C" const char* nf_build_info(void) { return "libnerface_hip gfx950 " __VERSION__ " " __DATE__; }

// A/B switch of the round-3 training kernels (profiles/r03_*): non-zero selects the round-2 exact-f32 training forward,
// dX chain and weight-gradient kernel.  Not declared in the public header; bound by tools/ab_train_f32.py only.
int g_nf_legacy_train = 0;
extern "C" void nf_debug_legacy_train(int on) { g_nf_legacy_train = on; }
