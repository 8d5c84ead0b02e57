// Library-level entry points of libnerface_hip.so.
#include "nf_common.h"

extern "C" int nf_abi_version(void) { return 5; }

extern "C" const char* nf_error_string(int code) {
    if (code == 0) return "ok";
    if (code == NF_EINVAL) return "nerface_hip: invalid argument";
    return hipGetErrorString((hipError_t)code);
}

extern "C" const char* nf_build_info(void) { return "libnerface_hip gfx950 " __VERSION__ " " __DATE__; }

