// abi.hip -- version / error-string entry points of libupp_hip.so (see include/upp_hip.h).
#include "common.h"

extern "C" int upp_abi_version(void) { return UPP_ABI_VERSION; }

extern "C" const char *upp_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case UPP_E_BADARG: return "upp_hip: null pointer or non-positive size";
        case UPP_E_RANGE: return "upp_hip: size outside the supported range (see include/upp_hip.h)";
        case UPP_E_KGTN: return "upp_hip: knn k exceeds the number of reference points";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "upp_hip: unknown error code";
}
