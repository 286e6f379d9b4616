// abi.hip -- version / error-string entry points of libupp_hip.so (see include/upp_hip.h).
#include "common.h"

#include <atomic>

extern "C" int upp_abi_version(void) { return UPP_ABI_VERSION; }

// ---- options: defaults are the product's (each one an A/B-measured choice, see include/upp_hip.h)
static std::atomic<int> g_options[UPP_OPT_COUNT] = {{1}, {2}, {1}, {1}};
static_assert(UPP_OPT_SB_TUNED == 0 && UPP_OPT_SB_XCD2D == 1 && UPP_OPT_STORE_WT == 2 && UPP_OPT_EMBED_SPLIT_BF16 == 3 && UPP_OPT_COUNT == 4, "defaults above are in key order");

int upp_option(int key) { return g_options[key].load(std::memory_order_relaxed); }

extern "C" int upp_get_option(int key) {
    if (key < 0 || key >= UPP_OPT_COUNT) return UPP_E_BADARG;
    return upp_option(key);
}

extern "C" int upp_set_option(int key, int value) {
    if (key < 0 || key >= UPP_OPT_COUNT) return UPP_E_BADARG;
    const bool ok = key == UPP_OPT_SB_XCD2D ? (value == 0 || value == 2 || value == 4) : (value == 0 || value == 1);
    if (!ok) return UPP_E_RANGE;
    g_options[key].store(value, std::memory_order_relaxed);
    return 0;
}

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
