// ToMe baseline kernels (K7/K8) -- placeholder entry points until the MFMA matcher lands.
#include "sttm_kernels.h"

extern "C" {

size_t sttm_tome_workspace_bytes(int, int, int) { return 0; }

int sttm_tome_match(const void*, int, int, int, int, void*, size_t, float*, int32_t*, void*) {
    return STTM_ERR_UNSUPPORTED;
}

int sttm_tome_merge(const void*, const float*, const int64_t*, int, int, int, const int64_t*, int, const int32_t*,
                    void*, size_t, void*, float*, int64_t*, void*) {
    return STTM_ERR_UNSUPPORTED;
}

}  // extern "C"
