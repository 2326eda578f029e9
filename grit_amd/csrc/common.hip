// ABI bookkeeping of libgrit_hip.so (see include/grit_hip.h).
#include <hip/hip_runtime.h>
#include "../../include/grit_hip.h"

extern "C" {

int grit_abi_version(void) { return GRIT_ABI_VERSION; }

const char* grit_status_string(int status) {
    switch (status) {
        case GRIT_OK: return "ok";
        case GRIT_ERR_BAD_ARG: return "bad argument (null pointer, non-positive or overflowing dimension)";
        case GRIT_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case GRIT_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown status";
    }
}

}  // extern "C"
