// Library-level entry points of the C ABI (include/picopose_hip.h).
#include "pp_common.h"

extern "C" {

const char* pp_strerror(int code) {
    switch (code) {
        case PP_OK: return "ok";
        case PP_EINVAL: return "invalid argument (null pointer, unsupported shape or mode)";
        case PP_EWORKSPACE: return "workspace too small or not 256-byte aligned";
        case PP_ELAUNCH: return "HIP launch or runtime call failed";
        default: return "unknown error code";
    }
}

int pp_version(void) { return 100; }

}  // extern "C"
