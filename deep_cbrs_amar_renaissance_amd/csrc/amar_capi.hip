// Library-level entry points of include/amar_hip.h (version, error reporting).
#include "amar_common.h"

thread_local int amar_tls_hip_error = 0;

extern "C" {

int amar_version(void) { return 100; }   // 0.1.0

const char *amar_error_string(int code) {
    switch (code) {
    case AMAR_OK: return "ok";
    case AMAR_EINVAL: return "invalid argument (null pointer, negative size, leading dimension or alignment)";
    case AMAR_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case AMAR_ELAUNCH: return "HIP launch error (see amar_last_hip_error)";
    default: return "unknown error";
    }
}

int amar_last_hip_error(void) { return amar_tls_hip_error; }

}  // extern "C"
