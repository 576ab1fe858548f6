// reart_amd/csrc/lib.hip -- library-level probes of libreart_hip.so (no device work).
#include "common.h"

extern "C" int reart_version(void) { return 200; }

extern "C" int reart_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *reart_status_string(int status) {
    switch (status) {
        case REART_OK: return "ok";
        case REART_ERR_INVALID_ARG: return "invalid argument";
        case REART_ERR_UNSUPPORTED: return "unsupported configuration";
        case REART_ERR_LAUNCH: return "kernel launch failed";
        case REART_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}
