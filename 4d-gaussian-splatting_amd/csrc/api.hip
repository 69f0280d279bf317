// Library-level entry points of libfdgs.so.
#include "fdgs_common.h"

extern "C" const char* fdgs_strerror(int status) {
    switch (status) {
        case FDGS_OK: return "ok";
        case FDGS_EINVAL: return "invalid argument (null pointer, negative size or unsupported configuration)";
        case FDGS_ELAUNCH: return "kernel launch rejected by the HIP runtime";
        case FDGS_EWORKSPACE: return "workspace too small";
        case FDGS_ETOOLARGE: return "input larger than the entry point's built-in limit";
        default: return "unknown fdgs status";
    }
}

extern "C" int fdgs_abi_version(void) { return 25; }
