// hf_api.cpp -- library-level entry points of libhfops.so (version / error text).
#include "hf_common.h"

namespace hf {
thread_local int g_last_hip_error = 0;
}

HF_API const char *hf_version(void) { return "hfops-mi355x 0.1.0 (gfx950)"; }

HF_API const char *hf_strerror(int status)
{
    switch (status) {
        case HF_OK: return "ok";
        case HF_EINVAL: return "invalid argument (shape / attribute check of the reference op failed)";
        case HF_EHIP: return "HIP runtime error (see hf_last_hip_error)";
        case HF_EWORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}

HF_API int hf_last_hip_error(void) { return hf::g_last_hip_error; }
