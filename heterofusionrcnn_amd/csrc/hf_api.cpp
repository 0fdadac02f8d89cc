// hf_api.cpp -- library-level entry points of libhfops.so (version / error text).
#include "hf_common.h"

#include <map>
#include <mutex>
#include <utility>

namespace hf {
thread_local int g_last_hip_error = 0;

int ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return HF_OK;
    int dev = 0;
    const hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return hip_status(e0);
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[std::make_pair(dev, kernel)];
    if (bytes <= have) return HF_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e != hipSuccess) return hip_status(e);
    have = bytes;
    return HF_OK;
}
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
