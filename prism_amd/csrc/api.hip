// Error plumbing and the trivial entry points of the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace prism {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace prism

extern "C" const char *prism_last_error(void) { return prism::g_err; }
extern "C" int prism_abi_version(void) { return PRISM_ABI_VERSION; }

extern "C" int prism_device_info(int device, int *cu_count, char *arch_name, int arch_name_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        prism::set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
        return PRISM_ERR_HIP;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return PRISM_OK;
}

extern "C" int prism_sync_target(float *target_params, const float *params, int64_t n_params, prism_stream_t stream) {
    PRISM_CHECK_ARG(target_params && params && n_params > 0, "null/empty buffers");
    hipError_t e = hipMemcpyAsync(target_params, params, sizeof(float) * (size_t)n_params, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) {
        prism::set_error("hipMemcpyAsync: %s", hipGetErrorString(e));
        return PRISM_ERR_HIP;
    }
    return PRISM_OK;
}
