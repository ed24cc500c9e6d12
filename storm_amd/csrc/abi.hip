// Error channel, version and device info of libstorm_hip.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "common.h"

namespace storm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace storm

extern "C" const char* storm_last_error(void) { return storm::g_err; }
extern "C" int storm_abi_version(void) { return 1; }
extern "C" int storm_device_info(char* name, int name_len, int* n_cu, size_t* hbm_bytes) {
    int dev = 0;
    STORM_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    STORM_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return STORM_OK;
}
