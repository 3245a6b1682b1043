// Error channel + version/introspection entry points of the C ABI.
#include "ppf_common.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

extern "C" {

void ppf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* ppf_last_error(void) { return g_err; }

int ppf_abi_version(void) { return 1; }

// Device facts the host side uses for launch sizing and reporting.
int ppf_device_info(int* cu_count, int* clock_mhz, char* name, int name_len) {
    hipDeviceProp_t prop;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { ppf_set_error("ppf_device_info: %s", hipGetErrorString(e)); return (int)e; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (name && name_len > 0) snprintf(name, name_len, "%s", prop.gcnArchName);
    return 0;
}

}  // extern "C"
