// errors.cpp — thread-local error text + trivial queries of the C ABI.
#include <cstdarg>
#include <cstdio>
#include <string_view>
#include <hip/hip_runtime.h>
#include "../../include/qdiff_hip.h"

static thread_local char g_err[512] = "";

void qd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int qd_abi_version(void) { return QD_ABI_VERSION; }
extern "C" const char* qd_last_error(void) { return g_err; }
extern "C" int qd_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return std::string_view(prop.gcnArchName).substr(0, 6) == "gfx950" ? 1 : 0;
}
