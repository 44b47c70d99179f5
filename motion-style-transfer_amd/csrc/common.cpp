// Error reporting shared by every entry point of libynet_hip.so (see include/ynet_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void ynet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ynet_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ynet_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

extern "C" const char* ynet_last_error(void) { return g_err; }
extern "C" int ynet_abi_version(void) { return 1; }
