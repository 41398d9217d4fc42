// Error plumbing, version and device checks of libvipant_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vipant_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vipant_last_error(void) { return g_err; }

extern "C" int32_t vipant_version(void) { return 100; }  // 0.1.0

extern "C" int32_t vipant_device_check(void) {
    int dev = 0;
    VIPANT_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    VIPANT_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vipant_set_error("device %d is %s; libvipant_hip.so is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return VIPANT_EHIP;
    }
    return VIPANT_OK;
}
