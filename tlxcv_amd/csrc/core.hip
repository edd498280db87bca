// Library plumbing: status strings, device probing.  No kernels here.
#include "common.h"
#include <string.h>

namespace tlxmi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// One process drives ONE device (the deployment model: one rank per GPU).  The per-kernel caches of this library —
// raised dynamic-LDS limits (hipFuncSetAttribute), the CU count persistent grids are sized by — are process-wide, so a
// second device in the same process is refused instead of silently running with the first device's settings.
static int g_bound_device = -1;

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "%s: HIP launch failed: %s", what, hipGetErrorString(e));
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) {
        if (g_bound_device < 0) g_bound_device = dev;
        else if (dev != g_bound_device)
            return fail(TLXMI_ERR_UNSUPPORTED, "%s: this process already runs libtlxmi on device %d; device %d needs its own process "
                        "(one rank per GPU)", what, g_bound_device, dev);
    }
    return TLXMI_OK;
}

}  // namespace tlxmi

extern "C" int tlxmi_version(void) { return TLXMI_VERSION; }

extern "C" const char* tlxmi_last_error(void) { return tlxmi::g_err; }

extern "C" int tlxmi_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return tlxmi::fail(TLXMI_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}
