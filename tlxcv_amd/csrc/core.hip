// Library plumbing: status strings, device probing.  No kernels here.
#include "common.h"
#include <string.h>
#include <mutex>
#include <utility>
#include <vector>

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

// Per-DEVICE caches (one process may drive several GPUs from several host threads): the CU count persistent grids are
// sized by, and which (device, kernel) pairs already had their dynamic-LDS limit raised (hipFuncSetAttribute is per
// device).  Both sit behind one mutex; a launch pays an uncontended lock and a short scan.
static std::mutex g_dev_mutex;
static int g_dev_cus[64];
static std::vector<std::pair<int, const void*>> g_raised;

int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    if (dev < 64 && g_dev_cus[dev] > 0) return g_dev_cus[dev];
    hipDeviceProp_t p;
    int n = 0;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
    if (dev < 64) g_dev_cus[dev] = n;
    return n;
}

int raise_lds_limit(const void* fn, int bytes, const char* who) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    for (const auto& e : g_raised)
        if (e.first == dev && e.second == fn) return TLXMI_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %d bytes: %s", who, bytes, hipGetErrorString(e));
    g_raised.emplace_back(dev, fn);
    return TLXMI_OK;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "%s: HIP launch failed: %s", what, hipGetErrorString(e));
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
