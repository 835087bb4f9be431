// core.hip -- library-level entry points of libbxmi: error text, device
// selection, raw HBM staging and the tuning knobs.
#include <mutex>

#include <chrono>
#include <cstring>
#include "common.hpp"

namespace bxmi {

std::string &last_error()
{
    thread_local std::string e;
    return e;
}

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

const DeviceProps &device_props()
{
    // refreshed when the current device changes (one process normally owns one GPU)
    thread_local DeviceProps p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return p;
    if (p.device != dev) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            p.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            p.device = dev;
        }
    }
    return p;
}

int ivl_set_option(const char *key, int64_t value);
int ivl_option_count();
int ivl_option_at(int i, const char **key, int64_t *value);
int bits_set_option(const char *key, int64_t value);
int64_t bits_get_grid();

}  // namespace bxmi

using namespace bxmi;

extern "C" int bxmi_version(void) { return 100; }  // 0.1.0

extern "C" const char *bxmi_last_error(void) { return last_error().c_str(); }

extern "C" int bxmi_device_count(int *n)
{
    if (!n) return fail(BXMI_EINVAL, "bxmi_device_count: n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(BXMI_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return BXMI_OK;
}

extern "C" int bxmi_set_device(int device)
{
    BXMI_HIP(hipSetDevice(device));
    return BXMI_OK;
}

extern "C" int bxmi_get_device(int *device)
{
    if (!device) return fail(BXMI_EINVAL, "bxmi_get_device: device is NULL");
    BXMI_HIP(hipGetDevice(device));
    return BXMI_OK;
}

extern "C" int bxmi_device_info(int device, char *name, int name_len, int *compute_units, int64_t *hbm_bytes)
{
    hipDeviceProp_t prop;
    BXMI_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return BXMI_OK;
}

extern "C" int bxmi_mem_info(int64_t *free_bytes, int64_t *total_bytes)
{
    size_t f = 0, t = 0;
    BXMI_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return BXMI_OK;
}

extern "C" int bxmi_synchronize(void *stream)
{
    BXMI_HIP(hipStreamSynchronize(as_stream(stream)));
    return BXMI_OK;
}

extern "C" int bxmi_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return fail(BXMI_EINVAL, "bxmi_malloc: dptr is NULL");
    *dptr = nullptr;
    BXMI_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return BXMI_OK;
}

extern "C" int bxmi_free(void *dptr)
{
    if (dptr) BXMI_HIP(hipFree(dptr));
    return BXMI_OK;
}

extern "C" int bxmi_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes) BXMI_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return BXMI_OK;
}

extern "C" int bxmi_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes) BXMI_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return BXMI_OK;
}

extern "C" int bxmi_memset(void *dst_dev, int value, size_t bytes)
{
    if (bytes) BXMI_HIP(hipMemset(dst_dev, value, bytes));
    return BXMI_OK;
}

namespace bxmi {
int64_t g_opt_poll = 1;

int wait_for_host_flag(const unsigned long long *flag, unsigned long long seq, hipStream_t st)
{
    if (g_opt_poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0;; spins++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return BXMI_OK;
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
        }
    }
    BXMI_HIP(hipStreamSynchronize(st));
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return fail(BXMI_EHIP, "completion word of a one-call kernel never arrived");
    return BXMI_OK;
}
}  // namespace bxmi

extern "C" int bxmi_set_option(const char *key, int64_t value)
{
    if (!key) return fail(BXMI_EINVAL, "bxmi_set_option: key is NULL");
    if (!strcmp(key, "core.poll")) {
        bxmi::g_opt_poll = value != 0;
        return BXMI_OK;
    }
    if (ivl_set_option(key, value) || bits_set_option(key, value)) return BXMI_OK;
    return fail(BXMI_EINVAL, "bxmi_set_option: unknown key '%s'", key);
}

// every option in turn: i = 0, 1, ... until BXMI_EINVAL (the interval path's table, then bits.grid and core.poll)
extern "C" int bxmi_option_at(int i, const char **key, int64_t *value)
{
    if (!key || !value) return fail(BXMI_EINVAL, "bxmi_option_at: NULL output");
    const int n = ivl_option_count();
    if (i >= 0 && i < n) return ivl_option_at(i, key, value) ? BXMI_OK : fail(BXMI_EINVAL, "bxmi_option_at: no option %d", i);
    if (i == n) {
        *key = "bits.grid", *value = bits_get_grid();
        return BXMI_OK;
    }
    if (i == n + 1) {
        *key = "core.poll", *value = bxmi::g_opt_poll ? 1 : 0;
        return BXMI_OK;
    }
    return fail(BXMI_EINVAL, "bxmi_option_at: no option %d", i);
}

extern "C" int bxmi_get_option(const char *key, int64_t *value)
{
    if (!key || !value) return fail(BXMI_EINVAL, "bxmi_get_option: NULL argument");
    // (a local: an unknown key must leave *value alone; the loop is bounded by the table, so no failing call overwrites
    // the thread's error text before the real message is set)
    const int n = ivl_option_count() + 2;
    for (int i = 0; i < n; i++) {
        const char *k = nullptr;
        int64_t v = 0;
        if (bxmi_option_at(i, &k, &v) != BXMI_OK) break;
        if (!strcmp(k, key)) {
            *value = v;
            return BXMI_OK;
        }
    }
    return fail(BXMI_EINVAL, "bxmi_get_option: unknown key '%s'", key);
}
