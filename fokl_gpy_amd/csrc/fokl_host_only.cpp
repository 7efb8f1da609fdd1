// Host-only stand-ins for what the host translation units (sampler, thread pool, vector log, integrator) expect from
// fokl_hip.hip: the error plumbing and the version call.  Linked into the sanitizer builds of the host side
// (`make host-tsan host-asan`, tools/sanitize_host.sh) -- GPU sanitizers are not available on this pool, and the
// hand-rolled lock-free pipeline of fokl_hostpool.cpp is exactly the part that wants a race detector.  Device entry
// points are absent from those libraries: _capi.load() skips them when FOKL_HOST_ONLY_LIBRARY=1.
#include <cstdint>
#include <mutex>
#include <string>

#include "../../include/fokl_hip_internal.h"

static std::mutex g_err_mutex;
static std::string g_err;

void fokl_set_global_error(const std::string &msg)
{
    std::lock_guard<std::mutex> lock(g_err_mutex);
    g_err = msg;
}

extern "C" int fokl_version(void) { return 100; }

extern "C" __attribute__((visibility("hidden"))) int64_t fokl_dchain_dispatcher_cpu_ns() { return 0; }   // (no engine here)

extern "C" const char *fokl_last_error(const fokl_ctx *)
{
    std::lock_guard<std::mutex> lock(g_err_mutex);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

extern "C" int fokl_device_count(int *count)
{
    if (count) *count = 0;
    return FOKL_ERR_HIP;
}

// The device chain engine and page-locked memory do not exist in these builds: the native search (fokl_search.cpp) is
// then created without an engine and falls back to ordinary memory for its tapes.
extern "C" int fokl_host_alloc(size_t, void **out)
{
    if (out) *out = nullptr;
    return FOKL_ERR_HIP;
}
extern "C" int fokl_host_free(void *) { return FOKL_ERR_HIP; }
extern "C" int fokl_dchain_submit(fokl_dchain *, int, int, const double *, const double *, double, double, double, double,
                                  double, const double *, const int32_t *, const double *, const double *, const int32_t *,
                                  const int32_t *, int, int, int, int64_t *, const double **)
{
    return FOKL_ERR_HIP;
}
extern "C" int fokl_dchain_submit_rows(fokl_dchain *, int, int, const double *, const double *, double, double, double, double,
                                       double, double, double, const fokl_tape_row *, const double *, const double *,
                                       const int32_t *, const uint64_t *, int, int64_t *, const double **)
{
    return FOKL_ERR_HIP;
}
extern "C" int fokl_dchain_wait(fokl_dchain *, int64_t, double *) { return FOKL_ERR_HIP; }
extern "C" int fokl_dchain_fetch_w(fokl_dchain *, int64_t, double *) { return FOKL_ERR_HIP; }
extern "C" int fokl_dchain_release(fokl_dchain *, int64_t) { return FOKL_ERR_HIP; }
extern "C" int fokl_dchain_try_release(fokl_dchain *, int64_t) { return 1; }
extern "C" int fokl_dchain_flush(fokl_dchain *) { return FOKL_ERR_HIP; }
// ... nor does the device's eigen-solver: a search in these builds is never bound to one.
extern "C" void fokl_device_dgemm(char *, char *, int *, int *, int *, double *, double *, int *, double *, int *, double *, double *,
                                  int *)
{
}
extern "C" int fokl_device_dgemm_configure(int, void *, int, void **) { return FOKL_ERR_HIP; }
extern "C" int fokl_device_dgemm_stats(int64_t *, int64_t *, int64_t *) { return FOKL_ERR_HIP; }
extern "C" int fokl_dspectral_max_columns(void) { return 0; }
extern "C" int fokl_dspectral_submit(fokl_dspectral *, const double *, int, const int32_t *, int, int, int, int64_t *, double **)
{
    return FOKL_ERR_HIP;
}
extern "C" int fokl_dspectral_flush(fokl_dspectral *) { return FOKL_ERR_HIP; }
extern "C" int fokl_dspectral_poll(fokl_dspectral *, int64_t) { return -FOKL_ERR_HIP; }
extern "C" int fokl_dspectral_wait(fokl_dspectral *, int64_t) { return FOKL_ERR_HIP; }
extern "C" int fokl_dspectral_release(fokl_dspectral *, int64_t) { return FOKL_ERR_HIP; }
