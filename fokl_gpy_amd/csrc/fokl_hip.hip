// libfokl_hip.so -- context management and kernel launches behind the C ABI of include/fokl_hip.h.
// gfx950 only; no PyTorch, no BLAS libraries: device work is the hand-written kernels of fokl_kernels.hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fokl_hip_internal.h"
#include "fokl_kernels.hip.h"

using namespace fokl;

// ---------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------

static std::mutex g_err_mutex;
static std::string g_err;

void fokl_set_global_error(const std::string &msg)
{
    std::lock_guard<std::mutex> lock(g_err_mutex);
    g_err = msg;
}

struct TimingSlot {
    double ms = 0.0;
    int64_t launches = 0;
    double bytes = 0.0;
    double flops = 0.0;
    double ideal_ms = 0.0;   // sum over launches of max(bytes / FOKL_PEAK_HBM_BYTES_PER_S, flops / FOKL_PEAK_F64_FLOPS)
};

struct PendingEvent {
    int kernel_id;
    hipEvent_t start, stop;
};

struct fokl_ctx {
    int device = 0;
    int cus = 0;                 // compute units of the device
    hipStream_t stream = nullptr;
    std::string err;

    // dataset
    bool have_data = false;
    int64_t n = 0, ld = 0;
    int m = 0, kernel = 0, n_basis = 0, width = 0;
    double *d_x = nullptr;      // [m][ld]
    double *d_zero = nullptr;   // [ld] zeros: stands in for padding columns of Gram panels
    double *d_staged = nullptr; // [staged_n][staged_m] raw inputs waiting for fokl_upload_staged (fokl_stage_inputs)
    int64_t staged_n = 0;
    int staged_m = 0;
    double *d_phis = nullptr;
    std::vector<double> h_phis;  // host copy of the coefficient table (launch planning of the matrix-free K3)
    size_t phis_doubles = 0;

    // slots
    static constexpr int CHUNK_SLOTS = 16;
    std::vector<double *> chunks;
    std::vector<double *> slot_ptr;     // host copy of the table
    double **d_slot_ptr = nullptr;      // device table, capacity table_cap
    int table_cap = 0;

    // workspaces (grow only)
    char *d_args = nullptr;
    char *h_args = nullptr;             // pinned
    size_t args_cap = 0;
    double *d_slab = nullptr;
    size_t slab_doubles = 0;
    double *d_out = nullptr;
    double *h_out = nullptr;            // pinned
    size_t out_doubles = 0;
    hipEvent_t args_free = nullptr;     // recorded after the last H2D copy out of h_args
    bool resid_pending = false;         // fokl_bic_resid_launch issued, result not fetched yet
    // residual moments of a launched pass (a pair of their own: Gram blocks may be computed before they are fetched)
    double *d_rout = nullptr;
    double *h_rout = nullptr;           // pinned
    // fokl_gram_launch: a Gram block on its way while other launches (which use d_out / h_out) go on
    double *d_gout = nullptr;
    double *h_gout = nullptr;           // pinned
    size_t gout_doubles = 0;
    size_t gram_pending = 0;            // doubles of the block fokl_gram_fetch will return; 0 = none on its way
    hipEvent_t gram_done = nullptr;
    hipEvent_t resid_done = nullptr;     // behind the residual pass's result copy: its fetch waits for this, not for what was queued after it

    // timing
    bool timing = false;
    TimingSlot tslot[FOKL_K_COUNT];
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;

    // RCCL (lazily loaded)
    void *comm = nullptr;
    int rank = 0, world = 1;
    double *d_comm = nullptr;
    size_t comm_doubles = 0;
};

static int fail(fokl_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    fokl_set_global_error(msg);
    return code;
}

#define HIP_TRY(ctx, call)                                                                                     \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess)                                                                                 \
            return fail(ctx, FOKL_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));                \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// timing helpers: HIP events on the context's own stream
// ---------------------------------------------------------------------------------------------------------

static hipEvent_t take_event(fokl_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct TimedRegion {
    fokl_ctx *ctx;
    int id;
    hipEvent_t start = nullptr, stop = nullptr;
    TimedRegion(fokl_ctx *c, int kernel_id, double bytes, double flops) : ctx(c), id(kernel_id)
    {
        if (!ctx->timing) return;
        ctx->tslot[id].launches += 1;
        ctx->tslot[id].bytes += bytes;
        ctx->tslot[id].flops += flops;
        ctx->tslot[id].ideal_ms += 1e3 * std::max(bytes / FOKL_PEAK_HBM_BYTES_PER_S, flops / FOKL_PEAK_F64_FLOPS);
        start = take_event(ctx);
        stop = take_event(ctx);
        if (start && stop) (void)hipEventRecord(start, ctx->stream);
    }
    ~TimedRegion()
    {
        if (!ctx->timing || !start || !stop) return;
        (void)hipEventRecord(stop, ctx->stream);
        ctx->pending.push_back({id, start, stop});
    }
};

static void drain_events(fokl_ctx *ctx)
{
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess)
            ctx->tslot[p.kernel_id].ms += ms;
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->pending.clear();
}

// ---------------------------------------------------------------------------------------------------------
// workspace management
// ---------------------------------------------------------------------------------------------------------

static int ensure_args(fokl_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->args_cap) return FOKL_OK;
    size_t cap = std::max<size_t>(bytes * 2, 1 << 16);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_args) HIP_TRY(ctx, hipFree(ctx->d_args));
    if (ctx->h_args) HIP_TRY(ctx, hipHostFree(ctx->h_args));
    ctx->d_args = nullptr;
    ctx->h_args = nullptr;
    ctx->args_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_args, cap));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_args, cap, hipHostMallocDefault));
    ctx->args_cap = cap;
    return FOKL_OK;
}

static int ensure_slab(fokl_ctx *ctx, size_t doubles)
{
    if (doubles <= ctx->slab_doubles) return FOKL_OK;
    size_t cap = std::max<size_t>(doubles + doubles / 2, 1 << 16);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_slab) HIP_TRY(ctx, hipFree(ctx->d_slab));
    ctx->d_slab = nullptr;
    ctx->slab_doubles = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_slab, cap * sizeof(double)));
    ctx->slab_doubles = cap;
    return FOKL_OK;
}

static int ensure_out(fokl_ctx *ctx, size_t doubles)
{
    if (doubles <= ctx->out_doubles) return FOKL_OK;
    size_t cap = std::max<size_t>(doubles * 2, 1 << 12);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_out) HIP_TRY(ctx, hipFree(ctx->d_out));
    if (ctx->h_out) HIP_TRY(ctx, hipHostFree(ctx->h_out));
    ctx->d_out = nullptr;
    ctx->h_out = nullptr;
    ctx->out_doubles = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_out, cap * sizeof(double)));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_out, cap * sizeof(double), hipHostMallocDefault));
    ctx->out_doubles = cap;
    return FOKL_OK;
}

static int ensure_rout(fokl_ctx *ctx)
{
    if (ctx->d_rout) return FOKL_OK;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_rout, 2 * sizeof(double)));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_rout, 2 * sizeof(double), hipHostMallocDefault));
    return FOKL_OK;
}

static int ensure_gout(fokl_ctx *ctx, size_t doubles)
{
    if (doubles <= ctx->gout_doubles) return FOKL_OK;
    size_t cap = std::max<size_t>(doubles * 2, 1 << 12);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_gout) HIP_TRY(ctx, hipFree(ctx->d_gout));
    if (ctx->h_gout) HIP_TRY(ctx, hipHostFree(ctx->h_gout));
    ctx->d_gout = nullptr;
    ctx->h_gout = nullptr;
    ctx->gout_doubles = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_gout, cap * sizeof(double)));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_gout, cap * sizeof(double), hipHostMallocDefault));
    ctx->gout_doubles = cap;
    return FOKL_OK;
}

// Stage `bytes` of launch arguments: wait until the previous copy out of the pinned buffer has finished,
// let the caller fill it, then copy asynchronously.
static int begin_args(fokl_ctx *ctx, size_t bytes)
{
    int rc = ensure_args(ctx, bytes);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventSynchronize(ctx->args_free));
    return FOKL_OK;
}

static int push_args(fokl_ctx *ctx, size_t bytes)
{
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_args, ctx->h_args, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->args_free, ctx->stream));
    return FOKL_OK;
}

static void free_slots(fokl_ctx *ctx)
{
    for (double *c : ctx->chunks) (void)hipFree(c);
    ctx->chunks.clear();
    ctx->slot_ptr.clear();
}

static int check_slots(fokl_ctx *ctx, const int32_t *slots, int count, const char *who)
{
    if (!slots && count > 0) return fail(ctx, FOKL_ERR_ARG, std::string(who) + ": null slot list");
    const int cap = (int)ctx->slot_ptr.size();
    for (int i = 0; i < count; ++i)
        if (slots[i] < 0 || slots[i] >= cap)
            return fail(ctx, FOKL_ERR_ARG, std::string(who) + ": slot " + std::to_string(slots[i]) +
                                               " outside [0, " + std::to_string(cap) + ")");
    return FOKL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// library / context
// ---------------------------------------------------------------------------------------------------------

extern "C" int fokl_version(void) { return 100; }

extern "C" int fokl_device_count(int *count)
{
    if (!count) return fail(nullptr, FOKL_ERR_ARG, "fokl_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, FOKL_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return FOKL_OK;
}

extern "C" const char *fokl_last_error(const fokl_ctx *ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lock(g_err_mutex);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

extern "C" int fokl_ctx_create(int device, fokl_ctx **out)
{
    if (!out) return fail(nullptr, FOKL_ERR_ARG, "fokl_ctx_create: null output pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, FOKL_ERR_HIP, "fokl_ctx_create: no HIP device available (" +
                                               std::string(hipGetErrorString(e)) + ")");
    if (device < 0 || device >= n)
        return fail(nullptr, FOKL_ERR_ARG, "fokl_ctx_create: device index out of range");
    fokl_ctx *ctx = new fokl_ctx();
    ctx->device = device;
    // FOKL_SYNC=blocking: host threads that wait for the device sleep instead of spinning -- for many processes sharing one
    // GPU under a CPU quota (bench.py's throughput workers), where the spinning of one is CPU the others do not get.
    // Must come before the runtime creates the device's context; an error (context exists already) is not fatal.
    if (const char *mode = std::getenv("FOKL_SYNC"))
        if (std::strcmp(mode, "blocking") == 0) {
            (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
            (void)hipGetLastError();
        }
    HIP_TRY(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        std::string arch = prop.gcnArchName;
        delete ctx;
        return fail(nullptr, FOKL_ERR_HIP, "fokl_ctx_create: device is " + arch + ", this library is built for gfx950 only");
    }
    const int total_cus = prop.multiProcessorCount;
    ctx->cus = total_cus;
    const hipError_t made = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (made != hipSuccess ||
        hipEventCreateWithFlags(&ctx->args_free, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->gram_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->resid_done, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return fail(nullptr, FOKL_ERR_HIP, "fokl_ctx_create: cannot create stream / event");
    }
    (void)hipEventRecord(ctx->args_free, ctx->stream);
    *out = ctx;
    return FOKL_OK;
}

extern "C" int fokl_comm_destroy(fokl_ctx *ctx);

extern "C" void fokl_ctx_destroy(fokl_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)fokl_comm_destroy(ctx);
    drain_events(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    free_slots(ctx);
    if (ctx->d_slot_ptr) (void)hipFree(ctx->d_slot_ptr);
    if (ctx->d_x) (void)hipFree(ctx->d_x);
    if (ctx->d_phis) (void)hipFree(ctx->d_phis);
    if (ctx->d_zero) (void)hipFree(ctx->d_zero);
    if (ctx->d_staged) (void)hipFree(ctx->d_staged);
    if (ctx->d_args) (void)hipFree(ctx->d_args);
    if (ctx->h_args) (void)hipHostFree(ctx->h_args);
    if (ctx->d_slab) (void)hipFree(ctx->d_slab);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->d_gout) (void)hipFree(ctx->d_gout);
    if (ctx->h_gout) (void)hipHostFree(ctx->h_gout);
    if (ctx->d_rout) (void)hipFree(ctx->d_rout);
    if (ctx->h_rout) (void)hipHostFree(ctx->h_rout);
    if (ctx->d_comm) (void)hipFree(ctx->d_comm);
    if (ctx->args_free) (void)hipEventDestroy(ctx->args_free);
    if (ctx->gram_done) (void)hipEventDestroy(ctx->gram_done);
    if (ctx->resid_done) (void)hipEventDestroy(ctx->resid_done);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int fokl_sync(fokl_ctx *ctx)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_sync: null context");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    return FOKL_OK;
}

extern "C" int fokl_slot_capacity(const fokl_ctx *ctx) { return ctx ? (int)ctx->slot_ptr.size() : 0; }
extern "C" int64_t fokl_rows(const fokl_ctx *ctx) { return ctx ? ctx->n : 0; }

// ---------------------------------------------------------------------------------------------------------
// slots
// ---------------------------------------------------------------------------------------------------------

extern "C" int fokl_reserve_slots(fokl_ctx *ctx, int n_slots)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_reserve_slots: null context");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_reserve_slots: call fokl_upload first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    bool grew = false;
    while ((int)ctx->slot_ptr.size() < n_slots) {
        double *chunk = nullptr;
        const size_t bytes = (size_t)fokl_ctx::CHUNK_SLOTS * ctx->ld * sizeof(double);
        HIP_TRY(ctx, hipMalloc((void **)&chunk, bytes));
        ctx->chunks.push_back(chunk);
        for (int i = 0; i < fokl_ctx::CHUNK_SLOTS; ++i) ctx->slot_ptr.push_back(chunk + (size_t)i * ctx->ld);
        grew = true;
    }
    if (grew) {
        const int cap = (int)ctx->slot_ptr.size();
        if (cap > ctx->table_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_slot_ptr) HIP_TRY(ctx, hipFree(ctx->d_slot_ptr));
            ctx->d_slot_ptr = nullptr;
            ctx->table_cap = 0;
            const int new_cap = std::max(cap * 2, 256);
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_slot_ptr, (size_t)new_cap * sizeof(double *)));
            ctx->table_cap = new_cap;
        }
        // synchronous copy: the host vector may be reallocated by a later call
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(ctx->d_slot_ptr, ctx->slot_ptr.data(), (size_t)cap * sizeof(double *),
                               hipMemcpyHostToDevice));
    }
    return FOKL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// dataset upload
// ---------------------------------------------------------------------------------------------------------

// x: host rows (fokl_upload) or NULL: the copy fokl_stage_inputs left on the device, normalised on the way
// (xT[k][i] = (x[i][k] - lows[k]) / spans[k], the reference's two separately rounded operations, FR:436-437).
static int upload_dataset(fokl_ctx *ctx, const double *x, const double *y, int64_t n, int m, int kernel, const double *phis,
                          int n_basis, int width, const double *lows, const double *spans, const char *who)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, std::string(who) + ": null context");
    if (!y || !phis) return fail(ctx, FOKL_ERR_ARG, std::string(who) + ": null pointer");
    if (n <= 0 || m <= 0 || m > 4096) return fail(ctx, FOKL_ERR_ARG, std::string(who) + ": need n > 0 and 0 < m <= 4096");
    if (kernel != FOKL_KERNEL_SPLINES && kernel != FOKL_KERNEL_BERNOULLI)
        return fail(ctx, FOKL_ERR_ARG, std::string(who) + ": unknown kernel id");
    if (n_basis <= 0 || width <= 0) return fail(ctx, FOKL_ERR_ARG, std::string(who) + ": empty coefficient table");
    if (kernel == FOKL_KERNEL_BERNOULLI && width < n_basis + 1)
        return fail(ctx, FOKL_ERR_ARG, std::string(who) + ": Bernoulli table needs width >= n_basis + 1");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);

    // release the previous dataset
    free_slots(ctx);
    if (ctx->d_x) HIP_TRY(ctx, hipFree(ctx->d_x));
    if (ctx->d_phis) HIP_TRY(ctx, hipFree(ctx->d_phis));
    if (ctx->d_zero) HIP_TRY(ctx, hipFree(ctx->d_zero));
    ctx->d_x = nullptr;
    ctx->d_phis = nullptr;
    ctx->d_zero = nullptr;
    ctx->have_data = false;

    ctx->n = n;
    ctx->m = m;
    ctx->ld = (n + 63) / 64 * 64;
    ctx->kernel = kernel;
    ctx->n_basis = n_basis;
    ctx->width = width;
    ctx->phis_doubles = (kernel == FOKL_KERNEL_SPLINES) ? (size_t)n_basis * 4 * width : (size_t)n_basis * width;

    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_x, (size_t)m * ctx->ld * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_phis, ctx->phis_doubles * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_zero, (size_t)ctx->ld * sizeof(double)));
    HIP_TRY(ctx, hipMemset(ctx->d_zero, 0, (size_t)ctx->ld * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_phis, phis, ctx->phis_doubles * sizeof(double), hipMemcpyHostToDevice));
    ctx->h_phis.assign(phis, phis + ctx->phis_doubles);

    ctx->have_data = true;
    int rc = fokl_reserve_slots(ctx, fokl_ctx::CHUNK_SLOTS);
    if (rc) return rc;

    // raw row-major copy in a temporary, transposed on the device
    double *d_raw = nullptr, *d_y = nullptr, *d_bounds = nullptr;
    if (x) {
        HIP_TRY(ctx, hipMalloc((void **)&d_raw, (size_t)n * m * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(d_raw, x, (size_t)n * m * sizeof(double), hipMemcpyHostToDevice));
    } else {
        d_raw = ctx->d_staged;
        ctx->d_staged = nullptr;
        HIP_TRY(ctx, hipMalloc((void **)&d_bounds, (size_t)2 * m * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(d_bounds, lows, (size_t)m * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(d_bounds + m, spans, (size_t)m * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMalloc((void **)&d_y, (size_t)n * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(d_y, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    const int blocks = (int)std::min<int64_t>((ctx->ld + 255) / 256, 4096);
    hipLaunchKernelGGL(transpose_inputs_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_raw, d_y, n, m, ctx->ld,
                       ctx->d_x, ctx->slot_ptr[FOKL_SLOT_ONES], ctx->slot_ptr[FOKL_SLOT_Y], d_bounds);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(d_raw));
    HIP_TRY(ctx, hipFree(d_y));
    if (d_bounds) HIP_TRY(ctx, hipFree(d_bounds));
    return FOKL_OK;
}

extern "C" int fokl_upload(fokl_ctx *ctx, const double *x, const double *y, int64_t n, int m, int kernel,
                           const double *phis, int n_basis, int width)
{
    if (ctx && !x) return fail(ctx, FOKL_ERR_ARG, "fokl_upload: null pointer");
    return upload_dataset(ctx, x, y, n, m, kernel, phis, n_basis, width, nullptr, nullptr, "fokl_upload");
}

// FoKL.clean's normalisation on the device, in two calls around the host's few lines of bookkeeping (minmax / pillow):
// fokl_stage_inputs copies the RAW rows and returns every column's minimum and maximum; fokl_upload_staged lays the
// staged rows out as fokl_upload does, each value normalised on the way.
extern "C" int fokl_stage_inputs(fokl_ctx *ctx, const double *x, int64_t n, int m, double *lows_out, double *highs_out)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_stage_inputs: null context");
    if (!x || !lows_out || !highs_out) return fail(ctx, FOKL_ERR_ARG, "fokl_stage_inputs: null pointer");
    if (n <= 0 || m <= 0 || m > 4096) return fail(ctx, FOKL_ERR_ARG, "fokl_stage_inputs: need n > 0 and 0 < m <= 4096");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->d_staged) HIP_TRY(ctx, hipFree(ctx->d_staged));
    ctx->d_staged = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_staged, (size_t)n * m * sizeof(double)));
    ctx->staged_n = n;
    ctx->staged_m = m;
    HIP_TRY(ctx, hipMemcpy(ctx->d_staged, x, (size_t)n * m * sizeof(double), hipMemcpyHostToDevice));
    // lane t of the grid owns column t mod m (the stride is a multiple of m); partial (min, max) per lane, then per column
    const int64_t total = n * (int64_t)m;
    int64_t lanes = std::min<int64_t>((int64_t)1024 * 256, (total + 7) / 8);
    lanes = std::max<int64_t>(m, lanes / m * m);
    const int blocks = (int)((lanes + 255) / 256);
    double *d_part = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_part, (size_t)(2 * lanes + 2 * m) * sizeof(double)));
    hipLaunchKernelGGL(column_bounds_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_staged, total, m, lanes, d_part);
    hipLaunchKernelGGL(column_bounds_finish_kernel, dim3(m), dim3(256), 0, ctx->stream, d_part, lanes, m, d_part + 2 * lanes);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<double> h((size_t)2 * m);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_part + 2 * lanes, (size_t)2 * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(d_part));
    for (int k = 0; k < m; ++k) {
        lows_out[k] = h[(size_t)k];
        highs_out[k] = h[(size_t)m + k];
    }
    return FOKL_OK;
}

extern "C" int fokl_upload_staged(fokl_ctx *ctx, const double *y, int64_t n, int m, int kernel, const double *phis,
                                  int n_basis, int width, const double *lows, const double *spans)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_upload_staged: null context");
    if (!lows || !spans) return fail(ctx, FOKL_ERR_ARG, "fokl_upload_staged: null pointer");
    if (!ctx->d_staged || ctx->staged_n != n || ctx->staged_m != m)
        return fail(ctx, FOKL_ERR_STATE, "fokl_upload_staged: no staged inputs of this shape (fokl_stage_inputs first)");
    return upload_dataset(ctx, nullptr, y, n, m, kernel, phis, n_basis, width, lows, spans, "fokl_upload_staged");
}

// The inputs as the kernels see them (normalised), back in rows: x_out [n, m].
extern "C" int fokl_download_inputs(fokl_ctx *ctx, double *x_out)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_download_inputs: null context");
    if (!x_out) return fail(ctx, FOKL_ERR_ARG, "fokl_download_inputs: null pointer");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_download_inputs: call fokl_upload first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double *d_rows = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_rows, (size_t)ctx->n * ctx->m * sizeof(double)));
    const int blocks = (int)std::min<int64_t>((ctx->n + 255) / 256, 4096);
    hipLaunchKernelGGL(rows_from_columns_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_x, ctx->n, ctx->m, ctx->ld, d_rows);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(x_out, d_rows, (size_t)ctx->n * ctx->m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(d_rows));
    return FOKL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K1 launch planning
// ---------------------------------------------------------------------------------------------------------

// Compute units of the device: every launch plan sizes its grid for this number.
static int cu_count(fokl_ctx *ctx) { return ctx->cus > 0 ? ctx->cus : 256; }

static int env_int(const char *name, int fallback);
static int dev_int(const char *name, int fallback);

// LDS available to one workgroup of the basis kernel; the factor table takes 4 KB per distinct factor.
static constexpr size_t K1_LDS_BUDGET = 144 * 1024;

static int launch_basis(fokl_ctx *ctx, const int32_t *terms, const int32_t *slots, int t_begin, int t_end,
                        const DerivSpec &deriv)
{
    const int m = ctx->m;
    const bool splines = ctx->kernel == FOKL_KERNEL_SPLINES;
    // distinct (input, order) factors of this group, ordered by input then order
    std::map<std::pair<int, int>, int> fac_id;
    for (int j = t_begin; j < t_end; ++j)
        for (int k = 0; k < m; ++k) {
            const int o = terms[(size_t)j * m + k];
            if (o != 0) fac_id.emplace(std::make_pair(k, o), 0);
        }
    int U = 0;
    for (auto &kv : fac_id) kv.second = U++;
    const int T = t_end - t_begin;

    // spline orders staged in LDS (most used first would be ideal; any K1_MAX_LDS_SLABS distinct ones do)
    std::vector<int> slab_orders;
    if (splines) {
        for (auto &kv : fac_id) {
            const int o = kv.first.second;
            if (std::find(slab_orders.begin(), slab_orders.end(), o) == slab_orders.end() &&
                (int)slab_orders.size() < K1_MAX_LDS_SLABS)
                slab_orders.push_back(o);
        }
    }
    const int NS = (int)slab_orders.size();
    size_t slab_doubles = splines ? (((size_t)NS * 4 * ctx->width + 1) & ~(size_t)1) : 0;
    // Workgroup size: the factor table costs 16 B per lane and factor, so big tables with 256 lanes leave only one
    // or two workgroups per CU; smaller workgroups pack the 160 KB of LDS with more wavefronts (each staged spline
    // slab is paid per workgroup, which pushes the other way).  FOKL_K1_THREADS overrides (experiments).
    int threads = K1_THREADS;
    {
        int best_waves = -1;
        for (int t : {256, 128, 64}) {
            const size_t need = slab_doubles * sizeof(double) + (size_t)std::max(U, 1) * t * sizeof(d2);
            const int blocks = (int)std::min<size_t>(16, (160 * 1024) / std::max<size_t>(need, 1));
            const int waves = std::min(32, blocks * (t / 64));
            if (waves > best_waves) {
                best_waves = waves;
                threads = t;
            }
        }
        if (const char *env = getenv("FOKL_K1_THREADS")) {
            const int t = atoi(env);
            if (t == 64 || t == 128 || t == 256) threads = t;
        }
    }
    const bool reg_table = U <= K1_REG_FACTORS;        // factor table in VGPRs: LDS only holds the spline slabs
    if (reg_table) threads = K1_THREADS;
    size_t lds_bytes = slab_doubles * sizeof(double) +
                       (reg_table ? 0 : (size_t)std::max(U, 1) * threads * sizeof(d2));

    size_t n_fac_entries = 0;
    for (int j = t_begin; j < t_end; ++j)
        for (int k = 0; k < m; ++k) n_fac_entries += terms[(size_t)j * m + k] != 0;

    const size_t n_ints = (size_t)3 * U + K1_MAX_LDS_SLABS + (T + 1) + n_fac_entries + T;
    const size_t plan_bytes = sizeof(BasisPlan);
    const size_t bytes = plan_bytes + n_ints * sizeof(int);
    int rc = begin_args(ctx, bytes);
    if (rc) return rc;
    BasisPlan *plan = reinterpret_cast<BasisPlan *>(ctx->h_args);
    plan->n_fac = U;
    plan->n_terms = T;
    plan->n_slabs = NS;
    plan->pad = 0;
    int *arr = reinterpret_cast<int *>(ctx->h_args + plan_bytes);
    int *fac_input = arr, *fac_order = arr + U, *fac_slab = arr + 2 * U, *slab_order = arr + 3 * U;
    int *term_off = slab_order + K1_MAX_LDS_SLABS, *term_fac = term_off + (T + 1);
    int *term_slot = term_fac + n_fac_entries;
    for (auto &kv : fac_id) {
        const int u = kv.second;
        fac_input[u] = kv.first.first;
        fac_order[u] = kv.first.second;
        int s = -1;
        for (int q = 0; q < NS; ++q)
            if (slab_orders[q] == kv.first.second) s = q;
        fac_slab[u] = s;
    }
    for (int q = 0; q < K1_MAX_LDS_SLABS; ++q) slab_order[q] = q < NS ? slab_orders[q] : 1;
    int off = 0;
    for (int j = t_begin; j < t_end; ++j) {
        term_off[j - t_begin] = off;
        for (int k = 0; k < m; ++k) {
            const int o = terms[(size_t)j * m + k];
            if (o != 0) term_fac[off++] = fac_id[std::make_pair(k, o)];
        }
        term_slot[j - t_begin] = slots[j];
    }
    term_off[T] = off;
    rc = push_args(ctx, bytes);
    if (rc) return rc;

    const int64_t tile_rows = (int64_t)threads * K1_ROWS_PER_THREAD;
    const int64_t n_tiles = (ctx->n + tile_rows - 1) / tile_rows;
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(reg_table ? 5 : 16, (160 * 1024) / std::max<size_t>(lds_bytes, 1)));
    per_cu = std::min(per_cu, std::max(1, dev_int("FOKL_K1_WGS", per_cu)));
    const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)cu_count(ctx) * per_cu);
    const BasisPlan *d_plan = reinterpret_cast<const BasisPlan *>(ctx->d_args);
    const int *d_arr = reinterpret_cast<const int *>(ctx->d_args + plan_bytes);

    int m_used = 0;
    {
        int last = -1;
        for (auto &kv : fac_id)
            if (kv.first.first != last) {
                last = kv.first.first;
                ++m_used;
            }
    }
    // (In a fit K1 comes behind Gram and residual launches that stream 0.5-1 GB of columns.  Those streams are marked
    // non-temporal, FOKL_STREAM_NT in fokl_kernels.hip.h, so that they do not evict the 8 N m bytes of inputs from the 256 MB
    // Infinity Cache: K1 reads them from there, 0.61-0.65 of the HBM roof with no helper launch.  Round 4's separate read of
    // the inputs ahead of every build -- 0.50-0.54 for the pair -- is gone.)
    const double alg_bytes = 8.0 * (double)ctx->n * (double)(m_used + T);
    TimedRegion timed(ctx, FOKL_K_BASIS, alg_bytes, 0.0);
    typedef void (*basis_fn)(const double *, int64_t, int64_t, const double *, int, const BasisPlan *, const int *,
                             double *const *, DerivSpec);
    basis_fn fn = splines ? (reg_table ? basis_build_reg_kernel<true> : basis_build_kernel<true>)
                          : (reg_table ? basis_build_reg_kernel<false> : basis_build_kernel<false>);
    if (lds_bytes > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)K1_LDS_BUDGET + 4096));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds_bytes, ctx->stream, ctx->d_x, ctx->ld, ctx->n, ctx->d_phis,
                       ctx->width, d_plan, d_arr, ctx->d_slot_ptr, deriv);
    HIP_TRY(ctx, hipGetLastError());
    return FOKL_OK;
}

static int build_terms_impl(fokl_ctx *ctx, const int32_t *terms, int T, const int32_t *slots, const DerivSpec &deriv);

extern "C" int fokl_build_terms(fokl_ctx *ctx, const int32_t *terms, int T, const int32_t *slots)
{
    const DerivSpec plain = {-1, 0, 1.0, 0};
    return build_terms_impl(ctx, terms, T, slots, plain);
}

extern "C" int fokl_build_terms_deriv(fokl_ctx *ctx, const int32_t *terms, int T, const int32_t *slots, int wrt_input,
                                      int order, double divisor)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_build_terms_deriv: null context");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_build_terms_deriv: call fokl_upload first");
    if (wrt_input < 0 || wrt_input >= ctx->m || (order != 1 && order != 2) || !(divisor != 0.0))
        return fail(ctx, FOKL_ERR_ARG, "fokl_build_terms_deriv: need 0 <= wrt_input < m, order 1 or 2, divisor != 0");
    for (int j = 0; j < T; ++j)
        if (terms && terms[(size_t)j * ctx->m + wrt_input] == 0)
            return fail(ctx, FOKL_ERR_ARG, "fokl_build_terms_deriv: a term does not contain the differentiated input "
                                           "(its derivative is the zero column; leave it out)");
    const DerivSpec spec = {wrt_input, order, divisor, 1};
    return build_terms_impl(ctx, terms, T, slots, spec);
}

static int build_terms_impl(fokl_ctx *ctx, const int32_t *terms, int T, const int32_t *slots, const DerivSpec &deriv)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_build_terms: null context");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_build_terms: call fokl_upload first");
    if (T < 0 || (T > 0 && (!terms || !slots))) return fail(ctx, FOKL_ERR_ARG, "fokl_build_terms: null pointer");
    if (T == 0) return FOKL_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_slots(ctx, slots, T, "fokl_build_terms");
    if (rc) return rc;
    const int m = ctx->m;
    for (int j = 0; j < T; ++j) {
        if (slots[j] < FOKL_SLOT_FIRST_FREE)
            return fail(ctx, FOKL_ERR_ARG, "fokl_build_terms: slots 0 and 1 are reserved (ones, y)");
        int nz = 0;
        for (int k = 0; k < m; ++k) {
            const int o = terms[(size_t)j * m + k];
            if (o < 0 || o > ctx->n_basis)
                return fail(ctx, FOKL_ERR_ARG, "fokl_build_terms: basis order " + std::to_string(o) +
                                                   " outside [0, " + std::to_string(ctx->n_basis) + "]");
            nz += o != 0;
        }
        if (nz == 0) return fail(ctx, FOKL_ERR_ARG, "fokl_build_terms: term with no input (all-zero row)");
    }
    // Split into launches whose distinct-factor table fits the LDS budget.
    const bool splines = ctx->kernel == FOKL_KERNEL_SPLINES;
    const size_t slab_bytes = splines ? (size_t)K1_MAX_LDS_SLABS * 4 * ctx->width * sizeof(double) + 16 : 0;
    const int lds_fac = (int)((K1_LDS_BUDGET - slab_bytes) / (K1_THREADS * sizeof(d2)));
    // groups of terms whose distinct factors fit the register table take the faster kernel; only a single term with
    // more factors than that falls back to the LDS table
    int max_fac = K1_REG_FACTORS;
    for (int j = 0; j < T; ++j) {
        int nz = 0;
        for (int k = 0; k < m; ++k) nz += terms[(size_t)j * m + k] != 0;
        if (nz > max_fac) max_fac = std::min(nz, lds_fac);
    }
    int begin = 0;
    while (begin < T) {
        std::map<std::pair<int, int>, int> seen;
        int end = begin;
        while (end < T) {
            std::vector<std::pair<int, int>> add;
            for (int k = 0; k < m; ++k) {
                const int o = terms[(size_t)end * m + k];
                if (o != 0 && !seen.count({k, o})) add.push_back({k, o});
            }
            if ((int)(seen.size() + add.size()) > max_fac) break;
            for (auto &p : add) seen[p] = 1;
            ++end;
        }
        if (end == begin)
            return fail(ctx, FOKL_ERR_ARG, "fokl_build_terms: a single term has more factors than fit in LDS");
        rc = launch_basis(ctx, terms, slots, begin, end, deriv);
        if (rc) return rc;
        begin = end;
    }
    return FOKL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K2: Gram blocks
// ---------------------------------------------------------------------------------------------------------

extern "C" int fokl_comm_allreduce_sum_f64(fokl_ctx *ctx, double *buf, int count);
static int comm_allreduce_device(fokl_ctx *ctx, double *d_buf, size_t count);   // fokl_comm.inc: in place, on the stream

// Output elements handled by one 256-thread block of the slab reduction: few elements -> many parts per element.
static int reduce_elements_per_block(int total)
{
    // 16 elements (one 128-byte line per slab) x 16 parts while that keeps the grid below ~4096 blocks: many blocks with
    // short per-thread chains; wider blocks only for very large outputs
    int epb = 1;
    while (epb < 16 && epb * 64 < total) epb *= 2;
    while (epb < 64 && (total + epb - 1) / epb > 4096) epb *= 2;
    return epb;
}

// Resident workgroups per CU for a kernel (occupancy API; advisory, only used to size the row split).
template <typename F>
static int blocks_per_cu(F fn, int threads, size_t dyn_lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, dyn_lds) != hipSuccess || nb < 1) nb = 1;
    return std::min(nb, 8);
}

#ifdef FOKL_DEV_KERNELS
// One instantiation per tile configuration; the dispatcher below picks the smallest that covers the block.
typedef void (*gram_mfma_fn)(double *const *, const int *, int, const int *, int, int64_t, double *, int, int,
                             const double *);

template <int TJ>
static gram_mfma_fn pick_isplit()
{
    return gram_mfma_kernel<4, TJ, true>;
}

static gram_mfma_fn isplit_kernel(int tj)
{
    switch (tj) {
        case 1: return pick_isplit<1>();
        case 2: return pick_isplit<2>();
        case 3: return pick_isplit<3>();
        case 4: return pick_isplit<4>();
        case 5: return pick_isplit<5>();
        case 6: return pick_isplit<6>();
        case 7: return pick_isplit<7>();
        case 8: return pick_isplit<8>();
        case 9: return pick_isplit<9>();
        case 10: return pick_isplit<10>();
        case 11: return pick_isplit<11>();
        default: return pick_isplit<12>();
    }
}

static gram_mfma_fn jsplit_kernel(int ti, int tj)
{
    if (ti == 1) {
        if (tj == 1) return gram_mfma_kernel<1, 1, false>;
        if (tj == 2) return gram_mfma_kernel<1, 2, false>;
        return gram_mfma_kernel<1, 3, false>;
    }
    if (tj == 1) return gram_mfma_kernel<2, 1, false>;
    if (tj == 2) return gram_mfma_kernel<2, 2, false>;
    return gram_mfma_kernel<2, 3, false>;
}

#endif  // FOKL_DEV_KERNELS

// ---- tile-list Gram (gram_tiles_kernel): launch planning ---------------------------------------------------

struct GramPlan {
    int nci = 0, it = 0, jt = 0;       // internal columns; i-tiles (row side), j-tiles (all internal columns)
    int nt = 1, ct = 1, ks = 1, rb_shift = 0, depth = 1;
    int kind = 0, waves = 4;           // 0: gram_tiles_kernel (16x16x4 MFMA), 1: gram_tiles4s_kernel (4x4x4 MFMA)
    bool half = false;                 // entries 0 and 1 of every list are half-tile slots (gram_tiles_dma_kernel<.., true>)
    std::vector<int32_t> icols;        // internal column -> slot: the row-side columns first
    std::vector<int32_t> perm;         // caller's column j -> internal column
    std::vector<GramGroup> groups;
};

static int env_int(const char *name, int fallback)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
}

// Knobs that select between measured variants of a kernel (A/B runs: docs/LAB_NOTEBOOK.md) exist in development builds
// only (make DEV=1): the product library runs one configuration, the one the `-m gpu` suite covers.
static int dev_int(const char *name, int fallback)
{
#ifdef FOKL_DEV_KERNELS
    return env_int(name, fallback);
#else
    (void)name;
    return fallback;
#endif
}

// Internal column order, tile groups and kernel parameters for an nr x nc block (see gram_tiles_kernel).  Pure host
// arithmetic: fokl_gram_plan exposes it to the CPU tests, which replay the lists with numpy.
static void plan_gram(const int32_t *row_slots, int nr, const int32_t *col_slots, int nc, int kind, GramPlan &pl,
                      bool allow_half = false)
{
    pl.kind = kind;
    pl.waves = 4;
#ifdef FOKL_DEV_KERNELS
    const int nt_max = kind == 1 ? G4S_MAX_NT : GT_MAX_NT;
    const int ct_max = kind == 1 ? G4S_MAX_CT : GT_MAX_CT;
#else
    const int nt_max = GT_MAX_NT, ct_max = GT_MAX_CT;
    (void)kind;
#endif
    pl.icols.assign(row_slots, row_slots + nr);
    pl.perm.resize(nc);
    {
        std::map<int32_t, int> where;
        for (int i = 0; i < nr; ++i) where.emplace(row_slots[i], i);            // first occurrence wins
        for (int j = 0; j < nc; ++j) {
            auto f = where.find(col_slots[j]);
            if (f != where.end()) {
                pl.perm[j] = f->second;
            } else {
                pl.perm[j] = (int)pl.icols.size();
                where.emplace(col_slots[j], (int)pl.icols.size());
                pl.icols.push_back(col_slots[j]);
            }
        }
    }
    pl.nci = (int)pl.icols.size();
    pl.it = (nr + 15) / 16;
    pl.jt = (pl.nci + 15) / 16;

    struct Cut {
        std::vector<std::pair<int, int>> tiles;
        std::vector<int> staged;
    };
    std::vector<Cut> cuts;
    const int max_tiles = pl.waves * nt_max;
    for (int b0 = 0; b0 < pl.it; b0 += 4) {
        const int i_hi = std::min(pl.it, b0 + 4);
        std::vector<std::pair<int, int>> seq;              // tiles on or above the diagonal, column of tiles by column
        for (int jt = b0; jt < pl.jt; ++jt)
            for (int it = b0; it < std::min(i_hi, jt + 1); ++it) seq.emplace_back(it, jt);
        const int T = (int)seq.size();
        for (int G = std::max(1, (T + max_tiles - 1) / max_tiles);; ++G) {
            std::vector<Cut> trial;
            bool fits = true;
            int at = 0;
            for (int gi = 0; gi < G && fits; ++gi) {
                const int size = T / G + (gi < T % G ? 1 : 0);
                Cut c;
                c.tiles.assign(seq.begin() + at, seq.begin() + at + size);
                at += size;
                for (auto &t : c.tiles) {
                    c.staged.push_back(t.first);
                    c.staged.push_back(t.second);
                }
                std::sort(c.staged.begin(), c.staged.end());
                c.staged.erase(std::unique(c.staged.begin(), c.staged.end()), c.staged.end());
                fits = (int)c.staged.size() <= ct_max && size <= max_tiles;
                if (size > 0) trial.push_back(std::move(c));
            }
            if (fits) {
                for (auto &c : trial) cuts.push_back(std::move(c));
                break;
            }
        }
    }

    int most_tiles = 1, most_staged = 1;
    for (auto &c : cuts) {
        most_tiles = std::max(most_tiles, (int)c.tiles.size());
        most_staged = std::max(most_staged, (int)c.staged.size());
    }
    pl.ks = kind != 0 ? 1 : most_tiles == 1 ? 4 : most_tiles == 2 ? 2 : 1;
    const int teams = pl.waves / pl.ks;
    pl.nt = (most_tiles + teams - 1) / teams;
    pl.ct = most_staged;
    // rows per chunk: as many sub-chunks of 32 rows as the 16 staging passes and FOKL_GRAM_RB allow
    const int rb_cap = std::max(1, std::min(16, dev_int("FOKL_GRAM_RB", 1)));
    pl.rb_shift = 0;
    while (kind == 0 && (pl.ct << (pl.rb_shift + 1)) <= GT_MAX_PASS && (2 << pl.rb_shift) <= rb_cap) ++pl.rb_shift;
    pl.depth = kind == 0 && pl.nt <= 4 ? std::max(1, std::min(2, dev_int("FOKL_GRAM_DEPTH", 1))) : 1;

    // Half-tile slots (gram_tiles_dma_kernel<NT, NBUF, true>): when 1 .. 8 of the 16 columns of the last row tile are
    // row-side columns, a tile of that row tile costs the matrix pipe half a tile if it is formed as 8 x 16.  Entries 0
    // and 1 of every list then are half-tile slots (one per wavefront; empty where there is nothing for them), the
    // ordinary tiles follow from entry 2 and go to the list with the least work so far (ordinary tile = 2, half = 1).
    // Only if every group keeps within 8 ordinary entries per list; the caller says whether the kernel is the one to run.
    // (not for a single row tile: such blocks are HBM-bound and only pay for the extra slot: 8 x 40 measured 57 -> 65 us)
    const int ragged = allow_half && kind == 0 && pl.ks == 1 && pl.it >= 2 && nr % 16 >= 1 && nr % 16 <= 8 ? pl.it - 1 : -1;
    const int normal_cap = GT_MAX_NT - 2;
    pl.half = ragged >= 0;
    for (const Cut &c : cuts) {
        int rag = 0;
        for (auto &t : c.tiles) rag += t.first == ragged ? 1 : 0;
        const int full = (int)c.tiles.size() - std::min(rag, 2 * teams);
        if (full > teams * normal_cap) pl.half = false;
    }
    pl.groups.resize(cuts.size());
    int most_normal = 1;
    for (size_t gi = 0; gi < cuts.size(); ++gi) {
        const Cut &c = cuts[gi];
        GramGroup &g = pl.groups[gi];
        std::memset(&g, 0, sizeof(g));
        for (int p = 0; p < GT_MAX_CT; ++p) g.ct[p] = p < (int)c.staged.size() ? c.staged[p] : -1;
        for (int w = 0; w < GT_MAX_WAVES; ++w)
            for (int k = 0; k < GT_MAX_NT; ++k) g.oi[w][k] = g.oj[w][k] = 0xFFFF;
        const int T = (int)c.tiles.size();
        std::vector<std::pair<int, int>> place(T);               // tile -> (team, entry)
        if (pl.half) {
            int len[GT_MAX_WAVES] = {0, 0, 0, 0}, load[GT_MAX_WAVES] = {0, 0, 0, 0}, halves = 0;
            std::vector<char> done(T, 0);
            for (int t = 0; t < T && halves < 2 * teams; ++t)
                if (c.tiles[t].first == ragged) {
                    place[t] = {halves % teams, halves / teams};
                    load[halves % teams] += 1;
                    done[t] = 1;
                    ++halves;
                }
            for (int t = 0; t < T; ++t) {
                if (done[t]) continue;
                int team = -1;
                for (int w = 0; w < teams; ++w)
                    if (len[w] < normal_cap && (team < 0 || load[w] < load[team])) team = w;
                place[t] = {team, 2 + len[team]++};
                load[team] += 2;
            }
            for (int w = 0; w < teams; ++w) most_normal = std::max(most_normal, len[w]);
        } else {
            for (int t = 0; t < T; ++t) place[t] = {t % teams, t / teams};
        }
        for (int t = 0; t < T; ++t) {
            const int team = place[t].first, k = place[t].second;
            const int a = (int)(std::lower_bound(c.staged.begin(), c.staged.end(), c.tiles[t].first) - c.staged.begin());
            const int b = (int)(std::lower_bound(c.staged.begin(), c.staged.end(), c.tiles[t].second) - c.staged.begin());
            for (int w = team * pl.ks; w < (team + 1) * pl.ks; ++w) {
                g.a[w][k] = (uint8_t)a;
                g.b[w][k] = (uint8_t)b;
                g.oi[w][k] = (uint16_t)c.tiles[t].first;
                g.oj[w][k] = (uint16_t)c.tiles[t].second;
            }
        }
    }
    if (pl.half) pl.nt = most_normal;                            // ordinary entries per list (the kernel adds the slots)
}

typedef void (*gram_tiles_fn)(double *const *, const int *, int, const GramGroup *, int, int, int64_t, double *, int,
                              int, const double *, const double *);

// Lowest address among the slot chunks and the zero column, if every one of them lies within 2^31 units of 256 bytes
// above it on that grid (gram_tiles_kernel keeps columns as 32-bit distances); nullptr otherwise.
static const double *slot_grid_base(fokl_ctx *ctx)
{
    uintptr_t lo = reinterpret_cast<uintptr_t>(ctx->d_zero), hi = lo;
    for (double *c : ctx->chunks) {
        lo = std::min(lo, reinterpret_cast<uintptr_t>(c));
        hi = std::max(hi, reinterpret_cast<uintptr_t>(c));
    }
    if ((ctx->ld * sizeof(double)) % 256 != 0 || (reinterpret_cast<uintptr_t>(ctx->d_zero) - lo) % 256 != 0) return nullptr;
    for (double *c : ctx->chunks)
        if ((reinterpret_cast<uintptr_t>(c) - lo) % 256 != 0) return nullptr;
    const uintptr_t span = hi - lo + (uintptr_t)fokl_ctx::CHUNK_SLOTS * ctx->ld * sizeof(double);
    if ((span >> 8) >= (uintptr_t(1) << 31)) return nullptr;          // bit 31 of a distance flags a padding column
    return reinterpret_cast<const double *>(lo);
}

// Instantiations: NT = 1..4 (HBM-bound shapes) for P = 4, 8, 16 staging passes and one or two chunks in flight, the
// k-split teams (NT = 1) likewise; NT = 5..10 (MFMA-bound) for P = 16, one chunk in flight.
template <int NT, int P>
static gram_tiles_fn tiles_kernel_np(int depth, int ks)
{
    if (NT == 1 && ks == 4) return depth == 2 ? gram_tiles_kernel<1, P, 2, 4> : gram_tiles_kernel<1, P, 1, 4>;
    if (NT == 1 && ks == 2) return depth == 2 ? gram_tiles_kernel<1, P, 2, 2> : gram_tiles_kernel<1, P, 1, 2>;
    if (depth == 2) return gram_tiles_kernel<NT, P, 2, 1>;
    return gram_tiles_kernel<NT, P, 1, 1>;
}

template <int NT>
static gram_tiles_fn tiles_kernel_n(int passes, int depth, int ks)
{
    if (passes <= 4) return tiles_kernel_np<NT, 4>(depth, ks);
    if (passes <= 8) return tiles_kernel_np<NT, 8>(depth, ks);
    return tiles_kernel_np<NT, 16>(depth, ks);
}

static gram_tiles_fn tiles_kernel(int nt, int passes, int depth, int ks)
{
    switch (nt) {
        case 1: return tiles_kernel_n<1>(passes, depth, ks);
        case 2: return tiles_kernel_n<2>(passes, depth, ks);
        case 3: return tiles_kernel_n<3>(passes, depth, ks);
        case 4: return tiles_kernel_n<4>(passes, depth, ks);
        case 5: return gram_tiles_kernel<5, 16, 1, 1>;
        case 6: return gram_tiles_kernel<6, 16, 1, 1>;
        case 7: return gram_tiles_kernel<7, 16, 1, 1>;
        case 8: return gram_tiles_kernel<8, 16, 1, 1>;
        case 9: return gram_tiles_kernel<9, 16, 1, 1>;
        default: return gram_tiles_kernel<10, 16, 1, 1>;
    }
}

#ifdef FOKL_DEV_KERNELS
typedef void (*gram_tiles4_fn)(double *const *, const int *, int, const GramGroup *, int, int64_t, double *, int, int,
                               const double *, const double *);

template <int NT>
static gram_tiles4_fn tiles4s_kernel_n(int passes)
{
    if (passes <= 2) return gram_tiles4s_kernel<NT, 2>;
    if (passes <= 4) return gram_tiles4s_kernel<NT, 4>;
    if (passes <= 6) return gram_tiles4s_kernel<NT, 6>;
    return gram_tiles4s_kernel<NT, 8>;
}

static gram_tiles4_fn tiles4s_kernel(int nt, int passes)
{
    switch (nt) {
        case 1: return tiles4s_kernel_n<1>(passes);
        case 2: return tiles4s_kernel_n<2>(passes);
        case 3: return tiles4s_kernel_n<3>(passes);
        default: return tiles4s_kernel_n<4>(passes);
    }
}

#endif  // FOKL_DEV_KERNELS

typedef void (*gram_dma_fn)(const GramGroup *, int, int, int64_t, double *, int, int, const double *, uint32_t);

// nt8: ordinary tiles per wavefront (1 .. 5; with half-tile slots 1 .. 4 + the slot); LW: loader wavefronts (0, 2, 4)
template <int NBUF, int LW>
static gram_dma_fn tiles_dma_kernel_b(int nt8, bool half)
{
    if (half) {
        switch (nt8) {
            case 1: return gram_tiles_dma_kernel<1, NBUF, true, LW>;
            case 2: return gram_tiles_dma_kernel<2, NBUF, true, LW>;
            case 3: return gram_tiles_dma_kernel<3, NBUF, true, LW>;
            default: return gram_tiles_dma_kernel<4, NBUF, true, LW>;
        }
    }
    switch (nt8) {
        case 1: return gram_tiles_dma_kernel<1, NBUF, false, LW>;
        case 2: return gram_tiles_dma_kernel<2, NBUF, false, LW>;
        case 3: return gram_tiles_dma_kernel<3, NBUF, false, LW>;
        case 4: return gram_tiles_dma_kernel<4, NBUF, false, LW>;
        default: return gram_tiles_dma_kernel<5, NBUF, false, LW>;
    }
}

static gram_dma_fn tiles_dma_kernel(int nt8, int nbuf, bool half, int loaders = 0)
{
#ifdef FOKL_DEV_KERNELS
    if (nbuf == 3) return tiles_dma_kernel_b<3, 0>(nt8, half);
#endif
    (void)nbuf;
    if (loaders == 2) return tiles_dma_kernel_b<2, 2>(nt8, half);
    if (loaders == 4) return tiles_dma_kernel_b<2, 4>(nt8, half);
    return tiles_dma_kernel_b<2, 0>(nt8, half);
}

// Half-tile slots need the LDS-DMA kernel: asked for only where it runs every block (the default), FOKL_GRAM_HALF=0 for A/B
static bool half_slots_wanted()
{
#ifdef FOKL_DEV_KERNELS
    if (dev_int("FOKL_GRAM_MFMA4", 0) == 2) return false;
#endif
    return dev_int("FOKL_GRAM_DMA", 2) == 2 && dev_int("FOKL_GRAM_HALF", 1) != 0;
}

extern "C" int fokl_gram_plan(const int32_t *row_slots, int nr, const int32_t *col_slots, int nc, int kind, int32_t *info,
                              int32_t *icols, int32_t *perm, int32_t *staged, int32_t *tiles, int cap_groups)
{
    if (!row_slots || !col_slots || !info || nr <= 0 || nc <= 0 || kind < 0 || kind > 1)
        return fail(nullptr, FOKL_ERR_ARG, "fokl_gram_plan: bad argument");
#ifndef FOKL_DEV_KERNELS
    if (kind == 1) return fail(nullptr, FOKL_ERR_ARG, "fokl_gram_plan: kind 1 (4x4x4 tile lists) needs a development build");
#endif
    GramPlan pl;
    plan_gram(row_slots, nr, col_slots, nc, kind, pl, half_slots_wanted());
    const int32_t head[10] = {pl.nci, pl.it, pl.jt, (int32_t)pl.groups.size(), pl.nt, pl.ct, pl.rb_shift, pl.ks, pl.depth,
                              pl.waves};
    std::memcpy(info, head, sizeof(head));
    if ((int)pl.groups.size() > cap_groups) return FOKL_OK;      // sizes only: call again with room for the lists
    if (icols) std::memcpy(icols, pl.icols.data(), pl.icols.size() * sizeof(int32_t));
    if (perm) std::memcpy(perm, pl.perm.data(), pl.perm.size() * sizeof(int32_t));
    for (size_t gi = 0; gi < pl.groups.size(); ++gi) {
        const GramGroup &g = pl.groups[gi];
        if (staged) std::memcpy(staged + gi * GT_MAX_CT, g.ct, sizeof(g.ct));
        if (tiles)
            for (int w = 0; w < GT_MAX_WAVES; ++w)
                for (int k = 0; k < GT_MAX_NT; ++k) {
                    int32_t *t = tiles + ((gi * GT_MAX_WAVES + w) * GT_MAX_NT + k) * 4;
                    t[0] = g.a[w][k] | (pl.half && k < 2 && g.oi[w][k] != 0xFFFF ? 256 : 0);   // + 256: in a half-tile slot
                    t[1] = g.b[w][k];
                    t[2] = g.oi[w][k] == 0xFFFF ? -1 : g.oi[w][k];
                    t[3] = g.oj[w][k] == 0xFFFF ? -1 : g.oj[w][k];
                }
    }
    return FOKL_OK;
}

// Everything of a Gram block up to the copy into pinned host memory, enqueued on the context's stream.  aside: the
// result goes to the d_gout / h_gout pair (fokl_gram_launch) instead of d_out / h_out.
static int gram_enqueue(fokl_ctx *ctx, const int32_t *row_slots, int nr, const int32_t *col_slots, int nc, int path,
                        int allreduce, bool aside)
{
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_gram: call fokl_upload first");
    if (nr <= 0 || nc <= 0) return fail(ctx, FOKL_ERR_ARG, "fokl_gram: empty block");
    if (path < 0 || path > 3) return fail(ctx, FOKL_ERR_ARG, "fokl_gram: path must be 0, 1, 2 or 3");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_slots(ctx, row_slots, nr, "fokl_gram");
    if (rc) return rc;
    rc = check_slots(ctx, col_slots, nc, "fokl_gram");
    if (rc) return rc;

    // path choice: the VALU kernel re-reads operands once per 4x4 register tile, fine while the block is small;
    // the MFMA kernels read every column of a group of tiles once per row chunk.
    if (path == 0) path = (int64_t)nr * nc > 64 ? dev_int("FOKL_GRAM_PATH", 2) : 1;
    if (path < 1 || path > 3) path = 2;
    const double *grid_base = path == 2 ? slot_grid_base(ctx) : nullptr;
    if (path == 2 && !grid_base) path = 3;                   // slots off the 256-byte grid (never seen): panel kernel
    const int cus = cu_count(ctx);
    rc = aside ? ensure_gout(ctx, (size_t)nr * nc) : ensure_out(ctx, (size_t)nr * nc);
    if (rc) return rc;
    double *d_dst = aside ? ctx->d_gout : ctx->d_out;
    double *h_dst = aside ? ctx->h_gout : ctx->h_out;

    // algorithmic traffic: every distinct column read once (row-side columns usually reappear on the column side)
    std::vector<int32_t> uniq(row_slots, row_slots + nr);
    uniq.insert(uniq.end(), col_slots, col_slots + nc);
    std::sort(uniq.begin(), uniq.end());
    const double distinct = (double)(std::unique(uniq.begin(), uniq.end()) - uniq.begin());
    const double bytes = 8.0 * (double)ctx->n * distinct;
    const double flops = 2.0 * (double)ctx->n * (double)nr * (double)nc;
    // the launch is booked under the roof that binds it
    const int gram_slot = flops / FOKL_PEAK_F64_FLOPS > bytes / FOKL_PEAK_HBM_BYTES_PER_S ? FOKL_K_GRAM_MFMA : FOKL_K_GRAM;
    if (const char *trace = std::getenv("FOKL_GRAM_TRACE")) {   // profile runs: which class each launch was booked under
        if (FILE *fh = std::fopen(trace, "a")) {
            std::fprintf(fh, "%d %d %d %s\n", nr, nc, (int)distinct, gram_slot == FOKL_K_GRAM_MFMA ? "gram_mfma" : "gram");
            std::fclose(fh);
        }
    }
    const int total = nr * nc;
    const int epb = reduce_elements_per_block(total);

    if (path == 2) {
        // tile lists over the internal column order; tiles below the diagonal of the row-side x row-side part skipped
        // FOKL_GRAM_MFMA4=2: the 4x4x4 form of the fp64 MFMA instruction (A/B runs; slower beyond the smallest blocks)
#ifdef FOKL_DEV_KERNELS
        const int kind = dev_int("FOKL_GRAM_MFMA4", 0) == 2 ? 1 : 0;
#else
        const int kind = 0;
#endif
        GramPlan pl;
        plan_gram(row_slots, nr, col_slots, nc, kind, pl, half_slots_wanted());
        // column addresses for the groups' descriptors (distances on the 256-byte slot grid)
        {
            const uintptr_t lo = reinterpret_cast<uintptr_t>(grid_base);
            const uint32_t zero_units = (uint32_t)((reinterpret_cast<uintptr_t>(ctx->d_zero) - lo) >> 8) | 0x80000000u;
            for (GramGroup &g : pl.groups)
                for (int p = 0; p < GT_MAX_CT; ++p)
                    for (int c = 0; c < 16; ++c) {
                        const int col = g.ct[p] >= 0 ? 16 * g.ct[p] + c : -1;
                        g.col_units[p][c] = col >= 0 && col < pl.nci
                                                ? (uint32_t)((reinterpret_cast<uintptr_t>(ctx->slot_ptr[pl.icols[col]]) - lo) >> 8)
                                                : zero_units;
                    }
        }
        const size_t ints = (size_t)pl.nci + (size_t)nc;
        const size_t group_off = (ints * sizeof(int32_t) + 15) / 16 * 16;
        const size_t arg_bytes = group_off + pl.groups.size() * sizeof(GramGroup);
        rc = begin_args(ctx, arg_bytes);
        if (rc) return rc;
        std::memcpy(ctx->h_args, pl.icols.data(), (size_t)pl.nci * sizeof(int32_t));
        std::memcpy(ctx->h_args + (size_t)pl.nci * sizeof(int32_t), pl.perm.data(), (size_t)nc * sizeof(int32_t));
        std::memcpy(ctx->h_args + group_off, pl.groups.data(), pl.groups.size() * sizeof(GramGroup));
        rc = push_args(ctx, arg_bytes);
        if (rc) return rc;
        const int *d_icols = reinterpret_cast<const int *>(ctx->d_args);
        const int *d_perm = d_icols + pl.nci;
        const GramGroup *d_groups = reinterpret_cast<const GramGroup *>(ctx->d_args + group_off);

        static std::mutex attr_mutex;
        static std::map<const void *, size_t> attr_set;            // kernels whose dynamic LDS limit was raised already
        auto raise_lds_limit = [&](const void *fn, size_t lds) -> int {
            std::lock_guard<std::mutex> lock(attr_mutex);
            size_t &have = attr_set[fn];
            if (have < lds) {
                HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                have = 160 * 1024;
            }
            return FOKL_OK;
        };
        const int nr_pad = 16 * pl.it, nc_pad = 16 * pl.jt;
        // workgroups per CU that share the rows of a group: more of them hide more latency but every one writes a
        // partial block the reduction has to read back (FOKL_GRAM_WGS caps it)
        const int wgs_cap = std::max(1, dev_int("FOKL_GRAM_WGS", 3));
        // The row cut S decides which rows meet in which partial sum, i.e. the last bits of the block.  It must move neither
        // with a compiler or an edit that changes a kernel's register count, nor with the device the fit happens to run on (a
        // partition with fewer CUs, another SKU), nor with what the occupancy query says today: it is the cut of kCutCus CUs
        // hosting a PINNED number of workgroups each -- what these kernels could host when the goldens' margins were
        // measured, under the 160 KB of LDS counted by arithmetic -- so the same rows meet in the same partial sums
        // everywhere and a golden measured on one device holds on another.  Resident or queued, every
        // workgroup has the same work.
        constexpr int kCutCus = 256;
        auto lds_fit = [](size_t lds) { return (int)std::max<size_t>(1, (size_t)(160 * 1024) / std::max<size_t>(lds, 1)); };
        int S;
#ifdef FOKL_DEV_KERNELS
        if (kind == 1) {
            const int P = pl.ct <= 2 ? 2 : pl.ct <= 4 ? 4 : pl.ct <= 6 ? 6 : 8;
            gram_tiles4_fn fn = tiles4s_kernel(pl.nt, pl.ct);
            const size_t lds = (size_t)P * 16 * G4_PITCH * sizeof(double);
            rc = raise_lds_limit(reinterpret_cast<const void *>(fn), lds);
            if (rc) return rc;
            const int64_t n_chunks = (ctx->n + 31) / 32;
            const int per_cu = std::min(wgs_cap, blocks_per_cu(fn, G4S_THREADS, lds));
            const int target = std::max(1, (per_cu * cus) / (int)pl.groups.size());      // (A/B kernel: no pinned row cut)
            S = (int)std::max<int64_t>(1, std::min<int64_t>(n_chunks, target));
            rc = ensure_slab(ctx, (size_t)S * nr_pad * nc_pad);
            if (rc) return rc;
            TimedRegion timed(ctx, gram_slot, bytes, flops);      // brackets the Gram kernel only
            hipLaunchKernelGGL(fn, dim3(S, (unsigned)pl.groups.size()), dim3(G4S_THREADS), lds, ctx->stream,
                               ctx->d_slot_ptr, d_icols, pl.nci, d_groups, pl.ct, ctx->n, ctx->d_slab, nr_pad, nc_pad,
                               ctx->d_zero, grid_base);
        } else
#endif
        if (pl.ks == 1 && (dev_int("FOKL_GRAM_DMA", 2) == 2 ||
                                  (dev_int("FOKL_GRAM_DMA", 2) == 1 && gram_slot == FOKL_K_GRAM_MFMA))) {
            // LDS-DMA staging, 8 wavefronts per workgroup, two LDS buffers: every block of three tiles or more
            // (FOKL_GRAM_DMA=1: only the launches the matrix pipe bounds, 0: gram_tiles_kernel for everything)
            const int pieces = (pl.ct * 16 * 34 * 8 + 1023) / 1024;
            // three LDS buffers (a chunk's pieces in flight across the barrier) where they fit, FOKL_GRAM_BUFS to force
            int nbuf = 2;
#ifdef FOKL_DEV_KERNELS
            nbuf = dev_int("FOKL_GRAM_BUFS", 2);
            if (nbuf == 3 && 3 * (size_t)pieces * 1024 > 160 * 1024) nbuf = 2;
#endif
            gram_dma_fn plain_fn = tiles_dma_kernel((pl.nt + 1) / 2, nbuf, pl.half);
            const size_t lds = (size_t)nbuf * pieces * 1024;
            rc = raise_lds_limit(reinterpret_cast<const void *>(plain_fn), lds);
            if (rc) return rc;
            const int64_t n_chunks = (ctx->n + 31) / 32;
            // The row cut decides which rows meet in which partial sum, i.e. the last bits of the block: it must not move
            // when a compiler or an edit changes the kernel's register count.  Pinned to what these kernels could host
            // when the goldens' margins were measured (three workgroups per CU for one tile per wavefront, two beyond),
            // under the LDS limit the occupancy query reports; resident or queued, every workgroup has the same work.
            const int by_registers = (pl.nt + 1) / 2 == 1 ? 3 : 2;
            const int per_cu = std::min({wgs_cap, blocks_per_cu(plain_fn, GD_THREADS, lds), by_registers});
            // Four loader wavefronts issue the LDS-DMA pieces instead of the matrix wavefronts (gram_tiles_dma_kernel<.., 4>)
            // wherever a CU can host as many of the larger workgroups as the row cut puts on it -- fewer resident
            // workgroups cost more than the loaders bring (56 x 98: 195 -> 216 us).  FOKL_GRAM_LOADERS=0: never.  Same tiles,
            // same order of summation: the block's bits do not depend on it.
            int loaders = nbuf == 2 ? dev_int("FOKL_GRAM_LOADERS", 4) : 0;
            if (loaders != 2 && loaders != 4) loaders = 0;
            gram_dma_fn fn = plain_fn;
            if (loaders) {
                gram_dma_fn with = tiles_dma_kernel((pl.nt + 1) / 2, nbuf, pl.half, loaders);
                rc = raise_lds_limit(reinterpret_cast<const void *>(with), lds);
                if (rc) return rc;
                if (blocks_per_cu(with, GD_THREADS + 64 * loaders, lds) >= per_cu)
                    fn = with;
                else
                    loaders = 0;
            }
            const int cut_per_cu = std::min({wgs_cap, by_registers, lds_fit(lds)});
            const int target = std::max(1, (cut_per_cu * kCutCus) / (int)pl.groups.size());
            S = (int)std::max<int64_t>(1, std::min<int64_t>(n_chunks, target));
            rc = ensure_slab(ctx, (size_t)S * nr_pad * nc_pad);
            if (rc) return rc;
            const uint32_t zero_units = (uint32_t)((reinterpret_cast<uintptr_t>(ctx->d_zero) - reinterpret_cast<uintptr_t>(grid_base)) >> 8);
            TimedRegion timed(ctx, gram_slot, bytes, flops);      // brackets the Gram kernel only
            hipLaunchKernelGGL(fn, dim3(S, (unsigned)pl.groups.size()), dim3(GD_THREADS + 64 * loaders), lds, ctx->stream,
                               d_groups, pl.ct, pieces, ctx->n, ctx->d_slab, nr_pad, nc_pad, grid_base, zero_units);
        } else {
            gram_tiles_fn fn = tiles_kernel(pl.nt, pl.ct << pl.rb_shift, pl.depth, pl.ks);
            const int R = 32 << pl.rb_shift;
            const size_t lds = (size_t)pl.ct * 16 * (R + 2) * sizeof(double);
            rc = raise_lds_limit(reinterpret_cast<const void *>(fn), lds);
            if (rc) return rc;
            const int64_t n_chunks = (ctx->n + R - 1) / R;
            // the product's launches here are the k-split teams of the smallest blocks (ks = 2, 4: one tile per wavefront):
            // three workgroups per CU by registers except with 16 staging passes and two chunks in flight (187 VGPRs: two);
            // ks = 1 only runs here under FOKL_GRAM_DMA != 2 (A/B runs): the occupancy query, as before
            const int passes = pl.ct << pl.rb_shift;
            const int by_registers = pl.ks > 1 ? (passes > 8 && pl.depth == 2 ? 2 : 3) : blocks_per_cu(fn, GT_THREADS, lds);
            const int cut_per_cu = std::min({wgs_cap, by_registers, lds_fit(lds)});
            const int target = std::max(1, (cut_per_cu * (pl.ks > 1 ? kCutCus : cus)) / (int)pl.groups.size());
            S = (int)std::max<int64_t>(1, std::min<int64_t>(n_chunks, target));
            rc = ensure_slab(ctx, (size_t)S * pl.ks * nr_pad * nc_pad);
            if (rc) return rc;
            TimedRegion timed(ctx, gram_slot, bytes, flops);      // brackets the Gram kernel only
            hipLaunchKernelGGL(fn, dim3(S, (unsigned)pl.groups.size()), dim3(GT_THREADS), lds, ctx->stream,
                               ctx->d_slot_ptr, d_icols, pl.nci, d_groups, pl.ct, pl.rb_shift, ctx->n, ctx->d_slab,
                               nr_pad, nc_pad, ctx->d_zero, grid_base);
        }
        HIP_TRY(ctx, hipGetLastError());
        {
            // the reduction is a kernel of its own in the timing table (and in rocprofv3's): slabs read + block written
            TimedRegion timed(ctx, FOKL_K_GRAM_REDUCE, 8.0 * ((double)S * pl.ks + 1.0) * total, 0.0);
            hipLaunchKernelGGL(reduce_slabs_sym_kernel, dim3((total + epb - 1) / epb), dim3(RD_THREADS), 0, ctx->stream,
                               ctx->d_slab, S * pl.ks, nr, nc, nr_pad, nc_pad, epb, d_perm, d_dst);
        }
        HIP_TRY(ctx, hipGetLastError());
    } else {
        const size_t arg_bytes = (size_t)(nr + nc) * sizeof(int);
        rc = begin_args(ctx, arg_bytes);
        if (rc) return rc;
        int *h = reinterpret_cast<int *>(ctx->h_args);
        std::memcpy(h, row_slots, (size_t)nr * sizeof(int));
        std::memcpy(h + nr, col_slots, (size_t)nc * sizeof(int));
        rc = push_args(ctx, arg_bytes);
        if (rc) return rc;
        const int *d_rows = reinterpret_cast<const int *>(ctx->d_args);
        const int *d_cols = d_rows + nr;
        int S, nr_pad, nc_pad;
        dim3 grid;
#ifndef FOKL_DEV_KERNELS
        if (path == 3) return fail(ctx, FOKL_ERR_ARG, "fokl_gram: path 3 (round-1 panel kernel) needs a development build");
        {
#else
        gram_mfma_fn mfma_fn = nullptr;
        if (path == 3) {                                     // rectangular panels, one i-tile or j-tile set per wavefront
            int BI, BJ;
            const int j_tiles = (nc + 15) / 16;
            if (nr > 32) {                                   // i-split: wave w <-> i-tile w, panel of TJ j-tiles
                const int panels = (j_tiles + 11) / 12;
                const int tj = (j_tiles + panels - 1) / panels;
                mfma_fn = isplit_kernel(tj);
                BI = 64;
                BJ = 16 * tj;
            } else {                                         // j-split: every wave all i-tiles, j-tiles dealt over waves
                const int ti = nr > 16 ? 2 : 1;
                const int per_wave = (j_tiles + 3) / 4;
                const int panels = (per_wave + 2) / 3;
                const int tj = (per_wave + panels - 1) / panels;
                mfma_fn = jsplit_kernel(ti, tj);
                BI = 16 * ti;
                BJ = 64 * tj;
            }
            const int gz = (nr + BI - 1) / BI, gy = (nc + BJ - 1) / BJ;
            nr_pad = gz * BI;
            nc_pad = gy * BJ;
            const int64_t n_chunks = (ctx->n + GM_R - 1) / GM_R;
            const int per_cu = blocks_per_cu(mfma_fn, GM_THREADS, 0);
            const int target = std::max(1, (per_cu * cus) / (gz * gy));
            S = (int)std::max<int64_t>(1, std::min<int64_t>(n_chunks, target));
            grid = dim3(S, gy, gz);
        } else {
#endif
            const int gz = (nr + GV_TI - 1) / GV_TI, gy = (nc + GV_TJ - 1) / GV_TJ;
            nr_pad = gz * GV_TI;
            nc_pad = gy * GV_TJ;
            const int64_t n_row_blocks = (ctx->n + GV_THREADS * 2 - 1) / (GV_THREADS * 2);
            const int target = std::max(1, (8 * cus) / (gz * gy));
            S = (int)std::max<int64_t>(1, std::min<int64_t>(n_row_blocks, target));
            grid = dim3(S, gy, gz);
        }
        rc = ensure_slab(ctx, (size_t)S * nr_pad * nc_pad);
        if (rc) return rc;
        {
            TimedRegion timed(ctx, gram_slot, bytes, flops);      // brackets the Gram kernel only
#ifdef FOKL_DEV_KERNELS
            if (path == 3) {
                hipLaunchKernelGGL(mfma_fn, grid, dim3(GM_THREADS), 0, ctx->stream, ctx->d_slot_ptr, d_rows, nr, d_cols,
                                   nc, ctx->n, ctx->d_slab, nr_pad, nc_pad, ctx->d_zero);
            } else
#endif
            {
                hipLaunchKernelGGL(gram_valu_kernel, grid, dim3(GV_THREADS), 0, ctx->stream, ctx->d_slot_ptr, d_rows,
                                   nr, d_cols, nc, ctx->n, ctx->d_slab, nr_pad, nc_pad);
            }
        }
        {
            TimedRegion timed(ctx, FOKL_K_GRAM_REDUCE, 8.0 * ((double)S + 1.0) * total, 0.0);
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((total + epb - 1) / epb), dim3(RD_THREADS), 0, ctx->stream,
                               ctx->d_slab, S, nr, nc, nr_pad, nc_pad, epb, d_dst);
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    // row-sharded fit: the block is summed over the ranks where it lies -- RCCL on the device, stream-ordered behind
    // the reduction kernel -- and crosses PCIe once
    if (allreduce && ctx->comm) {
        rc = comm_allreduce_device(ctx, d_dst, (size_t)nr * nc);
        if (rc) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_dst, (size_t)nr * nc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return FOKL_OK;
}

extern "C" int fokl_gram(fokl_ctx *ctx, const int32_t *row_slots, int nr, const int32_t *col_slots, int nc,
                         double *out, int path, int allreduce)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_gram: null context");
    if (!out) return fail(ctx, FOKL_ERR_ARG, "fokl_gram: null output");
    int rc = gram_enqueue(ctx, row_slots, nr, col_slots, nc, path, allreduce, false);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->h_out, (size_t)nr * nc * sizeof(double));
    return FOKL_OK;
}

// fokl_gram in two halves: the launch returns at once, other work may be launched behind it (residual passes, basis
// builds), fokl_gram_fetch waits for the block only.  One block can be on its way at a time: a launch drops a block
// that was not fetched.
extern "C" int fokl_gram_launch(fokl_ctx *ctx, const int32_t *row_slots, int nr, const int32_t *col_slots, int nc,
                                int allreduce)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_gram_launch: null context");
    ctx->gram_pending = 0;                  // a block nobody fetched is dropped (the stream orders the overwrite)
    int rc = gram_enqueue(ctx, row_slots, nr, col_slots, nc, 0, allreduce, true);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->gram_done, ctx->stream));
    ctx->gram_pending = (size_t)nr * nc;
    return FOKL_OK;
}

// 1: the block launched by fokl_gram_launch has arrived in host memory (fokl_gram_fetch will not wait), 0: not yet,
// < 0: nothing launched / an error.
extern "C" int fokl_gram_ready(fokl_ctx *ctx)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_gram_ready: null context");
    if (!ctx->gram_pending) return fail(ctx, FOKL_ERR_STATE, "fokl_gram_ready: nothing launched");
    const hipError_t st = hipEventQuery(ctx->gram_done);
    if (st == hipSuccess) return 1;
    if (st == hipErrorNotReady) return 0;
    return fail(ctx, FOKL_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(st));
}

extern "C" int fokl_gram_fetch(fokl_ctx *ctx, double *out, int64_t count)
{
    if (!ctx || !out) return fail(ctx, FOKL_ERR_ARG, "fokl_gram_fetch: null pointer");
    if (!ctx->gram_pending) return fail(ctx, FOKL_ERR_STATE, "fokl_gram_fetch: nothing launched");
    if (count != (int64_t)ctx->gram_pending)
        return fail(ctx, FOKL_ERR_ARG, "fokl_gram_fetch: the launched block has a different size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->gram_done));
    std::memcpy(out, ctx->h_gout, ctx->gram_pending * sizeof(double));
    ctx->gram_pending = 0;
    return FOKL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K3: residual moments
// ---------------------------------------------------------------------------------------------------------

extern "C" int fokl_bic_resid_launch(fokl_ctx *ctx, const int32_t *slots, int nc, const double *betahat)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_bic_resid_launch: null context");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_bic_resid_launch: call fokl_upload first");
    if (nc <= 0 || !betahat) return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_launch: empty model or null pointer");
    if (ctx->resid_pending) return fail(ctx, FOKL_ERR_STATE, "fokl_bic_resid_launch: previous launch not fetched");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_slots(ctx, slots, nc, "fokl_bic_resid_launch");
    if (rc) return rc;

    const size_t beta_off = ((size_t)nc * sizeof(int) + 7) & ~(size_t)7;
    const size_t arg_bytes = beta_off + (size_t)nc * sizeof(double);
    rc = begin_args(ctx, arg_bytes);
    if (rc) return rc;
    std::memcpy(ctx->h_args, slots, (size_t)nc * sizeof(int));
    std::memcpy(ctx->h_args + beta_off, betahat, (size_t)nc * sizeof(double));
    rc = push_args(ctx, arg_bytes);
    if (rc) return rc;

    const int64_t n_row_blocks = (ctx->n + RS_THREADS * 2 - 1) / (RS_THREADS * 2);
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(n_row_blocks, (int64_t)cu_count(ctx) * 8));
    rc = ensure_slab(ctx, (size_t)S * 2);
    if (rc) return rc;
    rc = ensure_rout(ctx);
    if (rc) return rc;
    {
        {
            TimedRegion timed(ctx, FOKL_K_RESID, 8.0 * (double)ctx->n * (double)(nc + 1), 2.0 * (double)ctx->n * nc);
            hipLaunchKernelGGL(resid_kernel, dim3(S), dim3(RS_THREADS), 0, ctx->stream, ctx->d_slot_ptr,
                               reinterpret_cast<const int *>(ctx->d_args),
                               nc, reinterpret_cast<const double *>(ctx->d_args + beta_off),
                               ctx->slot_ptr[FOKL_SLOT_Y], ctx->n, ctx->d_slab);
        }
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1), dim3(RD_THREADS), 0, ctx->stream, ctx->d_slab, S, 1, 2, 1, 2,
                           2, ctx->d_rout);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_rout, ctx->d_rout, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->resid_done, ctx->stream));
    ctx->resid_pending = true;
    return FOKL_OK;
}

typedef void (*resid_quad_fn)(const double *, int64_t, int64_t, const double *, int, const ResidQuadTable *, const double *,
                              double *);

// the slot layouts compiled: GM inputs x KM orders per input (fewest slots first); Bernoulli instances for orders up to 2, 4
// (what a search visits first) and RT_MAX_ORDER
struct ResidQuadLayout {
    int gm, km;
    resid_quad_fn bernoulli2, bernoulli4, bernoulli8, splines;
};
#define FOKL_RQ_LAYOUT(GM, KM)                                                                          \
    {GM, KM, resid_quadratic_kernel<false, GM, KM, 2>, resid_quadratic_kernel<false, GM, KM, 4>,        \
     resid_quadratic_kernel<false, GM, KM, RT_MAX_ORDER>, resid_quadratic_kernel<true, GM, KM, 2>}
static const ResidQuadLayout kResidQuadLayouts[] = {FOKL_RQ_LAYOUT(8, 1), FOKL_RQ_LAYOUT(16, 1), FOKL_RQ_LAYOUT(8, 2),
                                                    FOKL_RQ_LAYOUT(4, 4), FOKL_RQ_LAYOUT(2, 8), FOKL_RQ_LAYOUT(8, 4),
                                                    FOKL_RQ_LAYOUT(16, 2), FOKL_RQ_LAYOUT(4, 8)};
#undef FOKL_RQ_LAYOUT

extern "C" int fokl_bic_resid_terms_launch(fokl_ctx *ctx, const int32_t *terms, int n_terms, const double *betahat)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_bic_resid_terms_launch: null context");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_bic_resid_terms_launch: call fokl_upload first");
    if (n_terms < 0 || !betahat || (n_terms > 0 && !terms))
        return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_terms_launch: null pointer");
    if (ctx->resid_pending) return fail(ctx, FOKL_ERR_STATE, "fokl_bic_resid_terms_launch: previous launch not fetched");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int m = ctx->m;
    const bool splines = ctx->kernel == FOKL_KERNEL_SPLINES;

    // distinct (input, order) factors, ordered by input then order; a term has one or two of them
    std::map<int, std::vector<int>> orders_of;                   // input -> its orders, ascending
    for (int j = 0; j < n_terms; ++j) {
        int nz = 0;
        for (int k = 0; k < m; ++k) {
            const int o = terms[(size_t)j * m + k];
            if (o < 0 || o > ctx->n_basis)
                return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_terms_launch: basis order " + std::to_string(o) +
                                                   " outside [0, " + std::to_string(ctx->n_basis) + "]");
            if (o != 0) {
                if (!splines && o > RT_MAX_ORDER)
                    return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_terms_launch: Bernoulli order " + std::to_string(o) +
                                                       " beyond " + std::to_string(RT_MAX_ORDER) +
                                                       " (FOKL_RESID_TERMS_MAX_ORDER)");
                std::vector<int> &os = orders_of[k];
                auto at = std::lower_bound(os.begin(), os.end(), o);
                if (at == os.end() || *at != o) os.insert(at, o);
                ++nz;
            }
        }
        if (nz == 0) return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_terms_launch: term with no input (all-zero row)");
        if (nz > 2)
            return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_terms_launch: a term has more than two factors (the stored-column "
                                           "pass, fokl_bic_resid_launch, takes such models)");
    }
    const int G = (int)orders_of.size();
    int K = 1, top_order = 1;
    for (const auto &kv : orders_of) {
        K = std::max(K, (int)kv.second.size());
        top_order = std::max(top_order, kv.second.back());
    }
    const ResidQuadLayout *layout = nullptr;
    for (const ResidQuadLayout &l : kResidQuadLayouts)
        if (G <= l.gm && K <= l.km) {
            layout = &l;
            break;
        }
    if (!layout)
        return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_terms_launch: " + std::to_string(G) + " inputs with up to " +
                                           std::to_string(K) + " orders each fit none of the factor layouts (8 x 1, 16 x 1, "
                                           "8 x 2, 4 x 4, 2 x 8, 8 x 4, 16 x 2, 4 x 8: FOKL_RESID_TERMS_MAX_FACTORS slots)");
    const int GM = layout->gm, KM = layout->km, UM = GM * KM;

    int rc = begin_args(ctx, sizeof(ResidQuadTable));
    if (rc) return rc;
    ResidQuadTable *tab = reinterpret_cast<ResidQuadTable *>(ctx->h_args);
    std::memset(tab, 0, sizeof(ResidQuadTable));
    tab->n_groups = G;
    std::map<std::pair<int, int>, int> slot_of;                  // (input, order) -> slot
    double flops_per_row = 5.0;
    {
        int g = 0;
        for (const auto &kv : orders_of) {
            tab->input[g] = kv.first;
            tab->omax[g] = kv.second.back();
            flops_per_row += 8.0 * std::max(0, kv.second.back() - 1);        // the double-double power chain
            for (size_t k = 0; k < kv.second.size(); ++k) {
                const int o = kv.second[k], u = g * KM + (int)k;
                slot_of[std::make_pair(kv.first, o)] = u;
                if (!splines)
                    for (int q = 0; q <= o; ++q) tab->coef[u][q] = ctx->h_phis[(size_t)(o - 1) * ctx->width + q];
                tab->coef[u][9] = (double)o;
                flops_per_row += splines ? 14.0 : 2.0 * o;
            }
            ++g;
        }
    }
    tab->c0 = betahat[0];
    for (int j = 0; j < n_terms; ++j) {
        int fa = -1, fb = -1;
        for (int k = 0; k < m; ++k) {
            const int o = terms[(size_t)j * m + k];
            if (o != 0) (fa < 0 ? fa : fb) = slot_of[std::make_pair(k, o)];
        }
        if (fb < 0)
            tab->lin[fa] += betahat[j + 1];
        else                                                     // (ascending input order: fa < fb, different groups)
            tab->quad[resid_quad_index(fa, fb, UM, KM)] += betahat[j + 1];
    }
    flops_per_row += 2.0 * 2.0 * (UM + UM * (UM - 1) / 2 - GM * (KM * (KM - 1) / 2));     // the form's fused multiply-adds
    rc = push_args(ctx, sizeof(ResidQuadTable));
    if (rc) return rc;

    const int64_t n_row_blocks = (ctx->n + RS_THREADS * 2 - 1) / (RS_THREADS * 2);
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(n_row_blocks, (int64_t)cu_count(ctx) * 8));
    rc = ensure_slab(ctx, (size_t)S * 2);
    if (rc) return rc;
    rc = ensure_rout(ctx);
    if (rc) return rc;
    resid_quad_fn fn = splines ? layout->splines : top_order <= 2 ? layout->bernoulli2 : top_order <= 4 ? layout->bernoulli4 : layout->bernoulli8;
    {
        {
            TimedRegion timed(ctx, FOKL_K_RESID_MF, 8.0 * (double)ctx->n * (double)(G + 1),
                              (double)ctx->n * flops_per_row);
            hipLaunchKernelGGL(fn, dim3(S), dim3(RS_THREADS), 0, ctx->stream, ctx->d_x, ctx->ld, ctx->n, ctx->d_phis,
                               ctx->width, reinterpret_cast<const ResidQuadTable *>(ctx->d_args), ctx->slot_ptr[FOKL_SLOT_Y],
                               ctx->d_slab);
        }
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1), dim3(RD_THREADS), 0, ctx->stream, ctx->d_slab, S, 1, 2, 1, 2,
                           2, ctx->d_rout);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_rout, ctx->d_rout, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->resid_done, ctx->stream));
    ctx->resid_pending = true;
    return FOKL_OK;
}

extern "C" int fokl_bic_resid_fetch(fokl_ctx *ctx, double *out, int allreduce)
{
    if (!ctx || !out) return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid_fetch: null pointer");
    if (!ctx->resid_pending) return fail(ctx, FOKL_ERR_STATE, "fokl_bic_resid_fetch: nothing was launched");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->resid_pending = false;
    if (allreduce && ctx->comm) {
        // the launch left this rank's two moments in d_rout: sum them over the ranks there, copy again
        int rc = comm_allreduce_device(ctx, ctx->d_rout, 2);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_rout, ctx->d_rout, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->resid_done, ctx->stream));
    }
    // (not the stream: the driver may have queued the coming sub-stage's Gram launch behind this pass)
    HIP_TRY(ctx, hipEventSynchronize(ctx->resid_done));
    out[0] = ctx->h_rout[0];
    out[1] = ctx->h_rout[1];
    return FOKL_OK;
}

extern "C" int fokl_bic_resid(fokl_ctx *ctx, const int32_t *slots, int nc, const double *betahat, double *out,
                              int allreduce)
{
    if (!out) return fail(ctx, FOKL_ERR_ARG, "fokl_bic_resid: null output");
    int rc = fokl_bic_resid_launch(ctx, slots, nc, betahat);
    if (rc) return rc;
    return fokl_bic_resid_fetch(ctx, out, allreduce);
}

// ---------------------------------------------------------------------------------------------------------
// slot access (tests, evaluate)
// ---------------------------------------------------------------------------------------------------------

extern "C" int fokl_read_slot(fokl_ctx *ctx, int slot, int64_t row0, int64_t nrows, double *host)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_read_slot: null context");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_read_slot: call fokl_upload first");
    int32_t s = slot;
    int rc = check_slots(ctx, &s, 1, "fokl_read_slot");
    if (rc) return rc;
    if (row0 < 0 || nrows < 0 || row0 + nrows > ctx->n || (nrows > 0 && !host))
        return fail(ctx, FOKL_ERR_ARG, "fokl_read_slot: row range outside the dataset");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(host, ctx->slot_ptr[slot] + row0, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost));
    return FOKL_OK;
}

extern "C" int fokl_write_slot(fokl_ctx *ctx, int slot, int64_t row0, int64_t nrows, const double *host)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_write_slot: null context");
    if (!ctx->have_data) return fail(ctx, FOKL_ERR_STATE, "fokl_write_slot: call fokl_upload first");
    int32_t s = slot;
    int rc = check_slots(ctx, &s, 1, "fokl_write_slot");
    if (rc) return rc;
    if (row0 < 0 || nrows < 0 || row0 + nrows > ctx->n || (nrows > 0 && !host))
        return fail(ctx, FOKL_ERR_ARG, "fokl_write_slot: row range outside the dataset");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->slot_ptr[slot] + row0, host, (size_t)nrows * sizeof(double), hipMemcpyHostToDevice));
    return FOKL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------------------------------------

extern "C" int fokl_timing_enable(fokl_ctx *ctx, int on)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_timing_enable: null context");
    ctx->timing = on != 0;
    return FOKL_OK;
}

extern "C" int fokl_timing_reset(fokl_ctx *ctx)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_timing_reset: null context");
    (void)hipStreamSynchronize(ctx->stream);
    drain_events(ctx);
    for (auto &t : ctx->tslot) t = TimingSlot();
    return FOKL_OK;
}

extern "C" int fokl_timing_get(fokl_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches, double *bytes,
                               double *flops, double *ideal_ms)
{
    if (!ctx) return fail(nullptr, FOKL_ERR_ARG, "fokl_timing_get: null context");
    if (kernel_id < 0 || kernel_id >= FOKL_K_COUNT) return fail(ctx, FOKL_ERR_ARG, "fokl_timing_get: bad kernel id");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    if (total_ms) *total_ms = ctx->tslot[kernel_id].ms;
    if (launches) *launches = ctx->tslot[kernel_id].launches;
    if (bytes) *bytes = ctx->tslot[kernel_id].bytes;
    if (flops) *flops = ctx->tslot[kernel_id].flops;
    if (ideal_ms) *ideal_ms = ctx->tslot[kernel_id].ideal_ms;
    return FOKL_OK;
}

// the RCCL half lives in fokl_comm.hip; it needs the context layout
#include "fokl_comm.inc"
#include "fokl_chain_device.inc"
#include "fokl_spectral_device.inc"
#include "fokl_predict.inc"
#include "fokl_probe.inc"
#include "fokl_dgemm_device.inc"

#if defined(FOKL_GT_STAMP) || defined(FOKL_GD_STAMP)
// diagnostic builds only (tools/k2_clock.sh, tools/k2_phases.sh): the clock stamps of the last Gram launch
extern "C" int fokl_debug_stamps_read(unsigned long long *out, int count)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fokl::fokl_debug_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : -1;
}

extern "C" int fokl_debug_stamps_clear()
{
    static const unsigned long long zeros[8192] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(fokl::fokl_debug_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
}
#endif
