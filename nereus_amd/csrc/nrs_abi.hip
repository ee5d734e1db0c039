// nrs_abi.hip — the C ABI of libnereus_hip.so (include/nereus_hip.h): thin entry points over nrs::CtxBase.
// The contexts themselves are compiled per (precision, kernel set) in nrs_inst_*.hip.
#include "nrs_ctx_base.h"

#include <condition_variable>
#include <mutex>
#include <thread>

namespace nrs {

thread_local std::string g_err;

enum { KS_MONAGHAN_ = 0, KS_MULLER_ = 1 }; // = KS_MONAGHAN / KS_MULLER of nrs_math.h
extern template CtxBase *make_ctx2<float, KS_MULLER_>(bool);
extern template CtxBase *make_ctx2<float, KS_MONAGHAN_>(bool);
extern template CtxBase *make_ctx2<double, KS_MULLER_>(bool);
extern template CtxBase *make_ctx2<double, KS_MONAGHAN_>(bool);

static CtxBase *make_ctx(const nrs_config &c)
{
    if (c.precision == 32) return c.kernel_set == NRS_KERNELS_MULLER ? make_ctx2<float, KS_MULLER_>(c.surface_tension != 0) : make_ctx2<float, KS_MONAGHAN_>(c.surface_tension != 0);
    return c.kernel_set == NRS_KERNELS_MULLER ? make_ctx2<double, KS_MULLER_>(c.surface_tension != 0) : make_ctx2<double, KS_MONAGHAN_>(c.surface_tension != 0);
}

} // namespace nrs

using namespace nrs;

// nrs_step is asynchronous for the CALLER: a call with two or more steps hands them to a thread the context owns and returns at
// once (the step loop itself has to read one number back per step — the mover count that sizes the coherent re-sort's rocPRIM calls,
// or the IISPH solver's density error — so the thread that enqueues it is never more than about a step ahead of the device; that
// thread is the context's, not the caller's).  Every other entry point first waits until the queued steps have been enqueued
// (drain) and reports the error a queued step ran into.  Single steps run on the calling thread: a caller that alternates
// nrs_step(ctx, 1) with other calls (the slab exchange, a viewer that reads every frame) would only pay the hand-off.
struct nrs_ctx {
    CtxBase *impl = nullptr;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    int pending = 0;      // steps handed over, not yet taken by the worker
    bool busy = false;    // the worker is inside impl->step
    bool quit = false;
    int err = NRS_OK;     // first error of a queued step, reported by the next call
    std::string errMsg;

    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return pending > 0 || quit; });
            if (quit && pending == 0) return;
            const int k = pending;
            pending = 0;
            busy = true;
            lk.unlock();
            int rc = NRS_OK;
            if (hipSetDevice(impl->device) != hipSuccess) rc = fail(NRS_E_HIP, "hipSetDevice failed (step worker)");
            else rc = impl->step(k, 0);
            const std::string msg = rc != NRS_OK ? g_err : std::string();
            lk.lock();
            busy = false;
            if (rc != NRS_OK && err == NRS_OK) { err = rc; errMsg = msg; pending = 0; } // (steps queued behind a failed one are dropped)
            cv.notify_all();
        }
    }
    // wait for the worker to go idle; returns (and clears) the error of a queued step
    int drain()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return pending == 0 && !busy; });
        const int rc = err;
        if (rc != NRS_OK) { g_err = errMsg; err = NRS_OK; errMsg.clear(); }
        return rc;
    }
    void wait_idle() // (for the entry points that cannot report an error: a queued step's error stays for the next call that can)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return pending == 0 && !busy; });
    }
    // for the entry points that cannot report an error: run the deferred bookkeeping, park its error for the next call that can
    void settle_keep_error()
    {
        const int rc = impl->settle();
        if (rc == NRS_OK) return;
        std::lock_guard<std::mutex> lk(m);
        if (err == NRS_OK) { err = rc; errMsg = g_err; }
    }
    bool failed()
    {
        std::lock_guard<std::mutex> lk(m);
        return err != NRS_OK;
    }
    bool idle()
    {
        std::lock_guard<std::mutex> lk(m);
        return pending == 0 && !busy;
    }
    void submit(int k)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            if (!worker.joinable()) worker = std::thread([this] { run(); });
            pending += k;
        }
        cv.notify_all();
    }
    ~nrs_ctx()
    {
        if (worker.joinable()) {
            { std::lock_guard<std::mutex> lk(m); quit = true; }
            cv.notify_all();
            worker.join();
        }
        delete impl;
    }
};

#define CTX_GUARD(ctx)                                                   \
    if (!(ctx) || !(ctx)->impl) return fail(NRS_E_INVALID, "NULL context"); \
    if (hipSetDevice((ctx)->impl->device) != hipSuccess) return fail(NRS_E_HIP, "hipSetDevice failed"); \
    { const int drc_ = (ctx)->drain(); if (drc_ != NRS_OK) return drc_; } \
    { const int src_ = (ctx)->impl->settle(); if (src_ != NRS_OK) return src_; }
// (the two entry points of the exchange itself: nrs_slab_pack queues, nrs_slab_unpack settles together with its own wait)
#define CTX_GUARD_EXCHANGE(ctx)                                          \
    if (!(ctx) || !(ctx)->impl) return fail(NRS_E_INVALID, "NULL context"); \
    if (hipSetDevice((ctx)->impl->device) != hipSuccess) return fail(NRS_E_HIP, "hipSetDevice failed"); \
    { const int drc_ = (ctx)->drain(); if (drc_ != NRS_OK) return drc_; }

extern "C" {

const char *nrs_last_error(void) { return g_err.c_str(); }
uint32_t nrs_version(void) { return (0u << 16) | 2u; } // 0.2: NRS_FLAG_STAGED_SCAN / NO_WALL_WORKGROUPS, nrs_slab_last_counts, asynchronous nrs_step
int nrs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int nrs_create(const nrs_config *cfg, const void *params, nrs_ctx **out)
{
    if (!cfg || !params || !out) return fail(NRS_E_INVALID, "NULL argument");
    if (cfg->struct_size != sizeof(nrs_config)) return fail(NRS_E_INVALID, "nrs_config.struct_size mismatch");
    if (cfg->precision != 32 && cfg->precision != 64) return fail(NRS_E_INVALID, "precision must be 32 or 64");
    if (cfg->kernel_set != NRS_KERNELS_MULLER && cfg->kernel_set != NRS_KERNELS_MONAGHAN) return fail(NRS_E_INVALID, "bad kernel_set");
    if (cfg->solver != NRS_SOLVER_SESPH && cfg->solver != NRS_SOLVER_IISPH) return fail(NRS_E_INVALID, "bad solver");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NRS_E_NODEVICE, "no HIP device available: libnereus_hip has no CPU fallback");
    int dev = cfg->device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    if (dev >= ndev) return fail(NRS_E_INVALID, "device ordinal out of range");
    HIPCHK(hipSetDevice(dev));
    CtxBase *c = make_ctx(*cfg);
    c->device = dev;
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        hipStream_t s;
        hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(NRS_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
        c->stream = s;
        c->ownStream = true;
    }
    int r = c->init(*cfg, params);
    if (r != NRS_OK) { delete c; return r; }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { delete c; return fail(NRS_E_HIP, std::string("init sync: ") + hipGetErrorString(e)); }
    nrs_ctx *h = new nrs_ctx();
    h->impl = c;
    *out = h;
    return NRS_OK;
}
int nrs_destroy(nrs_ctx *ctx)
{
    if (!ctx) return NRS_OK;
    delete ctx; // (joins the step worker, then deletes the context)
    return NRS_OK;
}
int nrs_set_params(nrs_ctx *ctx, const void *params)
{
    CTX_GUARD(ctx);
    if (!params) return fail(NRS_E_INVALID, "NULL params");
    return ctx->impl->set_params(params);
}
int nrs_get_params(nrs_ctx *ctx, void *params)
{
    CTX_GUARD(ctx);
    if (!params) return fail(NRS_E_INVALID, "NULL params");
    return ctx->impl->get_params(params);
}
int nrs_upload_particles(nrs_ctx *ctx, const void *pos4, const void *vel4, const void *pres, uint64_t first, uint64_t count)
{
    CTX_GUARD(ctx);
    return ctx->impl->upload(pos4, vel4, pres, first, count);
}
int nrs_set_num_particles(nrs_ctx *ctx, uint64_t n)
{
    CTX_GUARD(ctx);
    return ctx->impl->set_n(n);
}
uint64_t nrs_num_particles(nrs_ctx *ctx)
{
    if (!ctx || !ctx->impl) return 0;
    ctx->wait_idle();
    ctx->settle_keep_error();
    return ctx->impl->get_n();
}
int nrs_set_boundaries(nrs_ctx *ctx, const void *bi4, const void *vbi, uint64_t nb, int update_grid)
{
    CTX_GUARD(ctx);
    return ctx->impl->set_boundaries(bi4, vbi, nb, update_grid);
}
int nrs_step(nrs_ctx *ctx, int nsteps)
{
    if (!ctx || !ctx->impl) return fail(NRS_E_INVALID, "NULL context");
    if (nsteps < 0) return fail(NRS_E_INVALID, "nsteps < 0");
    if (hipSetDevice(ctx->impl->device) != hipSuccess) return fail(NRS_E_HIP, "hipSetDevice failed");
    if (ctx->idle()) {
        // nothing in flight: refuse a call the context's state does not allow NOW, as a synchronous call would (mid-update after a
        // partial step, a host-driven IISPH step in progress, inconsistent state); single steps run right here
        { const int drc = ctx->drain(); if (drc != NRS_OK) return drc; }
        NRSCHK(ctx->impl->settle());
        if (nsteps <= 1) return ctx->impl->step(nsteps, 0);
        NRSCHK(ctx->impl->step(0, 0));
    }
    // (behind steps that are still being enqueued: their checks cover these too; if one of them has already failed, report that now
    // instead of queueing more work on a state the failed step left behind)
    if (ctx->failed()) return ctx->drain();
    ctx->submit(nsteps);
    return NRS_OK;
}
int nrs_step_partial(nrs_ctx *ctx, int stop_stage)
{
    CTX_GUARD(ctx);
    if (stop_stage <= 0 || stop_stage >= NRS_STAGE_COUNT) return fail(NRS_E_INVALID, "bad stop_stage");
    return ctx->impl->step(1, stop_stage);
}
int nrs_synchronize(nrs_ctx *ctx)
{
    CTX_GUARD(ctx);
    return ctx->impl->sync();
}
int nrs_snapshot_begin(nrs_ctx *ctx, int with_vel)
{
    CTX_GUARD(ctx);
    return ctx->impl->snapshot_begin(with_vel);
}
int nrs_snapshot_wait(nrs_ctx *ctx, int block, const void **pos4, const void **vel4, uint64_t *n, uint64_t *step)
{
    CTX_GUARD(ctx);
    return ctx->impl->snapshot_wait(block, pos4, vel4, n, step);
}
int nrs_download(nrs_ctx *ctx, void *pos4, void *vel4, void *pres)
{
    CTX_GUARD(ctx);
    return ctx->impl->download(pos4, vel4, pres);
}
int nrs_get_array(nrs_ctx *ctx, int which, void *dst, uint64_t dst_bytes, uint64_t *out_bytes)
{
    CTX_GUARD(ctx);
    void *p = nullptr;
    uint64_t sz = 0;
    NRSCHK(ctx->impl->array(which, &p, &sz));
    if (out_bytes) *out_bytes = sz;
    if (!dst) return NRS_OK;
    if (dst_bytes < sz) return fail(NRS_E_INVALID, "destination too small");
    if (sz) {
        HIPCHK(hipMemcpyAsync(dst, p, sz, hipMemcpyDeviceToHost, ctx->impl->stream));
        HIPCHK(hipStreamSynchronize(ctx->impl->stream));
    }
    return NRS_OK;
}
int nrs_device_ptr(nrs_ctx *ctx, int which, void **dptr, uint64_t *bytes)
{
    CTX_GUARD(ctx);
    if (!dptr || !bytes) return fail(NRS_E_INVALID, "NULL argument");
    return ctx->impl->array(which, dptr, bytes);
}
int nrs_last_iterations(nrs_ctx *ctx, uint32_t *iters)
{
    CTX_GUARD(ctx);
    if (!iters) return fail(NRS_E_INVALID, "NULL argument");
    *iters = ctx->impl->lastIters;
    return NRS_OK;
}
int nrs_set_max_iterations(nrs_ctx *ctx, uint32_t max_iters)
{
    CTX_GUARD(ctx);
    ctx->impl->maxIters = max_iters;
    return NRS_OK;
}
int nrs_set_profiling(nrs_ctx *ctx, uint32_t stage_mask)
{
    CTX_GUARD(ctx);
    return ctx->impl->set_profiling(stage_mask);
}
int nrs_stage_ms(nrs_ctx *ctx, int stage, float *ms, uint32_t *launches)
{
    CTX_GUARD(ctx);
    if (!ms) return fail(NRS_E_INVALID, "NULL argument");
    return ctx->impl->stage_ms(stage, ms, launches);
}
int nrs_slab_configure(nrs_ctx *ctx, int32_t cell_lo, int32_t cell_hi, int32_t halo_cells)
{
    CTX_GUARD(ctx);
    return ctx->impl->slab_configure(cell_lo, cell_hi, halo_cells);
}
int nrs_slab_pack(nrs_ctx *ctx, void *send_left, void *send_right, uint64_t capacity, uint32_t counts[6])
{
    CTX_GUARD_EXCHANGE(ctx);
    return ctx->impl->slab_pack(send_left, send_right, capacity, counts);
}
int nrs_slab_unpack(nrs_ctx *ctx, const void *recv_left, const void *recv_right, uint64_t capacity)
{
    CTX_GUARD_EXCHANGE(ctx);
    return ctx->impl->slab_unpack(recv_left, recv_right, capacity);
}
int nrs_slab_last_counts(nrs_ctx *ctx, uint32_t counts[6])
{
    CTX_GUARD(ctx);
    if (!counts) return fail(NRS_E_INVALID, "NULL argument");
    return ctx->impl->slab_last_counts(counts);
}
uint64_t nrs_num_owned(nrs_ctx *ctx)
{
    if (!ctx || !ctx->impl) return 0;
    ctx->wait_idle();
    ctx->settle_keep_error();
    return ctx->impl->num_owned();
}
int nrs_slab_histogram(nrs_ctx *ctx, int32_t first_cell, uint32_t ncells, uint32_t *counts)
{
    CTX_GUARD(ctx);
    return ctx->impl->slab_histogram(first_cell, ncells, counts);
}
int nrs_resort_stats(nrs_ctx *ctx, uint64_t *steps, uint64_t *fallbacks)
{
    CTX_GUARD(ctx);
    ctx->impl->resort_stats(steps, fallbacks);
    return NRS_OK;
}
int nrs_iisph_predict(nrs_ctx *ctx)
{
    CTX_GUARD(ctx);
    return ctx->impl->iisph_phase(0, nullptr, nullptr);
}
int nrs_iisph_iterate(nrs_ctx *ctx, double *sum_density_corr, uint64_t *count)
{
    CTX_GUARD(ctx);
    return ctx->impl->iisph_phase(1, sum_density_corr, count);
}
int nrs_iisph_finish(nrs_ctx *ctx)
{
    CTX_GUARD(ctx);
    return ctx->impl->iisph_phase(2, nullptr, nullptr);
}
int nrs_get_stat(nrs_ctx *ctx, int which, double *out)
{
    CTX_GUARD(ctx);
    if (!out) return fail(NRS_E_INVALID, "NULL argument");
    return ctx->impl->get_stat(which, out);
}
uint64_t nrs_slab_message_bytes(uint64_t capacity, int precision) { return 16 + capacity * 2 * (precision == 64 ? 32 : 16); }

int nrs_max_density(nrs_ctx *ctx, double *out)
{
    CTX_GUARD(ctx);
    if (!out) return fail(NRS_E_INVALID, "NULL argument");
    return ctx->impl->reduce_max(0, out);
}
int nrs_max_velocity(nrs_ctx *ctx, double *out)
{
    CTX_GUARD(ctx);
    if (!out) return fail(NRS_E_INVALID, "NULL argument");
    return ctx->impl->reduce_max(1, out);
}

} // extern "C"
