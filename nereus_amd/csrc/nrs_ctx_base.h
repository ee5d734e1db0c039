// nrs_ctx_base.h — what the C ABI translation unit (nrs_abi.hip) and the per-variant context translation units
// (nrs_inst_*.hip) share: error plumbing, device buffers, the abstract context.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/nereus_hip.h"

namespace nrs {

extern thread_local std::string g_err; // defined in nrs_abi.hip
static inline int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(NRS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                       std::to_string(__LINE__) + ")");                                  \
    } while (0)
#define NRSCHK(expr)              \
    do {                          \
        int r_ = (expr);          \
        if (r_ != NRS_OK) return r_; \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n)
    {
        if (n <= bytes && p) return NRS_OK;
        release();
        if (n == 0) return NRS_OK;
        HIPCHK(hipMalloc(&p, n));
        bytes = n;
        return NRS_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return (T *)p; }
};

// hipSetDevice for the duration of a call that is not bound to a context's own device bookkeeping (nrs_boundary_volumes,
// nrs_eval_smoothing): the caller's current device is put back on every way out.  device < 0 = stay on the current device.
struct DeviceScope {
    int prev = -1;
    bool good = true;
    explicit DeviceScope(int device)
    {
        if (device < 0) return;
        good = hipGetDevice(&prev) == hipSuccess && hipSetDevice(device) == hipSuccess;
        if (!good) prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    bool ok() const { return good; }
};

struct CtxBase {
    virtual ~CtxBase() {}
    virtual int init(const nrs_config &cfg, const void *params) = 0;
    virtual int set_params(const void *params) = 0;
    virtual int get_params(void *params) = 0;
    virtual int upload(const void *pos4, const void *vel4, const void *pres, uint64_t first, uint64_t count) = 0;
    virtual int set_n(uint64_t n) = 0;
    virtual uint64_t get_n() = 0;
    virtual int set_boundaries(const void *bi4, const void *vbi, uint64_t nb, int update_grid) = 0;
    virtual int step(int nsteps, int stop) = 0;
    virtual int sync() = 0;
    virtual int download(void *pos4, void *vel4, void *pres) = 0;
    virtual int snapshot_begin(int withVel) = 0;
    virtual int snapshot_wait(int block, const void **pos4, const void **vel4, uint64_t *n, uint64_t *step) = 0;
    virtual int array(int which, void **dptr, uint64_t *bytes) = 0;
    virtual int stage_ms(int stage, float *ms, uint32_t *launches) = 0;
    virtual int reduce_max(int which, double *out) = 0;
    virtual int slab_configure(int lo, int hi, int halo) = 0;
    virtual int slab_pack(void *sendL, void *sendR, uint64_t cap, uint32_t *counts) = 0;
    virtual int slab_unpack(const void *recvL, const void *recvR, uint64_t cap) = 0;
    virtual uint64_t num_owned() = 0;
    virtual int slab_histogram(int lo0, uint32_t nbins, uint32_t *out) = 0;
    virtual void resort_stats(uint64_t *steps, uint64_t *fallbacks) = 0;
    virtual int get_stat(int which, double *out) = 0;
    virtual int iisph_phase(int phase, double *sum, uint64_t *count) = 0;
    virtual int settle() = 0;                           // finish host bookkeeping a previous call deferred (nrs_slab_pack's totals)
    virtual int slab_last_counts(uint32_t *counts) = 0; // stream populations of the last nrs_slab_pack
    virtual int set_profiling(uint32_t mask) = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    int device = 0;
    uint32_t lastIters = 0, maxIters = 0;
    uint32_t profMask = 0;
};

// one context class per (precision, kernel set), each compiled in its own translation unit (nrs_inst_*.hip)
template <typename R, int KSET> CtxBase *make_ctx2(bool surf);

} // namespace nrs
