// nrs_debug.hip — test hook: the DEVICE smoothing kernels and vector helpers (nrs_math.h) evaluated on caller-supplied
// separations, so that tests can compare the product's arithmetic — not only the oracle's — with the reference's own
// common/kernels_impl.cuh + helper_math.h compiled unmodified (oracle/_ref; fixture tests/golden/ref_kernels_pin.npz).
// Same function numbering as oracle/ref_kernels_driver.cpp; Cakinci / Aboundary (6, 7) do not exist on the device (the
// reference never calls them).
#include "nrs_ctx_base.h"
#include "nrs_math.h"

namespace nrs {

template <typename R>
__global__ void k_eval_smoothing(int which, uint32_t n, const R *__restrict__ r3, const R *__restrict__ s3, R h, R c0, R c1, R *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3<R> r = mk3<R>(r3[3 * i], r3[3 * i + 1], r3[3 * i + 2]);
    const V3<R> s = s3 ? mk3<R>(s3[3 * i], s3[3 * i + 1], s3[3 * i + 2]) : mk3<R>(0, 0, 0);
    V3<R> v = mk3<R>(0, 0, 0);
    switch (which) {
    case 0: v.x = Wdefault<R>(r, h, c0); break;
    case 1: v = Wdefault_grad<R>(r, h, c0); break;
    case 2: v = Wpressure_grad<R>(r, h, c0); break;
    case 3: v = Wviscosity_grad<R>(r, h, c0, c1); break;
    case 4: v.x = Wmonaghan<R>(r, h); break;
    case 5: v = Wmonaghan_grad<R>(r, h); break;
    case 8: v.x = dot(r, s); break;
    case 9: v.x = length(r); break;
    case 10: v = r * (float)c0; break;
    case 11: v = (float)c0 * r; break;
    case 12: v = r / (float)c0; break;
    case 13: v = r; break;
    case 14: v = r + s; break;
    case 15: v = r - s; break;
    default: break;
    }
    out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
}

template <typename R> static int eval_smoothing(int which, uint64_t n, const void *r3, const void *s3, double h, double c0, double c1, void *out)
{
    DevBuf dr, ds, dout;
    const size_t bytes = sizeof(R) * 3 * n;
    auto done = [&](int rc) { dr.release(); ds.release(); dout.release(); return rc; };
    int rc = dr.alloc(bytes); // (every way out goes through done(): a failed second allocation must not leak the first)
    if (rc == NRS_OK) rc = dout.alloc(bytes);
    if (rc == NRS_OK && s3) rc = ds.alloc(bytes);
    if (rc != NRS_OK) return done(rc);
    if (hipMemcpy(dr.p, r3, bytes, hipMemcpyHostToDevice) != hipSuccess) return done(fail(NRS_E_HIP, "hipMemcpy"));
    if (s3 && hipMemcpy(ds.p, s3, bytes, hipMemcpyHostToDevice) != hipSuccess) return done(fail(NRS_E_HIP, "hipMemcpy"));
    hipLaunchKernelGGL((k_eval_smoothing<R>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, nullptr, which, (uint32_t)n, dr.as<R>(),
                       s3 ? ds.as<R>() : (const R *)nullptr, (R)h, (R)c0, (R)c1, dout.as<R>());
    if (hipDeviceSynchronize() != hipSuccess) return done(fail(NRS_E_HIP, "k_eval_smoothing failed"));
    if (hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return done(fail(NRS_E_HIP, "hipMemcpy"));
    return done(NRS_OK);
}

} // namespace nrs

using namespace nrs;

extern "C" int nrs_eval_smoothing(int precision, int which, uint64_t n, const void *r3, const void *s3, double h, double c0, double c1, void *out)
{
    if (precision != 32 && precision != 64) return fail(NRS_E_INVALID, "precision must be 32 or 64");
    if (which < 0 || which > 15 || which == 6 || which == 7) return fail(NRS_E_INVALID, "no such device function");
    if (!n) return NRS_OK;
    if (!r3 || !out || n > (1ull << 30)) return fail(NRS_E_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NRS_E_NODEVICE, "no HIP device available: libnereus_hip has no CPU fallback");
    return precision == 32 ? eval_smoothing<float>(which, n, r3, s3, h, c0, c1, out) : eval_smoothing<double>(which, n, r3, s3, h, c0, c1, out);
}
