// nrs_boundary.hip — Akinci boundary volumes on the device.
//
// The reference obtains the per-boundary-particle volumes from its un-vendored submodule
// (sample_spheres::boundary_forces::getVbi, call site main.cpp:546); that library is not in the container, so this is our own
// implementation behind the same host signature (nereus_amd/host/sph_boundary_particles/boundary_forces.h) — PARITY UNPINNED at
// this boundary, by nature.  Definition (Akinci et al. 2012, eq. 4): Vb_i = 1 / sum_k W_poly6(|x_i - x_k|, h) over all boundary
// particles k with |x_i - x_k| < h, i itself included.
//
// Same machinery as the solver's own neighbour search: grid hash -> rocPRIM radix sort -> cell ranges -> 27-cell gather, on a
// private grid (origin = AABB minimum, cell = h, one thread per sorted particle, sums in double).
#include "nrs_ctx_base.h"
#include <rocprim/rocprim.hpp>

namespace nrs {

template <typename R> struct BGrid { double ox, oy, oz, h; uint32_t gx, gy, gz; };

template <typename R> __device__ __forceinline__ void bcell(const BGrid<R> &g, double x, double y, double z, int &cx, int &cy, int &cz)
{
    cx = (int)floor((x - g.ox) / g.h); cy = (int)floor((y - g.oy) / g.h); cz = (int)floor((z - g.oz) / g.h);
    cx = min(max(cx, 0), (int)g.gx - 1); cy = min(max(cy, 0), (int)g.gy - 1); cz = min(max(cz, 0), (int)g.gz - 1);
}

template <typename R, typename T4>
__global__ __launch_bounds__(256) void k_bvol_hash(BGrid<R> g, const T4 *__restrict__ bi, uint32_t *__restrict__ key, uint32_t *__restrict__ val, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    bcell<R>(g, bi[i].x, bi[i].y, bi[i].z, cx, cy, cz);
    key[i] = ((uint32_t)cz * g.gy + (uint32_t)cy) * g.gx + (uint32_t)cx;
    val[i] = i;
}

static __global__ __launch_bounds__(256) void k_bvol_ranges(const uint32_t *__restrict__ key, uint32_t *__restrict__ cellStart, uint32_t *__restrict__ cellEnd, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = key[i];
    if (i == 0 || k != key[i - 1]) { cellStart[k] = i; if (i) cellEnd[key[i - 1]] = i; }
    if (i == n - 1) cellEnd[k] = n;
}

template <typename R, typename T4>
__global__ __launch_bounds__(256) void k_bvol_gather(BGrid<R> g, double kpoly, const T4 *__restrict__ bi, const uint32_t *__restrict__ val,
                                                     const uint32_t *__restrict__ cellStart, const uint32_t *__restrict__ cellEnd,
                                                     R *__restrict__ vbi, uint32_t n)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n) return;
    const uint32_t i = val[s];
    const double x = bi[i].x, y = bi[i].y, z = bi[i].z, h2 = g.h * g.h;
    int cx, cy, cz;
    bcell<R>(g, x, y, z, cx, cy, cz);
    double acc = 0.0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int qx = cx + dx, qy = cy + dy, qz = cz + dz;
                if (qx < 0 || qy < 0 || qz < 0 || qx >= (int)g.gx || qy >= (int)g.gy || qz >= (int)g.gz) continue;
                const uint32_t c = ((uint32_t)qz * g.gy + (uint32_t)qy) * g.gx + (uint32_t)qx;
                const uint32_t a = cellStart[c];
                if (a == 0xffffffffu) continue;
                const uint32_t b = cellEnd[c];
                for (uint32_t t = a; t < b; ++t) {
                    const T4 o = bi[val[t]];
                    const double rx = x - o.x, ry = y - o.y, rz = z - o.z, r2 = rx * rx + ry * ry + rz * rz;
                    if (r2 < h2) { const double w = h2 - r2; acc += kpoly * w * w * w; }
                }
            }
    vbi[i] = (R)(1.0 / acc);
}

template <typename R, typename T4> static int boundary_volumes(const void *bi4, uint64_t nb, double h, void *out)
{
    const T4 *hb = (const T4 *)bi4;
    double lo[3] = {hb[0].x, hb[0].y, hb[0].z}, hi[3] = {hb[0].x, hb[0].y, hb[0].z};
    for (uint64_t i = 0; i < nb; ++i) {
        const double c[3] = {hb[i].x, hb[i].y, hb[i].z};
        // (a NaN / inf coordinate would make the cell index of that particle undefined behaviour, here and on the device)
        if (!(std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]))) return fail(NRS_E_INVALID, "boundary particle with a non-finite coordinate");
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
    }
    BGrid<R> g;
    g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2]; g.h = h;
    uint64_t dims[3];
    for (int a = 0; a < 3; ++a) {
        const double d = std::floor((hi[a] - lo[a]) / h) + 1.0;
        if (!(d >= 1.0 && d <= 2147483648.0)) return fail(NRS_E_INVALID, "boundary AABB spans more than 2^31 cells of size h along one axis");
        dims[a] = (uint64_t)d;
    }
    // every factor is <= 2^31: the first product cannot wrap, the second is checked before it is formed
    if (dims[0] * dims[1] > (1ull << 31) || dims[0] * dims[1] * dims[2] > (1ull << 31))
        return fail(NRS_E_INVALID, "boundary AABB spans more than 2^31 cells of size h");
    const uint64_t cells = dims[0] * dims[1] * dims[2];
    g.gx = (uint32_t)dims[0]; g.gy = (uint32_t)dims[1]; g.gz = (uint32_t)dims[2];
    const double kpoly = 315.0 / (64.0 * 3.14159265358979323846 * std::pow(h, 9));
    const uint32_t n = (uint32_t)nb, nbk = (n + 255u) / 256u;
    DevBuf dBi, dKey, dKey2, dVal, dVal2, dStart, dEnd, dOut, dTmp;
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto done = [&](int rc) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
        DevBuf *all[] = {&dBi, &dKey, &dKey2, &dVal, &dVal2, &dStart, &dEnd, &dOut, &dTmp};
        for (DevBuf *b : all) b->release();
        return rc;
    };
#define BCHK(expr) do { int r_ = (expr); if (r_ != NRS_OK) return done(r_); } while (0)
#define BHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(NRS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); } while (0)
    BCHK(dBi.alloc(sizeof(T4) * nb)); BCHK(dKey.alloc(4 * nb)); BCHK(dKey2.alloc(4 * nb)); BCHK(dVal.alloc(4 * nb)); BCHK(dVal2.alloc(4 * nb));
    BCHK(dStart.alloc(4 * cells)); BCHK(dEnd.alloc(4 * cells)); BCHK(dOut.alloc(sizeof(R) * nb));
    BHIP(hipMemcpyAsync(dBi.p, bi4, sizeof(T4) * nb, hipMemcpyHostToDevice, st));
    BHIP(hipMemsetAsync(dStart.p, 0xff, 4 * cells, st));
    hipLaunchKernelGGL((k_bvol_hash<R, T4>), dim3(nbk), dim3(256), 0, st, g, dBi.as<T4>(), dKey.as<uint32_t>(), dVal.as<uint32_t>(), n);
    unsigned bits = 1;
    while (bits < 32 && (1ull << bits) < cells) ++bits;
    rocprim::double_buffer<uint32_t> k(dKey.as<uint32_t>(), dKey2.as<uint32_t>()), v(dVal.as<uint32_t>(), dVal2.as<uint32_t>());
    size_t tmp = 0;
    BHIP(rocprim::radix_sort_pairs(nullptr, tmp, k, v, (size_t)n, 0u, bits, st));
    BCHK(dTmp.alloc(tmp));
    BHIP(rocprim::radix_sort_pairs(dTmp.p, tmp, k, v, (size_t)n, 0u, bits, st));
    hipLaunchKernelGGL(k_bvol_ranges, dim3(nbk), dim3(256), 0, st, k.current(), dStart.as<uint32_t>(), dEnd.as<uint32_t>(), n);
    hipLaunchKernelGGL((k_bvol_gather<R, T4>), dim3(nbk), dim3(256), 0, st, g, kpoly, dBi.as<T4>(), v.current(), dStart.as<uint32_t>(),
                       dEnd.as<uint32_t>(), dOut.as<R>(), n);
    BHIP(hipGetLastError());
    BHIP(hipMemcpyAsync(out, dOut.p, sizeof(R) * nb, hipMemcpyDeviceToHost, st));
    BHIP(hipStreamSynchronize(st));
#undef BCHK
#undef BHIP
    return done(NRS_OK);
}

} // namespace nrs

using namespace nrs;

extern "C" int nrs_boundary_volumes(int device, int precision, const void *bi4, uint64_t nb, double h, void *vbi)
{
    if (precision != 32 && precision != 64) return fail(NRS_E_INVALID, "precision must be 32 or 64");
    if (nb == 0) return NRS_OK;
    if (!bi4 || !vbi || !(h > 0.0)) return fail(NRS_E_INVALID, "bad argument");
    if (nb >= (1ull << 31)) return fail(NRS_E_INVALID, "too many boundary particles");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NRS_E_NODEVICE, "no HIP device available: libnereus_hip has no CPU fallback");
    if (device >= ndev) return fail(NRS_E_INVALID, "device ordinal out of range");
    DeviceScope scope(device); // the caller's current device is restored on return
    if (!scope.ok()) return fail(NRS_E_HIP, "hipSetDevice failed");
    return precision == 32 ? boundary_volumes<float, float4>(bi4, nb, h, vbi) : boundary_volumes<double, double4>(bi4, nb, h, vbi);
}
