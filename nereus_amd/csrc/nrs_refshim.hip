// nrs_refshim.hip — libnereus_refshim.so: the reference's launcher names (sph/sph.cuh:19-230, sph/sph_cuda.cu) as thin launches of the
// reference-order gfx950 kernels (nrs_kernels_ref.h) on caller-owned HIP device pointers.  Contract, scope and the documented
// differences from the reference's defects: include/nereus_refshim.h.  Built once per reference configuration (DOUBLE_PRECISION x KERNEL_SET), see csrc/Makefile.
#include "nrs_ctx_base.h"
#include <rocprim/rocprim.hpp>

#include "nrs_kernels_ref.h"
#include <climits>
#include <cstdlib>

#include "../../include/nereus_refshim.h"

namespace nrs {

thread_local std::string g_err; // (this library has its own copy of the error plumbing: it does not link libnereus_hip.so)

namespace {

// the reference's compile-time switches (CMakeLists.txt:25-28, common/common.h:14-43), one shim library per combination:
// libnereus_refshim[_f64][_monaghan].so
#ifndef DOUBLE_PRECISION
#define DOUBLE_PRECISION 0
#endif
#ifndef KERNEL_SET
#define KERNEL_SET 1
#endif
#ifndef USE_SURFACE_TENSION
#define USE_SURFACE_TENSION 1
#endif
#if DOUBLE_PRECISION
typedef double SR;
#else
typedef float SR;
#endif
typedef Vec4T<SR>::type T4;
constexpr int KSET = KERNEL_SET ? KS_MULLER : KS_MONAGHAN;
constexpr bool SURF = USE_SURFACE_TENSION != 0;

Params<SR> g_params;          // the reference's `__constant__ SphSimParams sph_params` (sph_kernel_impl.cuh:66)
bool g_paramsSet = false;
uint32_t g_lastIters = 0;
DevBuf g_tmpKeys, g_tmpVals, g_sortTmp, g_sB, g_inv, g_partial, g_out;

[[noreturn]] void die(const char *what, hipError_t e)
{
    // checkCudaErrors / getLastCudaError of the reference: message, then exit(EXIT_FAILURE)
    fprintf(stderr, "libnereus_refshim: %s: %s\n", what, e == hipSuccess ? g_err.c_str() : hipGetErrorString(e));
    exit(EXIT_FAILURE);
}
#define SHIM_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) die(#expr, e_); } while (0)
#define SHIM_NRS(expr) do { if ((expr) != NRS_OK) die(#expr, hipSuccess); } while (0)

void need_params(const char *who)
{
    if (!g_paramsSet) { g_err = "setParameters has not been called"; die(who, hipSuccess); }
}
dim3 blocks(uint32_t n) { return dim3((n + BLOCK - 1) / BLOCK); }

GridView<SR> grid_view(const SUint_t *cellStart, const SUint_t *cellEnd, const SUint_t *bStart, const SUint_t *bEnd, const T4 *sB, uint32_t n)
{
    GridView<SR> G;
    G.cellStart = cellStart; G.cellEnd = cellEnd; G.bCellStart = bStart; G.bCellEnd = bEnd; G.sB = sB;
    G.actLo = INT_MIN; G.actHi = INT_MAX;
    G.nSorted = n;
    G.err = nullptr;
    G.qpos = nullptr; G.qT = 0u;
    G.qc = QuantCfg();
    return G;
}

// boundary particles as the kernels of this build read them: one vec4 per SORTED boundary slot, xyz + Vbi in w
// INDEXED: slot j is particle index[j] of the unsorted arrays (what the SESPH kernels index, sph_kernel_impl.cuh:341-344,568-572);
// otherwise slot j of arrays that are already sorted (the IISPH helpers, :747-748)
template <bool INDEXED>
__global__ __launch_bounds__(BLOCK) void k_pack_boundary(const T4 *__restrict__ pos, const SR *__restrict__ vbi, const uint32_t *__restrict__ index,
                                                         T4 *__restrict__ sB, uint32_t nb)
{
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= nb) return;
    const uint32_t o = INDEXED ? index[j] : j;
    const T4 p = pos[o];
    sB[j] = mk4<SR>(p.x, p.y, p.z, vbi[o]);
}
const T4 *pack_boundary(bool indexed, const SR *pos, const SR *vbi, const SUint_t *index, uint32_t nb)
{
    if (!nb) return nullptr;
    SHIM_NRS(g_sB.alloc(sizeof(T4) * (size_t)nb));
    if (indexed) hipLaunchKernelGGL((k_pack_boundary<true>), blocks(nb), dim3(BLOCK), 0, nullptr, (const T4 *)pos, vbi, index, g_sB.as<T4>(), nb);
    else hipLaunchKernelGGL((k_pack_boundary<false>), blocks(nb), dim3(BLOCK), 0, nullptr, (const T4 *)pos, vbi, index, g_sB.as<T4>(), nb);
    return g_sB.as<T4>();
}
__global__ __launch_bounds__(BLOCK) void k_unpack_boundary(const T4 *__restrict__ sB, T4 *__restrict__ sortedPos, SR *__restrict__ sortedVbi, uint32_t nb)
{
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= nb) return;
    const T4 b = sB[j];
    sortedPos[j] = mk4<SR>(b.x, b.y, b.z, (SR)1.0);
    sortedVbi[j] = b.w;
}
// inv[index[t]] = t: the slot -> reference-thread map the two IISPH kernels with the wrong self-exclusion need
__global__ __launch_bounds__(BLOCK) void k_invert(const uint32_t *__restrict__ index, uint32_t *__restrict__ inv, uint32_t n)
{
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < n) inv[index[t]] = t;
}
// component-wise extremes / the vector of largest length, one workgroup (setup-time reductions: BBMin/BBMax, maxVelocity)
__global__ __launch_bounds__(BLOCK) void k_vec_extremes(const T4 *__restrict__ a, uint32_t n, SR *__restrict__ out)
{
    // out[0..2] min xyz, out[3..5] max xyz, out[6..9] the vec4 of largest length (first one in array order among equals)
    __shared__ SR sm[BLOCK][6];
    __shared__ SR sl[BLOCK];
    __shared__ uint32_t si[BLOCK];
    SR mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, best = -1.0f;
    uint32_t bi = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < n; i += BLOCK) {
        const T4 v = a[i];
        const SR c[3] = {v.x, v.y, v.z};
        for (int k = 0; k < 3; ++k) { mn[k] = (SR)fmin((double)mn[k], (double)c[k]); mx[k] = (SR)fmax((double)mx[k], (double)c[k]); }
        const SR l = (SR)length(mk3<SR>(v.x, v.y, v.z)); // comp_length (sph_cuda.cu:20-26): length() of helper_math.h
        if (l > best) { best = l; bi = i; }
    }
    for (int k = 0; k < 3; ++k) { sm[threadIdx.x][k] = mn[k]; sm[threadIdx.x][3 + k] = mx[k]; }
    sl[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < BLOCK; ++t) {
            for (int k = 0; k < 3; ++k) { sm[0][k] = (SR)fmin((double)sm[0][k], (double)sm[t][k]); sm[0][3 + k] = (SR)fmax((double)sm[0][3 + k], (double)sm[t][3 + k]); }
            if (sl[t] > sl[0] || (sl[t] == sl[0] && si[t] < si[0])) { sl[0] = sl[t]; si[0] = si[t]; }
        }
        for (int k = 0; k < 6; ++k) out[k] = sm[0][k];
        const T4 v = si[0] != 0xffffffffu ? a[si[0]] : mk4<SR>((SR)0, (SR)0, (SR)0, (SR)0);
        out[6] = v.x; out[7] = v.y; out[8] = v.z; out[9] = v.w;
    }
}
void vec_extremes(const SR *a, uint32_t n, SR out[10])
{
    SHIM_NRS(g_out.alloc(16 * sizeof(double)));
    hipLaunchKernelGGL(k_vec_extremes, dim3(1), dim3(BLOCK), 0, nullptr, (const T4 *)a, n, g_out.as<SR>());
    SHIM_HIP(hipGetLastError());
    SHIM_HIP(hipMemcpy(out, g_out.p, 10 * sizeof(SR), hipMemcpyDeviceToHost));
}

IisphArrays<SR> iisph_view(SR *densAdv, SR *densCorr, SR *P_l, SR *prevP, SR *aii, SR *velAdv, SR *forcesAdv,
                              SR *forcesP, SR *diiF, SR *diiB, SR *sumDij)
{
    IisphArrays<SR> I;
    I.densAdv = densAdv; I.densCorr = densCorr; I.P_l = P_l; I.P_l_next = prevP; I.aii = aii;
    I.velAdv = (T4 *)velAdv; I.forcesAdv = (T4 *)forcesAdv; I.forcesP = (T4 *)forcesP;
    I.diiF = (T4 *)diiF; I.diiB = (T4 *)diiB; I.sumDij = (T4 *)sumDij;
    I.diiSum = nullptr; // (only the list-driven kernels of the context read it)
    I.inv = g_inv.as<uint32_t>();
    I.nonFinite = nullptr;
    return I;
}
void make_inv(const SUint_t *index, uint32_t n)
{
    SHIM_NRS(g_inv.alloc(4 * (size_t)n));
    hipLaunchKernelGGL(k_invert, blocks(n), dim3(BLOCK), 0, nullptr, index, g_inv.as<uint32_t>(), n);
}

} // namespace
} // namespace nrs

using namespace nrs;

extern "C" {

void allocateArray(void **devPtr, size_t size) { SHIM_HIP(hipMalloc(devPtr, size)); }
void freeArray(void *devPtr) { SHIM_HIP(hipFree(devPtr)); }
void threadSync(void) { SHIM_HIP(hipDeviceSynchronize()); }
void copyArrayToDevice(void *device, const void *host, int offset, int size)
{
    SHIM_HIP(hipMemcpy((char *)device + offset, host, (size_t)size, hipMemcpyHostToDevice));
}
void copyArrayFromDevice(void *host, const void *device, void **cuda_vbo_resource, int size)
{
    if (cuda_vbo_resource) { g_err = "GL interop is not part of this build: pass NULL for cuda_vbo_resource"; die("copyArrayFromDevice", hipSuccess); }
    SHIM_HIP(hipMemcpy(host, device, (size_t)size, hipMemcpyDeviceToHost));
}

void setParameters(void *hostParams)
{
    static_assert(sizeof(Params<SR>) == (DOUBLE_PRECISION ? sizeof(nrs_params_f64) : sizeof(nrs_params_f32)), "SphSimParams layout");
    std::memcpy(&g_params, hostParams, sizeof(g_params));
    g_params.numBodies = 0; // (this build's kernels read the unused numBodies field as the first column of a slab's cell-table window)
    g_paramsSet = true;
}

void integrateSystem(SR *pos, SR *vel, SR *forces, SR deltaTime, SUint_t numParticles)
{
    need_params("integrateSystem");
    if (!numParticles) return;
    Params<SR> P = g_params;
    P.timestep = deltaTime; // integrate_functor(deltaTime), sph_cuda.cu:224
    hipLaunchKernelGGL((k_integrate<SR>), blocks(numParticles), dim3(BLOCK), 0, nullptr, P, (T4 *)pos, (T4 *)vel, (const T4 *)forces, numParticles);
    SHIM_HIP(hipGetLastError());
}

void calcHash(SUint_t *gridParticleHash, SUint_t *gridParticleIndex, SR *pos, int numParticles)
{
    need_params("calcHash");
    if (numParticles <= 0) return;
    hipLaunchKernelGGL((k_hash<SR>), blocks((uint32_t)numParticles), dim3(BLOCK), 0, nullptr, g_params, (const T4 *)pos, gridParticleHash,
                       gridParticleIndex, (uint32_t)numParticles);
    SHIM_HIP(hipGetLastError());
}

void sortParticles(SUint_t *dGridParticleHash, SUint_t *dGridParticleIndex, SUint_t numParticles)
{
    if (numParticles < 2) return;
    // in place for the caller, as thrust::sort_by_key: rocPRIM sorts into a second pair of buffers owned by this library
    SHIM_NRS(g_tmpKeys.alloc(4 * (size_t)numParticles));
    SHIM_NRS(g_tmpVals.alloc(4 * (size_t)numParticles));
    rocprim::double_buffer<uint32_t> k(dGridParticleHash, g_tmpKeys.as<uint32_t>()), v(dGridParticleIndex, g_tmpVals.as<uint32_t>());
    size_t tmp = 0;
    SHIM_HIP(rocprim::radix_sort_pairs(nullptr, tmp, k, v, (size_t)numParticles, 0u, 32u, (hipStream_t) nullptr));
    SHIM_NRS(g_sortTmp.alloc(tmp));
    SHIM_HIP(rocprim::radix_sort_pairs(g_sortTmp.p, tmp, k, v, (size_t)numParticles, 0u, 32u, (hipStream_t) nullptr));
    if (k.current() != dGridParticleHash) SHIM_HIP(hipMemcpyAsync(dGridParticleHash, k.current(), 4 * (size_t)numParticles, hipMemcpyDeviceToDevice, nullptr));
    if (v.current() != dGridParticleIndex) SHIM_HIP(hipMemcpyAsync(dGridParticleIndex, v.current(), 4 * (size_t)numParticles, hipMemcpyDeviceToDevice, nullptr));
}

void reorderDataAndFindCellStartDBoundary(SUint_t *cellStart, SUint_t *cellEnd, SR *sortedPos, SR *sortedVbi, SUint_t *gridParticleHash,
                                          SUint_t *gridParticleIndex, SR *oldPos, SR *oldVbi, SUint_t numBoundaries, SUint_t numCells)
{
    SHIM_HIP(hipMemsetAsync(cellStart, 0xff, (size_t)numCells * sizeof(SUint_t), nullptr));
    if (!numBoundaries) return;
    SHIM_NRS(g_sB.alloc(sizeof(T4) * (size_t)numBoundaries));
    hipLaunchKernelGGL((k_reorder_boundary<SR>), blocks(numBoundaries), dim3(BLOCK), 0, nullptr, gridParticleHash, gridParticleIndex, (const T4 *)oldPos,
                       (const SR *)oldVbi, g_sB.as<T4>(), cellStart, cellEnd, numBoundaries);
    hipLaunchKernelGGL(k_unpack_boundary, blocks(numBoundaries), dim3(BLOCK), 0, nullptr, g_sB.as<T4>(), (T4 *)sortedPos, sortedVbi, numBoundaries);
    SHIM_HIP(hipGetLastError());
}

void reorderDataAndFindCellStart(SUint_t *cellStart, SUint_t *cellEnd, SR *sortedPos, SR *sortedVel, SR *sortedDens, SR *sortedPres,
                                 SR *sortedForces, SR *sortedCol, SUint_t *gridParticleHash, SUint_t *gridParticleIndex, SR *oldPos,
                                 SR *oldVel, SR *oldDens, SR *oldPres, SR *oldForces, SR *oldCol, SUint_t numParticles, SUint_t numCells)
{
    (void)sortedDens; (void)sortedForces; (void)sortedCol; (void)oldDens; (void)oldForces; (void)oldCol; // NULL in the reference's launch too
    SHIM_HIP(hipMemsetAsync(cellStart, 0xff, (size_t)numCells * sizeof(SUint_t), nullptr));
    if (!numParticles) return;
    hipLaunchKernelGGL((k_reorder<SR>), blocks(numParticles), dim3(BLOCK), 0, nullptr, gridParticleHash, gridParticleIndex, (const T4 *)oldPos,
                       (const T4 *)oldVel, (const SR *)oldPres, (T4 *)sortedPos, (T4 *)sortedVel, sortedPres, cellStart, cellEnd,
                       (uint32_t *)nullptr, numParticles, (const uint32_t *)nullptr, (uint32_t *)nullptr, (unsigned long long *)nullptr, QuantCfg(),
                       (qword_t *)nullptr);
    SHIM_HIP(hipGetLastError());
}

nrs_vec3 BBMin(SR *sortedBoundaryPos, SUint_t numBoundaries)
{
    SR o[10];
    vec_extremes(sortedBoundaryPos, numBoundaries, o);
    return nrs_vec3{o[0], o[1], o[2]};
}
nrs_vec3 BBMax(SR *sortedBoundaryPos, SUint_t numBoundaries)
{
    SR o[10];
    vec_extremes(sortedBoundaryPos, numBoundaries, o);
    return nrs_vec3{o[3], o[4], o[5]};
}

void computeDensityPressure(SR *sortedPos, SR *sortedVel, SR *sortedDens, SR *sortedPres, SR *sortedForces, SR *sortedCol,
                            SR *sortedBoundaryPos, SR *sortedBoundaryVbi, SUint_t *gridParticleIndex, SUint_t *cellStart, SUint_t *cellEnd,
                            SUint_t *gridBoundaryIndex, SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t numParticles, SUint_t numCells,
                            SUint_t numBoundaries)
{
    (void)sortedCol; (void)gridParticleIndex; (void)numCells;
    need_params("computeDensityPressure");
    if (!numParticles) return;
    const T4 *sB = pack_boundary(true, sortedBoundaryPos, sortedBoundaryVbi, gridBoundaryIndex, numBoundaries);
    const GridView<SR> G = grid_view(cellStart, cellEnd, cellBoundaryStart, cellBoundaryEnd, sB, numParticles);
    const dim3 g = blocks(numParticles), b(BLOCK);
    if (numBoundaries) {
        hipLaunchKernelGGL((k_density_ref<SR, KSET, true>), g, b, 0, nullptr, g_params, G, (const T4 *)sortedPos, sortedDens, sortedPres, numParticles);
        hipLaunchKernelGGL((k_forces_ref<SR, KSET, SURF, true>), g, b, 0, nullptr, g_params, G, (const T4 *)sortedPos, (const T4 *)sortedVel,
                           (const SR *)sortedDens, (const SR *)sortedPres, (T4 *)sortedForces, numParticles);
    } else {
        hipLaunchKernelGGL((k_density_ref<SR, KSET, false>), g, b, 0, nullptr, g_params, G, (const T4 *)sortedPos, sortedDens, sortedPres, numParticles);
        hipLaunchKernelGGL((k_forces_ref<SR, KSET, SURF, false>), g, b, 0, nullptr, g_params, G, (const T4 *)sortedPos, (const T4 *)sortedVel,
                           (const SR *)sortedDens, (const SR *)sortedPres, (T4 *)sortedForces, numParticles);
    }
    SHIM_HIP(hipGetLastError());
}

SR maxDensity(SR *dDensities, SUint_t numParticles)
{
    if (!numParticles) return 0.0f;
    const uint32_t nbk = std::min<uint32_t>(1024u, (numParticles + BLOCK - 1) / BLOCK);
    SHIM_NRS(g_partial.alloc(sizeof(double) * 1024));
    hipLaunchKernelGGL((k_max_partial<SR, false>), dim3(nbk), dim3(BLOCK), 0, nullptr, (const void *)dDensities, g_partial.as<double>(), numParticles);
    std::vector<double> h(nbk);
    SHIM_HIP(hipMemcpy(h.data(), g_partial.p, sizeof(double) * nbk, hipMemcpyDeviceToHost));
    double m = h[0];
    for (uint32_t i = 1; i < nbk; ++i) m = std::max(m, h[i]);
    return (SR)m;
}
nrs_vec4 maxVelocity(SR *dVelocities, SUint_t numParticles)
{
    SR o[10];
    vec_extremes(dVelocities, numParticles, o);
    return nrs_vec4{o[6], o[7], o[8], o[9]};
}

void predictAdvection(SR *sortedPos, SR *sortedVel, SR *sortedDens, SR *sortedPres, SR *sortedForces, SR *sortedCol,
                      SUint_t *cellStart, SUint_t *cellEnd, SUint_t *gridParticleIndex, SR *sortedBoundaryPos, SR *sortedBoundaryVbi,
                      SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t *gridBoundaryIndex, SR *sortedDensAdv, SR *sortedDensCorr,
                      SR *sortedP_l, SR *sortedPreviousP, SR *sortedAii, SR *sortedVelAdv, SR *sortedForcesAdv, SR *sortedForcesP,
                      SR *sortedDiiFluid, SR *sortedDiiBoundary, SR *sortedSumDij, SR *sortedNormal, SUint_t numParticles,
                      SUint_t numBoundaries, SUint_t numCells)
{
    (void)sortedForces; (void)sortedCol; (void)gridBoundaryIndex; (void)sortedNormal; (void)numCells;
    need_params("predictAdvection");
    if (!numParticles) return;
    make_inv(gridParticleIndex, numParticles);
    const T4 *sB = pack_boundary(false, sortedBoundaryPos, sortedBoundaryVbi, nullptr, numBoundaries);
    const GridView<SR> G = grid_view(cellStart, cellEnd, cellBoundaryStart, cellBoundaryEnd, sB, numParticles);
    const IisphArrays<SR> I = iisph_view(sortedDensAdv, sortedDensCorr, sortedP_l, sortedPreviousP, sortedAii, sortedVelAdv, sortedForcesAdv,
                                            sortedForcesP, sortedDiiFluid, sortedDiiBoundary, sortedSumDij);
    const dim3 g = blocks(numParticles), b(BLOCK);
    const T4 *sp = (const T4 *)sortedPos, *sv = (const T4 *)sortedVel;
    if (numBoundaries) {
        hipLaunchKernelGGL((k_density_ref<SR, KSET, true>), g, b, 0, nullptr, g_params, G, sp, sortedDens, (SR *)nullptr, numParticles);
        hipLaunchKernelGGL((k_displacement_ref<SR, KSET, SURF, true>), g, b, 0, nullptr, g_params, G, I, sp, sv, (const SR *)sortedDens, (const SR *)sortedPres, numParticles);
        hipLaunchKernelGGL((k_advection_ref<SR, KSET, true>), g, b, 0, nullptr, g_params, G, I, sp, sv, (const SR *)sortedDens, (const SR *)sortedPres, numParticles);
    } else {
        hipLaunchKernelGGL((k_density_ref<SR, KSET, false>), g, b, 0, nullptr, g_params, G, sp, sortedDens, (SR *)nullptr, numParticles);
        hipLaunchKernelGGL((k_displacement_ref<SR, KSET, SURF, false>), g, b, 0, nullptr, g_params, G, I, sp, sv, (const SR *)sortedDens, (const SR *)sortedPres, numParticles);
        hipLaunchKernelGGL((k_advection_ref<SR, KSET, false>), g, b, 0, nullptr, g_params, G, I, sp, sv, (const SR *)sortedDens, (const SR *)sortedPres, numParticles);
    }
    SHIM_HIP(hipGetLastError());
}

void pressureSolve(SR *sortedPos, SR *sortedVel, SR *sortedDens, SR *sortedPres, SR *sortedForces, SR *sortedCol,
                   SUint_t *cellStart, SUint_t *cellEnd, SUint_t *gridParticleIndex, SR *sortedBoundaryPos, SR *sortedBoundaryVbi,
                   SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t *gridBoundaryIndex, SR *sortedDensAdv, SR *sortedDensCorr,
                   SR *sortedP_l, SR *sortedPreviousP, SR *sortedAii, SR *sortedVelAdv, SR *sortedForcesAdv, SR *sortedForcesP,
                   SR *sortedDiiFluid, SR *sortedDiiBoundary, SR *sortedSumDij, SR *sortedNormal, SUint_t numParticles,
                   SUint_t numBoundaries, SUint_t numCells)
{
    (void)sortedForces; (void)sortedCol; (void)gridBoundaryIndex; (void)sortedNormal; (void)numCells;
    need_params("pressureSolve");
    if (!numParticles) return;
    make_inv(gridParticleIndex, numParticles);
    const T4 *sB = pack_boundary(false, sortedBoundaryPos, sortedBoundaryVbi, nullptr, numBoundaries);
    const GridView<SR> G = grid_view(cellStart, cellEnd, cellBoundaryStart, cellBoundaryEnd, sB, numParticles);
    IisphArrays<SR> I = iisph_view(sortedDensAdv, sortedDensCorr, sortedP_l, sortedPreviousP, sortedAii, sortedVelAdv, sortedForcesAdv,
                                      sortedForcesP, sortedDiiFluid, sortedDiiBoundary, sortedSumDij);
    const dim3 g = blocks(numParticles), b(BLOCK);
    const T4 *sp = (const T4 *)sortedPos;
    const uint32_t nbk = std::min<uint32_t>(1024u, g.x);
    SHIM_NRS(g_partial.alloc(sizeof(double) * 1024));
    SHIM_NRS(g_out.alloc(16 * sizeof(double)));
    // while ((rho_avg - 1000) > 1 || l < 2), sph_cuda.cu:736-741; the sum of the corrected densities in double (thrust::reduce's order
    // is unspecified), rounded to SReal and divided as the reference does (:818-819)
    uint32_t l = 0;
    SR rho_avg = (SR)0;
    const SR rd = (SR)1000, max_rho_err = (SR)1;
    while (((rho_avg - rd) > max_rho_err) || (l < 2)) {
        hipLaunchKernelGGL((k_sumdij_ref<SR, KSET>), g, b, 0, nullptr, g_params, G, I, sp, (const SR *)sortedDens, numParticles);
        if (numBoundaries) hipLaunchKernelGGL((k_pressure_ref<SR, KSET, true>), g, b, 0, nullptr, g_params, G, I, sp, (const SR *)sortedDens, sortedPres, numParticles);
        else hipLaunchKernelGGL((k_pressure_ref<SR, KSET, false>), g, b, 0, nullptr, g_params, G, I, sp, (const SR *)sortedDens, sortedPres, numParticles);
        std::swap(I.P_l, I.P_l_next); // (Q7: true Jacobi, the reference's unused sortedPreviousP is the second buffer)
        hipLaunchKernelGGL((k_sum_partial<SR>), dim3(nbk), b, 0, nullptr, (const SR *)sortedDensCorr, g_partial.as<double>(), numParticles,
                           (const T4 *)nullptr, (unsigned long long *)nullptr);
        hipLaunchKernelGGL(k_sum_final, dim3(1), b, 0, nullptr, g_partial.as<double>(), g_out.as<double>(), nbk);
        double acc = 0.0;
        SHIM_HIP(hipMemcpy(&acc, g_out.p, sizeof(double), hipMemcpyDeviceToHost));
        rho_avg = (SR)acc;
        rho_avg /= numParticles;
        l++;
    }
    g_lastIters = l;
    if (I.P_l != sortedP_l) { // an odd number of iterations left the newest pressures in the second buffer: the caller finds them in sortedP_l
        SHIM_HIP(hipMemcpyAsync(sortedP_l, I.P_l, sizeof(SR) * (size_t)numParticles, hipMemcpyDeviceToDevice, nullptr));
        I.P_l = sortedP_l; I.P_l_next = sortedPreviousP;
    }
    if (numBoundaries) hipLaunchKernelGGL((k_pforce_ref<SR, KSET, true>), g, b, 0, nullptr, g_params, G, I, sp, (const SR *)sortedDens, (const SR *)sortedPres, numParticles);
    else hipLaunchKernelGGL((k_pforce_ref<SR, KSET, false>), g, b, 0, nullptr, g_params, G, I, sp, (const SR *)sortedDens, (const SR *)sortedPres, numParticles);
    hipLaunchKernelGGL((k_iisph_integrate<SR>), g, b, 0, nullptr, g_params, (T4 *)sortedPos, (T4 *)sortedVel, (const T4 *)sortedVelAdv, (const T4 *)sortedForcesP,
                       numParticles, (uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0);
    SHIM_HIP(hipGetLastError());
}

SUint_t nrs_refshim_last_iterations(void) { return g_lastIters; }

} // extern "C"
