/*
 * nereus_refshim.h — the reference's own launcher names (sph/sph.cuh:19-230, defined in sph/sph_cuda.cu) on caller-owned HIP
 * device pointers: libnereus_refshim.so.
 *
 * "for symbol-level compatibility we additionally export the reference names that make sense stand-alone".
 * These are thin launches of the reference-order gfx950 kernels (nereus_amd/csrc/nrs_kernels_ref.h) — the same kernels
 * NRS_FLAG_REFERENCE_ORDER selects inside an nrs_ctx — with the signatures of sph.cuh, so that a caller written against that layer
 * (it owns every array, it sequences the stages itself) can run stage by stage on an MI355X.  The fast path is the context API of
 * nereus_hip.h (device-resident state, fused launches, coherent re-sort); this layer exists for A/B-ing single stages.
 *
 * What the reference fixes at COMPILE time is fixed per library here too: libnereus_refshim.so is SReal = float (DOUBLE_PRECISION=0),
 * Muller kernels (KERNEL_SET=1), USE_SURFACE_TENSION=1 — the reference's shipped build (CMakeLists.txt:25-28); libnereus_refshim_f64.so,
 * libnereus_refshim_monaghan.so and libnereus_refshim_f64_monaghan.so are the other three DOUBLE_PRECISION x KERNEL_SET combinations.
 *
 * What this layer does NOT provide: the CUDA runtime.  The reference's sph.cpp also calls cudaMalloc / cudaMemcpy / cudaMemset
 * directly (sph.cpp:141-185,233-284); a caller of this layer allocates with allocateArray (hipMalloc) or hands in any HIP device
 * pointer.  The GL interop entry points (registerGLBufferObject, mapGLBufferObject, ...) and cudaInit have no counterpart (no GL in
 * this build); computePciDensityPressure is declared by the reference but never defined (sph.cuh:135-151) and is not defined here.
 *
 * Semantics kept from the reference: void returns, fatal on error (message on stderr + exit(EXIT_FAILURE), as checkCudaErrors
 * does), default (NULL) stream, synchronous with respect to the host only where the reference is (the reductions and copies).
 * Deliberate differences, each a documented defect of the reference (SURVEY quirk register):
 *   Q1  reorderDataAndFindCellStartDBoundary fills sortedPos / sortedVbi (the reference's kernel writes oldPos in place and
 *       never touches them); the SESPH entry (computeDensityPressure) takes the UNSORTED boundary arrays indexed through
 *       gridBoundaryIndex, the IISPH entries the SORTED ones, exactly as the reference's kernels index them;
 *   Q3  kernels run one thread per sorted slot (the result per slot is the same; gridParticleIndex is only read where Q5 needs it);
 *   Q7  pressureSolve iterates true Jacobi: sortedPreviousP — allocated but unused by the reference — is the second buffer, and
 *       the final pressures are left in sortedP_l as the reference leaves them;
 *   allocateArray takes size_t (sph_cuda.cu:94 defines it so; sph.cuh:25 declares int).
 */
#ifndef NEREUS_REFSHIM_H
#define NEREUS_REFSHIM_H

#include <stddef.h>

#include "nereus_hip.h" /* nrs_params_f32 = SphSimParams with SReal = float */

#ifdef __cplusplus
extern "C" {
#endif

/* SReal as the reference's common/common.h:23-43 selects it: compile the CALLER with -DDOUBLE_PRECISION=1 and link
 * libnereus_refshim_f64[_monaghan].so for the double build (KERNEL_SET=0: ..._monaghan) */
#if defined(DOUBLE_PRECISION) && DOUBLE_PRECISION
typedef double nrs_sreal;
#else
typedef float nrs_sreal;
#endif
typedef unsigned int SUint_t;
typedef struct nrs_vec3 { nrs_sreal x, y, z; } nrs_vec3;       /* SVec3 */
typedef struct nrs_vec4 { nrs_sreal x, y, z, w; } nrs_vec4;    /* SVec4 = float4 */

/* sph.cuh:25-26, 28; sph_cuda.cu:94-110 */
void allocateArray(void **devPtr, size_t size);
void freeArray(void *devPtr);
void threadSync(void);
/* sph.cuh:30-31; sph_cuda.cu:115-118,150-178 (the cudaGraphicsResource argument must be NULL: no GL interop) */
void copyArrayToDevice(void *device, const void *host, int offset, int size);
void copyArrayFromDevice(void *host, const void *device, void **cuda_vbo_resource, int size);
/* sph.cuh:40; sph_cuda.cu:183-187: the parameter block every later launch uses (the reference's __constant__ sph_params) */
void setParameters(void *hostParams); /* SphSimParams = nrs_params_f32 / nrs_params_f64 by DOUBLE_PRECISION */
/* sph.cuh:50-54; sph_cuda.cu:211-225 (deltaTime is what integrate_functor uses; the reference passes params.timestep) */
void integrateSystem(nrs_sreal *pos, nrs_sreal *vel, nrs_sreal *forces, nrs_sreal deltaTime, SUint_t numParticles);
/* sph.cuh:59-62; sph_cuda.cu:230-246 */
void calcHash(SUint_t *gridParticleHash, SUint_t *gridParticleIndex, nrs_sreal *pos, int numParticles);
/* sph.cuh:160; sph_cuda.cu:58-63: ascending by key, stable (ties keep index order), as thrust::sort_by_key's radix sort */
void sortParticles(SUint_t *dGridParticleHash, SUint_t *dGridParticleIndex, SUint_t numParticles);
/* sph.cuh:67-77; sph_cuda.cu:251-293 (Q1: sortedPos / sortedVbi ARE filled) */
void reorderDataAndFindCellStartDBoundary(SUint_t *cellStart, SUint_t *cellEnd, nrs_sreal *sortedPos, nrs_sreal *sortedVbi, SUint_t *gridParticleHash,
                                          SUint_t *gridParticleIndex, nrs_sreal *oldPos, nrs_sreal *oldVbi, SUint_t numBoundaries, SUint_t numCells);
/* sph.cuh:82-99; sph_cuda.cu:295-358: memsets cellStart, gathers pos / vel / pres (the other arrays are passed NULL by the
 * reference's own launcher and are ignored here as there) */
void reorderDataAndFindCellStart(SUint_t *cellStart, SUint_t *cellEnd, nrs_sreal *sortedPos, nrs_sreal *sortedVel, nrs_sreal *sortedDens, nrs_sreal *sortedPres,
                                 nrs_sreal *sortedForces, nrs_sreal *sortedCol, SUint_t *gridParticleHash, SUint_t *gridParticleIndex, nrs_sreal *oldPos,
                                 nrs_sreal *oldVel, nrs_sreal *oldDens, nrs_sreal *oldPres, nrs_sreal *oldForces, nrs_sreal *oldCol, SUint_t numParticles,
                                 SUint_t numCells);
/* sph.cuh:104-108; sph_cuda.cu:461-505 */
nrs_vec3 BBMin(nrs_sreal *sortedBoundaryPos, SUint_t numBoundaries);
nrs_vec3 BBMax(nrs_sreal *sortedBoundaryPos, SUint_t numBoundaries);
/* sph.cuh:113-130; sph_cuda.cu:366-456: the density / Tait-pressure kernel AND the force kernel, as the reference's launcher */
void computeDensityPressure(nrs_sreal *sortedPos, nrs_sreal *sortedVel, nrs_sreal *sortedDens, nrs_sreal *sortedPres, nrs_sreal *sortedForces, nrs_sreal *sortedCol,
                            nrs_sreal *sortedBoundaryPos, nrs_sreal *sortedBoundaryVbi, SUint_t *gridParticleIndex, SUint_t *cellStart, SUint_t *cellEnd,
                            SUint_t *gridBoundaryIndex, SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t numParticles,
                            SUint_t numCells, SUint_t numBoundaries);
/* sph.cuh:155-158; sph_cuda.cu:32-53 (maxVelocity returns the velocity vector of largest length) */
nrs_sreal maxDensity(nrs_sreal *dDensities, SUint_t numParticles);
nrs_vec4 maxVelocity(nrs_sreal *dVelocities, SUint_t numParticles);
/* IISPH: sph.cuh:168-207; sph_cuda.cu:513-899 */
void predictAdvection(nrs_sreal *sortedPos, nrs_sreal *sortedVel, nrs_sreal *sortedDens, nrs_sreal *sortedPres, nrs_sreal *sortedForces, nrs_sreal *sortedCol,
                      SUint_t *cellStart, SUint_t *cellEnd, SUint_t *gridParticleIndex, nrs_sreal *sortedBoundaryPos, nrs_sreal *sortedBoundaryVbi,
                      SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t *gridBoundaryIndex, nrs_sreal *sortedDensAdv,
                      nrs_sreal *sortedDensCorr, nrs_sreal *sortedP_l, nrs_sreal *sortedPreviousP, nrs_sreal *sortedAii, nrs_sreal *sortedVelAdv,
                      nrs_sreal *sortedForcesAdv, nrs_sreal *sortedForcesP, nrs_sreal *sortedDiiFluid, nrs_sreal *sortedDiiBoundary, nrs_sreal *sortedSumDij,
                      nrs_sreal *sortedNormal, SUint_t numParticles, SUint_t numBoundaries, SUint_t numCells);
void pressureSolve(nrs_sreal *sortedPos, nrs_sreal *sortedVel, nrs_sreal *sortedDens, nrs_sreal *sortedPres, nrs_sreal *sortedForces, nrs_sreal *sortedCol,
                   SUint_t *cellStart, SUint_t *cellEnd, SUint_t *gridParticleIndex, nrs_sreal *sortedBoundaryPos, nrs_sreal *sortedBoundaryVbi,
                   SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t *gridBoundaryIndex, nrs_sreal *sortedDensAdv, nrs_sreal *sortedDensCorr,
                   nrs_sreal *sortedP_l, nrs_sreal *sortedPreviousP, nrs_sreal *sortedAii, nrs_sreal *sortedVelAdv, nrs_sreal *sortedForcesAdv, nrs_sreal *sortedForcesP,
                   nrs_sreal *sortedDiiFluid, nrs_sreal *sortedDiiBoundary, nrs_sreal *sortedSumDij, nrs_sreal *sortedNormal, SUint_t numParticles,
                   SUint_t numBoundaries, SUint_t numCells);
/* not in sph.cuh: solver iterations of the last pressureSolve (the `l` of sph_cuda.cu:736), for tests */
SUint_t nrs_refshim_last_iterations(void);

#ifdef __cplusplus
}
#endif
#endif /* NEREUS_REFSHIM_H */
