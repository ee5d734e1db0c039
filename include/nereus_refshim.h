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
 * What the reference fixes at COMPILE time is fixed here too: SReal = float (DOUBLE_PRECISION=0), Muller kernels (KERNEL_SET=1),
 * USE_SURFACE_TENSION=1 — the reference's shipped build (CMakeLists.txt:25-28).
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

typedef float SReal_f32;
typedef unsigned int SUint_t;
typedef struct nrs_vec3_f32 { float x, y, z; } nrs_vec3_f32;       /* SVec3 = float3 */
typedef struct nrs_vec4_f32 { float x, y, z, w; } nrs_vec4_f32;    /* SVec4 = float4 */

/* sph.cuh:25-26, 28; sph_cuda.cu:94-110 */
void allocateArray(void **devPtr, size_t size);
void freeArray(void *devPtr);
void threadSync(void);
/* sph.cuh:30-31; sph_cuda.cu:115-118,150-178 (the cudaGraphicsResource argument must be NULL: no GL interop) */
void copyArrayToDevice(void *device, const void *host, int offset, int size);
void copyArrayFromDevice(void *host, const void *device, void **cuda_vbo_resource, int size);
/* sph.cuh:40; sph_cuda.cu:183-187: the parameter block every later launch uses (the reference's __constant__ sph_params) */
void setParameters(nrs_params_f32 *hostParams);
/* sph.cuh:50-54; sph_cuda.cu:211-225 (deltaTime is what integrate_functor uses; the reference passes params.timestep) */
void integrateSystem(float *pos, float *vel, float *forces, float deltaTime, SUint_t numParticles);
/* sph.cuh:59-62; sph_cuda.cu:230-246 */
void calcHash(SUint_t *gridParticleHash, SUint_t *gridParticleIndex, float *pos, int numParticles);
/* sph.cuh:160; sph_cuda.cu:58-63: ascending by key, stable (ties keep index order), as thrust::sort_by_key's radix sort */
void sortParticles(SUint_t *dGridParticleHash, SUint_t *dGridParticleIndex, SUint_t numParticles);
/* sph.cuh:67-77; sph_cuda.cu:251-293 (Q1: sortedPos / sortedVbi ARE filled) */
void reorderDataAndFindCellStartDBoundary(SUint_t *cellStart, SUint_t *cellEnd, float *sortedPos, float *sortedVbi, SUint_t *gridParticleHash,
                                          SUint_t *gridParticleIndex, float *oldPos, float *oldVbi, SUint_t numBoundaries, SUint_t numCells);
/* sph.cuh:82-99; sph_cuda.cu:295-358: memsets cellStart, gathers pos / vel / pres (the other arrays are passed NULL by the
 * reference's own launcher and are ignored here as there) */
void reorderDataAndFindCellStart(SUint_t *cellStart, SUint_t *cellEnd, float *sortedPos, float *sortedVel, float *sortedDens, float *sortedPres,
                                 float *sortedForces, float *sortedCol, SUint_t *gridParticleHash, SUint_t *gridParticleIndex, float *oldPos,
                                 float *oldVel, float *oldDens, float *oldPres, float *oldForces, float *oldCol, SUint_t numParticles,
                                 SUint_t numCells);
/* sph.cuh:104-108; sph_cuda.cu:461-505 */
nrs_vec3_f32 BBMin(float *sortedBoundaryPos, SUint_t numBoundaries);
nrs_vec3_f32 BBMax(float *sortedBoundaryPos, SUint_t numBoundaries);
/* sph.cuh:113-130; sph_cuda.cu:366-456: the density / Tait-pressure kernel AND the force kernel, as the reference's launcher */
void computeDensityPressure(float *sortedPos, float *sortedVel, float *sortedDens, float *sortedPres, float *sortedForces, float *sortedCol,
                            float *sortedBoundaryPos, float *sortedBoundaryVbi, SUint_t *gridParticleIndex, SUint_t *cellStart, SUint_t *cellEnd,
                            SUint_t *gridBoundaryIndex, SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t numParticles,
                            SUint_t numCells, SUint_t numBoundaries);
/* sph.cuh:155-158; sph_cuda.cu:32-53 (maxVelocity returns the velocity vector of largest length) */
float maxDensity(float *dDensities, SUint_t numParticles);
nrs_vec4_f32 maxVelocity(float *dVelocities, SUint_t numParticles);
/* IISPH: sph.cuh:168-207; sph_cuda.cu:513-899 */
void predictAdvection(float *sortedPos, float *sortedVel, float *sortedDens, float *sortedPres, float *sortedForces, float *sortedCol,
                      SUint_t *cellStart, SUint_t *cellEnd, SUint_t *gridParticleIndex, float *sortedBoundaryPos, float *sortedBoundaryVbi,
                      SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t *gridBoundaryIndex, float *sortedDensAdv,
                      float *sortedDensCorr, float *sortedP_l, float *sortedPreviousP, float *sortedAii, float *sortedVelAdv,
                      float *sortedForcesAdv, float *sortedForcesP, float *sortedDiiFluid, float *sortedDiiBoundary, float *sortedSumDij,
                      float *sortedNormal, SUint_t numParticles, SUint_t numBoundaries, SUint_t numCells);
void pressureSolve(float *sortedPos, float *sortedVel, float *sortedDens, float *sortedPres, float *sortedForces, float *sortedCol,
                   SUint_t *cellStart, SUint_t *cellEnd, SUint_t *gridParticleIndex, float *sortedBoundaryPos, float *sortedBoundaryVbi,
                   SUint_t *cellBoundaryStart, SUint_t *cellBoundaryEnd, SUint_t *gridBoundaryIndex, float *sortedDensAdv, float *sortedDensCorr,
                   float *sortedP_l, float *sortedPreviousP, float *sortedAii, float *sortedVelAdv, float *sortedForcesAdv, float *sortedForcesP,
                   float *sortedDiiFluid, float *sortedDiiBoundary, float *sortedSumDij, float *sortedNormal, SUint_t numParticles,
                   SUint_t numBoundaries, SUint_t numCells);
/* not in sph.cuh: solver iterations of the last pressureSolve (the `l` of sph_cuda.cu:736), for tests */
SUint_t nrs_refshim_last_iterations(void);

#ifdef __cplusplus
}
#endif
#endif /* NEREUS_REFSHIM_H */
