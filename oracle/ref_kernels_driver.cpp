/*
 * ref_kernels_driver.cpp — batch driver around the REFERENCE's own smoothing kernels and vector helpers.
 * TEST INFRASTRUCTURE ONLY (same rules as nereus_oracle.cpp: tests/ and build() only).
 *
 * What this is: the reference files /root/reference/common/kernels_impl.cuh (+ the helper_math.h it
 * includes) are compiled BY PATH, unmodified, with plain g++ against the NVIDIA CUDA headers that ship in
 * this image (triton/backends/nvidia/include): those headers define __host__/__device__/__forceinline__ away
 * under a host compiler, so every `__device__ __host__` function of that file is an ordinary inline host
 * function.  No builtin shim, no stand-in header, no copy of reference source: this file only CALLS them.
 * The recipe is oracle/Makefile target `ref`; outputs go to oracle/_ref/ (git-ignored, not gpurun-ignored).
 *
 * What it pins: SURVEY §8 rows a1 (the SphSimParams layout of common/sph_kernel.cuh:13-59, ref_params_layout), a9 (smoothing
 * kernels, common/kernels_impl.cuh:85-247) and a13 (float-scalar
 * vector semantics, common/cuda_helpers/helper_math.h:817-829,1000-1008,1251-1301) — as the HOST compiler
 * resolves them (e.g. `pow(SReal,int)` at kernels_impl.cuh:95 is the double pow under g++).  Nothing else:
 * sph_kernel_impl.cuh needs nvcc builtins (threadIdx, __syncthreads, __umul24, <<<>>>) and is not built.
 */
#include <cmath>
#include <cstdio>
#include <cstring>

#ifndef REF_KERNELS_IMPL
#error "pass -DREF_KERNELS_IMPL='\"/root/reference/common/kernels_impl.cuh\"' (see oracle/Makefile)"
#endif
#include REF_KERNELS_IMPL
#ifdef REF_SPH_KERNEL
#include <cstddef>
#include REF_SPH_KERNEL /* common/sph_kernel.cuh: the SphSimParams POD, also by path and unmodified */
#endif

extern "C" {

int ref_sizeof_real(void) { return (int)sizeof(SReal); }

#ifdef REF_SPH_KERNEL
/* sizeof(SphSimParams) followed by the offset of every field, in declaration order (26 values) */
int ref_params_layout(unsigned *out)
{
    unsigned k = 0;
    out[k++] = (unsigned)sizeof(SphSimParams);
#define OFF(f) out[k++] = (unsigned)offsetof(SphSimParams, f)
    OFF(gridSize); OFF(numCells); OFF(worldOrigin); OFF(cellSize); OFF(numBodies); OFF(maxParticlesPerCell);
    OFF(gasStiffness); OFF(viscosity); OFF(surfaceTension); OFF(restDensity); OFF(particleMass); OFF(interactionRadius);
    OFF(timestep); OFF(particleRadius); OFF(gravity); OFF(soundSpeed); OFF(beta); OFF(kpoly); OFF(kpoly_grad); OFF(kpress_grad);
    OFF(kvisc_grad); OFF(kvisc_denum); OFF(ksurf1); OFF(ksurf2); OFF(bpol);
#undef OFF
    return (int)k;
}
#endif

/* which: 0 Wdefault(r,h,c0)  1 Wdefault_grad(r,h,c0)  2 Wpressure_grad(r,h,c0)  3 Wviscosity_grad(r,h,c0,c1)
 *        4 Wmonaghan(r,h)    5 Wmonaghan_grad(r,h)    6 Cakinci(r,h,c0,c1)      7 Aboundary(r,h,c0)
 *        8 dot(r,s)          9 length(r)              10 r*(float)c0            11 (float)c0*r
 *        12 r/(float)c0      13 make_SVec3(SVec4)     14 r+s                    15 r-s
 * r3/s3: n xyz triples of SReal; out: n*3 SReal (scalars in [3i], rest 0). */
int ref_eval(int which, unsigned n, const SReal *r3, const SReal *s3, SReal h, SReal c0, SReal c1, SReal *out)
{
    for (unsigned i = 0; i < n; ++i) {
        const SVec3 r = make_SVec3(r3[3 * i], r3[3 * i + 1], r3[3 * i + 2]);
        SVec3 s = make_SVec3(0, 0, 0);
        if (s3) s = make_SVec3(s3[3 * i], s3[3 * i + 1], s3[3 * i + 2]);
        SVec3 v = make_SVec3(0, 0, 0);
        switch (which) {
        case 0: v.x = Wdefault(r, h, c0); break;
        case 1: v = Wdefault_grad(r, h, c0); break;
        case 2: v = Wpressure_grad(r, h, c0); break;
        case 3: v = Wviscosity_grad(r, h, c0, c1); break;
        case 4: v.x = Wmonaghan(r, h); break;
        case 5: v = Wmonaghan_grad(r, h); break;
        case 6: v.x = Cakinci(r, h, c0, c1); break;
        case 7: v.x = Aboundary(r, h, c0); break;
        case 8: v.x = dot(r, s); break;
        case 9: v.x = length(r); break;
        case 10: v = r * (float)c0; break;
        case 11: v = (float)c0 * r; break;
        case 12: v = r / (float)c0; break;
        case 13: v = make_SVec3(make_SVec4(r.x, r.y, r.z, (SReal)7)); break;
        case 14: v = r + s; break;
        case 15: v = r - s; break;
        default: return -1;
        }
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
    return 0;
}

} /* extern "C" */
