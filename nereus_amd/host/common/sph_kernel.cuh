// sph_kernel.cuh — SphSimParams, the parameter block shared by the host classes and the device library.
// Field order and sizes are those of the reference's common/sph_kernel.cuh:13-59 (132 bytes for SReal=float,
// 240 for double), i.e. byte-identical to nrs_params_f32 / nrs_params_f64 of include/nereus_hip.h.
#pragma once
#ifndef SPH_KERNEL_H
#define SPH_KERNEL_H
#include "common.h"

struct SphSimParams {
    // uniform grid
    uint3 gridSize;
    unsigned int numCells;
    SVec3 worldOrigin;
    SVec3 cellSize;
    unsigned int numBodies;           // unused (kept for layout)
    unsigned int maxParticlesPerCell; // unused (kept for layout)
    // physics
    SReal gasStiffness, viscosity, surfaceTension, restDensity, particleMass, interactionRadius, timestep,
        particleRadius;
    SVec3 gravity;
    SReal soundSpeed;
    SReal beta; // boundary adhesion
    // pre-computed smoothing-kernel factors
    SReal kpoly, kpoly_grad, kpress_grad, kvisc_grad, kvisc_denum, ksurf1, ksurf2, bpol;
};
static_assert(sizeof(SphSimParams) == (DOUBLE_PRECISION == 1 ? 240 : 132), "SphSimParams must match the device ABI");

#endif // SPH_KERNEL_H
