// iisph.cpp — Nereus::IISPH over the nrs_* C ABI.  Defaults: sph/iisph/iisph.cpp:28-87 of the reference
// (they override the base-class values after SPH::SPH() ran); step: iisph.cpp:170-217.
#include "iisph.h"

#include <cmath>
#include <iostream>

#include "nereus_hip.h"

NEREUS_NAMESPACE_BEGIN

void nereusKernelFactors(SphSimParams &p, int flavour);
SReal nereusDefaultSoundSpeed();

IISPH::IISPH() : SPH(), m_maxIterations(0)
{
    std::cout << GREEN << "construction of iisph based system" << RESET << std::endl;
    m_params.restDensity = 1000.0;
    m_params.particleRadius = 0.02;
    m_params.timestep = 1e-3;
    m_params.viscosity = 0.01;
    m_params.surfaceTension = 0.01;
    m_params.gravity = make_SVec3(0.0, -9.81f, 0.0);
    m_params.interactionRadius = 0.0537;
    m_params.particleMass = 0.5 * powf(m_params.interactionRadius, 3) * m_params.restDensity;
    m_params.beta = 1050.0;
    m_params.soundSpeed = nereusDefaultSoundSpeed();
    m_params.worldOrigin = make_SVec3(-1.2, -1.2, -1.2);
    m_params.gridSize = make_uint3(128, 128, 128);
    const SReal h = m_params.interactionRadius;
    m_params.cellSize = make_SVec3(h, h, h);
    m_params.numCells = m_params.gridSize.x * m_params.gridSize.y * m_params.gridSize.z;
    nereusKernelFactors(m_params, 1);
    _initialize();
    m_numParticles = 0;
}

IISPH::IISPH(SphSimParams params) : SPH(params), m_maxIterations(0)
{
    nereusKernelFactors(m_params, 2);
    _initialize();
    m_numParticles = 0;
}

IISPH::~IISPH() {}

int IISPH::solverKind() const { return NRS_SOLVER_IISPH; }

void IISPH::_initialize()
{
    SPH::_initialize(); // the device context is created lazily, after construction, so it is an IISPH one
}

void IISPH::_finalize() { SPH::_finalize(); }

void IISPH::update()
{
    if (m_numParticles == 0) return;
    ensureContext();
    if (m_maxIterations && nrs_set_max_iterations(m_ctx, m_maxIterations) != NRS_OK) fatal("nrs_set_max_iterations");
    SPH::update(); // upload if dirty (pos, vel, warm-start pressure) → step → lazy download
}

SUint IISPH::getLastIterations()
{
    uint32_t it = 0;
    if (m_ctx && nrs_last_iterations(m_ctx, &it) != NRS_OK) fatal("nrs_last_iterations");
    return it;
}

void IISPH::setMaxIterations(SUint cap) { m_maxIterations = cap; }

NEREUS_NAMESPACE_END
