// pcisph.cpp — Nereus::PCISPH as the reference leaves it (sph/pcisph/pcisph.cpp:161-204): update() uploads the host state,
// builds the neighbour grid (calcHash / sortParticles / reorderDataAndFindCellStart), evaluates density and Tait pressure
// (pcisph_internalForces launches computeDensityPressure, sph_cuda.cu:905-936), runs the EMPTY pressure solve
// (pcisph_pressureSolve, sph_cuda.cu:944-952) and copies the SORTED positions / velocities back to the host arrays
// (:201-202): nothing moves, but the host arrays come back permuted into hash order.  Mirrored here exactly;
// densities and pressures of the step are additionally readable on the host (getHostDensity / getHostPressure).
#include "pcisph.h"

#include "nereus_hip.h"

NEREUS_NAMESPACE_BEGIN

PCISPH::PCISPH() : SPH() {}
PCISPH::PCISPH(SphSimParams params) : SPH(params) {}
PCISPH::~PCISPH() {}
void PCISPH::_initialize() { SPH::_initialize(); }
void PCISPH::_finalize() { SPH::_finalize(); }

void PCISPH::update()
{
    if (m_numParticles == 0) return;
    ensureContext();
    m_hostDirty = true; // the reference uploads the host arrays every step (pcisph.cpp:164-165)
    pushHostToDevice();
    if (nrs_set_params(m_ctx, &m_params) != NRS_OK) fatal("nrs_set_params");
    if (nrs_step_partial(m_ctx, NRS_STAGE_DENSITY) != NRS_OK) fatal("nrs_step_partial");
    // D2H of the sorted arrays (pcisph.cpp:201-202) + the step's density / pressure
    const uint64_t v = sizeof(SReal) * 4 * (uint64_t)m_numParticles, sc = sizeof(SReal) * (uint64_t)m_numParticles;
    if (nrs_get_array(m_ctx, NRS_ARR_SORTED_POS, m_pos, v, nullptr) != NRS_OK) fatal("nrs_get_array(sorted pos)");
    if (nrs_get_array(m_ctx, NRS_ARR_SORTED_VEL, m_vel, v, nullptr) != NRS_OK) fatal("nrs_get_array(sorted vel)");
    if (nrs_get_array(m_ctx, NRS_ARR_DENS, m_density, sc, nullptr) != NRS_OK) fatal("nrs_get_array(dens)");
    if (nrs_get_array(m_ctx, NRS_ARR_PRES, m_pressure, sc, nullptr) != NRS_OK) fatal("nrs_get_array(pres)");
    m_deviceNewer = false; // the host arrays are the state now (the partial step left the device mid-update)
    m_hostDirty = true;
}

NEREUS_NAMESPACE_END
