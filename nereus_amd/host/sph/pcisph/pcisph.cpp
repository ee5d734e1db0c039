// pcisph.cpp — see pcisph.h: density/pressure evaluation only, as in the (unfinished) reference.
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
    pushHostToDevice();
    if (nrs_set_params(m_ctx, &m_params) != NRS_OK) fatal("nrs_set_params");
    if (nrs_step_partial(m_ctx, NRS_STAGE_DENSITY) != NRS_OK) fatal("nrs_step_partial");
    m_hostDirty = true; // the partial step leaves the device state mid-update: next step re-uploads
}

NEREUS_NAMESPACE_END
