// iisph.h — Nereus::IISPH, implicit incompressible SPH (reference: sph/iisph/iisph.h:8-43).
// Same public surface; the twelve extra per-particle device arrays of the reference live inside the
// nrs_ctx created with NRS_SOLVER_IISPH.
#pragma once
#ifndef IISPH_H
#define IISPH_H
#include "sph.h"

NEREUS_NAMESPACE_BEGIN

class IISPH : public SPH {
public:
    IISPH();
    IISPH(SphSimParams params);
    virtual ~IISPH();

    virtual void _initialize();
    virtual void _finalize();
    void update();

    // additions: solver iterations of the last step (the `l` of sph_cuda.cu:736) and an optional cap
    SUint getLastIterations();
    void setMaxIterations(SUint cap);

protected:
    int solverKind() const override;
    SUint m_maxIterations;
};

NEREUS_NAMESPACE_END
#endif // IISPH_H
