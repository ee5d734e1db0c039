// pcisph.h — Nereus::PCISPH.  The reference's PCISPH is unfinished (README "soon finished"): its update()
// computes densities and its pressure solve is an empty stub (sph/pcisph/pcisph.cpp:161-204,
// sph_kernel_impl.cuh:1722-1730).  This header exists so main.cpp:6 still includes; the class behaves like
// the reference's: a step evaluates density/pressure and moves nothing.
#pragma once
#ifndef PCISPH_H
#define PCISPH_H
#include "sph.h"

NEREUS_NAMESPACE_BEGIN

class PCISPH : public SPH {
public:
    PCISPH();
    PCISPH(SphSimParams params);
    virtual ~PCISPH();
    virtual void _initialize();
    virtual void _finalize();
    void update();
};

NEREUS_NAMESPACE_END
#endif // PCISPH_H
