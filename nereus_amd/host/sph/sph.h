// sph.h — Nereus::SPH, the state-equation SPH (SESPH) solver, MI355X build.
//
// Public surface = the reference's sph/sph.h:26-94 (same method names, argument meaning and error behaviour),
// so main.cpp and other callers compile unchanged.  Everything below the API is new: the particle state
// lives on the GPU inside an nrs_ctx (include/nereus_hip.h); host arrays are synchronised on demand.
#pragma once
#ifndef SPH_H
#define SPH_H

#include <utility>
#include <vector>

#if defined(__has_include)
#if __has_include(<glm/glm.hpp>)
#ifndef GLM_SWIZZLE
#define GLM_SWIZZLE
#endif
#include <glm/glm.hpp> // optional: no GLM type appears in this API
#endif
#endif

#include "common.h"
#include "sph_kernel.cuh"
#include <colored_output.h>

// Initial particle capacity.  The reference hard-caps at this number (sph/sph.h:19); here it is only the
// starting size: host and device storage grow on demand (or set NEREUS_CAPACITY / call reserveParticles()).
#define MAX_PARTICLE_NUMBER 150000

struct nrs_ctx;

NEREUS_NAMESPACE_BEGIN

class SPH {
public:
    SPH();
    SPH(SphSimParams params);
    virtual ~SPH();

    // life cycle (idempotent: may be called again, as main.cpp:534 does)
    virtual void _initialize();
    virtual void _finalize();
    virtual void _initializeGrid();

    // fluid creation; particles may be appended between steps (main.cpp:499-513)
    virtual void addNewParticle(SVec4 p, SVec4 v);
    virtual void generateParticleCube(SVec4 center, SVec4 size, SVec4 vel);

    // one simulation step; on return the new state is observable through the getters
    virtual void update();

    // grid sizing from the boundary AABB
    void updateGrid();
    std::pair<SVec3, SVec3> computeGridMinMax() const;

    // physics constants
    SReal getGasStiffness() const { return m_params.gasStiffness; }
    SReal getRestDensity() const { return m_params.restDensity; }
    SReal getParticleMass() const { return m_params.particleMass; }
    SReal getParticleRadius() const { return m_params.particleRadius; }
    SReal getTimestep() const { return m_params.timestep; }
    SReal getViscosity() const { return m_params.viscosity; }
    SReal getSurfaceTension() const { return m_params.surfaceTension; }
    SReal getInteractionRadius() const { return m_params.interactionRadius; }
    SUint getNumCells() const { return m_params.numCells; }
    void setGasStiffness(SReal v) { m_params.gasStiffness = v; }
    void setRestDensity(SReal v) { m_params.restDensity = v; }
    void setParticleMass(SReal v) { m_params.particleMass = v; }
    void setViscosity(SReal v) { m_params.viscosity = v; }
    void setSurfaceTension(SReal v) { m_params.surfaceTension = v; }
    void setGravity(SReal gy) { m_params.gravity.y = gy; }

    // particle arrays, AoS xyzw, 4*getNumParticles() SReals, owned by the solver, valid until the next
    // update()/append/destruction.  The mutable accessors assume the caller may write and re-upload the
    // arrays before the next step; the const ones only read.
    SReal *&getPos();
    SReal *&getCol();
    SReal *&getVel();
    SReal *getHostPos() const;
    SReal *getHostCol() const;
    SUint getNumParticles() const { return m_numParticles; }

    // boundary particles: caller keeps ownership of bi (xyzw) / vbi; copied in updateGpuBoundaries
    void setBi(SReal *bi) { m_bi = bi; }
    void setVbi(SReal *vbi) { m_vbi = vbi; }
    void setNumBoundaries(SUint nb) { m_num_boundaries = nb; }
    void updateGpuBoundaries(SUint nb_boundary_spheres);

    // ---- additions of this build (not in the reference) ----
    void reserveParticles(SUint capacity);        // grow host+device storage up front
    void setEagerSync(bool on) { m_eagerSync = on; } // true: D2H at the end of every update() like the reference
    const SphSimParams &getParams() const { return m_params; }
    SReal *getHostVel() const;
    SReal *getHostPressure() const;
    void synchronize() const;                       // wait for the device
    nrs_ctx *deviceContext();                      // the underlying C-ABI handle (created on demand)
    // CFL time step, the block the reference keeps under `#if 0` (sph/sph.cpp:217-231): before each step
    // dt = lambda * h / max|v| (lambda = 0.4) from a device-side reduction; off by default as in the reference build
    void setAdaptiveTimestep(bool on, SReal lambda = 0.4f) { m_cfl = on; m_cflLambda = lambda; }
    // binary checkpoint of the particle state (parameters, positions, velocities, IISPH warm-start pressure);
    // the reference has none.  A restored solver continues bit-identically.
    bool saveState(const char *path) const;
    bool loadState(const char *path);
    // Viewer hand-off that does not stall the solver (the reference copies pos+vel back synchronously at the end of every
    // update(), sph.cpp:283-284, and main.cpp:587-588 uploads them to a VBO).  With asynchronous read-back on, update()
    // also starts a device -> page-locked-host copy of the new positions on a copy stream (nrs_snapshot_begin), which
    // overlaps with the updates enqueued after it.  latestFrame() never waits for the solver: it returns the newest
    // frame whose transfer has completed (at most two updates old; it blocks only while no frame has arrived yet);
    // getHostPos() keeps the reference's meaning (positions after the last update()) and waits for that frame.
    void setAsyncReadback(bool on);
    const SReal *latestFrame(SUint *numParticles = nullptr, unsigned long long *step = nullptr);

protected:
    virtual int solverKind() const; // NRS_SOLVER_*
    void ensureContext();
    void releaseContext();
    void growHost(SUint capacity);
    void pushHostToDevice();
    void pullDeviceToHost() const;
    [[noreturn]] void fatal(const char *what) const;

    SphSimParams m_params;
    SUint m_gridSortBits; // kept: the reference sets 32 and never uses it

    // host arrays (m_pos etc. keep the reference's names for derived classes)
    mutable SReal *m_pos, *m_vel, *m_density, *m_pressure, *m_forces, *m_colors;
    SUint m_numParticles;
    SUint m_hostCapacity;

    SReal *m_bi, *m_vbi; // caller-owned
    SUint m_num_boundaries;

    // device side
    nrs_ctx *m_ctx;
    SUint m_ctxCapacity;
    bool m_hostDirty;           // host arrays changed since the last upload
    mutable bool m_deviceNewer; // device holds a newer state than the host arrays
    bool m_boundariesPending;   // boundaries known but not yet uploaded into (a new) context
    bool m_eagerSync;
    bool m_initialized;
    bool m_cfl;
    SReal m_cflLambda;
    bool m_asyncReadback;
    mutable int m_framesInFlight;        // snapshots begun and not yet collected (0..2)
    mutable const SReal *m_frame;        // newest collected frame (library-owned pinned memory)
    mutable SUint m_frameCount;
    mutable unsigned long long m_frameStep;
    mutable bool m_frameIsCurrent;       // m_frame shows the state after the last update()
    void collectFrames(bool waitForAll) const;
};

NEREUS_NAMESPACE_END
#endif // SPH_H
