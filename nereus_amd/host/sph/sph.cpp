// sph.cpp — Nereus::SPH over the nrs_* C ABI (libnereus_hip.so).
//
// What each method has to do is defined by the reference's sph/sph.cpp (constructor defaults :29-93,
// _initialize :132-188, update :215-285, updateGrid :313-337, addNewParticle :341-368,
// generateParticleCube :373-386, updateGpuBoundaries :391-432).  How it is done here differs: no per-step
// PCIe copies (state stays on the GPU, host arrays are refreshed when somebody asks for them), run-time
// capacity instead of MAX_PARTICLE_NUMBER, idempotent _initialize().
#include "sph.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "nereus_hip.h"

NEREUS_NAMESPACE_BEGIN

namespace {

// Smoothing-kernel factors.  The three flavours differ only in where M_PI is narrowed to SReal:
// 0 = SPH::SPH() (sph.cpp:76-90), 1 = IISPH::IISPH() (iisph.cpp:70-80), 2 = the SphSimParams constructors.
void kernelFactors(SphSimParams &p, int flavour)
{
    const SReal h = p.interactionRadius;
    const float h3 = powf(h, 3.0), h6 = powf(h, 6.0), h9 = powf(h, 9.0);
    if (flavour == 1) {
        p.kpoly = 315.0 / (64.0 * M_PI * h9);
        p.kpoly_grad = -945.0 / (32.0 * M_PI * h9);
        p.kpress_grad = -45.0 / (M_PI * h6);
        p.kvisc_grad = 15.0 / (2 * M_PI * h3);
        p.ksurf1 = 32.0 / (M_PI * h9);
    } else {
        const SReal pi = (SReal)M_PI;
        p.kpoly = 315.0 / (64.0 * pi * h9);
        p.kpoly_grad = -945.0 / (32.0 * pi * h9);
        p.kpress_grad = -45.0 / (pi * h6);
        p.kvisc_grad = (flavour == 0) ? 15.0 / (2.0 * pi * h3) : 15.0 / (2 * pi * h3);
        p.ksurf1 = 32.0 / (pi * h9);
    }
    p.kvisc_denum = 2.0 * h3;
    p.ksurf2 = h6 / 64.0;
    p.bpol = 0.007f / (powf(h, 3.25));
}

SReal defaultSoundSpeed()
{
    const SReal eta = 0.01, H = 0.1;
    const SReal vf = std::sqrt(2.0 * 9.81 * H);
    return vf / (std::sqrt(eta));
}

SUint nextPow2(SUint v)
{
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

SUint envCapacity()
{
    const char *e = std::getenv("NEREUS_CAPACITY");
    if (e && *e) {
        const long long v = std::atoll(e);
        if (v > 0) return (SUint)std::min<long long>(v, (1ll << 30) - 1);
    }
    return MAX_PARTICLE_NUMBER;
}

} // namespace

void nereusKernelFactors(SphSimParams &p, int flavour) { kernelFactors(p, flavour); }
SReal nereusDefaultSoundSpeed() { return defaultSoundSpeed(); }

SPH::SPH()
    : m_gridSortBits(32), m_pos(nullptr), m_vel(nullptr), m_density(nullptr), m_pressure(nullptr), m_forces(nullptr),
      m_colors(nullptr), m_numParticles(0), m_hostCapacity(0), m_bi(nullptr), m_vbi(nullptr), m_num_boundaries(0),
      m_ctx(nullptr), m_ctxCapacity(0), m_hostDirty(true), m_deviceNewer(false), m_boundariesPending(false),
      m_eagerSync(false), m_initialized(false), m_cfl(false), m_cflLambda(0.4f), m_asyncReadback(false),
      m_framesInFlight(0), m_frame(nullptr), m_frameCount(0), m_frameStep(0), m_frameIsCurrent(false)
{
    std::cout << GREEN << "construction of sph based system" << RESET << std::endl;
    std::memset(&m_params, 0, sizeof(m_params));
    m_params.gasStiffness = 800;
    m_params.restDensity = 1000;
    m_params.particleRadius = 0.02;
    m_params.timestep = 1E-3;
    m_params.viscosity = 0.005;
    m_params.surfaceTension = 0.01;
    m_params.gravity = make_SVec3(0., -9.81, 0.);
    m_params.interactionRadius = 0.0457;
    m_params.particleMass = 0.5 * powf(m_params.interactionRadius, 3) * m_params.restDensity;
    m_params.beta = 450.0;
    m_params.soundSpeed = defaultSoundSpeed();
    m_params.worldOrigin = make_SVec3(-1.1, -1.1, -1.1);
    m_params.gridSize = make_uint3(64, 64, 64);
    const SReal h = m_params.interactionRadius;
    m_params.cellSize = make_SVec3(h, h, h);
    m_params.numCells = m_params.gridSize.x * m_params.gridSize.y * m_params.gridSize.z;
    kernelFactors(m_params, 0);
    _initialize();
    m_numParticles = 0;
}

SPH::SPH(SphSimParams params)
    : m_params(params), m_gridSortBits(32), m_pos(nullptr), m_vel(nullptr), m_density(nullptr), m_pressure(nullptr),
      m_forces(nullptr), m_colors(nullptr), m_numParticles(0), m_hostCapacity(0), m_bi(nullptr), m_vbi(nullptr),
      m_num_boundaries(0), m_ctx(nullptr), m_ctxCapacity(0), m_hostDirty(true), m_deviceNewer(false),
      m_boundariesPending(false), m_eagerSync(false), m_initialized(false), m_cfl(false), m_cflLambda(0.4f), m_asyncReadback(false),
      m_framesInFlight(0), m_frame(nullptr), m_frameCount(0), m_frameStep(0), m_frameIsCurrent(false)
{
    kernelFactors(m_params, 2);
    _initialize();
    m_numParticles = 0;
}

SPH::~SPH()
{
    releaseContext();
    std::free(m_pos); std::free(m_vel); std::free(m_density); std::free(m_pressure); std::free(m_forces);
    std::free(m_colors);
}

void SPH::fatal(const char *what) const
{
    std::fprintf(stderr, "Nereus: %s: %s\n", what, nrs_last_error());
    std::exit(EXIT_FAILURE); // same policy as the reference's checkCudaErrors
}

int SPH::solverKind() const { return NRS_SOLVER_SESPH; }

void SPH::growHost(SUint capacity)
{
    if (capacity <= m_hostCapacity) return;
    auto grow = [&](SReal *&a, size_t per) {
        SReal *n = (SReal *)std::realloc(a, sizeof(SReal) * per * capacity);
        if (!n) { std::fprintf(stderr, "Nereus: out of host memory\n"); std::exit(EXIT_FAILURE); }
        std::memset(n + per * m_hostCapacity, 0, sizeof(SReal) * per * (capacity - m_hostCapacity));
        a = n;
    };
    grow(m_pos, 4); grow(m_vel, 4); grow(m_forces, 4); grow(m_colors, 4);
    grow(m_density, 1); grow(m_pressure, 1);
    m_hostCapacity = capacity;
}

void SPH::_initialize()
{
    // The reference allocates here every time it is called (constructor, derived constructor, main.cpp:534)
    // and leaks the earlier allocations; this one only makes sure storage exists.  The device context is
    // created lazily so that it sees the final parameters and solver kind.
    if (!m_initialized) {
        growHost(envCapacity());
        m_initialized = true;
    }
}

void SPH::_finalize() { synchronize(); }

void SPH::_initializeGrid()
{
    // cell tables are (re)allocated by the device library when it sees a new numCells (nrs_set_params)
}

void SPH::reserveParticles(SUint capacity)
{
    growHost(capacity);
    if (m_ctx && capacity > m_ctxCapacity) {
        pullDeviceToHost();
        releaseContext();
        m_hostDirty = true;
    }
}

void SPH::releaseContext()
{
    if (m_ctx) {
        nrs_destroy(m_ctx); // also frees the pinned frames
        m_ctx = nullptr;
        m_ctxCapacity = 0;
        m_framesInFlight = 0;
        m_frame = nullptr;
        m_frameIsCurrent = false;
        if (m_num_boundaries) m_boundariesPending = true;
    }
}

void SPH::ensureContext()
{
    if (m_ctx && m_ctxCapacity >= m_numParticles) return;
    if (m_ctx) { // capacity exceeded: save the state and rebuild a larger context
        pullDeviceToHost();
        releaseContext();
        m_hostDirty = true;
    }
    nrs_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.device = -1;
    cfg.solver = solverKind();
    cfg.precision = (int)sizeof(SReal) * 8;
    cfg.kernel_set = KERNEL_SET;
    cfg.surface_tension = USE_SURFACE_TENSION;
    cfg.flags = std::getenv("NEREUS_REFERENCE_ORDER") ? NRS_FLAG_REFERENCE_ORDER : 0;
    cfg.capacity = std::max<SUint>(m_hostCapacity, m_numParticles);
    if (nrs_create(&cfg, &m_params, &m_ctx) != NRS_OK) fatal("nrs_create");
    m_ctxCapacity = (SUint)cfg.capacity;
    if (m_boundariesPending && m_bi && m_vbi && m_num_boundaries) {
        if (nrs_set_boundaries(m_ctx, m_bi, m_vbi, m_num_boundaries, 0) != NRS_OK) fatal("nrs_set_boundaries");
        m_boundariesPending = false;
    }
}

nrs_ctx *SPH::deviceContext()
{
    ensureContext();
    return m_ctx;
}

void SPH::pushHostToDevice()
{
    if (!m_hostDirty) return;
    if (nrs_set_num_particles(m_ctx, 0) != NRS_OK) fatal("nrs_set_num_particles");
    if (m_numParticles &&
        nrs_upload_particles(m_ctx, m_pos, m_vel, m_pressure, 0, m_numParticles) != NRS_OK)
        fatal("nrs_upload_particles");
    m_hostDirty = false;
    m_deviceNewer = false;
}

void SPH::pullDeviceToHost() const
{
    if (!m_deviceNewer || !m_ctx) return;
    const bool wantPressure = solverKind() == NRS_SOLVER_IISPH;
    if (nrs_download(m_ctx, m_pos, m_vel, wantPressure ? m_pressure : nullptr) != NRS_OK) fatal("nrs_download");
    m_deviceNewer = false;
}

void SPH::synchronize() const
{
    if (m_ctx && nrs_synchronize(m_ctx) != NRS_OK) fatal("nrs_synchronize");
}

void SPH::update()
{
    if (m_numParticles == 0) return;
    ensureContext();
    pushHostToDevice();                                                   // only if the host side changed
    if (m_cfl) { // sph.cpp:217-231 (disabled there): newDeltat = lambda * (ir / |v|max)
        double vmax = 0.0;
        if (nrs_max_velocity(m_ctx, &vmax) != NRS_OK) fatal("nrs_max_velocity");
        if (vmax > 0.0) m_params.timestep = m_cflLambda * (m_params.interactionRadius / (SReal)vmax);
    }
    if (nrs_set_params(m_ctx, &m_params) != NRS_OK) fatal("nrs_set_params"); // setParameters, every step
    if (nrs_step(m_ctx, 1) != NRS_OK) fatal("nrs_step");
    m_deviceNewer = true;
    m_frameIsCurrent = false;
    if (m_asyncReadback) {
        if (m_framesInFlight == 2) collectFrames(false); // the library would wait for the oldest anyway: keep it
        if (nrs_snapshot_begin(m_ctx, 0) != NRS_OK) fatal("nrs_snapshot_begin");
        ++m_framesInFlight;
    }
    if (m_eagerSync) pullDeviceToHost();
}

void SPH::setAsyncReadback(bool on)
{
    if (!on && m_ctx) collectFrames(true);
    m_asyncReadback = on;
}

// take every finished snapshot out of the library (all of them, blocking, if waitForAll; else at least the oldest when
// two are in flight or none has ever arrived)
void SPH::collectFrames(bool waitForAll) const
{
    while (m_framesInFlight > 0) {
        const bool mustWait = waitForAll || m_framesInFlight == 2 || m_frame == nullptr;
        const void *p = nullptr;
        uint64_t n = 0, step = 0;
        const int rc = nrs_snapshot_wait(m_ctx, mustWait ? 1 : 0, &p, nullptr, &n, &step);
        if (rc == NRS_E_NOTREADY) break;
        if (rc != NRS_OK) fatal("nrs_snapshot_wait");
        --m_framesInFlight;
        m_frame = (const SReal *)p;
        m_frameCount = (SUint)n;
        m_frameStep = step;
        m_frameIsCurrent = (m_framesInFlight == 0); // the newest begun snapshot was taken right after the last update()
    }
}

const SReal *SPH::latestFrame(SUint *numParticles, unsigned long long *step)
{
    if (!m_asyncReadback || !m_ctx) { // no snapshots: the synchronous path
        pullDeviceToHost();
        if (numParticles) *numParticles = m_numParticles;
        if (step) *step = 0;
        return m_pos;
    }
    collectFrames(false);
    if (numParticles) *numParticles = m_frameCount;
    if (step) *step = m_frameStep;
    return m_frame;
}

SReal *&SPH::getPos() { pullDeviceToHost(); m_hostDirty = true; return m_pos; }
SReal *&SPH::getVel() { pullDeviceToHost(); m_hostDirty = true; return m_vel; }
SReal *&SPH::getCol() { return m_colors; }
SReal *SPH::getHostPos() const
{
    if (m_asyncReadback && m_ctx && m_deviceNewer && m_framesInFlight > 0) collectFrames(true);
    if (m_asyncReadback && m_frameIsCurrent && m_deviceNewer && m_frame) return const_cast<SReal *>(m_frame);
    pullDeviceToHost();
    return m_pos;
}
SReal *SPH::getHostVel() const { pullDeviceToHost(); return m_vel; }
SReal *SPH::getHostPressure() const { pullDeviceToHost(); return m_pressure; }
SReal *SPH::getHostCol() const { return m_colors; }

std::pair<SVec3, SVec3> SPH::computeGridMinMax() const
{
    // BBMin/BBMax of the reference run three reductions per bound on the device copy; the values are the
    // per-axis extrema of the boundary positions, computed here on the caller's array.
    SVec3 lo = make_SVec3(0, 0, 0), hi = make_SVec3(0, 0, 0);
    if (m_bi && m_num_boundaries) {
        lo = hi = make_SVec3(m_bi[0], m_bi[1], m_bi[2]);
        for (SUint i = 1; i < m_num_boundaries; ++i) {
            const SReal *q = m_bi + 4 * (size_t)i;
            if (q[0] < lo.x) lo.x = q[0];
            if (q[1] < lo.y) lo.y = q[1];
            if (q[2] < lo.z) lo.z = q[2];
            if (hi.x < q[0]) hi.x = q[0];
            if (hi.y < q[1]) hi.y = q[1];
            if (hi.z < q[2]) hi.z = q[2];
        }
    }
    return std::make_pair(lo, hi);
}

void SPH::updateGrid()
{
    const std::pair<SVec3, SVec3> bb = computeGridMinMax();
    m_params.worldOrigin = make_SVec3(bb.first.x - 0.1, bb.first.y - 0.1, bb.first.z - 0.1);
    const SUint nx = std::ceil((bb.second.x - bb.first.x + 0.1) / m_params.interactionRadius);
    const SUint ny = std::ceil((bb.second.y - bb.first.y + 0.1) / m_params.interactionRadius);
    const SUint nz = std::ceil((bb.second.z - bb.first.z + 0.1) / m_params.interactionRadius);
    m_params.gridSize = make_uint3(nextPow2(nx), nextPow2(ny), nextPow2(nz));
    m_params.numCells = m_params.gridSize.x * m_params.gridSize.y * m_params.gridSize.z;
    _initializeGrid();
    if (m_ctx && nrs_set_params(m_ctx, &m_params) != NRS_OK) fatal("nrs_set_params");
}

void SPH::addNewParticle(SVec4 p, SVec4 v)
{
    pullDeviceToHost(); // appending to stale host arrays would lose the device state
    if (m_numParticles >= m_hostCapacity) growHost(std::max<SUint>(2 * m_hostCapacity, 1024));
    const size_t i = m_numParticles;
    m_pos[4 * i + 0] = p.x; m_pos[4 * i + 1] = p.y; m_pos[4 * i + 2] = p.z; m_pos[4 * i + 3] = p.w;
    m_vel[4 * i + 0] = v.x; m_vel[4 * i + 1] = v.y; m_vel[4 * i + 2] = v.z; m_vel[4 * i + 3] = v.w;
    m_density[i] = 0.0;
    m_pressure[i] = 0.0;
    for (int c = 0; c < 4; ++c) m_forces[4 * i + c] = 0.0;
    m_colors[4 * i + 0] = 1.0; m_colors[4 * i + 1] = 0.0; m_colors[4 * i + 2] = 0.0; m_colors[4 * i + 3] = 1.0;
    m_numParticles += 1;
    m_hostDirty = true;
}

void SPH::generateParticleCube(SVec4 center, SVec4 size, SVec4 vel)
{
    const SReal step = m_params.interactionRadius - 0.005f;
    for (SReal x = center.x - size.x / 2.0; x <= center.x + size.x / 2.0; x += step)
        for (SReal y = center.y - size.y / 2.0; y <= center.y + size.y / 2.0; y += step)
            for (SReal z = center.z - size.z / 2.0; z <= center.z + size.z / 2.0; z += step)
                addNewParticle(make_SVec4(x, y, z, 1.0), vel);
    std::cout << "There were " << m_numParticles << " particles generated." << std::endl;
}

namespace {
struct CkptHeader {
    char magic[8];
    uint32_t realBytes, solver, paramBytes, reserved;
    uint64_t n;
};
} // namespace

bool SPH::saveState(const char *path) const
{
    pullDeviceToHost();
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    CkptHeader h;
    std::memcpy(h.magic, "NRSCKPT1", 8);
    h.realBytes = sizeof(SReal); h.solver = (uint32_t)solverKind(); h.paramBytes = sizeof(SphSimParams); h.reserved = 0;
    h.n = m_numParticles;
    bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(&m_params, sizeof(m_params), 1, f) == 1;
    const size_t n = m_numParticles;
    ok = ok && (n == 0 || (std::fwrite(m_pos, sizeof(SReal) * 4, n, f) == n && std::fwrite(m_vel, sizeof(SReal) * 4, n, f) == n &&
                           std::fwrite(m_pressure, sizeof(SReal), n, f) == n));
    return std::fclose(f) == 0 && ok;
}

bool SPH::loadState(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    CkptHeader h;
    bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "NRSCKPT1", 8) == 0 && h.realBytes == sizeof(SReal) &&
              h.paramBytes == sizeof(SphSimParams) && h.solver == (uint32_t)solverKind() && h.n < (1ull << 27);
    SphSimParams p;
    ok = ok && std::fread(&p, sizeof(p), 1, f) == 1;
    if (ok) {
        const size_t n = (size_t)h.n;
        growHost((SUint)std::max<size_t>(n, 1));
        ok = n == 0 || (std::fread(m_pos, sizeof(SReal) * 4, n, f) == n && std::fread(m_vel, sizeof(SReal) * 4, n, f) == n &&
                        std::fread(m_pressure, sizeof(SReal), n, f) == n);
        if (ok) {
            // keep the grid of the running solver if boundaries already fixed it; take everything else from the file
            m_params = p;
            m_numParticles = (SUint)n;
            m_hostDirty = true;
            m_deviceNewer = false;
        }
    }
    std::fclose(f);
    return ok;
}

void SPH::updateGpuBoundaries(SUint nb_boundary_spheres)
{
    m_num_boundaries = nb_boundary_spheres ? nb_boundary_spheres : m_num_boundaries;
    if (!m_bi || !m_vbi || !m_num_boundaries) {
        std::fprintf(stderr, "Nereus: updateGpuBoundaries called without setBi/setVbi/setNumBoundaries\n");
        std::exit(EXIT_FAILURE);
    }
    ensureContext();
    // grid from the boundary AABB (updateGrid), then hash / sort / cell ranges of the boundary particles
    if (nrs_set_params(m_ctx, &m_params) != NRS_OK) fatal("nrs_set_params");
    if (nrs_set_boundaries(m_ctx, m_bi, m_vbi, m_num_boundaries, 1) != NRS_OK) fatal("nrs_set_boundaries");
    if (nrs_get_params(m_ctx, &m_params) != NRS_OK) fatal("nrs_get_params");
    m_boundariesPending = false;
    std::cout << "boundaries updated !" << std::endl;
}

NEREUS_NAMESPACE_END
