// sph_boundary_particles/ss.h — box sampler with the call signature main.cpp:545 uses.
//
// The reference takes this from the git submodule external/sph_boundary_particles, which is NOT vendored
// (empty directory, commit unknown): its exact sampling pattern is unknown, so this is our own sampler
// behind the same signature — PARITY UNPINNED at this boundary.  The solver itself
// is insensitive to how boundary particles were produced: they are plain input arrays.
#pragma once
#include <cmath>
#include <vector>

#include "common.h"

namespace sample_spheres {
namespace ss {

// Samples the six faces of the axis-aligned box [origin, origin+size] with one layer of particles on a
// regular lattice of pitch `radius` (the value main.cpp passes is the particle radius, 0.02); points on
// shared edges/corners are emitted once.  Appends xyz1 to `out`.
inline void sampleBox(std::vector<SVec4> &out, SVec3 origin, SVec3 size, double radius)
{
    const long nx = std::lround(size.x / radius), ny = std::lround(size.y / radius), nz = std::lround(size.z / radius);
    auto emit = [&](long i, long j, long k) {
        out.push_back(make_SVec4((SReal)(origin.x + i * radius), (SReal)(origin.y + j * radius),
                                 (SReal)(origin.z + k * radius), (SReal)1.0));
    };
    for (long i = 0; i <= nx; ++i)
        for (long k = 0; k <= nz; ++k) { emit(i, 0, k); emit(i, ny, k); }
    for (long j = 1; j < ny; ++j)
        for (long k = 0; k <= nz; ++k) { emit(0, j, k); emit(nx, j, k); }
    for (long i = 1; i < nx; ++i)
        for (long j = 1; j < ny; ++j) { emit(i, j, 0); emit(i, j, nz); }
}

} // namespace ss
} // namespace sample_spheres
