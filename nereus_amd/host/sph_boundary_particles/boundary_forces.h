// sph_boundary_particles/boundary_forces.h — Akinci boundary volumes with the signature main.cpp:546 uses.
// Our own implementation (the reference's library is not vendored; PARITY UNPINNED, see ss.h).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "nereus_hip.h"

namespace sample_spheres {
namespace boundary_forces {

// vbi[b] = 1 / sum_k W_poly6(|x_b - x_k|, h) over all boundary particles k within h of b (b itself included)
// (Akinci et al. 2012, eq. 4).  Uniform-grid search on the HOST, O(n * neighbours): kept as the cross-check of the device
// version below (tests/test_host_class.py), not called by the product path.
inline void getVbiHost(std::vector<SReal> &vbi, std::vector<SVec4> &bi, SReal h)
{
    const size_t n = bi.size();
    vbi.assign(n, (SReal)0);
    if (!n) return;
    const double hh = (double)h, h2 = hh * hh, kpoly = 315.0 / (64.0 * M_PI * std::pow(hh, 9));
    double lo[3] = {bi[0].x, bi[0].y, bi[0].z};
    for (const SVec4 &p : bi) { lo[0] = std::min<double>(lo[0], p.x); lo[1] = std::min<double>(lo[1], p.y); lo[2] = std::min<double>(lo[2], p.z); }
    auto cellOf = [&](const SVec4 &p, int64_t c[3]) {
        c[0] = (int64_t)std::floor((p.x - lo[0]) / hh); c[1] = (int64_t)std::floor((p.y - lo[1]) / hh); c[2] = (int64_t)std::floor((p.z - lo[2]) / hh);
    };
    auto keyOf = [](const int64_t c[3]) { return (uint64_t)((c[0] + 1) * 2097152ll * 2097152ll + (c[1] + 1) * 2097152ll + (c[2] + 1)); };
    std::vector<std::pair<uint64_t, uint32_t>> cells(n);
    for (size_t i = 0; i < n; ++i) { int64_t c[3]; cellOf(bi[i], c); cells[i] = std::make_pair(keyOf(c), (uint32_t)i); }
    std::sort(cells.begin(), cells.end());
    for (size_t i = 0; i < n; ++i) {
        int64_t c[3]; cellOf(bi[i], c);
        double acc = 0.0;
        for (int64_t dx = -1; dx <= 1; ++dx) for (int64_t dy = -1; dy <= 1; ++dy) for (int64_t dz = -1; dz <= 1; ++dz) {
            const int64_t q[3] = {c[0] + dx, c[1] + dy, c[2] + dz};
            const uint64_t key = keyOf(q);
            auto it = std::lower_bound(cells.begin(), cells.end(), std::make_pair(key, (uint32_t)0));
            for (; it != cells.end() && it->first == key; ++it) {
                const SVec4 &o = bi[it->second];
                const double rx = bi[i].x - o.x, ry = bi[i].y - o.y, rz = bi[i].z - o.z, r2 = rx * rx + ry * ry + rz * rz;
                if (r2 < h2) acc += kpoly * (h2 - r2) * (h2 - r2) * (h2 - r2);
            }
        }
        vbi[i] = (SReal)(1.0 / acc);
    }
}

// The signature main.cpp:546 calls.  Runs on the device (nrs_boundary_volumes: the solver's own hash / sort / cell-range /
// 27-cell gather machinery); like every call of the reference's launcher layer it is fatal on error.
inline void getVbi(std::vector<SReal> &vbi, std::vector<SVec4> &bi, SReal h)
{
    vbi.assign(bi.size(), (SReal)0);
    if (bi.empty()) return;
    const int rc = nrs_boundary_volumes(-1, (int)(8 * sizeof(SReal)), bi.data(), (uint64_t)bi.size(), (double)h, vbi.data());
    if (rc != 0) {
        std::fprintf(stderr, "getVbi: nrs_boundary_volumes failed (%d): %s\n", rc, nrs_last_error());
        std::exit(EXIT_FAILURE);
    }
}

} // namespace boundary_forces
} // namespace sample_spheres
