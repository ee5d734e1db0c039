// nrs_kernels_iisph.h — list-driven kernels of the IISPH chain (Muller kernels).
//
// The reference runs 6 + 2L neighbourhood sweeps per IISPH step, each re-walking the 27 cells.
// Here the neighbourhood is scanned ONCE per step (k_density_tiled<..., WIDE>, nrs_kernels_tiled.h): the hit lists
// it publishes keep the particle itself and every candidate with length(r)^2 <= h^2 — the widest cut-off any loop
// of the chain implies — tagged with the neighbour-cell number.  The kernels below only walk those lists, applying
// each reference loop's own exclusions (j != self, j != thread id (Q5), length(r) < h) and forming the sums in the
// reference's order (per-cell partial sums where the reference has them), so they are bit-identical to the
// reference-order kernels of nrs_kernels_ref.h.  No LDS, no cell-table traffic.
//
// The boundary loops of computePressure / computePressureForce run over an index range that mixes fluid and boundary
// indices; they are kept as cell walks with the reference's bounds, next to the list-driven fluid part.
// Not list-driven: Monaghan kernels (support 2h: no list cut-off is exact) and particles whose list overflowed
// (per-particle flag: the reference-order cell walk is used for that particle).
#pragma once
#include "nrs_kernels_tiled.h"

// waves per SIMD the fp32 list kernels are compiled for (measured at config C3, 4.1 M particles): displacement
// unbounded (92 VGPRs, 5 waves) 0.389 ms, 6 waves 0.366, 7 waves 0.351; advection 7 -> 8 waves 0.239 -> 0.234;
// pressure (2 iterations incl. the rest of the solve) unbounded 1.093, 5 waves 1.055, 6 waves 1.178 (spills)
#ifndef IISPH_DISP_WAVES
#define IISPH_DISP_WAVES 7
#endif
#ifndef IISPH_ADV_WAVES
#define IISPH_ADV_WAVES 8
#endif
#ifndef IISPH_PRES_WAVES
#define IISPH_PRES_WAVES 5
#endif

namespace nrs {


// Walk the merged hit lists; f(j, isBoundary, newPartial) with newPartial = first hit of a (cell, kind) group.
template <typename F> NRS_DEV void for_each_hit(const uint32_t *lbase, uint32_t lstride, HitCounts hc, F &&f)
{
    HitMerge it(lbase, lstride, hc);
    uint32_t j, key, prevKey = 0xffffffffu;
    bool isB;
    while (it.next(j, isB, key)) {
        const bool fresh = key != prevKey;
        prevKey = key;
        f(j, isB, fresh);
    }
}

// ---- computeDisplacementFactor (sph_kernel_impl.cuh:851-963) -----------------------------------------------
template <typename R, int KSET, bool SURF, bool HAS_B>
NRS_DEV void displacement_lists_particle(const Params<R> &P, const GridView<R> &G, const IisphArrays<R> &I, const HitBuffer &hb,
                                         const typename Vec4T<R>::type *__restrict__ sPos, const typename Vec4T<R>::type *__restrict__ sVel,
                                         const R *__restrict__ sDens, const R *__restrict__ sPres, uint32_t i)
{
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const V3<R> vel1 = xyz<R>(sVel[i]);
    const R pres = (R)0.0;
    const R dens = sDens[i];
    const R kpg = P.kpoly_grad, pm = P.particleMass, ir = P.interactionRadius, rd = P.restDensity, dt = P.timestep;
    const HitCounts hc = unpack_counts(hb.counts[i]);
    ForceAcc<R> A;
    V3<R> df = mk3<R>(0, 0, 0), db = mk3<R>(0, 0, 0);
    if (hc.over) {
        A = gather_forces<R, KSET, SURF, HAS_B>(P, G, i, pos1, vel1, dens, pres, sPos, sVel, sDens, sPres);
    } else {
        A = forces_from_hits<R, KSET, SURF, HAS_B, true>(P, G, sPos, sVel, sDens, sPres, pos1, vel1, dens, pres, hb.hits + i, hb.stride, hc, i);
    }
    V3<R> fvisc = 2.0 * A.fvisc;
    fvisc = (pm * P.viscosity) * fvisc;
    const V3<R> fgrav = pm * mk3<R>(P.gravity[0], P.gravity[1], P.gravity[2]);
    const V3<R> force_adv = fvisc + A.fsurf + A.fbound + fgrav;
    const V3<R> vel_adv = vel1 + dt * (force_adv / pm);
    I.forcesAdv[i] = mk4<R>(force_adv, (R)0.0);
    I.velAdv[i] = mk4<R>(vel_adv, (R)0.0);
    watch_finite<R>(I.nonFinite, vel_adv);
    // a particle with a NaN / inf / absurd coordinate (a caller's bug) is in nobody's hit list, while the reference's loops without
    // a cut-off multiply its NaN distance into their sums: such a step is repeated in reference order as well
    watch_finite<R>(I.nonFinite, pos1);
    if (hc.over) { // per-cell walk of the reference-order kernel for this particle
        const I3 gp = calcGridPos<R>(P, pos1);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                    V3<R> res = mk3<R>(0, 0, 0);
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            if (j == i) continue;
                            const V3<R> d = pos1 - xyz<R>(sPos[j]);
                            if (length(d) < ir) res = res - ((pm / (dens * dens)) * W_grad<R, KSET>(d, ir, kpg));
                        }
                    }
                    df = df + res;
                    if (HAS_B) {
                        V3<R> rb = mk3<R>(0, 0, 0);
                        const uint32_t sb = G.bCellStart[h];
                        if (sb != CELL_EMPTY) {
                            const uint32_t e = G.bCellEnd[h];
                            for (uint32_t j = sb; j < e; ++j) {
                                const typename Vec4T<R>::type b = G.sB[j];
                                const V3<R> d = pos1 - xyz<R>(b);
                                if (length(d) < ir) rb = rb - (((rd * b.w) / (dens * dens)) * W_grad<R, KSET>(d, ir, kpg));
                            }
                        }
                        db = db + rb;
                    }
                }
    } else if (!HAS_B || hc.nb == 0) { // no boundary hits: the fluid entries alone, batched
        V3<R> part = mk3<R>(0, 0, 0);
        uint32_t prevTag = 0xffffffffu;
        walk_fluid_batched(hb.hits + i, hb.stride, hc.nf, [&](uint32_t j) { return sPos[j]; },
                           [&](uint32_t j, uint32_t tag, const typename Vec4T<R>::type &q) {
                               if (tag != prevTag) { df = df + part; part = mk3<R>(0, 0, 0); prevTag = tag; }
                               if (j == i) return;
                               const V3<R> d = pos1 - xyz<R>(q);
                               const float rlen = length_listed(dot(d, d));
                               if (rlen < ir) part = part - ((pm / (dens * dens)) * Wdefault_grad_len<R>(d, rlen, ir, kpg));
                           });
        df = df + part;
    } else {
        V3<R> part = mk3<R>(0, 0, 0);
        bool partB = false;
        for_each_hit(hb.hits + i, hb.stride, hc, [&](uint32_t j, bool isB, bool fresh) {
            if (fresh) {
                if (partB) db = db + part; else df = df + part;
                part = mk3<R>(0, 0, 0);
                partB = isB;
            }
            if (HAS_B && isB) {
                const typename Vec4T<R>::type b = G.sB[j];
                const V3<R> d = pos1 - xyz<R>(b);
                const R psi = rd * b.w;
                if (length(d) < ir) part = part - ((psi / (dens * dens)) * W_grad<R, KSET>(d, ir, kpg));
            } else if (j != i) {
                const V3<R> d = pos1 - xyz<R>(sPos[j]);
                if (length(d) < ir) part = part - ((pm / (dens * dens)) * W_grad<R, KSET>(d, ir, kpg));
            }
        });
        if (partB) db = db + part; else df = df + part;
    }
    I.diiF[i] = mk4<R>(df, (R)0.0);
    I.diiB[i] = mk4<R>(db, (R)0.0);
    I.diiSum[i] = mk4<R>(df + db, (R)0.0); // the same sum computePressure forms per neighbour (sph_kernel_impl.cuh:1420)
    watch_finite<R>(I.nonFinite, df + db);
}
// one sorted slot per thread — or, with WALLS (see k_pressure_lists), wall workgroups over the step's wall list + interior workgroups
// compiled without the boundary code
#define NRS_IISPH_WALL_SPLIT(PARTICLE_CALL_B, PARTICLE_CALL_NOB, PARTICLE_CALL_ANY)                                                    \
    uint32_t block = blockIdx.x, blocks = gridDim.x;                                                                                  \
    if (WALLS) {                                                                                                                      \
        if (block < wallBlocks) {                                                                                                     \
            const uint32_t count = *wl.count;                                                                                         \
            for (uint32_t t = block * BLOCK + threadIdx.x; t < count; t += wallBlocks * BLOCK) { const uint32_t i = wl.list[t]; PARTICLE_CALL_B; } \
            return;                                                                                                                   \
        }                                                                                                                             \
        block -= wallBlocks; blocks -= wallBlocks;                                                                                    \
    }                                                                                                                                 \
    const uint32_t i = xcd_tile(block, blocks) * BLOCK + threadIdx.x;                                                                 \
    if (i >= n) return;                                                                                                               \
    if (WALLS) {                                                                                                                      \
        if (hb.counts[i] & COUNTS_DEFERRED) return;                                                                                   \
        PARTICLE_CALL_NOB;                                                                                                            \
    } else {                                                                                                                          \
        PARTICLE_CALL_ANY;                                                                                                            \
    }
template <typename R, int KSET, bool SURF, bool HAS_B, bool WALLS = false>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? IISPH_DISP_WAVES : 1)) void k_displacement_lists(Params<R> P, GridView<R> G, IisphArrays<R> I, HitBuffer hb,
                                                              const typename Vec4T<R>::type *__restrict__ sPos,
                                                              const typename Vec4T<R>::type *__restrict__ sVel,
                                                              const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                              uint32_t n, WallList wl, uint32_t wallBlocks)
{
    NRS_IISPH_WALL_SPLIT((displacement_lists_particle<R, KSET, SURF, true>(P, G, I, hb, sPos, sVel, sDens, sPres, i)),
                         (displacement_lists_particle<R, KSET, SURF, false>(P, G, I, hb, sPos, sVel, sDens, sPres, i)),
                         (displacement_lists_particle<R, KSET, SURF, HAS_B>(P, G, I, hb, sPos, sVel, sDens, sPres, i)))
}

// ---- computeAdvectionFactor (sph_kernel_impl.cuh:1114-1218) --------------------------------------------------
template <typename R, int KSET, bool HAS_B>
NRS_DEV void advection_lists_particle(const Params<R> &P, const GridView<R> &G, const IisphArrays<R> &I, const HitBuffer &hb,
                                      const typename Vec4T<R>::type *__restrict__ sPos, const typename Vec4T<R>::type *__restrict__ sVel,
                                      const R *__restrict__ sDens, const R *__restrict__ sPres, uint32_t i)
{
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const V3<R> vel1 = xyz<R>(sVel[i]);
    const V3<R> velAdv1 = xyz<R>(I.velAdv[i]);
    const R dens = sDens[i];
    const V3<R> diif = xyz<R>(I.diiF[i]);
    const V3<R> diib = xyz<R>(I.diiB[i]);
    const R kpg = P.kpoly_grad, pm = P.particleMass, ir = P.interactionRadius, rd = P.restDensity, dt = P.timestep;
    const HitCounts hc = unpack_counts(hb.counts[i]);
    R rho_advf = (R)0.0, rho_advb = (R)0.0, aii = (R)0.0;
    if (hc.over) {
        const I3 gp = calcGridPos<R>(P, pos1);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                    R res = (R)0.0;
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            if (j == i) continue;
                            const V3<R> vij = velAdv1 - xyz<R>(I.velAdv[j]);
                            const V3<R> d = pos1 - xyz<R>(sPos[j]);
                            if (length(d) < ir) res += (dt * pm * dot(vij, W_grad<R, KSET>(d, ir, kpg)));
                        }
                    }
                    rho_advf += res;
                    if (HAS_B) {
                        R rb = (R)0.0;
                        const uint32_t sb = G.bCellStart[h];
                        if (sb != CELL_EMPTY) {
                            const uint32_t e = G.bCellEnd[h];
                            for (uint32_t j = sb; j < e; ++j) {
                                const typename Vec4T<R>::type b = G.sB[j];
                                const V3<R> d = pos1 - xyz<R>(b);
                                rb += (dt * (rd * b.w) * dot(vel1, W_grad<R, KSET>(d, ir, kpg)));
                            }
                        }
                        rho_advb += rb;
                    }
                }
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                    R res = (R)0.0;
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            if (j == i) continue;
                            const V3<R> d = pos1 - xyz<R>(sPos[j]);
                            const R dpi = (pm) / (dens * dens);
                            const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                            const V3<R> dji = dpi * grad;
                            res += (pm * dot((diif + diib) - dji, grad));
                        }
                    }
                    aii += res;
                    if (HAS_B) {
                        R rb = (R)0.0;
                        const uint32_t sb = G.bCellStart[h];
                        if (sb != CELL_EMPTY) {
                            const uint32_t e = G.bCellEnd[h];
                            for (uint32_t j = sb; j < e; ++j) {
                                const typename Vec4T<R>::type b = G.sB[j];
                                const V3<R> d = pos1 - xyz<R>(b);
                                const R psi = rd * b.w;
                                const R dpi = (pm) / (dens * dens);
                                const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                                const V3<R> dji = dpi * grad;
                                rb += psi * dot((diif + diib) - dji, grad);
                            }
                        }
                        aii += rb;
                    }
                }
    } else if (!HAS_B || hc.nb == 0) {
        // no boundary hits (all but the particles next to a wall): both sums in ONE batched walk of the fluid entries — the same entries in
        // the same order feed rho_adv (one partial per cell, `length < h` as its loop tests) and a_ii (one partial per cell, no test), and
        // W_grad of a hit is formed once for both
        R part = (R)0.0, pa = (R)0.0;
        uint32_t prevTag = 0xffffffffu;
        const R dpi = (pm) / (dens * dens);
        struct Nb { typename Vec4T<R>::type q, va; };
        walk_fluid_batched(hb.hits + i, hb.stride, hc.nf, [&](uint32_t j) { return Nb{sPos[j], I.velAdv[j]}; },
                           [&](uint32_t j, uint32_t tag, const Nb &nb) {
                               if (tag != prevTag) { rho_advf += part; part = (R)0.0; aii += pa; pa = (R)0.0; prevTag = tag; }
                               if (j == i) return;
                               const V3<R> d = pos1 - xyz<R>(nb.q);
                               const float rlen = length_listed(dot(d, d));
                               const V3<R> grad = Wdefault_grad_len<R>(d, rlen, ir, kpg);
                               if (rlen < ir) part += (dt * pm * dot(velAdv1 - xyz<R>(nb.va), grad));
                               pa += (pm * dot((diif + diib) - dpi * grad, grad));
                           });
        rho_advf += part;
        aii += pa;
    } else {
        // rho_adv: fluid partials go to rho_advf, boundary partials to rho_advb, one partial per cell
        R part = (R)0.0;
        bool partB = false;
        for_each_hit(hb.hits + i, hb.stride, hc, [&](uint32_t j, bool isB, bool fresh) {
            if (fresh) {
                if (partB) rho_advb += part; else rho_advf += part;
                part = (R)0.0;
                partB = isB;
            }
            if (HAS_B && isB) {
                const typename Vec4T<R>::type b = G.sB[j];
                const V3<R> d = pos1 - xyz<R>(b);
                const R psi = (rd * b.w);
                part += (dt * psi * dot(vel1, W_grad<R, KSET>(d, ir, kpg)));
            } else if (j != i) {
                const V3<R> vij = velAdv1 - xyz<R>(I.velAdv[j]);
                const V3<R> d = pos1 - xyz<R>(sPos[j]);
                if (length(d) < ir) part += (dt * pm * dot(vij, W_grad<R, KSET>(d, ir, kpg)));
            }
        });
        if (partB) rho_advb += part; else rho_advf += part;
        // a_ii: every (cell, kind) partial is added to the same accumulator
        R pa = (R)0.0;
        for_each_hit(hb.hits + i, hb.stride, hc, [&](uint32_t j, bool isB, bool fresh) {
            if (fresh) { aii += pa; pa = (R)0.0; }
            const R dpi = (pm) / (dens * dens);
            if (HAS_B && isB) {
                const typename Vec4T<R>::type b = G.sB[j];
                const V3<R> d = pos1 - xyz<R>(b);
                const R psi = rd * b.w;
                const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                const V3<R> dji = dpi * grad;
                pa += psi * dot((diif + diib) - dji, grad);
            } else if (j != i) {
                const V3<R> d = pos1 - xyz<R>(sPos[j]);
                const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                const V3<R> dji = dpi * grad;
                pa += (pm * dot((diif + diib) - dji, grad));
            }
        });
        aii += pa;
    }
    const R rho_adv = dens + (rho_advf + rho_advb);
    I.densAdv[i] = rho_adv;
    I.P_l[i] = (R)(0.5 * sPres[i]);
    I.aii[i] = aii;
}
template <typename R, int KSET, bool HAS_B, bool WALLS = false>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? IISPH_ADV_WAVES : 1)) void k_advection_lists(Params<R> P, GridView<R> G, IisphArrays<R> I, HitBuffer hb,
                                                           const typename Vec4T<R>::type *__restrict__ sPos,
                                                           const typename Vec4T<R>::type *__restrict__ sVel,
                                                           const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                           uint32_t n, WallList wl, uint32_t wallBlocks)
{
    NRS_IISPH_WALL_SPLIT((advection_lists_particle<R, KSET, true>(P, G, I, hb, sPos, sVel, sDens, sPres, i)),
                         (advection_lists_particle<R, KSET, false>(P, G, I, hb, sPos, sVel, sDens, sPres, i)),
                         (advection_lists_particle<R, KSET, HAS_B>(P, G, I, hb, sPos, sVel, sDens, sPres, i)))
}

// ---- computeSumDijPj (sph_kernel_impl.cuh:1259-1325): fluid neighbours only -----------------------------------
template <typename R, int KSET>
__global__ __launch_bounds__(BLOCK) void k_sumdij_lists(Params<R> P, GridView<R> G, IisphArrays<R> I, HitBuffer hb,
                                                        const typename Vec4T<R>::type *__restrict__ sPos,
                                                        const R *__restrict__ sDens, uint32_t n)
{
    const uint32_t i = xcd_tile(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const R ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad;
    HitCounts hc = unpack_counts(hb.counts[i]);
    V3<R> dijpj = mk3<R>(0, 0, 0);
    if (hc.over) {
        const I3 gp = calcGridPos<R>(P, pos1);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                    V3<R> res = mk3<R>(0, 0, 0);
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            if (j == i) continue;
                            const V3<R> d = pos1 - xyz<R>(sPos[j]);
                            const R densj = sDens[j];
                            res = res - ((pm / (densj * densj)) * I.P_l[j] * W_grad<R, KSET>(d, ir, kpg));
                        }
                    }
                    dijpj = dijpj + res;
                }
    } else {
        // boundary hits play no part here: the fluid entries alone, one partial sum per cell tag (the list is in cell order)
        V3<R> part = mk3<R>(0, 0, 0);
        uint32_t prevTag = 0xffffffffu;
        struct Nb { typename Vec4T<R>::type q; R pl, dn; };
        walk_fluid_batched(hb.hits + i, hb.stride, hc.nf, [&](uint32_t j) { return Nb{sPos[j], I.P_l[j], sDens[j]}; },
                           [&](uint32_t j, uint32_t tag, const Nb &nb) {
                               if (tag != prevTag) { dijpj = dijpj + part; part = mk3<R>(0, 0, 0); prevTag = tag; }
                               if (j != i) part = part - ((pm / (nb.dn * nb.dn)) * nb.pl * W_grad_listed<R, KSET>(pos1 - xyz<R>(nb.q), ir, kpg));
                           });
        dijpj = dijpj + part;
    }
    I.sumDij[i] = mk4<R>(dijpj, (R)0.0);
    watch_finite<R>(I.nonFinite, dijpj);
}

// ---- computePressure without boundary particles (sph_kernel_impl.cuh:1330-1492; Q5: skips j == inv[i], keeps self;
//      Q7: reads P_l, writes P_l_next) ----------------------------------------------------------------------------
template <typename R, int KSET, bool HAS_B>
NRS_DEV void pressure_lists_particle(const Params<R> &P, const GridView<R> &G, const IisphArrays<R> &I, const HitBuffer &hb,
                                     const typename Vec4T<R>::type *__restrict__ sPos, const R *__restrict__ sDens, R *__restrict__ sPres,
                                     uint32_t i)
{
    const uint32_t skip = I.inv[i];
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const R dens = sDens[i];
    R p_l = I.P_l[i];
    const R previous_p_l = p_l;
    const R rho_adv = I.densAdv[i];
    const R aii = I.aii[i];
    const V3<R> dijpj = xyz<R>(I.sumDij[i]);
    const R ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad, dt = P.timestep, rd = P.restDensity;
    R fsum = (R)0.0;
    R bsum = (R)0.0;
    const R dpi = pm / (dens * dens);
    HitCounts hc = unpack_counts(hb.counts[i]);
    if (HAS_B && hc.anyB) { // (anyB: the scan saw boundary particles in at least one of the 27 cells; none: nothing to add)
        // boundary part: its own accumulator, so it can run apart from the fluid part.  The loop bounds are the
        // reference's: from the FLUID cell start to the BOUNDARY cell end, over the boundary array.
        const I3 gp = calcGridPos<R>(P, pos1);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                    if (G.bCellStart[h] != CELL_EMPTY) {
                        const uint32_t s = G.cellStart[h], eB = G.bCellEnd[h];
                        for (uint32_t j = s; j < eB; ++j) {
                            const typename Vec4T<R>::type b = G.sB[j];
                            const V3<R> d = pos1 - xyz<R>(b);
                            const R psi = rd * b.w;
                            bsum += psi * dot(dijpj, W_grad<R, KSET>(d, ir, kpg));
                        }
                    }
                }
    }
    auto term = [&](uint32_t j) {
        const V3<R> d = pos1 - xyz<R>(sPos[j]);
        const R p_lj = I.P_l[j];
        const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
        const V3<R> dji = dpi * (grad);
        const V3<R> d_ji_pi = dji * p_lj;
        const V3<R> diij = xyz<R>(I.diiSum[j]); // = diiF[j] + diiB[j]
        const V3<R> sum_dijj = xyz<R>(I.sumDij[j]);
        fsum += pm * dot(dijpj - diij * p_lj - (sum_dijj - d_ji_pi), grad);
    };
    if (hc.over) {
        const I3 gp = calcGridPos<R>(P, pos1);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j)
                            if (j != skip) term(j);
                    }
                }
    } else {
        struct Nb { typename Vec4T<R>::type q, dii, sdj; R pl; };
        walk_fluid_batched(hb.hits + i, hb.stride, hc.nf, [&](uint32_t j) { return Nb{sPos[j], I.diiSum[j], I.sumDij[j], I.P_l[j]}; },
                           [&](uint32_t j, uint32_t, const Nb &nb) {
                               if (j == skip) return;
                               const V3<R> grad = W_grad_listed<R, KSET>(pos1 - xyz<R>(nb.q), ir, kpg);
                               const V3<R> d_ji_pi = (dpi * (grad)) * nb.pl;
                               fsum += pm * dot(dijpj - xyz<R>(nb.dii) * nb.pl - (xyz<R>(nb.sdj) - d_ji_pi), grad);
                           });
    }
    const R omega = (R)0.5;
    R rho_corr = rho_adv + fsum + bsum;
    const R dt2 = dt * dt;
    const R diagDt2 = aii * dt2;
    const R b = rd - rho_adv;
    if (fabs(diagDt2) > 1.1920928955078125e-07f /* FLT_EPSILON */)
        p_l = (R)((1.0 - omega) * previous_p_l + (omega / diagDt2) * (b - dt2 * (bsum + fsum)));
    else
        p_l = (R)0.0;
    watch_finite<R>(I.nonFinite, p_l); // (before the clamp, which would hide a NaN)
    const R p = (R)fmax((double)p_l, 0.0);
    p_l = p;
    rho_corr += aii * previous_p_l;
    I.P_l_next[i] = p_l;
    sPres[i] = p_l;
    I.densCorr[i] = rho_corr;
}
// WALLS (the step's wall list exists: the scan of this step ran with wall workgroups, k_density_tiled): the first `wallBlocks` workgroups
// walk the wall list with the boundary code — every lane a particle with boundary cells in its neighbourhood —, the others take one sorted
// slot each, skip the slots flagged COUNTS_DEFERRED and are compiled WITHOUT the boundary loops (whose Q6 bounds make a lane with boundary
// cells run 27 cell walks while the other 63 lanes of its wave wait).  Every particle is evaluated once, by the same operations.
template <typename R, int KSET, bool HAS_B, bool WALLS = false>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? IISPH_PRES_WAVES : 1)) void k_pressure_lists(Params<R> P, GridView<R> G, IisphArrays<R> I, HitBuffer hb,
                                                          const typename Vec4T<R>::type *__restrict__ sPos,
                                                          const R *__restrict__ sDens, R *__restrict__ sPres, uint32_t n,
                                                          WallList wl, uint32_t wallBlocks)
{
    uint32_t block = blockIdx.x, blocks = gridDim.x;
    if (WALLS) {
        if (block < wallBlocks) {
            const uint32_t count = *wl.count;
            for (uint32_t t = block * BLOCK + threadIdx.x; t < count; t += wallBlocks * BLOCK)
                pressure_lists_particle<R, KSET, true>(P, G, I, hb, sPos, sDens, sPres, wl.list[t]);
            return;
        }
        block -= wallBlocks; blocks -= wallBlocks;
    }
    const uint32_t i = xcd_tile(block, blocks) * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (WALLS) {
        if (hb.counts[i] & COUNTS_DEFERRED) return;
        pressure_lists_particle<R, KSET, false>(P, G, I, hb, sPos, sDens, sPres, i);
    } else {
        pressure_lists_particle<R, KSET, HAS_B>(P, G, I, hb, sPos, sDens, sPres, i);
    }
}

// ---- computePressureForce (sph_kernel_impl.cuh:1497-1620, same Q5/Q6).  One accumulator takes the fluid terms of a
//      cell and then that cell's boundary terms, so with boundary particles the list is consumed cell by cell ------
template <typename R, int KSET, bool HAS_B>
NRS_DEV void pforce_lists_particle(const Params<R> &P, const GridView<R> &G, const IisphArrays<R> &I, const HitBuffer &hb,
                                   const typename Vec4T<R>::type *__restrict__ sPos, const R *__restrict__ sDens, const R *__restrict__ sPres,
                                   uint32_t i)
{
    const uint32_t skip = I.inv[i];
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const R p = sPres[i];
    const R dens = sDens[i];
    const R ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad, rd = P.restDensity;
    V3<R> fp = mk3<R>(0, 0, 0);
    HitCounts hc = unpack_counts(hb.counts[i]);
    auto term = [&](uint32_t j) {
        const V3<R> d = pos1 - xyz<R>(sPos[j]);
        const R pj = sPres[j];
        const R densj = sDens[j];
        const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
        const V3<R> contrib = -pm * pm * (p / (dens * dens) + pj / (densj * densj)) * grad;
        fp = fp + contrib;
    };
    if ((!HAS_B || !hc.anyB) && !hc.over) { // no boundary particles in any of the 27 cells: only the fluid list contributes
        struct Nb { typename Vec4T<R>::type q; R pj, dn; };
        walk_fluid_batched(hb.hits + i, hb.stride, hc.nf, [&](uint32_t j) { return Nb{sPos[j], sPres[j], sDens[j]}; },
                           [&](uint32_t j, uint32_t, const Nb &nb) {
                               if (j == skip) return;
                               const V3<R> grad = W_grad_listed<R, KSET>(pos1 - xyz<R>(nb.q), ir, kpg);
                               fp = fp + (-pm * pm * (p / (dens * dens) + nb.pj / (nb.dn * nb.dn)) * grad);
                           });
    } else {
        const I3 gp = calcGridPos<R>(P, pos1);
        int kf = 0; // cursor into the fluid list (ascending cell number)
        for (int c = 0; c < 27; ++c) {
            const int z = c / 9 - 1, y = (c / 3) % 3 - 1, x = c % 3 - 1;
            const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
            if (hc.over) {
                const uint32_t s = G.cellStart[h];
                if (s != CELL_EMPTY) {
                    const uint32_t e = G.cellEnd[h];
                    for (uint32_t j = s; j < e; ++j)
                        if (j != skip) term(j);
                }
            } else {
                while (kf < hc.nf) {
                    const uint32_t ent = hb.hits[(size_t)kf * hb.stride + i];
                    if ((ent >> HIT_TAG_SHIFT) != (uint32_t)c) break;
                    const uint32_t j = ent & HIT_INDEX;
                    if (j != skip) term(j);
                    ++kf;
                }
            }
            if (HAS_B && G.bCellStart[h] != CELL_EMPTY) {
                const uint32_t s = G.cellStart[h], eB = G.bCellEnd[h];
                for (uint32_t j = s; j < eB; ++j) { // Q6 bounds
                    const typename Vec4T<R>::type b = G.sB[j];
                    const V3<R> d = pos1 - xyz<R>(b);
                    const R psi = rd * b.w;
                    const V3<R> contrib = (pm * psi * (p / (dens * dens)) * W_grad<R, KSET>(d, ir, kpg));
                    fp = fp + contrib;
                }
            }
        }
    }
    I.forcesP[i] = mk4<R>(fp, (R)0.0);
}
// WALLS: as k_pressure_lists
template <typename R, int KSET, bool HAS_B, bool WALLS = false>
__global__ __launch_bounds__(BLOCK) void k_pforce_lists(Params<R> P, GridView<R> G, IisphArrays<R> I, HitBuffer hb,
                                                        const typename Vec4T<R>::type *__restrict__ sPos,
                                                        const R *__restrict__ sDens, const R *__restrict__ sPres, uint32_t n,
                                                        WallList wl, uint32_t wallBlocks)
{
    uint32_t block = blockIdx.x, blocks = gridDim.x;
    if (WALLS) {
        if (block < wallBlocks) {
            const uint32_t count = *wl.count;
            for (uint32_t t = block * BLOCK + threadIdx.x; t < count; t += wallBlocks * BLOCK)
                pforce_lists_particle<R, KSET, true>(P, G, I, hb, sPos, sDens, sPres, wl.list[t]);
            return;
        }
        block -= wallBlocks; blocks -= wallBlocks;
    }
    const uint32_t i = xcd_tile(block, blocks) * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (WALLS) {
        if (hb.counts[i] & COUNTS_DEFERRED) return;
        pforce_lists_particle<R, KSET, false>(P, G, I, hb, sPos, sDens, sPres, i);
    } else {
        pforce_lists_particle<R, KSET, HAS_B>(P, G, I, hb, sPos, sDens, sPres, i);
    }
}

} // namespace nrs
