// nrs_kernels_ref.h — "reference-order" gfx950 kernels for the SPH step.
//
// One thread per SORTED slot (coalesced own loads/stores; the reference's thread→slot indirection
// through gridParticleIndex is dropped), the 27 neighbour cells walked z,y,x with j ascending
// and one partial sum per cell, i.e. every floating-point sum is formed in the order the reference forms
// it.  Selected with NRS_FLAG_REFERENCE_ORDER; used for bit-level comparison with the oracle and as the
// semantic definition the tiled kernels (nrs_kernels_tiled.h) are checked against.
//
// What each kernel computes is specified by the reference kernel cited above it.  The physics statements of the force /
// density / IISPH loops (and the smoothing kernels of nrs_math.h) are restatements of the reference's expressions IN THE
// REFERENCE'S EVALUATION ORDER (the local names are this build's own; the oracle keeps the reference's for line-by-line reading): the parity goal
// (every float sum bit-identical to the reference's arithmetic) forces the expression order; everything around them —
// thread mapping, memory layout, templates, boundary packing, double-buffered P_l — is this build's own.
#pragma once
#include "nrs_math.h"
#include <climits>

namespace nrs {

template <typename R> struct GridView {
    typedef typename Vec4T<R>::type T4;
    const uint32_t *cellStart, *cellEnd;   // fluid cell table
    const uint32_t *bCellStart, *bCellEnd; // boundary cell table (valid only when the kernel has HAS_B)
    const T4 *sB;                          // sorted boundary particles: xyz + Vbi in w
    // slab decomposition: a gather kernel evaluates only particles whose (unwrapped) cell-x is in [actLo, actHi)
    // and writes zeros for the rest (halo copies whose neighbourhood is incomplete on this rank)
    int actLo, actHi;
    // consistency guard of the production scans: a fluid run [a, b) read from the cell table must lie inside the sorted array
    // (b >= a, b <= nSorted).  A table built from a wrongly sized merge breaks that, and an unguarded sweep then walks off the
    // allocation (the GPU memory fault of round 1); a run that fails the test is skipped and *err is set — the host reports
    // NRS_E_STATE at the next nrs_synchronize / nrs_download instead of the device faulting.
    uint32_t nSorted;
    uint32_t *err;
    // compact scan candidates (nrs_math.h, quantize_pos): one word per sorted slot, written by the reorder kernels; qT = integer
    // squared-distance threshold of the superset test; null = this context scans the exact positions
    const qword_t *qpos;
    uint32_t qT;
    QuantCfg qc;
};
template <typename R> NRS_DEV bool run_ok(const GridView<R> &G, uint32_t a, uint32_t b)
{
    const bool ok = (b >= a) & (b <= G.nSorted);
    if (!ok && G.err) *G.err = 1u;
    return ok;
}

template <typename R> NRS_DEV bool slab_active(const Params<R> &P, const GridView<R> &G, R x)
{
    // (a single-domain context has no inactive range: without the first test a particle with x = +inf — its cell index saturates past
    // INT_MAX — would be skipped, where the reference gives it its self term; found by the randomised soak against the oracle, round 3)
    if (G.actLo == INT_MIN) return true;
    const long long cx = (long long)floor((x - P.worldOrigin[0]) / P.cellSize[0]);
    return cx >= (long long)G.actLo && cx < (long long)G.actHi;
}

constexpr int BLOCK = 256;

// ---- calcHashD (sph_kernel_impl.cuh:127-145) -------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_hash(Params<R> P, const typename Vec4T<R>::type *__restrict__ pos,
                                                uint32_t *__restrict__ hash, uint32_t *__restrict__ index, uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    I3 g = calcGridPos<R>(P, xyz<R>(pos[i]));
    hash[i] = calcGridHash<R>(P, g.x, g.y, g.z);
    index[i] = i;
}

// Wall particles (see k_density_tiled, nrs_kernels_tiled.h): a sorted slot whose cell is flagged "some boundary particle in the
// 27-neighbourhood" in this static bit table is evaluated by the wall workgroups of the gather launches.
struct WallList {
    const uint32_t *nearBits; // bit per cell
    const uint32_t *hash;     // sorted keys of the step (the cell of every slot)
    const uint32_t *list;     // this step's wall slots, ascending
    const uint32_t *count;    // how many
    const unsigned long long *mask; // bit per sorted slot (one word per wavefront of the reorder kernel): is it a wall slot
};
// Wall slots per 256-slot tile, counted by the reorder kernels while they have each slot's key in a register (first step of the
// wall-list build: counts -> k_resort_scan_tiles -> k_wall_compact).  Called by EVERY thread of the block (it has a barrier);
// h is ignored when !live.
// forceWall: the slot goes to the wall workgroups whatever its cell (an owner the quantised scan cannot serve, see quant_far).
NRS_DEV void wall_tile_count(const uint32_t *__restrict__ nearBits, uint32_t *__restrict__ tileCount, uint32_t h, bool live,
                             unsigned long long *__restrict__ slotMask, bool forceWall = false)
{
    __shared__ uint32_t wallWaveCnt[256 / 64];
    const bool take = live && (forceWall || ((nearBits[h >> 5] >> (h & 31u)) & 1u));
    const unsigned long long m = __ballot(take);
    if ((threadIdx.x & 63u) == 0) {
        wallWaveCnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
        slotMask[(blockIdx.x * 256u + threadIdx.x) >> 6] = m; // (the wall-list compaction and the interior workgroups read this instead of the keys)
    }
    __syncthreads();
    if (threadIdx.x == 0) tileCount[blockIdx.x] = wallWaveCnt[0] + wallWaveCnt[1] + wallWaveCnt[2] + wallWaveCnt[3];
}

// ---- reorderDataAndFindCellStartD (sph_kernel_impl.cuh:210-281) -------------------------------------
// cellStart must have been filled with 0xff.  Also emits inv[index[i]] = i (needed for SURVEY Q5).
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_reorder(const uint32_t *__restrict__ hash, const uint32_t *__restrict__ index,
                                                   const typename Vec4T<R>::type *__restrict__ oldPos,
                                                   const typename Vec4T<R>::type *__restrict__ oldVel,
                                                   const R *__restrict__ oldPres,
                                                   typename Vec4T<R>::type *__restrict__ sPos,
                                                   typename Vec4T<R>::type *__restrict__ sVel, R *__restrict__ sPres,
                                                   uint32_t *__restrict__ cellStart, uint32_t *__restrict__ cellEnd,
                                                   uint32_t *__restrict__ inv, uint32_t n,
                                                   const uint32_t *__restrict__ nearBits, uint32_t *__restrict__ wallTileCount,
                                                   unsigned long long *__restrict__ wallMask, QuantCfg qc, qword_t *__restrict__ qpos)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t src = 0u;
    typename Vec4T<R>::type p4 = mk4<R>((R)0, (R)0, (R)0, (R)0);
    if (i < n) { src = index[i]; p4 = oldPos[src]; }
    // owners outside the range of the quantised scan (too far from the grid origin, NaN) are handed to the wall workgroups, whose
    // scan reads exact positions
    if (nearBits) wall_tile_count(nearBits, wallTileCount, i < n ? hash[i] : 0u, i < n, wallMask, qpos && quant_far<R>(qc, xyz<R>(p4)));
    if (i >= n) return;
    const uint32_t h = hash[i];
    if (i == 0) {
        cellStart[h] = 0;
    } else {
        const uint32_t hp = hash[i - 1];
        if (h != hp) { cellStart[h] = i; cellEnd[hp] = i; }
    }
    if (i == n - 1) cellEnd[h] = n;
    sPos[i] = p4;
    if (qpos) qpos[i] = quantize_pos<R>(qc, xyz<R>(p4));
    sVel[i] = oldVel[src];
    if (oldPres) sPres[i] = oldPres[src];
    if (inv) inv[src] = i;
}

// Undo of the cell table after a step: reset cellStart of exactly the cells the step filled (replaces the
// reference's per-step cudaMemset of 4*numCells bytes, sph_cuda.cu:318, whose cost grows with the EMPTY volume)
static __global__ __launch_bounds__(BLOCK) void k_clear_cells(const uint32_t *__restrict__ hash, uint32_t *__restrict__ cellStart,
                                                       uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = hash[i];
    if (i == 0 || h != hash[i - 1]) cellStart[h] = CELL_EMPTY;
}

// boundary flavour (sph_kernel_impl.cuh:150-205, intended semantics — SURVEY Q1): sorted xyz + vbi packed in one vec4
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_reorder_boundary(const uint32_t *__restrict__ hash,
                                                            const uint32_t *__restrict__ index,
                                                            const typename Vec4T<R>::type *__restrict__ bi,
                                                            const R *__restrict__ vbi,
                                                            typename Vec4T<R>::type *__restrict__ sB,
                                                            uint32_t *__restrict__ cellStart,
                                                            uint32_t *__restrict__ cellEnd, uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = hash[i];
    if (i == 0) {
        cellStart[h] = 0;
    } else {
        const uint32_t hp = hash[i - 1];
        if (h != hp) { cellStart[h] = i; cellEnd[hp] = i; }
    }
    if (i == n - 1) cellEnd[h] = n;
    const uint32_t src = index[i];
    typename Vec4T<R>::type b = bi[src];
    b.w = vbi[src];
    sB[i] = b;
}

// ---- density sum shared by computeDensityPressure (:365-433) and computeIisphDensity (:770-846) ----
template <typename R, int KSET, bool HAS_B>
NRS_DEV R density_of(const Params<R> &P, const GridView<R> &G, const typename Vec4T<R>::type *__restrict__ sPos,
                     uint32_t self)
{
    const V3<R> p = xyz<R>(sPos[self]);
    const I3 gp = calcGridPos<R>(P, p);
    const R ir = P.interactionRadius, kp = P.kpoly, pm = P.particleMass, rd = P.restDensity;
    R dens = (R)0.0;
    dens += pm * W_dens<R, KSET>(mk3<R>(0, 0, 0), ir, kp);
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                {
                    R c = (R)0.0;
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            if (j != self) {
                                const V3<R> d = p - xyz<R>(sPos[j]);
                                if (length(d) < ir) c += (pm * W_dens<R, KSET>(d, ir, kp));
                            }
                        }
                    }
                    dens += c;
                }
                if (HAS_B) {
                    R c = (R)0.0;
                    const uint32_t s = G.bCellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.bCellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            const typename Vec4T<R>::type b = G.sB[j];
                            const V3<R> d = p - xyz<R>(b);
                            if (length(d) < ir) {
                                const R psi = rd * b.w;
                                c += (psi * W_dens<R, KSET>(d, ir, kp));
                            }
                        }
                    }
                    dens += c;
                }
            }
    return dens;
}

// Tait equation of state (sph_kernel_impl.cuh:426)
template <typename R> NRS_DEV R tait_pressure(const Params<R> &P, R dens)
{
    return P.gasStiffness * (pow7f((float)(dens / P.restDensity)) - 1);
}

template <typename R, int KSET, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_density_ref(Params<R> P, GridView<R> G,
                                                       const typename Vec4T<R>::type *__restrict__ sPos,
                                                       R *__restrict__ dens, R *__restrict__ pres, uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (!slab_active<R>(P, G, sPos[i].x)) { dens[i] = (R)0; if (pres) pres[i] = (R)0; return; }
    const R d = density_of<R, KSET, HAS_B>(P, G, sPos, i);
    dens[i] = d;
    if (pres) pres[i] = tait_pressure<R>(P, d);
}

// ---- computeCellForces (sph_kernel_impl.cuh:442-604) -------------------------------------------------
template <typename R> struct ForceAcc { V3<R> fpres, fvisc, fsurf, fbound; };

template <typename R, int KSET, bool SURF, bool HAS_B>
NRS_DEV void cell_forces(const Params<R> &P, const GridView<R> &G, ForceAcc<R> &A, uint32_t h, uint32_t self,
                         V3<R> pos1, V3<R> vel1, R dens, R pres, const typename Vec4T<R>::type *__restrict__ sPos,
                         const typename Vec4T<R>::type *__restrict__ sVel, const R *__restrict__ sDens,
                         const R *__restrict__ sPres)
{
    const R pm = P.particleMass, m2 = P.particleMass, ir = P.interactionRadius, kp = P.kpoly;
    const R kappa = P.surfaceTension;
    const R kprg = P.kpress_grad, kvg = P.kvisc_grad, kvd = P.kvisc_denum;
    uint32_t s = G.cellStart[h];
    if (s != CELL_EMPTY) {
        const uint32_t e = G.cellEnd[h];
        for (uint32_t j = s; j < e; ++j) {
            if (j == self) continue;
            const V3<R> rij = pos1 - xyz<R>(sPos[j]);
            if (length(rij) < ir) {
                const R rhoNb = sDens[j];
                const R pNb = sPres[j];
                const V3<R> vel2 = xyz<R>(sVel[j]);
                const R diameter = (R)(2.0 * P.particleRadius);
                const R diameter2 = diameter * diameter;
                const V3<R> vij = vel1 - vel2;
                const R rhoSqOwn = dens * dens;
                const R rhoSqNb = rhoNb * rhoNb;
                V3<R> gradSpiky, gradVisc;
                R kernel, wAtDiameter;
                if (KSET == KS_MONAGHAN) {
                    gradSpiky = Wmonaghan_grad<R>(rij, ir);
                    gradVisc = gradSpiky;
                    kernel = Wmonaghan<R>(rij, ir);
                    wAtDiameter = Wmonaghan<R>(mk3<R>(diameter, 0, 0), ir);
                } else {
                    gradSpiky = Wpressure_grad<R>(rij, ir, kprg);
                    gradVisc = Wviscosity_grad<R>(rij, ir, kvg, kvd);
                    kernel = Wdefault<R>(rij, ir, kp);
                    wAtDiameter = Wdefault<R>(mk3<R>(diameter, 0, 0), ir, kp);
                }
                A.fpres = A.fpres + (m2 * (pres / rhoSqOwn + pNb / rhoSqNb) * gradSpiky);
                const R a = dot(rij, gradVisc);
                const R b = dot(rij, rij) + 0.01f * (ir * ir);
                A.fvisc = A.fvisc + (m2 / rhoNb * vij * (a / b));
                if (SURF) {
                    V3<R> ai = mk3<R>(0, 0, 0);
                    const R r2 = dot(rij, rij);
                    if (r2 > diameter2) ai = ai - (kappa / pm * pm * rij * kernel);
                    else ai = ai - (kappa / pm * pm * rij * wAtDiameter);
                    A.fsurf = A.fsurf + ai;
                }
            }
        }
    }
    if (HAS_B) {
        s = G.bCellStart[h];
        const R epsilon = (R)0.01;
        const R beta = P.beta;
        const R rd = P.restDensity;
        if (s != CELL_EMPTY) {
            const uint32_t e = G.bCellEnd[h];
            for (uint32_t j = s; j < e; ++j) { // no distance test on boundary particles (as the reference)
                const typename Vec4T<R>::type bq = G.sB[j];
                const R vbi = bq.w;
                const V3<R> vpos = xyz<R>(bq);
                const R psi = (rd * vbi);
                const V3<R> rij = pos1 - vpos;
                const V3<R> vij = vel1;
                R kernel;
                V3<R> grad;
                if (KSET == KS_MONAGHAN) {
                    kernel = Wmonaghan<R>(rij, ir);
                    grad = Wmonaghan_grad<R>(rij, ir);
                } else {
                    kernel = Wdefault<R>(rij, ir, P.kpoly);
                    grad = Wdefault_grad<R>(rij, ir, P.kpoly_grad);
                }
                A.fbound = A.fbound + (beta * psi * rij * kernel);
                A.fpres = A.fpres + (-pm * psi * (pres / (dens * dens)) * grad);
                const R nuWall = (P.viscosity * ir * P.soundSpeed) / (dens * dens);
                const R approach = (R)fmax((double)dot(vij, rij), 0.0);
                const R normSq = dot(rij / length(rij), rij / length(rij)) + epsilon * ir * ir;
                const R friction = -nuWall * (approach / normSq);
                A.fvisc = A.fvisc - (pm * psi * friction * grad);
            }
        }
    }
}

template <typename R, int KSET, bool SURF, bool HAS_B>
NRS_DEV ForceAcc<R> gather_forces(const Params<R> &P, const GridView<R> &G, uint32_t self, V3<R> pos, V3<R> vel, R dens,
                                  R pres, const typename Vec4T<R>::type *__restrict__ sPos,
                                  const typename Vec4T<R>::type *__restrict__ sVel, const R *__restrict__ sDens,
                                  const R *__restrict__ sPres)
{
    ForceAcc<R> A;
    A.fpres = A.fvisc = A.fsurf = A.fbound = mk3<R>(0, 0, 0);
    const I3 gp = calcGridPos<R>(P, pos);
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                cell_forces<R, KSET, SURF, HAS_B>(P, G, A, h, self, pos, vel, dens, pres, sPos, sVel, sDens, sPres);
            }
    return A;
}

// final combination of computeForces (sph_kernel_impl.cuh:663-674)
template <typename R> NRS_DEV V3<R> sesph_total_force(const Params<R> &P, ForceAcc<R> A, R dens)
{
    const R m1 = P.particleMass;
    V3<R> fpres = A.fpres * dens;
    V3<R> fvisc = A.fvisc * 2.0;
    fpres = fpres * -(m1 / dens);
    fvisc = fvisc * (m1 * P.viscosity);
    const V3<R> grav = mk3<R>(P.gravity[0], P.gravity[1], P.gravity[2]);
    return fpres + fvisc + (grav * m1) + A.fsurf + A.fbound;
}

// ---- computeForces (sph_kernel_impl.cuh:609-680) -----------------------------------------------------
template <typename R, int KSET, bool SURF, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_forces_ref(Params<R> P, GridView<R> G,
                                                      const typename Vec4T<R>::type *__restrict__ sPos,
                                                      const typename Vec4T<R>::type *__restrict__ sVel,
                                                      const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                      typename Vec4T<R>::type *__restrict__ forces, uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3<R> pos = xyz<R>(sPos[i]);
    if (!slab_active<R>(P, G, pos.x)) { forces[i] = mk4<R>((R)0, (R)0, (R)0, (R)0); return; }
    const V3<R> vel = xyz<R>(sVel[i]);
    const R dens = sDens[i], pres = sPres[i];
    ForceAcc<R> A = gather_forces<R, KSET, SURF, HAS_B>(P, G, i, pos, vel, dens, pres, sPos, sVel, sDens, sPres);
    const V3<R> f = sesph_total_force<R>(P, A, dens);
    forces[i] = mk4<R>(f, (R)0);
}

// IISPH slab runs: the warm-start pressure of a particle (p0 = 0.5 p of the last step, sph_kernel_impl.cuh:1187) has to travel
// with it through the partition and the messages.  vel.w is free for that — iisph_integrate zeroes it every step
// (sph_kernel_impl.cuh:1654) — so the pressure rides there from nrs_slab_pack to nrs_slab_unpack.
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_pressure_to_velw(typename Vec4T<R>::type *__restrict__ vel, const R *__restrict__ pres, uint32_t n)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) vel[i].w = pres[i];
}
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_velw_to_pressure(typename Vec4T<R>::type *__restrict__ vel, R *__restrict__ pres, uint32_t n)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) { pres[i] = vel[i].w; vel[i].w = (R)0; }
}

// ---- integrate_functor (sph_kernel_impl.cuh:71-100): symplectic Euler, w components kept --------------
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_integrate(Params<R> P, typename Vec4T<R>::type *__restrict__ pos,
                                                     typename Vec4T<R>::type *__restrict__ vel,
                                                     const typename Vec4T<R>::type *__restrict__ forces, uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const R dt = P.timestep, m1 = P.particleMass;
    typename Vec4T<R>::type p4 = pos[i], v4 = vel[i];
    V3<R> p = xyz<R>(p4), v = xyz<R>(v4), frc = xyz<R>(forces[i]);
    const V3<R> accel = dt * frc / m1;
    v = v + accel;
    p = p + dt * v;
    pos[i] = mk4<R>(p, p4.w);
    vel[i] = mk4<R>(v, v4.w);
}

// =========================================== IISPH ====================================================
template <typename R> struct IisphArrays {
    typedef typename Vec4T<R>::type T4;
    R *densAdv, *densCorr, *P_l, *P_l_next, *aii;
    T4 *velAdv, *forcesAdv, *forcesP, *diiF, *diiB, *sumDij;
    T4 *diiSum; // diiF + diiB, formed once per step by the list-driven displacement kernel (the pressure kernel's neighbour
                // term reads only the sum: one 16-byte gather per neighbour and iteration instead of two)
    const uint32_t *inv; // inv[slot] = id of the reference thread that handles the slot
    // Set by the list-driven kernels when a value that NEIGHBOURS will gather (velAdv, dii, sumDij, P_l) is not finite or so large
    // that a product or difference of two such values could overflow.  The list walks visit only neighbours inside the kernel
    // support; the reference's 27-cell walks also multiply the zero gradient of a particle beyond it with expressions of those
    // values (0 * inf = NaN): identical only while everything stays finite.  The context then repeats the step with the
    // reference-order kernels (Ctx::iisph_tail).  null: not watched.
    uint32_t *nonFinite;
};
// (1e12: the cube of it is still a finite float — fp64 builds pass scalars through float as well, SURVEY Q11 —; a fluid step has no
// quantity anywhere near it)
template <typename R> NRS_DEV R watch_limit() { return (R)1e12; }
template <typename R> NRS_DEV void watch_finite(uint32_t *flag, V3<R> v)
{
    const R lim = watch_limit<R>();
    if (flag && !((fabs(v.x) < lim) & (fabs(v.y) < lim) & (fabs(v.z) < lim))) *flag = 1u; // (false for NaN too)
}
template <typename R> NRS_DEV void watch_finite(uint32_t *flag, R v)
{
    if (flag && !(fabs(v) < watch_limit<R>())) *flag = 1u;
}
// sPres[i] = pres[index[i]]: the sorted warm-start pressures once more (a step repeated in reference order)
template <typename R>
static __global__ __launch_bounds__(BLOCK) void k_gather_scalar(const R *__restrict__ src, const uint32_t *__restrict__ index, R *__restrict__ dst, uint32_t n)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

// computeIisphDensity (:770-846) is k_density_ref with pres == nullptr.

// computeDisplacementFactor (sph_kernel_impl.cuh:851-963) + its two cell helpers (:689-765)
template <typename R, int KSET, bool SURF, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_displacement_ref(Params<R> P, GridView<R> G, IisphArrays<R> I,
                                                            const typename Vec4T<R>::type *__restrict__ sPos,
                                                            const typename Vec4T<R>::type *__restrict__ sVel,
                                                            const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                            uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const V3<R> vel1 = xyz<R>(sVel[i]);
    const R pres = (R)0.0;
    const R dens = sDens[i];
    const R kpg = P.kpoly_grad, pm = P.particleMass, ir = P.interactionRadius, rd = P.restDensity, dt = P.timestep;
    ForceAcc<R> A = gather_forces<R, KSET, SURF, HAS_B>(P, G, i, pos1, vel1, dens, pres, sPos, sVel, sDens, sPres);
    V3<R> fvisc = 2.0 * A.fvisc;
    fvisc = (pm * P.viscosity) * fvisc;
    const V3<R> fgrav = pm * mk3<R>(P.gravity[0], P.gravity[1], P.gravity[2]);
    const V3<R> force_adv = fvisc + A.fsurf + A.fbound + fgrav;
    const V3<R> vel_adv = vel1 + dt * (force_adv / pm);
    I.forcesAdv[i] = mk4<R>(force_adv, (R)0.0);
    I.velAdv[i] = mk4<R>(vel_adv, (R)0.0);

    const I3 gp = calcGridPos<R>(P, pos1);
    V3<R> df = mk3<R>(0, 0, 0), db = mk3<R>(0, 0, 0);
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                {
                    V3<R> res = mk3<R>(0, 0, 0);
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            if (j == i) continue;
                            const V3<R> d = pos1 - xyz<R>(sPos[j]);
                            if (length(d) < ir) {
                                const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                                res = res - ((pm / (dens * dens)) * grad);
                            }
                        }
                    }
                    df = df + res;
                }
                if (HAS_B) {
                    V3<R> res = mk3<R>(0, 0, 0);
                    const uint32_t s = G.bCellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.bCellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            const typename Vec4T<R>::type b = G.sB[j];
                            const V3<R> d = pos1 - xyz<R>(b);
                            const R psi = rd * b.w;
                            if (length(d) < ir) {
                                const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                                res = res - ((psi / (dens * dens)) * grad);
                            }
                        }
                    }
                    db = db + res;
                }
            }
    I.diiF[i] = mk4<R>(df, (R)0.0);
    I.diiB[i] = mk4<R>(db, (R)0.0);
}

// computeAdvectionFactor (sph_kernel_impl.cuh:1114-1218) + helpers (:968-1108)
template <typename R, int KSET, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_advection_ref(Params<R> P, GridView<R> G, IisphArrays<R> I,
                                                         const typename Vec4T<R>::type *__restrict__ sPos,
                                                         const typename Vec4T<R>::type *__restrict__ sVel,
                                                         const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                         uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const V3<R> vel1 = xyz<R>(sVel[i]);
    const V3<R> velAdv1 = xyz<R>(I.velAdv[i]);
    const R dens = sDens[i];
    const V3<R> diif = xyz<R>(I.diiF[i]);
    const V3<R> diib = xyz<R>(I.diiB[i]);
    const I3 gp = calcGridPos<R>(P, pos1);
    const R kpg = P.kpoly_grad, pm = P.particleMass, ir = P.interactionRadius, rd = P.restDensity, dt = P.timestep;

    R rho_advf = (R)0.0, rho_advb = (R)0.0;
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                {
                    R res = (R)0.0;
                    const uint32_t s = G.cellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.cellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            if (j == i) continue;
                            const V3<R> velAdv2 = xyz<R>(I.velAdv[j]);
                            const V3<R> vij = velAdv1 - velAdv2;
                            const V3<R> d = pos1 - xyz<R>(sPos[j]);
                            if (length(d) < ir) {
                                const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                                res += (dt * pm * dot(vij, grad));
                            }
                        }
                    }
                    rho_advf += res;
                }
                if (HAS_B) {
                    R res = (R)0.0;
                    const uint32_t s = G.bCellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.bCellEnd[h];
                        for (uint32_t j = s; j < e; ++j) { // no cut-off: relies on W_grad == 0 beyond h (Q8)
                            const typename Vec4T<R>::type b = G.sB[j];
                            const V3<R> d = pos1 - xyz<R>(b);
                            const V3<R> vij = vel1;
                            const R psi = (rd * b.w);
                            const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                            res += (dt * psi * dot(vij, grad));
                        }
                    }
                    rho_advb += res;
                }
            }
    const R rho_adv = dens + (rho_advf + rho_advb);
    I.densAdv[i] = rho_adv;
    I.P_l[i] = (R)(0.5 * sPres[i]);

    R aii = (R)0.0;
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                {
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
                }
                if (HAS_B) {
                    R res = (R)0.0;
                    const uint32_t s = G.bCellStart[h];
                    if (s != CELL_EMPTY) {
                        const uint32_t e = G.bCellEnd[h];
                        for (uint32_t j = s; j < e; ++j) {
                            const typename Vec4T<R>::type b = G.sB[j];
                            const V3<R> d = pos1 - xyz<R>(b);
                            const R psi = rd * b.w;
                            const R dpi = (pm) / (dens * dens);
                            const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                            const V3<R> dji = dpi * grad;
                            res += psi * dot((diif + diib) - dji, grad);
                        }
                    }
                    aii += res;
                }
            }
    I.aii[i] = aii;
}

// computeSumDijPj (sph_kernel_impl.cuh:1259-1325) + dijpjcell (:1224-1253)
template <typename R, int KSET>
__global__ __launch_bounds__(BLOCK) void k_sumdij_ref(Params<R> P, GridView<R> G, IisphArrays<R> I,
                                                      const typename Vec4T<R>::type *__restrict__ sPos,
                                                      const R *__restrict__ sDens, uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const I3 gp = calcGridPos<R>(P, pos1);
    const R ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad;
    V3<R> dijpj = mk3<R>(0, 0, 0);
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
                        const R p_lj = I.P_l[j];
                        const R densj = sDens[j];
                        const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                        res = res - ((pm / (densj * densj)) * p_lj * grad);
                    }
                }
                dijpj = dijpj + res;
            }
    I.sumDij[i] = mk4<R>(dijpj, (R)0.0);
}

// NRS_FLAG_IISPH_SELF_BY_SLOT: inv[slot] = slot, so that the two kernels below skip the particle itself (SURVEY Q5 off)
static __global__ __launch_bounds__(BLOCK) void k_identity(uint32_t *__restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) a[i] = i;
}

// computePressure (sph_kernel_impl.cuh:1330-1492): relaxed Jacobi, omega = 0.5.
// Q5: the fluid loop skips j == inv[i] (the reference thread id), not j == i.
// Q6: the boundary loop runs j from the FLUID cell start to the boundary cell end.
// Q7: reads P_l, writes P_l_next (true Jacobi; the reference updates in place and races).
template <typename R, int KSET, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_pressure_ref(Params<R> P, GridView<R> G, IisphArrays<R> I,
                                                        const typename Vec4T<R>::type *__restrict__ sPos,
                                                        const R *__restrict__ sDens, R *__restrict__ sPres, uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t skip = I.inv[i];
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const R dens = sDens[i];
    R p_l = I.P_l[i];
    const R previous_p_l = p_l;
    const R rho_adv = I.densAdv[i];
    const R aii = I.aii[i];
    const V3<R> dijpj = xyz<R>(I.sumDij[i]);
    const I3 gp = calcGridPos<R>(P, pos1);
    const R ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad, dt = P.timestep, rd = P.restDensity;
    R fsum = (R)0.0, bsum = (R)0.0;
    const R dpi = pm / (dens * dens);
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                const uint32_t s = G.cellStart[h];
                if (s != CELL_EMPTY) {
                    const uint32_t e = G.cellEnd[h];
                    for (uint32_t j = s; j < e; ++j) {
                        if (j == skip) continue;
                        const V3<R> d = pos1 - xyz<R>(sPos[j]);
                        const R p_lj = I.P_l[j];
                        const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                        const V3<R> dji = dpi * (grad);
                        const V3<R> d_ji_pi = dji * p_lj;
                        const V3<R> diifj = xyz<R>(I.diiF[j]);
                        const V3<R> diibj = xyz<R>(I.diiB[j]);
                        const V3<R> sum_dijj = xyz<R>(I.sumDij[j]);
                        fsum += pm * dot(dijpj - (diifj + diibj) * p_lj - (sum_dijj - d_ji_pi), grad);
                    }
                }
                if (HAS_B) {
                    const uint32_t sB = G.bCellStart[h];
                    if (sB != CELL_EMPTY) {
                        const uint32_t eB = G.bCellEnd[h];
                        for (uint32_t j = s; j < eB; ++j) {
                            const typename Vec4T<R>::type b = G.sB[j];
                            const V3<R> d = pos1 - xyz<R>(b);
                            const R psi = rd * b.w;
                            const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                            bsum += psi * dot(dijpj, grad);
                        }
                    }
                }
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
    const R p = (R)fmax((double)p_l, 0.0);
    p_l = p;
    rho_corr += aii * previous_p_l;
    I.P_l_next[i] = p_l;
    sPres[i] = p_l;
    I.densCorr[i] = rho_corr;
}

// computePressureForce (sph_kernel_impl.cuh:1497-1620), same Q5/Q6
template <typename R, int KSET, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_pforce_ref(Params<R> P, GridView<R> G, IisphArrays<R> I,
                                                      const typename Vec4T<R>::type *__restrict__ sPos,
                                                      const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                      uint32_t n)
{
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t skip = I.inv[i];
    const V3<R> pos1 = xyz<R>(sPos[i]);
    const R p = sPres[i];
    const R dens = sDens[i];
    const I3 gp = calcGridPos<R>(P, pos1);
    const R ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad, rd = P.restDensity;
    V3<R> fp = mk3<R>(0, 0, 0);
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                const uint32_t s = G.cellStart[h];
                if (s != CELL_EMPTY) {
                    const uint32_t e = G.cellEnd[h];
                    for (uint32_t j = s; j < e; ++j) {
                        if (j == skip) continue;
                        const V3<R> d = pos1 - xyz<R>(sPos[j]);
                        const R pj = sPres[j];
                        const R densj = sDens[j];
                        const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                        const V3<R> contrib = -pm * pm * (p / (dens * dens) + pj / (densj * densj)) * grad;
                        fp = fp + contrib;
                    }
                }
                if (HAS_B) {
                    const uint32_t sB = G.bCellStart[h];
                    if (sB != CELL_EMPTY) {
                        const uint32_t eB = G.bCellEnd[h];
                        for (uint32_t j = s; j < eB; ++j) {
                            const typename Vec4T<R>::type b = G.sB[j];
                            const V3<R> d = pos1 - xyz<R>(b);
                            const R psi = rd * b.w;
                            const V3<R> grad = W_grad<R, KSET>(d, ir, kpg);
                            const V3<R> contrib = (pm * psi * (p / (dens * dens)) * grad);
                            fp = fp + contrib;
                        }
                    }
                }
            }
    I.forcesP[i] = mk4<R>(fp, (R)0.0);
}

// iisph_integrate (sph_kernel_impl.cuh:1625-1655): sets pos.w = 1, vel.w = 0
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_iisph_integrate(Params<R> P, typename Vec4T<R>::type *__restrict__ pos,
                                                           typename Vec4T<R>::type *__restrict__ vel,
                                                           const typename Vec4T<R>::type *__restrict__ velAdv,
                                                           const typename Vec4T<R>::type *__restrict__ forcesP,
                                                           uint32_t n, uint32_t *__restrict__ nextHash,
                                                           uint32_t *__restrict__ nextIndex, const uint32_t *__restrict__ prevHash,
                                                           uint32_t *__restrict__ tileMovers, int keepHaloMark)
{
    // nextHash/nextIndex (both or neither): also emit the next step's sort keys (calcHashD of the new position);
    // prevHash/tileMovers (both or neither): count the slots whose key changes, for the coherent re-sort
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const R dt = P.timestep, pm = P.particleMass;
    const V3<R> pos1 = xyz<R>(pos[i]);
    const V3<R> velAdv1 = xyz<R>(velAdv[i]);
    const V3<R> fpres1 = xyz<R>(forcesP[i]);
    const V3<R> newVel = velAdv1 + (dt * fpres1 / pm);
    const V3<R> newPos = pos1 + (dt * newVel);
    // the reference sets w = 1 / 0 here (sph_kernel_impl.cuh:1653-1654); slab runs keep the w = 2 that marks a halo copy (not a
    // particle of this rank: it is dropped by the next partition)
    const R w0 = pos[i].w;
    pos[i] = mk4<R>(newPos, (keepHaloMark && w0 == (R)2.0) ? (R)2.0 : (R)1.0);
    vel[i] = mk4<R>(newVel, (R)0.0);
    if (nextHash) {
        const I3 g = calcGridPos<R>(P, newPos);
        const uint32_t h = calcGridHash<R>(P, g.x, g.y, g.z);
        nextHash[i] = h;
        nextIndex[i] = i;
        if (tileMovers && h != prevHash[i]) atomicAdd(&tileMovers[i / BLOCK], 1u);
    }
}

// deterministic two-pass sum of an SReal array in double (replaces thrust::reduce, sph_cuda.cu:816-819)
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_sum_partial(const R *__restrict__ a, double *__restrict__ partial, uint32_t n,
                                                       const typename Vec4T<R>::type *__restrict__ ownedPos = nullptr,
                                                       unsigned long long *__restrict__ ownedCount = nullptr)
{
    // ownedPos (slab runs): only the slots that hold a particle of this rank (pos.w == 1) count; their number goes to ownedCount
    __shared__ double sm[BLOCK / 64];
    double acc = 0.0;
    unsigned long long cnt = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        if (ownedPos && !(ownedPos[i].w == (R)1.0)) continue;
        acc += (double)a[i];
        ++cnt;
    }
    if (ownedCount) {
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(ownedCount, cnt);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sm[w];
        partial[blockIdx.x] = t;
    }
}
static __global__ __launch_bounds__(BLOCK) void k_sum_final(const double *__restrict__ partial, double *__restrict__ out,
                                                     uint32_t nblocks)
{
    __shared__ double sm[BLOCK / 64];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < nblocks; i += BLOCK) acc += partial[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sm[w];
        *out = t;
    }
}

// max of an SReal array / of |v| over a vec4 array (maxDensity / maxVelocity, sph_cuda.cu:32-53)
template <typename R, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_max_partial(const void *__restrict__ a, double *__restrict__ partial,
                                                       uint32_t n)
{
    __shared__ double sm[BLOCK / 64];
    double acc = -1.0e300;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        double v;
        if (VEC) {
            typename Vec4T<R>::type q = ((const typename Vec4T<R>::type *)a)[i];
            v = sqrt((double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z);
        } else {
            v = (double)((const R *)a)[i];
        }
        acc = fmax(acc, v);
    }
    for (int off = 32; off > 0; off >>= 1) acc = fmax(acc, __shfl_down(acc, off, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sm[0];
        for (int w = 1; w < BLOCK / 64; ++w) t = fmax(t, sm[w]);
        partial[blockIdx.x] = t;
    }
}

} // namespace nrs
