// nrs_kernels_tiled.h — production gather kernels of the SESPH step for gfx950.
//
// Shape of the work: per particle ≈40-55 candidate neighbours in the 27 surrounding
// cells, of which only ≈6-10 lie inside the support radius.  A one-pass "test and accumulate" loop
// wastes 80-85 % of every 64-lane wavefront in the heavy branch.  These kernels therefore run two phases
// per thread (one thread per SORTED slot, so own loads/stores are coalesced 16 B/lane):
//
//   scan    walk the 9 (dz,dy) rows; the three x-cells of a row are ONE contiguous run of the sorted array
//           (hash = (z*gy+y)*gx+x), so a row is a single [lo,hi) sweep of float4 positions with a
//           squared-distance compare against a precomputed threshold (no sqrt, no divide).  Hits are
//           appended to a per-thread list kept in LDS (lst[k][tid], k-major: conflict-free).
//   process the compacted hits (nearly equal counts across lanes → dense wavefronts) get the expensive
//           kernel evaluation, in the same order the reference visits them.
//
// Every floating-point sum is formed in the reference's order (per-cell partial sums for the density,
// running sums for the forces; flags in the hit list mark partial-sum boundaries), so the results are
// bit-identical to the reference-order kernels in nrs_kernels_ref.h — which are also the overflow path
// for a thread whose hit list would exceed HIT_CAP (correct for any neighbour count).
//
// No MFMA: this is a bandwidth/latency-bound gather, not a dense contraction.
#pragma once
#include "nrs_kernels_ref.h"

namespace nrs {

constexpr int HIT_CAP = 32;                 // hits kept per thread (LDS: HIT_CAP*BLOCK*4 B = 32 KiB per workgroup)
constexpr uint32_t HIT_BOUNDARY = 1u << 31; // entry refers to a boundary particle
constexpr uint32_t HIT_NEWPART = 1u << 30;  // a partial-sum boundary was crossed since the previous hit
constexpr uint32_t HIT_INDEX = (1u << 30) - 1;

// Thresholds that turn the reference's two cut-off predicates into one float compare on the float dot
// product d2 = dot(r,r)  (exact: sqrtf and the products are monotone, correctly rounded):
//   lenLtIr : smallest float T with  sqrtf(T) >= ir          ⇒  (length(r) <  ir)        ⇔ d2 < T
//   r2LeH2  : smallest float T with  fl(sqrtf(T)^2) > h*h    ⇒  !(length(r)^2 > h^2)     ⇔ d2 < T
struct CutThresholds { float lenLtIr, r2LeH2; };

template <typename R> struct Sweep {
    typedef typename Vec4T<R>::type T4;

    // Scan the neighbourhood of `self`; returns the hit count, or -1 when the list overflowed.
    // BFILT: cut-off used for boundary candidates: 0 = lenLtIr (explicit test of the density loop); 1 = r2LeH2
    // (the force loop has no explicit test, but every Muller kernel it evaluates returns 0 beyond h);
    // 2 = none (Monaghan kernels reach 2h, so every boundary particle of the 27 cells contributes).
    template <bool HAS_B, int BFILT>
    static NRS_DEV int scan(const Params<R> &P, const GridView<R> &G, const CutThresholds thr,
                            const T4 *__restrict__ sPos, uint32_t self, V3<R> p, uint32_t (*lst)[BLOCK])
    {
        const I3 gp = calcGridPos<R>(P, p);
        const uint32_t mx = P.gridSize[0] - 1, my = P.gridSize[1] - 1, mz = P.gridSize[2] - 1;
        const uint32_t cx = (uint32_t)gp.x & mx;
        const bool contiguous = (cx >= 1u) && (cx + 1u <= mx);
        const float tF = thr.lenLtIr;
        const float tB = BFILT == 2 ? INFINITY : (BFILT == 1 ? thr.r2LeH2 : thr.lenLtIr);
        const uint32_t tid = threadIdx.x;
        int cnt = 0;
        uint32_t pend = HIT_NEWPART;
        bool over = false;

        auto testFluid = [&](uint32_t j) {
            if (j != self) {
                const V3<R> d = p - xyz<R>(sPos[j]);
                if (dot(d, d) < tF) {
                    if (cnt < HIT_CAP) lst[cnt][tid] = j | pend; else over = true;
                    ++cnt;
                    pend = 0;
                }
            }
        };
        auto testBoundary = [&](uint32_t j) {
            const V3<R> d = p - xyz<R>(G.sB[j]);
            if (dot(d, d) < tB) {
                if (cnt < HIT_CAP) lst[cnt][tid] = j | pend | HIT_BOUNDARY; else over = true;
                ++cnt;
                pend = 0;
            }
        };

        for (int z = -1; z <= 1; z++) {
            const uint32_t cz = (uint32_t)(gp.z + z) & mz;
            for (int y = -1; y <= 1; y++) {
                const uint32_t cy = (uint32_t)(gp.y + y) & my;
                const uint32_t row = umul24(umul24(cz, P.gridSize[1]), P.gridSize[0]) + umul24(cy, P.gridSize[0]);
                const uint32_t h0 = row + ((cx - 1u) & mx), h1 = row + cx, h2 = row + ((cx + 1u) & mx);
                const uint32_t s0 = G.cellStart[h0], s1 = G.cellStart[h1], s2 = G.cellStart[h2];
                bool anyB = false;
                uint32_t b0 = CELL_EMPTY, b1 = CELL_EMPTY, b2 = CELL_EMPTY;
                if (HAS_B) {
                    b0 = G.bCellStart[h0]; b1 = G.bCellStart[h1]; b2 = G.bCellStart[h2];
                    anyB = (b0 & b1 & b2) != CELL_EMPTY;
                }
                if (contiguous && !anyB) {
                    // one contiguous run of the sorted array: [first non-empty start, last non-empty end)
                    uint32_t lo, hi;
                    if (s2 != CELL_EMPTY) hi = G.cellEnd[h2];
                    else if (s1 != CELL_EMPTY) hi = G.cellEnd[h1];
                    else if (s0 != CELL_EMPTY) hi = G.cellEnd[h0];
                    else hi = 0;
                    lo = (s0 != CELL_EMPTY) ? s0 : ((s1 != CELL_EMPTY) ? s1 : s2);
                    if (lo == CELL_EMPTY) hi = 0;
                    pend = HIT_NEWPART;
                    for (uint32_t j = lo; j < hi; ++j) {
                        if (j == s1 || j == s2) pend = HIT_NEWPART;
                        testFluid(j);
                    }
                } else {
                    const uint32_t hs[3] = {h0, h1, h2}, ss[3] = {s0, s1, s2}, bs[3] = {b0, b1, b2};
                    for (int c = 0; c < 3; ++c) {
                        pend = HIT_NEWPART;
                        if (ss[c] != CELL_EMPTY) {
                            const uint32_t e = G.cellEnd[hs[c]];
                            for (uint32_t j = ss[c]; j < e; ++j) testFluid(j);
                        }
                        if (HAS_B) {
                            pend = HIT_NEWPART;
                            if (bs[c] != CELL_EMPTY) {
                                const uint32_t e = G.bCellEnd[hs[c]];
                                for (uint32_t j = bs[c]; j < e; ++j) testBoundary(j);
                            }
                        }
                    }
                }
            }
        }
        return over ? -1 : cnt;
    }
};

// ---- density + Tait pressure (computeDensityPressure, sph_kernel_impl.cuh:365-433) -----------------------
template <typename R, int KSET, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_density_tiled(Params<R> P, GridView<R> G, CutThresholds thr,
                                                         const typename Vec4T<R>::type *__restrict__ sPos,
                                                         R *__restrict__ dens, R *__restrict__ pres, uint32_t n)
{
    __shared__ uint32_t lst[HIT_CAP][BLOCK];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t tid = threadIdx.x;
    const V3<R> p = xyz<R>(sPos[i]);
    if (!slab_active<R>(P, G, p.x)) { dens[i] = (R)0; if (pres) pres[i] = (R)0; return; }
    const int cnt = Sweep<R>::template scan<HAS_B, 0>(P, G, thr, sPos, i, p, lst);
    R d;
    if (cnt < 0) {
        d = density_of<R, KSET, HAS_B>(P, G, sPos, i); // overflow: reference-order path
    } else {
        const R ir = P.interactionRadius, kp = P.kpoly, pm = P.particleMass, rd = P.restDensity;
        d = (R)0.0;
        d += pm * W_dens<R, KSET>(mk3<R>(0, 0, 0), ir, kp);
        R part = (R)0.0;
        for (int k = 0; k < cnt; ++k) {
            const uint32_t ent = lst[k][tid];
            if (ent & HIT_NEWPART) { d += part; part = (R)0.0; }
            const uint32_t j = ent & HIT_INDEX;
            if (HAS_B && (ent & HIT_BOUNDARY)) {
                const typename Vec4T<R>::type b = G.sB[j];
                const V3<R> r = p - xyz<R>(b);
                const R psi = rd * b.w;
                part += (psi * W_dens<R, KSET>(r, ir, kp));
            } else {
                const V3<R> r = p - xyz<R>(sPos[j]);
                part += (pm * W_dens<R, KSET>(r, ir, kp));
            }
        }
        d += part;
    }
    dens[i] = d;
    if (pres) pres[i] = tait_pressure<R>(P, d);
}

// ---- forces (computeForces + computeCellForces, sph_kernel_impl.cuh:442-680) -----------------------------
template <typename R, int KSET, bool SURF, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_forces_tiled(Params<R> P, GridView<R> G, CutThresholds thr,
                                                        const typename Vec4T<R>::type *__restrict__ sPos,
                                                        const typename Vec4T<R>::type *__restrict__ sVel,
                                                        const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                        typename Vec4T<R>::type *__restrict__ forces, uint32_t n)
{
    __shared__ uint32_t lst[HIT_CAP][BLOCK];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t tid = threadIdx.x;
    const V3<R> pos1 = xyz<R>(sPos[i]);
    if (!slab_active<R>(P, G, pos1.x)) { forces[i] = mk4<R>((R)0, (R)0, (R)0, (R)0); return; }
    const V3<R> vel1 = xyz<R>(sVel[i]);
    const R dens = sDens[i], pres = sPres[i];
    const int cnt = Sweep<R>::template scan<HAS_B, (KSET == KS_MULLER ? 1 : 2)>(P, G, thr, sPos, i, pos1, lst);
    ForceAcc<R> A;
    if (cnt < 0) {
        A = gather_forces<R, KSET, SURF, HAS_B>(P, G, i, pos1, vel1, dens, pres, sPos, sVel, sDens, sPres);
    } else {
        A.fpres = A.fvisc = A.fsurf = A.fbound = mk3<R>(0, 0, 0);
        const R pm = P.particleMass, m2 = P.particleMass, ir = P.interactionRadius, kp = P.kpoly;
        const R kappa = P.surfaceTension;
        const R kprg = P.kpress_grad, kvg = P.kvisc_grad, kvd = P.kvisc_denum;
        const R diameter = (R)(2.0 * P.particleRadius);
        const R diameter2 = diameter * diameter;
        const R d1sq = dens * dens;
        R kernel_diameter;
        if (KSET == KS_MONAGHAN) kernel_diameter = Wmonaghan<R>(mk3<R>(diameter, 0, 0), ir);
        else kernel_diameter = Wdefault<R>(mk3<R>(diameter, 0, 0), ir, kp);
        const R epsilon = (R)0.01;
        const R beta = P.beta, rd = P.restDensity;
        for (int k = 0; k < cnt; ++k) {
            const uint32_t ent = lst[k][tid];
            const uint32_t j = ent & HIT_INDEX;
            if (HAS_B && (ent & HIT_BOUNDARY)) {
                const typename Vec4T<R>::type bq = G.sB[j];
                const R psi = (rd * bq.w);
                const V3<R> p1p2 = pos1 - xyz<R>(bq);
                const V3<R> v1v2 = vel1;
                R kernel;
                V3<R> grad;
                if (KSET == KS_MONAGHAN) {
                    kernel = Wmonaghan<R>(p1p2, ir);
                    grad = Wmonaghan_grad<R>(p1p2, ir);
                } else {
                    kernel = Wdefault<R>(p1p2, ir, P.kpoly);
                    grad = Wdefault_grad<R>(p1p2, ir, P.kpoly_grad);
                }
                A.fbound = A.fbound + (beta * psi * p1p2 * kernel);
                A.fpres = A.fpres + (-pm * psi * (pres / (dens * dens)) * grad);
                const R nu = (P.viscosity * ir * P.soundSpeed) / (dens * dens);
                const R nom = (R)fmax((double)dot(v1v2, p1p2), 0.0);
                const R denom = dot(p1p2 / length(p1p2), p1p2 / length(p1p2)) + epsilon * ir * ir;
                const R Pij = -nu * (nom / denom);
                A.fvisc = A.fvisc - (pm * psi * Pij * grad);
            } else {
                const V3<R> p1p2 = pos1 - xyz<R>(sPos[j]);
                const R dens2 = sDens[j];
                const R pres2 = sPres[j];
                const V3<R> v1v2 = vel1 - xyz<R>(sVel[j]);
                const R d2sq = dens2 * dens2;
                V3<R> kpressure_grad, kvisco_grad;
                R kernel;
                if (KSET == KS_MONAGHAN) {
                    kpressure_grad = Wmonaghan_grad<R>(p1p2, ir);
                    kvisco_grad = kpressure_grad;
                    kernel = Wmonaghan<R>(p1p2, ir);
                } else {
                    kpressure_grad = Wpressure_grad<R>(p1p2, ir, kprg);
                    kvisco_grad = Wviscosity_grad<R>(p1p2, ir, kvg, kvd);
                    kernel = Wdefault<R>(p1p2, ir, kp);
                }
                A.fpres = A.fpres + (m2 * (pres / d1sq + pres2 / d2sq) * kpressure_grad);
                const R a = dot(p1p2, kvisco_grad);
                const R b = dot(p1p2, p1p2) + 0.01f * (ir * ir);
                A.fvisc = A.fvisc + (m2 / dens2 * v1v2 * (a / b));
                if (SURF) {
                    V3<R> ai = mk3<R>(0, 0, 0);
                    const R r2 = dot(p1p2, p1p2);
                    if (r2 > diameter2) ai = ai - (kappa / pm * pm * p1p2 * kernel);
                    else ai = ai - (kappa / pm * pm * p1p2 * kernel_diameter);
                    A.fsurf = A.fsurf + ai;
                }
            }
        }
    }
    const V3<R> f = sesph_total_force<R>(P, A, dens);
    forces[i] = mk4<R>(f, (R)0);
}

// host-side threshold search (IEEE float arithmetic on the host)
template <typename R> static inline CutThresholds make_thresholds(const Params<R> &P)
{
    CutThresholds t;
    const R ir = P.interactionRadius;
    {
        float T = (float)(ir * ir);
        auto ge = [&](float x) { return !((R)sqrtf(x) < ir); }; // NOT (length < ir)
        while (ge(T) && T > 0.0f) T = nextafterf(T, 0.0f);
        while (!ge(T)) T = nextafterf(T, INFINITY);
        t.lenLtIr = T;
    }
    {
        const R h2 = ir * ir;
        float T = (float)h2;
        auto gt = [&](float x) { float l = sqrtf(x); R r2 = l * l; return r2 > h2; };
        while (gt(T) && T > 0.0f) T = nextafterf(T, 0.0f);
        while (!gt(T)) T = nextafterf(T, INFINITY);
        t.r2LeH2 = T;
    }
    return t;
}

static inline bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

template <typename R, int KSET, bool HAS_B>
static inline void launch_density_tiled(hipStream_t stream, const Params<R> &P, const GridView<R> &G,
                                        const uint32_t * /*hashSorted*/, const typename Vec4T<R>::type *sPos, R *dens,
                                        R *pres, uint32_t n)
{
    const CutThresholds thr = make_thresholds<R>(P);
    hipLaunchKernelGGL((k_density_tiled<R, KSET, HAS_B>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, P, G, thr,
                       sPos, dens, pres, n);
}
template <typename R, int KSET, bool SURF, bool HAS_B>
static inline void launch_forces_tiled(hipStream_t stream, const Params<R> &P, const GridView<R> &G,
                                       const uint32_t * /*hashSorted*/, const typename Vec4T<R>::type *sPos,
                                       const typename Vec4T<R>::type *sVel, const R *dens, const R *pres,
                                       typename Vec4T<R>::type *forces, uint32_t n)
{
    const CutThresholds thr = make_thresholds<R>(P);
    hipLaunchKernelGGL((k_forces_tiled<R, KSET, SURF, HAS_B>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, P, G,
                       thr, sPos, sVel, dens, pres, forces, n);
}

} // namespace nrs
