/*
 * nereus_oracle.cpp — CPU restatement of the Nereus SPH step.  TEST INFRASTRUCTURE ONLY.
 *
 *   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *     The product (nereus_amd/, libnereus_hip.so, libnereus_host.so) never links or calls it.
 *   * It restates, function by function, what /root/reference computes on its CUDA path
 *     (file:line cited at each function).  It is written from the reference's behaviour,
 *     not copied from it: one emulated "thread" at a time, sequential, IEEE arithmetic,
 *     no FMA contraction (build with -ffp-contract=off).
 *   * PARITY STATUS: the smoothing kernels and the vector helpers below are PINNED
 *     against the reference itself — /root/reference/common/kernels_impl.cuh + cuda_helpers/helper_math.h
 *     compile unmodified with g++ (oracle/Makefile target `ref`, oracle/ref_kernels_driver.cpp) and
 *     tests/test_oracle_ref_pin.py compares orc_eval with that build bit for bit (>= 10^5 inputs per variant,
 *     plus the committed fixture tests/golden/ref_kernels_pin.npz).  Everything else — grid hash, cell walk,
 *     force assembly, integration, the IISPH chain — is PARITY UNPINNED: the reference ships no tests, golden
 *     vectors or fixtures, and sph/sph_kernel_impl.cuh / sph_cuda.cu / sph.cpp need nvcc builtins, cudart and
 *     Thrust-CUDA (no stand-ins were written for them).  For those parts the only reference-derived numbers are
 *     the known answers recorded in SURVEY.md §8c (tests/test_oracle_kat.py).
 *
 * Build (see oracle/Makefile): one .so per reference compile-time configuration
 *   -DDOUBLE_PRECISION={0,1} -DKERNEL_SET={1 Muller,0 Monaghan} -DUSE_SURFACE_TENSION=1
 * which are the reference's own switches (CMakeLists.txt:25-28, common/common.h:14-43).
 *
 * Mixed-precision note: the reference's vector helpers take and return `float`
 * scalars even when SVec3 is double3 (common/cuda_helpers/helper_math.h:817-829,1000-1008,
 * 1251-1301).  The tiny vector layer below keeps exactly those signatures so that the
 * DOUBLE_PRECISION=1 build rounds where the reference rounds.
 */
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

#ifndef DOUBLE_PRECISION
#define DOUBLE_PRECISION 0
#endif
#ifndef KERNEL_SET
#define KERNEL_SET 1
#endif
#ifndef USE_SURFACE_TENSION
#define USE_SURFACE_TENSION 1
#endif
#define MONAGHAN 0
#define MULLER 1

#if DOUBLE_PRECISION == 1
typedef double SReal;
#else
typedef float SReal;
#endif
typedef unsigned int SUint;

struct SVec3 { SReal x, y, z; };
struct SVec4 { SReal x, y, z, w; };
struct I3 { int x, y, z; };
struct U3 { unsigned x, y, z; };

/* common/sph_kernel.cuh:13-59 — same field order; 132 B (fp32) / 240 B (fp64). */
struct SphSimParams {
    U3 gridSize;
    unsigned numCells;
    SVec3 worldOrigin;
    SVec3 cellSize;
    unsigned numBodies;
    unsigned maxParticlesPerCell;
    SReal gasStiffness, viscosity, surfaceTension, restDensity, particleMass, interactionRadius,
          timestep, particleRadius;
    SVec3 gravity;
    SReal soundSpeed;
    SReal beta;
    SReal kpoly, kpoly_grad, kpress_grad, kvisc_grad, kvisc_denum, ksurf1, ksurf2, bpol;
};
static_assert(sizeof(SphSimParams) == (DOUBLE_PRECISION ? 240 : 132), "SphSimParams layout");

/* ---- vector layer: helper_math.h semantics (float scalars, float dot/length) ------------- */
static inline SVec3 mk3(SReal x, SReal y, SReal z) { SVec3 v = {x, y, z}; return v; }
static inline SVec3 mk3(SVec4 a) { return mk3(a.x, a.y, a.z); }                 /* :134 */
static inline SVec4 mk4(SReal x, SReal y, SReal z, SReal w) { SVec4 v = {x, y, z, w}; return v; }
static inline SVec3 operator+(SVec3 a, SVec3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline SVec3 operator-(SVec3 a, SVec3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline SVec3 operator*(SVec3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }   /* :817 */
static inline SVec3 operator*(float b, SVec3 a) { return mk3(b * a.x, b * a.y, b * a.z); }   /* :821 */
static inline SVec3 operator/(SVec3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }   /* :1000 */
static inline float dot(SVec3 a, SVec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       /* :1251 */
static inline float length(SVec3 v) { return sqrtf(dot(v, v)); }                            /* :1294 */

/* ---- smoothing kernels: common/kernels_impl.cuh:85-203 -------------------------------------
 * Host-compiler (g++) overload resolution is what is restated: pow(SReal,int) is the double
 * pow, powf stays float even in fp64 builds. */
static inline SReal Wdefault(SVec3 r, SReal h, SReal kpoly)             /* :85-98 */
{
    SReal r2 = length(r) * length(r);
    SReal h2 = h * h;
    if (r2 > h2) return 0.0;
    SReal b = std::pow((double)(h2 - r2), 3.0);
    return kpoly * b;
}
static inline SVec3 Wdefault_grad(SVec3 r, SReal h, SReal kpoly_grad)   /* :103-116 */
{
    SReal r2 = length(r) * length(r);
    SReal h2 = h * h;
    if (r2 > h2) return mk3(0.0, 0.0, 0.0);
    SReal b = powf(h2 - r2, 2);
    return kpoly_grad * r * b;
}
static inline SVec3 Wpressure_grad(SVec3 r, SReal h, SReal kpress_grad) /* :121-135 */
{
    SReal l_r = length(r);
    SReal r2 = l_r * l_r;
    SReal h2 = h * h;
    if (r2 > h2) return mk3(0.0, 0.0, 0.0);
    SReal c = (h - l_r) * (h - l_r);
    return kpress_grad * (r / l_r) * c;
}
static inline SVec3 Wviscosity_grad(SVec3 r, SReal h, SReal kvisc_grad, SReal kvisc_denum) /* :140-154 */
{
    SReal l_r = length(r);
    SReal r2 = l_r * l_r;
    SReal h2 = h * h;
    if (r2 > h2) return mk3(0.0, 0.0, 0.0);
    SReal c = -(3 * l_r / kvisc_denum) + (2 / (h2)) - (h / (2 * l_r * l_r * l_r));
    return kvisc_grad * r * c;
}
static inline SReal Wmonaghan(SVec3 r, SReal h)                         /* :159-178 */
{
    SReal value = 0.0;
    SReal m_invH = 1.0 / h;
    SReal m_v = 1.0 / (4.0 * M_PI * h * h * h);
    SReal q = length(r) * m_invH;
    if (q >= 0 && q < 1)
        value = m_v * ((2 - q) * (2 - q) * (2 - q) - 4.0f * (1 - q) * (1 - q) * (1 - q));
    else if (q >= 1 && q < 2)
        value = m_v * ((2 - q) * (2 - q) * (2 - q));
    else
        value = 0.0f;
    return value;
}
static inline SVec3 Wmonaghan_grad(SVec3 r, SReal h)                    /* :183-203 */
{
    SReal m_g = 1.0 / (4.0 * M_PI * h * h * h);
    SReal dist = length(r);
    SReal m_invH = 1.0 / h;
    SReal q = dist * m_invH;
    SVec3 gradient = mk3(0.0, 0.0, 0.0);
    if (q >= 0 && q < 1) {
        SReal scalar = -3.0f * (2 - q) * (2 - q);
        scalar += 12.0f * (1 - q) * (1 - q);
        gradient = (m_g * m_invH * scalar / dist) * r;
    } else if (q >= 1 && q < 2) {
        SReal scalar = -3.0f * (2 - q) * (2 - q);
        gradient = (m_g * scalar * m_invH / dist) * r;
    }
    return gradient;
}
/* Akinci cohesion / adhesion kernels (:208-247) — defined by the reference, called nowhere on its path; restated
 * only so that every function of that file is pinned against oracle/_ref */
static inline SReal Cakinci(SVec3 r, SReal h, SReal ksurf1, SReal ksurf2)  /* :208-228 */
{
    SReal len = length(r);
    SReal poly = ksurf1;
    SReal hr = h - len;
    if (2.0 * len > h && len <= h) {
        SReal a = (hr * hr * hr) * (len * len * len);
        return poly * a;
    } else if (len > 0.0 && 2 * len <= h) {
        SReal a = 2 * (hr * hr * hr) * (len * len * len);
        SReal b = ksurf2;
        return poly * (a - b);
    }
    return 0.0;
}
static inline SReal Aboundary(SVec3 r, SReal h, SReal bpol)               /* :233-247 */
{
    SReal rl = length(r);
    if (2.0 * rl > h && rl <= h) {
        SReal a = -((4 * (rl * rl)) / (h));
        SReal b = (6.0 * rl - 2.0 * h);
        SReal res = powf(a + b, 1.0 / 4.0);
        return bpol * res;
    }
    return 0.0;
}
#if KERNEL_SET == MULLER
#define W_DENS(r, ir, kp) Wdefault(r, ir, kp)
#define W_GRAD(r, ir, kpg) Wdefault_grad(r, ir, kpg)
#else
#define W_DENS(r, ir, kp) Wmonaghan(r, ir)
#define W_GRAD(r, ir, kpg) Wmonaghan_grad(r, ir)
#endif

/* ---- grid: sph/sph_kernel_impl.cuh:105-125 ------------------------------------------------ */
/* `(int)` of a float ON THE DEVICE: CUDA converts with cvt.rzi.s32.f32 — NaN gives 0, values beyond the int range saturate — and so does
 * gfx950's v_cvt_i32_f32; in C++ on the host the same cast is undefined for those inputs (x86 returns INT_MIN for all of them).  The
 * path being restated is the CUDA one, so the oracle converts as the devices do.  Only particles with NaN / inf / absurd coordinates
 * (a caller's bug) are affected: which cell they are hashed into. */
static inline int device_f2i(SReal v)
{
    if (v != v) return 0;
    if (v >= (SReal)2147483648.0) return 2147483647;
    if (v <= (SReal)-2147483648.0) return (int)(-2147483647 - 1);
    return (int)v;
}
/* gridPos + offset as the devices form it: 32-bit two's-complement wrap-around (a saturated cell coordinate of a particle at +-inf plus one
 * is signed overflow — undefined — in host C++; the hash masks the low bits either way) */
static inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
static inline I3 calcGridPos(const SphSimParams &P, SVec3 p)
{
    I3 g;
    g.x = device_f2i(std::floor((p.x - P.worldOrigin.x) / P.cellSize.x));
    g.y = device_f2i(std::floor((p.y - P.worldOrigin.y) / P.cellSize.y));
    g.z = device_f2i(std::floor((p.z - P.worldOrigin.z) / P.cellSize.z));
    return g;
}
static inline unsigned umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline SUint calcGridHash(const SphSimParams &P, I3 g)
{
    unsigned x = (unsigned)g.x & (P.gridSize.x - 1);
    unsigned y = (unsigned)g.y & (P.gridSize.y - 1);
    unsigned z = (unsigned)g.z & (P.gridSize.z - 1);
    return umul24(umul24(z, P.gridSize.y), P.gridSize.x) + umul24(y, P.gridSize.x) + x;
}

static const SUint EMPTY = 0xffffffffu;

/* =============================================================================================
 * The simulator object: host arrays + "device" arrays of sph/sph.h:98-148, iisph/iisph.h:27-41
 * ===========================================================================================*/
struct Sim {
    SphSimParams P;
    SUint N = 0, Nb = 0;
    int jacobi = 1;          /* Q7: 1 = double-buffered P_l (our defined semantics), 0 = in place */
    int selfBySlot = 0;      /* Q5 off: exclude the particle itself (j == own slot) instead of the thread id; the result then no
                                longer depends on the ORDER of the input arrays (used to compare slab runs with single-domain ones) */
    int threads = 1;
    int surf = USE_SURFACE_TENSION; /* the reference's compile-time USE_SURFACE_TENSION (CMakeLists.txt:28, sph_kernel_impl.cuh:535-548) as a
                                run-time switch, so that one build checks both instantiations of the product (orc_set_surface_tension) */
    int taitDouble7 = 0;     /* Tait pressure (sph_kernel_impl.cuh:426): 0 = glibc powf(x, 7), what g++ gives the reference on the host;
                                1 = x^7 formed in double and rounded ONCE to float — what the device evaluates (nrs_math.h pow7f).  The two
                                differ by 1 ulp in ~0.1 % of the values; mode 1 lets the N-step bar measure the kernels instead of two
                                pow conventions (tests name the mode they assert in) */
    SUint lastIters = 0;
    /* host */
    std::vector<SVec4> pos, vel;
    std::vector<SReal> pressure;
    /* device: unsorted copies */
    std::vector<SVec4> dpos, dvel;
    std::vector<SReal> dpres;
    /* device: sorted */
    std::vector<SVec4> sPos, sVel, sForces;
    std::vector<SReal> sDens, sPres;
    std::vector<SUint> hash, index, cellStart, cellEnd;
    /* boundaries: unsorted (SESPH kernels go through bindex), sorted (IISPH helpers) — Q1 */
    std::vector<SVec4> bi, sbi;
    std::vector<SReal> vbi, svbi;
    std::vector<SUint> bhash, bindex, bCellStart, bCellEnd;
    /* IISPH */
    std::vector<SReal> densAdv, densCorr, P_l, aii;
    std::vector<SVec4> velAdv, forcesAdv, forcesP, diiF, diiB, sumDij;

    void ensureCells()
    {
        if (cellStart.size() != P.numCells) {
            cellStart.assign(P.numCells, EMPTY);
            cellEnd.assign(P.numCells, 0);
        }
        if (bCellStart.size() != P.numCells) {
            bCellStart.assign(P.numCells, EMPTY);
            bCellEnd.assign(P.numCells, 0);
        }
    }
    void resizeParticles(SUint n)
    {
        N = n;
        pos.resize(n); vel.resize(n); pressure.resize(n, 0);
        dpos.resize(n); dvel.resize(n); dpres.resize(n);
        sPos.resize(n); sVel.resize(n); sForces.resize(n); sDens.resize(n); sPres.resize(n);
        hash.resize(n); index.resize(n);
        densAdv.resize(n); densCorr.resize(n); P_l.resize(n); aii.resize(n);
        velAdv.resize(n); forcesAdv.resize(n); forcesP.resize(n);
        diiF.resize(n); diiB.resize(n); sumDij.resize(n);
    }
};

/* calcHashD: sph_kernel_impl.cuh:127-145 */
static void k_calcHash(const SphSimParams &P, const SVec4 *pos, SUint n, SUint *hash, SUint *index)
{
    for (SUint i = 0; i < n; ++i) {
        I3 g = calcGridPos(P, mk3(pos[i].x, pos[i].y, pos[i].z));
        hash[i] = calcGridHash(P, g);
        index[i] = i;
    }
}
/* sortParticles: sph_cuda.cu:58-63 — radix sort_by_key ⇒ stable; ties keep index order */
static void k_sort(SUint *hash, SUint *index, SUint n)
{
    std::vector<SUint> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::stable_sort(perm.begin(), perm.end(), [&](SUint a, SUint b) { return hash[a] < hash[b]; });
    std::vector<SUint> h2(n), i2(n);
    for (SUint k = 0; k < n; ++k) { h2[k] = hash[perm[k]]; i2[k] = index[perm[k]]; }
    std::copy(h2.begin(), h2.end(), hash);
    std::copy(i2.begin(), i2.end(), index);
}
/* cell ranges of reorderDataAndFindCellStartD(+Boundary): sph_kernel_impl.cuh:150-281,
 * preceded by the 0xff memset of cellStart (sph_cuda.cu:269,318). cellEnd of empty cells stays stale. */
static void k_cellRanges(const SUint *hash, SUint n, SUint *cellStart, SUint *cellEnd, SUint numCells)
{
    std::fill(cellStart, cellStart + numCells, EMPTY);
    for (SUint i = 0; i < n; ++i) {
        SUint h = hash[i];
        if (i == 0 || h != hash[i - 1]) {
            cellStart[h] = i;
            if (i > 0) cellEnd[hash[i - 1]] = i;
        }
        if (i == n - 1) cellEnd[h] = i + 1;
    }
}

/* Views handed to the gather "kernels" */
struct Grid {
    const SphSimParams *P;
    const SUint *cellStart, *cellEnd, *bCellStart, *bCellEnd, *bindex;
    const SVec4 *bi;   const SReal *vbi;    /* unsorted boundary, indexed through bindex (SESPH helpers) */
    const SVec4 *sbi;  const SReal *svbi;   /* sorted boundary, indexed directly (IISPH helpers) */
    int surf;                               /* USE_SURFACE_TENSION */
};

/* computeCellDensity / computeBoundaryCellDensity: sph_kernel_impl.cuh:290-360 */
static inline SReal cellDensity(const Grid &G, I3 gp, SUint self, SVec3 pos1, const SVec4 *sPos)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.cellStart[h];
    SReal dens = 0.0;
    if (s != EMPTY) {
        const SUint e = G.cellEnd[h];
        for (SUint j = s; j < e; ++j) {
            if (j != self) {
                SVec3 d = pos1 - mk3(sPos[j]);
                if (length(d) < P.interactionRadius)
                    dens += (P.particleMass * W_DENS(d, P.interactionRadius, P.kpoly));
            }
        }
    }
    return dens;
}
static inline SReal cellDensityBoundary(const Grid &G, I3 gp, SVec3 pos1)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.bCellStart[h];
    SReal dens = 0.0;
    if (s != EMPTY) {
        const SUint e = G.bCellEnd[h];
        for (SUint j = s; j < e; ++j) {
            const SUint o = G.bindex[j];
            SVec3 d = pos1 - mk3(G.bi[o]);
            if (length(d) < P.interactionRadius) {
                const SReal psi = P.restDensity * G.vbi[o];
                dens += (psi * W_DENS(d, P.interactionRadius, P.kpoly));
            }
        }
    }
    return dens;
}
/* density sum shared by computeDensityPressure (:365-433) and computeIisphDensity (:770-846) */
static inline SReal densityOf(const Grid &G, SUint slot, const SVec4 *sPos)
{
    const SphSimParams &P = *G.P;
    const SVec3 p = mk3(sPos[slot]);
    const I3 gp = calcGridPos(P, p);
    SReal dens = 0.0;
    dens += P.particleMass * W_DENS(mk3(0.0, 0.0, 0.0), P.interactionRadius, P.kpoly);
    for (int z = -1; z <= 1; z++)
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                dens += cellDensity(G, nb, slot, p, sPos);
                dens += cellDensityBoundary(G, nb, p);
            }
    return dens;
}

/* computeCellForces: sph_kernel_impl.cuh:442-604 */
static inline void cellForces(const Grid &G, SVec3 *fpres, SVec3 *fvisc, SVec3 *fsurf, SVec3 *fbound, I3 gp,
                              SUint self, SVec3 pos1, SVec3 vel1, SReal dens, SReal pres, const SVec4 *sPos,
                              const SReal *sDens, const SReal *sPres, const SVec4 *sVel)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    SUint s = G.cellStart[h];
    const SReal pm = P.particleMass, m2 = P.particleMass, ir = P.interactionRadius, kp = P.kpoly;
    const SReal kappa = P.surfaceTension;
    const SReal kprg = P.kpress_grad, kvg = P.kvisc_grad, kvd = P.kvisc_denum;
    (void)kp; (void)kprg; (void)kvg; (void)kvd; (void)kappa;
    if (s != EMPTY) {
        const SUint e = G.cellEnd[h];
        for (SUint j = s; j < e; ++j) {
            if (j == self) continue;
            const SVec3 pos2 = mk3(sPos[j]);
            const SReal dens2 = sDens[j];
            const SReal pres2 = sPres[j];
            const SVec3 vel2 = mk3(sVel[j]);
            const SVec3 p1p2 = pos1 - pos2;
            if (length(p1p2) < ir) {
                const SReal diameter = 2.0 * P.particleRadius;
                const SReal diameter2 = diameter * diameter;
                const SVec3 v1v2 = vel1 - vel2;
                const SReal d1sq = dens * dens;
                const SReal d2sq = dens2 * dens2;
#if KERNEL_SET == MONAGHAN
                const SVec3 kpressure_grad = Wmonaghan_grad(p1p2, ir);
                const SVec3 kvisco_grad = kpressure_grad;
                const SReal kernel = Wmonaghan(p1p2, ir);
                const SReal kernel_diameter = Wmonaghan(mk3(diameter, 0.0, 0.0), ir);
#else
                const SVec3 kpressure_grad = Wpressure_grad(p1p2, ir, kprg);
                const SVec3 kvisco_grad = Wviscosity_grad(p1p2, ir, kvg, kvd);
                const SReal kernel = Wdefault(p1p2, ir, kp);
                const SReal kernel_diameter = Wdefault(mk3(diameter, 0.0, 0.0), ir, kp);
#endif
                *fpres = *fpres + (m2 * (pres / d1sq + pres2 / d2sq) * kpressure_grad);
                const SReal a = dot(p1p2, kvisco_grad);
                const SReal b = dot(p1p2, p1p2) + 0.01f * (ir * ir);
                *fvisc = *fvisc + (m2 / dens2 * v1v2 * (a / b));
                if (G.surf) { /* #if USE_SURFACE_TENSION == 1, sph_kernel_impl.cuh:535-548 */
                    SVec3 ai = mk3(0.0, 0.0, 0.0);
                    const SReal r2 = dot(p1p2, p1p2);
                    if (r2 > diameter2)
                        ai = ai - (kappa / pm * pm * p1p2 * kernel);
                    else
                        ai = ai - (kappa / pm * pm * p1p2 * kernel_diameter);
                    *fsurf = *fsurf + ai;
                }
            }
        }
    }
    /* boundary part: no distance test */
    s = G.bCellStart[h];
    const SReal epsilon = 0.01;
    const SReal beta = P.beta;
    const SReal rd = P.restDensity;
    if (s != EMPTY) {
        const SUint e = G.bCellEnd[h];
        for (SUint j = s; j < e; ++j) {
            const SUint o = G.bindex[j];
            const SReal vbi = G.vbi[o];
            const SVec3 vpos = mk3(G.bi[o]);
            const SReal psi = (rd * vbi);
            const SVec3 p1p2 = pos1 - vpos;
            const SVec3 v1v2 = vel1;
#if KERNEL_SET == MONAGHAN
            const SReal kernel = Wmonaghan(p1p2, ir);
            const SVec3 grad = Wmonaghan_grad(p1p2, ir);
#else
            const SReal kernel = Wdefault(p1p2, ir, P.kpoly);
            const SVec3 grad = Wdefault_grad(p1p2, ir, P.kpoly_grad);
#endif
            *fbound = *fbound + (beta * psi * p1p2 * kernel);
            *fpres = *fpres + (-pm * psi * (pres / (dens * dens)) * grad);
            const SReal nu = (P.viscosity * ir * P.soundSpeed) / (dens * dens);
            const SReal nom = std::fmax((double)dot(v1v2, p1p2), 0.0);
            const SReal denom = dot(p1p2 / length(p1p2), p1p2 / length(p1p2)) + epsilon * ir * ir;
            const SReal Pij = -nu * (nom / denom);
            *fvisc = *fvisc - (pm * psi * Pij * grad);
        }
    }
}

/* computeDensityPressure kernel: sph_kernel_impl.cuh:365-433 */
static void k_densityPressure(Sim &S, const Grid &G)
{
    const SphSimParams &P = S.P;
#pragma omp parallel for schedule(static) num_threads(S.threads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        SReal dens = densityOf(G, slot, S.sPos.data());
        SReal pressure;
        if (S.taitDouble7) {
            const double x = (double)(float)(dens / P.restDensity), x2 = x * x, x4 = x2 * x2;
            pressure = P.gasStiffness * ((float)(x4 * x2 * x) - 1);
        } else {
            pressure = P.gasStiffness * (powf(dens / P.restDensity, 7) - 1);
        }
        S.sDens[slot] = dens;
        S.sPres[slot] = pressure;
    }
}
/* computeForces kernel: sph_kernel_impl.cuh:609-680 */
static void k_forces(Sim &S, const Grid &G)
{
    const SphSimParams &P = S.P;
#pragma omp parallel for schedule(static) num_threads(S.threads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        const SVec3 pos = mk3(S.sPos[slot]);
        const SVec3 vel = mk3(S.sVel[slot]);
        const SReal dens = S.sDens[slot];
        const SReal pres = S.sPres[slot];
        const SReal m1 = P.particleMass;
        const I3 gp = calcGridPos(P, pos);
        SVec3 fpres = mk3(0, 0, 0), fvisc = mk3(0, 0, 0), fsurf = mk3(0, 0, 0), fbound = mk3(0, 0, 0);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    cellForces(G, &fpres, &fvisc, &fsurf, &fbound, nb, slot, pos, vel, dens, pres, S.sPos.data(),
                               S.sDens.data(), S.sPres.data(), S.sVel.data());
                }
        fpres = fpres * dens;
        fvisc = fvisc * 2.0;
        fpres = fpres * -(m1 / dens);
        fvisc = fvisc * (m1 * P.viscosity);
        SVec3 f = fpres + fvisc + (P.gravity * m1) + fsurf + fbound;
        S.sForces[slot] = mk4(f.x, f.y, f.z, 0);
    }
}
/* integrate_functor via thrust::for_each: sph_kernel_impl.cuh:71-100, sph_cuda.cu:211-225 */
static void k_integrate(Sim &S)
{
    const SReal dt = S.P.timestep, m1 = S.P.particleMass;
    for (SUint i = 0; i < S.N; ++i) {
        SVec3 pos = mk3(S.sPos[i]), vel = mk3(S.sVel[i]), frc = mk3(S.sForces[i]);
        SVec3 accel = dt * frc / m1;
        vel = vel + accel;
        pos = pos + dt * vel;
        S.sPos[i] = mk4(pos.x, pos.y, pos.z, S.sPos[i].w);
        S.sVel[i] = mk4(vel.x, vel.y, vel.z, S.sVel[i].w);
    }
}

/* ----------------------------------- IISPH kernels ---------------------------------------- */
/* computeDisplacementFactorCell / ...BoundaryCell: sph_kernel_impl.cuh:689-765 */
static inline SVec3 dispCell(const Grid &G, SReal dens, I3 gp, SVec3 pos1, const SVec4 *sPos, SUint self)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.cellStart[h];
    SVec3 res = mk3(0, 0, 0);
    /* NB the caller passes kpoly_grad in the `kp` slot and pm in `pm` (sph_kernel_impl.cuh:952) */
    const SReal ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad;
    (void)kpg;
    if (s != EMPTY) {
        const SUint e = G.cellEnd[h];
        for (SUint j = s; j < e; ++j) {
            if (j == self) continue;
            const SVec3 d = pos1 - mk3(sPos[j]);
            if (length(d) < ir) {
                SVec3 grad = W_GRAD(d, ir, kpg);
                res = res - ((pm / (dens * dens)) * grad);
            }
        }
    }
    return res;
}
static inline SVec3 dispCellBoundary(const Grid &G, SReal dens, I3 gp, SVec3 pos1)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.bCellStart[h];
    SVec3 res = mk3(0, 0, 0);
    const SReal ir = P.interactionRadius, rd = P.restDensity, kpg = P.kpoly_grad;
    (void)kpg;
    if (s != EMPTY) {
        const SUint e = G.bCellEnd[h];
        for (SUint j = s; j < e; ++j) {
            const SVec3 d = pos1 - mk3(G.sbi[j]);
            const SReal vbi = G.svbi[j];
            const SReal psi = rd * vbi;
            if (length(d) < ir) {
                SVec3 grad = W_GRAD(d, ir, kpg);
                res = res - ((psi / (dens * dens)) * grad);
            }
        }
    }
    return res;
}
/* computeIisphDensity: sph_kernel_impl.cuh:770-846 */
static void k_iisphDensity(Sim &S, const Grid &G)
{
#pragma omp parallel for schedule(static) num_threads(S.threads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        S.sDens[slot] = densityOf(G, slot, S.sPos.data());
    }
}
/* computeDisplacementFactor: sph_kernel_impl.cuh:851-963 */
static void k_displacementFactor(Sim &S, const Grid &G)
{
    const SphSimParams &P = S.P;
#pragma omp parallel for schedule(static) num_threads(S.threads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        const SVec3 pos1 = mk3(S.sPos[slot]);
        const SVec3 vel1 = mk3(S.sVel[slot]);
        const SReal pres = 0.0;
        const SReal dens = S.sDens[slot];
        const SReal pm = P.particleMass, dt = P.timestep;
        const I3 gp = calcGridPos(P, pos1);
        SVec3 fvisc = mk3(0, 0, 0), fsurf = mk3(0, 0, 0), fgrav = mk3(0, 0, 0), fbound = mk3(0, 0, 0),
              fpres = mk3(0, 0, 0);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    cellForces(G, &fpres, &fvisc, &fsurf, &fbound, nb, slot, pos1, vel1, dens, pres, S.sPos.data(),
                               S.sDens.data(), S.sPres.data(), S.sVel.data());
                }
        fvisc = 2.0 * fvisc;
        fvisc = (pm * P.viscosity) * fvisc;
        fgrav = pm * P.gravity;
        SVec3 force_adv = fvisc + fsurf + fbound + fgrav;
        SVec3 vel_adv = vel1 + dt * (force_adv / pm);
        S.forcesAdv[slot] = mk4(force_adv.x, force_adv.y, force_adv.z, 0.0);
        S.velAdv[slot] = mk4(vel_adv.x, vel_adv.y, vel_adv.z, 0.0);
        SVec3 df = mk3(0, 0, 0), db = mk3(0, 0, 0);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    df = df + dispCell(G, dens, nb, pos1, S.sPos.data(), slot);
                    db = db + dispCellBoundary(G, dens, nb, pos1);
                }
        S.diiF[slot] = mk4(df.x, df.y, df.z, 0.0);
        S.diiB[slot] = mk4(db.x, db.y, db.z, 0.0);
    }
}
/* rho_adv_fluid / rho_adv_boundary / compute_aii_cell(_boundary): sph_kernel_impl.cuh:968-1108 */
static inline SReal rhoAdvFluid(const Grid &G, SUint self, SVec3 pos1, SVec3 velAdv1, const SVec4 *sPos,
                                const SVec4 *velAdv, I3 gp)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.cellStart[h];
    const SReal ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad;
    (void)kpg;
    SReal res = 0.0;
    if (s != EMPTY) {
        const SUint e = G.cellEnd[h];
        const SReal dt = P.timestep;
        for (SUint j = s; j < e; ++j) {
            if (j == self) continue;
            const SVec3 pos2 = mk3(sPos[j]);
            const SVec3 velAdv2 = mk3(velAdv[j]);
            const SVec3 v1v2 = velAdv1 - velAdv2;
            const SVec3 d = pos1 - pos2;
            if (length(d) < ir) {
                SVec3 grad = W_GRAD(d, ir, kpg);
                res += (dt * pm * dot(v1v2, grad));
            }
        }
    }
    return res;
}
static inline SReal rhoAdvBoundary(const Grid &G, SVec3 pos1, SVec3 vel1, I3 gp)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.bCellStart[h];
    const SReal ir = P.interactionRadius, rd = P.restDensity, kpg = P.kpoly_grad;
    (void)kpg;
    SReal res = 0.0;
    if (s != EMPTY) {
        const SUint e = G.bCellEnd[h];
        const SReal dt = P.timestep;
        for (SUint j = s; j < e; ++j) {
            const SVec3 bpos = mk3(G.sbi[j]);
            const SReal vbi = G.svbi[j];
            const SVec3 d = pos1 - bpos;
            const SVec3 v1v2 = vel1;
            const SReal psi = (rd * vbi);
            SVec3 grad = W_GRAD(d, ir, kpg);
            res += (dt * psi * dot(v1v2, grad));
        }
    }
    return res;
}
static inline SReal aiiCell(const Grid &G, SReal dens, SVec3 pos1, SVec3 diif, SVec3 diib, const SVec4 *sPos, I3 gp,
                            SUint self)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.cellStart[h];
    const SReal ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad;
    (void)kpg;
    SReal res = 0.0;
    if (s != EMPTY) {
        const SUint e = G.cellEnd[h];
        for (SUint j = s; j < e; ++j) {
            if (j == self) continue;
            const SVec3 d = pos1 - mk3(sPos[j]);
            const SReal dpi = (pm) / (dens * dens);
            SVec3 grad = W_GRAD(d, ir, kpg);
            SVec3 dji = dpi * grad;
            res += (pm * dot((diif + diib) - dji, grad));
        }
    }
    return res;
}
static inline SReal aiiCellBoundary(const Grid &G, SReal dens, SVec3 diif, SVec3 diib, SVec3 pos1, I3 gp)
{
    const SphSimParams &P = *G.P;
    const SUint h = calcGridHash(P, gp);
    const SUint s = G.bCellStart[h];
    const SReal ir = P.interactionRadius, rd = P.restDensity, kpg = P.kpoly_grad, pm = P.particleMass;
    (void)kpg;
    SReal res = 0.0;
    if (s != EMPTY) {
        const SUint e = G.bCellEnd[h];
        for (SUint j = s; j < e; ++j) {
            const SVec3 d = pos1 - mk3(G.sbi[j]);
            const SReal vbi = G.svbi[j];
            const SReal psi = rd * vbi;
            const SReal dpi = (pm) / (dens * dens);
            SVec3 grad = W_GRAD(d, ir, kpg);
            const SVec3 dji = dpi * grad;
            res += psi * dot((diif + diib) - dji, grad);
        }
    }
    return res;
}
/* computeAdvectionFactor: sph_kernel_impl.cuh:1114-1218 */
static void k_advectionFactor(Sim &S, const Grid &G)
{
    const SphSimParams &P = S.P;
#pragma omp parallel for schedule(static) num_threads(S.threads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        const SVec3 pos1 = mk3(S.sPos[slot]);
        const SVec3 vel1 = mk3(S.sVel[slot]);
        const SVec3 velAdv1 = mk3(S.velAdv[slot]);
        const SReal dens = S.sDens[slot];
        const SVec3 diif = mk3(S.diiF[slot]);
        const SVec3 diib = mk3(S.diiB[slot]);
        const I3 gp = calcGridPos(P, pos1);
        SReal rho_advf = 0.0, rho_advb = 0.0;
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    rho_advf += rhoAdvFluid(G, slot, pos1, velAdv1, S.sPos.data(), S.velAdv.data(), nb);
                    rho_advb += rhoAdvBoundary(G, pos1, vel1, nb);
                }
        SReal rho_adv = dens + (rho_advf + rho_advb);
        S.densAdv[slot] = rho_adv;
        S.P_l[slot] = 0.5 * S.sPres[slot];
        SReal aii = 0.0;
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    aii += aiiCell(G, dens, pos1, diif, diib, S.sPos.data(), nb, slot);
                    aii += aiiCellBoundary(G, dens, diif, diib, pos1, nb);
                }
        S.aii[slot] = aii;
    }
}
/* dijpjcell + computeSumDijPj: sph_kernel_impl.cuh:1224-1325 */
static void k_sumDijPj(Sim &S, const Grid &G)
{
    const SphSimParams &P = S.P;
#pragma omp parallel for schedule(static) num_threads(S.threads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        const SVec3 pos1 = mk3(S.sPos[slot]);
        const I3 gp = calcGridPos(P, pos1);
        const SReal ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad;
        (void)kpg;
        SVec3 dijpj = mk3(0, 0, 0);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    SVec3 res = mk3(0, 0, 0);
                    const SUint h = calcGridHash(P, nb);
                    const SUint s = G.cellStart[h];
                    if (s != EMPTY) {
                        const SUint e = G.cellEnd[h];
                        for (SUint j = s; j < e; ++j) {
                            if (j == slot) continue;
                            const SVec3 d = pos1 - mk3(S.sPos[j]);
                            const SReal p_lj = S.P_l[j];
                            const SReal densj = S.sDens[j];
                            SVec3 grad = W_GRAD(d, ir, kpg);
                            res = res - ((pm / (densj * densj)) * p_lj * grad);
                        }
                    }
                    dijpj = dijpj + res;
                }
        S.sumDij[slot] = mk4(dijpj.x, dijpj.y, dijpj.z, 0.0);
    }
}
/* computePressure: sph_kernel_impl.cuh:1330-1492.  Q5: self-exclusion tests the THREAD id `t`,
 * Q6: the boundary loop starts at the FLUID cell start, Q7: P_l read/written in place (racy on a GPU);
 * jacobi=1 reads neighbours' P_l from a snapshot taken before the sweep. */
static void k_pressure(Sim &S, const Grid &G)
{
    const SphSimParams &P = S.P;
    std::vector<SReal> snap;
    const SReal *Pold = S.P_l.data();
    if (S.jacobi) { snap = S.P_l; Pold = snap.data(); }
    const int nthreads = S.jacobi ? S.threads : 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        const SVec3 pos1 = mk3(S.sPos[slot]);
        const SReal dens = S.sDens[slot];
        SReal p_l = Pold[slot];
        const SReal previous_p_l = p_l;
        const SReal rho_adv = S.densAdv[slot];
        const SReal aii = S.aii[slot];
        const SVec3 dijpj = mk3(S.sumDij[slot]);
        const I3 gp = calcGridPos(P, pos1);
        const SReal ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad, dt = P.timestep,
                    rd = P.restDensity;
        (void)kpg;
        SReal fsum = 0.0, bsum = 0.0;
        const SReal dpi = pm / (dens * dens);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    const SUint h = calcGridHash(P, nb);
                    const SUint s = G.cellStart[h];
                    if (s != EMPTY) {
                        const SUint e = G.cellEnd[h];
                        for (SUint j = s; j < e; ++j) {
                            if (j == (S.selfBySlot ? slot : t)) continue; /* Q5 */
                            const SVec3 d = pos1 - mk3(S.sPos[j]);
                            const SReal p_lj = Pold[j];
                            SVec3 grad = W_GRAD(d, ir, kpg);
                            const SVec3 dji = dpi * (grad);
                            const SVec3 d_ji_pi = dji * p_lj;
                            const SVec3 diifj = mk3(S.diiF[j]);
                            const SVec3 diibj = mk3(S.diiB[j]);
                            const SVec3 sum_dijj = mk3(S.sumDij[j]);
                            fsum += pm * dot(dijpj - (diifj + diibj) * p_lj - (sum_dijj - d_ji_pi), grad);
                        }
                    }
                    const SUint sB = G.bCellStart[h];
                    if (sB != EMPTY) {
                        const SUint eB = G.bCellEnd[h];
                        for (SUint j = s; j < eB; ++j) { /* Q6 */
                            const SVec3 d = pos1 - mk3(G.sbi[j]);
                            const SReal vbi = G.svbi[j];
                            const SReal psi = rd * vbi;
                            SVec3 grad = W_GRAD(d, ir, kpg);
                            bsum += psi * dot(dijpj, grad);
                        }
                    }
                }
        SReal omega = 0.5;
        SReal rho_corr = rho_adv + fsum + bsum;
        const SReal dt2 = dt * dt;
        const SReal denom = aii * dt2;
        const SReal b = rd - rho_adv;
        if (std::fabs(denom) > FLT_EPSILON)
            p_l = (1.0 - omega) * previous_p_l + (omega / denom) * (b - dt2 * (bsum + fsum));
        else
            p_l = 0.0;
        SReal p = std::fmax((double)p_l, 0.0);
        p_l = p;
        rho_corr += aii * previous_p_l;
        S.P_l[slot] = p_l;
        S.sPres[slot] = p_l;
        S.densCorr[slot] = rho_corr;
    }
}
/* computePressureForce: sph_kernel_impl.cuh:1497-1620 (same Q5/Q6) */
static void k_pressureForce(Sim &S, const Grid &G)
{
    const SphSimParams &P = S.P;
#pragma omp parallel for schedule(static) num_threads(S.threads)
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        const SVec3 pos1 = mk3(S.sPos[slot]);
        const SReal p = S.sPres[slot];
        const SReal dens = S.sDens[slot];
        const I3 gp = calcGridPos(P, pos1);
        const SReal ir = P.interactionRadius, pm = P.particleMass, kpg = P.kpoly_grad, rd = P.restDensity;
        (void)kpg;
        SVec3 fp = mk3(0, 0, 0);
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    I3 nb = {wadd(gp.x, x), wadd(gp.y, y), wadd(gp.z, z)};
                    const SUint h = calcGridHash(P, nb);
                    const SUint s = G.cellStart[h];
                    if (s != EMPTY) {
                        const SUint e = G.cellEnd[h];
                        for (SUint j = s; j < e; ++j) {
                            if (j == (S.selfBySlot ? slot : t)) continue; /* Q5 */
                            const SVec3 d = pos1 - mk3(S.sPos[j]);
                            const SReal pj = S.sPres[j];
                            const SReal densj = S.sDens[j];
                            SVec3 grad = W_GRAD(d, ir, kpg);
                            const SVec3 contrib = -pm * pm * (p / (dens * dens) + pj / (densj * densj)) * grad;
                            fp = fp + contrib;
                        }
                    }
                    const SUint sB = G.bCellStart[h];
                    if (sB != EMPTY) {
                        const SUint eB = G.bCellEnd[h];
                        for (SUint j = s; j < eB; ++j) { /* Q6 */
                            const SVec3 d = pos1 - mk3(G.sbi[j]);
                            const SReal vbi = G.svbi[j];
                            const SReal psi = rd * vbi;
                            SVec3 grad = W_GRAD(d, ir, kpg);
                            const SVec3 contrib = (pm * psi * (p / (dens * dens)) * grad);
                            fp = fp + contrib;
                        }
                    }
                }
        S.forcesP[slot] = mk4(fp.x, fp.y, fp.z, 0.0);
    }
}
/* iisph_integrate: sph_kernel_impl.cuh:1625-1655 */
static void k_iisphIntegrate(Sim &S)
{
    const SReal dt = S.P.timestep, pm = S.P.particleMass;
    for (SUint t = 0; t < S.N; ++t) {
        const SUint slot = S.index[t];
        const SVec3 pos1 = mk3(S.sPos[slot]);
        const SVec3 velAdv1 = mk3(S.velAdv[slot]);
        const SVec3 fpres1 = mk3(S.forcesP[slot]);
        SVec3 newVel = velAdv1 + (dt * fpres1 / pm);
        SVec3 newPos = pos1 + (dt * newVel);
        S.sPos[slot] = mk4(newPos.x, newPos.y, newPos.z, 1.0);
        S.sVel[slot] = mk4(newVel.x, newVel.y, newVel.z, 0.0);
    }
}

static Grid makeGrid(Sim &S)
{
    S.ensureCells();
    Grid G;
    G.P = &S.P;
    G.cellStart = S.cellStart.data(); G.cellEnd = S.cellEnd.data();
    G.bCellStart = S.bCellStart.data(); G.bCellEnd = S.bCellEnd.data();
    G.bindex = S.bindex.data();
    G.bi = S.bi.data(); G.vbi = S.vbi.data();
    G.sbi = S.sbi.data(); G.svbi = S.svbi.data();
    G.surf = S.surf;
    return G;
}

/* hash → sort → reorder common prefix of SPH::update (sph.cpp:233-260) and IISPH::update (iisph.cpp:172-200) */
static void stagePrefix(Sim &S, int stop)
{
    S.ensureCells();
    S.dpos = S.pos;   /* H2D */
    S.dvel = S.vel;
    S.dpres = S.pressure;
    k_calcHash(S.P, S.dpos.data(), S.N, S.hash.data(), S.index.data());
    if (stop == 1) return;
    k_sort(S.hash.data(), S.index.data(), S.N);
    if (stop == 2) return;
    k_cellRanges(S.hash.data(), S.N, S.cellStart.data(), S.cellEnd.data(), S.P.numCells);
    for (SUint i = 0; i < S.N; ++i) { /* gather of reorderDataAndFindCellStartD :272-279 */
        SUint src = S.index[i];
        S.sVel[i] = S.dvel[src];
        S.sPos[i] = S.dpos[src];
        S.sPres[i] = S.dpres[src];
    }
}

enum { STOP_HASH = 1, STOP_SORT = 2, STOP_REORDER = 3, STOP_DENSITY = 4, STOP_FORCES = 5, STOP_NONE = 100 };

/* SPH::update(): sph/sph.cpp:215-285 */
static void sesphStep(Sim &S, int stop)
{
    stagePrefix(S, stop);
    if (stop <= STOP_REORDER) return;
    Grid G = makeGrid(S);
    k_densityPressure(S, G);
    if (stop == STOP_DENSITY) return;
    k_forces(S, G);
    if (stop == STOP_FORCES) return;
    k_integrate(S);
    S.pos = S.sPos;   /* D2H from the SORTED arrays (Q2) */
    S.vel = S.sVel;
}
/* IISPH::update(): sph/iisph/iisph.cpp:170-217; predictAdvection sph_cuda.cu:513-697; pressureSolve :702-899.
 * stop codes ≥ 10 are IISPH-specific: 10 density, 11 displacement, 12 advection, 13 after the solver loop,
 * 14 pressure force. */
static void iisphStep(Sim &S, int stop, int maxIters)
{
    stagePrefix(S, stop);
    if (stop <= STOP_REORDER) return;
    Grid G = makeGrid(S);
    k_iisphDensity(S, G);
    if (stop == 10) return;
    k_displacementFactor(S, G);
    if (stop == 11) return;
    k_advectionFactor(S, G);
    if (stop == 12) return;
    SUint l = 0;
    SReal rho_avg = 0.f;
    const SReal rd = 1000.f;
    const SReal max_rho_err = 1.f;
    /* maxIters < 0 (slab checker engine, tests/slab_check_engine.py): exactly -maxIters iterations, whatever the LOCAL average
     * says — in a slab run the exit test uses the average over all ranks */
    while (maxIters < 0 ? (int)l < -maxIters : (((rho_avg - rd) > max_rho_err) || (l < 2))) {
        k_sumDijPj(S, G);
        k_pressure(S, G);
        /* thrust::reduce order is unspecified; both this oracle and the HIP path accumulate in double */
        double acc = 0.0;
        for (SUint i = 0; i < S.N; ++i) acc += (double)S.densCorr[i];
        rho_avg = (SReal)acc;
        rho_avg /= S.N;
        l++;
        if (maxIters > 0 && (int)l >= maxIters) break;
    }
    S.lastIters = l;
    if (stop == 13) return;
    k_pressureForce(S, G);
    if (stop == 14) return;
    k_iisphIntegrate(S);
    S.pos = S.sPos;
    S.vel = S.sVel;
    S.pressure = S.sPres;
}

/* ----------------------------------- host-side restatements -------------------------------- */
/* Pre-computed kernel constants of the four constructors.  They differ only in where M_PI is cast:
 *   ctor 0  SPH::SPH()              sph/sph.cpp:76-90     (SReal)M_PI, 2.0*(SReal)M_PI in kvisc_grad
 *   ctor 1  IISPH::IISPH()          iisph/iisph.cpp:70-80 double M_PI everywhere
 *   ctor 2  SPH/IISPH(SphSimParams) sph.cpp:105-116, iisph.cpp:98-109  (SReal)M_PI, 2*(SReal)M_PI (float product) */
static void kernelConstants(SphSimParams &p, int ctor)
{
    const SReal ir = p.interactionRadius;
    if (ctor != 1) {
        p.kpoly = 315.0 / (64.0 * (SReal)M_PI * powf(ir, 9.0));
        p.kpoly_grad = -945.0 / (32.0 * (SReal)M_PI * powf(ir, 9.0));
        p.kpress_grad = -45.0 / ((SReal)M_PI * powf(ir, 6.0));
        if (ctor == 0) p.kvisc_grad = 15.0 / (2.0 * (SReal)M_PI * powf(ir, 3.0));
        else           p.kvisc_grad = 15.0 / (2 * (SReal)M_PI * powf(ir, 3.0));
        p.kvisc_denum = 2.0 * powf(ir, 3.0);
        p.ksurf1 = 32.0 / ((SReal)M_PI * powf(ir, 9.0));
    } else {
        p.kpoly = 315.0 / (64.0 * M_PI * powf(ir, 9.0));
        p.kpoly_grad = -945.0 / (32.0 * M_PI * powf(ir, 9.0));
        p.kpress_grad = -45.0 / (M_PI * powf(ir, 6.0));
        p.kvisc_grad = 15.0 / (2 * M_PI * powf(ir, 3.0));
        p.kvisc_denum = 2.0 * powf(ir, 3.0);
        p.ksurf1 = 32.0 / (M_PI * powf(ir, 9));
    }
    p.ksurf2 = powf(ir, 6) / 64.0;
    p.bpol = 0.007f / (powf(ir, 3.25));
}
static SReal soundSpeedDefault()
{
    const SReal eta = 0.01;
    const SReal H = 0.1;
    const SReal vf = std::sqrt(2.0 * 9.81 * H);
    return vf / (std::sqrt(eta));
}
/* SPH::SPH(): sph/sph.cpp:29-93 */
static void defaultsSESPH(SphSimParams &p)
{
    std::memset(&p, 0, sizeof(p));
    p.gasStiffness = 800;
    p.restDensity = 1000;
    p.particleRadius = 0.02;
    p.timestep = 1E-3;
    p.viscosity = 0.005;
    p.surfaceTension = 0.01;
    p.gravity.x = 0.; p.gravity.y = -9.81; p.gravity.z = 0.;
    p.interactionRadius = 0.0457;
    p.particleMass = 0.5 * powf(p.interactionRadius, 3) * p.restDensity;
    p.beta = 450.0;
    p.soundSpeed = soundSpeedDefault();
    p.worldOrigin = mk3(-1.1, -1.1, -1.1);
    p.gridSize.x = p.gridSize.y = p.gridSize.z = 64;
    p.cellSize = mk3(p.interactionRadius, p.interactionRadius, p.interactionRadius);
    p.numCells = p.gridSize.x * p.gridSize.y * p.gridSize.z;
    kernelConstants(p, 0);
}
/* IISPH::IISPH(): sph/iisph/iisph.cpp:28-87 (runs after the base constructor) */
static void defaultsIISPH(SphSimParams &p)
{
    defaultsSESPH(p);
    p.restDensity = 1000.0;
    p.particleRadius = 0.02;
    p.timestep = 1e-3;
    p.viscosity = 0.01;
    p.surfaceTension = 0.01;
    p.gravity.x = 0.0; p.gravity.y = -9.81f; p.gravity.z = 0.0;
    p.interactionRadius = 0.0537;
    p.particleMass = 0.5 * powf(p.interactionRadius, 3) * p.restDensity;
    p.beta = 1050.0;
    p.soundSpeed = soundSpeedDefault();
    p.worldOrigin = mk3(-1.2, -1.2, -1.2);
    p.gridSize.x = p.gridSize.y = p.gridSize.z = 128;
    p.cellSize = mk3(p.interactionRadius, p.interactionRadius, p.interactionRadius);
    p.numCells = p.gridSize.x * p.gridSize.y * p.gridSize.z;
    kernelConstants(p, 1);
}
/* nextPower2 + SPH::updateGrid: sph/sph.cpp:300-337; BBMin/BBMax sph_cuda.cu:461-505 */
static SUint nextPower2(SUint v)
{
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
    return v;
}
static void updateGrid(SphSimParams &p, const SVec4 *bi, SUint nb)
{
    SVec3 mn = mk3(bi[0]), mx = mk3(bi[0]);
    for (SUint i = 1; i < nb; ++i) {
        if (bi[i].x < mn.x) mn.x = bi[i].x;
        if (bi[i].y < mn.y) mn.y = bi[i].y;
        if (bi[i].z < mn.z) mn.z = bi[i].z;
        if (mx.x < bi[i].x) mx.x = bi[i].x;
        if (mx.y < bi[i].y) mx.y = bi[i].y;
        if (mx.z < bi[i].z) mx.z = bi[i].z;
    }
    p.worldOrigin = mk3(mn.x - 0.1, mn.y - 0.1, mn.z - 0.1);
    SUint sizex = std::ceil((mx.x - mn.x + 0.1) / p.interactionRadius);
    SUint sizey = std::ceil((mx.y - mn.y + 0.1) / p.interactionRadius);
    SUint sizez = std::ceil((mx.z - mn.z + 0.1) / p.interactionRadius);
    p.gridSize.x = nextPower2(sizex);
    p.gridSize.y = nextPower2(sizey);
    p.gridSize.z = nextPower2(sizez);
    p.numCells = p.gridSize.x * p.gridSize.y * p.gridSize.z;
}

/* =============================== C interface for ctypes ===================================== */
extern "C" {

int orc_sizeof_real(void) { return (int)sizeof(SReal); }
int orc_kernel_set(void) { return KERNEL_SET; }
int orc_sizeof_params(void) { return (int)sizeof(SphSimParams); }

void orc_default_params(int solver, void *out)
{
    SphSimParams p;
    if (solver == 1) defaultsIISPH(p); else defaultsSESPH(p);
    std::memcpy(out, &p, sizeof(p));
}
/* recompute the precomputed kernel parts for a changed interactionRadius, SPH(SphSimParams) ctor rule
 * (sph.cpp:98-118) */
void orc_recompute_constants(void *params)
{
    SphSimParams p;
    std::memcpy(&p, params, sizeof(p));
    kernelConstants(p, 2);
    std::memcpy(params, &p, sizeof(p));
}

/* Batch evaluation of the smoothing kernels / vector helpers, same numbering as oracle/ref_kernels_driver.cpp
 * (which calls the reference's own functions): tests/test_oracle_ref_pin.py compares the two bit for bit. */
int orc_eval(int which, unsigned n, const SReal *r3, const SReal *s3, SReal h, SReal c0, SReal c1, SReal *out)
{
    for (unsigned i = 0; i < n; ++i) {
        const SVec3 r = mk3(r3[3 * i], r3[3 * i + 1], r3[3 * i + 2]);
        SVec3 s = mk3(0, 0, 0);
        if (s3) s = mk3(s3[3 * i], s3[3 * i + 1], s3[3 * i + 2]);
        SVec3 v = mk3(0, 0, 0);
        switch (which) {
        case 0: v.x = Wdefault(r, h, c0); break;
        case 1: v = Wdefault_grad(r, h, c0); break;
        case 2: v = Wpressure_grad(r, h, c0); break;
        case 3: v = Wviscosity_grad(r, h, c0, c1); break;
        case 4: v.x = Wmonaghan(r, h); break;
        case 5: v = Wmonaghan_grad(r, h); break;
        case 6: v.x = Cakinci(r, h, c0, c1); break;
        case 7: v.x = Aboundary(r, h, c0); break;
        case 8: v.x = dot(r, s); break;
        case 9: v.x = length(r); break;
        case 10: v = r * (float)c0; break;
        case 11: v = (float)c0 * r; break;
        case 12: v = r / (float)c0; break;
        case 13: v = mk3(mk4(r.x, r.y, r.z, (SReal)7)); break;
        case 14: v = r + s; break;
        case 15: v = r - s; break;
        default: return -1;
        }
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
    return 0;
}

void *orc_create(const void *params)
{
    Sim *S = new Sim();
    std::memcpy(&S->P, params, sizeof(SphSimParams));
    return S;
}
void orc_destroy(void *h) { delete (Sim *)h; }
void orc_set_params(void *h, const void *params) { std::memcpy(&((Sim *)h)->P, params, sizeof(SphSimParams)); }
void orc_get_params(void *h, void *params) { std::memcpy(params, &((Sim *)h)->P, sizeof(SphSimParams)); }
void orc_set_mode(void *h, int jacobi, int threads)
{
    Sim *S = (Sim *)h;
    S->jacobi = jacobi;
    S->threads = threads < 1 ? 1 : threads;
}
void orc_set_self_by_slot(void *h, int on) { ((Sim *)h)->selfBySlot = on; }
void orc_set_surface_tension(void *h, int on) { ((Sim *)h)->surf = on ? 1 : 0; }
void orc_set_tait_mode(void *h, int double7) { ((Sim *)h)->taitDouble7 = double7 ? 1 : 0; }
void orc_set_particles(void *h, const SReal *pos4, const SReal *vel4, const SReal *pres, SUint n)
{
    Sim *S = (Sim *)h;
    S->resizeParticles(n);
    std::memcpy(S->pos.data(), pos4, sizeof(SVec4) * n);
    std::memcpy(S->vel.data(), vel4, sizeof(SVec4) * n);
    if (pres) std::memcpy(S->pressure.data(), pres, sizeof(SReal) * n);
    else std::fill(S->pressure.begin(), S->pressure.end(), (SReal)0);
}
/* SPH::generateParticleCube + addNewParticle: sph/sph.cpp:341-386.  Returns the count; writes up to cap. */
SUint orc_generate_cube(const void *params, const SReal *center, const SReal *size, SReal *pos4, SUint cap)
{
    SphSimParams P;
    std::memcpy(&P, params, sizeof(P));
    SUint n = 0;
    for (SReal x = center[0] - size[0] / 2.0; x <= center[0] + size[0] / 2.0; x += P.interactionRadius - 0.005f)
        for (SReal y = center[1] - size[1] / 2.0; y <= center[1] + size[1] / 2.0; y += P.interactionRadius - 0.005f)
            for (SReal z = center[2] - size[2] / 2.0; z <= center[2] + size[2] / 2.0;
                 z += P.interactionRadius - 0.005f) {
                if (n < cap) { pos4[4 * n + 0] = x; pos4[4 * n + 1] = y; pos4[4 * n + 2] = z; pos4[4 * n + 3] = 1.0; }
                n++;
            }
    return n;
}
/* SPH::updateGpuBoundaries: sph/sph.cpp:391-432 (intended semantics, SURVEY Q1: sorted copies ARE filled) */
void orc_set_boundaries(void *h, const SReal *bi4, const SReal *vbi, SUint nb, int update_grid)
{
    Sim *S = (Sim *)h;
    S->Nb = nb;
    S->bi.resize(nb); S->vbi.resize(nb); S->sbi.resize(nb); S->svbi.resize(nb);
    S->bhash.resize(nb); S->bindex.resize(nb);
    if (nb) {
        std::memcpy(S->bi.data(), bi4, sizeof(SVec4) * nb);
        std::memcpy(S->vbi.data(), vbi, sizeof(SReal) * nb);
        if (update_grid) updateGrid(S->P, S->bi.data(), nb);
    }
    S->cellStart.clear(); S->bCellStart.clear();
    S->ensureCells();
    if (nb) {
        k_calcHash(S->P, S->bi.data(), nb, S->bhash.data(), S->bindex.data());
        k_sort(S->bhash.data(), S->bindex.data(), nb);
        k_cellRanges(S->bhash.data(), nb, S->bCellStart.data(), S->bCellEnd.data(), S->P.numCells);
        for (SUint i = 0; i < nb; ++i) { S->sbi[i] = S->bi[S->bindex[i]]; S->svbi[i] = S->vbi[S->bindex[i]]; }
    }
}
/* solver: 0 SESPH, 1 IISPH.  stop: see enum / iisphStep. */
void orc_step(void *h, int solver, int stop, int max_iters)
{
    Sim *S = (Sim *)h;
    if (S->N == 0) return;
    if (solver == 1) iisphStep(*S, stop <= 0 ? STOP_NONE : stop, max_iters);
    else sesphStep(*S, stop <= 0 ? STOP_NONE : stop);
}
SUint orc_num_particles(void *h) { return ((Sim *)h)->N; }
SUint orc_last_iters(void *h) { return ((Sim *)h)->lastIters; }

/* Array access by name; returns element count (in scalars of the array's type) or -1. */
long orc_get(void *h, const char *name, void *dst)
{
    Sim *S = (Sim *)h;
#define GETV(nm, vec)                                                                      \
    if (!std::strcmp(name, nm)) {                                                          \
        if (dst) std::memcpy(dst, (vec).data(), (vec).size() * sizeof((vec)[0]));          \
        return (long)((vec).size() * sizeof((vec)[0]));                                    \
    }
    GETV("pos", S->pos) GETV("vel", S->vel) GETV("pressure", S->pressure)
    GETV("hash", S->hash) GETV("index", S->index) GETV("cellStart", S->cellStart) GETV("cellEnd", S->cellEnd)
    GETV("sortedPos", S->sPos) GETV("sortedVel", S->sVel) GETV("dens", S->sDens) GETV("pres", S->sPres)
    GETV("forces", S->sForces)
    GETV("bhash", S->bhash) GETV("bindex", S->bindex) GETV("bCellStart", S->bCellStart) GETV("bCellEnd", S->bCellEnd)
    GETV("sbi", S->sbi) GETV("svbi", S->svbi)
    GETV("densAdv", S->densAdv) GETV("densCorr", S->densCorr) GETV("P_l", S->P_l) GETV("aii", S->aii)
    GETV("velAdv", S->velAdv) GETV("forcesAdv", S->forcesAdv) GETV("forcesP", S->forcesP)
    GETV("diiFluid", S->diiF) GETV("diiBoundary", S->diiB) GETV("sumDij", S->sumDij)
#undef GETV
    return -1;
}

} /* extern "C" */
