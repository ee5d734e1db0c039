// nrs_kernels_tiled.h — production gather kernels of the SESPH step for gfx950.
//
// Shape of the work: per particle ≈40-55 candidate neighbours in the 27 surrounding
// cells, of which only ≈6-10 lie inside the support radius.  A one-pass "test and accumulate" loop
// wastes 80-85 % of every 64-lane wavefront in the heavy branch.  These kernels therefore run two phases
// per thread (one thread per SORTED slot, so own loads/stores are coalesced 16 B/lane):
//
//   scan    walk the 9 (dz,dy) rows; the three x-cells of a row are ONE contiguous run of the sorted array
//           (hash = (z*gy+y)*gx+x), so a row is a single [lo,hi) sweep of float4 positions with a
//           squared-distance compare against a precomputed threshold (no sqrt, no divide).  The loop is
//           organised for memory-level parallelism (measured: the plain cell walk is latency-bound with one
//           dependent load in flight per thread): 27 table entries per z-plane requested at once, candidate
//           positions fetched SCAN_BATCH (4) at a time.  Hits are appended to a per-thread list kept in LDS
//           (lst[k][tid], k-major: conflict-free).  This exact-position form (Sweep::scan) serves the wall workgroups, the
//           kernels without shared lists and contexts whose geometry rules the quantised form out.
//           Round 2: the interior workgroups scan 4-BYTE QUANTISED candidates instead (Sweep::scan_compact: position modulo
//           four cells, 10 bits per axis, written by the reorder kernels; integer superset test, branch-free append), and
//           the process phase applies the exact float cut-off to the exact position it gathers anyway — same lists, same
//           sums.  Measured: these kernels are bound by vector-instruction issue (VALUBusy 85-90 %), with a latency term
//           that six waves per SIMD do not hide completely.
//   process the compacted hits (nearly equal counts across lanes → dense wavefronts) get the expensive
//           kernel evaluation, in the same order the reference visits them.
//
// Every floating-point sum is formed in the reference's order (per-cell partial sums for the density,
// running sums for the forces; every hit carries its cell number, fluid and boundary hits are merged back
// into the reference's cell-by-cell order), so the results are
// bit-identical to the reference-order kernels in nrs_kernels_ref.h — which are also the overflow path
// for a thread whose hit list would exceed HIT_CAP (correct for any neighbour count).
//
// The density kernel publishes its lists to global memory (k-major, hits[k][i]) and the force kernel of the same step
// walks them (k_forces_lists: no second scan, no LDS), fetching the next list head one hit ahead.
//
// No MFMA: this is an issue/latency-bound gather with ~2 kflop per particle-step, not a dense contraction.
#pragma once
#include "nrs_kernels_ref.h"
#include "nrs_kernels_slab.h"
#include <type_traits>

namespace nrs {

constexpr int HIT_CAP = 20; // hits kept per thread.  LDS = HIT_CAP*BLOCK*4 B = 20 KiB per workgroup ⇒ 8 workgroups/CU: measured
                            // 32 → 20 entries: density 0.221 → 0.159 ms, forces 0.273 → 0.211 ms (2.1 M particles); 16 gives no more
// A hit is one u32: bits 31..27 = neighbour-cell number 0..26 in the reference's z,y,x visiting order,
// bits 26..0 = index into the sorted fluid array (or the sorted boundary array).
constexpr uint32_t HIT_INDEX = (1u << 27) - 1;
constexpr int HIT_TAG_SHIFT = 27;

// Thresholds that turn the reference's two cut-off predicates into one float compare on the float dot
// product d2 = dot(r,r)  (exact: sqrtf and the products are monotone, correctly rounded):
//   lenLtIr : smallest float T with  sqrtf(T) >= ir          ⇒  (length(r) <  ir)        ⇔ d2 < T
//   r2LeH2  : smallest float T with  fl(sqrtf(T)^2) > h*h    ⇒  !(length(r)^2 > h^2)     ⇔ d2 < T
struct CutThresholds { float lenLtIr, r2LeH2; };

// waves per SIMD the fp32 list-driven force kernel is compiled for: 7 caps it at 72 VGPRs (3 dwords spilled with
// boundaries) and measured 0.487 -> 0.465 ms at 10 M particles; 8 (64 VGPRs) spills into the hit loop: 0.563 ms
#ifndef FORCES_LISTS_MIN_WAVES
#define FORCES_LISTS_MIN_WAVES 7
#endif
#ifndef FORCES_PACKED_MIN_WAVES
#define FORCES_PACKED_MIN_WAVES 5
#endif
// the force launch of a PARTIAL step (forces array out, no integration; tests and diagnostics): its register allocation spills ten dwords
// into the hit loop at the 96-VGPR bound where the fused launch spills none (0.99 against 0.79 ms) — bounded for four waves instead
#ifndef FORCES_PACKED_MIN_WAVES_UNFUSED
#define FORCES_PACKED_MIN_WAVES_UNFUSED 4
#endif
// value of `v` in lane `srcLane` (wave-uniform lane number), for every lane
NRS_DEV float bcast_lane(float v, int srcLane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srcLane)); }
NRS_DEV double bcast_lane(double v, int srcLane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), srcLane), hi = __builtin_amdgcn_readlane((int)(b >> 32), srcLane);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
#ifndef NRS_COMPACT_SCAN
#define NRS_COMPACT_SCAN 1 // interior workgroups scan the 4-byte quantised candidates (Sweep::scan_compact); 0: the exact positions
#endif
#ifndef NRS_FORCE_PAIRS
#define NRS_FORCE_PAIRS 1 // the list-driven force kernel gathers (p / rho^2, m / rho) pairs written by the density kernel (HitBuffer::pairs)
#endif
#ifndef NRS_DBG_LDS_PAD
#define NRS_DBG_LDS_PAD 0 // bytes of dynamic LDS added to every density workgroup (occupancy experiments only)
#endif
#ifndef NRS_DBG_LDS_PAD_F
#define NRS_DBG_LDS_PAD_F 0 // ... to every fused force workgroup
#endif
#ifndef QP_WALK
#define QP_WALK 4 // list entries whose exact positions are gathered together in density_from_superset
#endif
#ifndef QP_PRE
#define QP_PRE 2 // (4: 104 VGPRs unbounded, 5 dwords spilled at the 80-VGPR bound, 0.649 vs 0.583 ms) dwordx4 candidate loads (two candidates each) per row issued before the first test (x 3 rows of a z-plane)
#endif
constexpr int SCAN_CAP = 21; // list entries the compact scan can hold: HIT_CAP + the particle itself (+ 3 spill rows behind them)
// a * b + c on 24-bit signed operands in ONE instruction (the compiler prefers three multiplies and a three-operand add)
NRS_DEV int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
constexpr int SCAN_BATCH = 4; // candidate positions fetched per thread per memory round trip (6: 79 VGPRs = 6 waves, 0.81 vs 0.71 ms; 8: slower still)

// Result of the scan phase: fluid hits are lst[0 .. nf) (ascending), boundary hits are
// lst[HIT_CAP-1 .. HIT_CAP-nb] (descending slots, ascending visiting order); both lists are ordered by cell
// number, so the process phase restores the reference's order (cell by cell: fluid, then boundary) by merging.
struct HitCounts { int nf, nb; bool over; bool anyB; }; // anyB: some of the 27 cells holds boundary particles
// NRS_GATHER_INTERLEAVED (the gather records of HitBuffer): the two records of a slot side by side (slot j at 32 j bytes: both loads of a hit fall into one 128-byte line)
#ifndef NRS_GATHER_INTERLEAVED
#define NRS_GATHER_INTERLEAVED 1
#endif
constexpr uint32_t GATHER_STRIDE = NRS_GATHER_INTERLEAVED ? 2u : 1u;
template <typename R> struct PrePair { R prq, mrho; }; // (p / rho^2, m / rho) of one sorted slot, see HitBuffer::pairs

template <typename R> struct Sweep {
    typedef typename Vec4T<R>::type T4;
    // element `idx` of a vec4 array through a 32-bit byte offset from the (wave-uniform) base pointer: lets the
    // compiler use the scalar-base + 32-bit-vector-offset form of global_load (no 64-bit address arithmetic, two
    // VGPRs less per address in flight).  Valid because capacities are below 2^27 elements (nrs_create checks).
    static NRS_DEV T4 at32(const T4 *__restrict__ base, uint32_t idx)
    {
        const uint32_t off = idx * (uint32_t)sizeof(T4);
        return *reinterpret_cast<const T4 *>(reinterpret_cast<const char *>(base) + off);
    }

    // BFILT: cut-off used for boundary candidates: 0 = lenLtIr (explicit test of the density loop); 1 = r2LeH2
    // (the force loop has no explicit test, but every Muller kernel it evaluates returns 0 beyond h);
    // 2 = none (Monaghan kernels reach 2h, so every boundary particle of the 27 cells contributes).
    //
    // The loop is built for memory-level parallelism and few branches (profiled: the plain cell walk keeps one
    // dependent load in flight per thread, and a branchy scan spends as many scalar exec-mask instructions as
    // vector ones): the cell-table entries of a whole z-plane (3 rows x 3 cells, start and end) are requested
    // together; the three x-cells of a row are ONE run [lo,hi) of the sorted array; candidate positions are
    // fetched SCAN_BATCH at a time and tested branch-free; only the (rare) hit takes a branch.  Boundary cells
    // are swept afterwards, only by the lanes that have any, so the fluid sweep stays uniform across the wave.
    // FWIDE: fluid candidates are kept up to r2LeH2 instead of lenLtIr (IISPH: several of its loops have no explicit
    // cut-off and rely on the kernel gradient being 0 beyond h); pass self = 0xffffffff to keep the
    // particle itself in the list (two IISPH kernels do not exclude it).
    template <bool HAS_B, int BFILT, int W, bool FWIDE = false>
    static NRS_DEV HitCounts scan(const Params<R> &P, const GridView<R> &G, const CutThresholds thr,
                                  const T4 *__restrict__ sPos, uint32_t self, V3<R> p, uint32_t (*lst)[W])
    {
        const I3 gp = calcGridPos<R>(P, p);
        const uint32_t mx = P.gridSize[0] - 1, my = P.gridSize[1] - 1, mz = P.gridSize[2] - 1;
        const uint32_t cx = grid_x<R>(P, gp.x);
        const uint32_t x0 = (cx - 1u) & mx, x2 = (cx + 1u) & mx;
        const bool contiguous = (cx >= 1u) && (cx + 1u <= mx);
        const float tF = FWIDE ? thr.r2LeH2 : thr.lenLtIr;
        const float tB = BFILT == 2 ? INFINITY : (BFILT == 1 ? thr.r2LeH2 : thr.lenLtIr);
        const uint32_t tid = threadIdx.x;
        int nf = 0, nb = 0;
        bool over = false, anyB = false;

        // fluid candidates j in [a, b); the cell number advances at j == m1 and j == m2 (starts of the 2nd / 3rd
        // cell of a merged run; CELL_EMPTY when that cell is empty)
        auto sweepFluid = [&](uint32_t a, uint32_t b, uint32_t m1, uint32_t m2, uint32_t tag0) {
            const uint32_t nT = run_ok<R>(G, a, b) ? b - a : 0u;
            for (uint32_t base = 0; base < nT; base += SCAN_BATCH) {
                T4 c[SCAN_BATCH];
#pragma unroll
                for (int u = 0; u < SCAN_BATCH; ++u) c[u] = at32(sPos, a + min(base + (uint32_t)u, nT - 1u));
#pragma unroll
                for (int u = 0; u < SCAN_BATCH; ++u) {
                    const uint32_t q = base + (uint32_t)u;
                    const uint32_t j = a + q;
                    const V3<R> d = p - xyz<R>(c[u]);
                    const bool hit = (q < nT) & (j != self) & (dot(d, d) < tF);
                    if (hit) {
                        uint32_t tag = tag0 + (((m1 != CELL_EMPTY) & (j >= m1)) ? 1u : 0u);
                        if ((m2 != CELL_EMPTY) & (j >= m2)) tag = tag0 + 2u;
                        if (nf + nb < HIT_CAP) lst[nf][tid] = j | (tag << HIT_TAG_SHIFT); else over = true;
                        ++nf;
                    }
                }
            }
        };

        for (int z = -1; z <= 1; z++) {
            const uint32_t cz = (uint32_t)(gp.z + z) & mz;
            const uint32_t plane = umul24(umul24(cz, P.gridSize[1]), P.gridSize[0]);
            // all table entries of this z-plane in one go: 3 rows x {start,end} x 3 cells (+ boundary starts)
            uint32_t st[3][3], en[3][3], hrow[3];
            uint32_t bmask = 0; // bit (y*3+c): boundary particles in that cell
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                const uint32_t cy = (uint32_t)(gp.y + y - 1) & my;
                hrow[y] = plane + umul24(cy, P.gridSize[0]);
                const uint32_t h[3] = {hrow[y] + x0, hrow[y] + cx, hrow[y] + x2};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    st[y][c] = G.cellStart[h[c]];
                    en[y][c] = G.cellEnd[h[c]];
                    if (HAS_B) bmask |= (G.bCellStart[h[c]] != CELL_EMPTY) ? (1u << (y * 3 + c)) : 0u;
                }
            }
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                const uint32_t tag0 = (uint32_t)((z + 1) * 9 + y * 3);
                const uint32_t s0 = st[y][0], s1 = st[y][1], s2 = st[y][2];
                if (contiguous) {
                    // one contiguous run of the sorted array: [first non-empty start, last non-empty end)
                    uint32_t lo = (s0 != CELL_EMPTY) ? s0 : ((s1 != CELL_EMPTY) ? s1 : s2);
                    uint32_t hi = (s2 != CELL_EMPTY) ? en[y][2] : ((s1 != CELL_EMPTY) ? en[y][1] : en[y][0]);
                    if (lo == CELL_EMPTY) lo = hi = 0;
                    sweepFluid(lo, hi, s1, s2, tag0);
                } else { // the 3-cell window wraps around the grid edge: cell by cell
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (st[y][c] != CELL_EMPTY) sweepFluid(st[y][c], en[y][c], CELL_EMPTY, CELL_EMPTY, tag0 + (uint32_t)c);
                }
            }
            if (HAS_B) {
#if defined(NRS_ABL_NOBSWEEP) // timing ablation: boundary cells are not swept
                bmask = 0u;
#endif
                anyB = anyB || bmask != 0u;
                // BFILT 2 (Monaghan, shared lists): EVERY boundary particle of the 27 cells contributes to the force loop, so a list of
                // them is the boundary cell table itself — density_from_hits / forces_from_hits walk it (BOUNDARY_BY_CELL)
                if (BFILT == 2) bmask = 0u;
                // boundary cells of this plane, visited in ascending cell number by the lanes that have any
                // (a wave-cooperative sweep — one wall lane at a time, 64 candidates per round, hits ranked with a ballot — was
                // measured SLOWER, 0.756 vs 0.711 ms at 10 M particles: the two wall lanes of a wave already share one instruction
                // stream, and the cooperative form pays its broadcasts and table reloads per wall lane)
                while (bmask) {
                    const int bit = __builtin_ctz(bmask);
                    bmask &= bmask - 1u;
                    const int y = bit / 3, c = bit - y * 3;
                    const uint32_t cy = (uint32_t)(gp.y + y - 1) & my;
                    const uint32_t hc = plane + umul24(cy, P.gridSize[0]) + (c == 0 ? x0 : (c == 1 ? cx : x2));
                    const uint32_t a = G.bCellStart[hc], nT = G.bCellEnd[hc] - a;
                    const uint32_t tag = (uint32_t)((z + 1) * 9 + bit);
                    for (uint32_t base = 0; base < nT; base += 4) {
                        T4 cb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) cb[u] = G.sB[a + min(base + (uint32_t)u, nT - 1u)];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t q = base + (uint32_t)u;
                            const V3<R> d = p - xyz<R>(cb[u]);
                            if ((q < nT) & (dot(d, d) < tB)) {
                                if (nf + nb < HIT_CAP) lst[HIT_CAP - 1 - nb][tid] = (a + q) | (tag << HIT_TAG_SHIFT); else over = true;
                                ++nb;
                            }
                        }
                    }
                }
            }
        }
        HitCounts hc;
        // An owner with a NaN / inf coordinate (a caller's bug) fails every distance test, so its lists are empty — but the reference's
        // loops without a cut-off (boundary particles in the force loop) multiply its infinite distances into their sums
        // (0 * inf = NaN): such an owner takes the reference-order walk, as a list overflow does.  (Found by the randomised soak,
        // round 3: seed 20728, a particle with y = inf next to a wall sheet.)
        const bool finite = (fabs(p.x) < (R)INFINITY) & (fabs(p.y) < (R)INFINITY) & (fabs(p.z) < (R)INFINITY); // (false for NaN too)
        hc.nf = nf; hc.nb = nb; hc.over = over | !finite; hc.anyB = anyB;
        return hc;
    }

    // ---- compact scan (interior code: no boundary particles) ---------------------------------------------------------------
    // Same walk as scan() — z-plane by z-plane, the three x-cells of a row as one run — but the candidates are the 8-byte words
    // of G.qpos (nrs_math.h, quantize_pos) and the test is an integer squared distance against G.qT (two v_pk_sub_i16 + two
    // v_dot2_i32_i16): the list is a SUPERSET of the exact hits (< 1 % more entries), in the visiting order.  Two candidates
    // travel per global_load_dwordx4, and the first QP_PRE loads of all three rows of a plane are issued together before the
    // first test; a load may run one slot past the end of its row (masked by q < nT; the array is padded).
    // The append is BRANCH-FREE: every slot stores its entry at the thread's write cursor and the cursor advances only on a hit,
    // so a miss is overwritten by the next slot (rows SCAN_CAP .. SCAN_CAP + 2 of the LDS array are the spill rows of a full
    // list).  In a wavefront some lane hits at almost every slot, so a branchy append ran its body ~70 times per wave anyway,
    // plus the exec-mask bookkeeping.  An entry carries the ROW number only (tag = 3 * row); which of the row's three cells the
    // candidate sits in — the reference's summation order needs it — is decided in density_from_superset() from the exact
    // position, once per HIT instead of once per slot; the particle itself is dropped there too (it costs one list slot here:
    // SCAN_CAP = HIT_CAP + 1).
    // (Measured and dropped: skipping the corner rows the owner cannot reach geometrically — the lanes of a wave rarely agree, 0.538
    // vs 0.546 ms at rest, 0.915 vs 0.910 developed.)
    typedef short qs2 __attribute__((ext_vector_type(2)));
    static constexpr int QSLOTS = 16 / QP_BYTES; // candidates per global_load_dwordx4
    struct __attribute__((packed, aligned(QP_BYTES))) Q2 { uint32_t v[4]; };
    static NRS_DEV Q2 ldq(const qword_t *__restrict__ qpos, uint32_t idx)
    {
        return *reinterpret_cast<const Q2 *>(reinterpret_cast<const char *>(qpos) + idx * (uint32_t)QP_BYTES);
    }
    template <int W>
    static NRS_DEV HitCounts scan_compact(const Params<R> &P, const GridView<R> &G, V3<R> p, uint32_t (*lst)[W])
    {
        const I3 gp = calcGridPos<R>(P, p);
        const uint32_t mx = P.gridSize[0] - 1, my = P.gridSize[1] - 1, mz = P.gridSize[2] - 1;
        const uint32_t cx = grid_x<R>(P, gp.x);
        const uint32_t x0 = (cx - 1u) & mx, x2 = (cx + 1u) & mx;
        const bool contiguous = (cx >= 1u) && (cx + 1u <= mx);
        float tx, ty, tz;
        quantize_t<R>(G.qc, p, tx, ty, tz);
        // an owner too far from the grid origin for the error budget of the quanta (or with a NaN coordinate) is served by the
        // reference-order walk, like a list overflow
        const bool far = !((fabsf(tx) < QP_FAR) & (fabsf(ty) < QP_FAR) & (fabsf(tz) < QP_FAR));
        const qword_t Qw = pack_quanta(tx, ty, tz);
#if QP_BYTES == 8
        const qs2 Qxy = __builtin_bit_cast(qs2, Qw.x), Qz = __builtin_bit_cast(qs2, Qw.y);
#else
        const uint32_t Qi = Qw | QP_GUARD;
#endif
        const uint32_t qT = G.qT;
        const qword_t *__restrict__ qpos = G.qpos;
        // write cursor: byte offset of lst[nf][tid] from lst[0][0]
        // (the clamp sits one row BEHIND the last valid row, so that a cursor that ever passed SCAN_CAP entries stays there: sticky overflow)
        const uint32_t col = threadIdx.x * 4u, rowBytes = (uint32_t)W * 4u, capOff = col + (uint32_t)(SCAN_CAP + 1) * rowBytes;
        uint32_t cur = col;
        char *const lbase = reinterpret_cast<char *>(&lst[0][0]);

        // candidates a + base .. a + base + QSLOTS - 1 of a run of nT slots; aTag = a | (3 * row) << HIT_TAG_SHIFT
        auto test2 = [&](const Q2 &c, uint32_t aTag, uint32_t base, uint32_t nT) {
#pragma unroll
            for (int u = 0; u < QSLOTS; ++u) {
                if ((u & 1) == 0) cur = min(cur, capOff);
#if QP_BYTES == 8
                const qs2 dxy = Qxy - __builtin_bit_cast(qs2, c.v[2 * u]), dz = Qz - __builtin_bit_cast(qs2, c.v[2 * u + 1]);
                const uint32_t d2 = (uint32_t)__builtin_amdgcn_sdot2(dz, dz, __builtin_amdgcn_sdot2(dxy, dxy, 0, false), false);
#else
                const uint32_t t = Qi - c.v[u];
                const int dx = ((int)(t << 22)) >> 22, dy = ((int)(t << 11)) >> 22, dz = ((int)t) >> 22;
                const uint32_t d2 = (uint32_t)mad24(dz, dz, mad24(dy, dy, __mul24(dx, dx)));
#endif
                const uint32_t q = base + (uint32_t)u;
                const bool hit = (d2 < qT) & (q < nT);
#if !defined(NRS_ABL_NOAPPEND)
                *reinterpret_cast<uint32_t *>(lbase + cur) = aTag + q;
#endif
                cur += hit ? rowBytes : 0u;
            }
        };
        auto sweepRun = [&](uint32_t aTag, uint32_t nT, uint32_t from) {
            for (uint32_t base = from; base < nT; base += QSLOTS) test2(ldq(qpos, (aTag & HIT_INDEX) + base), aTag, base, nT);
        };

        for (int z = -1; z <= 1; z++) {
            const uint32_t cz = (uint32_t)(gp.z + z) & mz;
            const uint32_t plane = umul24(umul24(cz, P.gridSize[1]), P.gridSize[0]);
            uint32_t st[3][3], en[3][3];
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                const uint32_t cy = (uint32_t)(gp.y + y - 1) & my;
                const uint32_t hrow = plane + umul24(cy, P.gridSize[0]);
                const uint32_t h[3] = {hrow + x0, hrow + cx, hrow + x2};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // one 32-bit byte offset serves both tables (scalar base + vector offset form of global_load: no 64-bit
                    // address arithmetic per entry); 4 * numCells <= 2^32 because a hash has at most 30 bits
                    const uint32_t off = h[c] * 4u;
                    st[y][c] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(G.cellStart) + off);
                    en[y][c] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(G.cellEnd) + off);
                }
            }
            if (contiguous) {
                uint32_t lo[3], nT[3];
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    const uint32_t s0 = st[y][0], s1 = st[y][1], s2 = st[y][2];
                    uint32_t a = (s0 != CELL_EMPTY) ? s0 : ((s1 != CELL_EMPTY) ? s1 : s2);
                    uint32_t b = (s2 != CELL_EMPTY) ? en[y][2] : ((s1 != CELL_EMPTY) ? en[y][1] : en[y][0]);
                    if (a == CELL_EMPTY) a = b = 0;
                    // a run that fails the guard (stale / corrupt table) is neither swept nor PREFETCHED: its loads go to slot 0
                    // (invariant otherwise: a <= b <= nSorted <= capacity, and qpos has QSLOTS slots of padding behind capacity)
                    const bool ok = run_ok<R>(G, a, b);
                    lo[y] = ok ? a : 0u;
                    nT[y] = ok ? b - a : 0u;
                }
                Q2 c[3][QP_PRE];
#pragma unroll
                for (int y = 0; y < 3; ++y)
#pragma unroll
                    for (int k = 0; k < QP_PRE; ++k) c[y][k] = ldq(qpos, lo[y] + min((uint32_t)(QSLOTS * k), nT[y]));
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    const uint32_t aTag = lo[y] | ((uint32_t)((z + 1) * 9 + y * 3) << HIT_TAG_SHIFT);
#pragma unroll
                    for (int k = 0; k < QP_PRE; ++k)
                        if (k == 0 || nT[y] > (uint32_t)(QSLOTS * k)) // (rows of neighbouring lanes are alike: a whole wave often skips this)
                            test2(c[y][k], aTag, (uint32_t)(QSLOTS * k), nT[y]);
                    sweepRun(aTag, nT[y], (uint32_t)(QSLOTS * QP_PRE));
                }
            } else { // the 3-cell window wraps around the grid edge: cell by cell, in the reference's order
#pragma unroll
                for (int y = 0; y < 3; ++y)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (st[y][c] != CELL_EMPTY && run_ok<R>(G, st[y][c], en[y][c]))
                            sweepRun(st[y][c] | ((uint32_t)((z + 1) * 9 + y * 3) << HIT_TAG_SHIFT), en[y][c] - st[y][c], 0u);
            }
        }
        const uint32_t nf = (cur - col) / rowBytes;
        HitCounts hc;
        hc.nf = (int)nf; hc.nb = 0; hc.over = far || nf > (uint32_t)SCAN_CAP; hc.anyB = false;
        return hc;
    }
};

// Merge cursor over the two hit lists: yields hits in the reference's order (cell 0 fluid, cell 0 boundary,
// cell 1 fluid, ...).  key = 2*cell + kind identifies the partial sum a hit belongs to.
// The lists are addressed as base[k * stride]: the thread's column of the LDS array (stride = workgroup size) or of
// the global hit buffer written by the density kernel (stride = particle capacity); both are k-major, so a wave
// reads/writes 64 consecutive words per k.
struct HitMerge {
    const uint32_t *base;
    uint32_t stride;
    int nf, nb, kf, kb;
    // the two list heads are fetched one step ahead: when a hit is handed out, the load of its successor is already on
    // its way and overlaps with the gathers and the arithmetic of the hit (one dependent memory round trip per hit
    // instead of two when the lists live in global memory)
    uint32_t headF, headB;
    NRS_DEV HitMerge(const uint32_t *b, uint32_t st, HitCounts hc) : base(b), stride(st), nf(hc.nf), nb(hc.nb), kf(0), kb(0)
    {
        headF = nf > 0 ? base[0] : 0xffffffffu;
        headB = nb > 0 ? base[(uint32_t)(HIT_CAP - 1) * stride] : 0xffffffffu;
    }
    NRS_DEV int last_boundary() const { return kb - 1; } // visiting number of the boundary hit next() just handed out
    NRS_DEV bool next(uint32_t &index, bool &boundary, uint32_t &key)
    {
        if (kf >= nf && kb >= nb) return false;
        const uint32_t tf = headF >> HIT_TAG_SHIFT, tb = headB >> HIT_TAG_SHIFT;
        boundary = tb < tf; // fluid first inside a cell; the 0xffffffff sentinel has tag 31 > 26
        const uint32_t e = boundary ? headB : headF;
        if (boundary) {
            ++kb;
            headB = kb < nb ? base[(uint32_t)(HIT_CAP - 1 - kb) * stride] : 0xffffffffu;
        } else {
            ++kf;
            headF = kf < nf ? base[(uint32_t)kf * stride] : 0xffffffffu;
        }
        index = e & HIT_INDEX;
        key = (e >> HIT_TAG_SHIFT) * 2u + (boundary ? 1u : 0u);
        return true;
    }
};

// ---- phase 2 of the density (computeDensityPressure, sph_kernel_impl.cuh:365-433): hits → rho -----------
// STRICT: the list is the wide IISPH one (self included, fluid kept up to r2LeH2): apply the density loop's own
// `j != self` and `length < h` tests (sph_kernel_impl.cuh:305-309).
template <typename R, int KSET, bool HAS_B, bool STRICT = false, bool BOUNDARY_BY_CELL = false>
NRS_DEV R density_from_hits(const Params<R> &P, const GridView<R> &G, const typename Vec4T<R>::type *__restrict__ sPos,
                            V3<R> p, const uint32_t *lbase, uint32_t lstride, HitCounts hc, uint32_t self = 0xffffffffu)
{
    const R ir = P.interactionRadius, kp = P.kpoly, pm = P.particleMass, rd = P.restDensity;
    R d = (R)0.0;
    d += pm * W_dens<R, KSET>(mk3<R>(0, 0, 0), ir, kp);
    if (BOUNDARY_BY_CELL) {
        // no boundary list (see Sweep::scan, BFILT 2): cell by cell in the reference's order, the fluid hits of the cell from the
        // list, then the cell's boundary particles straight from the boundary cell table; one partial sum per group, as below
        const I3 gp = calcGridPos<R>(P, p);
        int k = 0;
        uint32_t head = hc.nf > 0 ? lbase[0] : 0xffffffffu;
        uint32_t c = 0;
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++, c++) {
                    if ((head >> HIT_TAG_SHIFT) == c) {
                        R part = (R)0.0;
                        do {
                            const uint32_t j = head & HIT_INDEX;
                            const V3<R> r = p - xyz<R>(sPos[j]);
                            if (!STRICT || ((j != self) && (length(r) < ir))) part += (pm * W_dens<R, KSET>(r, ir, kp));
                            ++k;
                            head = k < hc.nf ? lbase[(uint32_t)k * lstride] : 0xffffffffu;
                        } while ((head >> HIT_TAG_SHIFT) == c);
                        d += part;
                    }
                    if (hc.anyB) {
                        const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                        const uint32_t sB = G.bCellStart[h];
                        if (sB != CELL_EMPTY) {
                            const uint32_t eB = G.bCellEnd[h];
                            R part = (R)0.0;
                            for (uint32_t j = sB; j < eB; ++j) {
                                const typename Vec4T<R>::type b = G.sB[j];
                                const V3<R> r = p - xyz<R>(b);
                                if (length(r) < ir) part += ((rd * b.w) * W_dens<R, KSET>(r, ir, kp));
                            }
                            d += part;
                        }
                    }
                }
        return d;
    }
    R part = (R)0.0; // the reference adds one partial sum per (cell, fluid|boundary)
    uint32_t prevKey = 0xffffffffu;
    HitMerge it(lbase, lstride, hc);
    uint32_t j, key;
    bool isB;
    while (it.next(j, isB, key)) {
        if (key != prevKey) { d += part; part = (R)0.0; prevKey = key; }
        if (HAS_B && isB) {
            const typename Vec4T<R>::type b = G.sB[j];
            const V3<R> r = p - xyz<R>(b);
            // the list may have been built with the wider cut-off of the force loop (shared lists): apply the
            // density loop's own test (sph_kernel_impl.cuh:347); contributions are unchanged when it was not
            if (length(r) < ir) {
                const R psi = rd * b.w;
                part += (psi * W_dens<R, KSET>(r, ir, kp));
            }
        } else {
            const V3<R> r = p - xyz<R>(sPos[j]);
            if (!STRICT || ((j != self) && (length(r) < ir))) part += (pm * W_dens<R, KSET>(r, ir, kp));
        }
    }
    d += part;
    return d;
}

// Process phase behind Sweep::scan_compact (no boundary list): the exact cut-off of the list (`dot(r,r) < tKeep`, the test scan()
// applies to its candidates) on the exact position, which the density sum needs anyway.  A survivor gets its full cell tag here
// — 3 * row (from the scan) + the cell's place in the row, from calcGridPos's own expression on the candidate's exact x — and is
// written back to the front of the thread's LDS column (w <= k): what is published afterwards is exactly the list scan() would
// have produced, and the sums are formed in the same order (one partial sum per cell).
// INRANGE (fp32, Muller kernels; the caller's wave-uniform decision, density_inrange_ok): the cell-tag division (x - origin) / cellSize and the
// square root of a kept entry run as the bare steps of nrs_math.h "operands in range".  Guards: the cell size in [2^-20, 2^10], and the
// numerator x - origin either +0 or at least 2^-90 — which holds for every particle when the grid origin itself is at least 2^-60 in
// magnitude (a non-zero difference of two floats is a multiple of the smaller ulp), and otherwise for the kept entries of an owner that is at
// least four cells from the origin (a kept entry is less than h < 1.99 cells from its owner); owners are at most 4096 cells from the origin
// (Sweep::scan_compact), so the quotient stays below 2^13.  The square root's argument is below tKeep; an entry below 2^-96 (coincident
// particles) takes sqrtf.
template <typename R, int KSET, bool STRICT, bool INRANGE = false>
NRS_DEV R density_from_superset(const Params<R> &P, const typename Vec4T<R>::type *__restrict__ sPos, V3<R> p, uint32_t (*lst)[BLOCK],
                                int &nf, float tKeep, uint32_t self)
{
    const uint32_t tid = threadIdx.x;
    const R ir = P.interactionRadius, kp = P.kpoly, pm = P.particleMass;
    const uint32_t mx = P.gridSize[0] - 1;
    const int gxi = (int)floor((p.x - P.worldOrigin[0]) / P.cellSize[0]);
    float yCell = 0.f;
    if constexpr (INRANGE) yCell = rcp_refined((float)P.cellSize[0]);
    R d = (R)0.0;
    d += pm * W_dens<R, KSET>(mk3<R>(0, 0, 0), ir, kp);
    R part = (R)0.0;
    uint32_t prevTag = 0xffffffffu;
    int w = 0;
    // the exact positions of QP_WALK entries are requested together (the entries are independent; a plain loop exposes one
    // memory round trip per entry)
    for (int k0 = 0; k0 < nf; k0 += QP_WALK) {
        uint32_t e[QP_WALK];
        V3<R> q[QP_WALK];
#pragma unroll
        for (int u = 0; u < QP_WALK; ++u) e[u] = lst[min(k0 + u, nf - 1)][tid];
#pragma unroll
        for (int u = 0; u < QP_WALK; ++u) q[u] = xyz<R>(sPos[e[u] & HIT_INDEX]);
#pragma unroll
        for (int u = 0; u < QP_WALK; ++u) {
            const V3<R> r = p - q[u];
            const uint32_t j = e[u] & HIT_INDEX;
            const float d2 = dot(r, r);
            if ((k0 + u < nf) && (d2 < tKeep) && (STRICT || j != self)) { // (STRICT: the wide IISPH list keeps the particle itself)
#if defined(NRS_ABL_NODIV) // timing ablation: row tags only (summation order within a row is then not the reference's)
                const uint32_t tag = (e[u] >> HIT_TAG_SHIFT);
#else
                int gxj;
                if constexpr (INRANGE) gxj = (int)floor(div_steps((float)(q[u].x - P.worldOrigin[0]), (float)P.cellSize[0], yCell));
                else gxj = (int)floor((q[u].x - P.worldOrigin[0]) / P.cellSize[0]);
                const uint32_t tag = (e[u] >> HIT_TAG_SHIFT) + (((uint32_t)(gxj - gxi) + 1u) & mx);
#endif
                lst[min(w, HIT_CAP - 1)][tid] = j | (tag << HIT_TAG_SHIFT);
                ++w;
                if (tag != prevTag) { d += part; part = (R)0.0; prevTag = tag; }
                if constexpr (INRANGE) {
                    // Wdefault (kernels_impl.cuh:85-98) on length(r) = sqrt(dot(r, r)), the square root without its scaling
                    const float len = d2 >= 0x1p-96f ? sqrt_inrange(d2) : sqrt_rn(d2);
                    if (!STRICT || ((j != self) && (len < ir))) {
                        const float r2 = len * len, h2 = ir * ir;
                        part += (pm * (r2 > h2 ? 0.0f : kp * cube_via_double<float>(h2 - r2)));
                    }
                } else {
                    if (!STRICT || ((j != self) && (length(r) < ir))) part += (pm * W_dens<R, KSET>(r, ir, kp));
                }
            }
        }
    }
    d += part;
    nf = w;
    return d;
}
// the decision behind INRANGE: true for the whole wave or for none of its lanes (launch constants, else a vote)
template <typename R> NRS_DEV bool density_inrange_ok(const Params<R> &P, V3<R> p)
{
    if constexpr (!std::is_same<R, float>::value) return false;
    else {
#if NRS_INRANGE_DIV
        const float cs = P.cellSize[0], wo = fabsf(P.worldOrigin[0]);
        if (!(cs >= 0x1p-20f && cs <= 0x1p10f && wo <= 0x1p60f)) return false;
        if (wo >= 0x1p-60f) return true;
        return __all(fabsf(p.x - P.worldOrigin[0]) >= 4.0f * cs) != 0;
#else
        return false;
#endif
    }
}

// The three boundary terms of one (fluid particle, boundary particle) pair, computeCellForces sph_kernel_impl.cuh:566-602:
// adhesion, pressure mirror, friction — evaluated here ONCE, by whichever lane does it (the owner, or a helper lane of the
// cooperative pre-pass of k_forces_lists), with the same operations in the same order; the owner adds them to its running
// sums in the reference's order.
template <typename R, int KSET> struct BoundaryTerms { V3<R> bound, pres, visc; };
template <typename R, int KSET>
NRS_DEV BoundaryTerms<R, KSET> boundary_terms(const Params<R> &P, V3<R> pos1, V3<R> vel1, R dens, R pres, typename Vec4T<R>::type bq)
{
    const R pm = P.particleMass, ir = P.interactionRadius;
    const R epsilon = (R)0.01;
    const R beta = P.beta, rd = P.restDensity;
    const R psi = (rd * bq.w);
    const V3<R> rij = pos1 - xyz<R>(bq);
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
    BoundaryTerms<R, KSET> T;
    T.bound = (beta * psi * rij * kernel);
    T.pres = (-pm * psi * (pres / (dens * dens)) * grad);
    const R nuWall = (P.viscosity * ir * P.soundSpeed) / (dens * dens);
    const R approach = (R)fmax((double)dot(vij, rij), 0.0);
    const R normSq = dot(rij / length(rij), rij / length(rij)) + epsilon * ir * ir;
    const R friction = -nuWall * (approach / normSq);
    T.visc = (pm * psi * friction * grad);
    return T;
}

// ---- two hits per iteration on packed fp32 (fp32, Muller kernels, interior code without boundary hits) ---------------------------
// gfx950 executes v_pk_mul_f32 / v_pk_add_f32 on two floats per lane and instruction, IEEE-rounded like their scalar forms.  The hit
// loop of the force walk is bound by vector-instruction issue, so two CONSECUTIVE hits of a particle are
// evaluated side by side — component .x of every pair is hit k, .y is hit k + 1 — with exactly the scalar code's operations in the
// scalar code's order per hit; the divisions, square roots and the double-precision cube stay scalar (no packed forms exist), the
// running sums are formed hit k first, then hit k + 1, as before: bit-identical by construction, checked against the reference-order
// kernels by the whole suite.
// Measured on the bench's own state (NS scene after 3000 steps at dt = 2.5e-4 s, tools/ab_flowing.sh): force stage 1.078 -> 0.887 ms
// (-18 %) with 94 VGPRs / 5 waves per SIMD; bounded to 80 / 72 VGPRs it spills into the loop and loses (1.11 / 1.45 ms); at rest (six
// hits per particle) 0.386 ms either way.  368 -> 272 vector instructions per two hits.
#ifndef NRS_PACKED_HITS
#define NRS_PACKED_HITS 1
#endif
// (f2, V3x2, splat2, pair2, dot2 and the packed forms of the in-range steps: nrs_math.h)
// Two IEEE divisions a.x / b.x, a.y / b.y.  The compiler expands a correctly rounded fp32 division into v_div_scale (x2), v_rcp, five
// fused multiply-adds and a multiply, v_div_fmas, v_div_fixup (AMDGPU LowerFDIV32); here the six arithmetic steps of the two
// divisions run as v_pk_fma_f32 / v_pk_mul_f32 on both at once — the same operations on the same operands, 16 instructions instead
// of 22, the quotients are the correctly rounded ones either way (tools/check_div2.hip: all 2^32 numerators for twelve denominators and
// 2^32 random operand pairs incl. zeros, denormals, infinities, NaNs: 0 differing quotients).  Bit-identical — and SLOWER: the force stage
// 1.053 ms against 0.887 ms with the compiler's own expansion (two more registers tip the 96-VGPR bound into spills; at 4 waves, no
// spills: 0.912 ms — the hand-ordered chain leaves the scheduler less to interleave).  Kept behind the macro, off.
#ifndef NRS_PACKED_DIV
#define NRS_PACKED_DIV 0
#endif
NRS_DEV f2 div2(f2 a, f2 b)
{
#if NRS_PACKED_DIV
    bool da, db, na, nb;
    const f2 den = pair2(__builtin_amdgcn_div_scalef(a.x, b.x, false, &da), __builtin_amdgcn_div_scalef(a.y, b.y, false, &db));
    const f2 num = pair2(__builtin_amdgcn_div_scalef(a.x, b.x, true, &na), __builtin_amdgcn_div_scalef(a.y, b.y, true, &nb));
    const f2 rcp = pair2(__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y));
    const f2 nden = -den;
    const f2 e0 = __builtin_elementwise_fma(nden, rcp, splat2(1.0f));
    const f2 y = __builtin_elementwise_fma(e0, rcp, rcp);
    const f2 q0 = num * y;
    const f2 e1 = __builtin_elementwise_fma(nden, q0, num);
    const f2 q1 = __builtin_elementwise_fma(e1, y, q0);
    const f2 e2 = __builtin_elementwise_fma(nden, q1, num);
    const float qa = __builtin_amdgcn_div_fmasf(e2.x, y.x, q1.x, na);
    const float qb = __builtin_amdgcn_div_fmasf(e2.y, y.y, q1.y, nb);
    return pair2(__builtin_amdgcn_div_fixupf(qa, b.x, a.x), __builtin_amdgcn_div_fixupf(qb, b.y, a.y));
#else
    return pair2(a.x / b.x, a.y / b.y);
#endif
}
// ---- the force walk's divisions and square roots for operands in range (nrs_math.h "operands in range") ----------------------------------
// Five of the six divisions of a hit and its square root run as the bare arithmetic steps of the compiler's own expansions, two hits at a
// time on packed fp32, behind three guards:
//   launch constants (wave-uniform): 2^-10 <= h <= 2^10, 2^-40 <= |kvisc_denum| <= 2^40;
//   owner (wave-uniform vote): every coordinate |x| >= 2^-66 — then a component of rij = pos1 - pos2 is either +0 or at least 2^-90 in
//     magnitude (a non-zero difference of two floats is a multiple of the smaller ulp; -0 needs a zero owner coordinate);
//   hit: 2^-16 h^2 <= dot(rij, rij) < 2 h^2 (one unsigned range test on the bits: NaN, inf and negative patterns fail it).
// With them: rij.c / rlen has |n| <= rlen (1 + 2^-22), rlen in [2^-18, 2^11]; 3 rlen / kvisc_denum lies in [2^-58, 2^53];
// h / (2 rlen^3) has its denominator in [2^-53, 2^34] and lies in [2^-44, 2^63] — all inside v_div_scale's pass-through region.  A wave whose
// vote fails walks with the compiler's divisions as before; a lane whose hit fails the range test repeats its walk that way after the loop.
// rij.c = +0: v_div_fixup returns +0, the steps return +0 too (0 * y = +0, fma(-d, +0, +0) = +0, fma(+0, y, +0) = +0).
// a / b of the viscosity term keeps the compiler's division (the range of a = dot(rij, gradVisc) does not follow from the guards).
// Measured (tools/ab_flowing.sh, tools/busy_flowing.sh): vector instructions per wave of the force launch 2280 -> 1710 in the bench's window
// and VALU busy 81 % -> 60 % — and the launch takes the same 0.88 ms there, because in that window it is bound by the L1 (0.57 line
// accesses per cycle and CU, TA 71-77 % busy: 34 distinct lines per gather instruction); at rest, six hits per particle, 0.382 -> 0.351 ms.
struct InRange { float yVisc; uint32_t loBits, spanBits; bool constantsOk; };
NRS_DEV InRange in_range_setup(const Params<float> &P)
{
    InRange G;
    const float h = P.interactionRadius, h2 = h * h, kvd = fabsf(P.kvisc_denum);
    G.constantsOk = h >= 0x1p-10f && h <= 0x1p10f && kvd >= 0x1p-40f && kvd <= 0x1p40f;
    G.yVisc = rcp_refined(P.kvisc_denum);
    G.loBits = __float_as_uint(h2 * 0x1p-16f);
    G.spanBits = __float_as_uint(h2 * 2.0f) - G.loBits;
    return G;
}

struct PairTerms2 { V3x2 pres, visc, surf; };
// the three pair terms of computeCellForces (sph_kernel_impl.cuh:520-548) for hits (a, b) of one owner; `own` = pres / (dens * dens)
// of the owner, c0 = kappa / pm * pm, both formed once per owner with the scalar code's operations.  INRANGE: the forms above; `bad` is
// set when one of the two hits is outside their range (the caller then repeats the owner's walk with INRANGE = false).
template <bool SURF, bool INRANGE>
NRS_DEV PairTerms2 fluid_terms2_muller(const Params<float> &P, V3<float> pos1, V3<float> vel1, float own, float c0, float wAtDiameter, float diameter2,
                                       float4 pa, float4 pb, float4 va, float4 vb, PrePair<float> qa, PrePair<float> qb, const InRange &G, bool &bad)
{
    const float ir = P.interactionRadius, m2 = P.particleMass;
    // The twelve differences are scalar subtractions, and stay so (the empty asm takes them as twelve scalars): as packed subtractions —
    // written so, or put together by the SLP vectoriser — they need the owner's six coordinates splat into register pairs for the whole
    // loop, which costs the fused launch a register spilled INTO the loop (0.87 -> 0.80 ms in the bench's window, 0.44 -> 0.39 at rest).
    float rxa = pos1.x - pa.x, rxb = pos1.x - pb.x, rya = pos1.y - pa.y, ryb = pos1.y - pb.y, rza = pos1.z - pa.z, rzb = pos1.z - pb.z;
    float vxa = vel1.x - va.x, vxb = vel1.x - vb.x, vya = vel1.y - va.y, vyb = vel1.y - vb.y, vza = vel1.z - va.z, vzb = vel1.z - vb.z;
    asm volatile("" : "+v"(rxa), "+v"(rxb), "+v"(rya), "+v"(ryb), "+v"(rza), "+v"(rzb));
    asm volatile("" : "+v"(vxa), "+v"(vxb), "+v"(vya), "+v"(vyb), "+v"(vza), "+v"(vzb));
    const V3x2 r = {pair2(rxa, rxb), pair2(rya, ryb), pair2(rza, rzb)};
    const V3x2 v = {pair2(vxa, vxb), pair2(vya, vyb), pair2(vza, vzb)};
    const f2 d2 = dot2(r, r);                       // dot(rij, rij)
    f2 len;                                         // length(rij)
    if (INRANGE) {
        bad = bad || (__float_as_uint(d2.x) - G.loBits >= G.spanBits) || (__float_as_uint(d2.y) - G.loBits >= G.spanBits);
        len = sqrt_inrange2(d2);
    } else {
        len = pair2(sqrt_rn(d2.x), sqrt_rn(d2.y));
    }
    const f2 r2 = len * len;
    const float h2 = ir * ir;
    const bool outA = r2.x > h2, outB = r2.y > h2;  // every Muller kernel returns 0 beyond h (kernels_impl.cuh:92,110,129,148)
    // Wpressure_grad: kpress_grad * (r / rlen) * c, c = (h - rlen)^2
    const f2 hm = splat2(ir) - len;
    const f2 c = hm * hm;
    V3x2 gs;
    // Wviscosity_grad: kvisc_grad * r * c, c = -(3 rlen / kvisc_denum) + (2 / h2) - (h / (2 rlen rlen rlen))
    const f2 t3 = splat2(3.0f) * len;
    const f2 l3 = splat2(2.0f) * len * len * len;
    f2 cv;
    if (INRANGE) {
        const f2 y = rcp_refined2(len);
        gs.x = splat2(P.kpress_grad) * div_steps2(r.x, len, y) * c;
        gs.y = splat2(P.kpress_grad) * div_steps2(r.y, len, y) * c;
        gs.z = splat2(P.kpress_grad) * div_steps2(r.z, len, y) * c;
        cv = -div_steps2(t3, splat2(P.kvisc_denum), splat2(G.yVisc)) + splat2(2.0f / h2) - div_steps2(splat2(ir), l3, rcp_refined2(l3));
    } else {
        gs.x = splat2(P.kpress_grad) * div2(r.x, len) * c;
        gs.y = splat2(P.kpress_grad) * div2(r.y, len) * c;
        gs.z = splat2(P.kpress_grad) * div2(r.z, len) * c;
        cv = -div2(t3, splat2(P.kvisc_denum)) + splat2(2.0f / h2) - div2(splat2(ir), l3);
    }
    V3x2 gv = {splat2(P.kvisc_grad) * r.x * cv, splat2(P.kvisc_grad) * r.y * cv, splat2(P.kvisc_grad) * r.z * cv};
    // Wdefault: kpoly * (h2 - r2)^3, the cube formed in double and rounded once
    const f2 hr = splat2(h2) - r2;
    f2 kern = splat2(P.kpoly) * pair2(cube_via_double<float>(hr.x), cube_via_double<float>(hr.y));
    if (outA) { gs.x.x = 0.f; gs.y.x = 0.f; gs.z.x = 0.f; gv.x.x = 0.f; gv.y.x = 0.f; gv.z.x = 0.f; kern.x = 0.f; }
    if (outB) { gs.x.y = 0.f; gs.y.y = 0.f; gs.z.y = 0.f; gv.x.y = 0.f; gv.y.y = 0.f; gv.z.y = 0.f; kern.y = 0.f; }
    PairTerms2 T;
    // fpres += m2 * (pres / rhoSqOwn + pOverRhoSqNb) * gradSpiky
    const f2 sp = splat2(m2) * (splat2(own) + pair2(qa.prq, qb.prq));
    T.pres.x = sp * gs.x; T.pres.y = sp * gs.y; T.pres.z = sp * gs.z;
    // fvisc += mOverRhoNb * vij * (a / b), a = dot(rij, gradVisc), b = dot(rij, rij) + 0.01 (ir * ir)
    const f2 a = dot2(r, gv);
    const f2 b = d2 + splat2(0.01f * (ir * ir));
    const f2 q = div2(a, b);
    const f2 mr = pair2(qa.mrho, qb.mrho);
    T.visc.x = mr * v.x * q; T.visc.y = mr * v.y * q; T.visc.z = mr * v.z * q;
    if (SURF) {
        // ai = 0 - (kappa / pm * pm * rij * (r2 > diameter2 ? kernel : wAtDiameter)), r2 = dot(rij, rij)
        const f2 ks = pair2(d2.x > diameter2 ? kern.x : wAtDiameter, d2.y > diameter2 ? kern.y : wAtDiameter);
        const f2 z = splat2(0.f);
        T.surf.x = z - splat2(c0) * r.x * ks; T.surf.y = z - splat2(c0) * r.y * ks; T.surf.z = z - splat2(c0) * r.z * ks;
    }
    return T;
}

constexpr int LIST_BATCH = 4; // list entries whose gathers a fluid-only list walk of the IISPH chain requests together
// Fluid entries only, LIST_BATCH at a time: the entries are fetched together and `gather(j)` — the loads of everything the walk needs of
// neighbour j, returned by value — is called for all of them before `use(j, tag, data)` runs entry by entry, in list order.  The list walks
// of the chain are bound by the round trip entry -> gathers -> arithmetic of each hit (config C3, k_sumdij_lists: 145 -> 107 us with four
// hits in flight and the same instructions), not by instruction issue; the sums are formed exactly as before.
template <int NB = LIST_BATCH, typename GATHER, typename USE>
NRS_DEV void walk_fluid_batched(const uint32_t *col, uint32_t stride, int nf, GATHER &&gather, USE &&use)
{
    for (int k0 = 0; k0 < nf; k0 += NB) {
        uint32_t e[NB];
        decltype(gather(0u)) d[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) e[u] = col[(size_t)min(k0 + u, nf - 1) * stride];
#pragma unroll
        for (int u = 0; u < NB; ++u) d[u] = gather(e[u] & HIT_INDEX);
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (k0 + u < nf) use(e[u] & HIT_INDEX, e[u] >> HIT_TAG_SHIFT, d[u]);
    }
}

// ---- phase 2 of the forces (computeCellForces, sph_kernel_impl.cuh:442-604): hits → accumulators ----------
template <typename R, int KSET, bool SURF, bool HAS_B, bool STRICT = false, bool PAIRS = false, bool BOUNDARY_BY_CELL = false>
NRS_DEV ForceAcc<R> forces_from_hits(const Params<R> &P, const GridView<R> &G,
                                     const typename Vec4T<R>::type *__restrict__ sPos,
                                     const typename Vec4T<R>::type *__restrict__ sVel, const R *__restrict__ sDens,
                                     const R *__restrict__ sPres, V3<R> pos1, V3<R> vel1, R dens, R pres,
                                     const uint32_t *lbase, uint32_t lstride, HitCounts hc, uint32_t self = 0xffffffffu,
                                     const R *pre = nullptr, const typename Vec4T<R>::type *__restrict__ gpos = nullptr,
                                     const typename Vec4T<R>::type *__restrict__ gvel = nullptr, const uint32_t *__restrict__ lroot = nullptr,
                                     uint32_t lcol = 0u)
{
    ForceAcc<R> A;
    A.fpres = A.fvisc = A.fsurf = A.fbound = mk3<R>(0, 0, 0);
    const R pm = P.particleMass, m2 = P.particleMass, ir = P.interactionRadius, kp = P.kpoly;
    const R kappa = P.surfaceTension;
    const R kprg = P.kpress_grad, kvg = P.kvisc_grad, kvd = P.kvisc_denum;
    const R diameter = (R)(2.0 * P.particleRadius);
    const R diameter2 = diameter * diameter;
    const R rhoSqOwn = dens * dens;
    R wAtDiameter;
    if (KSET == KS_MONAGHAN) wAtDiameter = Wmonaghan<R>(mk3<R>(diameter, 0, 0), ir);
    else wAtDiameter = Wdefault<R>(mk3<R>(diameter, 0, 0), ir, kp);
    const R epsilon = (R)0.01;
    const R beta = P.beta, rd = P.restDensity;
    auto boundaryHit = [&](const BoundaryTerms<R, KSET> &T) {
        A.fbound = A.fbound + T.bound;
        A.fpres = A.fpres + T.pres;
        A.fvisc = A.fvisc - T.visc;
    };
    auto fluidHit = [&](uint32_t j) {
        V3<R> rij, vij;
        R pOverRhoSqNb, mOverRhoNb; // pNb / (rhoNb * rhoNb), m2 / rhoNb
        if (PAIRS) { // the neighbour's two gather records (HitBuffer)
            const typename Vec4T<R>::type a = gpos[GATHER_STRIDE * j], b = gvel[GATHER_STRIDE * j];
            rij = pos1 - xyz<R>(a);
            if (STRICT && ((j == self) || !(length(rij) < ir))) return; // the loop's own tests (:494,:505)
            vij = vel1 - xyz<R>(b);
            pOverRhoSqNb = a.w; mOverRhoNb = b.w;
        } else {
            rij = pos1 - xyz<R>(sPos[j]);
            if (STRICT && ((j == self) || !(length(rij) < ir))) return;
            const R rhoNb = sDens[j];
            const R pNb = sPres[j];
            const R rhoSqNb = rhoNb * rhoNb;
            pOverRhoSqNb = pNb / rhoSqNb; mOverRhoNb = m2 / rhoNb;
            vij = vel1 - xyz<R>(sVel[j]);
        }
        V3<R> gradSpiky, gradVisc;
        R kernel;
        if (KSET == KS_MONAGHAN) {
            gradSpiky = Wmonaghan_grad<R>(rij, ir);
            gradVisc = gradSpiky;
            kernel = Wmonaghan<R>(rij, ir);
        } else {
            gradSpiky = Wpressure_grad<R>(rij, ir, kprg);
            gradVisc = Wviscosity_grad<R>(rij, ir, kvg, kvd);
            kernel = Wdefault<R>(rij, ir, kp);
        }
        A.fpres = A.fpres + (m2 * (pres / rhoSqOwn + pOverRhoSqNb) * gradSpiky);
        const R a = dot(rij, gradVisc);
        const R b = dot(rij, rij) + 0.01f * (ir * ir);
        A.fvisc = A.fvisc + (mOverRhoNb * vij * (a / b));
        if (SURF) {
            V3<R> ai = mk3<R>(0, 0, 0);
            const R r2 = dot(rij, rij);
            if (r2 > diameter2) ai = ai - (kappa / pm * pm * rij * kernel);
            else ai = ai - (kappa / pm * pm * rij * wAtDiameter);
            A.fsurf = A.fsurf + ai;
        }
    };
    if (BOUNDARY_BY_CELL) {
        // no boundary list (Sweep::scan, BFILT 2): the fluid hits of each cell from the list, then ALL boundary particles of the cell
        // from the boundary cell table, in the reference's cell order (computeForces walks z, y, x; sph_kernel_impl.cuh:642-660)
        const I3 gp = calcGridPos<R>(P, pos1);
        int k = 0;
        uint32_t head = hc.nf > 0 ? lbase[0] : 0xffffffffu;
        uint32_t c = 0;
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++, c++) {
                    while ((head >> HIT_TAG_SHIFT) == c) {
                        fluidHit(head & HIT_INDEX);
                        ++k;
                        head = k < hc.nf ? lbase[(uint32_t)k * lstride] : 0xffffffffu;
                    }
                    if (hc.anyB) {
                        const uint32_t h = calcGridHash<R>(P, gp.x + x, gp.y + y, gp.z + z);
                        const uint32_t sB = G.bCellStart[h];
                        if (sB != CELL_EMPTY) {
                            const uint32_t eB = G.bCellEnd[h];
                            for (uint32_t j = sB; j < eB; ++j) boundaryHit(boundary_terms<R, KSET>(P, pos1, vel1, dens, pres, G.sB[j]));
                        }
                    }
                }
        return A;
    }
#if NRS_PACKED_HITS
    if constexpr (!HAS_B && !STRICT && PAIRS && KSET == KS_MULLER && std::is_same<R, float>::value) {
        // interior code, fluid hits only: two consecutive hits per iteration on packed fp32 (fluid_terms2_muller); the list heads of
        // the NEXT iteration are requested before this iteration's gathers are used
        const int nf = hc.nf;
        const float own = pres / rhoSqOwn, c0 = kappa / pm * pm;
        auto walk = [&](auto inRange, const InRange &IG, bool &bad) {
            ForceAcc<R> S;
            S.fpres = S.fvisc = S.fsurf = S.fbound = mk3<R>(0, 0, 0);
            // (the lists of the packed walk live in global memory: row k of the wave-uniform root + the lane's column, so that the address is a
            // scalar base + a 32-bit lane offset and no 64-bit per-lane pointer stays live across the loop)
            auto entry = [&](int k) { return lroot[(size_t)k * lstride + lcol]; };
            uint32_t e0 = nf > 0 ? entry(0) : 0u, e1 = nf > 1 ? entry(1) : e0;
            for (int k = 0; k < nf; k += 2) {
#if defined(NRS_ABL_FORCE_COALESCED) // timing ablation: the gathers of a wave hit consecutive slots (what a staged walk could reach at best)
                const uint32_t j0 = (blockIdx.x * BLOCK + threadIdx.x + (uint32_t)k * 3u) % G.nSorted, j1 = (j0 + 1u) % G.nSorted;
#else
                const uint32_t j0 = e0 & HIT_INDEX, j1 = e1 & HIT_INDEX;
#endif
                const bool two = k + 1 < nf;
                const uint32_t n0 = k + 2 < nf ? entry(k + 2) : 0u;
                const uint32_t n1 = k + 3 < nf ? entry(k + 3) : n0;
#if defined(NRS_ABL_FORCE_NOVEL) // timing ablation: one gather per hit
                const float4 pa = gpos[GATHER_STRIDE * j0], pb = gpos[GATHER_STRIDE * j1], va = pa, vb = pb;
#else
                const float4 pa = gpos[GATHER_STRIDE * j0], pb = gpos[GATHER_STRIDE * j1], va = gvel[GATHER_STRIDE * j0], vb = gvel[GATHER_STRIDE * j1];
#endif
                const PairTerms2 T = fluid_terms2_muller<SURF, decltype(inRange)::value>(P, pos1, vel1, own, c0, wAtDiameter, diameter2, pa, pb, va, vb,
                                                                                        PrePair<float>{pa.w, va.w}, PrePair<float>{pb.w, vb.w}, IG, bad);
                S.fpres = S.fpres + mk3<R>(T.pres.x.x, T.pres.y.x, T.pres.z.x);
                S.fvisc = S.fvisc + mk3<R>(T.visc.x.x, T.visc.y.x, T.visc.z.x);
                if (SURF) S.fsurf = S.fsurf + mk3<R>(T.surf.x.x, T.surf.y.x, T.surf.z.x);
                if (two) {
                    S.fpres = S.fpres + mk3<R>(T.pres.x.y, T.pres.y.y, T.pres.z.y);
                    S.fvisc = S.fvisc + mk3<R>(T.visc.x.y, T.visc.y.y, T.visc.z.y);
                    if (SURF) S.fsurf = S.fsurf + mk3<R>(T.surf.x.y, T.surf.y.y, T.surf.z.y);
                }
                e0 = n0; e1 = n1;
            }
            return S;
        };
#if NRS_INRANGE_DIV
        const InRange IG = in_range_setup(P);
        const bool ownerOk = fminf(fminf(fabsf(pos1.x), fabsf(pos1.y)), fabsf(pos1.z)) >= 0x1p-66f;
        bool again = true;
        if (IG.constantsOk && __all(ownerOk)) { // (a vote of the lanes that walk: the branch is wave-uniform)
            again = false;
            A = walk(std::true_type{}, IG, again);
        }
        if (again) A = walk(std::false_type{}, IG, again);
#else
        bool unused = false;
        A = walk(std::false_type{}, InRange{}, unused);
#endif
        return A;
    }
#endif
    HitMerge it(lbase, lstride, hc);
    uint32_t j, key;
    bool isB;
    while (it.next(j, isB, key)) {
        if (HAS_B && isB) {
            BoundaryTerms<R, KSET> T;
            if (pre) { // evaluated by the cooperative pre-pass (same operations): pre[k][0..8]
                const R *t = pre + (size_t)it.last_boundary() * 9;
                T.bound = mk3<R>(t[0], t[1], t[2]); T.pres = mk3<R>(t[3], t[4], t[5]); T.visc = mk3<R>(t[6], t[7], t[8]);
            } else {
                T = boundary_terms<R, KSET>(P, pos1, vel1, dens, pres, G.sB[j]);
            }
            boundaryHit(T);
        } else {
            fluidHit(j);
        }
    }
    return A;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2).  Consecutive tiles of the sorted
// array share most of their neighbour rows, so give every XCD one contiguous eighth of the tiles: measured L2
// miss traffic of the gathers was 4x the compulsory bytes with the identity mapping.  (Placement is not a
// contract: this is a speed-only remap, any placement gives the same results.)
NRS_DEV uint32_t xcd_tile(uint32_t b, uint32_t nb)
{
    const uint32_t per = nb >> 3;
    return (b < (per << 3)) ? (b & 7u) * per + (b >> 3) : b;
}

// Hit lists shared between the two gathers of a step: the density kernel scans once, uses the hits, and (when
// `hb.hits` is set) leaves them in global memory for the force kernel, which then needs no scan and no LDS.
// hits[k * stride + i] is particle i's k-th list slot (k-major ⇒ coalesced), counts[i] = nf | nb << 8 | over << 16.
// gather records (SESPH): per sorted slot (x, y, z, p / rho^2) and (vx, vy, vz, m / rho) — what the force walk needs of a NEIGHBOUR, in two
// 16-byte records written by the density kernel.  The two quotients are formed ONCE per particle with the operands and the divisions
// the force loop would use for that neighbour (computeCellForces, sph_kernel_impl.cuh:531,541: two IEEE divisions less per hit there,
// bit-identical by construction), and they ride in the w lanes of copies of the sorted position and velocity, so that a hit costs TWO
// gathers (2 x global_load_dwordx4) instead of three (position, velocity, 8-byte pair).  The walk is bound by the L1 — the lanes of a wave
// sit in different rows of the neighbourhood, so every gather instruction touches ~34 distinct lines whatever it loads — and the third
// gather was a third of its line accesses: force launch 0.889 -> 0.758 ms in the bench's window (tools/ab_flowing.sh), for 0.02-0.03 ms
// more in the density launch (one coalesced read of the velocities, 32 instead of 8 bytes stored per particle).  The sorted arrays
// themselves keep their w lanes (the caller's payload).  gpos == nullptr: the force loop divides itself and gathers sPos / sVel / dens / pres.
// fast (NRS_FLAG_FAST_ARITH): per sorted slot (p * (1/rho)^2, 1/rho) with a hardware reciprocal, for the tolerance-mode force kernel
// (nrs_kernels_staged.h, k_forces_fast); the density itself stays exact.
struct FastPair { float pr, invRho; };
struct HitBuffer {
    uint32_t *hits; uint32_t *counts; uint32_t stride;
    void *gpos = nullptr, *gvel = nullptr; // the gather records (Vec4T<R>::type[capacity] each)
    const void *svel = nullptr;            // sorted velocities of this step (the density kernel copies xyz into gvel)
    FastPair *fast = nullptr;
};
// (p / rho^2, m / rho) of one particle into its gather records
template <typename R> NRS_DEV typename Vec4T<R>::type own_sorted_velocity(const HitBuffer &hb, uint32_t i)
{
    return reinterpret_cast<const typename Vec4T<R>::type *>(hb.svel)[i];
}
template <typename R>
NRS_DEV void publish_gather_records(const Params<R> &P, const HitBuffer &hb, uint32_t i, V3<R> p, typename Vec4T<R>::type v, R d, R pr)
{
    typedef typename Vec4T<R>::type T4;
    reinterpret_cast<T4 *>(hb.gpos)[GATHER_STRIDE * i] = mk4<R>(p, pr / (d * d));
    reinterpret_cast<T4 *>(hb.gvel)[GATHER_STRIDE * i] = mk4<R>(xyz<R>(v), P.particleMass / d);
}
NRS_DEV uint32_t pack_counts(HitCounts hc)
{
    return (uint32_t)hc.nf | ((uint32_t)hc.nb << 8) | (hc.over ? 1u << 16 : 0u) | (hc.anyB ? 1u << 17 : 0u);
}
constexpr uint32_t COUNTS_UNSTAGED = 1u << 18; // diagnostics: the particle's wave took the global-memory scan (nrs_kernels_staged.h)
NRS_DEV HitCounts unpack_counts(uint32_t c)
{
    HitCounts hc;
    hc.nf = (int)(c & 0xffu); hc.nb = (int)((c >> 8) & 0xffu); hc.over = ((c >> 16) & 1u) != 0u;
    hc.anyB = ((c >> 17) & 1u) != 0u;
    return hc;
}

// diagnostics over the published hit counts: out[0] = particles whose list overflowed, out[1] = sum of hits (lists that
// did not overflow), out[2] = longest list, out[3] = particles whose wave could not stage its neighbour rows in LDS
static __global__ __launch_bounds__(BLOCK) void k_hit_stats(const uint32_t *__restrict__ counts, unsigned long long *__restrict__ out, uint32_t n)
{
    unsigned long long over = 0, sum = 0, mx = 0, unst = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t c = counts[i];
        const HitCounts hc = unpack_counts(c);
        unst += (c & COUNTS_UNSTAGED) ? 1ull : 0ull;
        if (hc.over) { ++over; continue; }
        const unsigned long long k = (unsigned long long)(hc.nf + hc.nb);
        sum += k;
        mx = k > mx ? k : mx;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        over += __shfl_down(over, d);
        sum += __shfl_down(sum, d);
        unst += __shfl_down(unst, d);
        const unsigned long long o = __shfl_down(mx, d);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&out[0], over);
        atomicAdd(&out[1], sum);
        atomicMax(&out[2], mx);
        atomicAdd(&out[3], unst);
    }
}

// ---- density + Tait pressure (computeDensityPressure, sph_kernel_impl.cuh:365-433) -----------------------
// SHARE: build the lists with the force loop's (wider) boundary cut-off and publish them for the force kernel.
// WIDE (IISPH, Muller kernels): the published lists keep self and every candidate up to r2LeH2, for the six other
// kernels of the IISPH chain; the density itself applies its own tests while summing.
// Wall particles.  A fluid particle whose 27 cells hold any boundary particle pays ~50-100 extra candidates and up to HIT_CAP
// boundary hits — and in the dam-break such particles are the first one or two of every x-row: half of all wavefronts carry
// one or two of them and run the whole boundary machinery at 2/64 lane utilisation (measured: 37 % of the density kernel and
// 27 % of the force kernel).  A static bit per cell ("boundary particles in the 27-neighbourhood", built once per
// nrs_set_boundaries) tells which sorted slots these are, and each gather launch has two kinds of workgroups running the SAME
// per-particle code:
//   wall workgroups      (the first blocks of the grid) walk this step's WALL LIST — the wall slots in ascending order, built
//                        by the reorder kernel (per-tile counts), a prefix sum and k_wall_compact (ballots: no atomics) —
//                        with every lane busy, code compiled with the boundary machinery;
//   interior workgroups  one thread per sorted slot as before, compiled WITHOUT the boundary machinery (no boundary cell-table
//                        loads, no second list); a wall slot is skipped.
// Each particle is still evaluated by exactly one thread with exactly the same operations: results are bit-identical.
// (Tried first: the list appended by the reorder kernel through one atomic counter — +0.31 ms of atomic contention at 10 M
// particles; the wall work as a launch of its own behind the interior one — a single generation of latency-bound waves that cost
// more than it saved — or beside it on a second stream — 0.89 vs 0.65 ms, the two launches did not overlap usefully; wall
// workgroups that each compact a fixed range of slots themselves — the floor rows make a few ranges all-wall: 0.78 ms.  The
// one-launch form kept here allocates registers for both code paths (84 VGPRs = 5 waves/SIMD instead of 7; bounded to 72 it
// spills: 0.68 vs 0.65 ms), which is why it recovers only part of the 37 %.)
constexpr uint32_t COUNTS_DEFERRED = 1u << 19; // counts[]: the particle is handled by the wall workgroups

NRS_DEV bool cell_near_boundary(const WallList &wl, uint32_t h) { return ((wl.nearBits[h >> 5] >> (h & 31u)) & 1u) != 0u; }

// wall list: step 1 (wall slots per 256-slot tile) rides in the reorder kernels (wall_tile_count, nrs_kernels_ref.h),
// step 3 (step 2 is k_resort_scan_tiles over the tile counts): stable compaction of the wall slots
static __global__ __launch_bounds__(BLOCK) void k_wall_compact(WallList wl, const uint32_t *__restrict__ tileOffset, const uint32_t *__restrict__ groupPrefix,
                                                                 uint32_t groupSize, uint32_t *__restrict__ list, uint32_t n)
{
    __shared__ uint32_t waveCnt[BLOCK / 64];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long m = wl.mask[i >> 6]; // (written by the reorder kernel; bits of slots >= n are 0)
    const bool take = ((m >> lane) & 1ull) != 0ull;
    if (lane == 0) waveCnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (!take) return;
    uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; ++w) before += waveCnt[w];
    list[groupPrefix[blockIdx.x / groupSize] + tileOffset[blockIdx.x] + before] = i;
}

template <typename R, int KSET, bool HAS_B, bool SHARE, bool WIDE>
NRS_DEV void density_tiled_particle(const Params<R> &P, const GridView<R> &G, const CutThresholds thr,
                                    const typename Vec4T<R>::type *__restrict__ sPos, R *__restrict__ dens, R *__restrict__ pres,
                                    const HitBuffer &hb, uint32_t i, V3<R> p, uint32_t (*lst)[BLOCK], uint32_t countFlags = 0u)
{
    const uint32_t tid = threadIdx.x;
    constexpr int BF = SHARE ? (KSET == KS_MULLER ? 1 : 2) : 0;
    constexpr bool COMPACT = NRS_COMPACT_SCAN && !HAS_B && SHARE; // (the context publishes lists only when its qpos array is valid)
    HitCounts hc;
    R d;
    if constexpr (COMPACT) {
#if defined(NRS_ABL_NOSCAN) // timing ablations (tools/density_ablate2.py): results are wrong by design
        hc.nf = 0; hc.nb = 0; hc.over = false; hc.anyB = false;
#else
        hc = Sweep<R>::template scan_compact<BLOCK>(P, G, p, lst);
#endif
#if defined(NRS_ABL_NOPROCESS) || defined(NRS_ABL_NOAPPEND) // (NOAPPEND: the scan left its entries unwritten — nobody may read them)
        d = (R)hc.nf;
        hc.nf = 0; hc.over = false; // (publish empty lists: the entries carry row tags only)
#else
        if (hc.over) d = density_of<R, KSET, HAS_B>(P, G, sPos, i); // list overflow / far owner: reference-order path
        else {
            constexpr bool CAN = KSET == KS_MULLER && std::is_same<R, float>::value;
            const float tKeep = WIDE ? thr.r2LeH2 : thr.lenLtIr;
            if (CAN && density_inrange_ok<R>(P, p)) d = density_from_superset<R, KSET, WIDE, CAN>(P, sPos, p, lst, hc.nf, tKeep, i); // (a vote: wave-uniform)
            else d = density_from_superset<R, KSET, WIDE>(P, sPos, p, lst, hc.nf, tKeep, i);
            if (hc.nf > HIT_CAP) { hc.over = true; d = density_of<R, KSET, HAS_B>(P, G, sPos, i); } // (only when the list held HIT_CAP + 1 exact hits)
        }
#endif
    } else {
        hc = Sweep<R>::template scan<HAS_B, BF, BLOCK, WIDE>(P, G, thr, sPos, WIDE ? 0xffffffffu : i, p, lst);
        if (hc.over) d = density_of<R, KSET, HAS_B>(P, G, sPos, i); // list overflow: reference-order path
        else d = density_from_hits<R, KSET, HAS_B, WIDE, (HAS_B && BF == 2)>(P, G, sPos, p, &lst[0][tid], BLOCK, hc, i);
    }
    dens[i] = d;
    if (pres) {
        const R pr = tait_pressure<R>(P, d);
        pres[i] = pr;
        // (the own velocity is requested HERE: asked for before the walk it costs 8 spilled registers at the 80-VGPR bound, 0.608 against 0.561 ms at rest)
        if (SHARE && hb.gpos) publish_gather_records<R>(P, hb, i, p, own_sorted_velocity<R>(hb, i), d, pr);
        if (SHARE && hb.fast) {
            const float inv = __builtin_amdgcn_rcpf((float)d);
            FastPair z;
            z.pr = (float)pr * inv * inv; z.invRho = inv;
            hb.fast[i] = z;
        }
    }
    if (SHARE) {
        hb.counts[i] = pack_counts(hc) | countFlags;
        if (!hc.over) {
            for (int k = 0; k < hc.nf; ++k) hb.hits[(size_t)k * hb.stride + i] = lst[k][tid];
            for (int k = 0; k < hc.nb; ++k) hb.hits[(size_t)(HIT_CAP - 1 - k) * hb.stride + i] = lst[HIT_CAP - 1 - k][tid];
        }
    }
}

// DEFER (only with boundaries and shared lists): the two kinds of workgroups described above; HAS_B is then false for the
// interior code, and the first `wallBlocks` blocks of the grid walk the wall list with the boundary code.
#ifndef DENSITY_DEFER_MIN_WAVES
#define DENSITY_DEFER_MIN_WAVES 6 // waves/SIMD the two-kinds-of-workgroups kernel is register-bounded for: unbounded 84 VGPRs (5 waves) 0.677 ms, 6 (80 VGPRs, 3 dwords spilled) 0.650, 7 (72, 11 spilled) 0.651
#endif
template <typename R, int KSET, bool HAS_B, bool SHARE, bool WIDE = false, bool DEFER = false>
__global__ __launch_bounds__(BLOCK, (((DEFER || (NRS_COMPACT_SCAN && !HAS_B && SHARE)) && sizeof(R) == 4) ? DENSITY_DEFER_MIN_WAVES : 1)) void k_density_tiled(Params<R> P, GridView<R> G, CutThresholds thr,
                                                         const typename Vec4T<R>::type *__restrict__ sPos,
                                                         R *__restrict__ dens, R *__restrict__ pres, HitBuffer hb,
                                                         uint32_t n, WallList wl, uint32_t wallBlocks)
{
    // (rows SCAN_CAP .. SCAN_CAP + 2: spill rows of the branch-free append, Sweep::scan_compact; kernels without the quantised scan
    // keep the 20 KiB that let 8 workgroups share a CU)
    constexpr int LIST_ROWS = (NRS_COMPACT_SCAN && SHARE && (DEFER || !HAS_B)) ? SCAN_CAP + 3 : HIT_CAP;
    __shared__ uint32_t lst[LIST_ROWS][BLOCK];
    uint32_t block = blockIdx.x, blocks = gridDim.x;
    if (DEFER) {
        if (block < wallBlocks) {
            const uint32_t count = *wl.count;
            for (uint32_t t = block * BLOCK + threadIdx.x; t < count; t += wallBlocks * BLOCK) {
                const uint32_t i = wl.list[t];
                const V3<R> p = xyz<R>(sPos[i]);
#if defined(NRS_ABL_NOWALL) // timing ablation: wall particles get no work at all
                if (true) {
#else
                if (!slab_active<R>(P, G, p.x)) {
#endif
                    dens[i] = (R)0;
                    if (pres) pres[i] = (R)0;
                    hb.counts[i] = COUNTS_DEFERRED;
                    continue;
                }
                density_tiled_particle<R, KSET, true, SHARE, WIDE>(P, G, thr, sPos, dens, pres, hb, i, p, lst, COUNTS_DEFERRED);
            }
            return;
        }
        block -= wallBlocks; blocks -= wallBlocks;
    }
    const uint32_t i = xcd_tile(block, blocks) * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (DEFER && ((wl.mask[i >> 6] >> (i & 63u)) & 1ull)) return; // a wall slot
    const V3<R> p = xyz<R>(sPos[i]);
    if (!slab_active<R>(P, G, p.x)) {
        dens[i] = (R)0;
        if (pres) pres[i] = (R)0;
        if (SHARE) hb.counts[i] = 0u;
        return;
    }
    density_tiled_particle<R, KSET, HAS_B, SHARE, WIDE>(P, G, thr, sPos, dens, pres, hb, i, p, lst);
}

// ---- forces (computeForces, sph_kernel_impl.cuh:609-680).  FUSE: the same launch also integrates
//      (integrate_functor :71-100) and hashes the new position (calcHashD :127-145), writing the next step's input
//      arrays and radix-sort keys directly: saves the force-array round trip and two launches per step ---------
template <typename R> struct FusedOut {
    typedef typename Vec4T<R>::type T4;
    T4 *newPos, *newVel;    // next step's "unsorted" arrays
    uint32_t *hash, *index; // next step's keys / values
    // coherent re-sort (nrs_kernels_resort.h): this step's sorted keys and the per-tile count of slots whose key
    // changes; both null when the next step sorts from scratch
    const uint32_t *prevHash;
    uint32_t *tileMovers;
    // slab runs (nrs_kernels_slab.h): classify the particle for the next partition here, where its new position is in
    // registers — stream flags per slot, per-2048-slot populations of the message/ghost streams, dead slots get the key
    // 0xffffffff and are counted per 256-slot tile.  slabFlags == nullptr: off.
    uint8_t *slabFlags;
    uint32_t *slabBlockCounts;
    uint32_t slabBlocks;
    uint32_t *tileDead;
    SlabCfg slab;
};

template <typename R, int KSET, bool SURF, bool HAS_B, bool FUSE>
NRS_DEV void forces_epilogue(const Params<R> &P, typename Vec4T<R>::type p4, typename Vec4T<R>::type v4, V3<R> f,
                             typename Vec4T<R>::type *__restrict__ forces, const FusedOut<R> &fo, uint32_t i)
{
    if (forces) forces[i] = mk4<R>(f, (R)0);
    if (FUSE) {
        const R dt = P.timestep, m1 = P.particleMass;
        const V3<R> pos1 = xyz<R>(p4), vel1 = xyz<R>(v4);
        const V3<R> accel = dt * f / m1;
        const V3<R> v = vel1 + accel;
        const V3<R> pn = pos1 + dt * v;
        fo.newPos[i] = mk4<R>(pn, p4.w);
        fo.newVel[i] = mk4<R>(v, v4.w);
        const I3 g = calcGridPos<R>(P, pn);
        const uint32_t h = calcGridHash<R>(P, g.x, g.y, g.z);
        uint32_t key = h;
        if (fo.slabFlags) {
            const uint32_t f = slab_flags<R>(P, fo.slab, mk4<R>(pn, p4.w));
            fo.slabFlags[i] = (uint8_t)f;
            if (!(f & (1u << ST_STAY))) {
                key = 0xffffffffu;
                atomicAdd(&fo.tileDead[i / BLOCK], 1u);
            } else if (h != fo.prevHash[i]) {
                atomicAdd(&fo.tileMovers[i / BLOCK], 1u);
            }
            if (f & ~(1u << ST_STAY))
                for (int s = 1; s < ST_COUNT; ++s)
                    if ((f >> s) & 1u) atomicAdd(&fo.slabBlockCounts[(uint32_t)s * fo.slabBlocks + i / SLAB_TILE], 1u);
        } else if (fo.tileMovers && h != fo.prevHash[i]) {
            atomicAdd(&fo.tileMovers[i / BLOCK], 1u);
        }
        fo.hash[i] = key;
        fo.index[i] = i;
    }
}

// own scan (used when no shared lists exist: partial steps after nrs_step_partial(DENSITY) etc.)
template <typename R, int KSET, bool SURF, bool HAS_B, bool FUSE>
__global__ __launch_bounds__(BLOCK) void k_forces_tiled(Params<R> P, GridView<R> G, CutThresholds thr,
                                                        const typename Vec4T<R>::type *__restrict__ sPos,
                                                        const typename Vec4T<R>::type *__restrict__ sVel,
                                                        const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                        typename Vec4T<R>::type *__restrict__ forces, FusedOut<R> fo,
                                                        uint32_t n)
{
    typedef typename Vec4T<R>::type T4;
    __shared__ uint32_t lst[HIT_CAP][BLOCK];
    const uint32_t i = xcd_tile(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t tid = threadIdx.x;
    const T4 p4 = sPos[i];
    const T4 v4 = sVel[i];
    const V3<R> pos1 = xyz<R>(p4), vel1 = xyz<R>(v4);
    V3<R> f = mk3<R>(0, 0, 0);
    if (slab_active<R>(P, G, pos1.x)) {
        const R dens = sDens[i], pres = sPres[i];
        constexpr int BF = (KSET == KS_MULLER ? 1 : 2);
        const HitCounts hc = Sweep<R>::template scan<HAS_B, BF, BLOCK>(P, G, thr, sPos, i, pos1, lst);
        ForceAcc<R> A;
        if (hc.over) A = gather_forces<R, KSET, SURF, HAS_B>(P, G, i, pos1, vel1, dens, pres, sPos, sVel, sDens, sPres);
        else A = forces_from_hits<R, KSET, SURF, HAS_B, false, false, (HAS_B && BF == 2)>(P, G, sPos, sVel, sDens, sPres, pos1, vel1, dens, pres, &lst[0][tid], BLOCK, hc);
        f = sesph_total_force<R>(P, A, dens);
    }
    forces_epilogue<R, KSET, SURF, HAS_B, FUSE>(P, p4, v4, f, forces, fo, i);
}

// scan-free form: consumes the hit lists the density kernel of the same step published (no LDS ⇒ occupancy is
// bounded by registers only)
template <typename R, int KSET, bool SURF, bool HAS_B, bool FUSE>
NRS_DEV void forces_lists_particle(const Params<R> &P, const GridView<R> &G, const HitBuffer &hb,
                                   const typename Vec4T<R>::type *__restrict__ sPos, const typename Vec4T<R>::type *__restrict__ sVel,
                                   const R *__restrict__ sDens, const R *__restrict__ sPres, typename Vec4T<R>::type *__restrict__ forces,
                                   const FusedOut<R> &fo, uint32_t i, typename Vec4T<R>::type p4, typename Vec4T<R>::type v4,
                                   bool active, HitCounts hc)
{
    const V3<R> pos1 = xyz<R>(p4), vel1 = xyz<R>(v4);
    V3<R> f = mk3<R>(0, 0, 0);
    if (active) {
        const R dens = sDens[i], pres = sPres[i];
        ForceAcc<R> A;
        if (hc.over) A = gather_forces<R, KSET, SURF, HAS_B>(P, G, i, pos1, vel1, dens, pres, sPos, sVel, sDens, sPres);
        else // (the context hands out lists only together with the pairs array of the same density launch)
            A = forces_from_hits<R, KSET, SURF, HAS_B, false, NRS_FORCE_PAIRS != 0, (HAS_B && KSET == KS_MONAGHAN)>(P, G, sPos, sVel, sDens, sPres, pos1, vel1, dens, pres, hb.hits + i, hb.stride,
                                                                                     hc, 0xffffffffu, nullptr, reinterpret_cast<const typename Vec4T<R>::type *>(hb.gpos),
                                                                                     reinterpret_cast<const typename Vec4T<R>::type *>(hb.gvel), hb.hits, i);
        f = sesph_total_force<R>(P, A, dens);
    }
    forces_epilogue<R, KSET, SURF, HAS_B, FUSE>(P, p4, v4, f, forces, fo, i);
}

// DEFER: wall workgroups first, interior workgroups skip the particles flagged COUNTS_DEFERRED (see k_density_tiled); HAS_B is
// false for the interior code
// (the packed two-hit walk of the interior code needs 94 VGPRs: 5 waves per SIMD; the scalar walk is bounded for 7)
template <typename R, int KSET, bool SURF, bool HAS_B, bool FUSE, bool DEFER = false>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? ((NRS_PACKED_HITS && NRS_FORCE_PAIRS && KSET == KS_MULLER && !HAS_B) ? (FUSE ? FORCES_PACKED_MIN_WAVES : FORCES_PACKED_MIN_WAVES_UNFUSED) : FORCES_LISTS_MIN_WAVES) : 1)) void k_forces_lists(Params<R> P, GridView<R> G, HitBuffer hb,
                                                        const typename Vec4T<R>::type *__restrict__ sPos,
                                                        const typename Vec4T<R>::type *__restrict__ sVel,
                                                        const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                        typename Vec4T<R>::type *__restrict__ forces, FusedOut<R> fo,
                                                        uint32_t n, WallList wl, uint32_t wallBlocks)
{
    typedef typename Vec4T<R>::type T4;
    uint32_t block = blockIdx.x, blocks = gridDim.x;
    if (DEFER) {
        if (block < wallBlocks) {
#if defined(NRS_ABL_NOWALL_F) // timing ablation: the wall particles get no force evaluation (and are not integrated)
            return;
#endif
            const uint32_t count = *wl.count;
            for (uint32_t t = block * BLOCK + threadIdx.x; t < count; t += wallBlocks * BLOCK) {
                const uint32_t i = wl.list[t];
                const T4 p4 = sPos[i];
                const bool active = slab_active<R>(P, G, p4.x);
                forces_lists_particle<R, KSET, SURF, true, FUSE>(P, G, hb, sPos, sVel, sDens, sPres, forces, fo, i, p4, sVel[i], active,
                                                                 unpack_counts(active ? hb.counts[i] : 0u));
            }
            return;
        }
        block -= wallBlocks; blocks -= wallBlocks;
    }
    const uint32_t i = xcd_tile(block, blocks) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = hb.counts[i];
    if (DEFER && (c & COUNTS_DEFERRED)) return;
    const T4 p4 = sPos[i];
    const bool active = slab_active<R>(P, G, p4.x);
    forces_lists_particle<R, KSET, SURF, HAS_B, FUSE>(P, G, hb, sPos, sVel, sDens, sPres, forces, fo, i, p4, sVel[i], active,
                                                      unpack_counts(active ? c : 0u));
}

// marks, for every cell that holds boundary particles, the 27 cells around it (power-of-two grids: the reference's wrap)
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_mark_near_boundary(Params<R> P, const uint32_t *__restrict__ bHash, uint32_t nb, uint32_t *__restrict__ nearBits)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nb) return;
    const uint32_t h = bHash[i];
    if (i && bHash[i - 1] == h) return; // one thread per occupied cell
    const uint32_t gx = P.gridSize[0], gy = P.gridSize[1];
    const int x = (int)((h & (gx - 1)) + P.numBodies), y = (int)((h / gx) & (gy - 1)), z = (int)(h / (gx * gy)); // (x: global column)
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const uint32_t c = calcGridHash<R>(P, x + dx, y + dy, z + dz);
                atomicOr(&nearBits[c >> 5], 1u << (c & 31u));
            }
}

static inline bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

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

// wall workgroups of a gather launch (grid-stride over the wall list, whose length is only known on the device)
static inline uint32_t wall_blocks(uint32_t interiorBlocks) { return std::min<uint32_t>(1024u, std::max<uint32_t>(1u, interiorBlocks / 16u)); }

// wall: this step's wall list (tile counts of the reorder kernel / scan / k_wall_compact) — needs `share`; null = one kind of workgroup
template <typename R, int KSET, bool HAS_B>
static inline void launch_density_tiled(hipStream_t stream, const Params<R> &P, const GridView<R> &G, const HitBuffer *share,
                                        const typename Vec4T<R>::type *sPos, R *dens, R *pres, uint32_t n, const WallList *wall = nullptr)
{
    const CutThresholds thr = make_thresholds<R>(P);
    const dim3 g((n + BLOCK - 1) / BLOCK), b(BLOCK);
    HitBuffer hb = {nullptr, nullptr, 0};
    const WallList none = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // occupancy experiment (tools/occupancy_sweep.sh builds variants with -DNRS_DBG_LDS_PAD=bytes): extra dynamic LDS per
    // workgroup lowers the workgroups per CU
    constexpr unsigned pad = NRS_DBG_LDS_PAD;
    if (share) {
        hb = *share;
        if (HAS_B && wall) {
            const uint32_t wb = wall_blocks(g.x);
            hipLaunchKernelGGL((k_density_tiled<R, KSET, false, true, false, true>), dim3(g.x + wb), b, pad, stream, P, G, thr, sPos, dens, pres, hb, n,
                               *wall, wb);
        } else {
            hipLaunchKernelGGL((k_density_tiled<R, KSET, HAS_B, true>), g, b, pad, stream, P, G, thr, sPos, dens, pres, hb, n, none, 0u);
        }
    } else {
        hipLaunchKernelGGL((k_density_tiled<R, KSET, HAS_B, false>), g, b, 0, stream, P, G, thr, sPos, dens, pres, hb, n, none, 0u);
    }
}
template <typename R, int KSET, bool HAS_B>
static inline void launch_density_wide(hipStream_t stream, const Params<R> &P, const GridView<R> &G, const HitBuffer &hb,
                                       const typename Vec4T<R>::type *sPos, R *dens, uint32_t n, const WallList *wall = nullptr)
{
    const CutThresholds thr = make_thresholds<R>(P);
    const WallList none = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const uint32_t g = (n + BLOCK - 1) / BLOCK;
    if (HAS_B && wall) { // wall workgroups (boundary code, exact positions) + interior workgroups (quantised scan), see k_density_tiled
        const uint32_t wb = wall_blocks(g);
        hipLaunchKernelGGL((k_density_tiled<R, KSET, false, true, true, true>), dim3(g + wb), dim3(BLOCK), 0, stream, P, G, thr, sPos, dens,
                           (R *)nullptr, hb, n, *wall, wb);
        return;
    }
    hipLaunchKernelGGL((k_density_tiled<R, KSET, HAS_B, true, true>), dim3(g), dim3(BLOCK), 0, stream, P, G, thr,
                       sPos, dens, (R *)nullptr, hb, n, none, 0u);
}
// `lists`: hit lists published by launch_density_tiled of the same step (then no scan), or nullptr
template <typename R, int KSET, bool SURF, bool HAS_B>
static inline void launch_forces_tiled(hipStream_t stream, const Params<R> &P, const GridView<R> &G, const HitBuffer *lists,
                                       const typename Vec4T<R>::type *sPos, const typename Vec4T<R>::type *sVel, const R *dens,
                                       const R *pres, typename Vec4T<R>::type *forces, const FusedOut<R> *fused, uint32_t n,
                                       const WallList *wall = nullptr)
{
    FusedOut<R> fo;
    fo.newPos = fo.newVel = nullptr;
    fo.hash = fo.index = nullptr;
    fo.prevHash = nullptr;
    fo.tileMovers = nullptr;
    fo.slabFlags = nullptr;
    fo.slabBlockCounts = nullptr;
    fo.slabBlocks = 0;
    fo.tileDead = nullptr;
    fo.slab = SlabCfg{0, 0, 0};
    if (fused) fo = *fused;
    const dim3 g((n + BLOCK - 1) / BLOCK), b(BLOCK);
    constexpr unsigned padF = NRS_DBG_LDS_PAD_F; // occupancy experiment
    const WallList none = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (lists && HAS_B && wall) { // wall workgroups + interior workgroups without the boundary code (see k_density_tiled)
        const uint32_t wb = wall_blocks(g.x);
        const dim3 gd(g.x + wb);
        if (fused) hipLaunchKernelGGL((k_forces_lists<R, KSET, SURF, false, true, true>), gd, b, padF, stream, P, G, *lists, sPos, sVel, dens, pres, forces, fo, n, *wall, wb);
        else hipLaunchKernelGGL((k_forces_lists<R, KSET, SURF, false, false, true>), gd, b, 0, stream, P, G, *lists, sPos, sVel, dens, pres, forces, fo, n, *wall, wb);
    } else if (lists) {
        if (fused) hipLaunchKernelGGL((k_forces_lists<R, KSET, SURF, HAS_B, true>), g, b, padF, stream, P, G, *lists, sPos, sVel, dens, pres, forces, fo, n, none, 0u);
        else hipLaunchKernelGGL((k_forces_lists<R, KSET, SURF, HAS_B, false>), g, b, 0, stream, P, G, *lists, sPos, sVel, dens, pres, forces, fo, n, none, 0u);
    } else {
        const CutThresholds thr = make_thresholds<R>(P);
        if (fused) hipLaunchKernelGGL((k_forces_tiled<R, KSET, SURF, HAS_B, true>), g, b, 0, stream, P, G, thr, sPos, sVel, dens, pres, forces, fo, n);
        else hipLaunchKernelGGL((k_forces_tiled<R, KSET, SURF, HAS_B, false>), g, b, 0, stream, P, G, thr, sPos, sVel, dens, pres, forces, fo, n);
    }
}

} // namespace nrs
