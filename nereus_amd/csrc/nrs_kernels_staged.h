// nrs_kernels_staged.h — density/pressure gather of the SESPH step with the neighbour rows STAGED IN LDS (fp32).
//
// What the reference does (computeDensityPressure, sph_kernel_impl.cuh:365-433): one thread per particle walks the 27
// neighbour cells through cellStart/cellEnd and tests every candidate.  What nrs_kernels_tiled.h does: one thread per
// sorted slot, the three x-cells of a (dz,dy) row swept as ONE run of the sorted array, candidates fetched from global
// memory four at a time.  That scan was measured latency-bound on the L1/L2 path (54 % of a wave's life waiting, 11 of
// 64 lanes busy per vector instruction): 64 consecutive slots request the same ~68 candidates of a row 64 times over.
//
// Here one WAVEFRONT owns 64 consecutive sorted slots (workgroup = 1 wave: no workgroup barrier couples waves).
//   1. every lane reads its 27 cell-table entries (one round trip) and forms its nine row runs [lo, hi);
//   2. the slots are sorted by hash, so the runs of a row are monotone across the lanes: the row's hull is
//      [lo of the first non-empty lane, hi of the last] — two ballots and two v_readlane per row, no reduction tree;
//   3. the nine hulls (~9 x 68 positions) are staged into LDS with coalesced 16-byte loads (one round trip, every load in
//      flight at once: the occupancy is LDS-bound at ~3 waves/SIMD, which leaves > 128 VGPRs per lane);
//   4. each lane scans its runs out of LDS (ds_read_b128, ~64-cycle latency instead of an L2 round trip).
// The hit lists (LDS, then published for the force kernel of the step) and everything after the scan are those of
// nrs_kernels_tiled.h.  A wave whose hulls do not fit the pool, or that holds a particle in a grid-edge cell (the
// reference's power-of-two wrap makes its rows non-contiguous), takes the global-memory scan of nrs_kernels_tiled.h.
//
// FAST (NRS_FLAG_FAST_ARITH, Muller kernels): the density is accumulated in the scan itself as
// m*kpoly*sum (h^2 - r^2)^3 — no square root, no double-precision cube, no second pass over the hits — and the Tait
// pressure is a float x^7.  Results agree with the reference-order arithmetic to ~1e-6 relative (the reference itself
// is built with --use_fast_math, CMakeLists.txt:85); hash / index / cell tables do not depend on this flag.
#pragma once
#include "nrs_kernels_tiled.h"
#include <type_traits>

namespace nrs {

constexpr int STG_WAVE = 64;  // slots per workgroup = one wavefront
#ifndef STG_POOL
#define STG_POOL 320          // float4 slots staged per wave and z-plane: three hulls of ~68 at rest (5 KiB; also holds the
                              // 32-bit lists of the unstaged path)
#endif
constexpr int STG_ROWS = 9;

#ifndef FORCE_BATCH
#define FORCE_BATCH 4 // pairs whose gathers are in flight together in the FAST force walk
#endif
// (FastPair — p / rho^2, 1 / rho per particle for the FAST force kernel — is declared in nrs_kernels_tiled.h)

// 16-byte LDS read that stays ONE ds_read_b128 (4 LDS cycles): left alone, the compiler narrows a float4 load whose w is
// unused to ds_read_b96, which takes 8 (MI355X_MICROARCH.md, LDS table)
NRS_DEV float4 lds_read16(const float4 *q)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const volatile v4f *lds_ptr; // (a volatile GENERIC load would become flat_load)
    const v4f v = *(lds_ptr)(q);
    return make_float4(v.x, v.y, v.z, v.w);
}

NRS_DEV float pow7_fast(float x)
{
    const float x2 = x * x, x4 = x2 * x2;
    return x4 * x2 * x;
}

// three consecutive table entries (the x-1, x, x+1 cells of a row) in ONE 12-byte load through a 32-bit byte offset from the
// wave-uniform base (tables of at most 2^30 cells)
struct Cell3 { uint32_t a, b, c; };
NRS_DEV Cell3 load_cell3(const uint32_t *__restrict__ table, uint32_t firstCell)
{
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    u3 v;
    __builtin_memcpy(&v, reinterpret_cast<const char *>(table) + (firstCell << 2), 12);
    Cell3 r; r.a = v.x; r.b = v.y; r.c = v.z;
    return r;
}

#ifndef STG_MIN_WAVES
#define STG_MIN_WAVES 1 // waves per SIMD the register allocation is bounded for (experiments)
#endif
#ifndef STG_COOP_LANES
#define STG_COOP_LANES 4 // up to this many lanes with boundary cells: the wave sweeps them cooperatively, one such lane at a time
#endif
#ifndef STG_BATCH
#define STG_BATCH SCAN_BATCH // candidates read from LDS per lane per round
#endif
template <int KSET, bool HAS_B, bool SHARE, bool FAST>
__global__ __launch_bounds__(STG_WAVE, STG_MIN_WAVES) void k_density_staged(Params<float> P, GridView<float> G, CutThresholds thr,
                                                             const float4 *__restrict__ sPos, float *__restrict__ dens,
                                                             float *__restrict__ pres, FastPair *__restrict__ fq, HitBuffer hb, uint32_t n)
{
    typedef float R;
    // FAST keeps 16-bit hits in LDS (row << 12 | offset into the row's hull; boundary: cell << 11 | offset into the cell) and
    // widens them when it publishes; the exact path keeps the 32-bit (tag, index) entries density_from_hits walks
    typedef typename std::conditional<FAST, uint16_t, uint32_t>::type Hit;
    __shared__ float4 pool[STG_POOL];              // one z-plane at a time: the hulls of its three rows
    __shared__ Hit lst[HIT_CAP + 1][STG_WAVE];     // row HIT_CAP: where the non-hits are stored (branch-free append)
    __shared__ uint32_t hullStart[STG_ROWS];
    static_assert(STG_POOL * 16 >= HIT_CAP * STG_WAVE * 4, "the pool doubles as the 32-bit hit list of the unstaged path");
    const uint32_t lane = threadIdx.x;
    const uint32_t i = xcd_tile(blockIdx.x, gridDim.x) * STG_WAVE + lane;
    const bool inb = i < n;
    const float4 p4 = inb ? sPos[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const V3<R> p = xyz<R>(p4);
    const bool active = inb && slab_active<R>(P, G, p.x);

    const I3 gp = calcGridPos<R>(P, p);
    const uint32_t mx = P.gridSize[0] - 1, my = P.gridSize[1] - 1, mz = P.gridSize[2] - 1;
    const uint32_t cx = grid_x<R>(P, gp.x), cy = (uint32_t)gp.y & my, cz = (uint32_t)gp.z & mz;
    const bool edge = active && (cx == 0u || cx == mx || cy == 0u || cy == my || cz == 0u || cz == mz);
    constexpr int BF = SHARE ? (KSET == KS_MULLER ? 1 : 2) : 0;

    // ---- 1. cell-table entries of the nine rows: 18 (27 with boundaries) 12-byte loads, all in flight together -----------
    uint32_t lo[STG_ROWS], hi[STG_ROWS], m1[STG_ROWS], m2[STG_ROWS];
    uint32_t bmask = 0; // bit (row * 3 + c): boundary particles in that cell
    bool staged = __ballot(edge) == 0ull;
    if (staged) {
        Cell3 st[STG_ROWS], en[STG_ROWS], bs[STG_ROWS];
#pragma unroll
        for (int r = 0; r < STG_ROWS; ++r) {
            const uint32_t zc = cz + (uint32_t)(r / 3) - 1u, yc = cy + (uint32_t)(r % 3) - 1u; // (no wrap: not an edge cell)
            const uint32_t h0 = active ? umul24(umul24(zc, P.gridSize[1]), P.gridSize[0]) + umul24(yc, P.gridSize[0]) + cx - 1u : 0u;
            st[r] = load_cell3(G.cellStart, h0);
            en[r] = load_cell3(G.cellEnd, h0);
            if (HAS_B) bs[r] = load_cell3(G.bCellStart, h0);
        }
#pragma unroll
        for (int r = 0; r < STG_ROWS; ++r) {
            const uint32_t s0 = st[r].a, s1 = st[r].b, s2 = st[r].c;
            uint32_t a = (s0 != CELL_EMPTY) ? s0 : ((s1 != CELL_EMPTY) ? s1 : s2);
            uint32_t b = (s2 != CELL_EMPTY) ? en[r].c : ((s1 != CELL_EMPTY) ? en[r].b : en[r].a);
            if (a == CELL_EMPTY || !active || !run_ok<R>(G, a, b)) a = b = 0u;
            lo[r] = a; hi[r] = b;
            // cell number inside the run: + (j >= m1) + (j >= m2), with m1 = start of the 2nd cell or, when that one is
            // empty, of the 3rd (CELL_EMPTY = 0xffffffff compares greater than every j)
            m2[r] = s2;
            m1[r] = s1 < s2 ? s1 : s2;
            if (HAS_B && active)
                bmask |= ((bs[r].a != CELL_EMPTY) ? (1u << (r * 3)) : 0u) | ((bs[r].b != CELL_EMPTY) ? (2u << (r * 3)) : 0u) |
                         ((bs[r].c != CELL_EMPTY) ? (4u << (r * 3)) : 0u);
        }
    }

    // ---- 2. hull of every row over the wave (the runs are monotone across the lanes: the slots are sorted by hash) --------
    uint32_t L[STG_ROWS], len[STG_ROWS];
    if (staged) {
#pragma unroll
        for (int r = 0; r < STG_ROWS; ++r) {
            const unsigned long long m = __ballot(lo[r] != hi[r]);
            uint32_t a = 0, b = 0;
            if (m) {
                a = (uint32_t)__builtin_amdgcn_readlane((int)lo[r], __builtin_ctzll(m));
                b = (uint32_t)__builtin_amdgcn_readlane((int)hi[r], 63 - __builtin_clzll(m));
            }
            L[r] = a;
            len[r] = b - a;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) staged = staged && (len[3 * pl] + len[3 * pl + 1] + len[3 * pl + 2] <= (uint32_t)STG_POOL);
        if (FAST) {
#pragma unroll
            for (int r = 0; r < STG_ROWS; ++r) staged = staged && len[r] < 4096u; // 12-bit offsets in the 16-bit hits
        }
    }

    const float tF = thr.lenLtIr;
    const float tB = BF == 2 ? INFINITY : (BF == 1 ? thr.r2LeH2 : thr.lenLtIr);
    const float h2 = P.interactionRadius * P.interactionRadius;
    HitCounts hc;
    hc.nf = 0; hc.nb = 0; hc.over = false; hc.anyB = false;
    float acc = 0.f, bacc = 0.f; // FAST: sum (h^2-r^2)^3 over fluid (self included) / sum Vb (h^2-r^2)^3 over boundary
    if (staged) {
        if (FAST && SHARE && lane < (uint32_t)STG_ROWS) {
            uint32_t v = L[0];
#pragma unroll
            for (int r = 1; r < STG_ROWS; ++r) v = (lane == (uint32_t)r) ? L[r] : v;
            hullStart[lane] = v;
        }
        int nf = 0, nb = 0;
        // ---- 3. one z-plane at a time: its three hulls go to LDS while the next plane's loads are in flight -----------
        float4 t0[3], t1[3];
        auto fetch_plane = [&](int pl) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = 3 * pl + k;
                if (lane < len[r]) t0[k] = Sweep<R>::at32(sPos, L[r] + lane);
                if (lane + 64u < len[r]) t1[k] = Sweep<R>::at32(sPos, L[r] + 64u + lane);
            }
        };
        fetch_plane(0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint32_t O[4];
            O[0] = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) O[k + 1] = O[k] + len[3 * pl + k];
            if (pl) __syncthreads(); // the previous plane has been scanned
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = 3 * pl + k;
                if (lane < len[r]) pool[O[k] + lane] = t0[k];
                if (lane + 64u < len[r]) pool[O[k] + 64u + lane] = t1[k];
                for (uint32_t q = lane + 128u; q < len[r]; q += 64u) pool[O[k] + q] = Sweep<R>::at32(sPos, L[r] + q);
            }
            __syncthreads();
            if (pl < 2) fetch_plane(pl + 1);
            // ---- 4. scan the three rows of the plane out of LDS -------------------------------------------------------
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = 3 * pl + k;
                const uint32_t cnt = hi[r] - lo[r];
                const uint32_t off = lo[r] - L[r]; // first candidate, relative to the row's hull
                const uint32_t a = O[k] + off;
                for (uint32_t base = 0; base < cnt; base += STG_BATCH) {
                    float4 c[STG_BATCH];
#pragma unroll
                    for (int u = 0; u < STG_BATCH; ++u) c[u] = lds_read16(&pool[a + min(base + (uint32_t)u, cnt - 1u)]);
#pragma unroll
                    for (int u = 0; u < STG_BATCH; ++u) {
                        const uint32_t q = base + (uint32_t)u;
                        const uint32_t j = lo[r] + q;
                        const V3<R> d = p - xyz<R>(c[u]);
                        // the entry is ALWAYS stored: a hit goes to list slot nf, anything else to the spare row (no branch, no
                        // exec-mask traffic per candidate); a full list is marked `over` below and never read
                        if (FAST) {
                            const float d2 = __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x));
                            const bool in = (q < cnt) & (d2 < tF);
                            const float w = h2 - d2;
                            acc += in ? w * w * w : 0.f;
                            const bool hit = in & (j != i);
                            lst[hit ? min(nf, HIT_CAP - 1) : HIT_CAP][lane] = (Hit)(((uint32_t)r << 12) | (off + q));
                            nf += hit ? 1 : 0;
                        } else {
                            const bool hit = (q < cnt) & (j != i) & (dot(d, d) < tF);
                            const uint32_t tag = (uint32_t)(r * 3) + (j >= m1[r] ? 1u : 0u) + (j >= m2[r] ? 1u : 0u);
                            lst[hit ? min(nf, HIT_CAP - 1) : HIT_CAP][lane] = (Hit)(j | (tag << HIT_TAG_SHIFT));
                            nf += hit ? 1 : 0;
                        }
                    }
                }
            }
        }
        if (HAS_B) {
            hc.anyB = bmask != 0u;
            // Boundary cells.  A wall touches a wave of 64 consecutive slots typically with ONE OR TWO lanes (the first
            // particles of an x-row), each of which has ~50-100 boundary candidates: swept lane by lane that is ~20 rounds of
            // global loads at 2/64 lane utilisation — measured a quarter of this kernel's instructions.  So when few lanes
            // have boundary cells, the whole wave sweeps them for one such lane at a time (64 candidates per round, coalesced),
            // ranks the hits with a ballot and stores them straight into that lane's list.
            const unsigned long long wall = __ballot(bmask != 0u);
            if (wall != 0ull && __popcll(wall) <= STG_COOP_LANES) {
                unsigned long long todo = wall;
                while (todo) {
                    const int Ln = __builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.x), Ln));
                    const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.y), Ln));
                    const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.z), Ln));
                    uint32_t bm = (uint32_t)__builtin_amdgcn_readlane((int)bmask, Ln);
                    const uint32_t cxL = (uint32_t)__builtin_amdgcn_readlane((int)cx, Ln), cyL = (uint32_t)__builtin_amdgcn_readlane((int)cy, Ln),
                                   czL = (uint32_t)__builtin_amdgcn_readlane((int)cz, Ln);
                    const uint32_t nfL = (uint32_t)__builtin_amdgcn_readlane(nf, Ln);
                    uint32_t nbL = 0;
                    float part = 0.f;
                    while (bm) {
                        const int bit = __builtin_ctz(bm);
                        bm &= bm - 1u;
                        const int r = bit / 3, c = bit - r * 3;
                        const uint32_t hcell = umul24(umul24(czL + (uint32_t)(r / 3) - 1u, P.gridSize[1]), P.gridSize[0]) +
                                               umul24(cyL + (uint32_t)(r % 3) - 1u, P.gridSize[0]) + cxL - 1u + (uint32_t)c;
                        const uint32_t a = G.bCellStart[hcell], nT = G.bCellEnd[hcell] - a;
                        for (uint32_t base = 0; base < nT; base += 64u) {
                            const uint32_t q = base + lane;
                            const bool valid = q < nT;
                            const float4 cb = G.sB[a + (valid ? q : 0u)];
                            const V3<R> d = mk3<R>(qx, qy, qz) - xyz<R>(cb);
                            bool hit;
                            if (FAST) {
                                const float d2 = __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x));
                                const float w = fmaxf(h2 - d2, 0.f);
                                part += valid ? cb.w * (w * w * w) : 0.f;
                                hit = valid & (d2 < tB);
                            } else {
                                hit = valid & (dot(d, d) < tB);
                            }
                            const unsigned long long hm = __ballot(hit);
                            if (hit) {
                                const int slot = HIT_CAP - 1 - (int)(nbL + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull)));
                                const uint32_t e = FAST ? (((uint32_t)bit << 11) | (q & 2047u)) : ((a + q) | ((uint32_t)bit << HIT_TAG_SHIFT));
                                if (slot >= (int)nfL) lst[slot][Ln] = (Hit)e; // (slots below the fluid hits: the list overflows, see `over`)
                            }
                            nbL += (uint32_t)__popcll(hm);
                            if (FAST && nT > 2048u) nbL = HIT_CAP + 1; // (offset not encodable in 11 bits: reference-order walk)
                        }
                    }
                    if (FAST) {
                        for (int dlt = 32; dlt >= 1; dlt >>= 1) part += __shfl_xor(part, dlt);
                        if (lane == (uint32_t)Ln) bacc = part;
                    }
                    if (lane == (uint32_t)Ln) nb = (int)nbL;
                }
            } else
            while (bmask) { // boundary cells, ascending cell number, only on the lanes that have any (near a wall)
                const int bit = __builtin_ctz(bmask);
                bmask &= bmask - 1u;
                const int r = bit / 3, c = bit - r * 3;
                const uint32_t hcell = umul24(umul24(cz + (uint32_t)(r / 3) - 1u, P.gridSize[1]), P.gridSize[0]) +
                                       umul24(cy + (uint32_t)(r % 3) - 1u, P.gridSize[0]) + cx - 1u + (uint32_t)c;
                const uint32_t a = G.bCellStart[hcell], nT = G.bCellEnd[hcell] - a;
                for (uint32_t base = 0; base < nT; base += 4) {
                    float4 cb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) cb[u] = G.sB[a + min(base + (uint32_t)u, nT - 1u)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t q = base + (uint32_t)u;
                        const V3<R> d = p - xyz<R>(cb[u]);
                        if (FAST) {
                            const float d2 = __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x));
                            const float w = fmaxf(h2 - d2, 0.f);
                            bacc += (q < nT) ? cb[u].w * (w * w * w) : 0.f;
                            if ((q < nT) & (d2 < tB)) {
                                if (q >= 2048u) nb = HIT_CAP + 1; // (cannot be encoded in 11 bits: forces the reference-order walk)
                                lst[max(HIT_CAP - 1 - nb, 0)][lane] = (Hit)(((uint32_t)bit << 11) | (q & 2047u));
                                ++nb;
                            }
                        } else if ((q < nT) & (dot(d, d) < tB)) {
                            lst[max(HIT_CAP - 1 - nb, 0)][lane] = (Hit)((a + q) | ((uint32_t)bit << HIT_TAG_SHIFT));
                            ++nb;
                        }
                    }
                }
            }
        }
        // (an owner with a NaN / inf coordinate takes the reference-order walk: see Sweep::scan)
        const bool finiteOwner = (fabsf(p.x) < INFINITY) & (fabsf(p.y) < INFINITY) & (fabsf(p.z) < INFINITY);
        hc.nf = nf; hc.nb = nb; hc.over = (nf + nb > HIT_CAP) | !finiteOwner;
    } else {
        // grid-edge cells or hulls longer than the pool: the global-memory scan of nrs_kernels_tiled.h, its 32-bit lists in the
        // (unused) pool
        uint32_t (*wide)[STG_WAVE] = reinterpret_cast<uint32_t (*)[STG_WAVE]>(pool);
        if (active) hc = Sweep<R>::template scan<HAS_B, BF, STG_WAVE>(P, G, thr, sPos, i, p, wide);
        if (inb) {
            R d = 0.f, pr = 0.f;
            if (active) {
                if (hc.over) d = density_of<R, KSET, HAS_B>(P, G, sPos, i);
                else d = density_from_hits<R, KSET, HAS_B>(P, G, sPos, p, &wide[0][lane], STG_WAVE, hc, i);
                pr = tait_pressure<R>(P, d);
            }
            dens[i] = d;
            if (pres) pres[i] = pr;
            if (SHARE && !FAST && hb.gpos && active) publish_gather_records<R>(P, hb, i, p, own_sorted_velocity<R>(hb, i), d, pr); // what the list-driven force kernel gathers (HitBuffer, k_density_tiled)
            if (FAST && fq) {
                const float inv = active ? 1.0f / d : 0.f;
                FastPair z; z.pr = pr * inv * inv; z.invRho = inv;
                fq[i] = z;
            }
            if (SHARE) {
                hb.counts[i] = active ? (pack_counts(hc) | COUNTS_UNSTAGED) : 0u;
                if (active && !hc.over) {
                    for (int k = 0; k < hc.nf; ++k) hb.hits[(size_t)k * hb.stride + i] = wide[k][lane];
                    for (int k = 0; k < hc.nb; ++k) hb.hits[(size_t)(HIT_CAP - 1 - k) * hb.stride + i] = wide[HIT_CAP - 1 - k][lane];
                }
            }
        }
        return;
    }
    if (!inb) return;
    if (!active) {
        dens[i] = 0.f;
        if (pres) pres[i] = 0.f;
        if (FAST && fq) { FastPair z; z.pr = 0.f; z.invRho = 0.f; fq[i] = z; }
        if (SHARE) hb.counts[i] = 0u;
        return;
    }
    float d, pr;
    if (FAST) {
        d = (P.particleMass * P.kpoly) * acc + (P.restDensity * P.kpoly) * bacc;
        pr = P.gasStiffness * (pow7_fast(d * (1.0f / P.restDensity)) - 1.0f);
    } else {
        if (hc.over) d = density_of<R, KSET, HAS_B>(P, G, sPos, i);
        else d = density_from_hits<R, KSET, HAS_B>(P, G, sPos, p, reinterpret_cast<const uint32_t *>(&lst[0][lane]), STG_WAVE, hc, i);
        pr = tait_pressure<R>(P, d);
    }
    dens[i] = d;
    if (pres) pres[i] = pr;
    // the gather records of the list-driven force kernel, as k_density_tiled leaves them: round 2 made that kernel gather per-slot pairs, and this
    // launch did not write them — found by the first test that ran the staged launch (round 3)
    if (SHARE && !FAST && hb.gpos) publish_gather_records<float>(P, hb, i, p, own_sorted_velocity<float>(hb, i), d, pr);
    if (FAST && fq) {
        const float inv = 1.0f / d;
        FastPair z; z.pr = pr * inv * inv; z.invRho = inv;
        fq[i] = z;
    }
    if (SHARE) {
        hb.counts[i] = pack_counts(hc);
        if (!hc.over) {
            if (FAST) {
                for (int k = 0; k < hc.nf; ++k) {
                    const uint32_t e = lst[k][lane];
                    hb.hits[(size_t)k * hb.stride + i] = hullStart[e >> 12] + (e & 4095u);
                }
                for (int k = 0; k < hc.nb; ++k) {
                    const uint32_t e = lst[HIT_CAP - 1 - k][lane];
                    const uint32_t bit = e >> 11, r = bit / 3u, c = bit - r * 3u;
                    const uint32_t hcell = umul24(umul24(cz + r / 3u - 1u, P.gridSize[1]), P.gridSize[0]) + umul24(cy + r % 3u - 1u, P.gridSize[0]) +
                                           cx - 1u + c;
                    hb.hits[(size_t)(HIT_CAP - 1 - k) * hb.stride + i] = G.bCellStart[hcell] + (e & 2047u);
                }
            } else {
                for (int k = 0; k < hc.nf; ++k) hb.hits[(size_t)k * hb.stride + i] = lst[k][lane];
                for (int k = 0; k < hc.nb; ++k) hb.hits[(size_t)(HIT_CAP - 1 - k) * hb.stride + i] = lst[HIT_CAP - 1 - k][lane];
            }
        }
    }
}

// ---- FAST force terms (computeCellForces, sph_kernel_impl.cuh:442-604, Muller kernels of kernels_impl.cuh:85-154) ------
// Same sums as forces_from_hits, evaluated with reciprocals instead of IEEE divisions, one v_rsq per pair instead of
// three square roots, float powers, and fused multiply-adds; the order of the hits is irrelevant here, so the lists carry
// no cell tags.  Per neighbour: pos (16 B), vel (16 B) and the packed (p/rho^2, 1/rho) pair the density kernel left.
template <bool SURF, bool HAS_B>
NRS_DEV ForceAcc<float> forces_from_hits_fast(const Params<float> &P, const GridView<float> &G, const float4 *__restrict__ sPos,
                                              const float4 *__restrict__ sVel, const FastPair *__restrict__ fq, V3<float> pos1,
                                              V3<float> vel1, FastPair own, const uint32_t *lbase, uint32_t lstride, HitCounts hc)
{
    typedef float R;
    ForceAcc<R> A;
    A.fpres = A.fvisc = A.fsurf = A.fbound = mk3<R>(0, 0, 0);
    const float h = P.interactionRadius, h2 = h * h;
    const float m = P.particleMass;
    const float kprg = P.kpress_grad, kvg = P.kvisc_grad;
    const float c1 = -3.0f / P.kvisc_denum, c2 = 2.0f / h2, c3 = -0.5f * h; // c(len) = c1 len + c2 + c3 / len^3
    const float eps = 0.01f * h2;
    const float diameter = 2.0f * P.particleRadius, diameter2 = diameter * diameter;
    const float td = fmaxf(h2 - diameter2, 0.f);
    const float wAtDiameter = P.kpoly * td * td * td;
    const float ks = P.surfaceTension / P.particleMass * P.particleMass; // the reference's `kappa / m * m`
    // fluid hits, FORCE_BATCH at a time: the list entries of a batch are requested together, then the three gathers of
    // every pair of the batch (memory-level parallelism: the walk is bound by the latency of dependent gathers — one
    // exposed round trip per hit — not by arithmetic or bandwidth), then the arithmetic
    for (int k0 = 0; k0 < hc.nf; k0 += FORCE_BATCH) {
        uint32_t j[FORCE_BATCH];
#pragma unroll
        for (int u = 0; u < FORCE_BATCH; ++u) j[u] = lbase[(uint32_t)min(k0 + u, hc.nf - 1) * lstride] & HIT_INDEX;
        float4 pj[FORCE_BATCH], vj[FORCE_BATCH];
        FastPair qj[FORCE_BATCH];
#pragma unroll
        for (int u = 0; u < FORCE_BATCH; ++u) { pj[u] = sPos[j[u]]; vj[u] = sVel[j[u]]; qj[u] = fq[j[u]]; }
#pragma unroll
        for (int u = 0; u < FORCE_BATCH; ++u) {
            const bool on = k0 + u < hc.nf; // slots beyond the list repeat its last pair with zero weight
            const float rx = pos1.x - pj[u].x, ry = pos1.y - pj[u].y, rz = pos1.z - pj[u].z;
            const float d2 = __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
            const float inv = __builtin_amdgcn_rsqf(d2);
            const float len = d2 * inv;
            const float hl = fmaxf(h - len, 0.f);
            // pressure: m (p_i/rho_i^2 + p_j/rho_j^2) * kpress_grad * (r/|r|) (h-|r|)^2
            const float cp = on ? m * (own.pr + qj[u].pr) * kprg * (hl * hl) * inv : 0.f;
            A.fpres.x = __builtin_fmaf(cp, rx, A.fpres.x); A.fpres.y = __builtin_fmaf(cp, ry, A.fpres.y); A.fpres.z = __builtin_fmaf(cp, rz, A.fpres.z);
            // viscosity: m/rho_j (v_i - v_j) (r . gradW_visc) / (r^2 + 0.01 h^2), r . gradW_visc = kvisc_grad r^2 c(|r|)
            const float inv3 = inv * inv * inv;
            const float cl = (on & (d2 <= h2)) ? __builtin_fmaf(c3, inv3, __builtin_fmaf(c1, len, c2)) : 0.f;
            const float cv = m * qj[u].invRho * (kvg * cl * d2) * __builtin_amdgcn_rcpf(d2 + eps);
            A.fvisc.x = __builtin_fmaf(cv, vel1.x - vj[u].x, A.fvisc.x); A.fvisc.y = __builtin_fmaf(cv, vel1.y - vj[u].y, A.fvisc.y);
            A.fvisc.z = __builtin_fmaf(cv, vel1.z - vj[u].z, A.fvisc.z);
            if (SURF) {
                const float t = fmaxf(h2 - d2, 0.f);
                const float kern = (d2 > diameter2) ? P.kpoly * (t * t * t) : wAtDiameter;
                const float cs = on ? -ks * kern : 0.f;
                A.fsurf.x = __builtin_fmaf(cs, rx, A.fsurf.x); A.fsurf.y = __builtin_fmaf(cs, ry, A.fsurf.y); A.fsurf.z = __builtin_fmaf(cs, rz, A.fsurf.z);
            }
        }
    }
    return A;
}

// one boundary particle's contribution to the TOTAL force of a fluid particle (fast arithmetic): adhesion, pressure mirror
// and friction terms of computeCellForces (sph_kernel_impl.cuh:566-602) already combined as computeForces combines them
// (-m * fpres + 2 m mu * fvisc + fbound, :663-674) — all three are multiples of r
NRS_DEV V3<float> boundary_pair_force_fast(const Params<float> &P, FastPair own, V3<float> pos1, V3<float> vel1, float4 bq)
{
    const float h = P.interactionRadius, h2 = h * h, m = P.particleMass;
    const float psi = P.restDensity * bq.w;
    const float rx = pos1.x - bq.x, ry = pos1.y - bq.y, rz = pos1.z - bq.z;
    const float d2 = __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
    const float t = fmaxf(h2 - d2, 0.f);
    const float kern = P.kpoly * (t * t * t);
    const float g = P.kpoly_grad * (t * t); // gradW_poly6 = g * r
    const float cb = P.beta * psi * kern;
    const float cp = -m * psi * own.pr * g;
    const float nuWall = (P.viscosity * h * P.soundSpeed) * (own.invRho * own.invRho);
    const float approach = fmaxf(__builtin_fmaf(vel1.z, rz, __builtin_fmaf(vel1.y, ry, vel1.x * rx)), 0.f);
    const float friction = -nuWall * approach * (1.0f / (1.0f + 0.01f * h * h)); // |r/|r||^2 + eps h^2 in the denominator
    const float cv = -(m * psi * friction * g);
    const float k = cb - m * cp + (2.0f * m * P.viscosity) * cv;
    return mk3<float>(k * rx, k * ry, k * rz);
}

#ifndef FORCE_COOP_LANES
#define FORCE_COOP_LANES 4 // up to this many lanes of a wave with boundary hits: the wave evaluates them cooperatively
#endif
// FAST force launch: consumes the lists and the (p/rho^2, 1/rho) pairs of k_density_staged<FAST>; fused with integrate + hash
// like k_forces_lists (the epilogue — integration and the grid hash with its true division — is the exact one).
// Boundary hits belong to the one or two lanes of a wave that sit at a wall (up to HIT_CAP each): when few lanes have any,
// the wave evaluates them together — lane t takes hit t of the wall lane — and reduces the three force components with
// shuffles, instead of idling 62 lanes for up to HIT_CAP rounds.
template <bool SURF, bool HAS_B, bool FUSE>
__global__ __launch_bounds__(BLOCK) void k_forces_fast(Params<float> P, GridView<float> G, HitBuffer hb, const float4 *__restrict__ sPos,
                                                       const float4 *__restrict__ sVel, const float *__restrict__ sDens,
                                                       const float *__restrict__ sPres, const FastPair *__restrict__ fq,
                                                       float4 *__restrict__ forces, FusedOut<float> fo, uint32_t n)
{
    typedef float R;
    const uint32_t i = xcd_tile(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool inb = i < n;
    const float4 p4 = inb ? sPos[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v4 = inb ? sVel[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const V3<R> pos1 = xyz<R>(p4), vel1 = xyz<R>(v4);
    V3<R> f = mk3<R>(0, 0, 0);
    const bool active = inb && slab_active<R>(P, G, pos1.x);
    HitCounts hc;
    hc.nf = 0; hc.nb = 0; hc.over = false; hc.anyB = false;
    if (active) hc = unpack_counts(hb.counts[i]);
    FastPair own; own.pr = 0.f; own.invRho = 0.f;
    bool listed = false; // the particle's force comes from its hit lists (as opposed to the overflow walk)
    if (active) {
        if (hc.over) { // list overflow: the reference-order cell walk (exact arithmetic)
            const R dens = sDens[i], pres = sPres[i];
            const ForceAcc<R> A = gather_forces<R, KS_MULLER, SURF, HAS_B>(P, G, i, pos1, vel1, dens, pres, sPos, sVel, sDens, sPres);
            f = sesph_total_force<R>(P, A, dens);
        } else {
            listed = true;
            own = fq[i];
            const ForceAcc<R> A = forces_from_hits_fast<SURF, HAS_B>(P, G, sPos, sVel, fq, pos1, vel1, own, hb.hits + i, hb.stride, hc);
            // computeForces' final combination (sph_kernel_impl.cuh:663-674); rho cancels in the pressure term
            const float m = P.particleMass, mv = 2.0f * m * P.viscosity;
            f.x = -m * A.fpres.x + mv * A.fvisc.x + P.gravity[0] * m + A.fsurf.x;
            f.y = -m * A.fpres.y + mv * A.fvisc.y + P.gravity[1] * m + A.fsurf.y;
            f.z = -m * A.fpres.z + mv * A.fvisc.z + P.gravity[2] * m + A.fsurf.z;
        }
    }
    if (HAS_B) {
        const int nbMine = listed ? hc.nb : 0;
        const unsigned long long wall = __ballot(nbMine > 0);
        if (wall != 0ull && __popcll(wall) <= FORCE_COOP_LANES) {
            unsigned long long todo = wall;
            while (todo) {
                const int Ln = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int nbL = __builtin_amdgcn_readlane(nbMine, Ln);
                const uint32_t iL = (uint32_t)__builtin_amdgcn_readlane((int)i, Ln);
                V3<R> pl, vl;
                pl.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pos1.x), Ln));
                pl.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pos1.y), Ln));
                pl.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pos1.z), Ln));
                vl.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vel1.x), Ln));
                vl.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vel1.y), Ln));
                vl.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vel1.z), Ln));
                FastPair ol;
                ol.pr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own.pr), Ln));
                ol.invRho = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own.invRho), Ln));
                V3<R> df = mk3<R>(0, 0, 0);
                if ((int)lane < nbL) { // nbL <= HIT_CAP < 64: one round
                    const uint32_t j = hb.hits[(size_t)(HIT_CAP - 1 - (int)lane) * hb.stride + iL] & HIT_INDEX;
                    df = boundary_pair_force_fast(P, ol, pl, vl, G.sB[j]);
                }
                for (int dlt = 32; dlt >= 1; dlt >>= 1) {
                    df.x += __shfl_xor(df.x, dlt); df.y += __shfl_xor(df.y, dlt); df.z += __shfl_xor(df.z, dlt);
                }
                if (lane == (uint32_t)Ln) f = f + df;
            }
        } else if (nbMine > 0) {
            for (int k = 0; k < nbMine; ++k) {
                const uint32_t j = hb.hits[(size_t)(HIT_CAP - 1 - k) * hb.stride + i] & HIT_INDEX;
                f = f + boundary_pair_force_fast(P, own, pos1, vel1, G.sB[j]);
            }
        }
    }
    if (inb) forces_epilogue<R, KS_MULLER, SURF, HAS_B, FUSE>(P, p4, v4, f, forces, fo, i);
}

template <int KSET, bool HAS_B>
static inline void launch_density_staged(hipStream_t stream, const Params<float> &P, const GridView<float> &G, const HitBuffer *share,
                                         bool fast, const float4 *sPos, float *dens, float *pres, FastPair *fq, uint32_t n)
{
    const CutThresholds thr = make_thresholds<float>(P);
    const dim3 g((n + STG_WAVE - 1) / STG_WAVE), b(STG_WAVE);
    HitBuffer hb = {nullptr, nullptr, 0};
    if (share) hb = *share;
    constexpr unsigned pad = NRS_DBG_LDS_PAD; // occupancy experiment (compile-time, see nrs_kernels_tiled.h)
    if (fast && KSET == KS_MULLER) {
        if (share) hipLaunchKernelGGL((k_density_staged<KS_MULLER, HAS_B, true, true>), g, b, pad, stream, P, G, thr, sPos, dens, pres, fq, hb, n);
        else hipLaunchKernelGGL((k_density_staged<KS_MULLER, HAS_B, false, true>), g, b, 0, stream, P, G, thr, sPos, dens, pres, fq, hb, n);
    } else {
        if (share) hipLaunchKernelGGL((k_density_staged<KSET, HAS_B, true, false>), g, b, pad, stream, P, G, thr, sPos, dens, pres, (FastPair *)nullptr, hb, n);
        else hipLaunchKernelGGL((k_density_staged<KSET, HAS_B, false, false>), g, b, 0, stream, P, G, thr, sPos, dens, pres, (FastPair *)nullptr, hb, n);
    }
}

template <bool SURF, bool HAS_B>
static inline void launch_forces_fast(hipStream_t stream, const Params<float> &P, const GridView<float> &G, const HitBuffer &lists,
                                      const float4 *sPos, const float4 *sVel, const float *dens, const float *pres, const FastPair *fq,
                                      float4 *forces, const FusedOut<float> *fused, uint32_t n)
{
    FusedOut<float> fo;
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
    if (fused) hipLaunchKernelGGL((k_forces_fast<SURF, HAS_B, true>), g, b, 0, stream, P, G, lists, sPos, sVel, dens, pres, fq, forces, fo, n);
    else hipLaunchKernelGGL((k_forces_fast<SURF, HAS_B, false>), g, b, 0, stream, P, G, lists, sPos, sVel, dens, pres, fq, forces, fo, n);
}

} // namespace nrs
