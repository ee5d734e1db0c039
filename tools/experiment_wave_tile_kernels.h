// nrs_kernels_wave.h — LDS-staged gather kernels: one wavefront (64 lanes) per tile of 64 consecutive SORTED
// particles.  These are the production density / force kernels of the SESPH step on gfx950.
//
// Why: in the global-memory form of the two-phase gather (nrs_kernels_tiled.h) every thread issues ≈100
// divergent 4-16 B loads (cell table + candidate positions); rocprof showed both gathers bound by the vector
// memory pipeline (L1/TA), equally slow although the force kernel does 10x the arithmetic per hit.  Here the
// candidate set of a tile is fetched ONCE, coalesced, into LDS and the scan runs out of LDS:
//
//   A. pieces   64 consecutive sorted particles cover 1-4 runs of cells inside x-rows of the grid ("pieces");
//               found with one ballot over the row id decoded from the sorted hash.
//   B. segments for every piece and each of the 9 (dz,dy) neighbour rows, the cells [xa-1, xb+1] are ONE
//               contiguous run of the sorted array.  The wave reads that slice of cellStart coalesced, turns it
//               into a "next non-empty start" table (suffix-min by DPP shuffles) kept in LDS, and gets the run
//               [segLo, segHi).  Boundary-cell occupancy of the slice is recorded per segment.
//   C. staging  the ≤36 runs (≈9 x (64 + margins) positions) are copied into LDS with coalesced 16 B loads.
//   D. scan     each lane looks up its own 3-cell range with one 16 B LDS read, then tests its candidates in
//               batches of 8 independent ds_read_b128 against the squared-distance threshold; hits are appended
//               to the lane's list in LDS.
//   E. process  as in nrs_kernels_tiled.h (same device functions, same summation order ⇒ bit-identical results).
//
// A single wave per workgroup means no s_barrier anywhere (LDS traffic of one wave is ordered).  Tiles that do
// not fit the fixed LDS budget (more than 4 pieces, an x-wrap at the grid edge, very sparse rows) fall back to
// the global-memory scan of nrs_kernels_tiled.h for that tile; a lane whose hit list overflows falls back to the
// reference-order routine.  LDS: 25 KiB per tile (fp32) ⇒ 6 tiles per CU.
#pragma once
#include "nrs_kernels_tiled.h"

namespace nrs {

constexpr int WT = 64;        // lanes per tile
constexpr int WT_MAXP = 4;    // row pieces per tile
constexpr int WT_SEGS = WT_MAXP * 9;
constexpr int WT_CAND = 768;  // staged candidate positions per tile
constexpr int WT_TBL = 1024;  // "next non-empty start" table entries per tile
constexpr int WT_BATCH = 8;   // candidates fetched per lane per LDS round trip

struct TileGeom { uint32_t lgx, lgy; int dbg; }; // log2 of gridSize.x / .y (power-of-two grids)

template <typename R> struct WaveTileLds {
    typename Vec4T<R>::type cand[WT_CAND];
    uint32_t tbl[WT_TBL];
    uint32_t segLo[WT_SEGS], segHi[WT_SEGS], segCand[WT_SEGS + 1], segTbl[WT_SEGS], segHasB[WT_SEGS];
    uint32_t pieceX0[WT_MAXP]; // first table cell of the piece = xa-1
    uint32_t lst[HIT_CAP][WT];
};

NRS_DEV uint32_t rdlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// Steps A-C.  Returns (wave-uniform) whether the tile was staged; outputs the lane's piece and cell-x.
template <typename R, bool HAS_B>
NRS_DEV bool wave_tile_build(const Params<R> &P, const GridView<R> &G, const TileGeom tg, const uint32_t *__restrict__ hashS,
                             const typename Vec4T<R>::type *__restrict__ sPos, uint32_t n, WaveTileLds<R> &L,
                             int &pieceIdx, uint32_t &cxOut)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WT + lane;
    const bool valid = i < n;
    const uint32_t h = hashS[valid ? i : n - 1];
    const uint32_t mx = P.gridSize[0] - 1, my = P.gridSize[1] - 1, mz = P.gridSize[2] - 1;
    const uint32_t row = h >> tg.lgx, cx = h & mx;
    const uint32_t prevRow = __shfl_up(row, 1);
    const bool head = valid && (lane == 0 || row != prevRow);
    const unsigned long long hm = __ballot(head);
    const int nP = __popcll(hm);
    pieceIdx = __popcll(hm & ((2ull << lane) - 1ull)) - 1;
    cxOut = cx;
    if (nP > WT_MAXP) return false;
    const int nValid = __popcll(__ballot(valid));

    uint32_t tblUsed = 0;
    unsigned long long rem = hm;
    for (int p = 0; p < nP; ++p) {
        const int hl = __builtin_ctzll(rem);
        rem &= rem - 1;
        const int tl = (rem ? __builtin_ctzll(rem) : nValid) - 1;
        const uint32_t prow = rdlane(row, hl), xa = rdlane(cx, hl), xb = rdlane(cx, tl);
        if (xa == 0 || xb == mx) return false;                 // the 3-cell window would wrap in x
        const uint32_t ncell = xb - xa + 4;                     // table cells xa-1 .. xb+2 (last one is a sentinel)
        if (tblUsed + 9 * ncell > (uint32_t)WT_TBL) return false; // very sparse row: table does not fit
        const uint32_t cy = prow & my, cz = prow >> tg.lgy;
        if (lane == 0) L.pieceX0[p] = xa - 1;
        const int nchunks = (int)((ncell + 63) / 64);
        // The nine neighbour rows are handled together so that their table loads are all in flight at once
        // (one memory round trip per 64-cell chunk instead of nine dependent ones).
        uint32_t rowBase[9], carry[9], lastCell[9];
        bool anyB[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int dz = r / 3 - 1, dy = r % 3 - 1;
            const uint32_t nrow = (((cz + (uint32_t)dz) & mz) << tg.lgy) + ((cy + (uint32_t)dy) & my);
            rowBase[r] = nrow << tg.lgx;
            carry[r] = CELL_EMPTY;    // min start seen so far, scanning cells from high to low
            lastCell[r] = CELL_EMPTY; // highest non-empty cell
            anyB[r] = false;
        }
        for (int ch = nchunks - 1; ch >= 0; --ch) {
            const uint32_t k = (uint32_t)ch * 64u + lane;
            const bool inR = k + 1 < ncell; // real cells; k == ncell-1 is the sentinel
            uint32_t v[9], bs[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                v[r] = inR ? G.cellStart[rowBase[r] + xa - 1 + k] : CELL_EMPTY;
                bs[r] = (HAS_B && inR) ? G.bCellStart[rowBase[r] + xa - 1 + k] : CELL_EMPTY;
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                if (HAS_B) anyB[r] = anyB[r] || (__ballot(bs[r] != CELL_EMPTY) != 0ull);
                const unsigned long long ne = __ballot(v[r] != CELL_EMPTY);
                if (lastCell[r] == CELL_EMPTY && ne)
                    lastCell[r] = xa - 1 + (uint32_t)ch * 64u + (63u - (uint32_t)__builtin_clzll(ne));
                uint32_t w = v[r];
                for (int off = 1; off < 64; off <<= 1) { // inclusive suffix-min across lanes
                    const uint32_t t = __shfl_down(w, off);
                    w = min(w, t);
                }
                w = min(w, carry[r]);
                carry[r] = rdlane(w, 0);
                if (k < ncell) L.tbl[tblUsed + (uint32_t)r * ncell + k] = w;
            }
        }
        uint32_t hi[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) hi[r] = (lastCell[r] != CELL_EMPTY) ? G.cellEnd[rowBase[r] + lastCell[r]] : 0u;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (lane == 0) {
                const int s = p * 9 + r;
                L.segLo[s] = (lastCell[r] != CELL_EMPTY) ? carry[r] : 0u;
                L.segHi[s] = hi[r];
                L.segTbl[s] = tblUsed + (uint32_t)r * ncell;
                L.segHasB[s] = anyB[r] ? 1u : 0u;
            }
        }
        tblUsed += 9 * ncell;
    }
    __syncthreads();
    // exclusive prefix of the run lengths → LDS offsets of the staged runs
    const int S = nP * 9;
    const uint32_t len = (int)lane < S ? L.segHi[lane] - L.segLo[lane] : 0u;
    uint32_t incl = len;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)lane >= off) incl += t;
    }
    const uint32_t total = rdlane(incl, 63);
    if (total > (uint32_t)WT_CAND) return false;
    if ((int)lane <= S) L.segCand[lane] = incl - len;
    __syncthreads();
    int s = 0;
    for (uint32_t idx0 = lane; idx0 < total; idx0 += 4 * 64) { // coalesced 16 B loads (4 in flight) → ds_write_b128
        typename Vec4T<R>::type q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t idx = idx0 + (uint32_t)u * 64u;
            if (idx < total) {
                while (idx >= L.segCand[s + 1]) ++s;
                q[u] = sPos[L.segLo[s] + (idx - L.segCand[s])];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t idx = idx0 + (uint32_t)u * 64u;
            if (idx < total) L.cand[idx] = q[u];
        }
    }
    __syncthreads();
    return true;
}

// Step D for one lane.  Returns the hit count or -1 on list overflow.
template <typename R, bool HAS_B, int BFILT>
NRS_DEV int wave_tile_scan(const Params<R> &P, const GridView<R> &G, const TileGeom tg, const CutThresholds thr,
                           WaveTileLds<R> &L, uint32_t self, uint32_t selfHash, V3<R> p, int pieceIdx, uint32_t cx)
{
    typedef typename Vec4T<R>::type T4;
    const uint32_t lane = threadIdx.x;
    const uint32_t mx = P.gridSize[0] - 1, my = P.gridSize[1] - 1, mz = P.gridSize[2] - 1;
    const float tF = thr.lenLtIr;
    const float tB = BFILT == 2 ? INFINITY : (BFILT == 1 ? thr.r2LeH2 : thr.lenLtIr);
    int cnt = 0;
    uint32_t pend = HIT_NEWPART;
    bool over = false;
    const uint32_t k0 = cx - 1 - L.pieceX0[pieceIdx];
    const uint32_t row = selfHash >> tg.lgx;
    const uint32_t cy = row & my, cz = row >> tg.lgy;

    auto push = [&](uint32_t ent) {
        if (cnt < HIT_CAP) L.lst[cnt][lane] = ent | pend; else over = true;
        ++cnt;
        pend = 0;
    };

    for (int r = 0; r < 9; ++r) {
        const int s = pieceIdx * 9 + r;
        const uint32_t tb = L.segTbl[s] + k0;
        const uint32_t segHi = L.segHi[s], segLo = L.segLo[s], candBase = L.segCand[s];
        // next-non-empty starts of cells cx-1, cx, cx+1, cx+2, clamped to the run
        const uint32_t t0 = min(L.tbl[tb], segHi), t1 = min(L.tbl[tb + 1], segHi), t2 = min(L.tbl[tb + 2], segHi),
                       t3 = min(L.tbl[tb + 3], segHi);
        if (!HAS_B || !L.segHasB[s]) {
            // merged 3-cell run [t0, t3); partial sums restart where cell cx and cell cx+1 begin
            const uint32_t nT = t3 - t0;
            const uint32_t off = candBase + (t0 - segLo);
            pend = HIT_NEWPART;
            for (uint32_t base = 0; base < nT; base += WT_BATCH) {
                T4 c[WT_BATCH];
#pragma unroll
                for (int u = 0; u < WT_BATCH; ++u) c[u] = L.cand[min(off + base + u, (uint32_t)WT_CAND - 1u)];
#pragma unroll
                for (int u = 0; u < WT_BATCH; ++u) {
                    const uint32_t q = base + u;
                    if (q < nT) {
                        const uint32_t j = t0 + q;
                        if (j == t1 || j == t2) pend = HIT_NEWPART;
                        if (j != self) {
                            const V3<R> d = p - xyz<R>(c[u]);
                            if (dot(d, d) < tF) push(j);
                        }
                    }
                }
            }
        } else {
            // boundary particles near this row: keep the reference's cell-by-cell order (fluid, then boundary)
            const int dz = r / 3 - 1, dy = r % 3 - 1;
            const uint32_t nrow = (((cz + (uint32_t)dz) & mz) << tg.lgy) + ((cy + (uint32_t)dy) & my);
            const uint32_t hb = (nrow << tg.lgx) + cx - 1;
            const uint32_t ts[4] = {t0, t1, t2, t3};
            for (int c = 0; c < 3; ++c) {
                pend = HIT_NEWPART;
                for (uint32_t j = ts[c]; j < ts[c + 1]; ++j) {
                    if (j != self) {
                        const V3<R> d = p - xyz<R>(L.cand[candBase + (j - segLo)]);
                        if (dot(d, d) < tF) push(j);
                    }
                }
                pend = HIT_NEWPART;
                const uint32_t bs = G.bCellStart[hb + c];
                if (bs != CELL_EMPTY) {
                    const uint32_t be = G.bCellEnd[hb + c];
                    for (uint32_t j = bs; j < be; ++j) {
                        const V3<R> d = p - xyz<R>(G.sB[j]);
                        if (dot(d, d) < tB) push(j | HIT_BOUNDARY);
                    }
                }
            }
        }
    }
    return over ? -1 : cnt;
}

// ---- density + Tait pressure ----------------------------------------------------------------------------
template <typename R, int KSET, bool HAS_B>
__global__ __launch_bounds__(WT) void k_density_wave(Params<R> P, GridView<R> G, TileGeom tg, CutThresholds thr,
                                                     const uint32_t *__restrict__ hashS,
                                                     const typename Vec4T<R>::type *__restrict__ sPos,
                                                     R *__restrict__ dens, R *__restrict__ pres, uint32_t n)
{
    __shared__ WaveTileLds<R> L;
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WT + lane;
    int pieceIdx;
    uint32_t cx;
    const bool staged = wave_tile_build<R, HAS_B>(P, G, tg, hashS, sPos, n, L, pieceIdx, cx);
    if (i >= n) return;
    if (tg.dbg == 1) { dens[i] = (R)(staged ? L.segLo[0] : 0); return; }
    const V3<R> p = xyz<R>(sPos[i]);
    if (!slab_active<R>(P, G, p.x)) { dens[i] = (R)0; if (pres) pres[i] = (R)0; return; }
    int cnt;
    if (staged) cnt = wave_tile_scan<R, HAS_B, 0>(P, G, tg, thr, L, i, hashS[i], p, pieceIdx, cx);
    else cnt = Sweep<R>::template scan<HAS_B, 0, WT>(P, G, thr, sPos, i, p, L.lst);
    if (tg.dbg == 2) { dens[i] = (R)cnt; return; }
    if (tg.dbg == 3) { dens[i] = (R)(staged ? 1 : 0); return; }
    R d;
    if (cnt < 0) d = density_of<R, KSET, HAS_B>(P, G, sPos, i);
    else d = density_from_hits<R, KSET, HAS_B, WT>(P, G, sPos, p, L.lst, lane, cnt);
    dens[i] = d;
    if (pres) pres[i] = tait_pressure<R>(P, d);
}

// ---- forces; FUSE: also integrate (integrate_functor, sph_kernel_impl.cuh:71-100) and hash the new position
//      (calcHashD :127-145) for the next step, writing straight into the next step's input arrays ------------
template <typename R> struct FusedOut {
    typedef typename Vec4T<R>::type T4;
    T4 *newPos, *newVel;    // next step's "unsorted" arrays
    uint32_t *hash, *index; // next step's keys / values
};

template <typename R, int KSET, bool SURF, bool HAS_B, bool FUSE>
__global__ __launch_bounds__(WT) void k_forces_wave(Params<R> P, GridView<R> G, TileGeom tg, CutThresholds thr,
                                                    const uint32_t *__restrict__ hashS,
                                                    const typename Vec4T<R>::type *__restrict__ sPos,
                                                    const typename Vec4T<R>::type *__restrict__ sVel,
                                                    const R *__restrict__ sDens, const R *__restrict__ sPres,
                                                    typename Vec4T<R>::type *__restrict__ forces, FusedOut<R> fo,
                                                    uint32_t n)
{
    typedef typename Vec4T<R>::type T4;
    __shared__ WaveTileLds<R> L;
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WT + lane;
    int pieceIdx;
    uint32_t cx;
    const bool staged = wave_tile_build<R, HAS_B>(P, G, tg, hashS, sPos, n, L, pieceIdx, cx);
    if (i >= n) return;
    const T4 p4 = sPos[i];
    const V3<R> pos1 = xyz<R>(p4);
    V3<R> f = mk3<R>(0, 0, 0);
    T4 v4;
    if (FUSE) v4 = sVel[i];
    if (slab_active<R>(P, G, pos1.x)) {
        if (!FUSE) v4 = sVel[i];
        const V3<R> vel1 = xyz<R>(v4);
        const R dens = sDens[i], pres = sPres[i];
        constexpr int BF = (KSET == KS_MULLER ? 1 : 2);
        int cnt;
        if (staged) cnt = wave_tile_scan<R, HAS_B, BF>(P, G, tg, thr, L, i, hashS[i], pos1, pieceIdx, cx);
        else cnt = Sweep<R>::template scan<HAS_B, BF, WT>(P, G, thr, sPos, i, pos1, L.lst);
        ForceAcc<R> A;
        if (cnt < 0) A = gather_forces<R, KSET, SURF, HAS_B>(P, G, i, pos1, vel1, dens, pres, sPos, sVel, sDens, sPres);
        else A = forces_from_hits<R, KSET, SURF, HAS_B, WT>(P, G, sPos, sVel, sDens, sPres, pos1, vel1, dens, pres, L.lst, lane, cnt);
        f = sesph_total_force<R>(P, A, dens);
    }
    if (forces) forces[i] = mk4<R>(f, (R)0);
    if (FUSE) {
        const R dt = P.timestep, m1 = P.particleMass;
        V3<R> v = xyz<R>(v4);
        const V3<R> accel = dt * f / m1;
        v = v + accel;
        const V3<R> pn = pos1 + dt * v;
        fo.newPos[i] = mk4<R>(pn, p4.w);
        fo.newVel[i] = mk4<R>(v, v4.w);
        const I3 g = calcGridPos<R>(P, pn);
        fo.hash[i] = calcGridHash<R>(P, g.x, g.y, g.z);
        fo.index[i] = i;
    }
}

static inline uint32_t ilog2(uint32_t v) { uint32_t l = 0; while ((1u << l) < v) ++l; return l; }

// the wave-tile kernels decode (row, cell-x) from the hash: needs power-of-two grids whose hash is exactly
// (z*gy + y)*gx + x, i.e. the 24-bit multiplies of calcGridHash must not truncate
template <typename R> static inline bool wave_tile_grid_ok(const Params<R> &P)
{
    return is_pow2(P.gridSize[0]) && is_pow2(P.gridSize[1]) && is_pow2(P.gridSize[2]) && P.gridSize[0] <= (1u << 24) &&
           (uint64_t)P.gridSize[1] * P.gridSize[2] <= (1ull << 24) &&
           (uint64_t)P.gridSize[0] * P.gridSize[1] * P.gridSize[2] <= (1ull << 31);
}

template <typename R, int KSET, bool HAS_B>
static inline void launch_density_tiled(hipStream_t stream, const Params<R> &P, const GridView<R> &G, const uint32_t *hashSorted,
                                        const typename Vec4T<R>::type *sPos, R *dens, R *pres, uint32_t n)
{
    const CutThresholds thr = make_thresholds<R>(P);
    const TileGeom tg = {ilog2(P.gridSize[0]), ilog2(P.gridSize[1]), getenv("NEREUS_DBG_STOP") ? atoi(getenv("NEREUS_DBG_STOP")) : 0};
    hipLaunchKernelGGL((k_density_wave<R, KSET, HAS_B>), dim3((n + WT - 1) / WT), dim3(WT), 0, stream, P, G, tg, thr, hashSorted,
                       sPos, dens, pres, n);
}
template <typename R, int KSET, bool SURF, bool HAS_B>
static inline void launch_forces_tiled(hipStream_t stream, const Params<R> &P, const GridView<R> &G, const uint32_t *hashSorted,
                                       const typename Vec4T<R>::type *sPos, const typename Vec4T<R>::type *sVel, const R *dens,
                                       const R *pres, typename Vec4T<R>::type *forces, const FusedOut<R> *fused, uint32_t n)
{
    const CutThresholds thr = make_thresholds<R>(P);
    const TileGeom tg = {ilog2(P.gridSize[0]), ilog2(P.gridSize[1]), getenv("NEREUS_DBG_STOP") ? atoi(getenv("NEREUS_DBG_STOP")) : 0};
    FusedOut<R> fo;
    fo.newPos = fo.newVel = nullptr;
    fo.hash = fo.index = nullptr;
    if (fused) {
        fo = *fused;
        hipLaunchKernelGGL((k_forces_wave<R, KSET, SURF, HAS_B, true>), dim3((n + WT - 1) / WT), dim3(WT), 0, stream, P, G, tg, thr,
                           hashSorted, sPos, sVel, dens, pres, forces, fo, n);
    } else {
        hipLaunchKernelGGL((k_forces_wave<R, KSET, SURF, HAS_B, false>), dim3((n + WT - 1) / WT), dim3(WT), 0, stream, P, G, tg, thr,
                           hashSorted, sPos, sVel, dens, pres, forces, fo, n);
    }
}

} // namespace nrs
