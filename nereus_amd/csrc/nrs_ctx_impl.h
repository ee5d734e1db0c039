// nrs_ctx_impl.h — the context object behind the C ABI of libnereus_hip.so (include/nereus_hip.h; entry points in nrs_abi.hip).
//
// The context owns the device-resident particle state and sequences one update() exactly as the
// reference's host classes do (SPH::update sph/sph.cpp:215-285, IISPH::update sph/iisph/iisph.cpp:170-217,
// predictAdvection/pressureSolve sph/sph_cuda.cu:513-899) — minus the per-step PCIe copies: the
// "unsorted" arrays of step t+1 are the integrated sorted arrays of step t (buffer swap), which is what
// the reference obtains by copying sorted→host→device.
#pragma once
#include <sched.h>
#include "nrs_ctx_base.h"
#include <rocprim/rocprim.hpp>

#include "nrs_kernels_ref.h"
#include "nrs_kernels_tiled.h"
#include "nrs_kernels_staged.h"
#include <type_traits>
#include "nrs_kernels_iisph.h"
#include "nrs_kernels_slab.h"
#include "nrs_kernels_resort.h"
#include <climits>

namespace nrs {

static inline uint32_t nblocks(uint64_t n) { return (uint32_t)((n + BLOCK - 1) / BLOCK); }

// Radix sort of (hash, index) pairs.  rocPRIM's onesweep sorts 8 key bits per pass by default, so the 25-27-bit
// hashes of the dam-break grids take 4 passes; with 9 bits per pass they take 3.
using SortCfg9 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, 9,
                                                                                rocprim::block_radix_rank_algorithm::match>>;
using SortCfg10 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                             rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, 10,
                                                                                 rocprim::block_radix_rank_algorithm::match>>;
static inline hipError_t sort_pairs(void *tmp, size_t &bytes, rocprim::double_buffer<uint32_t> &k, rocprim::double_buffer<uint32_t> &v,
                                    size_t n, unsigned bits, hipStream_t stream)
{
    if (bits > 24 && bits <= 27) return rocprim::radix_sort_pairs<SortCfg9>(tmp, bytes, k, v, n, 0u, bits, stream);
    if (bits > 27 && bits <= 30) return rocprim::radix_sort_pairs<SortCfg10>(tmp, bytes, k, v, n, 0u, bits, stream);
    return rocprim::radix_sort_pairs(tmp, bytes, k, v, n, 0u, bits, stream);
}

// Radix sort of the movers of the coherent re-sort (nrs_kernels_resort.h): u64 keys "hash << 32 | slot", only the hash
// bits are sorted (the slots are already ascending and the sort is stable).  A few hundred thousand keys: onesweep
// from 8192 keys on (rocPRIM's default switches to its merge sort below 1 M keys: measured 104 vs 66 us at 300 k).
template <unsigned BITS>
using MoverSortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, BITS,
                                                                                    rocprim::block_radix_rank_algorithm::match>, 8192>;
static inline hipError_t sort_movers(void *tmp, size_t &bytes, rocprim::double_buffer<uint64_t> &k, size_t m, unsigned bits, hipStream_t stream)
{
    if (bits > 24 && bits <= 27) return rocprim::radix_sort_keys<MoverSortCfg<9>>(tmp, bytes, k, m, 32u, 32u + bits, stream);
    if (bits > 27 && bits <= 30) return rocprim::radix_sort_keys<MoverSortCfg<10>>(tmp, bytes, k, m, 32u, 32u + bits, stream);
    return rocprim::radix_sort_keys<MoverSortCfg<8>>(tmp, bytes, k, m, 32u, 32u + bits, stream);
}

static uint32_t next_pow2(uint32_t v) // sph/sph.cpp:300-311
{
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    v++;
    return v;
}

template <typename R, int KSET, bool SURF> struct Ctx : CtxBase {
    typedef typename Vec4T<R>::type T4;
    // PU: the parameters as the caller set them (GLOBAL grid; what nrs_get_params returns).  P: what the kernels get — PU
    // with numBodies = first column of the cell-table window (0 = whole grid) and, for a slab rank, gridSize[0] / numCells of
    // that window: a rank keeps cell tables only for its own cell-x columns + halo, so table memory (and the radix-sort key
    // width) stop growing with the number of ranks.  Cell coordinates stay global (calcGridPos is unchanged), only the hash
    // rebases x: the sort order, and with it every per-particle result, is the one of the global grid.
    Params<R> PU, P;
    int winBase = 0;
    uint32_t winW = 0; // 0: no window
    void derive_kernel_params()
    {
        P = PU;
        P.numBodies = 0;
        if (winW && winW < PU.gridSize[0]) {
            P.numBodies = (uint32_t)winBase;
            P.gridSize[0] = winW;
            P.numCells = winW * PU.gridSize[1] * PU.gridSize[2];
        }
        // compact scan candidates (nrs_math.h): quanta per metre, threshold of the superset test, and whether the geometry allows
        // it at all (a pair inside the interaction radius must be less than two cells apart on every axis)
        float hq = 0.0f;
        qOk = NRS_COMPACT_SCAN != 0 && P.gridSize[0] >= 4u; // (narrower grids alias the row's three cells: tags by position fail)
        for (int a = 0; a < 3; ++a) {
            qc.o[a] = (double)P.worldOrigin[a];
            qc.s[a] = QP_PER_CELL / (float)P.cellSize[a];
            const float r = (float)P.interactionRadius * qc.s[a];
            qOk = qOk && P.cellSize[a] > (R)0 && std::isfinite(qc.s[a]) && (r + QP_MARGIN < QP_HALF);
            hq = std::max(hq, r);
        }
        const double lim = (double)hq + (double)QP_MARGIN;
        qT = qOk ? (uint32_t)std::ceil(lim * lim) + 1u : 0u;
    }
    QuantCfg qc;
    uint32_t qT = 0;
    bool qOk = false;
    DevBuf gatherPos, gatherVel; // (x, y, z, p / rho^2) and (vx, vy, vz, m / rho) per sorted slot: density kernel -> force kernel of the same step (HitBuffer)
    DevBuf qpos; // two words per sorted slot (+ 4 slots of padding), written by the reorder kernels
    // hit lists are built (and the kernels that consume them used) only when the scan that builds them can run
    bool lists_ok() const { return hitBuf.p != nullptr && (NRS_COMPACT_SCAN == 0 || (qOk && qpos.p != nullptr)); }
    nrs_config cfg;
    uint64_t cap = 0, n = 0, nb = 0;
    bool midStep = false; // a partial step left the state mid-update
    // particle state: A = current ("unsorted" input of the next step), B = sorted work arrays
    DevBuf posA, posB, velA, velB, presA, presB, dens, forces;
    DevBuf hashA, hashB, indexA, indexB, inv, sortTmp;
    uint32_t *hashCur = nullptr, *indexCur = nullptr; // sorted keys/values after the sort stage
    bool hashReady = false;                           // the fused force kernel already wrote the next step's keys/values
    uint32_t *hashNext = nullptr, *indexNext = nullptr;
    DevBuf cellStart, cellEnd, bCellStart, bCellEnd;
    uint32_t cellsAllocated = 0;
    // boundaries
    DevBuf bSorted, bHash, bIndex, bHashAlt, bIndexAlt;
    uint32_t *bHashCur = nullptr, *bIndexCur = nullptr;
    // IISPH
    DevBuf densAdv, densCorr, P_l, P_l2, aii, velAdv, forcesAdv, forcesP, diiF, diiB, sumDij, diiSum;
    DevBuf redPartial, redOut;
    DevBuf errWord; // set by the device-side consistency guard of the scans (GridView::err)
    DevBuf hitBuf, hitCounts; // hit lists shared by the density and force kernels of a step
    // wall-particle deferral (nrs_kernels_tiled.h): static near-boundary bit per cell, this step's wall list
    DevBuf nearBits, wallList, wallTile, wallTileOffset, wallGroupTotal, wallGroupPrefix, wallScalars, wallMask;
    bool nearBitsValid = false;
    bool wallListed = false; // this step's gathers run with wall workgroups
    bool deferWalls() const
    {
        return !(cfg.flags & NRS_FLAG_NO_WALL_WORKGROUPS) && nearBitsValid && nb != 0 && (!iisph() || iisph_lists()) && !refOrder() && lists_ok();
    }
    WallList wall_view() const { return WallList{nearBits.as<uint32_t>(), hashCur, wallList.as<uint32_t>(), wallScalars.as<uint32_t>() + 1, wallMask.as<unsigned long long>()}; }
    // this step's wall list: tile counts (reorder kernel) -> two-level scan (the re-sort's scan kernel) -> stable compaction
    int build_wall_list(uint32_t N)
    {
        const uint32_t nTiles = nblocks(N), nGroups = (nTiles + RESORT_GROUP - 1) / RESORT_GROUP;
        const WallList wl = wall_view(); // (the tile counts were left by the reorder kernel of this step)
        uint32_t *sc = wallScalars.as<uint32_t>();
        const ResortScan a = {wallTile.as<uint32_t>(), wallTileOffset.as<uint32_t>(), wallGroupTotal.as<uint32_t>(), wallGroupPrefix.as<uint32_t>(), sc + 1};
        const ResortScan none = {nullptr, nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(k_resort_scan_tiles, dim3(nGroups), dim3(RESORT_GROUP), 0, stream, a, none, sc, (volatile uint64_t *)nullptr, 0u, nTiles);
        hipLaunchKernelGGL(k_wall_compact, dim3(nTiles), dim3(BLOCK), 0, stream, wl, wallTileOffset.as<uint32_t>(), wallGroupPrefix.as<uint32_t>(),
                           (uint32_t)RESORT_GROUP, wallList.as<uint32_t>(), N);
        HIPCHK(hipGetLastError());
        return NRS_OK;
    }
    DevBuf fastQ;             // NRS_FLAG_FAST_ARITH: (p/rho^2, 1/rho) per sorted slot, density kernel -> force kernel
    // LDS-staged density scan (nrs_kernels_staged.h): fp32 SESPH on power-of-two grids.  Measured at 10 M particles it is
    // SLOWER than the global-memory scan in the exact arithmetic (0.84 vs 0.71 ms: the kernel is bound by vector-instruction
    // issue, not by the latency the staging removes), and since the quantised scan (0.52 ms) also slower than the
    // exact path in its own fast arithmetic (0.70-0.88 ms): it runs only when NRS_FLAG_STAGED_SCAN asks for it.
    bool stagedScan() const
    {
        if (!(cfg.flags & NRS_FLAG_STAGED_SCAN) || !std::is_same<R, float>::value || iisph() || refOrder() || P.numCells > (1u << 30)) return false;
        return KSET == KS_MULLER && lists_ok();
    }
    // fast arithmetic (reciprocals, rsq, fused multiply-adds) in the FORCE walk: fp32 Muller SESPH on the production kernels with
    // shared lists; the density kernel (exact) leaves the (p/rho^2, 1/rho) pairs it needs; everything else keeps IEEE arithmetic
    bool fastArith() const
    {
        return (cfg.flags & NRS_FLAG_FAST_ARITH) && std::is_same<R, float>::value && KSET == KS_MULLER && !iisph() && !refOrder() && lists_ok();
    }
    // coherent re-sort (nrs_kernels_resort.h)
    DevBuf rsMovers, rsMoversAlt, rsStayers, rsMerged, rsTileMovers, rsTileOffset, rsGroupTotal, rsGroupPrefix, rsScalars, rsPrevPacked;
    bool slotOrderValid = false; // posA/velA are in the slot order of hashCur (a full fused step was the last thing that happened)
    uint32_t *packKeys = nullptr, *packVals = nullptr; // where slab_pack / slab_unpack write the next step's keys / values
    uint64_t *rsHostTotal = nullptr, *rsHostTotalDev = nullptr; // (launch number << 32 | mover count), written by
                                                                // k_resort_scan_tiles into pinned, mapped host memory
    uint32_t rsSeq = 0;
    hipEvent_t rsEvent = nullptr;
    bool rsPending = false; // movers/stayers of the keys in hashNext have been split; the count is on its way
    bool splitClearedCells = false; // this step's k_resort_split also reset the cell table
    uint64_t rsSteps = 0, rsFallbacks = 0;
    double lastMovers = -1.0; // mover count of the last coherent re-sort
    bool few_movers(uint64_t M, uint64_t N) const { return M * 100ull <= N * (uint64_t)RESORT_MAX_MOVER_PCT; }
    // slab decomposition
    bool slabOn = false;
    SlabCfg slab = {INT_MIN / 2, INT_MAX / 2, 2};
    DevBuf ghostPos, ghostVel, slabCounts, slabTotals;
    uint64_t nOwned = 0;
    uint32_t ghostCount = 0;
    bool cellsClean = false; // cellStart is all-EMPTY
    bool packedHashValid = false;
    bool packResort = false; // this pack compacted the previous sorted keys next to the new ones
    uint32_t packChanged = 0; // ... and counted the owned particles that stay but changed cell
    bool packInplace = false; // the last pack left the owned particles where they were (holesPending until the next reorder)
    // fused classification: the force kernel of the last step already wrote stream flags / counts / dead marks for these cuts
    DevBuf slabFlags;
    bool classifiedValid = false;
    uint32_t classifiedN = 0;
    // slab runs, in-place partition: the owned particles are not compacted; dead slots carry the key 0xffffffff
    DevBuf rsTileDead, rsTileDeadOffset, rsGroupDeadTotal, rsGroupDeadPrefix;
    hipEvent_t packEvent = nullptr;
    uint32_t *slabHostTotals = nullptr; // page-locked landing place of the stream totals
    bool holesPending = false; // posA/velA[0, physN) contain dead slots (keys in hashNext tell which); n counts live ones
    uint32_t physN = 0;        // physical extent of the arrays while holesPending
    bool rsTilesDirty = false; // rsTileMovers holds counts no scan has consumed
    int clean_tile_counts()
    {
        if (rsTilesDirty) {
            HIPCHK(hipMemsetAsync(rsTileMovers.p, 0, rsTileMovers.bytes, stream));
            HIPCHK(hipMemsetAsync(rsTileDead.p, 0, rsTileDead.bytes, stream));
        }
        rsTilesDirty = false;
        return NRS_OK;
    }
    bool rsCountKnown = false; // the mover count of the pending split is already on the host (slab runs)
    uint32_t rsKnownCount = 0;
    bool fusedThisStep = false;
    // profiling
    struct Ev { int stage; hipEvent_t a, b; bool cont; };
    std::vector<Ev> evPool;
    size_t evUsed = 0;
    float stageMs[NRS_STAGE_COUNT] = {0};
    uint32_t stageLaunches[NRS_STAGE_COUNT] = {0};
    bool evOpen = false;

    // ---- the state of the particle arrays and of the keys prepared for the next step --------------------------------
    // The fields above are not independent: they encode ONE of the states below (DESIGN.md §5 has the transition table).
    // Every public entry point calls validate() first, so a sequence of calls that would leave them inconsistent returns
    // NRS_E_STATE instead of handing a wrong count or a stale table to a kernel (the GPU memory fault of round 1 was exactly
    // that: a merge sized with a mover count that had been reset before it was read).
    enum ArrayState {
        AS_FRESH,        // arrays compact, any order; the next step hashes and sorts from scratch
        AS_KEYS_READY,   // + hashNext/indexNext hold the next step's keys/values (fused kernel, or slab pack/unpack)
        AS_SPLIT_QUEUED, // + their movers/stayers split is queued (coherent re-sort); the count is pending or known
        AS_SLOT_ORDER,   // slab run after a fused step: arrays in the slot order of hashCur, keys per slot, to be re-partitioned
        AS_HOLES,        // slab in-place partition: arrays [0, physN) with dead slots, split queued, count known
        AS_INVALID
    };
    ArrayState array_state() const
    {
        if (n > cap || (slabOn && nOwned > n)) return AS_INVALID;
        if (!slabOn && (holesPending || classifiedValid)) return AS_INVALID;
        if (hashReady && (!hashNext || !indexNext)) return AS_INVALID;
        if (rsPending && (!hashReady || !rsMovers.p)) return AS_INVALID;
        if (rsCountKnown && !rsPending) return AS_INVALID;
        if (classifiedValid && (!slotOrderValid || !hashCur || !hashNext)) return AS_INVALID;
        if (slotOrderValid && (!hashCur || !hashNext)) return AS_INVALID;
        if (holesPending) {
            if (!(packInplace && hashReady && rsPending && rsCountKnown) || physN < n || physN > cap || rsKnownCount > physN) return AS_INVALID;
            return AS_HOLES;
        }
        if (rsCountKnown && rsKnownCount > n) return AS_INVALID;
        if (rsPending) return AS_SPLIT_QUEUED;
        if (hashReady) return AS_KEYS_READY;
        if (slotOrderValid) return AS_SLOT_ORDER;
        return AS_FRESH;
    }
    // Transitions.  Every write to the coupled fields above goes through one of these (round 3: the fields used to be set one by one
    // at ~30 places, which is how a count could be reset before it was read); array_state() reads the result back as ONE state.
    void drop_prepared_keys() { hashReady = false; rsPending = false; rsCountKnown = false; }   // the keys were consumed, or are void
    void to_fresh() { drop_prepared_keys(); slotOrderValid = false; classifiedValid = false; } // -> AS_FRESH: compact arrays, any order
    void keys_ready(uint32_t *h, uint32_t *i) { hashNext = h; indexNext = i; hashReady = true; } // -> AS_KEYS_READY
    void split_queued() { rsPending = true; }                                                    // -> AS_SPLIT_QUEUED, count still on the device
    void split_queued_known(uint32_t movers) { rsPending = true; rsCountKnown = true; rsKnownCount = movers; } // ..., count on the host
    void to_holes(uint32_t extent, uint32_t movers)                                             // -> AS_HOLES (in-place slab partition)
    {
        holesPending = true; physN = extent; packedHashValid = true; hashReady = true;
        split_queued_known(movers);
    }
    int validate(const char *where) const
    {
        if (array_state() != AS_INVALID) return NRS_OK;
        char buf[320];
        snprintf(buf, sizeof(buf), "internal state inconsistent at %s (n %llu cap %llu physN %u owned %llu | hashReady %d rsPending %d countKnown %d "
                 "known %u holes %d inplace %d classified %d slotOrder %d slab %d)", where, (unsigned long long)n, (unsigned long long)cap, physN,
                 (unsigned long long)nOwned, hashReady, rsPending, rsCountKnown, rsKnownCount, holesPending, packInplace, classifiedValid,
                 slotOrderValid, slabOn);
        return fail(NRS_E_STATE, buf);
    }

    // the tiled kernels assume the power-of-two grids the reference's hash assumes (sph_kernel_impl.cuh:120)
    bool refOverride = false; // this IISPH step is being repeated with the reference-order kernels (iisph_tail)
    bool refOrder() const
    {
        return refOverride || (cfg.flags & NRS_FLAG_REFERENCE_ORDER) != 0 || !is_pow2(P.gridSize[0]) || !is_pow2(P.gridSize[1]) ||
               !is_pow2(P.gridSize[2]);
    }
    bool iisph() const { return cfg.solver == NRS_SOLVER_IISPH; }

    ~Ctx() override
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto &e : evPool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        DevBuf *all[] = {&posA, &posB, &velA, &velB, &presA, &presB, &dens, &forces, &hashA, &hashB, &indexA, &indexB,
                         &inv, &sortTmp, &cellStart, &cellEnd, &bCellStart, &bCellEnd, &bSorted, &bHash, &bIndex,
                         &bHashAlt, &bIndexAlt, &densAdv, &densCorr, &P_l, &P_l2, &aii, &velAdv, &forcesAdv, &forcesP,
                         &diiF, &diiB, &sumDij, &diiSum, &redPartial, &redOut, &errWord, &hitBuf, &hitCounts, &qpos, &gatherPos, &gatherVel, &fastQ, &nearBits, &wallList, &wallMask, &wallTile, &wallTileOffset, &wallGroupTotal, &wallGroupPrefix, &wallScalars, &ghostPos, &ghostVel, &slabCounts, &slabTotals,
                         &rsMovers, &rsMoversAlt, &rsStayers, &rsMerged, &rsTileMovers, &rsTileOffset, &rsGroupTotal, &rsGroupPrefix, &rsScalars, &rsPrevPacked,
                         &rsTileDead, &rsTileDeadOffset, &rsGroupDeadTotal, &rsGroupDeadPrefix, &slabFlags};
        for (DevBuf *b : all) b->release();
        if (rsEvent) (void)hipEventDestroy(rsEvent);
        if (packEvent) (void)hipEventDestroy(packEvent);
        if (slabHostTotals) (void)hipHostFree(slabHostTotals);
        if (rsHostTotal) (void)hipHostFree(rsHostTotal);
        snapshot_release();
        if (ownStream && stream) (void)hipStreamDestroy(stream);
    }

    int alloc_cells()
    {
        const uint64_t C = P.numCells;
        if (C == 0 || C > (1ull << 31)) return fail(NRS_E_INVALID, "numCells out of range");
        NRSCHK(cellStart.alloc(C * 4));
        NRSCHK(cellEnd.alloc(C * 4));
        if (nb) {
            NRSCHK(bCellStart.alloc(C * 4));
            NRSCHK(bCellEnd.alloc(C * 4));
        }
        if (cellsAllocated != C) {
            cellsClean = false;
            HIPCHK(hipMemsetAsync(cellEnd.p, 0, C * 4, stream));
            if (nb) HIPCHK(hipMemsetAsync(bCellEnd.p, 0, C * 4, stream));
            cellsAllocated = (uint32_t)C;
        }
        return NRS_OK;
    }

    int init(const nrs_config &c, const void *params) override
    {
        cfg = c;
        cap = c.capacity;
        if (cap == 0 || cap > (uint64_t)HIT_INDEX) return fail(NRS_E_INVALID, "capacity must be in 1..2^27-1");
        std::memcpy(&PU, params, sizeof(PU));
        derive_kernel_params();
        const size_t v = sizeof(T4) * cap, s = sizeof(R) * cap, u = 4 * cap;
        NRSCHK(posA.alloc(v)); NRSCHK(posB.alloc(v)); NRSCHK(velA.alloc(v)); NRSCHK(velB.alloc(v));
        NRSCHK(presA.alloc(s)); NRSCHK(presB.alloc(s)); NRSCHK(dens.alloc(s)); NRSCHK(forces.alloc(v));
        NRSCHK(hashA.alloc(u)); NRSCHK(hashB.alloc(u)); NRSCHK(indexA.alloc(u)); NRSCHK(indexB.alloc(u));
        HIPCHK(hipMemsetAsync(presA.p, 0, s, stream));
        HIPCHK(hipMemsetAsync(presB.p, 0, s, stream));
        HIPCHK(hipMemsetAsync(dens.p, 0, s, stream));
        HIPCHK(hipMemsetAsync(forces.p, 0, v, stream));
        if (iisph()) {
            NRSCHK(inv.alloc(u));
            NRSCHK(densAdv.alloc(s)); NRSCHK(densCorr.alloc(s)); NRSCHK(P_l.alloc(s)); NRSCHK(P_l2.alloc(s));
            NRSCHK(aii.alloc(s));
            NRSCHK(velAdv.alloc(v)); NRSCHK(forcesAdv.alloc(v)); NRSCHK(forcesP.alloc(v));
            NRSCHK(diiF.alloc(v)); NRSCHK(diiB.alloc(v)); NRSCHK(sumDij.alloc(v)); NRSCHK(diiSum.alloc(v));
            DevBuf *z[] = {&densAdv, &densCorr, &P_l, &P_l2, &aii, &velAdv, &forcesAdv, &forcesP, &diiF, &diiB, &sumDij};
            for (DevBuf *b : z) HIPCHK(hipMemsetAsync(b->p, 0, b->bytes, stream));
        }
        // SESPH: density → forces; IISPH (Muller kernels only: the Monaghan support is 2h): one scan feeds the chain
        if ((!iisph() || KSET == KS_MULLER) && !(cfg.flags & (NRS_FLAG_REFERENCE_ORDER | NRS_FLAG_NO_SHARED_LISTS))) {
            NRSCHK(hitBuf.alloc((size_t)HIT_CAP * cap * 4));
            NRSCHK(hitCounts.alloc((size_t)cap * 4));
            if (NRS_COMPACT_SCAN) NRSCHK(qpos.alloc(((size_t)cap + 4) * QP_BYTES));
            if (NRS_FORCE_PAIRS && !iisph()) {
                if (NRS_GATHER_INTERLEAVED) NRSCHK(gatherPos.alloc((size_t)cap * 2 * sizeof(T4)));
                else { NRSCHK(gatherPos.alloc((size_t)cap * sizeof(T4))); NRSCHK(gatherVel.alloc((size_t)cap * sizeof(T4))); }
            }
            if ((cfg.flags & NRS_FLAG_FAST_ARITH) && !iisph() && std::is_same<R, float>::value && KSET == KS_MULLER)
                NRSCHK(fastQ.alloc((size_t)cap * sizeof(FastPair)));
        }
        NRSCHK(errWord.alloc(8)); // [0] run guard of the scans, [1] IISPH: a gathered value went non-finite (IisphArrays::nonFinite)
        HIPCHK(hipMemsetAsync(errWord.p, 0, 8, stream));
        NRSCHK(redPartial.alloc(sizeof(double) * 1024));
        NRSCHK(redOut.alloc(2 * sizeof(double)));
        // radix sort workspace for the largest problem
        size_t tmp = 0;
        rocprim::double_buffer<uint32_t> k(hashA.as<uint32_t>(), hashB.as<uint32_t>());
        rocprim::double_buffer<uint32_t> vv(indexA.as<uint32_t>(), indexB.as<uint32_t>());
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp, k, vv, (size_t)cap, 0u, 32u, stream));
        size_t tmp9 = 0;
        HIPCHK(rocprim::radix_sort_pairs<SortCfg9>(nullptr, tmp9, k, vv, (size_t)cap, 0u, 27u, stream));
        size_t tmp10 = 0;
        HIPCHK(rocprim::radix_sort_pairs<SortCfg10>(nullptr, tmp10, k, vv, (size_t)cap, 0u, 30u, stream));
        size_t tmpAll = std::max(tmp, std::max(tmp9, tmp10));
        // coherent re-sort: SESPH steps on the production kernels re-use the previous step's order
        if (!(cfg.flags & (NRS_FLAG_REFERENCE_ORDER | NRS_FLAG_NO_FUSION | NRS_FLAG_FULL_SORT)) && cap >= RESORT_MIN_PARTICLES) {
            const size_t nTiles = (cap + BLOCK - 1) / BLOCK, nGroups = (nTiles + RESORT_GROUP - 1) / RESORT_GROUP;
            const size_t mcap = cap; // any share of the particles may be movers (see rsMaxPct)
            NRSCHK(rsMovers.alloc(8 * cap)); NRSCHK(rsMoversAlt.alloc(8 * mcap)); NRSCHK(rsStayers.alloc(8 * cap)); NRSCHK(rsMerged.alloc(8 * cap));
            NRSCHK(rsTileMovers.alloc(4 * nTiles)); NRSCHK(rsTileOffset.alloc(4 * nTiles));
            NRSCHK(rsGroupTotal.alloc(4 * nGroups)); NRSCHK(rsGroupPrefix.alloc(4 * nGroups)); NRSCHK(rsScalars.alloc(16));
            NRSCHK(rsPrevPacked.alloc(4 * cap));
            NRSCHK(rsTileDead.alloc(4 * nTiles)); NRSCHK(rsTileDeadOffset.alloc(4 * nTiles));
            NRSCHK(rsGroupDeadTotal.alloc(4 * nGroups)); NRSCHK(rsGroupDeadPrefix.alloc(4 * nGroups));
            HIPCHK(hipMemsetAsync(rsTileDead.p, 0, 4 * nTiles, stream));
            HIPCHK(hipEventCreateWithFlags(&packEvent, hipEventDisableTiming));
            HIPCHK(hipMemsetAsync(rsTileMovers.p, 0, 4 * nTiles, stream));
            HIPCHK(hipMemsetAsync(rsScalars.p, 0, 16, stream));
            HIPCHK(hipHostMalloc((void **)&rsHostTotal, 64, hipHostMallocMapped));
            std::memset(rsHostTotal, 0, 64);
            HIPCHK(hipHostGetDevicePointer((void **)&rsHostTotalDev, rsHostTotal, 0));
            HIPCHK(hipEventCreateWithFlags(&rsEvent, hipEventDisableTiming));
            rocprim::double_buffer<uint64_t> mk(rsMovers.as<uint64_t>(), rsMoversAlt.as<uint64_t>());
            for (unsigned bits : {24u, 27u, 30u}) {
                size_t t = 0;
                HIPCHK(sort_movers(nullptr, t, mk, mcap, bits, stream));
                tmpAll = std::max(tmpAll, t);
            }
            size_t t = 0;
            HIPCHK(rocprim::merge(nullptr, t, rsStayers.as<uint64_t>(), rsMovers.as<uint64_t>(), rsMerged.as<uint64_t>(), (size_t)cap, mcap,
                                  rocprim::less<uint64_t>(), stream));
            tmpAll = std::max(tmpAll, t);
        }
        NRSCHK(sortTmp.alloc(tmpAll));
        NRSCHK(alloc_cells());
        return NRS_OK;
    }

    // a host-driven IISPH step (nrs_iisph_predict .. nrs_iisph_finish) holds hit lists, factors and a halo budget that belong to the
    // arrays, grid and cuts it was predicted on: everything that would change those is refused until it is finished (or abandoned
    // by uploading particles)
    int refuse_mid_iisph(const char *what) const
    {
        if (!iisphPhase) return NRS_OK;
        char buf[200];
        snprintf(buf, sizeof(buf), "%s while a host-driven IISPH step is in progress (nrs_iisph_finish first, or upload particles to abandon it)", what);
        return fail(NRS_E_STATE, buf);
    }
    int set_params(const void *params) override
    {
        NRSCHK(refuse_mid_iisph("nrs_set_params"));
        Params<R> q;
        std::memcpy(&q, params, sizeof(q));
        // the keys the fused force kernel left for the next step depend on the grid only (a new time step or viscosity
        // does not invalidate them: the reference calls setParameters every update(), the CFL variant with a new dt)
        const bool sameGrid = std::memcmp(PU.gridSize, q.gridSize, sizeof(PU.gridSize)) == 0 && q.numCells == PU.numCells &&
                              std::memcmp(PU.worldOrigin, q.worldOrigin, sizeof(PU.worldOrigin)) == 0 &&
                              std::memcmp(PU.cellSize, q.cellSize, sizeof(PU.cellSize)) == 0;
        if (!sameGrid) NRSCHK(invalidate_grid_state());
        const uint32_t cellsBefore = P.numCells;
        PU = q;
        if (!sameGrid && slabOn) choose_window(slab.lo, slab.hi, slab.halo, true);
        derive_kernel_params();
        const bool regrid = P.numCells != cellsBefore;
        if (regrid) NRSCHK(alloc_cells());
        if (!sameGrid && nb) NRSCHK(rebuild_boundary_tables()); // the boundary hashes / cell table depend on origin, cell size and extents
        return NRS_OK;
    }
    // The grid (origin, cell size or extents) is about to change: every key computed for the old grid is void — the
    // keys the fused kernel prepared for the next step, the split of the coherent re-sort, the slot order, a slab
    // classification — and the cell table has to be reset in full (k_clear_cells undoes only cells of the OLD keys).
    int invalidate_grid_state()
    {
        NRSCHK(compact_holes());
        to_fresh();
        packedHashValid = false; cellsClean = false;
        return NRS_OK;
    }
    int get_params(void *params) override
    {
        std::memcpy(params, &PU, sizeof(PU));
        return NRS_OK;
    }

    int upload(const void *pos4, const void *vel4, const void *pres, uint64_t first, uint64_t count) override
    {
        NRSCHK(validate("nrs_upload_particles"));
        if (first + count > cap) return fail(NRS_E_CAPACITY, "upload exceeds capacity");
        NRSCHK(compact_holes());
        if (count) {
            if (!pos4) return fail(NRS_E_INVALID, "pos4 is NULL");
            HIPCHK(hipMemcpyAsync(posA.as<T4>() + first, pos4, sizeof(T4) * count, hipMemcpyHostToDevice, stream));
            if (vel4) HIPCHK(hipMemcpyAsync(velA.as<T4>() + first, vel4, sizeof(T4) * count, hipMemcpyHostToDevice, stream));
            else HIPCHK(hipMemsetAsync(velA.as<T4>() + first, 0, sizeof(T4) * count, stream));
            if (pres) HIPCHK(hipMemcpyAsync(presA.as<R>() + first, pres, sizeof(R) * count, hipMemcpyHostToDevice, stream));
            else HIPCHK(hipMemsetAsync(presA.as<R>() + first, 0, sizeof(R) * count, stream));
            HIPCHK(hipStreamSynchronize(stream)); // the caller may reuse its host buffers on return
        }
        if (first + count > n) n = first + count;
        if (slabOn) nOwned = n; // (until the next partition says otherwise)
        midStep = false;
        iisphPhase = 0; iisphIter = 0; // new particles abandon a host-driven IISPH step that was in progress
        to_fresh();
        return NRS_OK;
    }
    int set_n(uint64_t nn) override
    {
        NRSCHK(validate("nrs_set_num_particles"));
        if (nn > cap) return fail(NRS_E_CAPACITY, "n exceeds capacity");
        NRSCHK(compact_holes());
        if (nn != n) to_fresh();
        if (nn != n) { iisphPhase = 0; iisphIter = 0; } // (the hit lists of a predicted step belong to the old particle set)
        n = nn;
        if (slabOn) nOwned = n;
        return NRS_OK;
    }
    uint64_t get_n() override { return n; }

    // ---- boundaries: SPH::updateGpuBoundaries / updateGrid (sph/sph.cpp:313-337, 391-432) ----------
    std::vector<T4> hostBi;
    std::vector<R> hostVbi;

    uint32_t sort_end_bit() const
    {
        uint32_t bits = 1;
        while (bits < 32 && (1ull << bits) < (uint64_t)P.numCells) ++bits;
        return bits;
    }

    int rebuild_boundary_tables()
    {
        if (!nb) return NRS_OK;
        NRSCHK(alloc_cells());
        DevBuf dBi, dVbi;
        NRSCHK(dBi.alloc(sizeof(T4) * nb));
        NRSCHK(dVbi.alloc(sizeof(R) * nb));
        HIPCHK(hipMemcpyAsync(dBi.p, hostBi.data(), sizeof(T4) * nb, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(dVbi.p, hostVbi.data(), sizeof(R) * nb, hipMemcpyHostToDevice, stream));
        NRSCHK(bHash.alloc(4 * nb)); NRSCHK(bIndex.alloc(4 * nb)); NRSCHK(bHashAlt.alloc(4 * nb)); NRSCHK(bIndexAlt.alloc(4 * nb));
        NRSCHK(bSorted.alloc(sizeof(T4) * nb));
        hipLaunchKernelGGL((k_hash<R>), dim3(nblocks(nb)), dim3(BLOCK), 0, stream, P, dBi.as<T4>(), bHash.as<uint32_t>(),
                           bIndex.as<uint32_t>(), (uint32_t)nb);
        size_t tmp = 0;
        rocprim::double_buffer<uint32_t> k(bHash.as<uint32_t>(), bHashAlt.as<uint32_t>());
        rocprim::double_buffer<uint32_t> v(bIndex.as<uint32_t>(), bIndexAlt.as<uint32_t>());
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp, k, v, (size_t)nb, 0u, sort_end_bit(), stream));
        DevBuf t;
        NRSCHK(t.alloc(tmp));
        HIPCHK(rocprim::radix_sort_pairs(t.p, tmp, k, v, (size_t)nb, 0u, sort_end_bit(), stream));
        bHashCur = k.current();
        bIndexCur = v.current();
        HIPCHK(hipMemsetAsync(bCellStart.p, 0xff, (size_t)P.numCells * 4, stream));
        hipLaunchKernelGGL((k_reorder_boundary<R>), dim3(nblocks(nb)), dim3(BLOCK), 0, stream, bHashCur, bIndexCur,
                           dBi.as<T4>(), dVbi.as<R>(), bSorted.as<T4>(), bCellStart.as<uint32_t>(),
                           bCellEnd.as<uint32_t>(), (uint32_t)nb);
        nearBitsValid = false;
        if ((!iisph() || KSET == KS_MULLER) && is_pow2(P.gridSize[0]) && is_pow2(P.gridSize[1]) && is_pow2(P.gridSize[2])) {
            const size_t words = ((size_t)P.numCells + 31) / 32;
            NRSCHK(nearBits.alloc(words * 4));
            const size_t nTiles = (cap + BLOCK - 1) / BLOCK, nGroups = (nTiles + RESORT_GROUP - 1) / RESORT_GROUP;
            NRSCHK(wallList.alloc((size_t)cap * 4));
            NRSCHK(wallTile.alloc(nTiles * 4)); NRSCHK(wallTileOffset.alloc(nTiles * 4));
            NRSCHK(wallMask.alloc(nTiles * 4 * 8));
            NRSCHK(wallGroupTotal.alloc(nGroups * 4)); NRSCHK(wallGroupPrefix.alloc(nGroups * 4)); NRSCHK(wallScalars.alloc(16));
            HIPCHK(hipMemsetAsync(wallScalars.p, 0, 16, stream));
            HIPCHK(hipMemsetAsync(nearBits.p, 0, words * 4, stream));
            hipLaunchKernelGGL((k_mark_near_boundary<R>), dim3(nblocks(nb)), dim3(BLOCK), 0, stream, P, bHashCur, (uint32_t)nb, nearBits.as<uint32_t>());
            nearBitsValid = true;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        t.release(); dBi.release(); dVbi.release();
        return NRS_OK;
    }

    int set_boundaries(const void *bi4, const void *vbi, uint64_t nbNew, int update_grid) override
    {
        NRSCHK(refuse_mid_iisph("nrs_set_boundaries"));
        if (nbNew > (uint64_t)HIT_INDEX) return fail(NRS_E_INVALID, "too many boundary particles (max 2^27-1)");
        if (nbNew && (!bi4 || !vbi)) return fail(NRS_E_INVALID, "bi4/vbi is NULL");
        nb = nbNew;
        hostBi.assign((const T4 *)bi4, (const T4 *)bi4 + nb);
        hostVbi.assign((const R *)vbi, (const R *)vbi + nb);
        if (!nb) { nearBitsValid = false; return NRS_OK; }
        if (update_grid) {
            NRSCHK(invalidate_grid_state());
            // BBMin/BBMax (sph_cuda.cu:461-505) + SPH::updateGrid (sph.cpp:313-337)
            R mn[3] = {hostBi[0].x, hostBi[0].y, hostBi[0].z}, mx[3] = {hostBi[0].x, hostBi[0].y, hostBi[0].z};
            for (uint64_t i = 1; i < nb; ++i) {
                const R c[3] = {hostBi[i].x, hostBi[i].y, hostBi[i].z};
                for (int a = 0; a < 3; ++a) {
                    if (c[a] < mn[a]) mn[a] = c[a];
                    if (mx[a] < c[a]) mx[a] = c[a];
                }
            }
            uint32_t g[3];
            for (int a = 0; a < 3; ++a) {
                PU.worldOrigin[a] = (R)(mn[a] - 0.1);
                const uint32_t sz = (uint32_t)std::ceil((mx[a] - mn[a] + 0.1) / PU.interactionRadius);
                g[a] = next_pow2(sz);
            }
            const uint64_t C = (uint64_t)g[0] * g[1] * g[2];
            if (C > (1ull << 31)) return fail(NRS_E_INVALID, "grid from boundary AABB exceeds 2^31 cells");
            PU.gridSize[0] = g[0]; PU.gridSize[1] = g[1]; PU.gridSize[2] = g[2];
            PU.numCells = (uint32_t)C;
            if (slabOn) choose_window(slab.lo, slab.hi, slab.halo, true);
            derive_kernel_params();
        }
        return rebuild_boundary_tables();
    }

    // ---- profiling helpers ------------------------------------------------------------------------
    // cont: second part of a stage whose first part ran earlier (time is added, the launch count is not)
    int ev_begin(int stage, bool cont = false)
    {
        evOpen = (profMask >> stage) & 1u;
        if (!evOpen) return NRS_OK;
        if (evUsed == evPool.size()) {
            Ev e; e.stage = stage;
            HIPCHK(hipEventCreate(&e.a));
            HIPCHK(hipEventCreate(&e.b));
            evPool.push_back(e);
        }
        evPool[evUsed].stage = stage;
        evPool[evUsed].cont = cont;
        HIPCHK(hipEventRecord(evPool[evUsed].a, stream));
        return NRS_OK;
    }
    int ev_end()
    {
        if (!evOpen) return NRS_OK;
        evOpen = false;
        HIPCHK(hipEventRecord(evPool[evUsed].b, stream));
        ++evUsed;
        return NRS_OK;
    }
    int set_profiling(uint32_t mask) override
    {
        NRSCHK(ev_collect());
        profMask = mask;
        std::memset(stageMs, 0, sizeof(stageMs));
        std::memset(stageLaunches, 0, sizeof(stageLaunches));
        return NRS_OK;
    }
    int ev_collect()
    {
        if (!evUsed) return NRS_OK;
        HIPCHK(hipStreamSynchronize(stream));
        for (size_t i = 0; i < evUsed; ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, evPool[i].a, evPool[i].b));
            stageMs[evPool[i].stage] += ms;
            stageLaunches[evPool[i].stage] += evPool[i].cont ? 0 : 1;
        }
        evUsed = 0;
        return NRS_OK;
    }
    int stage_ms(int stage, float *ms, uint32_t *launches) override
    {
        if (stage < 0 || stage >= NRS_STAGE_COUNT) return fail(NRS_E_INVALID, "bad stage");
        NRSCHK(ev_collect()); // resolves the pending event pairs (synchronizes the stream)
        *ms = stageMs[stage];
        if (launches) *launches = stageLaunches[stage];
        return NRS_OK;
    }

    GridView<R> grid_view() const
    {
        GridView<R> G;
        G.cellStart = cellStart.as<uint32_t>(); G.cellEnd = cellEnd.as<uint32_t>();
        G.bCellStart = bCellStart.as<uint32_t>(); G.bCellEnd = bCellEnd.as<uint32_t>();
        G.sB = bSorted.as<T4>();
        G.actLo = INT_MIN;
        G.actHi = INT_MAX;
        G.nSorted = (uint32_t)n;
        G.err = errWord.as<uint32_t>();
        G.qpos = lists_ok() ? qpos.as<qword_t>() : (const qword_t *)nullptr;
        G.qT = qT;
        G.qc = qc;
        return G;
    }
    IisphArrays<R> iisph_view() const
    {
        IisphArrays<R> I;
        I.densAdv = densAdv.as<R>(); I.densCorr = densCorr.as<R>(); I.P_l = P_l.as<R>(); I.P_l_next = P_l2.as<R>();
        I.aii = aii.as<R>();
        I.velAdv = velAdv.as<T4>(); I.forcesAdv = forcesAdv.as<T4>(); I.forcesP = forcesP.as<T4>();
        I.diiF = diiF.as<T4>(); I.diiB = diiB.as<T4>(); I.sumDij = sumDij.as<T4>(); I.diiSum = diiSum.as<T4>();
        I.inv = inv.as<uint32_t>();
        I.nonFinite = (iisph_lists() && !slabOn) ? errWord.as<uint32_t>() + 1 : (uint32_t *)nullptr; // (second word of the error buffer)
        return I;
    }

    // hash → sort → cell ranges + reorder: common prefix of both solvers
    int stage_prefix(int stop)
    {
        const uint32_t N = (uint32_t)n;
        const dim3 g(nblocks(N)), b(BLOCK);
        if (holesPending) { // in-place slab partition: only the merge path can consume arrays with holes
            const bool canMerge = hashReady && rsPending && rsCountKnown && stop != NRS_STAGE_HASH && stop != NRS_STAGE_SORT &&
                                  few_movers(rsKnownCount, n);
            if (!canMerge) {
                if (hashReady && rsPending && rsCountKnown) { ++rsSteps; ++rsFallbacks; }
                NRSCHK(compact_holes()); // also drops the prepared keys: hash and sort from scratch below
            }
        }
        uint32_t *kIn = hashA.as<uint32_t>(), *kAlt = hashB.as<uint32_t>();
        uint32_t *vIn = indexA.as<uint32_t>(), *vAlt = indexB.as<uint32_t>();
        if (hashReady) { // keys/values of this step were produced by the previous step's fused force kernel
            kIn = hashNext; vIn = indexNext;
            kAlt = (kIn == hashA.as<uint32_t>()) ? hashB.as<uint32_t>() : hashA.as<uint32_t>();
            vAlt = (vIn == indexA.as<uint32_t>()) ? indexB.as<uint32_t>() : indexA.as<uint32_t>();
        } else {
            NRSCHK(ev_begin(NRS_STAGE_HASH));
            hipLaunchKernelGGL((k_hash<R>), g, b, 0, stream, P, posA.as<T4>(), kIn, vIn, N);
            NRSCHK(ev_end());
        }
        const bool resort = hashReady && rsPending && stop != NRS_STAGE_HASH && stop != NRS_STAGE_SORT;
        const bool countKnown = rsCountKnown; // (slab runs: the host already has the mover count)
        drop_prepared_keys(); // (consumed by this step)
        hashCur = kIn; indexCur = vIn;
        if (stop == NRS_STAGE_HASH) return NRS_OK;

        const uint64_t *merged = nullptr;
        NRSCHK(ev_begin(NRS_STAGE_SORT, resort));
        if (resort) {
            // the split of these keys into movers / stayers was queued behind the force kernel; its mover count sizes
            // the mover sort and the merge (see nrs_kernels_resort.h)
            uint32_t M = rsKnownCount;
            if (!countKnown) NRSCHK(wait_mover_count(&M));
            ++rsSteps;
            lastMovers = (double)M;
            if (M > N) return fail(NRS_E_STATE, "coherent re-sort: mover count exceeds the particle count (stale count)");
            if (few_movers(M, N)) {
                if (M == 0) {
                    merged = rsStayers.as<uint64_t>();
                } else {
                    rocprim::double_buffer<uint64_t> mk(rsMovers.as<uint64_t>(), rsMoversAlt.as<uint64_t>());
                    size_t tmp = sortTmp.bytes;
                    HIPCHK(sort_movers(sortTmp.p, tmp, mk, (size_t)M, sort_end_bit(), stream));
                    tmp = sortTmp.bytes;
                    HIPCHK(rocprim::merge(sortTmp.p, tmp, rsStayers.as<uint64_t>(), mk.current(), rsMerged.as<uint64_t>(), (size_t)(N - M),
                                          (size_t)M, rocprim::less<uint64_t>(), stream));
                    merged = rsMerged.as<uint64_t>();
                }
            } else {
                ++rsFallbacks;
            }
        }
        if (!merged) {
            if (!resort) lastMovers = -1.0;
            rocprim::double_buffer<uint32_t> k(kIn, kAlt);
            rocprim::double_buffer<uint32_t> v(vIn, vAlt);
            size_t tmp = sortTmp.bytes;
            HIPCHK(sort_pairs(sortTmp.p, tmp, k, v, (size_t)N, sort_end_bit(), stream));
            hashCur = k.current(); indexCur = v.current();
        } else {
            hashCur = kAlt; indexCur = vAlt; // plain sorted arrays, written by k_reorder_merged below
        }
        NRSCHK(ev_end());
        if (stop == NRS_STAGE_SORT) return NRS_OK;

        NRSCHK(ev_begin(NRS_STAGE_REORDER));
        if (!cellsClean) HIPCHK(hipMemsetAsync(cellStart.p, 0xff, (size_t)P.numCells * 4, stream));
        cellsClean = false;
        holesPending = false; // the gather below reads only live slots
        wallListed = deferWalls() && !stagedScan(); // (wall workgroups read the sorted keys this stage leaves in hashCur)
        if (merged)
            hipLaunchKernelGGL((k_reorder_merged<R>), g, b, 0, stream, merged, hashCur, indexCur, posA.as<T4>(), velA.as<T4>(),
                               iisph() ? presA.as<R>() : (const R *)nullptr, posB.as<T4>(), velB.as<T4>(), presB.as<R>(),
                               cellStart.as<uint32_t>(), cellEnd.as<uint32_t>(), iisph() ? inv.as<uint32_t>() : (uint32_t *)nullptr, N,
                               wallListed ? nearBits.as<uint32_t>() : (const uint32_t *)nullptr, wallTile.as<uint32_t>(), wallMask.as<unsigned long long>(), qc,
                               lists_ok() ? qpos.as<qword_t>() : (qword_t *)nullptr);
        else
            hipLaunchKernelGGL((k_reorder<R>), g, b, 0, stream, hashCur, indexCur, posA.as<T4>(), velA.as<T4>(),
                               iisph() ? presA.as<R>() : (const R *)nullptr, posB.as<T4>(), velB.as<T4>(), presB.as<R>(),
                               cellStart.as<uint32_t>(), cellEnd.as<uint32_t>(), iisph() ? inv.as<uint32_t>() : (uint32_t *)nullptr, N,
                               wallListed ? nearBits.as<uint32_t>() : (const uint32_t *)nullptr, wallTile.as<uint32_t>(), wallMask.as<unsigned long long>(), qc,
                               lists_ok() ? qpos.as<qword_t>() : (qword_t *)nullptr);
        if (iisph() && (cfg.flags & NRS_FLAG_IISPH_SELF_BY_SLOT)) // Q5 off: the pressure kernels skip j == own slot
            hipLaunchKernelGGL(k_identity, g, b, 0, stream, inv.as<uint32_t>(), N);
        NRSCHK(ev_end());
        return NRS_OK;
    }

    ResortScan scan_movers() const
    {
        uint32_t *sc = rsScalars.as<uint32_t>();
        return ResortScan{rsTileMovers.as<uint32_t>(), rsTileOffset.as<uint32_t>(), rsGroupTotal.as<uint32_t>(), rsGroupPrefix.as<uint32_t>(), sc + 1};
    }
    ResortScan scan_dead() const
    {
        uint32_t *sc = rsScalars.as<uint32_t>();
        return ResortScan{rsTileDead.as<uint32_t>(), rsTileDeadOffset.as<uint32_t>(), rsGroupDeadTotal.as<uint32_t>(), rsGroupDeadPrefix.as<uint32_t>(), sc + 2};
    }
    ResortOffsets offsets_movers() const { return ResortOffsets{rsTileOffset.as<uint32_t>(), rsGroupPrefix.as<uint32_t>()}; }
    ResortOffsets offsets_dead() const { return ResortOffsets{rsTileDeadOffset.as<uint32_t>(), rsGroupDeadPrefix.as<uint32_t>()}; }
    int launch_resort_scan(uint32_t nTiles, bool withDead)
    {
        const uint32_t nGroups = (nTiles + RESORT_GROUP - 1) / RESORT_GROUP;
        ResortScan none = {nullptr, nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(k_resort_scan_tiles, dim3(nGroups), dim3(RESORT_GROUP), 0, stream, scan_movers(), withDead ? scan_dead() : none,
                           rsScalars.as<uint32_t>(), (volatile uint64_t *)rsHostTotalDev, ++rsSeq, nTiles);
        HIPCHK(hipEventRecord(rsEvent, stream));
        return NRS_OK;
    }
    // Somebody wants to look at (or re-partition) the particle arrays while they still have the holes of an in-place slab
    // partition: compact them now (stable) and forget the prepared re-sort; the next step hashes and sorts from scratch.
    int compact_holes()
    {
        if (!holesPending) return NRS_OK;
        holesPending = false;
        const uint32_t NP = physN, nTiles = nblocks(NP);
        drop_prepared_keys(); packedHashValid = false;
        if (!NP) return NRS_OK;
        NRSCHK(clean_tile_counts());
        hipLaunchKernelGGL(k_holes_count, dim3(nTiles), dim3(BLOCK), 0, stream, hashNext, rsTileDead.as<uint32_t>(), NP);
        const uint32_t nGroups = (nTiles + RESORT_GROUP - 1) / RESORT_GROUP;
        ResortScan none = {nullptr, nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(k_resort_scan_tiles, dim3(nGroups), dim3(RESORT_GROUP), 0, stream, scan_dead(), none, rsScalars.as<uint32_t>(),
                           (volatile uint64_t *)nullptr, 0u, nTiles);
        hipLaunchKernelGGL((k_holes_compact<R>), dim3(nTiles), dim3(BLOCK), 0, stream, hashNext, offsets_dead(), posA.as<T4>(), velA.as<T4>(),
                           posB.as<T4>(), velB.as<T4>(), NP);
        HIPCHK(hipGetLastError());
        std::swap(posA.p, posB.p);
        std::swap(velA.p, velB.p);
        return NRS_OK;
    }

    // first half of the next step's sort, queued right behind the kernel that produced the keys in hashNext and counted
    // the movers per tile: scan of the tile counts (total to the host) + stable split into movers / stayers
    int queue_resort_split(uint32_t N)
    {
        NRSCHK(ev_begin(NRS_STAGE_SORT));
        const uint32_t nTiles = nblocks(N);
        NRSCHK(launch_resort_scan(nTiles, false));
        const bool clear = (uint64_t)P.numCells > 8ull * n; // the step's cell-table reset rides along (see step())
        hipLaunchKernelGGL((k_resort_split<false>), dim3(nTiles), dim3(BLOCK), 0, stream, hashCur, hashNext, offsets_movers(), offsets_movers(),
                           rsMovers.as<uint64_t>(), rsStayers.as<uint64_t>(), N, clear ? cellStart.as<uint32_t>() : (uint32_t *)nullptr);
        splitClearedCells = clear;
        split_queued();
        NRSCHK(ev_end());
        return NRS_OK;
    }

    template <bool HAS_B> int sesph_tail(int stop)
    {
        const uint32_t N = (uint32_t)n;
        const dim3 g(nblocks(N)), b(BLOCK);
        GridView<R> G = grid_view();
        if (slabOn) { G.actLo = slab.lo - 1; G.actHi = slab.hi + 1; } // density is also needed one cell beyond the cuts
        // the density kernel's hit lists are handed to the force kernel when both run in this call
        HitBuffer hb = {hitBuf.as<uint32_t>(), hitCounts.as<uint32_t>(), (uint32_t)cap};
        if (NRS_FORCE_PAIRS) { hb.gpos = gatherPos.p; hb.gvel = NRS_GATHER_INTERLEAVED ? (void *)(gatherPos.as<T4>() + 1) : gatherVel.p; hb.svel = velB.p; }
        if constexpr (std::is_same<R, float>::value) { if (fastArith() && fastQ.p && !stagedScan()) hb.fast = fastQ.as<FastPair>(); }
        const bool share = !refOrder() && lists_ok() && stop != NRS_STAGE_DENSITY;
        const bool fast = fastArith() && share && fastQ.p;
        if (HAS_B && share && wallListed && !refOrder()) { // (timed with the reorder stage, whose tile counts it finishes: the density stage is its one launch)
            NRSCHK(ev_begin(NRS_STAGE_REORDER, true));
            NRSCHK(build_wall_list(N));
            NRSCHK(ev_end());
        }
        NRSCHK(ev_begin(NRS_STAGE_DENSITY));
        const WallList wv = wall_view();
        bool didStaged = false;
        if constexpr (std::is_same<R, float>::value) {
            if (stagedScan()) {
                launch_density_staged<KSET, HAS_B>(stream, P, G, share ? &hb : (const HitBuffer *)nullptr, fast, posB.as<T4>(), dens.as<R>(),
                                                   presB.as<R>(), fast ? fastQ.as<FastPair>() : (FastPair *)nullptr, N);
                didStaged = true;
            }
        }
        const bool wallsDeferred = !didStaged && !refOrder() && HAS_B && share && wallListed;
        if (didStaged) {
        } else if (refOrder())
            hipLaunchKernelGGL((k_density_ref<R, KSET, HAS_B>), g, b, 0, stream, P, G, posB.as<T4>(), dens.as<R>(), presB.as<R>(), N);
        else
            launch_density_tiled<R, KSET, HAS_B>(stream, P, G, share ? &hb : (const HitBuffer *)nullptr, posB.as<T4>(), dens.as<R>(),
                                                 presB.as<R>(), N, (HAS_B && share && wallListed) ? &wv : (const WallList *)nullptr);
        if (slabOn) { G.actLo = slab.lo; G.actHi = slab.hi; }
        NRSCHK(ev_end());
        if (stop == NRS_STAGE_DENSITY) return NRS_OK;
        // A full step on the production kernels fuses forces + integrate + next-step hash into one launch that
        // writes the new state straight into the A ("current") arrays, which reorder has finished reading.
        const bool fuse = !refOrder() && stop == 0 && !(cfg.flags & NRS_FLAG_NO_FUSION);
        NRSCHK(ev_begin(NRS_STAGE_FORCES));
        if (refOrder()) {
            hipLaunchKernelGGL((k_forces_ref<R, KSET, SURF, HAS_B>), g, b, 0, stream, P, G, posB.as<T4>(), velB.as<T4>(),
                               dens.as<R>(), presB.as<R>(), forces.as<T4>(), N);
        } else if (fuse) {
            FusedOut<R> fo;
            fo.newPos = posA.as<T4>(); fo.newVel = velA.as<T4>();
            fo.hash = (hashCur == hashA.as<uint32_t>()) ? hashB.as<uint32_t>() : hashA.as<uint32_t>();
            fo.index = (indexCur == indexA.as<uint32_t>()) ? indexB.as<uint32_t>() : indexA.as<uint32_t>();
            const bool resort = rsMovers.p && !slabOn && (uint64_t)N >= RESORT_MIN_PARTICLES;
            // slab runs: the next partition's classification rides in the same launch (k_slab_count and most of
            // k_slab_scatter then have nothing left to do)
            const bool classify = rsMovers.p && slabOn && (uint64_t)N >= RESORT_MIN_PARTICLES;
            if (resort || classify) NRSCHK(clean_tile_counts());
            fo.prevHash = (resort || classify) ? hashCur : nullptr;
            fo.tileMovers = (resort || classify) ? rsTileMovers.as<uint32_t>() : nullptr;
            fo.slabFlags = nullptr; fo.slabBlockCounts = nullptr; fo.slabBlocks = 0; fo.tileDead = nullptr; fo.slab = slab;
            classifiedValid = false;
            if (classify) {
                const uint32_t nbk = std::max<uint32_t>(1u, (N + SLAB_TILE - 1) / SLAB_TILE);
                NRSCHK(slabFlags.alloc(cap));
                NRSCHK(slabCounts.alloc((size_t)ST_TOTALS * ((cap_blocks() > nbk) ? cap_blocks() : nbk) * 4));
                HIPCHK(hipMemsetAsync(slabCounts.p, 0, (size_t)ST_TOTALS * nbk * 4, stream));
                fo.slabFlags = slabFlags.as<uint8_t>();
                fo.slabBlockCounts = slabCounts.as<uint32_t>();
                fo.slabBlocks = nbk;
                fo.tileDead = rsTileDead.as<uint32_t>();
                classifiedValid = true;
                classifiedN = N;
                rsTilesDirty = true; // until a pack's scan consumes the tile counts
            }
            bool didFast = false;
            if constexpr (std::is_same<R, float>::value && KSET == KS_MULLER) {
                if (fast) {
                    launch_forces_fast<SURF, HAS_B>(stream, P, G, hb, posB.as<T4>(), velB.as<T4>(), dens.as<R>(), presB.as<R>(),
                                                    fastQ.as<FastPair>(), (T4 *)nullptr, &fo, N);
                    didFast = true;
                }
            }
            if (!didFast)
                launch_forces_tiled<R, KSET, SURF, HAS_B>(stream, P, G, share ? &hb : (const HitBuffer *)nullptr, posB.as<T4>(),
                                                          velB.as<T4>(), dens.as<R>(), presB.as<R>(), (T4 *)nullptr, &fo, N,
                                                          (HAS_B && share && wallsDeferred) ? &wv : (const WallList *)nullptr);
            if (!slabOn) keys_ready(fo.hash, fo.index);
            else { hashNext = fo.hash; indexNext = fo.index; } // a slab run re-partitions the arrays before the next step (AS_SLOT_ORDER)
            fusedThisStep = true;
            if (resort) {
                NRSCHK(ev_end());
                NRSCHK(queue_resort_split(N));
            }
        } else {
            bool didFast = false;
            if constexpr (std::is_same<R, float>::value && KSET == KS_MULLER) {
                if (fast) {
                    launch_forces_fast<SURF, HAS_B>(stream, P, G, hb, posB.as<T4>(), velB.as<T4>(), dens.as<R>(), presB.as<R>(),
                                                    fastQ.as<FastPair>(), forces.as<T4>(), (const FusedOut<float> *)nullptr, N);
                    didFast = true;
                }
            }
            if (!didFast)
                launch_forces_tiled<R, KSET, SURF, HAS_B>(stream, P, G, share ? &hb : (const HitBuffer *)nullptr, posB.as<T4>(),
                                                          velB.as<T4>(), dens.as<R>(), presB.as<R>(), forces.as<T4>(),
                                                          (const FusedOut<R> *)nullptr, N,
                                                          (HAS_B && share && wallsDeferred) ? &wv : (const WallList *)nullptr);
        }
        NRSCHK(ev_end());
        if (stop == NRS_STAGE_FORCES || fuse) return NRS_OK;
        NRSCHK(ev_begin(NRS_STAGE_INTEGRATE));
        hipLaunchKernelGGL((k_integrate<R>), g, b, 0, stream, P, posB.as<T4>(), velB.as<T4>(), forces.as<T4>(), N);
        NRSCHK(ev_end());
        return NRS_OK;
    }

    int reduce_sum(const R *a, uint32_t N, double *out)
    {
        const uint32_t nbk = std::min<uint32_t>(1024u, nblocks(N));
        hipLaunchKernelGGL((k_sum_partial<R>), dim3(nbk), dim3(BLOCK), 0, stream, a, redPartial.as<double>(), N);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(BLOCK), 0, stream, redPartial.as<double>(), redOut.as<double>(), nbk);
        HIPCHK(hipMemcpyAsync(out, redOut.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return NRS_OK;
    }
    int reduce_max(int which, double *out) override
    {
        if (!n) { *out = 0; return NRS_OK; }
        NRSCHK(compact_holes());
        const uint32_t N = (uint32_t)n;
        const uint32_t nbk = std::min<uint32_t>(1024u, nblocks(N));
        if (which == 0)
            hipLaunchKernelGGL((k_max_partial<R, false>), dim3(nbk), dim3(BLOCK), 0, stream, (const void *)dens.p, redPartial.as<double>(), N);
        else
            hipLaunchKernelGGL((k_max_partial<R, true>), dim3(nbk), dim3(BLOCK), 0, stream, (const void *)velA.p, redPartial.as<double>(), N);
        std::vector<double> h(nbk);
        HIPCHK(hipMemcpyAsync(h.data(), redPartial.p, sizeof(double) * nbk, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        double m = h[0];
        for (uint32_t i = 1; i < nbk; ++i) m = std::max(m, h[i]);
        *out = m;
        return NRS_OK;
    }

    // ---- IISPH step in three phases (the single-domain nrs_step runs them back to back; a slab run has the host decide after
    //      every solver iteration, on the sum over ALL ranks: nrs_iisph_predict / _iterate / _finish) -----------------------------
    uint32_t iisphIter = 0;  // solver iterations done in the current step
    int iisphPhase = 0;      // 0 idle, 1 predicted (iterations may follow)
    bool iisph_lists() const { return !refOrder() && lists_ok() && KSET == KS_MULLER; }

    // predictAdvection (sph_cuda.cu:513-697)
    template <bool HAS_B> int iisph_predict(int stop)
    {
        const uint32_t N = (uint32_t)n;
        const dim3 g(nblocks(N)), b(BLOCK);
        const GridView<R> G = grid_view();
        IisphArrays<R> I = iisph_view();
        // one neighbourhood scan per step: its hit lists drive the rest of the chain (nrs_kernels_iisph.h)
        const bool lists = iisph_lists();
        const HitBuffer hb = {hitBuf.as<uint32_t>(), hitCounts.as<uint32_t>(), (uint32_t)cap};
        // (wall workgroups: for the scan and, round 3, for the list kernels with boundary loops — displacement, advection, pressure, pressure force)
        const bool walls = HAS_B && lists && wallListed;
        if (walls) {
            NRSCHK(ev_begin(NRS_STAGE_REORDER, true));
            NRSCHK(build_wall_list(N));
            NRSCHK(ev_end());
        }
        NRSCHK(ev_begin(NRS_STAGE_I_DENSITY));
        const WallList wv = wall_view();
        if (lists) launch_density_wide<R, KSET, HAS_B>(stream, P, G, hb, posB.as<T4>(), dens.as<R>(), N, walls ? &wv : (const WallList *)nullptr);
        else hipLaunchKernelGGL((k_density_ref<R, KSET, HAS_B>), g, b, 0, stream, P, G, posB.as<T4>(), dens.as<R>(), (R *)nullptr, N);
        NRSCHK(ev_end());
        if (stop == NRS_STAGE_I_DENSITY) return NRS_OK;
        NRSCHK(ev_begin(NRS_STAGE_I_DISPLACEMENT));
        const WallList noWalls = {nullptr, nullptr, nullptr, nullptr, nullptr};
        const uint32_t wb = wall_blocks(g.x);
        if (lists && walls) // (wall workgroups + interior workgroups without the boundary code, as the scan: k_pressure_lists)
            hipLaunchKernelGGL((k_displacement_lists<R, KSET, SURF, HAS_B, true>), dim3(g.x + wb), b, 0, stream, P, G, I, hb, posB.as<T4>(), velB.as<T4>(),
                               dens.as<R>(), presB.as<R>(), N, wv, wb);
        else if (lists)
            hipLaunchKernelGGL((k_displacement_lists<R, KSET, SURF, HAS_B>), g, b, 0, stream, P, G, I, hb, posB.as<T4>(), velB.as<T4>(),
                               dens.as<R>(), presB.as<R>(), N, noWalls, 0u);
        else
            hipLaunchKernelGGL((k_displacement_ref<R, KSET, SURF, HAS_B>), g, b, 0, stream, P, G, I, posB.as<T4>(), velB.as<T4>(),
                               dens.as<R>(), presB.as<R>(), N);
        NRSCHK(ev_end());
        if (stop == NRS_STAGE_I_DISPLACEMENT) return NRS_OK;
        NRSCHK(ev_begin(NRS_STAGE_I_ADVECTION));
        if (lists && walls)
            hipLaunchKernelGGL((k_advection_lists<R, KSET, HAS_B, true>), dim3(g.x + wb), b, 0, stream, P, G, I, hb, posB.as<T4>(), velB.as<T4>(),
                               dens.as<R>(), presB.as<R>(), N, wv, wb);
        else if (lists)
            hipLaunchKernelGGL((k_advection_lists<R, KSET, HAS_B>), g, b, 0, stream, P, G, I, hb, posB.as<T4>(), velB.as<T4>(),
                               dens.as<R>(), presB.as<R>(), N, noWalls, 0u);
        else
            hipLaunchKernelGGL((k_advection_ref<R, KSET, HAS_B>), g, b, 0, stream, P, G, I, posB.as<T4>(), velB.as<T4>(),
                               dens.as<R>(), presB.as<R>(), N);
        NRSCHK(ev_end());
        iisphIter = 0;
        return NRS_OK;
    }
    // one relaxed-Jacobi iteration of pressureSolve (sph_cuda.cu:736-823): sum d_ij p_j, pressure update (double-buffered P_l)
    template <bool HAS_B> int iisph_iteration()
    {
        const uint32_t N = (uint32_t)n;
        const dim3 g(nblocks(N)), b(BLOCK);
        const GridView<R> G = grid_view();
        IisphArrays<R> I = iisph_view();
        const bool lists = iisph_lists();
        const HitBuffer hb = {hitBuf.as<uint32_t>(), hitCounts.as<uint32_t>(), (uint32_t)cap};
        if (lists) hipLaunchKernelGGL((k_sumdij_lists<R, KSET>), g, b, 0, stream, P, G, I, hb, posB.as<T4>(), dens.as<R>(), N);
        else hipLaunchKernelGGL((k_sumdij_ref<R, KSET>), g, b, 0, stream, P, G, I, posB.as<T4>(), dens.as<R>(), N);
        const bool walls = HAS_B && lists && wallListed; // (this step's wall list: built for the scan, iisph_predict)
        const WallList wv = wall_view(), noWalls = {nullptr, nullptr, nullptr, nullptr, nullptr};
        const uint32_t wb = wall_blocks(g.x);
        if (lists && walls)
            hipLaunchKernelGGL((k_pressure_lists<R, KSET, HAS_B, true>), dim3(g.x + wb), b, 0, stream, P, G, I, hb, posB.as<T4>(), dens.as<R>(), presB.as<R>(), N, wv, wb);
        else if (lists)
            hipLaunchKernelGGL((k_pressure_lists<R, KSET, HAS_B>), g, b, 0, stream, P, G, I, hb, posB.as<T4>(), dens.as<R>(), presB.as<R>(), N, noWalls, 0u);
        else
            hipLaunchKernelGGL((k_pressure_ref<R, KSET, HAS_B>), g, b, 0, stream, P, G, I, posB.as<T4>(), dens.as<R>(),
                               presB.as<R>(), N);
        std::swap(P_l.p, P_l2.p);
        ++iisphIter;
        return NRS_OK;
    }
    // computePressureForce + iisph_integrate (sph_cuda.cu:827-867)
    template <bool HAS_B> int iisph_finish(int stop)
    {
        const uint32_t N = (uint32_t)n;
        const dim3 g(nblocks(N)), b(BLOCK);
        const GridView<R> G = grid_view();
        IisphArrays<R> I = iisph_view();
        const bool lists = iisph_lists();
        const HitBuffer hb = {hitBuf.as<uint32_t>(), hitCounts.as<uint32_t>(), (uint32_t)cap};
        NRSCHK(ev_begin(NRS_STAGE_I_PFORCE));
        const bool walls = HAS_B && lists && wallListed;
        const WallList wv = wall_view(), noWalls = {nullptr, nullptr, nullptr, nullptr, nullptr};
        const uint32_t wb = wall_blocks(g.x);
        if (lists && walls)
            hipLaunchKernelGGL((k_pforce_lists<R, KSET, HAS_B, true>), dim3(g.x + wb), b, 0, stream, P, G, I, hb, posB.as<T4>(), dens.as<R>(), presB.as<R>(), N, wv, wb);
        else if (lists)
            hipLaunchKernelGGL((k_pforce_lists<R, KSET, HAS_B>), g, b, 0, stream, P, G, I, hb, posB.as<T4>(), dens.as<R>(), presB.as<R>(), N, noWalls, 0u);
        else
            hipLaunchKernelGGL((k_pforce_ref<R, KSET, HAS_B>), g, b, 0, stream, P, G, I, posB.as<T4>(), dens.as<R>(), presB.as<R>(), N);
        NRSCHK(ev_end());
        if (stop == NRS_STAGE_I_PFORCE) return NRS_OK;
        NRSCHK(ev_begin(NRS_STAGE_I_INTEGRATE));
        // a full step on the production kernels also leaves the next step's sort keys (and the split of the coherent
        // re-sort), as the fused SESPH force kernel does — not in slab runs, whose arrays are re-partitioned first
        const bool keys = !refOrder() && stop == 0 && !(cfg.flags & NRS_FLAG_NO_FUSION) && !slabOn;
        const bool resort = keys && rsMovers.p && (uint64_t)N >= RESORT_MIN_PARTICLES;
        uint32_t *nh = nullptr, *ni = nullptr;
        if (keys) {
            nh = (hashCur == hashA.as<uint32_t>()) ? hashB.as<uint32_t>() : hashA.as<uint32_t>();
            ni = (indexCur == indexA.as<uint32_t>()) ? indexB.as<uint32_t>() : indexA.as<uint32_t>();
        }
        if (resort) NRSCHK(clean_tile_counts());
        hipLaunchKernelGGL((k_iisph_integrate<R>), g, b, 0, stream, P, posB.as<T4>(), velB.as<T4>(), velAdv.as<T4>(), forcesP.as<T4>(), N,
                           nh, ni, resort ? (const uint32_t *)hashCur : (const uint32_t *)nullptr,
                           resort ? rsTileMovers.as<uint32_t>() : (uint32_t *)nullptr, slabOn ? 1 : 0);
        NRSCHK(ev_end());
        if (keys) {
            keys_ready(nh, ni);
            if (resort) NRSCHK(queue_resort_split(N));
        }
        return NRS_OK;
    }

    // The list-driven chain is the reference-order chain only while every value a neighbour gathers is finite (IisphArrays::nonFinite).
    // A solve that overflows raises the flag; the step is then repeated from the sorted input with the reference-order kernels,
    // so that even a diverging run produces what the reference's loops produce.  (Not in slab runs, whose loop the host drives.)
    template <bool HAS_B> int iisph_tail(int stop)
    {
        const bool watch = iisph_lists() && !slabOn;
        uint32_t *flag = errWord.as<uint32_t>() + 1;
        if (watch) HIPCHK(hipMemsetAsync(flag, 0, 4, stream));
        NRSCHK(iisph_tail_once<HAS_B>(stop, watch));
        if (!watch || !iisphDiverged) return NRS_OK;
        refOverride = true;
        ++iisphRestarts;
        hipLaunchKernelGGL((k_gather_scalar<R>), dim3(nblocks((uint32_t)n)), dim3(BLOCK), 0, stream, presA.as<R>(), indexCur, presB.as<R>(), (uint32_t)n);
        const int rc = iisph_tail_once<HAS_B>(stop, false);
        refOverride = false;
        return rc;
    }
    bool iisphDiverged = false;
    uint64_t iisphRestarts = 0;
    template <bool HAS_B> int iisph_tail_once(int stop, bool watch)
    {
        const uint32_t N = (uint32_t)n;
        iisphDiverged = false;
        uint32_t hflag = 0u;
        uint32_t *flag = errWord.as<uint32_t>() + 1;
        NRSCHK(iisph_predict<HAS_B>(stop));
        if (stop && stop <= NRS_STAGE_I_ADVECTION) {
            if (watch) {
                HIPCHK(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                iisphDiverged = hflag != 0u;
            }
            return NRS_OK;
        }
        // pressureSolve (sph_cuda.cu:702-899): while ((rho_avg - 1000) > 1 || l < 2)
        NRSCHK(ev_begin(NRS_STAGE_I_SOLVE));
        uint32_t l = 0;
        R rho_avg = 0.f;
        const R rd = 1000.f;
        const R max_rho_err = 1.f;
        bool flagRead = false;
        while (((rho_avg - rd) > max_rho_err) || (l < 2)) {
            NRSCHK(iisph_iteration<HAS_B>());
            l++;
            flagRead = false;
            if (maxIters && l >= maxIters) break;
            // the loop condition reads rho_avg only once l >= 2 (sph_cuda.cu:736: `|| l < 2`): the average of the first
            // iteration is never looked at, so its reduction and host round trip are skipped
            if (l >= 2) {
                double acc = 0.0;
                if (watch) HIPCHK(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, stream)); // (rides in the reduction's round trip)
                NRSCHK(reduce_sum(densCorr.as<R>(), N, &acc));
                flagRead = true;
                if (watch && hflag) break;
                rho_avg = (R)acc;
                rho_avg /= N;
            }
        }
        if (watch && !flagRead) {
            HIPCHK(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        lastIters = l;
        NRSCHK(ev_end());
        if (watch && hflag) { iisphDiverged = true; return NRS_OK; } // (the caller repeats the step in reference order)
        if (stop == NRS_STAGE_I_SOLVE) return NRS_OK;
        return iisph_finish<HAS_B>(stop);
    }

    // ---- host-driven IISPH step (multi-GPU: the loop exit needs the average over ALL ranks) ---------------------------------
    int iisph_phase(int phase, double *sum, uint64_t *count) override
    {
        if (!iisph()) return fail(NRS_E_STATE, "not an IISPH context");
        NRSCHK(validate("nrs_iisph_*"));
        if (phase == 0) { // predict
            if (midStep || iisphPhase) return fail(NRS_E_STATE, "a step is already in progress");
            if (n == 0) return NRS_OK;
            fusedThisStep = false; splitClearedCells = false;
            NRSCHK(stage_prefix(0));
            if (nb) NRSCHK(iisph_predict<true>(0)); else NRSCHK(iisph_predict<false>(0));
            iisphPhase = 1;
            return NRS_OK;
        }
        if (!iisphPhase) return fail(NRS_E_STATE, "nrs_iisph_predict first");
        if (phase == 1) { // one iteration + the density-error sum over the particles this rank owns
            if (slabOn && (int)iisphIter >= slabMaxIters())
                return fail(NRS_E_STATE, "IISPH slab run: more solver iterations than the halo width supports (halo >= 2 * iterations + 4 cells)");
            if (nb) NRSCHK(iisph_iteration<true>()); else NRSCHK(iisph_iteration<false>());
            const uint32_t N = (uint32_t)n, nbk = std::min<uint32_t>(1024u, nblocks(N));
            unsigned long long *cnt = (unsigned long long *)((char *)redOut.p); // redOut: [double sum][u64 count]
            HIPCHK(hipMemsetAsync(redOut.p, 0, 16, stream));
            hipLaunchKernelGGL((k_sum_partial<R>), dim3(nbk), dim3(BLOCK), 0, stream, densCorr.as<R>(), redPartial.as<double>(), N,
                               slabOn ? posB.as<T4>() : (const T4 *)nullptr, cnt + 1);
            hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(BLOCK), 0, stream, redPartial.as<double>(), redOut.as<double>(), nbk);
            double h[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(h, redOut.p, 16, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            unsigned long long c;
            std::memcpy(&c, &h[1], 8);
            if (sum) *sum = h[0];
            if (count) *count = slabOn ? (uint64_t)c : (uint64_t)N;
            lastIters = iisphIter;
            return NRS_OK;
        }
        // finish
        if (iisphIter == 0) return fail(NRS_E_STATE, "nrs_iisph_iterate at least once before nrs_iisph_finish");
        if (nb) NRSCHK(iisph_finish<true>(0)); else NRSCHK(iisph_finish<false>(0));
        HIPCHK(hipGetLastError());
        NRSCHK(end_of_step());
        iisphPhase = 0;
        return NRS_OK;
    }
    int slabMaxIters() const { return (slab.halo - 4) / 2; }

    // ---- slab decomposition (nrs_kernels_slab.h) -------------------------------------------------------
    int slab_configure(int lo, int hi, int halo) override
    {
        // IISPH: every solver iteration consumes two cells of halo validity, the predict stages three and the pressure force one
        // (DESIGN.md §5): 2 iterations — the reference's minimum — need 8 cells
        NRSCHK(refuse_mid_iisph("nrs_slab_configure"));
        if (iisph() && halo < 8) return fail(NRS_E_INVALID, "IISPH slabs need a halo of at least 8 cells (2 * iterations + 4)");
        if (halo < 2) return fail(NRS_E_INVALID, "halo must be >= 2 cells (one cell for the density of the ring + one)");
        if ((long long)hi - lo < 2ll * halo) return fail(NRS_E_INVALID, "slab narrower than two halos");
        if (classifiedValid && (slab.lo != lo || slab.hi != hi || slab.halo != halo)) {
            // the last force kernel classified (and marked dead keys) for the old cuts: partition the slow way once
            classifiedValid = false;
            slotOrderValid = false;
        }
        slab.lo = lo; slab.hi = hi; slab.halo = halo;
        slabOn = true;
        nOwned = n;
        // cell-table window of this rank (see PU / P): re-chosen only when the slab no longer fits the current one
        const uint32_t cellsBefore = P.numCells, baseBefore = P.numBodies;
        if (choose_window(lo, hi, halo, false)) {
            NRSCHK(invalidate_grid_state());
            derive_kernel_params();
            if (P.numCells != cellsBefore || P.numBodies != baseBefore) {
                cellsAllocated = 0; // (same size, other columns: the tables still have to be reset)
                NRSCHK(alloc_cells());
                if (nb) NRSCHK(rebuild_boundary_tables());
            }
        }
        return NRS_OK;
    }
    // Window [winBase, winBase + winW) of cell-x columns covering the slab, its halo, two columns of drift and WINDOW_SLACK columns
    // of room for moving cuts; returns true when it changed.  force: choose afresh (the global grid changed).
    static constexpr int WINDOW_SLACK = 8;
    bool choose_window(int lo, int hi, int halo, bool force)
    {
        const long long GX = (long long)PU.gridSize[0];
        const bool pow2 = is_pow2(PU.gridSize[0]) && is_pow2(PU.gridSize[1]) && is_pow2(PU.gridSize[2]);
        long long a = std::max<long long>(0, (long long)lo - halo - 2), b = std::min<long long>(GX, (long long)hi + halo + 2);
        if (!pow2 || b <= a) {
            const bool changed = winW != 0;
            winW = 0; winBase = 0;
            return changed;
        }
        if (!force && winW && a >= winBase && b <= (long long)winBase + (long long)winW) return false; // still fits
        a = std::max<long long>(0, a - WINDOW_SLACK); b = std::min<long long>(GX, b + WINDOW_SLACK);
        const uint32_t w = next_pow2((uint32_t)(b - a));
        const int baseOld = winBase; const uint32_t wOld = winW;
        if (w >= (uint32_t)GX) { winW = 0; winBase = 0; }
        else { winW = w; winBase = (int)a; }
        return winW != wOld || winBase != baseOld;
    }
    uint64_t num_owned() override { return slabOn ? nOwned : n; }
    int slab_histogram(int lo0, uint32_t nbins, uint32_t *out) override
    {
        if (!nbins || !out) return fail(NRS_E_INVALID, "bad histogram request");
        NRSCHK(compact_holes());
        DevBuf bins;
        NRSCHK(bins.alloc((size_t)nbins * 4));
        HIPCHK(hipMemsetAsync(bins.p, 0, (size_t)nbins * 4, stream));
        if (n)
            hipLaunchKernelGGL((k_slab_histogram<R>), dim3((uint32_t)((n + SLAB_BLOCK - 1) / SLAB_BLOCK)), dim3(SLAB_BLOCK), 0, stream, P,
                               posA.as<T4>(), (uint32_t)n, lo0, nbins, bins.as<uint32_t>());
        HIPCHK(hipMemcpyAsync(out, bins.p, (size_t)nbins * 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        bins.release();
        return NRS_OK;
    }

    int slab_pack(void *sendL, void *sendR, uint64_t mcap, uint32_t *counts) override
    {
        NRSCHK(validate("nrs_slab_pack"));
        if (!slabOn) return fail(NRS_E_STATE, "nrs_slab_configure first");
        if (midStep) return fail(NRS_E_STATE, "state is mid-update");
        if (mcap == 0 || mcap > (uint64_t)HIT_INDEX) return fail(NRS_E_INVALID, "bad message capacity");
        NRSCHK(finish_pack());   // (a pack right behind a pack whose totals nobody has looked at yet)
        NRSCHK(compact_holes()); // (a pack right after a pack/unpack without a step in between)
        if (iisphPhase) return fail(NRS_E_STATE, "a host-driven IISPH step is in progress");
        if (iisph() && n) // the warm-start pressure travels in vel.w (k_pressure_to_velw)
            hipLaunchKernelGGL((k_pressure_to_velw<R>), dim3(nblocks(n)), dim3(BLOCK), 0, stream, velA.as<T4>(), presA.as<R>(), (uint32_t)n);
        const uint32_t N = (uint32_t)n;
        const uint32_t nbk = std::max<uint32_t>(1u, (N + SLAB_TILE - 1) / SLAB_TILE);
        NRSCHK(slabCounts.alloc((size_t)ST_TOTALS * ((cap_blocks() > nbk) ? cap_blocks() : nbk) * 4));
        NRSCHK(slabTotals.alloc(ST_TOTALS * 4));
        NRSCHK(ghostPos.alloc(sizeof(T4) * mcap));
        NRSCHK(ghostVel.alloc(sizeof(T4) * mcap));
        bool inplace = false;
        if (!slabHostTotals) HIPCHK(hipHostMalloc((void **)&slabHostTotals, 128, hipHostMallocDefault));
        if (!packEvent) HIPCHK(hipEventCreateWithFlags(&packEvent, hipEventDisableTiming)); // (contexts without re-sort buffers have none yet)
        if (N) {
            const bool fusedClass = classifiedValid && slotOrderValid && classifiedN == N && rsMovers.p && hashCur && hashNext &&
                                    hashNext != hashCur;
            classifiedValidAtPack = fusedClass;
            if (fusedClass) {
                // the force kernel of the last step classified every slot for these cuts (flags, stream populations per
                // 2048 slots, dead marks in the keys, movers / dead per 256 slots): scan, copy out the few particles of
                // the message and ghost streams, split
                inplace = true;
                packResort = true;
                packKeys = hashNext; packVals = indexNext;
                hipLaunchKernelGGL(k_slab_scan, dim3(ST_TOTALS), dim3(SLAB_BLOCK), 0, stream, slabCounts.as<uint32_t>(), nbk,
                                   slabTotals.as<uint32_t>());
                SlabOut<R> out;
                out.stayPos = posB.as<T4>(); out.stayVel = velB.as<T4>();
                out.hash = packKeys; out.index = packVals;
                out.prevHash = hashCur; out.prevPacked = nullptr;
                out.tileMovers = rsTileMovers.as<uint32_t>(); out.tileDead = rsTileDead.as<uint32_t>();
                out.flags = slabFlags.as<uint8_t>();
                out.ghostPos = ghostPos.as<T4>(); out.ghostVel = ghostVel.as<T4>();
                out.sendL = (unsigned char *)sendL; out.sendR = (unsigned char *)sendR;
                out.cap = (uint32_t)mcap;
                hipLaunchKernelGGL((k_slab_scatter<R, true, true>), dim3(nbk), dim3(SLAB_BLOCK), 0, stream, P, slab, posA.as<T4>(),
                                   velA.as<T4>(), N, slabCounts.as<uint32_t>(), nbk, slabTotals.as<uint32_t>(), out);
                hipLaunchKernelGGL(k_slab_headers, dim3(1), dim3(64), 0, stream, slabTotals.as<uint32_t>(), (unsigned char *)sendL,
                                   (unsigned char *)sendR);
                const uint32_t nTiles = nblocks(N);
                NRSCHK(launch_resort_scan(nTiles, true)); // also totals the cell changers and the dead slots
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(slabHostTotals, slabTotals.p, ST_TOTALS * 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipMemcpyAsync(slabHostTotals + 8, rsScalars.p, 16, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipEventRecord(packEvent, stream));
                hipLaunchKernelGGL((k_resort_split<true>), dim3(nTiles), dim3(BLOCK), 0, stream, hashCur, hashNext, offsets_movers(),
                                   offsets_dead(), rsMovers.as<uint64_t>(), rsStayers.as<uint64_t>(), N, (uint32_t *)nullptr);
                HIPCHK(hipGetLastError());
                rsTilesDirty = false; // the scan resets the counts it reads
            } else {
                // coherent re-sort of the next step: possible when the arrays are still in the slot order of the last sort and
                // the fused force kernel left the new keys per slot
                const bool resort = rsMovers.p && slotOrderValid && hashCur && hashNext && hashNext != hashCur;
                // ... and then the owned particles need not be moved at all (in-place partition, see k_slab_scatter)
                inplace = resort && (uint64_t)N >= RESORT_MIN_PARTICLES;
                hipLaunchKernelGGL((k_slab_count<R>), dim3(nbk), dim3(SLAB_BLOCK), 0, stream, P, slab, posA.as<T4>(), N,
                                   slabCounts.as<uint32_t>(), nbk, resort ? hashCur : (const uint32_t *)nullptr,
                                   resort ? hashNext : (const uint32_t *)nullptr);
                hipLaunchKernelGGL(k_slab_scan, dim3(ST_TOTALS), dim3(SLAB_BLOCK), 0, stream, slabCounts.as<uint32_t>(), nbk,
                                   slabTotals.as<uint32_t>());
                SlabOut<R> out;
                out.stayPos = posB.as<T4>(); out.stayVel = velB.as<T4>();
                // the hash pass of the next step, done here (into the key buffers the last sort did not end in)
                packKeys = (hashCur == hashA.as<uint32_t>()) ? hashB.as<uint32_t>() : hashA.as<uint32_t>();
                packVals = (indexCur == indexA.as<uint32_t>()) ? indexB.as<uint32_t>() : indexA.as<uint32_t>();
                if (inplace) { packKeys = hashNext; packVals = indexNext; } // the fused kernel's keys / slot numbers stay where they are
                out.hash = packKeys; out.index = packVals;
                out.prevHash = resort ? hashCur : nullptr;
                out.prevPacked = (resort && !inplace) ? rsPrevPacked.as<uint32_t>() : nullptr;
                out.tileMovers = resort ? rsTileMovers.as<uint32_t>() : nullptr;
                out.tileDead = inplace ? rsTileDead.as<uint32_t>() : nullptr;
                if (resort) NRSCHK(clean_tile_counts());
                if (resort) rsTilesDirty = true; // (never clear it here: the counts of an unused classification may still be in the arrays)
                packResort = resort;
                out.ghostPos = ghostPos.as<T4>(); out.ghostVel = ghostVel.as<T4>();
                out.sendL = (unsigned char *)sendL; out.sendR = (unsigned char *)sendR;
                out.cap = (uint32_t)mcap;
                if (inplace)
                    hipLaunchKernelGGL((k_slab_scatter<R, true>), dim3(nbk), dim3(SLAB_BLOCK), 0, stream, P, slab, posA.as<T4>(), velA.as<T4>(), N,
                                       slabCounts.as<uint32_t>(), nbk, slabTotals.as<uint32_t>(), out);
                else
                    hipLaunchKernelGGL((k_slab_scatter<R, false>), dim3(nbk), dim3(SLAB_BLOCK), 0, stream, P, slab, posA.as<T4>(), velA.as<T4>(), N,
                                       slabCounts.as<uint32_t>(), nbk, slabTotals.as<uint32_t>(), out);
                hipLaunchKernelGGL(k_slab_headers, dim3(1), dim3(64), 0, stream, slabTotals.as<uint32_t>(), (unsigned char *)sendL,
                                   (unsigned char *)sendR);
                HIPCHK(hipGetLastError());
                // page-locked destination: the copy is complete when the event behind it is (a pageable destination is only
                // guaranteed after a stream synchronization, which would also wait for the split queued below)
                HIPCHK(hipMemcpyAsync(slabHostTotals, slabTotals.p, ST_TOTALS * 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipEventRecord(packEvent, stream));
                if (inplace) {
                    // the split of the slots we keep does not depend on what arrives: queue it now, so that it runs while the
                    // messages travel
                    const uint32_t nTiles = nblocks(N);
                    NRSCHK(launch_resort_scan(nTiles, true));
                    hipLaunchKernelGGL((k_resort_split<true>), dim3(nTiles), dim3(BLOCK), 0, stream, hashCur, hashNext, offsets_movers(),
                                       offsets_dead(), rsMovers.as<uint64_t>(), rsStayers.as<uint64_t>(), N, (uint32_t *)nullptr);
                    HIPCHK(hipGetLastError());
                    rsTilesDirty = false; // the scan resets the counts it reads
                }
            }
        } else {
            HIPCHK(hipMemsetAsync(slabTotals.p, 0, ST_TOTALS * 4, stream));
            hipLaunchKernelGGL(k_slab_headers, dim3(1), dim3(64), 0, stream, slabTotals.as<uint32_t>(), (unsigned char *)sendL,
                               (unsigned char *)sendR);
            HIPCHK(hipEventRecord(packEvent, stream));
        }
        // Round 3: nothing above waits.  The messages are complete in stream order, so the caller can enqueue its sends right behind
        // this call; the stream totals (how many stay, leave, ghost) are read back by finish_pack() — in nrs_slab_unpack, together
        // with the headers of the received messages: ONE host synchronisation per exchange instead of two — or by whichever entry
        // point needs the particle count first (settle()).
        packPending = true;
        pendFused = N && classifiedValidAtPack;
        pendInplace = inplace;
        pendN = N;
        pendCap = mcap;
        if (counts) { // the caller wants the counts now: that is the synchronisation it asked for
            NRSCHK(finish_pack());
            std::memcpy(counts, lastCounts, ST_COUNT * sizeof(uint32_t));
        }
        return NRS_OK;
    }
    // ---- the host half of nrs_slab_pack, run when the stream totals are needed ---------------------------------------------
    bool packPending = false, pendFused = false, pendInplace = false, classifiedValidAtPack = false;
    uint32_t pendN = 0;
    uint64_t pendCap = 0;
    uint32_t lastCounts[ST_COUNT] = {0, 0, 0, 0, 0, 0};
    int finish_pack()
    {
        if (!packPending) return NRS_OK;
        packPending = false;
        const uint32_t N = pendN;
        const bool inplace = pendInplace;
        uint32_t tot[ST_TOTALS] = {0, 0, 0, 0, 0, 0, 0};
        HIPCHK(hipEventSynchronize(packEvent));
        if (N) {
            std::memcpy(tot, slabHostTotals, sizeof(tot));
            if (pendFused) { // (pre-classified partition: cell changers and dead slots come from the re-sort's scan)
                tot[ST_CHANGED] = slabHostTotals[8 + 1];
                tot[ST_STAY] = N - slabHostTotals[8 + 2];
            }
        }
        if ((uint64_t)tot[ST_STAY] + tot[ST_MIG_L] + tot[ST_MIG_R] > N || tot[ST_CHANGED] > tot[ST_STAY])
            return fail(NRS_E_HIP, "inconsistent slab stream totals");
        const bool overflow = (uint64_t)tot[ST_MIG_L] + tot[ST_HALO_L] > pendCap || (uint64_t)tot[ST_MIG_R] + tot[ST_HALO_R] > pendCap ||
                              tot[ST_GHOST] > pendCap;
        to_fresh();
        if (inplace) {
            // hashNext / indexNext hold key and slot of every live slot, 0xffffffff marks the dead ones; arrivals are added to the mover
            // count by nrs_slab_unpack
            to_holes(N, tot[ST_CHANGED]);
        } else {
            if (N) { std::swap(posA.p, posB.p); std::swap(velA.p, velB.p); }
            else { packKeys = hashA.as<uint32_t>(); packVals = indexA.as<uint32_t>(); packResort = false; }
            packedHashValid = N != 0; // k_slab_scatter hashed the particles that stay (with the current parameters)
        }
        packInplace = inplace;
        n = tot[ST_STAY];
        nOwned = n;
        ghostCount = tot[ST_GHOST];
        packChanged = tot[ST_CHANGED];
        std::memcpy(lastCounts, tot, ST_COUNT * sizeof(uint32_t));
        if (overflow) return fail(NRS_E_CAPACITY, "slab message capacity exceeded");
        return NRS_OK;
    }
    int settle() override { return finish_pack(); }
    int slab_last_counts(uint32_t *counts) override
    {
        NRSCHK(finish_pack());
        std::memcpy(counts, lastCounts, ST_COUNT * sizeof(uint32_t));
        return NRS_OK;
    }
    uint32_t cap_blocks() const { return (uint32_t)((cap + SLAB_TILE - 1) / SLAB_TILE); }

    int slab_unpack(const void *recvL, const void *recvR, uint64_t mcap) override
    {
        NRSCHK(validate("nrs_slab_unpack"));
        NRSCHK(refuse_mid_iisph("nrs_slab_unpack"));
        if (!slabOn) return fail(NRS_E_STATE, "nrs_slab_configure first");
        // ONE host synchronisation for the exchange: the headers of the received messages (how many migrants, how many halo copies)
        // are copied to page-locked memory behind the receives, and the same wait covers the stream totals of the pack (finish_pack)
        uint32_t hL[4] = {0, 0, 0, 0}, hR[4] = {0, 0, 0, 0};
        if (!slabHostTotals) HIPCHK(hipHostMalloc((void **)&slabHostTotals, 128, hipHostMallocDefault));
        if (recvL) HIPCHK(hipMemcpyAsync(slabHostTotals + 16, recvL, 16, hipMemcpyDeviceToHost, stream));
        if (recvR) HIPCHK(hipMemcpyAsync(slabHostTotals + 20, recvR, 16, hipMemcpyDeviceToHost, stream));
        if (recvL || recvR) HIPCHK(hipStreamSynchronize(stream));
        if (recvL) std::memcpy(hL, slabHostTotals + 16, 16);
        if (recvR) std::memcpy(hR, slabHostTotals + 20, 16);
        NRSCHK(finish_pack());
        if ((uint64_t)hL[0] + hL[1] > mcap || (uint64_t)hR[0] + hR[1] > mcap) return fail(NRS_E_INVALID, "corrupt slab message header");
        const bool inplace = packInplace && holesPending;
        const uint64_t arrivals = (uint64_t)hL[0] + hR[0] + ghostCount + hL[1] + hR[1];
        const uint64_t total = n + arrivals;                           // live particles of the next step
        const uint64_t base = inplace ? (uint64_t)physN : (uint64_t)n; // first free physical slot
        if (base + arrivals > cap) return fail(NRS_E_CAPACITY, "owned + halo particles exceed the context capacity");
        const unsigned char *bL = (const unsigned char *)recvL, *bR = (const unsigned char *)recvR;
        auto mp = [&](const unsigned char *b) { return (const T4 *)(b + 16); };
        auto mv = [&](const unsigned char *b) { return (const T4 *)(b + 16 + (size_t)mcap * sizeof(T4)); };
        AppendPieces<R> A;
        const uint32_t len[5] = {hL[0], hR[0], ghostCount, hL[1], hR[1]};
        A.srcPos[0] = bL ? mp(bL) : nullptr;           A.srcVel[0] = bL ? mv(bL) : nullptr;            // migrants from the left
        A.srcPos[1] = bR ? mp(bR) : nullptr;           A.srcVel[1] = bR ? mv(bR) : nullptr;            // migrants from the right
        A.srcPos[2] = ghostPos.as<T4>();               A.srcVel[2] = ghostVel.as<T4>();               // our ghosts
        A.srcPos[3] = bL ? mp(bL) + hL[0] : nullptr;   A.srcVel[3] = bL ? mv(bL) + hL[0] : nullptr;    // halo from the left
        A.srcPos[4] = bR ? mp(bR) + hR[0] : nullptr;   A.srcVel[4] = bR ? mv(bR) + hR[0] : nullptr;    // halo from the right
        A.start[0] = 0;
        for (int k = 0; k < 5; ++k) A.start[k + 1] = A.start[k] + len[k];
        const bool compactResort = !inplace && packResort;
        if (A.start[5])
            hipLaunchKernelGGL((k_slab_append<R>), dim3((A.start[5] + SLAB_BLOCK - 1) / SLAB_BLOCK), dim3(SLAB_BLOCK), 0, stream, P, A,
                               posA.as<T4>(), velA.as<T4>(), packKeys, packVals,
                               compactResort ? rsPrevPacked.as<uint32_t>() : (uint32_t *)nullptr,
                               compactResort ? rsTileMovers.as<uint32_t>() : (uint32_t *)nullptr, (uint32_t)base,
                               inplace ? rsMovers.as<uint64_t>() : (uint64_t *)nullptr, inplace ? rsKnownCount : 0u);
        HIPCHK(hipGetLastError());
        nOwned = n + hL[0] + hR[0];
        n = total;
        if (iisph() && n)
            hipLaunchKernelGGL((k_velw_to_pressure<R>), dim3(nblocks(n)), dim3(BLOCK), 0, stream, velA.as<T4>(), presA.as<R>(), (uint32_t)n);
        // pack + unpack have written the radix keys/values of every local particle
        hashNext = packKeys; indexNext = packVals;
        hashReady = packedHashValid;
        if (inplace) {
            physN += (uint32_t)arrivals;
            rsKnownCount += (uint32_t)arrivals; // every arrival is a mover (k_slab_append put it behind the cell changers)
        } else if (hashReady && packResort && n >= RESORT_MIN_PARTICLES) {
            // coherent re-sort: the owned particles that stayed in their cell are still in sorted order
            // (the partition and the append have counted the movers of every tile of the new arrays)
            const uint32_t N = (uint32_t)n, nTiles = nblocks(N);
            NRSCHK(launch_resort_scan(nTiles, false));
            hipLaunchKernelGGL((k_resort_split<false>), dim3(nTiles), dim3(BLOCK), 0, stream, rsPrevPacked.as<uint32_t>(), hashNext,
                               offsets_movers(), offsets_movers(), rsMovers.as<uint64_t>(), rsStayers.as<uint64_t>(), N, (uint32_t *)nullptr);
            split_queued_known(packChanged + A.start[5]); // everything appended is a mover, and the partition counted the cell changers
            rsTilesDirty = false; // the scan resets the counts it reads
        }
        packResort = false;
        return NRS_OK;
    }

    // The scan kernel stores (launch number, count) straight into mapped host memory; polling that word costs a PCIe
    // write latency, where hipEventSynchronize on an otherwise idle host thread was measured to cost ~0.1 ms per step.
    int wait_mover_count(uint32_t *M)
    {
        volatile uint64_t *w = (volatile uint64_t *)rsHostTotal;
        {
            for (uint64_t spins = 0;; ++spins) {
                const uint64_t v = *w;
                if ((uint32_t)(v >> 32) == rsSeq) { *M = (uint32_t)v; return NRS_OK; }
                // polite spin: on a host with fewer free cores than ranks (8 ranks on a 16-CPU share) the poller hands its
                // time slice to whoever is runnable; with an idle core the yield returns at once and costs no latency
                if ((spins & 63u) == 63u) sched_yield();
                if ((spins & 0xfffff) == 0xfffff) { // every ~1 M polls: has the stream failed or finished without us seeing it?
                    const hipError_t e = hipEventQuery(rsEvent);
                    if (e == hipSuccess) break;
                    if (e != hipErrorNotReady) HIPCHK(e);
                }
            }
        }
        HIPCHK(hipEventSynchronize(rsEvent));
        *M = (uint32_t)*w;
        return NRS_OK;
    }
    void resort_stats(uint64_t *steps, uint64_t *fallbacks) override
    {
        if (steps) *steps = rsSteps;
        if (fallbacks) *fallbacks = rsFallbacks;
    }
    int get_stat(int which, double *out) override
    {
        if (which == NRS_STAT_MOVERS) { *out = lastMovers; return NRS_OK; }
        if (which != NRS_STAT_HIT_OVERFLOW && which != NRS_STAT_HIT_MEAN && which != NRS_STAT_HIT_MAX && which != NRS_STAT_UNSTAGED)
            return fail(NRS_E_INVALID, "unknown statistic");
        if (!hitCounts.p || !n || midStep) return fail(NRS_E_STATE, "no shared hit lists (reference-order kernels, or no step yet)");
        const uint32_t N = (uint32_t)n;
        HIPCHK(hipMemsetAsync(redPartial.p, 0, 4 * sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(k_hit_stats, dim3(std::min<uint32_t>(1024u, nblocks(N))), dim3(BLOCK), 0, stream, hitCounts.as<uint32_t>(),
                           (unsigned long long *)redPartial.p, N);
        unsigned long long h[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(h, redPartial.p, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        *out = which == NRS_STAT_HIT_OVERFLOW ? (double)h[0] : (which == NRS_STAT_HIT_MEAN ? (double)h[1] / (double)N : (which == NRS_STAT_HIT_MAX ? (double)h[2] : (double)h[3]));
        return NRS_OK;
    }
    // bookkeeping at the end of a completed step (cell-table undo, buffer swaps)
    int end_of_step()
    {
        if ((uint64_t)P.numCells > 8ull * n) { // big, mostly empty table: undo only the touched cells
            if (!splitClearedCells)
                hipLaunchKernelGGL(k_clear_cells, dim3(nblocks(n)), dim3(BLOCK), 0, stream, hashCur, cellStart.as<uint32_t>(), (uint32_t)n);
            cellsClean = true;
        }
        // the integrated sorted arrays become the next step's input (replaces D2H + H2D)
        ++stepsDone;
        slotOrderValid = fusedThisStep; // A holds the new state in the slot order of hashCur
        if (!fusedThisStep) { // the fused kernel already wrote the new state into A
            std::swap(posA.p, posB.p);
            std::swap(velA.p, velB.p);
        }
        if (iisph()) std::swap(presA.p, presB.p);
        return NRS_OK;
    }
    int step(int nsteps, int stop) override
    {
        if (iisphPhase) return fail(NRS_E_STATE, "a host-driven IISPH step is in progress (nrs_iisph_finish first)");
        NRSCHK(validate("nrs_step"));
        if (midStep) return fail(NRS_E_STATE, "state is mid-update after nrs_step_partial; upload particles first");
        if (n == 0) return NRS_OK;
        for (int s = 0; s < nsteps; ++s) {
            fusedThisStep = false;
            splitClearedCells = false;
            NRSCHK(stage_prefix(stop));
            if (stop && stop <= NRS_STAGE_REORDER) { midStep = true; break; }
            if (iisph()) { if (nb) NRSCHK(iisph_tail<true>(stop)); else NRSCHK(iisph_tail<false>(stop)); }
            else { if (nb) NRSCHK(sesph_tail<true>(stop)); else NRSCHK(sesph_tail<false>(stop)); }
            HIPCHK(hipGetLastError());
            if (stop) { midStep = true; break; }
            NRSCHK(end_of_step());
        }
        HIPCHK(hipGetLastError());
        if (evUsed > 8192) NRSCHK(ev_collect()); // bound the pool of pending event pairs
        return NRS_OK;
    }
    // the device-side guard (GridView::err) fired since the last check: report it, once
    int check_device_error()
    {
        uint32_t e = 0;
        HIPCHK(hipMemcpyAsync(&e, errWord.p, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (!e) return NRS_OK;
        HIPCHK(hipMemsetAsync(errWord.p, 0, 4, stream));
        return fail(NRS_E_STATE, "device-side consistency check failed: a cell range lies outside the sorted particle array (the cell table was "
                                 "built from an inconsistent sort); the neighbour sweep skipped it, results of the last steps are invalid");
    }
    int sync() override
    {
        HIPCHK(hipStreamSynchronize(stream));
        return check_device_error();
    }
    // ---- asynchronous snapshots for a viewer (include/nereus_hip.h: nrs_snapshot_*) ---------------------------
    struct Snap {
        DevBuf dPos, dVel;
        void *hPos = nullptr, *hVel = nullptr;
        size_t hBytes = 0;
        hipEvent_t staged = nullptr, done = nullptr;
        uint64_t n = 0, step = 0;
        bool withVel = false, pending = false;
    };
    Snap snaps[2];
    int snapHead = 0, snapTail = 0; // next slot to fill / oldest pending slot
    hipStream_t copyStream = nullptr;
    uint64_t stepsDone = 0;
    void snapshot_release()
    {
        for (Snap &sn : snaps) {
            if (sn.pending && sn.done) (void)hipEventSynchronize(sn.done);
            sn.dPos.release(); sn.dVel.release();
            if (sn.hPos) (void)hipHostFree(sn.hPos);
            if (sn.hVel) (void)hipHostFree(sn.hVel);
            if (sn.staged) (void)hipEventDestroy(sn.staged);
            if (sn.done) (void)hipEventDestroy(sn.done);
            sn = Snap();
        }
        if (copyStream) (void)hipStreamDestroy(copyStream);
        copyStream = nullptr;
    }
    int snapshot_begin(int withVel) override
    {
        NRSCHK(validate("nrs_snapshot_begin"));
        if (midStep) return fail(NRS_E_STATE, "state is mid-update");
        NRSCHK(compact_holes());
        if (!copyStream) HIPCHK(hipStreamCreateWithFlags(&copyStream, hipStreamNonBlocking));
        Snap &sn = snaps[snapHead];
        if (sn.pending) { // both slots in flight: the oldest is this one
            HIPCHK(hipEventSynchronize(sn.done));
            sn.pending = false;
            snapTail = (snapHead + 1) % 2;
        }
        const size_t bytes = sizeof(T4) * (size_t)cap;
        if (!sn.staged) {
            HIPCHK(hipEventCreateWithFlags(&sn.staged, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sn.done, hipEventDisableTiming));
        }
        NRSCHK(sn.dPos.alloc(bytes));
        if (!sn.hPos) HIPCHK(hipHostMalloc(&sn.hPos, bytes, hipHostMallocDefault));
        if (withVel) {
            NRSCHK(sn.dVel.alloc(bytes));
            if (!sn.hVel) HIPCHK(hipHostMalloc(&sn.hVel, bytes, hipHostMallocDefault));
        }
        const size_t live = sizeof(T4) * (size_t)n;
        if (live) {
            HIPCHK(hipMemcpyAsync(sn.dPos.p, posA.p, live, hipMemcpyDeviceToDevice, stream));
            if (withVel) HIPCHK(hipMemcpyAsync(sn.dVel.p, velA.p, live, hipMemcpyDeviceToDevice, stream));
        }
        HIPCHK(hipEventRecord(sn.staged, stream));
        HIPCHK(hipStreamWaitEvent(copyStream, sn.staged, 0));
        if (live) {
            HIPCHK(hipMemcpyAsync(sn.hPos, sn.dPos.p, live, hipMemcpyDeviceToHost, copyStream));
            if (withVel) HIPCHK(hipMemcpyAsync(sn.hVel, sn.dVel.p, live, hipMemcpyDeviceToHost, copyStream));
        }
        HIPCHK(hipEventRecord(sn.done, copyStream));
        sn.n = n; sn.step = stepsDone; sn.withVel = withVel != 0; sn.pending = true;
        snapHead = (snapHead + 1) % 2;
        return NRS_OK;
    }
    int snapshot_wait(int block, const void **pos4, const void **vel4, uint64_t *np, uint64_t *step) override
    {
        Snap &sn = snaps[snapTail];
        if (!sn.pending) return fail(NRS_E_STATE, "no snapshot in flight (nrs_snapshot_begin first)");
        if (block) {
            HIPCHK(hipEventSynchronize(sn.done));
        } else {
            const hipError_t e = hipEventQuery(sn.done);
            if (e == hipErrorNotReady) return NRS_E_NOTREADY;
            HIPCHK(e);
        }
        sn.pending = false;
        snapTail = (snapTail + 1) % 2;
        if (pos4) *pos4 = sn.hPos;
        if (vel4) *vel4 = sn.withVel ? sn.hVel : nullptr;
        if (np) *np = sn.n;
        if (step) *step = sn.step;
        return NRS_OK;
    }

    int download(void *pos4, void *vel4, void *pres) override
    {
        NRSCHK(validate("nrs_download"));
        NRSCHK(compact_holes());
        if (pos4) HIPCHK(hipMemcpyAsync(pos4, posA.p, sizeof(T4) * n, hipMemcpyDeviceToHost, stream));
        if (vel4) HIPCHK(hipMemcpyAsync(vel4, velA.p, sizeof(T4) * n, hipMemcpyDeviceToHost, stream));
        if (pres) HIPCHK(hipMemcpyAsync(pres, presA.p, sizeof(R) * n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return check_device_error();
    }
    int array(int which, void **dptr, uint64_t *bytes) override
    {
        if (which == NRS_ARR_POS || which == NRS_ARR_VEL) NRSCHK(compact_holes());
        const uint64_t v = sizeof(T4) * n, s = sizeof(R) * n, u = 4 * n, c = 4ull * P.numCells;
        void *p = nullptr;
        uint64_t sz = 0;
        // after a completed step the sorted arrays ARE the current arrays (buffers were swapped)
        const bool sortedIsCurrent = !midStep;
        switch (which) {
        case NRS_ARR_POS: p = posA.p; sz = v; break;
        case NRS_ARR_VEL: p = velA.p; sz = v; break;
        case NRS_ARR_PRESSURE: p = presA.p; sz = s; break;
        case NRS_ARR_HASH: p = hashCur; sz = u; break;
        case NRS_ARR_INDEX: p = indexCur; sz = u; break;
        case NRS_ARR_CELL_START: p = cellStart.p; sz = c; break;
        case NRS_ARR_CELL_END: p = cellEnd.p; sz = c; break;
        case NRS_ARR_SORTED_POS: p = sortedIsCurrent ? posA.p : posB.p; sz = v; break;
        case NRS_ARR_SORTED_VEL: p = sortedIsCurrent ? velA.p : velB.p; sz = v; break;
        case NRS_ARR_DENS: p = dens.p; sz = s; break;
        case NRS_ARR_PRES: p = (iisph() && sortedIsCurrent) ? presA.p : presB.p; sz = s; break;
        case NRS_ARR_FORCES: p = forces.p; sz = v; break;
        case NRS_ARR_B_HASH: p = bHashCur; sz = 4 * nb; break;
        case NRS_ARR_B_INDEX: p = bIndexCur; sz = 4 * nb; break;
        case NRS_ARR_B_CELL_START: p = nb ? bCellStart.p : nullptr; sz = nb ? c : 0; break;
        case NRS_ARR_B_CELL_END: p = nb ? bCellEnd.p : nullptr; sz = nb ? c : 0; break;
        case NRS_ARR_B_SORTED: p = bSorted.p; sz = sizeof(T4) * nb; break;
        case NRS_ARR_DENS_ADV: p = densAdv.p; sz = s; break;
        case NRS_ARR_DENS_CORR: p = densCorr.p; sz = s; break;
        case NRS_ARR_P_L: p = P_l.p; sz = s; break;
        case NRS_ARR_AII: p = aii.p; sz = s; break;
        case NRS_ARR_VEL_ADV: p = velAdv.p; sz = v; break;
        case NRS_ARR_FORCES_ADV: p = forcesAdv.p; sz = v; break;
        case NRS_ARR_FORCES_P: p = forcesP.p; sz = v; break;
        case NRS_ARR_DII_FLUID: p = diiF.p; sz = v; break;
        case NRS_ARR_DII_BOUNDARY: p = diiB.p; sz = v; break;
        case NRS_ARR_SUM_DIJ: p = sumDij.p; sz = v; break;
        default: return fail(NRS_E_INVALID, "unknown array id");
        }
        if (which >= NRS_ARR_DENS_ADV && !iisph()) return fail(NRS_E_STATE, "IISPH array requested from a SESPH context");
        *dptr = p;
        *bytes = p ? sz : 0;
        return NRS_OK;
    }
};

template <typename R, int KSET> CtxBase *make_ctx2(bool surf)
{
    if (surf) return new Ctx<R, KSET, true>();
    return new Ctx<R, KSET, false>();
}

} // namespace nrs
