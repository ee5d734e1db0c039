// nrs_kernels_resort.h — coherent re-sort of the (hash, index) pairs between two steps.
//
// The reference sorts all pairs from scratch every step (thrust::sort_by_key, sph_cuda.cu:310-313).  Between two
// steps only a few per cent of the particles change grid cell (CFL: < 1 cell per step; measured 0.3-7 % over the
// first 400 steps of the 1 M dam-break), and the arrays are still in the previous step's sorted order.  With
// k_i = next hash of the particle in sorted slot i, the stable sort by (k_i, i) is therefore a MERGE of
//   stayers  (k_i == previous hash of slot i): already sorted, because their keys are the previous sorted keys,
//   movers   (k_i != previous hash):           few; sorted on their own with a radix sort.
// Pairs travel as one u64 "hash << 32 | slot" so that the merge order IS the stable-sort order (no ties); the
// result is identical to the full stable radix sort, element for element.
//
//   k_forces_*  (fused epilogue)   counts the movers of every 256-slot tile            → tileMovers[tile]
//   k_resort_scan_tiles            exclusive scan of the tile counts, total to the host → tileOffset[], *total
//   k_resort_split                 stable two-way compaction into movers[] / stayers[]
//   rocprim::radix_sort_keys       movers only (bits 32 .. 32+log2(numCells))
//   rocprim::merge                 stayers + movers → merged[]
//   k_reorder_merged               cell ranges + gather from merged[], also leaves plain hash[] / index[] arrays
// The host needs the mover count to size the last two calls: it is read after the split has been queued, so the
// copy overlaps the split and the cell-table reset; above RESORT_MAX_MOVER_PCT % movers the step falls back to the full
// radix sort.
#pragma once
#include "nrs_kernels_ref.h"

namespace nrs {

constexpr int RESORT_GROUP = 1024;              // tiles per scan workgroup
constexpr uint64_t RESORT_MIN_PARTICLES = 32768; // below this the full sort is launch-bound either way
constexpr uint32_t RESORT_MAX_MOVER_PCT = 50;    // more movers than this share of N: full radix sort (measured break-even,
                                                 // DESIGN.md §4)

// Exclusive scan of the per-tile mover counts, two levels: every workgroup scans RESORT_GROUP counts (coalesced; the
// counts are reset to 0 for the next step's atomics) and the last one to finish scans the group totals.
// tileOffset[t] is local to the group; groupPrefix[t / RESORT_GROUP] is added by the consumer.
struct ResortScan { // one scanned array: counts in, offsets (local to the group) + group totals/prefixes out
    uint32_t *tile, *tileOffset, *groupTotal, *groupPrefix, *total;
};
static __global__ __launch_bounds__(RESORT_GROUP) void k_resort_scan_tiles(ResortScan a, ResortScan b, uint32_t *__restrict__ done,
                                                                     volatile uint64_t *hostTotal, uint32_t seq, uint32_t nTiles)
{
    // a = movers (always), b = dead slots (slab runs that leave holes; b.tile == nullptr otherwise).  The total of `a`
    // goes to the host.
    __shared__ uint32_t waveSum[RESORT_GROUP / 64];
    __shared__ bool last;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    auto block_exclusive = [&](uint32_t v, uint32_t &sum) { // exclusive prefix of v over the workgroup, sum = total
        uint32_t inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += o;
        }
        __syncthreads();
        if (lane == 63) waveSum[wave] = inc;
        __syncthreads();
        uint32_t base = 0, all = 0;
        for (uint32_t w = 0; w < RESORT_GROUP / 64; ++w) { const uint32_t c = waveSum[w]; if (w < wave) base += c; all += c; }
        sum = all;
        return base + inc - v;
    };
    const uint32_t t = blockIdx.x * RESORT_GROUP + tid;
    const ResortScan arr[2] = {a, b};
    for (int k = 0; k < 2; ++k) {
        if (!arr[k].tile) continue;
        uint32_t v = 0;
        if (t < nTiles) { v = arr[k].tile[t]; arr[k].tile[t] = 0; }
        uint32_t sum;
        const uint32_t ex = block_exclusive(v, sum);
        if (t < nTiles) arr[k].tileOffset[t] = ex;
        if (tid == 0) arr[k].groupTotal[blockIdx.x] = sum;
    }
    if (tid == 0) {
        __threadfence();
        last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // gridDim.x <= RESORT_GROUP (checked by the host): one more scan over the group totals
    for (int k = 0; k < 2; ++k) {
        if (!arr[k].tile) continue;
        const uint32_t g = tid < gridDim.x ? __atomic_load_n(&arr[k].groupTotal[tid], __ATOMIC_RELAXED) : 0u;
        uint32_t all;
        const uint32_t gex = block_exclusive(g, all);
        if (tid < gridDim.x) arr[k].groupPrefix[tid] = gex;
        if (tid == 0) {
            *arr[k].total = all;
            if (k == 0 && hostTotal) *hostTotal = ((uint64_t)seq << 32) | all; // one 8-byte store to mapped host memory: (launch number, count)
        }
    }
    if (tid == 0) *done = 0;
}

// stable split of slot i (tile = i / BLOCK) by "hash changed": movers[rank among movers], stayers[rank among stayers].
// HOLES (slab runs that do not compact their arrays): a slot whose next key is 0xffffffff is dead (its particle left
// the slab or was a halo copy) and goes to neither list; `dead` holds the scanned per-tile counts of such slots.
struct ResortOffsets { const uint32_t *tileOffset, *groupPrefix; };
template <bool HOLES>
__global__ __launch_bounds__(BLOCK) void k_resort_split(const uint32_t *__restrict__ prevHash, const uint32_t *__restrict__ nextHash,
                                                        ResortOffsets mov, ResortOffsets dead, uint64_t *__restrict__ movers,
                                                        uint64_t *__restrict__ stayers, uint32_t n, uint32_t *__restrict__ clearCells)
{
    // clearCells (cellStart, or null): also undo the cell table of the step that just ended — the work of k_clear_cells
    // (nrs_kernels_ref.h), folded in here because this kernel reads the step's sorted keys anyway and runs after every
    // reader of the table
    __shared__ uint32_t waveCount[2][BLOCK / 64];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t i = tile * BLOCK + tid;
    const bool live = i < n;
    uint32_t k = 0;
    bool mover = false, hole = false;
    if (live) {
        k = nextHash[i];
        hole = HOLES && k == 0xffffffffu;
        const uint32_t prev = prevHash[i];
        mover = !hole && k != prev;
        if (clearCells && (i == 0 || prev != prevHash[i - 1])) clearCells[prev] = CELL_EMPTY;
    }
    const uint64_t mask = __ballot(mover), hmask = HOLES ? __ballot(hole) : 0ull;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    if (lane == 0) { waveCount[0][wave] = (uint32_t)__popcll(mask); waveCount[1][wave] = (uint32_t)__popcll(hmask); }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t before = (uint32_t)__popcll(mask & below), hbefore = (uint32_t)__popcll(hmask & below);
    for (uint32_t w = 0; w < wave; ++w) { before += waveCount[0][w]; hbefore += waveCount[1][w]; }
    if (!live || hole) return;
    const uint32_t moversBefore = mov.groupPrefix[tile / RESORT_GROUP] + mov.tileOffset[tile] + before;
    const uint32_t holesBefore = HOLES ? dead.groupPrefix[tile / RESORT_GROUP] + dead.tileOffset[tile] + hbefore : 0u;
    const uint64_t e = ((uint64_t)k << 32) | i;
    if (mover) movers[moversBefore] = e;
    else stayers[i - moversBefore - holesBefore] = e;
}

// stable compaction of (pos, vel) by "slot is live", used when somebody looks at the arrays while they have holes
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_holes_compact(const uint32_t *__restrict__ keys, ResortOffsets dead,
                                                         const typename Vec4T<R>::type *__restrict__ pos,
                                                         const typename Vec4T<R>::type *__restrict__ vel,
                                                         typename Vec4T<R>::type *__restrict__ outPos,
                                                         typename Vec4T<R>::type *__restrict__ outVel, uint32_t n)
{
    __shared__ uint32_t waveCount[BLOCK / 64];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t i = tile * BLOCK + tid;
    const bool hole = i < n && keys[i] == 0xffffffffu;
    const uint64_t hmask = __ballot(hole);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    if (lane == 0) waveCount[wave] = (uint32_t)__popcll(hmask);
    __syncthreads();
    uint32_t hbefore = (uint32_t)__popcll(hmask & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; ++w) hbefore += waveCount[w];
    if (i >= n || hole) return;
    const uint32_t d = i - (dead.groupPrefix[tile / RESORT_GROUP] + dead.tileOffset[tile] + hbefore);
    outPos[d] = pos[i];
    outVel[d] = vel[i];
}
// per-tile count of dead slots (input of the scan that k_holes_compact needs)
static __global__ __launch_bounds__(BLOCK) void k_holes_count(const uint32_t *__restrict__ keys, uint32_t *__restrict__ tileDead, uint32_t n)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t m = __ballot(i < n && keys[i] == 0xffffffffu);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&tileDead[blockIdx.x], (uint32_t)__popcll(m));
}

// reorderDataAndFindCellStartD (sph_kernel_impl.cuh:210-281) fed by the merged u64 pairs; also writes the plain
// sorted hash / index arrays the rest of the step (and nrs_get_array) use.
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_reorder_merged(const uint64_t *__restrict__ merged, uint32_t *__restrict__ hashOut,
                                                          uint32_t *__restrict__ indexOut,
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
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t e = i < n ? merged[i] : 0ull;
    const uint32_t h = (uint32_t)(e >> 32), src = (uint32_t)e;
    typename Vec4T<R>::type p4 = mk4<R>((R)0, (R)0, (R)0, (R)0);
    if (i < n) p4 = oldPos[src];
    if (nearBits) wall_tile_count(nearBits, wallTileCount, h, i < n, wallMask, qpos && quant_far<R>(qc, xyz<R>(p4))); // (far owners: see k_reorder)
    if (i >= n) return;
    if (i == 0) {
        cellStart[h] = 0;
    } else {
        const uint32_t hp = (uint32_t)(merged[i - 1] >> 32);
        if (h != hp) { cellStart[h] = i; cellEnd[hp] = i; }
    }
    if (i == n - 1) cellEnd[h] = n;
    hashOut[i] = h;
    indexOut[i] = src;
    sPos[i] = p4;
    if (qpos) qpos[i] = quantize_pos<R>(qc, xyz<R>(p4));
    sVel[i] = oldVel[src];
    if (oldPres) sPres[i] = oldPres[src];
    if (inv) inv[src] = i;
}

} // namespace nrs
