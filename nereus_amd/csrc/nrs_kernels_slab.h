// nrs_kernels_slab.h — device side of the multi-GPU slab decomposition (new: the reference is
// single-GPU).  Each rank owns the particles whose global grid cell-x lies in [lo, hi).  Once per step, before
// update(), every rank partitions its current particles into six streams with ONE stable multi-way
// compaction (count → scan → scatter; deterministic, no atomics on the output order):
//
//   STAY    cell-x in [lo,hi)                      → stays owned (compacted to the front of the local arrays)
//   MIG_L/R cell-x < lo / >= hi                    → ownership moves to the left / right neighbour
//   HALO_L/R STAY particles within `halo` cells of the left / right cut → copied to that neighbour as read-only halo
//   GHOST   migrants still within `halo` cells of our cut → we keep a read-only copy (they are the neighbour's now)
//
// Halo/ghost copies carry pos.w = 2 (owned particles have w = 1, as the reference sets it, sph/sph.cpp:381):
// that is how they are recognised and dropped by the next partition, after the sort has interleaved them with
// owned particles.  With a halo of two cells the density of every particle within one cell of a cut is complete
// locally, so the force on every OWNED particle is exact without a second message per step.
//
// Message buffer per direction (device memory owned by the caller, e.g. a torch tensor sent with RCCL):
//   [ u32 nMigrants, u32 nHalo, u32 0, u32 0 | vec4 pos[cap] | vec4 vel[cap] ]   migrants first, then halo.
#pragma once
#include "nrs_math.h"

namespace nrs {

enum { ST_STAY = 0, ST_MIG_L = 1, ST_HALO_L = 2, ST_MIG_R = 3, ST_HALO_R = 4, ST_GHOST = 5, ST_COUNT = 6,
       ST_CHANGED = 6,  // counted only (no output stream): STAY particles whose grid cell changed in the last step
       ST_TOTALS = 7 };
constexpr int SLAB_BLOCK = 256;
constexpr int SLAB_ITERS = 8;                       // sub-tiles per workgroup: 2048 particles per workgroup keeps the block-
constexpr int SLAB_TILE = SLAB_BLOCK * SLAB_ITERS;  // offset scan short (it was the most expensive slab kernel at 256)

struct SlabCfg { int lo, hi, halo; };

template <typename R> NRS_DEV uint32_t slab_flags(const Params<R> &P, const SlabCfg c, const typename Vec4T<R>::type q)
{
    if (!(q.w == (R)1)) return 0u; // last step's halo / ghost copies are dropped here
    const long long cx = (long long)floor((q.x - P.worldOrigin[0]) / P.cellSize[0]);
    uint32_t f = 0;
    if (cx < c.lo) {
        f = 1u << ST_MIG_L;
        if (cx >= (long long)c.lo - c.halo) f |= 1u << ST_GHOST;
    } else if (cx >= c.hi) {
        f = 1u << ST_MIG_R;
        if (cx < (long long)c.hi + c.halo) f |= 1u << ST_GHOST;
    } else {
        f = 1u << ST_STAY;
        if (cx < (long long)c.lo + c.halo) f |= 1u << ST_HALO_L;
        if (cx >= (long long)c.hi - c.halo) f |= 1u << ST_HALO_R;
    }
    return f;
}

// pass 1: per-workgroup population of every stream
template <typename R>
__global__ __launch_bounds__(SLAB_BLOCK) void k_slab_count(Params<R> P, SlabCfg c, const typename Vec4T<R>::type *__restrict__ pos,
                                                          uint32_t n, uint32_t *__restrict__ blockCounts, uint32_t nBlocks,
                                                          const uint32_t *__restrict__ prevHash, const uint32_t *__restrict__ nextHash)
{
    // prevHash / nextHash (both or neither): sorted keys of the last step and the keys its fused force kernel computed
    // for the new positions, per slot — lets the host know how many particles the coherent re-sort has to sort
    __shared__ uint32_t cnt[ST_TOTALS];
    if (threadIdx.x < ST_TOTALS) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int it = 0; it < SLAB_ITERS; ++it) {
        const uint32_t i = blockIdx.x * SLAB_TILE + it * SLAB_BLOCK + threadIdx.x;
        uint32_t f = i < n ? slab_flags<R>(P, c, pos[i]) : 0u;
        if (prevHash && (f & (1u << ST_STAY)) && prevHash[i] != nextHash[i]) f |= 1u << ST_CHANGED;
        for (int s = 0; s < ST_TOTALS; ++s) {
            const unsigned long long m = __ballot((f >> s) & 1u);
            if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt[s], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < ST_TOTALS) blockCounts[threadIdx.x * nBlocks + blockIdx.x] = cnt[threadIdx.x];
}

// pass 2: exclusive scan of the block counts, one workgroup per stream; totals[s] = stream population
static __global__ __launch_bounds__(SLAB_BLOCK) void k_slab_scan(uint32_t *__restrict__ blockCounts, uint32_t nBlocks,
                                                          uint32_t *__restrict__ totals)
{
    __shared__ uint32_t waveSum[SLAB_BLOCK / 64];
    uint32_t *row = blockCounts + (size_t)blockIdx.x * nBlocks;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nBlocks; base += SLAB_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nBlocks ? row[i] : 0u;
        uint32_t inc = v; // inclusive scan inside the wave, then across the 4 waves
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += o;
        }
        __syncthreads();
        if (lane == 63) waveSum[wave] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < SLAB_BLOCK / 64; ++w) { const uint32_t c = waveSum[w]; if (w < wave) before += c; all += c; }
        if (i < nBlocks) row[i] = carry + before + inc - v;
        carry += all;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <typename R> struct SlabOut {
    typedef typename Vec4T<R>::type T4;
    T4 *stayPos, *stayVel;   // compacted owned particles
    uint32_t *hash, *index;  // radix-sort keys/values of the next step for the compacted particles (saves a hash pass)
    const uint32_t *prevHash; // sorted keys of the step that produced `pos` (slot order), or null
    uint32_t *prevPacked;     // their compacted copy for the coherent re-sort (nrs_kernels_resort.h), or null
    uint32_t *tileMovers;     // with prevPacked: per 256-slot tile of the COMPACTED array, slots whose key changed
    // INPLACE partition (owned particles are not moved; `hash` holds the keys the fused force kernel computed per slot):
    uint32_t *tileDead;       // per 256-slot tile: slots that hold no particle of the next step (left, or old halo copy)
    const uint8_t *flags;     // FROM_FLAGS: stream flags per slot, written by the fused force kernel of the step
    T4 *ghostPos, *ghostVel; // our read-only copies of fresh migrants
    unsigned char *sendL, *sendR; // message buffers (may be null at the ends of the chain)
    uint32_t cap;            // particles per message buffer
};

template <typename R> NRS_DEV typename Vec4T<R>::type *msg_pos(unsigned char *buf) { return (typename Vec4T<R>::type *)(buf + 16); }
template <typename R> NRS_DEV typename Vec4T<R>::type *msg_vel(unsigned char *buf, uint32_t cap)
{
    return (typename Vec4T<R>::type *)(buf + 16 + (size_t)cap * sizeof(typename Vec4T<R>::type));
}

// pass 3: stable scatter of every stream (also hashes the particles that stay, for the next step's sort).
// INPLACE: the particles that stay are not moved at all — their slot keeps its key (out.hash[i], computed by the fused
// force kernel), dead slots get the key 0xffffffff, and the per-tile counts of cell changers / dead slots feed the
// coherent re-sort (nrs_kernels_resort.h), whose merged order then skips the holes.
// FROM_FLAGS (with INPLACE): the fused force kernel has already classified every slot, marked the dead ones and
// counted the streams; only the particles of the message / ghost streams are touched here.
template <typename R, bool INPLACE, bool FROM_FLAGS = false>
__global__ __launch_bounds__(SLAB_BLOCK) void k_slab_scatter(Params<R> P, SlabCfg c, const typename Vec4T<R>::type *__restrict__ pos,
                                                            const typename Vec4T<R>::type *__restrict__ vel, uint32_t n,
                                                            const uint32_t *__restrict__ blockOffsets, uint32_t nBlocks,
                                                            const uint32_t *__restrict__ totals, SlabOut<R> out)
{
    typedef typename Vec4T<R>::type T4;
    __shared__ uint32_t waveCnt[ST_COUNT][SLAB_BLOCK / 64];
    __shared__ uint32_t run[ST_COUNT]; // particles of each stream already emitted by this workgroup
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < ST_COUNT) run[threadIdx.x] = 0;
    for (int it = 0; it < SLAB_ITERS; ++it) {
        const uint32_t i = blockIdx.x * SLAB_TILE + it * SLAB_BLOCK + threadIdx.x;
        T4 p, v;
        uint32_t f = 0;
        if (i < n) {
            if (FROM_FLAGS) {
                f = out.flags[i];
                if (f & ~(1u << ST_STAY)) { p = pos[i]; v = vel[i]; }
            } else {
                p = pos[i]; v = vel[i]; f = slab_flags<R>(P, c, p);
            }
        }
        uint32_t rankInWave[ST_COUNT];
        for (int s = 0; s < ST_COUNT; ++s) {
            const unsigned long long m = __ballot((f >> s) & 1u);
            rankInWave[s] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) waveCnt[s][wave] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (INPLACE && !FROM_FLAGS && i < n) {
            if (!(f & (1u << ST_STAY))) {
                out.hash[i] = 0xffffffffu;
                atomicAdd(&out.tileDead[i / SLAB_BLOCK], 1u);
            } else if (out.hash[i] != out.prevHash[i]) {
                atomicAdd(&out.tileMovers[i / SLAB_BLOCK], 1u);
            }
        }
        if (f) {
            T4 tagged = p;
            tagged.w = (R)2; // read-only copy
            for (int s = 0; s < ST_COUNT; ++s) {
                if (!((f >> s) & 1u)) continue;
                uint32_t idx = blockOffsets[(size_t)s * nBlocks + blockIdx.x] + run[s] + rankInWave[s];
                for (uint32_t w = 0; w < wave; ++w) idx += waveCnt[s][w];
                switch (s) {
                case ST_STAY: {
                    if (INPLACE) break;
                    out.stayPos[idx] = p; out.stayVel[idx] = v;
                    const I3 g = calcGridPos<R>(P, xyz<R>(p));
                    const uint32_t key = calcGridHash<R>(P, g.x, g.y, g.z);
                    out.hash[idx] = key;
                    out.index[idx] = idx;
                    if (out.prevPacked) {
                        const uint32_t prev = out.prevHash ? out.prevHash[i] : 0xffffffffu;
                        out.prevPacked[idx] = prev;
                        if (prev != key) atomicAdd(&out.tileMovers[idx / SLAB_BLOCK], 1u);
                    }
                    break;
                }
                case ST_GHOST: // (the ghost arrays hold out.cap particles; the host reports the overflow from the totals)
                    if (idx < out.cap) { out.ghostPos[idx] = tagged; out.ghostVel[idx] = v; }
                    break;
                case ST_MIG_L:
                    if (out.sendL && idx < out.cap) { msg_pos<R>(out.sendL)[idx] = p; msg_vel<R>(out.sendL, out.cap)[idx] = v; }
                    break;
                case ST_HALO_L:
                    idx += totals[ST_MIG_L];
                    if (out.sendL && idx < out.cap) { msg_pos<R>(out.sendL)[idx] = tagged; msg_vel<R>(out.sendL, out.cap)[idx] = v; }
                    break;
                case ST_MIG_R:
                    if (out.sendR && idx < out.cap) { msg_pos<R>(out.sendR)[idx] = p; msg_vel<R>(out.sendR, out.cap)[idx] = v; }
                    break;
                case ST_HALO_R:
                    idx += totals[ST_MIG_R];
                    if (out.sendR && idx < out.cap) { msg_pos<R>(out.sendR)[idx] = tagged; msg_vel<R>(out.sendR, out.cap)[idx] = v; }
                    break;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < ST_COUNT) {
            uint32_t t = 0;
            for (int w = 0; w < SLAB_BLOCK / 64; ++w) t += waveCnt[threadIdx.x][w];
            run[threadIdx.x] += t;
        }
        __syncthreads();
    }
}

static __global__ void k_slab_headers(const uint32_t *__restrict__ totals, unsigned char *sendL, unsigned char *sendR)
{
    if (threadIdx.x == 0 && sendL) {
        uint32_t *h = (uint32_t *)sendL;
        h[0] = totals[ST_MIG_L]; h[1] = totals[ST_HALO_L]; h[2] = 0; h[3] = 0;
    }
    if (threadIdx.x == 1 && sendR) {
        uint32_t *h = (uint32_t *)sendR;
        h[0] = totals[ST_MIG_R]; h[1] = totals[ST_HALO_R]; h[2] = 0; h[3] = 0;
    }
}

// owned particles per grid cell-x column, bins [lo0, lo0 + nbins): input of the count-balanced re-cut
template <typename R>
__global__ __launch_bounds__(SLAB_BLOCK) void k_slab_histogram(Params<R> P, const typename Vec4T<R>::type *__restrict__ pos, uint32_t n,
                                                              int lo0, uint32_t nbins, uint32_t *__restrict__ bins)
{
    const uint32_t i = blockIdx.x * SLAB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const typename Vec4T<R>::type q = pos[i];
    if (!(q.w == (R)1)) return;
    const long long cx = (long long)floor((q.x - P.worldOrigin[0]) / P.cellSize[0]) - lo0;
    if (cx >= 0 && cx < (long long)nbins) atomicAdd(&bins[cx], 1u);
}

// append up to five (source, count) pieces behind the compacted owned particles
template <typename R> struct AppendPieces {
    typedef typename Vec4T<R>::type T4;
    const T4 *srcPos[5];
    const T4 *srcVel[5];
    uint32_t start[6]; // exclusive prefix of the piece lengths; start[5] = total
};
template <typename R>
__global__ __launch_bounds__(SLAB_BLOCK) void k_slab_append(Params<R> P, AppendPieces<R> A, typename Vec4T<R>::type *__restrict__ dstPos,
                                                           typename Vec4T<R>::type *__restrict__ dstVel, uint32_t *__restrict__ hash,
                                                           uint32_t *__restrict__ index, uint32_t *__restrict__ prevPacked,
                                                           uint32_t *__restrict__ tileMovers, uint32_t dstBase,
                                                           uint64_t *__restrict__ movers, uint32_t moverBase)
{
    const uint32_t i = blockIdx.x * SLAB_BLOCK + threadIdx.x;
    if (i >= A.start[5]) return;
    int k = 0;
    while (k < 4 && i >= A.start[k + 1]) ++k;
    const uint32_t j = i - A.start[k];
    const typename Vec4T<R>::type p = A.srcPos[k][j];
    dstPos[dstBase + i] = p;
    dstVel[dstBase + i] = A.srcVel[k][j];
    const I3 g = calcGridPos<R>(P, xyz<R>(p));
    const uint32_t key = calcGridHash<R>(P, g.x, g.y, g.z);
    hash[dstBase + i] = key;
    index[dstBase + i] = dstBase + i;
    // in-place partition: every arrival goes straight to the tail of the mover list of the coherent re-sort
    if (movers) movers[moverBase + i] = ((uint64_t)key << 32) | (dstBase + i);
    if (prevPacked) { // new to this rank's arrays: never a "stayer" of the coherent re-sort
        prevPacked[dstBase + i] = 0xffffffffu;
        atomicAdd(&tileMovers[(dstBase + i) / SLAB_BLOCK], 1u);
    }
}

} // namespace nrs
