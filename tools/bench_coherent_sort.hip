// Micro-benchmark: radix sort of (hash,index) vs. "coherent" re-sort (partition stayers/movers, sort movers, merge).
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/bench_coherent_sort.hip -o tools/_bin/bench_coherent_sort
// run  : tools/_bin/bench_coherent_sort [n] [mover fraction]
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../nereus_amd/csrc/nrs_kernels_resort.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using SortCfg9 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, 9,
                                                                                rocprim::block_radix_rank_algorithm::match>>;

using SortCfgS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 4>, rocprim::kernel_config<256, 4>, 9,
                                                                                rocprim::block_radix_rank_algorithm::match>, 8192>;
using SortCfgM = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 8>, rocprim::kernel_config<256, 8>, 9,
                                                                                rocprim::block_radix_rank_algorithm::match>, 8192>;
using SortCfgL = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, 9,
                                                                                rocprim::block_radix_rank_algorithm::match>, 8192>;
struct Compose {
    const uint32_t *next;
    __host__ __device__ uint64_t operator()(uint32_t i) const { return ((uint64_t)next[i] << 32) | i; }
};
struct IsMover {
    const uint32_t *old;
    __host__ __device__ bool operator()(uint64_t v) const { return old[(uint32_t)v] != (uint32_t)(v >> 32); }
};

// stand-in for the fused force kernel's epilogue: movers per 256-slot tile
__global__ void k_count(const uint32_t *prev, const uint32_t *next, uint32_t *tileMovers, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mover = i < n && prev[i] != next[i];
    const uint64_t mask = __ballot(mover);
    if ((threadIdx.x & 63u) == 0 && mask) atomicAdd(&tileMovers[blockIdx.x], (uint32_t)__popcll(mask));
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 10077696u;
    const double frac = argc > 2 ? atof(argv[2]) : 0.03;
    const uint32_t bits = 27, gx = 1024, gy = 512;
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t mcap = n; // any share of movers
    std::mt19937_64 rng(12345);
    std::vector<uint32_t> oldh(n), nexth(n), idx(n);
    for (uint32_t i = 0; i < n; ++i) oldh[i] = (uint32_t)(rng() % (1u << bits)) & ~0x3u; // ~2.4 per used cell
    std::sort(oldh.begin(), oldh.end());
    const int64_t deltas[6] = {1, -1, gx, -(int64_t)gx, (int64_t)gx * gy, -(int64_t)gx * gy};
    std::uniform_real_distribution<double> U(0, 1);
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        nexth[i] = oldh[i];
        idx[i] = i;
        if (U(rng) < frac) { nexth[i] = (uint32_t)((int64_t)oldh[i] + deltas[rng() % 6]) & ((1u << bits) - 1); if (nexth[i] != oldh[i]) ++m; }
    }
    // expected result on the host
    std::vector<uint64_t> expect(n);
    for (uint32_t i = 0; i < n; ++i) expect[i] = ((uint64_t)nexth[i] << 32) | i;
    std::sort(expect.begin(), expect.end());

    uint32_t *dOld, *dNext, *dIdx, *dK2, *dV2, *dCount;
    uint64_t *dMov, *dMov2, *dStay, *dOut;
    CHK(hipMalloc(&dOld, 4ull * n)); CHK(hipMalloc(&dNext, 4ull * n)); CHK(hipMalloc(&dIdx, 4ull * n));
    CHK(hipMalloc(&dK2, 4ull * n)); CHK(hipMalloc(&dV2, 4ull * n)); CHK(hipMalloc(&dCount, 4));
    CHK(hipMalloc(&dMov, 8ull * n)); CHK(hipMalloc(&dMov2, 8ull * n)); CHK(hipMalloc(&dStay, 8ull * n));
    CHK(hipMalloc(&dOut, 8ull * (n + mcap)));
    CHK(hipMemcpy(dOld, oldh.data(), 4ull * n, hipMemcpyHostToDevice));
    hipStream_t st = nullptr;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    void *tmp = nullptr;
    size_t tmpBytes = 0, need = 0;

    // ---- (a) radix sort pairs, 9-bit passes
    {
        rocprim::double_buffer<uint32_t> k(dNext, dK2), v(dIdx, dV2);
        CHK(rocprim::radix_sort_pairs<SortCfg9>(nullptr, need, k, v, n, 0u, bits, st));
        tmpBytes = need;
    }
    {
        rocprim::double_buffer<uint64_t> k(dMov, dMov2);
        CHK(rocprim::radix_sort_keys<SortCfg9>(nullptr, need, k, mcap, 32u, 32u + bits, st));
        tmpBytes = std::max(tmpBytes, need);
        CHK(rocprim::radix_sort_keys<SortCfgS>(nullptr, need, k, mcap, 32u, 32u + bits, st));
        tmpBytes = std::max(tmpBytes, need);
        CHK(rocprim::radix_sort_keys<SortCfgM>(nullptr, need, k, mcap, 32u, 32u + bits, st));
        tmpBytes = std::max(tmpBytes, need);
        CHK(rocprim::radix_sort_keys<SortCfgL>(nullptr, need, k, mcap, 32u, 32u + bits, st));
        tmpBytes = std::max(tmpBytes, need);
    }
    CHK(rocprim::merge(nullptr, need, dStay, dMov, dOut, n, mcap, rocprim::less<uint64_t>(), st));
    tmpBytes = std::max(tmpBytes, need);
    CHK(hipMalloc(&tmp, tmpBytes));
    printf("n %u movers %u (%.2f%%) mcap %u tmp %zu MB\n", n, m, 100.0 * m / n, mcap, tmpBytes >> 20);

    const int reps = 10;
    float ms;
    float tRadix = 0;
    for (int r = 0; r < reps + 1; ++r) {
        CHK(hipMemcpy(dNext, nexth.data(), 4ull * n, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dIdx, idx.data(), 4ull * n, hipMemcpyHostToDevice));
        rocprim::double_buffer<uint32_t> k(dNext, dK2), v(dIdx, dV2);
        size_t b = tmpBytes;
        CHK(hipEventRecord(e0, st));
        CHK(rocprim::radix_sort_pairs<SortCfg9>(tmp, b, k, v, n, 0u, bits, st));
        CHK(hipEventRecord(e1, st));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r) tRadix += ms;
        if (r == reps) {
            std::vector<uint32_t> hk(n), hv(n);
            CHK(hipMemcpy(hk.data(), k.current(), 4ull * n, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(hv.data(), v.current(), 4ull * n, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (uint32_t i = 0; i < n; ++i) bad += ((((uint64_t)hk[i] << 32) | hv[i]) != expect[i]);
            printf("radix  : %.1f us  mismatches %zu\n", 1e3 * tRadix / reps, bad);
        }
    }
    CHK(hipMemcpy(dNext, nexth.data(), 4ull * n, hipMemcpyHostToDevice));
    float tPart = 0, tSort = 0, tMerge = 0, tAll = 0;
    hipEvent_t ev[5];
    for (auto &e : ev) CHK(hipEventCreate(&e));
    const uint32_t nTiles = (n + 255) / 256;
    uint32_t *dTileMovers, *dTileOffset;
    uint64_t *hTotal;
    CHK(hipMalloc(&dTileMovers, 4ull * nTiles)); CHK(hipMalloc(&dTileOffset, 4ull * nTiles));
    CHK(hipMemset(dTileMovers, 0, 4ull * nTiles));
    const uint32_t nGroups = (nTiles + nrs::RESORT_GROUP - 1) / nrs::RESORT_GROUP;
    uint32_t *dGroupTotal, *dGroupPrefix, *dDone;
    CHK(hipMalloc(&dGroupTotal, 4ull * nGroups)); CHK(hipMalloc(&dGroupPrefix, 4ull * nGroups)); CHK(hipMalloc(&dDone, 4));
    CHK(hipMemset(dDone, 0, 4));
    CHK(hipHostMalloc(&hTotal, 8, hipHostMallocMapped));
    uint64_t *hTotalDev = nullptr;
    CHK(hipHostGetDevicePointer((void **)&hTotalDev, hTotal, 0));
    hipEvent_t evM;
    CHK(hipEventCreateWithFlags(&evM, hipEventDisableTiming));
    for (int r = 0; r < reps + 1; ++r) {
        size_t b = tmpBytes;
        k_count<<<nTiles, 256, 0, st>>>(dOld, dNext, dTileMovers, n);
        CHK(hipEventRecord(ev[0], st));
        const nrs::ResortScan sc = {dTileMovers, dTileOffset, dGroupTotal, dGroupPrefix, dCount}, none = {nullptr, nullptr, nullptr, nullptr, nullptr};
        nrs::k_resort_scan_tiles<<<nGroups, nrs::RESORT_GROUP, 0, st>>>(sc, none, dDone, hTotalDev, (uint32_t)(r + 1), nTiles);
        CHK(hipEventRecord(evM, st));
        const nrs::ResortOffsets off = {dTileOffset, dGroupPrefix};
        nrs::k_resort_split<false><<<nTiles, 256, 0, st>>>(dOld, dNext, off, off, dMov, dStay, n, (uint32_t *)nullptr);
        CHK(hipEventRecord(ev[1], st));
        CHK(hipEventSynchronize(evM));
        const uint32_t M = (uint32_t)*hTotal;
        if (r == 0) printf("host sees %u movers\n", M);
        rocprim::double_buffer<uint64_t> k(dMov, dMov2);
        if (variant == 0) CHK(rocprim::radix_sort_keys<SortCfg9>(tmp, b, k, M, 32u, 32u + bits, st));
        else if (variant == 1) CHK(rocprim::radix_sort_keys<SortCfgS>(tmp, b, k, M, 32u, 32u + bits, st));
        else if (variant == 2) CHK(rocprim::radix_sort_keys<SortCfgM>(tmp, b, k, M, 32u, 32u + bits, st));
        else CHK(rocprim::radix_sort_keys<SortCfgL>(tmp, b, k, M, 32u, 32u + bits, st));
        CHK(hipEventRecord(ev[3], st));
        b = tmpBytes;
        CHK(rocprim::merge(tmp, b, dStay, k.current(), dOut, n - M, M, rocprim::less<uint64_t>(), st));
        CHK(hipEventRecord(ev[4], st));
        CHK(hipEventSynchronize(ev[4]));
        if (r) {
            CHK(hipEventElapsedTime(&ms, ev[0], ev[1])); tPart += ms;
            CHK(hipEventElapsedTime(&ms, ev[1], ev[3])); tSort += ms;
            CHK(hipEventElapsedTime(&ms, ev[3], ev[4])); tMerge += ms;
            CHK(hipEventElapsedTime(&ms, ev[0], ev[4])); tAll += ms;
        }
    }
    std::vector<uint64_t> got(n);
    CHK(hipMemcpy(got.data(), dOut, 8ull * n, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += got[i] != expect[i];
    printf("coherent: %.1f us (scan+split %.1f, mover sort %.1f, merge %.1f)  mismatches %zu\n", 1e3 * tAll / reps,
           1e3 * tPart / reps, 1e3 * tSort / reps, 1e3 * tMerge / reps, bad);
    return bad ? 1 : 0;
}
