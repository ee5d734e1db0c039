// mfma_tile_cost.hip — bounded experiment behind DESIGN.md section 4 "matrix-core distance tiles" (VERDICT r2 item 3).
//
// Question: can the neighbourhood scan's superset test (integer squared distance of 10-bit quantised coordinates against a
// threshold) be moved from the vector ALU to the matrix cores?  For a tile of 32 owners x 32 candidates,
//     d2 = |a|^2 + |b|^2 - 2 a.b
// is ONE v_mfma_f32_32x32x16_f16 (coordinates are integers < 2048: exact in f16; the norms are split hi * 2048 + lo across spare
// k-slots; every partial sum is an integer < 2^24: exact in the fp32 accumulators).  What the matrix core cannot do is hand the
// hits back: the 1024 distances of a tile land in 16 accumulator registers per lane and have to be compared and gathered into
// per-owner bit masks by vector instructions (v_cmp + v_addc per register — the cheapest form found).
//
// This tool measures, on the device, wave-cycles per OWNER-ROW (one owner against the candidates of one of its nine neighbour rows)
//   direct : the production form — every lane is an owner and tests 8 candidate slots of the row (Sweep::scan_compact's test, the
//            branch-free LDS append included)
//   mfma   : 32 owners x 32 candidates per tile: operand build from the same 4-byte words, the MFMA, compare + mask build, and
//            the append of the hits found (mask decode)
// with identical inputs and checked against each other (same hit sets).  A row of 32 consecutive owners spans ~24 cells + 2 = ~34-37
// candidates, i.e. ONE to TWO tiles per owner-row; the direct form needs 8 slots per owner-row.
//
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mfma_tile_cost.hip -o tools/_bin/mfma_tile_cost ; run: tools/_bin/mfma_tile_cost
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int WAVES_PER_BLOCK = 4;
constexpr uint32_t GUARD = (1u << 10) | (1u << 21);

// quantised word: x bits 0-9, y bits 11-20, z bits 22-31 (as nrs_math.h, quantize_pos)
__host__ __device__ inline uint32_t packq(uint32_t x, uint32_t y, uint32_t z) { return x | (y << 11) | (z << 22); }

__device__ inline int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---- direct: lane = owner, 8 candidate slots per row (two dwordx4 loads), branch-free append of the hits into LDS ----------------
// tiles[t]: 32 owner words then 32 candidate words; a wave takes two tiles' owners (64 lanes) and, for parity of the tested set with
// the mfma kernel, its 8 slots are candidates 4*(lane&7) .. +7 (mod 32) of the lane's own tile — the arithmetic per slot is exactly
// the production scan's, which is what is being timed.
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_direct(const uint32_t *__restrict__ tiles, uint32_t nTilePairs, uint32_t qT, uint32_t rows,
                                                                 uint32_t *__restrict__ hitCount, unsigned long long *__restrict__ cycles)
{
    __shared__ uint32_t lst[24][64 * WAVES_PER_BLOCK];
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= nTilePairs) return;
    const uint32_t *T = tiles + (size_t)(2u * wave + (lane >> 5)) * 64u;
    const uint32_t Qi = T[lane & 31u] | GUARD;
    const uint32_t col = threadIdx.x * 4u, rowBytes = 64u * WAVES_PER_BLOCK * 4u, capOff = col + 22u * rowBytes;
    char *const lbase = reinterpret_cast<char *>(&lst[0][0]);
    uint32_t cur = col, total = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t r = 0; r < rows; ++r) {
        const uint32_t first = (4u * (lane & 7u) + r) & 31u;
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = T[32u + ((first + (uint32_t)u) & 31u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if ((u & 1) == 0) cur = min(cur, capOff);
            const uint32_t t = Qi - c[u];
            const int dx = ((int)(t << 22)) >> 22, dy = ((int)(t << 11)) >> 22, dz = ((int)t) >> 22;
            const uint32_t d2 = (uint32_t)mad24(dz, dz, mad24(dy, dy, __mul24(dx, dx)));
            const bool hit = d2 < qT;
            *reinterpret_cast<uint32_t *>(lbase + cur) = (r << 8) + (uint32_t)u;
            cur += hit ? rowBytes : 0u;
        }
        total += (cur - col) / rowBytes; // (drain the list every row so that it never overflows: the append is what is timed)
        cur = col;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    hitCount[(size_t)wave * 64u + lane] = total;
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

// ---- mfma: one 32 x 32 tile per row and wave ----------------------------------------------------------------------------------
__device__ inline half8 operandA(uint32_t w, bool lower)
{
    // k-slots (lane half h = 0): ax, ay, az, na_hi, na_lo, 2048, 1, 0 ; lane half 1 (k = 8..15): zeros
    const int x = (int)(w & 1023u), y = (int)((w >> 11) & 1023u), z = (int)(w >> 22);
    const int n = x * x + y * y + z * z;
    half8 a = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lower) {
        a[0] = (_Float16)x; a[1] = (_Float16)y; a[2] = (_Float16)z;
        a[3] = (_Float16)(n >> 11); a[4] = (_Float16)(n & 2047);
        a[5] = (_Float16)2048.0f; a[6] = (_Float16)1.0f;
    }
    return a;
}
__device__ inline half8 operandB(uint32_t w, bool lower)
{
    // k-slots: -2bx, -2by, -2bz, 2048, 1, nb_hi, nb_lo, 0
    const int x = (int)(w & 1023u), y = (int)((w >> 11) & 1023u), z = (int)(w >> 22);
    const int n = x * x + y * y + z * z;
    half8 b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lower) {
        b[0] = (_Float16)(-2 * x); b[1] = (_Float16)(-2 * y); b[2] = (_Float16)(-2 * z);
        b[3] = (_Float16)2048.0f; b[4] = (_Float16)1.0f;
        b[5] = (_Float16)(n >> 11); b[6] = (_Float16)(n & 2047);
    }
    return b;
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_mfma(const uint32_t *__restrict__ tiles, uint32_t nTiles, float qT, uint32_t rows,
                                                               uint32_t *__restrict__ hitCount, unsigned long long *__restrict__ cycles)
{
    __shared__ uint32_t lst[24][64 * WAVES_PER_BLOCK];
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= nTiles) return;
    const uint32_t *T = tiles + (size_t)wave * 64u;
    const bool lower = lane < 32u;
    // owners are the COLUMNS of the result (col = lane & 31): owners go into B, candidates into A (rows)
    const half8 own = operandB(T[lane & 31u], lower);
    const uint32_t col = threadIdx.x * 4u, rowBytes = 64u * WAVES_PER_BLOCK * 4u;
    char *const lbase = reinterpret_cast<char *>(&lst[0][0]);
    uint32_t total = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t r = 0; r < rows; ++r) {
        const half8 cand = operandA(T[32u + ((lane + r) & 31u)], lower); // (row permutation per r: keeps the compiler from hoisting)
        float16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cand, own, acc, 0, 0, 0);
        // lane (col, half) holds rows (reg & 3) + 8 (reg >> 2) + 4 half of its owner's column: compare and shift into a 16-bit mask
        uint32_t m = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v)
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(acc[v]), "v"(qT) : "vcc");
        // decode: append the hits of this lane's mask to its LDS column (the owner's list is the two half-lanes' lists, concatenated later)
        uint32_t cur = col;
        while (m) {
            const uint32_t b = 31u - (uint32_t)__builtin_clz(m);
            *reinterpret_cast<uint32_t *>(lbase + cur) = (r << 8) + b;
            cur += rowBytes;
            m &= ~(1u << b);
        }
        total += (cur - col) / rowBytes;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    hitCount[(size_t)wave * 64u + lane] = total;
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

int main(int argc, char **argv)
{
    const uint32_t nTiles = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144u, rows = 9u * 8u; // 8 neighbourhoods' worth of rows per wave
    const uint32_t qT = (uint32_t)((256.0 + 2.5) * (256.0 + 2.5)) + 1u;
    std::mt19937 rng(7);
    // owners and candidates of a tile inside a window of 3 cells (768 quanta) along x and `wy` quanta in y, z: with wy = 768 about 12 %
    // of the pairs hit, as in a resting row of the dam-break
    std::vector<uint32_t> h((size_t)nTiles * 64u);
    const uint32_t wy = argc > 2 ? (uint32_t)atoi(argv[2]) : 768u; // window in y and z, quanta (768 = three cells: ~12 % hits; 256: ~45 %)
    for (auto &w : h) w = packq(rng() % 768u, rng() % wy, rng() % wy);
    uint32_t *d = nullptr, *c1 = nullptr, *c2 = nullptr;
    unsigned long long *cyc = nullptr;
    CHK(hipMalloc(&d, h.size() * 4)); CHK(hipMalloc(&c1, (size_t)nTiles * 64 * 4)); CHK(hipMalloc(&c2, (size_t)nTiles * 64 * 4));
    CHK(hipMalloc(&cyc, 16));
    CHK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const dim3 b(64 * WAVES_PER_BLOCK);
    float msD = 0, msM = 0;
    unsigned long long cyD = 0, cyM = 0;
    for (int it = 0; it < 3; ++it) {
        CHK(hipMemset(cyc, 0, 16));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_direct, dim3((nTiles / 2 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), b, 0, 0, d, nTiles / 2, qT, rows, c1, cyc);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&msD, e0, e1));
        CHK(hipMemcpy(&cyD, cyc, 8, hipMemcpyDeviceToHost));
        CHK(hipMemset(cyc, 0, 16));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma, dim3((nTiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), b, 0, 0, d, nTiles, (float)qT, rows, c2, cyc);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&msM, e0, e1));
        CHK(hipMemcpy(&cyM, cyc, 8, hipMemcpyDeviceToHost));
    }
    // check the mfma distances against the direct arithmetic on the host: hits of tile 0 .. 63, every (owner, candidate) pair
    std::vector<uint32_t> g2((size_t)nTiles * 64);
    CHK(hipMemcpy(g2.data(), c2, g2.size() * 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0, hits = 0;
    for (uint32_t t = 0; t < 64 && t < nTiles; ++t)
        for (uint32_t o = 0; o < 32; ++o) {
            uint32_t want = 0;
            const uint32_t a = h[(size_t)t * 64 + o];
            for (uint32_t k = 0; k < 32; ++k) {
                const uint32_t bb = h[(size_t)t * 64 + 32 + k];
                const int dx = (int)(a & 1023) - (int)(bb & 1023), dy = (int)((a >> 11) & 1023) - (int)((bb >> 11) & 1023), dz = (int)(a >> 22) - (int)(bb >> 22);
                want += (uint32_t)(dx * dx + dy * dy + dz * dz) < qT ? 1u : 0u;
            }
            const uint32_t got = g2[(size_t)t * 64 + o] + g2[(size_t)t * 64 + 32 + o]; // the two half-lanes of the owner's column
            hits += want;
            bad += got != want * rows ? 1 : 0; // (every row re-tests the same 32 candidates, permuted)
        }
    const double ownerRowsD = (double)(nTiles / 2) * 64.0 * rows, ownerRowsM = (double)nTiles * 32.0 * rows;
    printf("tiles %u, rows per wave %u, hit rate %.1f %% (host check of the mfma hit counts: %llu mismatches)\n", nTiles, rows,
           100.0 * hits / (64.0 * 32 * 32), (unsigned long long)bad);
    printf("direct: %.3f ms, %.1f wave-cycles per wave-row  = %.2f cycles per owner-row (64 owners x 8 slots per wave-row)\n", msD,
           (double)cyD / ((double)(nTiles / 2) * rows), (double)cyD / ownerRowsD);
    printf("mfma  : %.3f ms, %.1f wave-cycles per tile      = %.2f cycles per owner-row at ONE tile per row, %.2f at two (32 owners x 32 candidates)\n",
           msM, (double)cyM / ((double)nTiles * rows), (double)cyM / ownerRowsM, 2.0 * (double)cyM / ownerRowsM);
    printf("throughput: direct %.2f G owner-rows/s, mfma %.2f G owner-rows/s (one tile per row)\n", ownerRowsD / msD * 1e-6, ownerRowsM / msM * 1e-6);
    return bad ? 1 : 0;
}
