// check_div2.hip — the packed division of the force walk (nrs_kernels_tiled.h, div2) against the compiler's `/` on the device: all 2^32
// bit patterns of the numerator for a set of denominators, and 2^32 random (numerator, denominator) pairs incl. zeros, denormals,
// infinities and NaNs; counts operand pairs whose quotient bits differ (NaN payloads compared as "both NaN").
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/check_div2.hip -o tools/_bin/check_div2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define NRS_DEV __device__ __forceinline__
typedef float f2 __attribute__((ext_vector_type(2)));
NRS_DEV f2 splat2(float v) { f2 r = {v, v}; return r; }
NRS_DEV f2 pair2(float a, float b) { f2 r = {a, b}; return r; }
NRS_DEV f2 div2(f2 a, f2 b)   // (a copy of nrs_kernels_tiled.h's, so that this tool builds stand-alone)
{
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
}
__device__ inline bool same(float x, float y) { return (__float_as_uint(x) == __float_as_uint(y)) || (x != x && y != y); }
__device__ inline uint32_t mix(uint64_t &s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (uint32_t)(z ^ (z >> 31)); }
// mode 0: numerator = every bit pattern (two per thread-iteration), denominator fixed; mode 1: random pairs
__global__ void k_check(int mode, float den, uint64_t seed, unsigned long long *bad)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    if (mode == 0) {
        for (uint64_t v = 2 * tid; v < (1ull << 32); v += 2 * nth) {
            const f2 a = pair2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)v + 1u)), b = splat2(den);
            const f2 q = div2(a, b);
            mine += !same(q.x, a.x / b.x) + !same(q.y, a.y / b.y);
        }
    } else {
        uint64_t s = seed + tid * 0x632BE59BD9B4E019ull;
        for (int it = 0; it < 2048; ++it) {
            const f2 a = pair2(__uint_as_float(mix(s)), __uint_as_float(mix(s))), b = pair2(__uint_as_float(mix(s)), __uint_as_float(mix(s)));
            const f2 q = div2(a, b);
            mine += !same(q.x, a.x / b.x) + !same(q.y, a.y / b.y);
        }
    }
    if (mine) atomicAdd(bad, mine);
}
int main()
{
    unsigned long long *d = nullptr, h = 0, total = 0;
    if (hipMalloc(&d, 8) != hipSuccess) { printf("no device\n"); return 2; }
    const float dens[] = {0.0457f, 1.0f, 3.0f, 1.9999999f, 1.1754944e-38f, 1e-42f, 3.4e38f, 0.0f, -7.25e-5f, 2.0e-4f, 1.3e-12f, INFINITY};
    for (float den : dens) {
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, 0, den, 0ull, d);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("denominator %-14.8g: all 2^32 numerators, %llu differing quotients\n", den, h);
        total += h;
    }
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, 1, 0.f, 12345ull, d);   // 4096 * 256 * 2048 * 2 = 2^32 pairs
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("random operand pairs (all bit patterns equally likely): 2^32 pairs, %llu differing quotients\n", h);
    total += h;
    printf("check_div2: %s\n", total ? "MISMATCHES" : "identical");
    return total ? 1 : 0;
}
