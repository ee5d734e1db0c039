// check_div2.hip — (1) the packed division of the force walk (nrs_kernels_tiled.h, div2) and (2) the divisions and square roots "for
// operands in range" (nrs_math.h: rcp_refined / div_steps / sqrt_inrange, the bare steps of the compiler's expansions) against the
// compiler's `/` and sqrtf on the device.  (2): for a set of denominators every numerator bit pattern that lies in v_div_scale's
// pass-through region (or is +0), 2^32 random pairs filtered the same way, and every float in [2^-96, inf) for the square root.
// (1): all 2^32
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

// ---- (2) operands in range: copies of nrs_math.h's forms ----
__device__ __forceinline__ float rcp_refined(float d) { const float r = __builtin_amdgcn_rcpf(d); return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r); }
__device__ __forceinline__ float div_steps(float n, float d, float y)
{
    const float q0 = n * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), y, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), y, q1);
}
__device__ __forceinline__ float sqrt_inrange(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-dn, s, x), vs = __builtin_fmaf(-up, s, x);
    float o = 0.f >= vp ? dn : s;
    o = 0.f < vs ? up : o;
    return o;
}
// the region nrs_math.h uses the forms in: numerator +0, or 2^-90 <= |n| <= 2^90, 2^-90 <= |d| <= 2^90, 2^-104 < |n / d| < 2^96
__device__ inline bool in_range(float n, float d)
{
    const uint32_t bn = __float_as_uint(n), bd = __float_as_uint(d);
    const int en = (int)((bn >> 23) & 0xff) - 127, ed = (int)((bd >> 23) & 0xff) - 127;
    if (ed < -90 || ed >= 90) return false;
    if (bn == 0u) return true;
    if (en < -90 || en >= 90) return false;
    return (en - ed) < 95 && (en - ed) > -103;
}
__global__ void k_check_steps(int mode, float den, uint64_t seed, unsigned long long *bad, unsigned long long *tested)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0, cnt = 0;
    if (mode == 0) {
        const float y = rcp_refined(den);
        for (uint64_t v = tid; v < (1ull << 32); v += nth) {
            const float n = __uint_as_float((uint32_t)v);
            if (!in_range(n, den)) continue;
            ++cnt;
            mine += !same(div_steps(n, den, y), n / den);
        }
    } else if (mode == 1) {
        uint64_t s = seed + tid * 0x632BE59BD9B4E019ull;
        for (int it = 0; it < 4096; ++it) {
            const float n = __uint_as_float(mix(s)), d = __uint_as_float(mix(s));
            if (!in_range(n, d)) continue;
            ++cnt;
            mine += !same(div_steps(n, d, rcp_refined(d)), n / d);
        }
    } else {
        for (uint64_t v = 0x0f800000ull + tid; v < 0x7f800000ull; v += nth) { // 2^-96 .. below inf
            const float x = __uint_as_float((uint32_t)v);
            ++cnt;
            mine += !same(sqrt_inrange(x), sqrtf(x));
        }
    }
    if (mine) atomicAdd(bad, mine);
    atomicAdd(tested, cnt);
}
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
    // (2) the forms for operands in range
    unsigned long long *t = nullptr, ht = 0;
    hipMalloc(&t, 8);
    const float dens2[] = {0.0457f, 1.0f, 3.0f, 1.9999999f, 1.0e-27f, 1.2e27f, -7.25e-5f, 2.0e-4f, 1.3e-12f, 1.90885e-4f, 0.33333334f, 6.0221e23f};
    for (float den : dens2) {
        hipMemset(d, 0, 8); hipMemset(t, 0, 8);
        hipLaunchKernelGGL(k_check_steps, dim3(4096), dim3(256), 0, 0, 0, den, 0ull, d, t);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
        printf("in-range steps, denominator %-14.8g: %llu numerators in range, %llu differing quotients\n", den, ht, h);
        total += h;
    }
    hipMemset(d, 0, 8); hipMemset(t, 0, 8);
    hipLaunchKernelGGL(k_check_steps, dim3(4096), dim3(256), 0, 0, 1, 0.f, 777ull, d, t);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    printf("in-range steps, random pairs: %llu pairs in range, %llu differing quotients\n", ht, h);
    total += h;
    hipMemset(d, 0, 8); hipMemset(t, 0, 8);
    hipLaunchKernelGGL(k_check_steps, dim3(4096), dim3(256), 0, 0, 2, 0.f, 0ull, d, t);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    printf("square root without scaling: %llu arguments in [2^-96, inf), %llu differing roots\n", ht, h);
    total += h;
    printf("check_div2: %s\n", total ? "MISMATCHES" : "identical");
    return total ? 1 : 0;
}
