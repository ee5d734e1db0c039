// exhaustive check: for a fixed divisor b, does q' = fma(r, y, q) with q = a*y, r = fma(-b, q, a), y = RN(1/b) equal a / b for EVERY float a?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
__global__ void k(float b, float y, unsigned long long *mis, unsigned long long *misNormal, uint32_t *firstBad)
{
    const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096ull;
    unsigned long long m = 0, mn = 0;
    for (uint32_t k2 = 0; k2 < 4096; ++k2) {
        const uint32_t bits = (uint32_t)(base + k2);
        const float a = __uint_as_float(bits);
        const float ref = a / b;
        const float q = a * y;
        const float r = fmaf(-b, q, a);
        const float got = fmaf(r, y, q);
        const bool same = (__float_as_uint(ref) == __float_as_uint(got)) || (ref != ref && got != got);
        if (!same) {
            ++m;
            const float aa = fabsf(a), rr = fabsf(ref);
            if (aa >= 7.8886091e-31f /* 2^-100 */ && aa <= 1.2676506e30f /* 2^100 */) { ++mn; atomicMin(firstBad, bits); }
        }
    }
    if (m) atomicAdd(mis, m);
    if (mn) atomicAdd(misNormal, mn);
}
int main()
{
    const float bs[] = {0.0457f, 0.045700002f, 0.0537f, 1.0f / 3.0f, 0.1f, 7.13e-5f, 0.020565f, 1.9999999f};
    unsigned long long *d; uint32_t *fb;
    hipMalloc(&d, 16); hipMalloc(&fb, 4);
    for (float b : bs) {
        const float y = (float)(1.0 / (double)b);
        hipMemset(d, 0, 16); hipMemset(fb, 0xff, 4);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, b, y, d, d + 1, fb);
        unsigned long long h[2]; uint32_t f;
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
        printf("b = %.9g  y = %.9g : mismatches %llu (of them with 2^-100 <= |a| <= 2^100: %llu, first 0x%08x)\n", b, y, h[0], h[1], f);
    }
    return 0;
}
