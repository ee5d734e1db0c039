// nrs_math.h — device-side vector layer and smoothing kernels for the gfx950 SPH step.
//
// Semantics to match (not code): the reference's helper_math.h overload set
// (common/cuda_helpers/helper_math.h:817-829, 1000-1008, 1251-1301) takes and returns *float* scalars
// even when SVec3 is double3, and common/kernels_impl.cuh:85-203 mixes pow/powf.
// All arithmetic is IEEE (this file is compiled with -ffp-contract=off; division and sqrt are the
// correctly rounded forms), so the fp32 path can be compared bit-for-bit with the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrs {

#define NRS_DEV __device__ __forceinline__
#define NRS_HD __host__ __device__ __forceinline__

template <typename R> struct V3 { R x, y, z; };

template <typename R> struct Vec4T;
template <> struct Vec4T<float> { typedef float4 type; };
template <> struct Vec4T<double> { typedef double4 type; };

template <typename R> NRS_DEV V3<R> mk3(R x, R y, R z) { V3<R> v; v.x = x; v.y = y; v.z = z; return v; }
template <typename R, typename T4> NRS_DEV V3<R> xyz(const T4 &a) { return mk3<R>(a.x, a.y, a.z); }
template <typename R> NRS_DEV typename Vec4T<R>::type mk4(R x, R y, R z, R w)
{
    typename Vec4T<R>::type v; v.x = x; v.y = y; v.z = z; v.w = w; return v;
}
template <typename R> NRS_DEV typename Vec4T<R>::type mk4(V3<R> a, R w) { return mk4<R>(a.x, a.y, a.z, w); }

template <typename R> NRS_DEV V3<R> operator+(V3<R> a, V3<R> b) { return mk3<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename R> NRS_DEV V3<R> operator-(V3<R> a, V3<R> b) { return mk3<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
// scalar operands are float by signature, as in the reference
template <typename R> NRS_DEV V3<R> operator*(V3<R> a, float b) { return mk3<R>(a.x * b, a.y * b, a.z * b); }
template <typename R> NRS_DEV V3<R> operator*(float b, V3<R> a) { return mk3<R>(b * a.x, b * a.y, b * a.z); }
template <typename R> NRS_DEV V3<R> operator/(V3<R> a, float b) { return mk3<R>(a.x / b, a.y / b, a.z / b); }
template <typename R> NRS_DEV float dot(V3<R> a, V3<R> b) { return (float)(a.x * b.x + a.y * b.y + a.z * b.z); }
// NB: __fsqrt_rn is the NATIVE (1-ulp) sqrt in this HIP; sqrtf is the correctly rounded one (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).
NRS_DEV float sqrt_rn(float x) { return sqrtf(x); }
template <typename R> NRS_DEV float length(V3<R> v) { return sqrt_rn(dot(v, v)); }

// ---- divisions and square roots for operands in range --------------------------------------------------------------------------------
// The compiler expands a correctly rounded fp32 division into v_div_scale x2, v_rcp, fma, fma, mul, fma, fma, fma, v_div_fmas, v_div_fixup
// (AMDGPU LowerFDIV32; the ISA of `a / b` on gfx950 is exactly that, 11 instructions) and sqrtf into a scaling select, v_sqrt, two integer
// neighbours, two residual fmas, two selects, the unscaling and a class test (15).  The scaling exists for operands near the ends of the
// exponent range: v_div_scale returns its operand UNCHANGED (and clears the flag v_div_fmas reads, which then is a plain fma) when numerator
// and denominator are non-zero, the denominator and its reciprocal are normal, the numerator is not tiny, the quotient is not denormal and
// the exponents are less than 96 apart; v_div_fixup then only re-applies the quotient's sign.  For such operands the eight arithmetic steps
// in between ARE the division, bit for bit — nothing is approximated, the scaling is simply not needed — and divisions by the same
// denominator share the first three steps (for a launch constant they are formed once per thread).  Likewise sqrtf without its scaling for
// 2^-96 <= x < inf.
// The region the forms are used in (every use states the guard that puts its operands there, and what runs when the guard fails — the plain
// `/` and sqrtf):  numerator +0, or 2^-90 <= |n| <= 2^90 with 2^-90 <= |d| <= 2^90 and 2^-104 < |n / d| < 2^96.
// Why 2^-90 and not the ISA manual's "biased exponent above 23" (2^-103): the residual n - d q of a quotient step is a multiple of
// 2^(e_n - 47); it is a float — and the step exact — only while that is at least 2^-149.  tools/check_div2.hip found the case: for d = 2e-4
// the numerators +-1.34756e-31 (biased exponent 24) are a last-bit tie that the bare steps round the other way.  The same tool compares the
// forms with `/` and sqrtf on the device over the region: all 2^32 numerator patterns for twelve denominators, 2^32 random pairs, and every
// float in [2^-96, inf) for the square root — 0 differences.
#ifndef NRS_INRANGE_DIV
#define NRS_INRANGE_DIV 1 // 0: every division and square root as the compiler expands it
#endif
NRS_DEV float rcp_refined(float d) // steps 1-3: the reciprocal every quotient by d is built from
{
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
NRS_DEV float div_steps(float n, float d, float y) // steps 4-8, y = rcp_refined(d)
{
    const float q0 = n * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), y, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), y, q1);
}
NRS_DEV float sqrt_inrange(float x) // 2^-96 <= x < inf
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-dn, s, x), vs = __builtin_fmaf(-up, s, x);
    float o = 0.f >= vp ? dn : s;
    o = 0.f < vs ? up : o;
    return o;
}

// two floats per lane on packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 round each component like the scalar forms): the pairs of
// the force walk (two hits) and of the density walk (two list entries), and the in-range steps above for both components at once
typedef float f2 __attribute__((ext_vector_type(2)));
struct V3x2 { f2 x, y, z; };
NRS_DEV f2 splat2(float v) { f2 r = {v, v}; return r; }
NRS_DEV f2 pair2(float a, float b) { f2 r = {a, b}; return r; }
NRS_DEV f2 dot2(const V3x2 &a, const V3x2 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // ((x + y) + z), as dot()
NRS_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
NRS_DEV f2 rcp_refined2(f2 d) // steps 1-3: the reciprocal every quotient by d is built from
{
    const f2 r = pair2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
    return fma2(fma2(-d, r, splat2(1.0f)), r, r);
}
NRS_DEV f2 div_steps2(f2 n, f2 d, f2 y) // steps 4-8
{
    const f2 nd = -d;
    const f2 q0 = n * y;
    const f2 q1 = fma2(fma2(nd, q0, n), y, q0);
    return fma2(fma2(nd, q1, n), y, q1);
}
NRS_DEV f2 sqrt_inrange2(f2 x)
{
    const float sa = __builtin_amdgcn_sqrtf(x.x), sb = __builtin_amdgcn_sqrtf(x.y);
    const f2 s = pair2(sa, sb);
    const f2 dn = pair2(__uint_as_float(__float_as_uint(sa) - 1u), __uint_as_float(__float_as_uint(sb) - 1u));
    const f2 up = pair2(__uint_as_float(__float_as_uint(sa) + 1u), __uint_as_float(__float_as_uint(sb) + 1u));
    const f2 vp = fma2(-dn, s, x), vs = fma2(-up, s, x);
    f2 o = s;
    o.x = 0.f >= vp.x ? dn.x : o.x; o.x = 0.f < vs.x ? up.x : o.x;
    o.y = 0.f >= vp.y ? dn.y : o.y; o.y = 0.f < vs.y ? up.y : o.y;
    return o;
}

// x^3 the way g++ evaluates pow(SReal,int): in double, rounded once to SReal (kernels_impl.cuh:95)
template <typename R> NRS_DEV R cube_via_double(R x) { double d = (double)x; return (R)(d * d * d); }
// powf(x,2) (kernels_impl.cuh:113): float square, also in fp64 builds
template <typename R> NRS_DEV R square_via_float(R x) { float f = (float)x; return (R)(f * f); }
// powf(x,7) (sph_kernel_impl.cuh:426): float result of x^7; computed in double and rounded once
NRS_DEV float pow7f(float x)
{
    double d = (double)x, d2 = d * d, d4 = d2 * d2;
    return (float)(d4 * d2 * d);
}

enum { KS_MONAGHAN = 0, KS_MULLER = 1 };

// Muller poly6 W (kernels_impl.cuh:85-98)
template <typename R> NRS_DEV R Wdefault(V3<R> r, R h, R kpoly)
{
    R r2 = length(r) * length(r);
    R h2 = h * h;
    if (r2 > h2) return (R)0.0;
    R b = cube_via_double<R>(h2 - r2);
    return kpoly * b;
}
// gradient of poly6 (:103-116)
template <typename R> NRS_DEV V3<R> Wdefault_grad(V3<R> r, R h, R kpoly_grad)
{
    R r2 = length(r) * length(r);
    R h2 = h * h;
    if (r2 > h2) return mk3<R>(0, 0, 0);
    R b = square_via_float<R>(h2 - r2);
    return kpoly_grad * r * b;
}
// spiky gradient (:121-135)
template <typename R> NRS_DEV V3<R> Wpressure_grad(V3<R> r, R h, R kpress_grad)
{
    R rlen = length(r);
    R r2 = rlen * rlen;
    R h2 = h * h;
    if (r2 > h2) return mk3<R>(0, 0, 0);
    R c = (h - rlen) * (h - rlen);
    return kpress_grad * (r / rlen) * c;
}
// viscosity kernel "gradient" (:140-154)
template <typename R> NRS_DEV V3<R> Wviscosity_grad(V3<R> r, R h, R kvisc_grad, R kvisc_denum)
{
    R rlen = length(r);
    R r2 = rlen * rlen;
    R h2 = h * h;
    if (r2 > h2) return mk3<R>(0, 0, 0);
    R c = -(3 * rlen / kvisc_denum) + (2 / (h2)) - (h / (2 * rlen * rlen * rlen));
    return kvisc_grad * r * c;
}
// Monaghan cubic spline (:159-203); constants evaluated in double as the host compiler does
template <typename R> NRS_DEV R Wmonaghan(V3<R> r, R h)
{
    R value = (R)0.0;
    R invH = (R)(1.0 / h);
    R m_v = (R)(1.0 / (4.0 * 3.14159265358979323846 * h * h * h));
    R q = length(r) * invH;
    if (q >= 0 && q < 1)
        value = m_v * ((2 - q) * (2 - q) * (2 - q) - 4.0f * (1 - q) * (1 - q) * (1 - q));
    else if (q >= 1 && q < 2)
        value = m_v * ((2 - q) * (2 - q) * (2 - q));
    else
        value = 0.0f;
    return value;
}
template <typename R> NRS_DEV V3<R> Wmonaghan_grad(V3<R> r, R h)
{
    R m_g = (R)(1.0 / (4.0 * 3.14159265358979323846 * h * h * h));
    R dist = length(r);
    R invH = (R)(1.0 / h);
    R q = dist * invH;
    V3<R> gradient = mk3<R>(0, 0, 0);
    if (q >= 0 && q < 1) {
        R scalar = -3.0f * (2 - q) * (2 - q);
        scalar += 12.0f * (1 - q) * (1 - q);
        gradient = (m_g * invH * scalar / dist) * r;
    } else if (q >= 1 && q < 2) {
        R scalar = -3.0f * (2 - q) * (2 - q);
        gradient = (m_g * scalar * invH / dist) * r;
    }
    return gradient;
}

template <typename R, int KSET> NRS_DEV R W_dens(V3<R> r, R ir, R kp)
{
    if (KSET == KS_MULLER) return Wdefault<R>(r, ir, kp);
    return Wmonaghan<R>(r, ir);
}
template <typename R, int KSET> NRS_DEV V3<R> W_grad(V3<R> r, R ir, R kpg)
{
    if (KSET == KS_MULLER) return Wdefault_grad<R>(r, ir, kpg);
    return Wmonaghan_grad<R>(r, ir);
}
// length() and W_grad() for the hit-list walks of the IISPH chain (Muller kernels): the square root as the bare steps of the compiler's
// expansion ("operands in range" above; an argument below 2^-96 — coincident particles — takes sqrtf).  Same values, bit for bit: dot() and
// length() are float in both precisions, Wdefault_grad squares that float length.
NRS_DEV float length_listed(float d2)
{
#if NRS_INRANGE_DIV
    return d2 >= 0x1p-96f ? sqrt_inrange(d2) : sqrt_rn(d2);
#else
    return sqrt_rn(d2);
#endif
}
template <typename R> NRS_DEV V3<R> Wdefault_grad_len(V3<R> r, float rlen, R h, R kpoly_grad) // Wdefault_grad with length(r) handed in
{
    R r2 = rlen * rlen;
    R h2 = h * h;
    if (r2 > h2) return mk3<R>(0, 0, 0);
    R b = square_via_float<R>(h2 - r2);
    return kpoly_grad * r * b;
}
template <typename R, int KSET> NRS_DEV V3<R> W_grad_listed(V3<R> r, R ir, R kpg)
{
    if constexpr (KSET == KS_MULLER) return Wdefault_grad_len<R>(r, length_listed(dot(r, r)), ir, kpg);
    else return W_grad<R, KSET>(r, ir, kpg);
}

// SphSimParams, byte-identical to nrs_params_f32 / nrs_params_f64 (include/nereus_hip.h)
template <typename R> struct Params {
    uint32_t gridSize[3];
    uint32_t numCells;
    R worldOrigin[3];
    R cellSize[3];
    uint32_t numBodies;
    uint32_t maxParticlesPerCell;
    R gasStiffness, viscosity, surfaceTension, restDensity, particleMass, interactionRadius, timestep, particleRadius;
    R gravity[3];
    R soundSpeed;
    R beta;
    R kpoly, kpoly_grad, kpress_grad, kvisc_grad, kvisc_denum, ksurf1, ksurf2, bpol;
};
static_assert(sizeof(Params<float>) == 132, "fp32 SphSimParams must be 132 bytes");
static_assert(sizeof(Params<double>) == 240, "fp64 SphSimParams must be 240 bytes");

struct I3 { int x, y, z; };

// calcGridPos (sph_kernel_impl.cuh:105-113): true division, then floor
template <typename R> NRS_DEV I3 calcGridPos(const Params<R> &P, V3<R> p)
{
    I3 g;
    g.x = (int)floor((p.x - P.worldOrigin[0]) / P.cellSize[0]);
    g.y = (int)floor((p.y - P.worldOrigin[1]) / P.cellSize[1]);
    g.z = (int)floor((p.z - P.worldOrigin[2]) / P.cellSize[2]);
    return g;
}
NRS_DEV uint32_t umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
// calcGridHash (:118-125): power-of-two wrap, 24-bit multiplies.
// P.numBodies (unused by the reference, common/sph_kernel.cuh:27) carries the first cell-x column of the context's cell-table
// WINDOW in the device-side copy of the parameters: 0 for a single domain (then this is the reference's hash exactly); a slab
// rank keeps tables only for its own columns + halo, gridSize[0] being the (power-of-two) window width (nrs_ctx_impl.h).
template <typename R> NRS_DEV uint32_t grid_x(const Params<R> &P, int gx) { return ((uint32_t)gx - P.numBodies) & (P.gridSize[0] - 1); }
template <typename R> NRS_DEV uint32_t calcGridHash(const Params<R> &P, int gx, int gy, int gz)
{
    uint32_t x = grid_x<R>(P, gx);
    uint32_t y = (uint32_t)gy & (P.gridSize[1] - 1);
    uint32_t z = (uint32_t)gz & (P.gridSize[2] - 1);
    return umul24(umul24(z, P.gridSize[1]), P.gridSize[0]) + umul24(y, P.gridSize[0]) + x;
}

static constexpr uint32_t CELL_EMPTY = 0xffffffffu;

// ---- compact scan candidates (the neighbour scan of the interior workgroups, nrs_kernels_tiled.h) -----------------------------
// A quantised copy of every sorted position, written by the reorder kernels next to the exact one: the position modulo FOUR
// cells per axis.  Differences of such coordinates wrap, so for two particles less than two cells apart on every axis — which
// every pair inside the interaction radius is (host check: h < ~2 cellSize, else the context does not use the compact scan) —
// the wrapped difference IS the coordinate difference in quanta, whatever cells the two sit in.  The scan keeps a candidate when
// the integer squared distance is below (h / quantum + QP_MARGIN)^2: a SUPERSET of the exact hits; the process phase applies the
// exact float test to the exact position, which it has to gather anyway, and compacts the list before it is published.
// Two forms, chosen at compile time (both pass the parity suite):
//   QP_BYTES 8: 16 bits per axis in units of cellSize/16384, words (x | y << 16, z): two v_pk_sub_i16 (wrap per 16-bit lane) and
//               two v_dot2_i32_i16 give the squared distance (< 3 * 2^30: no overflow as an unsigned number) — 4 VALU
//               instructions, 8 bytes and two VGPRs per candidate in flight;
//   QP_BYTES 4: 10 bits per axis in units of cellSize/256 in one word — 8 VALU instructions, 4 bytes, one VGPR (details below).
// Measured at 10 M particles the 4-byte form wins (density stage 0.536 vs 0.602 ms at rest, 0.904 vs 1.021 ms in the developed
// flow; exact positions: 0.646 / 1.091): what a candidate costs is its bytes through the L1 path and its register, not the
// distance arithmetic.
#ifndef QP_BYTES
#define QP_BYTES 4 // 4 (default, measured faster: 0.536 vs 0.602 ms at rest, 0.904 vs 1.021 developed): one word, 10 bits per axis, see below.  8: the 16-bit form described above
#endif
struct QuantCfg { double o[3]; float s[3]; }; // grid origin, quanta per metre (host: QP_PER_CELL / cellSize)
template <typename R> NRS_DEV void quantize_t(const QuantCfg &q, V3<R> p, float &tx, float &ty, float &tz)
{
    tx = (float)(p.x - (R)q.o[0]) * q.s[0];
    ty = (float)(p.y - (R)q.o[1]) * q.s[1];
    tz = (float)(p.z - (R)q.o[2]) * q.s[2];
}
#if QP_BYTES == 8
typedef uint2 qword_t;
constexpr float QP_PER_CELL = 16384.0f;
// error budget, per axis, in quanta: t = fl(fl(x - o) * s) with s = fl(16384 / cs) carries a relative error <= 3 * 2^-24, i.e.
// up to 6 quanta at |t| = 2^25 (2048 cells from the grid origin; an fp32 position itself is no finer there); two floors add < 1:
// |dk - dt| < 13 per axis, 13 * sqrt(3) = 22.6 on the distance; 32 leaves room for the float rounding of the exact test.
// 32 / 16384 = 0.2 % of h: about 0.6 % more list entries than exact hits.
constexpr float QP_MARGIN = 32.0f;
constexpr float QP_FAR = 33554432.0f; // 2^25 quanta: an owner beyond that (or with a NaN coordinate) takes the exact path
constexpr float QP_HALF = 32767.0f;   // a difference must stay below half the period (4 cells)
NRS_DEV qword_t pack_quanta(float tx, float ty, float tz)
{
    const uint32_t kx = (uint32_t)(int)floorf(tx) & 0xffffu, ky = (uint32_t)(int)floorf(ty) & 0xffffu, kz = (uint32_t)(int)floorf(tz) & 0xffffu;
    return make_uint2(kx | (ky << 16), kz);
}
#else
// 4-byte form: 10 bits per axis at bits 0 / 11 / 22 (units of cellSize/256, modulo 4 cells); bits 10 and 21 are guard bits, left
// 0.  With the guard bits SET in the owner's word, ONE 32-bit subtraction gives the three differences modulo 1024 (each guard
// absorbs its field's borrow); three sign extensions, three 24-bit multiplies and an add give the squared distance.  Half the
// bytes of the 16-bit form, 8 VALU instructions per distance instead of 4, about 3 % false positives instead of 0.6 %.
typedef uint32_t qword_t;
constexpr uint32_t QP_GUARD = (1u << 10) | (1u << 21);
constexpr float QP_PER_CELL = 256.0f;
// |k_i - k_j - (t_i - t_j)| < 1 (two floors) + 2 * 0.19 (relative error <= 3 * 2^-24 on |t| < 2^20 + 2^9) per axis, so
// |dk| <= |dt| + 1.38 * sqrt(3) = |dt| + 2.39; + the float rounding of the exact test itself
constexpr float QP_MARGIN = 2.5f;
constexpr float QP_FAR = 1048576.0f; // 2^20 quanta = 4096 cells from the grid origin: an owner beyond that (or with a NaN coordinate) is scanned on exact positions (quant_far)
constexpr float QP_HALF = 511.0f;
NRS_DEV qword_t pack_quanta(float tx, float ty, float tz)
{
    const uint32_t kx = (uint32_t)(int)floorf(tx) & 1023u, ky = (uint32_t)(int)floorf(ty) & 1023u, kz = (uint32_t)(int)floorf(tz) & 1023u;
    return kx | (ky << 11) | (kz << 22);
}
#endif
// an owner too far from the grid origin for the error budget of the quanta, or with a NaN coordinate
template <typename R> NRS_DEV bool quant_far(const QuantCfg &q, V3<R> p)
{
    float tx, ty, tz;
    quantize_t<R>(q, p, tx, ty, tz);
    return !((fabsf(tx) < QP_FAR) && (fabsf(ty) < QP_FAR) && (fabsf(tz) < QP_FAR));
}
template <typename R> NRS_DEV qword_t quantize_pos(const QuantCfg &q, V3<R> p)
{
    float tx, ty, tz;
    quantize_t<R>(q, p, tx, ty, tz);
    return pack_quanta(tx, ty, tz);
}

} // namespace nrs
