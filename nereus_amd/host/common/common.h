// common.h — precision switch and vocabulary types of the Nereus host API, for the MI355X build.
//
// Same names and meaning as the reference's common/common.h:9-52 (SReal/SVec3/SVec4/SUint, make_SVec3/4,
// GL_REAL, KERNEL_SET ids, namespace macros) so that main.cpp and user code compile unchanged; the vector
// types come from HIP's <hip/hip_vector_types.h> (float3/float4/double3/double4, same layout as CUDA's)
// and are usable from plain g++.
#pragma once
#ifndef COMMON_H
#define COMMON_H

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_vector_types.h>

// launch geometry of the reference (kept for source compatibility; our kernels choose their own)
#define CUDA_BLOCKSIZE 256

// smoothing-kernel families selectable with -DKERNEL_SET=
#define MONAGHAN 0
#define MULLER 1

#ifndef DOUBLE_PRECISION
#define DOUBLE_PRECISION 0
#endif
#ifndef KERNEL_SET
#define KERNEL_SET MULLER
#endif
#ifndef USE_SURFACE_TENSION
#define USE_SURFACE_TENSION 1
#endif

typedef unsigned int SUint;

#if DOUBLE_PRECISION == 1
typedef double SReal;
typedef double3 SVec3;
typedef double4 SVec4;
#define make_SVec3 make_double3
#define make_SVec4 make_double4
#define GL_REAL GL_DOUBLE
#else
typedef float SReal;
typedef float3 SVec3;
typedef float4 SVec4;
#define make_SVec3 make_float3
#define make_SVec4 make_float4
#define GL_REAL GL_FLOAT
#endif

#define NEREUS_NAMESPACE_BEGIN namespace Nereus {
#define NEREUS_NAMESPACE_END }
#define EXTERN_C_BEGIN extern "C" {
#define EXTERN_C_END }

#endif // COMMON_H
