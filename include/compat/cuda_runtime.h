/*
 * Types-only forwarder so the reference's UNMODIFIED front end (src/display.cpp:3
 * and src/simulator.h:3 include <cuda_runtime.h> for float3 / int2 / make_int2)
 * compiles against this repository's headers.  It supplies the HIP vector types
 * and nothing else: no runtime API, no CUDA symbol is mapped.  Put this
 * directory on the include path ONLY when building that unmodified front end.
 */
#ifndef SPH_COMPAT_CUDA_RUNTIME_H
#define SPH_COMPAT_CUDA_RUNTIME_H
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_vector_types.h>
#endif
