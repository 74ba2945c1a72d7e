// Shared helpers for the libdsamd.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ds_engine.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DS_CHECK_LAUNCH()                              \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

static inline bool ds_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// SiLU with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division sequence (~10 VALU instructions):
// the activation sits in the convolution's halo loader and epilogue, where every VALU instruction is exposed.
__device__ __forceinline__ float ds_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// EDM preconditioning coefficients (diff-solvers-main/models/networks_edm.py:488-491), fp32, same operation order.
__device__ __forceinline__ float ds_c_skip(float s, float sd) { return (sd * sd) / (s * s + sd * sd); }
__device__ __forceinline__ float ds_c_out(float s, float sd) { return s * sd / sqrtf(s * s + sd * sd); }
__device__ __forceinline__ float ds_c_in(float s, float sd) { return 1.0f / sqrtf(sd * sd + s * s); }
