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

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that drives several GPUs must set it on
// each of them.  One bit per device id, per call site (the static lives in the launcher's own -- possibly templated -- scope);
// thread-safe; costs one hipGetDevice per launch once set.
#include <atomic>
#define DS_ENSURE_DYN_LDS(kernel_ptr, bytes)                                                                                   \
    do {                                                                                                                        \
        static std::atomic<unsigned long long> ds_lds_done_{0};                                                                 \
        int ds_dev_ = 0;                                                                                                        \
        (void)hipGetDevice(&ds_dev_);                                                                                           \
        const unsigned long long ds_bit_ = 1ull << (ds_dev_ & 63);                                                              \
        if (!(ds_lds_done_.load(std::memory_order_acquire) & ds_bit_)) {                                                        \
            const hipError_t ds_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                             \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));                  \
            if (ds_e_ != hipSuccess) return (int)ds_e_;                                                                         \
            ds_lds_done_.fetch_or(ds_bit_, std::memory_order_release);                                                          \
        }                                                                                                                       \
    } while (0)

static inline bool ds_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// SiLU with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division sequence (~10 VALU instructions):
// the activation sits in the convolution's halo loader and epilogue, where every VALU instruction is exposed.
__device__ __forceinline__ float ds_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// EDM preconditioning coefficients (diff-solvers-main/models/networks_edm.py:488-491), fp32, same operation order.
__device__ __forceinline__ float ds_c_skip(float s, float sd) { return (sd * sd) / (s * s + sd * sd); }
__device__ __forceinline__ float ds_c_out(float s, float sd) { return s * sd / sqrtf(s * s + sd * sd); }
__device__ __forceinline__ float ds_c_in(float s, float sd) { return 1.0f / sqrtf(sd * sd + s * s); }
