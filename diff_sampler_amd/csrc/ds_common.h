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

// ---- Race-stress build (tests only; round 6).  -DDS_RACE_STRESS=<wave mask> (build.py: build_variant('stress_a' / 'stress_b')) makes the waves of
// the mask sleep ~2 us (s_sleep 75 = 4 800 cycles: several L2 -> LDS round trips) at every point where a wave PRODUCES shared LDS contents --
// right before it issues an LDS-DMA (global_load_lds) or a staging ds_write that other waves consume.  A slept wave's data lands late and its
// own later reads come late, while the other waves run ahead to their next reads AND their next overwrites: an ordering that rests on timing
// instead of on `s_waitcnt` + `s_barrier` (docs/HISTORY.md G.5: coefficient rows fetched by wave 0, read by waves 1 - 7 with no barrier) then
// fails on every run instead of once per cold box.  Two complementary masks are built: 0x21 (waves 0 and 5 late: the single-wave producers
// are late -> read-before-landed) and 0xDE (every other wave late: the producers are early -> overwritten-before-read).  Results of a
// correct kernel do not depend on the mask; tests/test_hip_race_stress.py runs the kernel / fp16 suites against both libraries.
#ifdef DS_RACE_STRESS
// The wave index goes through v_readfirstlane: the test is then a SCALAR branch around the s_sleep.  On the per-lane value the compiler only
// masks EXEC (s_and_saveexec) and lets the scalar s_sleep run in every wave -- a uniform delay, no skew (session r9a: the library with the
// round-5 race re-introduced passed its tests until this was fixed).
#define DS_RACE_SKEW(wave_)                                                                                                    \
    do {                                                                                                                       \
        if (((unsigned)(DS_RACE_STRESS) >> ((unsigned)__builtin_amdgcn_readfirstlane((int)(wave_)) & 31u)) & 1u) __builtin_amdgcn_s_sleep(75); \
    } while (0)
#else
#define DS_RACE_SKEW(wave_) do { } while (0)
#endif

// ---- Timeline build (diagnostics only; round 6).  -DDS_TIMELINE=1 (build.py variant 'timeline', loaded through DS_LIB_PATH by tools/timeline_gemm.py):
// a launch whose ds_conv_args.tune.ablate has bit 15 set and which was given a workspace writes, per workgroup and wave, s_memtime stamps of its
// phases (slots 0 .. 6) and {HW_ID, XCC_ID} (slot 7) into the workspace as u64[tile][8 waves][8 slots].  The product library does not
// contain a single instruction of it.
#ifdef DS_TIMELINE
__device__ __forceinline__ void ds_timeline_mark(float* buf, int abl, int slot, int tile) {
    if (!(abl & 0x8000) || !buf) return;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* b = reinterpret_cast<unsigned long long*>(buf) + ((size_t)tile * 8 + (threadIdx.x >> 6)) * 8;
        b[slot] = __builtin_amdgcn_s_memtime();
        if (slot == 0) b[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}
#define DS_TL(buf_, abl_, slot_, tile_) ds_timeline_mark((buf_), (abl_), (slot_), (tile_))
#else
#define DS_TL(buf_, abl_, slot_, tile_) do { } while (0)
#endif

static inline bool ds_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// SiLU with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division sequence (~10 VALU instructions):
// the activation sits in the convolution's halo loader and epilogue, where every VALU instruction is exposed.
__device__ __forceinline__ float ds_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// EDM preconditioning coefficients (diff-solvers-main/models/networks_edm.py:488-491), fp32, the reference's operation order EXACTLY: every
// product and sum rounded on its own (contraction off inside these bodies).  Round 5: with the compiler free to contract `s * s + sd * sd`
// either way, two kernels inlining the same source line disagreed in the last bit of c_skip / c_out at small sigma -- invisible at every
// tolerance, but the head-fused solver update (conv3x3_thin.hip) and the stand-alone update kernel (solver.hip) must produce EQUAL bits.
__device__ __forceinline__ float ds_c_skip(float s, float sd) {
#pragma clang fp contract(off)
    const float s2 = s * s, d2 = sd * sd;
    return d2 / (s2 + d2);
}
__device__ __forceinline__ float ds_c_out(float s, float sd) {
#pragma clang fp contract(off)
    const float s2 = s * s, d2 = sd * sd, num = s * sd;
    return num / sqrtf(s2 + d2);
}
__device__ __forceinline__ float ds_c_in(float s, float sd) {
#pragma clang fp contract(off)
    const float s2 = s * s, d2 = sd * sd;
    return 1.0f / sqrtf(d2 + s2);
}

// ---- The solver update's per-element arithmetic (ds_solver_update, include/ds_engine.h), shared by the update kernels (solver.hip) and by the
// network head that applies it in its epilogue (conv3x3_thin.hip, ds_conv_args.update): ONE definition with explicit fused multiply-adds,
// so that the fused and the two-launch form produce equal bits by construction (no compiler contraction choice is involved).
//     D = raw ? c_skip * x + c_out * F : F;  d = (x - D) / t;  m = store_d ? d : D;  x' = cx * xb + cm * m + sum_k ch[k] * hist[k]
struct DsUpdCoefs { float cx, cm, ch0, ch1, ch2, t, sig, pad; };

__device__ __forceinline__ DsUpdCoefs ds_upd_load_coefs(const ds_update_args& a, int img) {
    DsUpdCoefs c;
    if (a.coefs) {
        const float* r = a.coefs + (size_t)(a.coef_rows == 1 ? 0 : img) * 8;
        c.cx = r[0]; c.cm = r[1]; c.ch0 = r[2]; c.ch1 = r[3]; c.ch2 = r[4]; c.t = r[5]; c.sig = r[6]; c.pad = 0.f;
    } else {
        c.cx = a.hcoefs[0]; c.cm = a.hcoefs[1]; c.ch0 = a.hcoefs[2]; c.ch1 = a.hcoefs[3]; c.ch2 = a.hcoefs[4];
        c.t = a.hcoefs[5]; c.sig = a.hcoefs[6]; c.pad = 0.f;
    }
    return c;
}

// one element; h0 / h1 / h2 are read only where has0 / has1 / has2 (the history pointers) say so
__device__ __forceinline__ void ds_upd_element(const DsUpdCoefs& k, float cskip, float cout_, bool raw, bool store_d, float x, float xb, float f,
                                               bool has0, float h0, bool has1, float h1, bool has2, float h2, float& m, float& xo) {
#pragma clang fp contract(off)
    const float D = raw ? __builtin_fmaf(cskip, x, cout_ * f) : f;             // networks_edm.py:495
    const float d = (x - D) / k.t;                                             // solvers.py:80
    m = store_d ? d : D;
    float acc = __builtin_fmaf(k.cx, xb, k.cm * m);
    if (has0) acc = __builtin_fmaf(k.ch0, h0, acc);
    if (has1) acc = __builtin_fmaf(k.ch1, h1, acc);
    if (has2) acc = __builtin_fmaf(k.ch2, h2, acc);
    xo = acc;
}
