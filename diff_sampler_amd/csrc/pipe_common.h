// Hand-ordered pipeline pieces shared by the second-generation kernels (conv3x3_halo2.hip, gemm_f16.hip): volatile-asm LDS and
// global accesses with hand-counted waits, fragment sets, compile-time loops.  See conv3x3_halo2.hip for the rules they follow
// (cdna_hip_programming.md section 5.7, form (ii): the wait statement names the loaded registers "+v").
#pragma once
#include <type_traits>

#include "igemm_common.h"

namespace igemm {
namespace {

struct Frag2 { f32x4 a0, a1, b0, b1; };
template <int K> using IC = std::integral_constant<int, K>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF>
__device__ __forceinline__ f32x4 lds_rd(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_wr(unsigned addr, const f32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ f32x4 gld16(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned lds_addr2(const float* p) {
    typedef __attribute__((address_space(3))) const void* lcptr_t;
    return (unsigned)(size_t)(lcptr_t)(p);
}
#define DS2_FRAG_WAIT(N, f) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"((f).a0), "+v"((f).a1), "+v"((f).b0), "+v"((f).b1))
#define DS2_FENCE() __builtin_amdgcn_sched_barrier(0)

// A operand: two 32-row MFMA tiles per wave, base address + immediate; B operand: swizzled 16-B chunk of the LDS-DMA weight image
template <int AOFF>
__device__ __forceinline__ void frag_read2(Frag2& f, unsigned va0, unsigned va1, unsigned vb) {
    f.a0 = lds_rd<AOFF>(va0);
    f.a1 = lds_rd<AOFF>(va1);
    f.b0 = lds_rd<0>(vb);
    f.b1 = lds_rd<4096>(vb);
}

// Split mode ("fp32 emulated on the fp16 pipe", MODE 2): every operand is hi + lo with hi = fp16(x), lo = fp16(x - hi); a product is the
// three MFMAs hi*hi + hi*lo + lo*hi (the dropped lo*lo term and the rounding of lo are 2**-22 relative, fp32 class).  One 128-B LDS
// row holds a 32-channel slab as [32 hi | 32 lo] halfs, so the byte offsets 0 / 32 / 64 / 96 of the fp32 kernel's four K steps now
// select (hi, k 0..15), (hi, k 16..31), (lo, k 0..15), (lo, k 16..31).
struct Frag2S { f32x4 a0h, a1h, b0h, b1h, a0l, a1l, b0l, b1l; };
template <int AOFF>
__device__ __forceinline__ void frag_read2s(Frag2S& f, unsigned va0, unsigned va1, unsigned vbh, unsigned vbl) {
    f.a0h = lds_rd<AOFF>(va0);
    f.a1h = lds_rd<AOFF>(va1);
    f.a0l = lds_rd<AOFF + 64>(va0);
    f.a1l = lds_rd<AOFF + 64>(va1);
    f.b0h = lds_rd<0>(vbh);
    f.b1h = lds_rd<4096>(vbh);
    f.b0l = lds_rd<0>(vbl);
    f.b1l = lds_rd<4096>(vbl);
}
#define DS2_FRAG_WAIT_S(N, f)                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"((f).a0h), "+v"((f).a1h), "+v"((f).b0h), "+v"((f).b1h), "+v"((f).a0l), "+v"((f).a1l), \
                 "+v"((f).b0l), "+v"((f).b1l))
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_wr64(unsigned addr, const f32x2& v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pack_h2(float a, float b) {            // two fp32 -> one dword of two fp16 (round to nearest even)
    const h2 pk = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(float, pk);
}


}  // namespace
}  // namespace igemm
