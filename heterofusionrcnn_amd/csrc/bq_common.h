// bq_common.h -- helpers shared by the ball-query kernels (ballquery.hip: the single-launch cell kernel;
// ballquery_sorted.hip: the cell-sorted two-kernel path).  gfx950 only.
#pragma once
#include "hf_common.h"

namespace hf {

// Points and grouped rows are 12-byte elements.  They go through raw buffer accesses of exactly 96 bits (clang widens a
// plain vec3 load / store to 16 bytes: the store would clobber the next element, the load could leave the tensor),
// whose range check also returns zeros / drops the store past the end, so no index needs clamping.
typedef unsigned u3v __attribute__((ext_vector_type(3)));
struct P3 { float x, y, z; };
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, static_cast<int>(bytes), 0x00020000);
}
__device__ __forceinline__ P3 load_p3(__amdgpu_buffer_rsrc_t r, unsigned elem)
{
    const u3v v = __builtin_amdgcn_raw_buffer_load_b96(r, static_cast<int>(elem * 12u), 0, 0);
    return P3{ __uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z) };
}

// a * b + c on the low 24 bits of a and b: one full-rate instruction (hipcc turns __umul24 + add into the
// quarter-rate 64-bit v_mad_u64_u32 here)
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// stores in three flavours: SM = 0 plain, 1 nontemporal, 2 write-through (sc0 sc1)
template <int SM> struct StoreAux { static constexpr int value = SM == 1 ? 2 : (SM == 2 ? 17 : 0); };
template <int SM>
__device__ __forceinline__ void store_i32(__amdgpu_buffer_rsrc_t r, unsigned elem, int v)
{
    __builtin_amdgcn_raw_buffer_store_b32(static_cast<unsigned>(v), r, static_cast<int>(elem * 4u), 0, StoreAux<SM>::value);
}
template <int SM>
__device__ __forceinline__ void store_p3(__amdgpu_buffer_rsrc_t r, unsigned elem, float x, float y, float z)
{
    const u3v v = { __float_as_uint(x), __float_as_uint(y), __float_as_uint(z) };
    __builtin_amdgcn_raw_buffer_store_b96(v, r, static_cast<int>(elem * 12u), 0, StoreAux<SM>::value);
}

}  // namespace hf
