// gnnpe_dpp.hip.h -- wave-wide reductions on the VALU's data-parallel primitives (DPP), shared by the auxiliary-index pass
// (gnnpe_aux.hip) and the index leaf kernel (gnnpe_index.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gnnpe {

// Wave-wide reductions on the VALU's data-parallel primitives (DPP: quad permutes, row mirrors, row broadcasts -- gfx9
// family), result broadcast from lane 63.  The first version used __shfl_xor butterflies = ds_bpermute through the LDS
// crossbar: 162 of them per leaf (27 dwords x 6 steps) made the LDS pipe of every CU the pass's bottleneck.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)b), hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
#define GNNPE_DPP_REDUCE(T, DPP, OP)                                            \
    v = OP(v, DPP<0xB1, 0xF>(v));   /* quad_perm [1,0,3,2] */                   \
    v = OP(v, DPP<0x4E, 0xF>(v));   /* quad_perm [2,3,0,1] */                   \
    v = OP(v, DPP<0x141, 0xF>(v));  /* row_half_mirror */                       \
    v = OP(v, DPP<0x140, 0xF>(v));  /* row_mirror: every row of 16 is reduced */ \
    v = OP(v, DPP<0x142, 0xA>(v));  /* row_bcast15 into rows 1 and 3 */         \
    v = OP(v, DPP<0x143, 0xC>(v));  /* row_bcast31 into rows 2 and 3: lane 63 holds the wave's result */
__device__ __forceinline__ double lane63(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double dpp_min_f64(double v)  // result in lane 63
{
    GNNPE_DPP_REDUCE(double, dpp_f64, fmin)
    return v;
}
__device__ __forceinline__ double dpp_max_f64(double v)
{
    GNNPE_DPP_REDUCE(double, dpp_f64, fmax)
    return v;
}
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v)
{
    GNNPE_DPP_REDUCE(uint32_t, dpp_u32, max)
    return v;
}
// two 16-bit lanes per dword (v_pk_max_u16): the index leaf kernel's label ranks
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const u16x2 z = __builtin_elementwise_max(x, y);
    uint32_t r;
    __builtin_memcpy(&r, &z, 4);
    return r;
}
__device__ __forceinline__ uint32_t dpp_max_pk_u16(uint32_t v)
{
    GNNPE_DPP_REDUCE(uint32_t, dpp_u32, pk_max_u16)
    return v;
}
// wave sum (result in lane 63): the masked row-broadcast steps must contribute the identity to the rows they skip, so the
// "old" operand of the DPP move is 0 here (the min / max reductions above pass the value itself, their identity)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_u32_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v)
{
    v += dpp_u32_zero<0xB1, 0xF>(v);
    v += dpp_u32_zero<0x4E, 0xF>(v);
    v += dpp_u32_zero<0x141, 0xF>(v);
    v += dpp_u32_zero<0x140, 0xF>(v);
    v += dpp_u32_zero<0x142, 0xA>(v);
    v += dpp_u32_zero<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)dpp_add_u32(v), 63); }
__device__ __forceinline__ double wave_min(double v) { return lane63(dpp_min_f64(v)); }
__device__ __forceinline__ double wave_max(double v) { return lane63(dpp_max_f64(v)); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)dpp_max_u32(v), 63); }

// wave-wide inclusive scans on DPP (row shifts inside rows of 16, then the two row broadcasts): no LDS crossbar, so the
// six steps cost VALU issue only -- __shfl_up is ds_bpermute, one LDS round trip per step on the kernel's critical path
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_shift0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);  // lanes without a source read 0
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += dpp_shift0<0x111, 0xF>(v);  // row_shr:1
    v += dpp_shift0<0x112, 0xF>(v);  // row_shr:2
    v += dpp_shift0<0x114, 0xF>(v);  // row_shr:4
    v += dpp_shift0<0x118, 0xF>(v);  // row_shr:8
    v += dpp_shift0<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_shift0<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)
{
    v = max(v, dpp_shift0<0x111, 0xF>(v));
    v = max(v, dpp_shift0<0x112, 0xF>(v));
    v = max(v, dpp_shift0<0x114, 0xF>(v));
    v = max(v, dpp_shift0<0x118, 0xF>(v));
    v = max(v, dpp_shift0<0x142, 0xA>(v));
    v = max(v, dpp_shift0<0x143, 0xC>(v));
    return v;
}

}  // namespace gnnpe
