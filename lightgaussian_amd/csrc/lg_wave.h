// lg_wave.h -- wave64 primitives: lane id, prefix popcount, DPP / permlane reductions, XCD-aware tile mapping
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// ------------------------------------------------------------------------------------------------
// wave64 helpers
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t prefix_popc(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// DPP wave reduction: after the call lane 63 holds the sum over all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111, 0xf>(v); // row_shr:1
    v = dpp_add<0x112, 0xf>(v); // row_shr:2
    v = dpp_add<0x114, 0xf>(v); // row_shr:4
    v = dpp_add<0x118, 0xf>(v); // row_shr:8   -> lane 15 of every row holds the row total
    v = dpp_add<0x142, 0xa>(v); // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v); // row_bcast:31 into rows 2,3 -> lane 63 = total
    return v;
}


// integer add across lanes through DPP (one v_add_u32_dpp): every lane receives v + v[partner]
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
