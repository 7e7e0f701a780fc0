// Wave-level sums on gfx950 without LDS round trips: DPP inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap across rows.
// Every lane must be active (a DPP source lane that is not returns 0).
#pragma once
#include <hip/hip_runtime.h>

namespace aar {

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {   // lane permutation inside a row of 16 lanes (every lane has a source)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;

// sum over the 8 lanes of a row group (result in all 8)
__device__ __forceinline__ double sum8(double v) {
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    return v;
}
// v(row r of 16 lanes) <- v(r) + v(r ^ 1), then + the other pair of rows: gfx950's v_permlane16_swap / v_permlane32_swap exchange whole
// rows between two registers, so a cross-row all-reduce of a 64-bit value is 4 + 4 exchanges and two additions, in the same order
// in every row (same bits everywhere)
__device__ __forceinline__ double sum_across_rows(double v) {
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    return v;
}
// sum over the 64 lanes of the wavefront (result in every lane)
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = sum8(v);
    v += dpp<DPP_ROR8>(v);
    return sum_across_rows(v);
}

}  // namespace aar
