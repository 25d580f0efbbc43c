// Row-kernel helpers shared by the encoder and transformer kernels: a row of H channels is owned by a
// group of G lanes, 4*VPL channels per lane, as float4s v*G + gl (v < VPL) so a group's accesses are contiguous.
#pragma once
#include <hip/hip_runtime.h>

namespace cpc {

template <int H> struct RowCfg {
    static_assert(H % 32 == 0, "row width must be a multiple of 32");
    static constexpr int G = (H / 4 < 64) ? H / 4 : 64;
    static constexpr int VPL = H / (4 * G);
    static constexpr int RPW = 64 / G;   // rows per wave pass
};

template <int CTRL> __device__ __forceinline__ float row_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over an aligned group of G lanes (8, 16, 32 or 64), left in every lane.  Inside a row of 16 lanes the partners
// come through data-parallel primitives (VALU speed); only the 16- and 32-lane hops go through the LDS crossbar.
template <int G> __device__ __forceinline__ float group_sum(float v)
{
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "group of 8, 16, 32 or 64 lanes");
    v += row_dpp<0xB1>(v);                      // quad_perm [1,0,3,2]
    v += row_dpp<0x4E>(v);                      // quad_perm [2,3,0,1]
    v += row_dpp<0x141>(v);                     // row_half_mirror
    if (G >= 16) v += row_dpp<0x140>(v);        // row_mirror
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// max over an aligned group of 32 lanes, left in every lane (same hops as group_sum)
__device__ __forceinline__ float group_max32(float v)
{
    v = fmaxf(v, row_dpp<0xB1>(v));
    v = fmaxf(v, row_dpp<0x4E>(v));
    v = fmaxf(v, row_dpp<0x141>(v));
    v = fmaxf(v, row_dpp<0x140>(v));
    return fmaxf(v, __shfl_xor(v, 16, 64));
}

static inline bool supported_row_width(int H) { return H == 32 || H == 64 || H == 128 || H == 256 || H == 512; }
static inline int rows_per_wave(int H) { return 64 / (H / 4 < 64 ? H / 4 : 64); }

#define CPC_DISPATCH_H(H, ...)                                  \
    switch (H) {                                                \
    case 32: { constexpr int HH = 32; __VA_ARGS__; } break;    \
    case 64: { constexpr int HH = 64; __VA_ARGS__; } break;    \
    case 128: { constexpr int HH = 128; __VA_ARGS__; } break;  \
    case 256: { constexpr int HH = 256; __VA_ARGS__; } break;  \
    case 512: { constexpr int HH = 512; __VA_ARGS__; } break;  \
    default: break;                                             \
    }

}  // namespace cpc
