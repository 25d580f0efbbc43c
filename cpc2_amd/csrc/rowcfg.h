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

template <int G> __device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
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
