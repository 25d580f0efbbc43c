// LDS-DMA and LDS reads issued from inline asm (gfx950).  hipcc drains vmcnt to 0 before any LDS read it can see behind an
// LDS-DMA, which ends every load's flight at the next use of LDS; with the DMA, the reads and their waits all in asm it
// sees none of them, and the waits are ours: s_waitcnt vmcnt(N) counted by hand (loads, stores and LDS-DMA count together,
// in issue order), then -- where another wave reads -- a barrier.
#pragma once
#include <hip/hip_runtime.h>

namespace cpc {

typedef unsigned frag_t __attribute__((ext_vector_type(4)));   // 16 bytes as four dwords (an MFMA operand or a float4)

// one LDS-DMA piece: lane l writes 16 bytes at lds_dst + 16 l, read from sbase + voff (bytes); lds_dst and sbase wave-uniform
__device__ __forceinline__ void glds16(unsigned lds_dst, unsigned voff, const void *sbase)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(sbase)
                 : "memory");
}
// the same with a 64-bit per-lane address (rows gathered by index: every lane its own row)
__device__ __forceinline__ void glds16_addr(unsigned lds_dst, const void *lane_src)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(lane_src)
                 : "memory");
}

template <int OFF> __device__ __forceinline__ frag_t lds_read16(unsigned addr)
{
    frag_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long long)p; }

}  // namespace cpc
