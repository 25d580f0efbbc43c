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

// Eight pieces of one stream element in ONE statement: lane l of piece i writes 16 bytes at lds_dst + i * PSTEP + 16 l, read
// from sbase + vo[i] + GOFF.  M0 is saved and restored once and stepped by an s_add between the pieces (the single-piece form
// spends four scalar instructions per piece on it), and the per-lane offsets are inputs: no vector arithmetic between the
// pieces.  GOFF (0 <= GOFF < 4096) goes into the instruction's offset field, which moves the LDS address by the same amount
// (tools/scratch/ldsdma_offset_probe.hip) -- compensated in M0, which must stay >= 0: a negative M0 drops the write -- so the K
// slices of a row share one offset register.  NT: non-temporal loads (data read once -- prediction rows -- should not push the
// gathered table out of L2).
#define CPC_GLDS_FIRST(MOD) "s_mov_b32 %0, m0\n\ts_sub_u32 m0, %1, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %10 offset:%11" MOD "\n\t"
#define CPC_GLDS_NEXT(N, MOD) "s_add_u32 m0, m0, %12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %" #N ", %10 offset:%11" MOD "\n\t"
#define CPC_GLDS_BODY8(MOD)                                                                                                     \
    CPC_GLDS_FIRST(MOD) CPC_GLDS_NEXT(3, MOD) CPC_GLDS_NEXT(4, MOD) CPC_GLDS_NEXT(5, MOD) CPC_GLDS_NEXT(6, MOD) CPC_GLDS_NEXT(7, MOD)   \
    CPC_GLDS_NEXT(8, MOD) CPC_GLDS_NEXT(9, MOD)
#define CPC_GLDS_OPS                                                                                                              \
    : "=&s"(keep)                                                                                                                \
    : "s"(lds_dst), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(vo[4]), "v"(vo[5]), "v"(vo[6]), "v"(vo[7]), "s"(sbase), "n"(GOFF), \
      "n"(PSTEP)                                                                                                                 \
    : "memory", "scc"
template <int PSTEP, int GOFF, bool NT = false>
__device__ __forceinline__ void glds16x8(unsigned lds_dst, const unsigned (&vo)[8], const void *sbase)
{
    unsigned keep;
    if constexpr (NT) asm volatile(CPC_GLDS_BODY8(" nt") "s_mov_b32 m0, %0" CPC_GLDS_OPS);
    else asm volatile(CPC_GLDS_BODY8("") "s_mov_b32 m0, %0" CPC_GLDS_OPS);
}
#undef CPC_GLDS_OPS

// two dwords of LDS, `addr` + 4 * D0 and + 4 * D1 (asm for the same reason as lds_read16: no wait of the compiler's).  The
// PAIR must go through the asm wait as it is ("+v" on the u32x2): elements taken out before the wait are copies of registers
// the load has not written yet.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int D0, int D1> __device__ __forceinline__ u32x2_t lds_read2(unsigned addr)
{
    u32x2_t v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(D0), "n"(D1) : "memory");
    return v;
}

template <int OFF> __device__ __forceinline__ frag_t lds_read16(unsigned addr)
{
    frag_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long long)p; }

}  // namespace cpc
