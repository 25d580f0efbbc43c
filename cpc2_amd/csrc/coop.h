// Pieces shared by the cooperative (weights-in-registers) recurrent kernels of gru.hip and lstm.hip.
#pragma once
#include "common.h"

namespace cpc {

// A GROUP of G workgroups (512 threads each, one per CU) shares NB windows.  Forward: thread (u, q) = unit u of the
// member's U units, K slice q of 32 columns, all gates.  H = 256: 8 slices, U = 64, G = 4;  H = 512: 16 slices,
// U = 32, G = 16.
template <int H> struct CoopCfg {
    static constexpr int QS = H / 32;            // K slices of 32 columns
    static constexpr int U = 512 / QS;           // units per member
    static constexpr int G = H / U;              // workgroups per group
    static constexpr int LDH = QS * 36 + 4;      // padded h row: chunk q of 32 floats at q*36; rows 4 banks apart (the gather
                                                 // writes the NB windows of a unit from neighbouring lanes)
    static constexpr int HALVES = 512 / H;       // backward: threads per W_hh column
};

typedef unsigned long long gu64_t;               // {epoch, value} granule
#define COOP_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ int coop_pad(int k) { return (k >> 5) * 36 + (k & 31); }

// Forward granules: [group][set][member][unit u][window].  A member's piece of a set is ONE run of NB * U granules that nobody
// else writes, and what a wave publishes (its 4 or 8 units x NB windows) is whole 128-byte lines.  Round 2 had
// [group][set][window][H]: 32-byte pieces of lines shared with other members, which with a group on one XCD cost 5.4 us per
// step at H = 512 against 2.35 (tools/exchange_probe.hip, profiles/r03_exchange_probe.txt).
template <int H, int NB> __device__ __forceinline__ long coop_fwd_slot(int group, int set, int member, int u, int window)
{
    return (((long)group * 2 + set) * CoopCfg<H>::G + member) * (NB * CoopCfg<H>::U) + u * NB + window;
}
// granule idx (0 .. NB * H) of a set, in memory order -> window and unit k (0 .. H) of the group
template <int H, int NB> __device__ __forceinline__ void coop_fwd_who(int idx, int &window, int &k)
{
    constexpr int U = CoopCfg<H>::U;
    const int g = idx / (NB * U), p = idx - g * (NB * U);
    window = p % NB;
    k = g * U + p / NB;
}

// two f32 lanes as one value.  (Device code is built WITHOUT the packed-f32 VALU instructions -- build.py, DEVICE_FLAGS: round 3's
// launch-to-launch differences -- so pk_fma compiles to two v_fma_f32; the type is kept for the register-pair layout of the weights.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_lo(const float4 &v) { return f32x2{v.x, v.y}; }
__device__ __forceinline__ f32x2 pk_hi(const float4 &v) { return f32x2{v.z, v.w}; }

// Barrier between the phases of a time step.  The waves of a member talk to each other through LDS only, so only LDS
// traffic has to have landed: __syncthreads() also waits for every global load and store of the wave (vmcnt(0)) -- the loads
// requested a step ahead, the saved activations nobody waits for -- and put a memory round trip on every step's serial
// path (H = 512: 2.2 us of the backward step's 8.2, profiles/r03_gru_stamps.txt).
__device__ __forceinline__ void coop_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The weights are loaded once, before the time loop; used here so that the compiler waits for them HERE.  Left alone it waits
// for the last of them at their first use inside the loop, with a vmcnt that counts the memory operations issued behind them
// -- which in steady state are the loop's own requests and stores: every step then sat out a memory round trip in the middle
// of its arithmetic (gru_bwd_coop_kernel<512>: s_waitcnt vmcnt(7) .. vmcnt(1) between the FMAs).
template <int N> __device__ __forceinline__ void coop_weights_ready(f32x2 (&w)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(w[i]));
}

template <int CTRL> __device__ __forceinline__ float dpp_get(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over an aligned group of QS = 8 or 16 consecutive lanes, left in every lane of the group: data-parallel
// primitives (VALU speed) instead of ds_bpermute round trips through the LDS crossbar
template <int QS> __device__ __forceinline__ float coop_group_sum(float v)
{
    static_assert(QS == 8 || QS == 16, "half a DPP row or a full one");
    v += dpp_get<0xB1>(v);                  // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);                  // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);                 // row_half_mirror: the other quad of the 8
    if (QS == 16) v += dpp_get<0x140>(v);   // row_mirror: the other half of the 16
    return v;
}

template <int G> __device__ __forceinline__ void coop_who(int groups, int xcd_map, int &group, int &member)
{
    if (xcd_map) {                          // members of a group on one XCD (speed only: blocks b, b+8 share one)
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        group = xcd * (groups / 8) + i / G;
        member = i % G;
    } else {
        group = blockIdx.x / G;
        member = blockIdx.x % G;
    }
}

// window count -> windows per group (0: not covered), for the hidden sizes that have a cooperative kernel; every
// workgroup must be resident at once (1 per CU)
static inline int coop_windows_per_group(int H, int N, int n_cus, int *groups_per)
{
    const int G = H == 256 ? CoopCfg<256>::G : (H == 512 ? CoopCfg<512>::G : 0);
    if (G == 0 || n_cus < G) return 0;
    const int max_groups = n_cus / G;
    static const int nb_min = getenv("CPC_COOP_NB_MIN") != nullptr ? atoi(getenv("CPC_COOP_NB_MIN")) : 1;      // A/B switch
    for (int nb = 1; nb <= 8; nb *= 2)
        if (nb >= nb_min && (int)cdiv(N, nb) <= max_groups) { *groups_per = G; return nb; }
    return 0;
}

static inline int coop_cu_count()
{
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return v;
}

// ---- failing loudly -------------------------------------------------------------------------------------------------
// The cooperative kernels need every workgroup of a launch resident at once.  (1) Before choosing them the launcher asks
// the occupancy API whether the kernel fits a CU at all and compares the grid with the CU count (coop_fits).  (2) Every
// wait is bounded; a wave whose wait runs out still poisons its outputs with NaN, and writes a code into ONE host-visible
// word (mapped, coherent host memory: no synchronisation needed to read it).  The recurrent entry points look at that word
// before they launch anything and cpc_async_error_check(stream) after synchronising the stream: both return CPC_ERR_HIP
// with a message and clear the word -- like a HIP asynchronous error, a time-out surfaces at the next call.
int *coop_error_word();                         // device-visible address of the word (nullptr: allocation failed)
int coop_error_take(const char *where);         // host: CPC_OK, or CPC_ERR_HIP (message set, word cleared)
int coop_fault_injection();                     // tests: CPC_COOP_FAULT=1 makes member 0 of group 0 withhold one publish
// Granule memory of the cooperative kernels: ONE buffer per (device, stream) that the library owns and nothing else ever writes,
// zeroed once; every launch gets a range of epochs of its own ([*epoch0 + 1, *epoch0 + T]), so a granule left by an earlier launch
// can never be taken for a current one and nothing has to be cleared between launches (rounds 2-4 carved the granules out of the
// caller's scratch arena -- which other kernels write -- and cleared them in front of every launch: two memsets of 7-14 us on the
// step's critical path).  The buffer grows when a launch needs more; the epoch counter wraps by clearing it once.
int coop_comm_acquire(size_t bytes, int T, hipStream_t st, gu64_t **comm, unsigned *epoch0);
bool coop_allowed();                            // process-wide policy (cpc_coop_set_policy): false = streaming kernels only
void coop_count_launch();                       // every cooperative recurrent launch is counted (cpc_coop_launches)
long coop_launches();
void coop_count_backward_call();                 // every recurrent backward entry (any kernel kind) is counted (cpc_recurrent_backward_calls)
enum { COOP_ERR_FWD_WAIT = 1, COOP_ERR_BWD_WAIT = 2, COOP_ERR_NONFINITE_GRAD = 3, COOP_ERR_BAD_INDEX = 4 };     // 3: the Adam kernel (rowops.hip), 4: the criterion's index check (infonce.hip)
__device__ __forceinline__ void coop_report(int *err, int code)
{
    if (err != nullptr) __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename K> static inline bool coop_fits(K kernel, unsigned grid, int n_cus)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 512, 0) != hipSuccess) return false;
    return per_cu >= 1 && (int)grid <= n_cus;   // one workgroup per CU is what the grouping assumes
}

// granules of the backward kernels: [groups][2][G][NB][H] with groups*NB < N + 8 windows and G <= 16
static inline size_t coop_comm_bytes(int H, int N)
{
    return (H == 256 || H == 512) ? sizeof(gu64_t) * 2 * 16 * (size_t)(N + 8) * H : 256;
}

}  // namespace cpc
