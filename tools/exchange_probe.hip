// What does one step of the cooperative recurrent kernels' hand-off cost, by itself, and which form of it is cheapest?
// (gru.hip / lstm.hip: a GROUP of G workgroups, one per CU, exchanges P new values per member and step; every member
// needs all G * P of them before its next step.)  No arithmetic here: a step is publish -> gather -> LDS -> barrier, plus
// an optional busy wait that stands for the step's arithmetic.
//
//   form 0  8-byte {epoch, value} granules, laid out [window][H] as in round 2 (a member's piece of a window row is U * 8
//           bytes; a wave stores 8 pieces of 32 bytes), two sets (epoch parity)
//   form 1  the same granules, member-major [member][P]: whole 128-byte lines per store instruction
//   form 2  4-byte values, "not written yet" = a reserved NaN pattern; three sets, a member re-arms its piece of the set
//           after next at the start of a step (ordered before its next publish by a vmcnt wait)
//   form 3  4-byte values + one flag per member and step: data stores, vmcnt(0), barrier, flag store; readers poll the G
//           flags, then load the data
//   form 4  form 1 with workgroup-scope cache bits (sc0) instead of agent scope (sc1) on the granule loads and stores: meant
//           for a group whose members share an XCD, i.e. one L2 -- does the load then come from L2 instead of the fabric?
//   form 5  form 1 with plain stores and, per polling attempt, buffer_inv sc0 (drop the CU's vector L1) + plain loads
//   form 6  agent-scope (sc1) stores, sc0 loads          form 7  agent-scope stores, buffer_inv sc0 + plain loads
//
//   hipcc -O3 --offload-arch=gfx950 tools/exchange_probe.hip -o tools/exchange_probe.bin && tools/exchange_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
#define GLOBAL_AS __attribute__((address_space(1)))
constexpr unsigned ARMED = 0xFFFFFFFFu;

template <int FORM> __device__ __forceinline__ void granule_store(u64 *p, u64 v)
{
    if (FORM == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (FORM == 5) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if (FORM >= 6) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else __hip_atomic_store((GLOBAL_AS u64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Args {
    u64 *comm;            // forms 0, 1
    unsigned *comm32;     // forms 2, 3
    unsigned *flags;      // form 3: [groups][2][G]
    int G, P, U, H, NB, groups, steps, xcd_map, math_ticks;   // math_ticks: busy wait per step, 10 ns ticks
    int *err;
    u64 *ticks;           // [blocks]: 10 ns ticks of the whole loop
};

template <int FORM, int KP> __global__ __launch_bounds__(512) void xchg(Args a)
{
    __shared__ float hs[2][4096];
    const int tid = threadIdx.x;
    int group, member;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        group = xcd * (a.groups / 8) + i / a.G;
        member = i % a.G;
    } else {
        group = blockIdx.x / a.G;
        member = blockIdx.x % a.G;
    }
    const int GP = a.G * a.P;                                // = 512 KP
    for (int i = tid; i < GP; i += 512) hs[0][i] = 0.25f;
    __syncthreads();
    // publishing role.  forms 0: thread (u = tid / QS, q = tid % QS), q < NB publishes (window q, unit u); others: tid < P
    const int QS = a.H / 32;
    const int u = tid / QS, q = tid % QS;
    const bool pub0 = q < a.NB && u < a.U;
    bool dead = false;
    const u64 t_begin = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < a.steps; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const unsigned epoch = (unsigned)(t + 1);
        if (FORM == 2 && tid < a.P) {                       // re-arm this member's piece of the set after next
            GLOBAL_AS unsigned *slot = (GLOBAL_AS unsigned *)(a.comm32 + ((long)group * 3 + (epoch + 1) % 3) * GP + member * a.P + tid);
            __hip_atomic_store(slot, ARMED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.math_ticks > 0) {
            const u64 until = __builtin_amdgcn_s_memrealtime() + (u64)a.math_ticks;
            while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(1);
        }
        const float v = hs[cur][(tid * 7 + t) % GP] * 0.5f + 0.125f;
        if (FORM == 0) {
            if (pub0) {
                GLOBAL_AS u64 *slot = (GLOBAL_AS u64 *)(a.comm + (((long)group * 2 + nxt) * a.NB + q) * a.H + member * a.U + u);
                __hip_atomic_store(slot, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (FORM == 1 || FORM >= 4) {
            if (tid < a.P)
                granule_store<FORM>(a.comm + ((long)group * 2 + nxt) * GP + member * a.P + tid, ((u64)epoch << 32) | (u64)__float_as_uint(v));
        } else if (FORM == 2) {
            if (tid < a.P) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the re-arming stores of this step are acknowledged
                GLOBAL_AS unsigned *slot = (GLOBAL_AS unsigned *)(a.comm32 + ((long)group * 3 + epoch % 3) * GP + member * a.P + tid);
                __hip_atomic_store(slot, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (tid < a.P) {
                GLOBAL_AS unsigned *slot = (GLOBAL_AS unsigned *)(a.comm32 + ((long)group * 2 + nxt) * GP + member * a.P + tid);
                __hip_atomic_store(slot, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store((GLOBAL_AS unsigned *)(a.flags + ((long)group * 2 + nxt) * a.G + member), epoch, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        // gather
        unsigned spins = dead ? (1u << 20) : 0u;
        if (FORM <= 1 || FORM >= 4) {
            u64 x[KP];
            if (tid < GP) {
                for (;;) {
                    bool ready = true;
                    if (FORM == 5 || FORM == 7) asm volatile("buffer_inv sc0" ::: "memory");
                    u64 *src[KP];
#pragma unroll
                    for (int i = 0; i < KP; ++i) {
                        const int idx = tid + 512 * i;
                        long off;
                        if (FORM == 0) off = (((long)group * 2 + nxt) * a.NB + idx / a.H) * a.H + idx % a.H;
                        else off = ((long)group * 2 + nxt) * GP + idx;
                        src[i] = a.comm + off;
                    }
                    if (FORM >= 4) {
                        // no branch between a load issued from asm and the wait: the compiler believes the register is written
                        // where the asm statement stands
#pragma unroll
                        for (int i = 0; i < KP; ++i) {
                            if (FORM == 4 || FORM == 6) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(x[i]) : "v"(src[i]) : "memory");
                            else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(x[i]) : "v"(src[i]) : "memory");
                        }
                        if constexpr (KP == 8) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : : "memory");
                        else if constexpr (KP == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : : "memory");
                        else if constexpr (KP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]) : : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]) : : "memory");
                    } else {
#pragma unroll
                        for (int i = 0; i < KP; ++i) x[i] = __hip_atomic_load((GLOBAL_AS u64 *)src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int i = 0; i < KP; ++i) ready = ready && (unsigned)(x[i] >> 32) == epoch;
                    if (ready) break;
                    if (++spins > (1u << 20)) { dead = true; *a.err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int i = 0; i < KP; ++i) hs[nxt][tid + 512 * i] = __uint_as_float((unsigned)x[i]);
            }
        } else if (FORM == 2) {
            unsigned x[KP];
            if (tid < GP) {
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int i = 0; i < KP; ++i)
                        x[i] = __hip_atomic_load((GLOBAL_AS unsigned *)(a.comm32 + ((long)group * 3 + epoch % 3) * GP + tid + 512 * i),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < KP; ++i) ready = ready && x[i] != ARMED;
                    if (ready) break;
                    if (++spins > (1u << 20)) { dead = true; *a.err = 2; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int i = 0; i < KP; ++i) hs[nxt][tid + 512 * i] = __uint_as_float(x[i]);
            }
        } else {
            if (tid < a.G) {
                for (;;) {
                    const unsigned f = __hip_atomic_load((GLOBAL_AS unsigned *)(a.flags + ((long)group * 2 + nxt) * a.G + tid), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                    if (f == epoch) break;
                    if (++spins > (1u << 20)) { dead = true; *a.err = 3; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (tid < GP) {
#pragma unroll
                for (int i = 0; i < KP; ++i)
                    hs[nxt][tid + 512 * i] = __uint_as_float(__hip_atomic_load(
                        (GLOBAL_AS unsigned *)(a.comm32 + ((long)group * 2 + nxt) * GP + tid + 512 * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.ticks[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t_begin;
}

template <int FORM> static void run(const char *name, Args a, int blocks, size_t comm_bytes)
{
    hipMemset(a.comm, 0, comm_bytes);
    hipMemset(a.comm32, 0xFF, comm_bytes);       // form 2: everything armed;  form 3 overwrites before it reads
    hipMemset(a.flags, 0, 1 << 20);
    hipMemset(a.err, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int kp = a.G * a.P / 512;
    if (kp == 8) hipLaunchKernelGGL((xchg<FORM, 8>), dim3(blocks), dim3(512), 0, 0, a);
    else if (kp == 4) hipLaunchKernelGGL((xchg<FORM, 4>), dim3(blocks), dim3(512), 0, 0, a);
    else if (kp == 2) hipLaunchKernelGGL((xchg<FORM, 2>), dim3(blocks), dim3(512), 0, 0, a);
    else if (kp == 1) hipLaunchKernelGGL((xchg<FORM, 1>), dim3(blocks), dim3(512), 0, 0, a);
    else { printf("  unsupported shape\n"); return; }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    int err = 0;
    hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost);
    std::vector<u64> ticks(blocks);
    hipMemcpy(ticks.data(), a.ticks, blocks * sizeof(u64), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (u64 v : ticks) { mean += (double)v; mx = mx > (double)v ? mx : (double)v; }
    mean /= blocks;
    printf("  %-34s %6.2f us/step (slowest workgroup %6.2f; kernel %7.3f ms)%s\n", name, mean * 0.01 / a.steps, mx * 0.01 / a.steps, ms,
           err ? "  TIMED OUT" : "");
}

// which XCD does workgroup b of a 256-workgroup launch (512 threads, one per CU) run on?  coop_who() assumes b % 8
__global__ __launch_bounds__(512) void where_kernel(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw;
    }
    // keep the CU busy for a while so that every workgroup gets its own
    const u64 until = __builtin_amdgcn_s_memrealtime() + 2000;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(4);
}

int main(int argc, char **argv)
{
    {
        unsigned *d;
        hipMalloc(&d, 256 * 8);
        hipLaunchKernelGGL(where_kernel, dim3(256), dim3(512), 0, 0, d);
        std::vector<unsigned> h(512);
        hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
        int agree = 0;
        printf("XCC_ID & 15 of workgroups 0..31:");
        for (int b = 0; b < 32; ++b) printf(" %u", h[2 * b] & 15);
        for (int b = 0; b < 256; ++b) agree += (int)((h[2 * b] & 15) == (unsigned)(b & 7));
        printf("\nworkgroups with XCC_ID == b %% 8: %d of 256 (raw XCC_ID of workgroup 0: 0x%x, HW_ID 0x%x)\n", agree, h[0], h[1]);
    }
    const size_t comm_bytes = 64u << 20;
    Args a{};
    hipMalloc(&a.comm, comm_bytes); hipMalloc(&a.comm32, comm_bytes); hipMalloc(&a.flags, 1 << 20);
    hipMalloc(&a.err, 4); hipMalloc(&a.ticks, 4096 * sizeof(u64));
    a.steps = 512;
    const bool quick = argc > 1;
    struct Shape { int H, NB; } shapes[] = {{256, 2}, {512, 8}, {512, 4}};
    for (const Shape &s : shapes)
        for (int math : {0, 240})
            for (int xm : {1, 0}) {
                a.H = s.H; a.NB = s.NB;
                const int QS = s.H / 32;
                a.U = 512 / QS; a.G = s.H / a.U; a.P = s.NB * a.U;
                a.groups = 256 / a.G;
                a.xcd_map = xm; a.math_ticks = math;
                if (a.G * a.P > 4096) continue;
                printf("H %d, %d windows per group: G %d members x P %d values, %d groups, busy wait %.1f us, members %s\n", s.H, s.NB, a.G, a.P, a.groups,
                       math * 0.01, xm ? "on one XCD" : "in dispatch order (spread over the XCDs)");
                const int blocks = a.groups * a.G;
                run<0>("8-byte granules, [window][H]", a, blocks, comm_bytes);
                run<1>("8-byte granules, member-major", a, blocks, comm_bytes);
                run<2>("4-byte values, armed pattern", a, blocks, comm_bytes);
                run<3>("4-byte values + flag per member", a, blocks, comm_bytes);
                if (!quick || !xm) continue;               // the two below time out (1 s each): see profiles/r03_exchange_probe.txt
                run<4>("8-byte granules, member-major, sc0", a, blocks, comm_bytes);
                run<5>("... plain + buffer_inv sc0 per poll", a, blocks, comm_bytes);
                run<6>("... sc1 stores, sc0 loads", a, blocks, comm_bytes);
                run<7>("... sc1 stores, inv sc0 + plain loads", a, blocks, comm_bytes);
            }
    return 0;
}
