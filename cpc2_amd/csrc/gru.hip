// CPCAR (mode="GRU") on gfx950.  Reference: /root/reference/cpc/model.py:158-207 -> torch.nn.GRU
// (batch_first, gate order r,z,n):
//     r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)        z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//     n = tanh(W_in x + b_in + r * (W_hn h + b_hn))      h' = (1 - z) * n + z * h
//
// Forward per layer: one MFMA GEMM for all input projections GI = X W_ih^T + b_ih, then ONE
// persistent kernel for the T sequential steps.  Windows are independent, so each workgroup owns
// one window for the whole sequence (no inter-workgroup synchronisation); thread (j, q) owns hidden
// unit j and K slice q, h lives in LDS and W_hh (re-laid out so that lanes read consecutive float4s)
// is streamed from L2 every step.  Backward mirrors it (BPTT), then three GEMMs give dW_hh, dW_ih and dX.
#include "common.h"
#include "coop.h"

#include <algorithm>
#include <cstdlib>

#ifndef GRU_ABL_F
#define GRU_ABL_F 8      // probes: K quarters of the slice actually multiplied (forward), of 8
#endif
#ifndef GRU_ABL_B
#define GRU_ABL_B 24     // probes: row quarters multiplied (backward), of 24
#endif

namespace cpc {

// W_hh [3H][H] -> wf[(k4*3 + g)*H + j] = W[g*H + j][4*k4 .. 4*k4+3]   (forward: thread j, all k)
__global__ void gru_pack_fwd_kernel(const float *w, float4 *wf, int H)
{
    const int total = 3 * H * (H / 4);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx % H;
        const int g = (idx / H) % 3;
        const int k4 = idx / (3 * H);
        const float *src = w + (long)(g * H + j) * H + 4 * k4;
        wf[idx] = make_float4(src[0], src[1], src[2], src[3]);
    }
}

// W_hh [3H][H] -> wb[g4*H + j] = (W[4*g4][j], W[4*g4+1][j], W[4*g4+2][j], W[4*g4+3][j])   (backward)
__global__ void gru_pack_bwd_kernel(const float *w, float4 *wb, int H)
{
    const int total = (3 * H / 4) * H;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx % H;
        const int g4 = idx / H;
        wb[idx] = make_float4(w[(long)(4 * g4) * H + j], w[(long)(4 * g4 + 1) * H + j], w[(long)(4 * g4 + 2) * H + j],
                              w[(long)(4 * g4 + 3) * H + j]);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct GruArgs {
    const float *gi;      // [N*T][3H]   input projections incl. b_ih
    const float4 *wpack;  // packed W_hh
    const float *whh;     // W_hh [3H][H] as stored (cooperative kernel)
    const float *bhh;     // [3H]
    const float *h0;      // [N][H] or null
    float *out;           // [N][T][H]
    float *hall;          // [N][T+1][H]  row 0 = h0, row t+1 = h_t
    float *gates;         // [N*T][3H]    r, z, n
    float *hn;            // [N*T][H]     W_hn h + b_hn
    float *hlast;         // [N][H] or null
    int N, T, H;
    int hp, kq;           // threads = kq * hp: hp = H rounded up to 64, kq = K-split factor
    // backward
    const float *dout;    // [N][T][H]
    float *dgi;           // [N*T][3H]
    float *dgh;           // [N][T+1][3H], row T zero
};

// One workgroup per window; thread (j, q): hidden unit j, K slice q.  The K split puts kq x more waves
// (and W_hh loads) in flight per CU: the step time is the L2 -> CU stream of W_hh (3H*H*4 bytes).
__global__ void gru_fwd_kernel(GruArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hs[H] | red[kq][3][hp]
    const int H = a.H, T = a.T, hp = a.hp, kq = a.kq;
    float *hs = smem;
    float *red = smem + ((H + 3) / 4) * 4;
    const int j = threadIdx.x % hp, q = threadIdx.x / hp;
    const bool act = j < H;
    const int n = blockIdx.x;
    const int k4_per = (H / 4 + kq - 1) / kq;
    const int k4_lo = q * k4_per, k4_hi = min(H / 4, k4_lo + k4_per);

    float hprev = 0.f;
    float bh[3] = {0.f, 0.f, 0.f};
    if (act && q == 0) {
        hprev = a.h0 != nullptr ? a.h0[(long)n * H + j] : 0.f;
        hs[j] = hprev;
        a.hall[((long)n * (T + 1)) * H + j] = hprev;
        bh[0] = a.bhh[j]; bh[1] = a.bhh[H + j]; bh[2] = a.bhh[2 * H + j];
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        if (act) {
            const float4 *wp = a.wpack + j;
#pragma unroll 4
            for (int k4 = k4_lo; k4 < k4_hi; ++k4) {
                const float4 wr = wp[(long)(k4 * 3 + 0) * H];
                const float4 wz = wp[(long)(k4 * 3 + 1) * H];
                const float4 wn = wp[(long)(k4 * 3 + 2) * H];
                const float4 h4 = reinterpret_cast<const float4 *>(hs)[k4];
                acc0 = fmaf(wr.x, h4.x, fmaf(wr.y, h4.y, fmaf(wr.z, h4.z, fmaf(wr.w, h4.w, acc0))));
                acc1 = fmaf(wz.x, h4.x, fmaf(wz.y, h4.y, fmaf(wz.z, h4.z, fmaf(wz.w, h4.w, acc1))));
                acc2 = fmaf(wn.x, h4.x, fmaf(wn.y, h4.y, fmaf(wn.z, h4.z, fmaf(wn.w, h4.w, acc2))));
            }
            red[(q * 3 + 0) * hp + j] = acc0;
            red[(q * 3 + 1) * hp + j] = acc1;
            red[(q * 3 + 2) * hp + j] = acc2;
        }
        __syncthreads();                       // partial sums visible; nobody reads hs any more
        if (act && q == 0) {
            float g0 = bh[0], g1 = bh[1], g2 = bh[2];
            for (int qq = 0; qq < kq; ++qq) {
                g0 += red[(qq * 3 + 0) * hp + j];
                g1 += red[(qq * 3 + 1) * hp + j];
                g2 += red[(qq * 3 + 2) * hp + j];
            }
            const long row = (long)n * T + t;
            const float *g = a.gi + row * 3 * H;
            const float r = sigmoidf_(g[j] + g0);
            const float z = sigmoidf_(g[H + j] + g1);
            const float c = tanhf(g[2 * H + j] + r * g2);
            const float hv = (1.f - z) * c + z * hprev;
            float *gs = a.gates + row * 3 * H;
            gs[j] = r; gs[H + j] = z; gs[2 * H + j] = c;
            a.hn[row * H + j] = g2;
            a.out[row * H + j] = hv;
            a.hall[((long)n * (T + 1) + t + 1) * H + j] = hv;
            hs[j] = hv;
            hprev = hv;
        }
        __syncthreads();
    }
    if (act && q == 0 && a.hlast != nullptr) a.hlast[(long)n * H + j] = hprev;
}

__global__ void gru_bwd_kernel(GruArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // dgh[3H] | red[kq][hp]
    const int H = a.H, T = a.T, hp = a.hp, kq = a.kq;
    float *dg = smem;
    float *red = smem + 3 * H;
    const int j = threadIdx.x % hp, q = threadIdx.x / hp;
    const bool act = j < H;
    const int n = blockIdx.x;
    const int g4_total = 3 * H / 4;
    const int g4_per = (g4_total + kq - 1) / kq;
    const int g4_lo = q * g4_per, g4_hi = min(g4_total, g4_lo + g4_per);

    float carry = 0.f;
    if (act && q == 0) {                                         // zero junk row T of dGH
        float *zr = a.dgh + ((long)n * (T + 1) + T) * 3 * H;
        zr[j] = 0.f; zr[H + j] = 0.f; zr[2 * H + j] = 0.f;
    }
    for (int t = T - 1; t >= 0; --t) {
        float keep = 0.f;
        if (act && q == 0) {
            const long row = (long)n * T + t;
            const float dh = a.dout[row * H + j] + carry;
            const float *gs = a.gates + row * 3 * H;
            const float r = gs[j], z = gs[H + j], c = gs[2 * H + j];
            const float hnv = a.hn[row * H + j];
            const float hp_ = a.hall[((long)n * (T + 1) + t) * H + j];
            const float dc = dh * (1.f - z);
            const float dz = dh * (hp_ - c);
            const float dpn = dc * (1.f - c * c);
            const float dpr = dpn * hnv * r * (1.f - r);
            const float dpz = dz * z * (1.f - z);
            const float dhn = dpn * r;
            keep = dh * z;
            float *gi = a.dgi + row * 3 * H;
            gi[j] = dpr; gi[H + j] = dpz; gi[2 * H + j] = dpn;
            float *gh = a.dgh + ((long)n * (T + 1) + t) * 3 * H;
            gh[j] = dpr; gh[H + j] = dpz; gh[2 * H + j] = dhn;
            dg[j] = dpr; dg[H + j] = dpz; dg[2 * H + j] = dhn;
        }
        __syncthreads();
        if (act) {
            float acc = 0.f;
            const float4 *wp = a.wpack + j;
#pragma unroll 4
            for (int g4 = g4_lo; g4 < g4_hi; ++g4) {
                const float4 w4 = wp[(long)g4 * H];
                const float4 d4 = reinterpret_cast<const float4 *>(dg)[g4];
                acc = fmaf(w4.x, d4.x, fmaf(w4.y, d4.y, fmaf(w4.z, d4.z, fmaf(w4.w, d4.w, acc))));
            }
            red[q * hp + j] = acc;
        }
        __syncthreads();
        if (act && q == 0) {
            float sum = keep;
            for (int qq = 0; qq < kq; ++qq) sum += red[qq * hp + j];
            carry = sum;
        }
        // the next iteration's writes to dg happen after every thread passed the barrier above; its reads of
        // red happen after the next two barriers -> no extra barrier needed here
    }
}

// ------------------------------------------------------------------------------------------------
// On-chip recurrence for H = 256 and 512: W_hh never leaves the register file.  A GROUP of G workgroups (512
// threads each, one per CU) shares NB windows; every thread holds 96 weights: thread (u, q) = unit u of the member's
// U units, K slice q of 32 columns, 3 gates.  H = 256: 8 slices, U = 64, G = 4;  H = 512: 16 slices, U = 32, G = 16.
// Per step every member multiplies its slice by the full h (LDS), finishes the K reduction with wave shuffles,
// applies the gates for its U units and publishes them; the members exchange the new h through 8-byte
// {epoch, value} granules in L2 (one sc1 store each, polled with relaxed agent-scope loads: MI355X guide,
// Guideline 16 R2).  A member can be at most one step ahead of another, so two granule sets (epoch parity)
// suffice.  Every spin is bounded: on time-out the workgroup poisons its outputs with NaN and leaves.
struct GruCoopArgs {
    GruArgs g;
    gu64_t *comm;          // granules (coop_comm_acquire: the library's own buffer): fwd [groups][2][G][U][NB] (coop_fwd_slot), bwd [groups][2][G][NB][H]
    unsigned epoch0;       // this launch's epochs are epoch0 + 1 .. epoch0 + T: no granule of an earlier launch carries one of them
    int groups, xcd_map;
    int *err;              // host-visible error word (coop.h), or nullptr
    int fault;             // tests: member 0 of group 0 withholds its publish of step 1
    unsigned long long *stamps;   // -DGRU_STAMPS builds (probes): [workgroup][8] ticks of 10 ns summed over the steps, per phase
};
#ifdef GRU_STAMPS
#define GRU_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); ph[i] += now_ - last_; last_ = now_; } while (0)
#else
#define GRU_STAMP(i) do { } while (0)
#endif

template <int H, int NB> __global__ __launch_bounds__(512) void gru_fwd_coop_kernel(GruCoopArgs ca)
{
    using C = CoopCfg<H>;
    constexpr int QS = C::QS, U = C::U, G = C::G;
    constexpr int KP = NB * H / 512 > 0 ? NB * H / 512 : 1;          // granules gathered per thread and step
    static_assert(NB <= QS, "one finishing lane per window");
    __shared__ __attribute__((aligned(16))) float hs[2][NB][C::LDH];
    const GruArgs &a = ca.g;
    const int T = a.T;
    int group, member;
    coop_who<G>(ca.groups, ca.xcd_map, group, member);
    const int tid = threadIdx.x;
    const int q = tid & (QS - 1), u = tid / QS;
    const int j = member * U + u;
    const int n0 = group * NB;

    f32x2 w[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i4 = 0; i4 < 8; ++i4) {
            const float4 v = *reinterpret_cast<const float4 *>(a.whh + (long)(g * H + j) * H + q * 32 + 4 * i4);
            w[g][2 * i4] = pk_lo(v); w[g][2 * i4 + 1] = pk_hi(v);
        }
    const float bh0 = a.bhh[j], bh1 = a.bhh[H + j], bh2 = a.bhh[2 * H + j];

    for (int idx = tid; idx < NB * H; idx += 512) {
        const int s = idx / H, k = idx - s * H;
        const int n = n0 + s;
        const float v = (n < a.N && a.h0 != nullptr) ? a.h0[(long)n * H + k] : 0.f;
        hs[0][s][coop_pad(k)] = v;
        if (n < a.N && (k / U) == member) a.hall[((long)n * (T + 1)) * H + k] = v;
    }
    __syncthreads();

    const int ns = n0 + q;
    const bool mine = q < NB && ns < a.N;
    float gin0 = 0.f, gin1 = 0.f, gin2 = 0.f;
    if (mine) {
        const float *gp = a.gi + (long)ns * T * 3 * H;
        gin0 = gp[j]; gin1 = gp[H + j]; gin2 = gp[2 * H + j];
    }
    bool dead = false;
#ifdef GRU_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // this lane finishes sample q (if q < NB); its input projections were requested one step ago, the next
        // step's are requested now: a load consumed in the step that issues it puts an L2/HBM round trip on the
        // serial path of every time step
        const float gi0 = gin0, gi1 = gin1, gi2 = gin2;
        if (mine && t + 1 < T) {
            const float *gp = a.gi + ((long)ns * T + t + 1) * 3 * H;
            gin0 = gp[j]; gin1 = gp[H + j]; gin2 = gp[2 * H + j];
        }
        float acc[NB][3];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            f32x2 a2[3] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};    // even / odd columns of the slice
#pragma unroll
            for (int i4 = 0; i4 < GRU_ABL_F; ++i4) {
                const float4 h4 = *reinterpret_cast<const float4 *>(&hs[cur][s][q * 36 + 4 * i4]);
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    a2[g] = pk_fma(w[g][2 * i4 + 1], pk_hi(h4), pk_fma(w[g][2 * i4], pk_lo(h4), a2[g]));
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[s][g] = coop_group_sum<QS>(a2[g].x + a2[g].y);
        }
        GRU_STAMP(0);
        if (q < NB) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int s = 0; s < NB; ++s)
                if (s == q) { g0 = acc[s][0]; g1 = acc[s][1]; g2 = acc[s][2]; }
            g0 += bh0; g1 += bh1; g2 += bh2;
            const float hp = hs[cur][q][coop_pad(j)];
            const float r = sigmoidf_(gi0 + g0);
            const float z = sigmoidf_(gi1 + g1);
            const float c = tanhf(gi2 + r * g2);
            float hv = (1.f - z) * c + z * hp;
            if (dead) hv = NAN;
            // publish first (also for padding windows, so that every granule of the epoch gets written): the other
            // members wait for this store, nobody waits for the saved activations below
            COOP_GLOBAL gu64_t *slot = (COOP_GLOBAL gu64_t *)(ca.comm + coop_fwd_slot<H, NB>(group, nxt, member, u, q));
            if (!(ca.fault && group == 0 && member == 0 && t == 1))
                __hip_atomic_store(slot, ((gu64_t)(ca.epoch0 + (unsigned)(t + 1)) << 32) | (gu64_t)__float_as_uint(mine ? hv : 0.f),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine) {
                const long row = (long)ns * T + t;
                float *gs = a.gates + row * 3 * H;
                gs[j] = r; gs[H + j] = z; gs[2 * H + j] = c;
                a.hn[row * H + j] = g2;
                a.out[row * H + j] = hv;
                a.hall[((long)ns * (T + 1) + t + 1) * H + j] = hv;
            }
        }
        GRU_STAMP(1);
        // gather the whole new h (all members) into the other LDS buffer; a thread's KP granules are polled together
        // (one L2 round trip per attempt, not KP in a row)
        if (t + 1 < T) {
            COOP_GLOBAL gu64_t *slot[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int idx = tid + 512 * i;
                slot[i] = (COOP_GLOBAL gu64_t *)(ca.comm + ((long)group * 2 + nxt) * (NB * H) + idx);      // memory order: coalesced
            }
            gu64_t x[KP];
            unsigned spins = dead ? (1u << 22) : 0u;            // once timed out, never wait again
            if (tid < NB * H) {
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int i = 0; i < KP; ++i) x[i] = __hip_atomic_load(slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < KP; ++i) ready = ready && (unsigned)(x[i] >> 32) == ca.epoch0 + (unsigned)(t + 1);
                    if (ready) break;
                    if (++spins > (1u << 22)) { dead = true; coop_report(ca.err, COOP_ERR_FWD_WAIT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#ifdef GRU_STAMPS
                ph[4] += spins;
#endif
                GRU_STAMP(2);
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int idx = tid + 512 * i;
                    int gw, gk;
                    coop_fwd_who<H, NB>(idx, gw, gk);
                    hs[nxt][gw][coop_pad(gk)] = dead ? NAN : __uint_as_float((unsigned)x[i]);
                }
            }
            coop_lds_barrier();            // `dead` stays with the thread that timed out: what it gathered is poisoned above
            GRU_STAMP(3);
        }
    }
#ifdef GRU_STAMPS
    if (ca.stamps != nullptr && (tid == 0 || tid == 511))
        for (int i = 0; i < 5; ++i) ca.stamps[(blockIdx.x * 2 + (tid != 0)) * 8 + i] = ph[i];
#endif
    if (a.hlast != nullptr && q < NB && n0 + q < a.N)
        a.hlast[(long)(n0 + q) * H + j] = a.hall[((long)(n0 + q) * (T + 1) + T) * H + j];
}

// ------------------------------------------------------------------------------------------------
// The same recurrence with the step's product on the bf16 matrix pipe (many windows per group: H = 512, NB = 8, where the
// VALU form spends 2.4 of a 5.8 us step on 768 FMAs per thread).  Per member and step out[16][3U] = A[16][H] B[H][3U]:
//   B = the member's 3U rows of W_hh (column c = gate * U + unit) as the three bf16 terms of every weight, held in registers
//       as v_mfma_f32_16x16x32_bf16 operands: the 3U / 16 column tiles x 4 (or 2) K parts make 24 units of 4 K steps, three
//       per wave -- three tiles of ONE K part -- (144 VGPRs);
//   A = the three bf16 terms of h (LDS, [plane][window][H + 8]), with windows 0-7 of term 0 in rows 0-7 and of term 1 -- or 2 --
//       in rows 8-15, so that FOUR products per K step give the six of the exact split (gemm_f32.hip):
//       [h0|h1] w0 -> h0w0, h1w0;  [h0|h1] w1 -> h0w1, h1w1;  [h0|h2] w2 -> h0w2 (+ h2w2: 2^-32, harmless);  [h2|0] w0 -> h2w0.
// The units' 16 x 16 partial tiles go through LDS ([unit][column][row]); the lane that finishes (window q, unit u) adds the
// K parts and the two row halves in a fixed order.  Everything else -- granules, polling, saved activations -- is the
// cooperative kernel's.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct MfmaFrag { unsigned d[4]; };                     // 8 bf16: one operand of v_mfma_f32_16x16x32_bf16

__device__ __forceinline__ unsigned short bf16_rne(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// v = t0 + t1 + t2 (bf16 each, round to nearest): exact to 2^-27 |v| (gemm_f32.hip)
__device__ __forceinline__ void split3(float v, unsigned short (&t)[3])
{
    t[0] = bf16_rne(v);
    v -= __uint_as_float((unsigned)t[0] << 16);
    t[1] = bf16_rne(v);
    v -= __uint_as_float((unsigned)t[1] << 16);
    t[2] = bf16_rne(v);
}

template <int H> struct MfmaCfg {
    using C = CoopCfg<H>;
    static constexpr int N3 = 3 * C::U;                  // columns of the member: gate * U + unit
    static constexpr int TILES = N3 / 16;                // 6 (H = 512), 12 (H = 256)
    static constexpr int KS = 24 / TILES;                // K parts per tile: 24 units for 8 waves
    static constexpr int KPU = (H / 32) / KS;            // K steps of 32 per unit
    static constexpr int LDK = H + 8;                    // bf16 per (plane, window) row: rows 4 banks apart
    static_assert(N3 % 16 == 0 && 24 % TILES == 0 && (H / 32) % KS == 0 && KPU == 4, "24 units of 4 K steps");
};

template <int H, int NB> __global__ __launch_bounds__(512) void gru_fwd_mfma_kernel(GruCoopArgs ca)
{
    using C = CoopCfg<H>;
    using M = MfmaCfg<H>;
    constexpr int QS = C::QS, U = C::U, G = C::G;
    constexpr int KP = NB * H / 512 > 0 ? NB * H / 512 : 1;
    constexpr int LDK = M::LDK;
    static_assert(NB <= 8 && NB <= QS, "eight windows fill half the rows of a 16-row tile");
    __shared__ __attribute__((aligned(16))) unsigned short hp[2][3][8][LDK];     // the three terms of h, two steps
    __shared__ __attribute__((aligned(16))) unsigned short zrow[LDK];            // rows 8-15 of the fourth product
    __shared__ __attribute__((aligned(16))) float part[24][16][16];              // [unit][column][row]
    const GruArgs &a = ca.g;
    const int T = a.T;
    int group, member;
    coop_who<G>(ca.groups, ca.xcd_map, group, member);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = tid & (QS - 1), u = tid / QS;
    const int j = member * U + u;
    const int n0 = group * NB;

    // ---- this wave's three units: B operands (all three terms) of 4 K steps each
    MfmaFrag bw[3][4][3];
    // (the three units of a wave share their K part: its A fragments are read once per K step, not once per unit -- the
    //  step's arithmetic is bound by LDS reads, 295 KB per member and step otherwise)
    const int kpart = wave % M::KS, tile0 = 3 * (wave / M::KS);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int tile = tile0 + i;
        const int col = tile * 16 + (lane & 15);
        const long row = (long)((col / U) * H + member * U + (col % U)) * H;          // row of W_hh
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = (kpart * 4 + ks) * 32 + 8 * (lane >> 4);
            const float4 v0 = *reinterpret_cast<const float4 *>(a.whh + row + k);
            const float4 v1 = *reinterpret_cast<const float4 *>(a.whh + row + k + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            unsigned short t[8][3];
#pragma unroll
            for (int e = 0; e < 8; ++e) split3(v[e], t[e]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int d = 0; d < 4; ++d) bw[i][ks][pl].d[d] = (unsigned)t[2 * d][pl] | ((unsigned)t[2 * d + 1][pl] << 16);
        }
    }
    const float bh0 = a.bhh[j], bh1 = a.bhh[H + j], bh2 = a.bhh[2 * H + j];

    // ---- h0 -> planes of step 0 (windows >= NB: zeros), the zero row
    for (int idx = tid; idx < 8 * H; idx += 512) {
        const int s = idx / H, k = idx - s * H;
        const int n = n0 + s;
        const float v = (s < NB && n < a.N && a.h0 != nullptr) ? a.h0[(long)n * H + k] : 0.f;
        unsigned short t[3];
        split3(v, t);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { hp[0][pl][s][k] = t[pl]; hp[1][pl][s][k] = 0; }
        if (s < NB && n < a.N && (k / U) == member) a.hall[((long)n * (T + 1)) * H + k] = v;
    }
    for (int k = tid; k < LDK; k += 512) zrow[k] = 0;
    __syncthreads();

    const int ns = n0 + q;
    const bool mine = q < NB && ns < a.N;
    float hprev = (mine && a.h0 != nullptr) ? a.h0[(long)ns * H + j] : 0.f;
    float gin0 = 0.f, gin1 = 0.f, gin2 = 0.f;
    if (mine) {
        const float *gp = a.gi + (long)ns * T * 3 * H;
        gin0 = gp[j]; gin1 = gp[H + j]; gin2 = gp[2 * H + j];
    }
    // A operand addresses (bytes inside one step's planes): lane -> row lane % 16, K group lane / 16
    const int arow = lane & 15, akg = lane >> 4;
    const unsigned aoff0 = (unsigned)(((0 * 8 + (arow & 7)) * LDK + 8 * akg) * 2);
    const unsigned aoff1 = (unsigned)(((1 * 8 + (arow & 7)) * LDK + 8 * akg) * 2);
    const unsigned aoff2 = (unsigned)(((2 * 8 + (arow & 7)) * LDK + 8 * akg) * 2);
    const bool upper = arow >= 8;
    // where this lane's finished sums are: column c = gate * U + u -> tile c / 16, column c % 16 of the tile
    int ptile[3], pcol[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { ptile[g] = (g * U + u) / 16; pcol[g] = (g * U + u) % 16; }
    bool dead = false;
#ifdef GRU_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const float gi0 = gin0, gi1 = gin1, gi2 = gin2;
        if (mine && t + 1 < T) {
            const float *gp = a.gi + ((long)ns * T + t + 1) * 3 * H;
            gin0 = gp[j]; gin1 = gp[H + j]; gin2 = gp[2 * H + j];
        }
        const char *planes = reinterpret_cast<const char *>(&hp[cur][0][0][0]);
        const char *pa1 = planes + (upper ? aoff1 : aoff0);                  // [h0 | h1]
        const char *pa2 = planes + (upper ? aoff2 : aoff0);                  // [h0 | h2]
        const char *pa3 = upper ? reinterpret_cast<const char *>(zrow) + 16 * akg : planes + aoff2;   // [h2 | 0]
        f32x4_t acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kb = (kpart * 4 + ks) * 64;                                       // bytes: 32 bf16 per K step
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t *>(pa1 + kb);
            const bf16x8_t a2 = *reinterpret_cast<const bf16x8_t *>(pa2 + kb);
            const bf16x8_t a3 = *reinterpret_cast<const bf16x8_t *>(pa3 + (upper ? 0 : kb));
#pragma unroll
            for (int i = 0; i < 3; ++i) {                                               // three independent accumulator chains
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8_t, bw[i][ks][0]), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8_t, bw[i][ks][1]), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8_t, bw[i][ks][2]), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, __builtin_bit_cast(bf16x8_t, bw[i][ks][0]), acc[i], 0, 0, 0);
            }
        }
        // lane: column lane % 16, rows 4 (lane / 16) .. + 3 of unit (tile, kpart)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            *reinterpret_cast<f32x4_t *>(&part[(tile0 + i) * M::KS + kpart][lane & 15][4 * (lane >> 4)]) = acc[i];
        GRU_STAMP(0);
        coop_lds_barrier();
        if (q < NB) {
            float gs[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float v = 0.f;
#pragma unroll
                for (int p = 0; p < M::KS; ++p) {
                    v += part[ptile[g] * M::KS + p][pcol[g]][q];
                    v += part[ptile[g] * M::KS + p][pcol[g]][q + 8];
                }
                gs[g] = v;
            }
            const float g0 = gs[0] + bh0, g1 = gs[1] + bh1, g2 = gs[2] + bh2;
            const float r = sigmoidf_(gi0 + g0);
            const float z = sigmoidf_(gi1 + g1);
            const float c = tanhf(gi2 + r * g2);
            float hv = (1.f - z) * c + z * hprev;
            if (dead) hv = NAN;
            hprev = hv;
            COOP_GLOBAL gu64_t *slot = (COOP_GLOBAL gu64_t *)(ca.comm + coop_fwd_slot<H, NB>(group, nxt, member, u, q));
            if (!(ca.fault && group == 0 && member == 0 && t == 1))
                __hip_atomic_store(slot, ((gu64_t)(ca.epoch0 + (unsigned)(t + 1)) << 32) | (gu64_t)__float_as_uint(mine ? hv : 0.f),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine) {
                const long row = (long)ns * T + t;
                float *gsv = a.gates + row * 3 * H;
                gsv[j] = r; gsv[H + j] = z; gsv[2 * H + j] = c;
                a.hn[row * H + j] = g2;
                a.out[row * H + j] = hv;
                a.hall[((long)ns * (T + 1) + t + 1) * H + j] = hv;
            }
        }
        GRU_STAMP(1);
        if (t + 1 < T) {
            COOP_GLOBAL gu64_t *slot[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i)
                slot[i] = (COOP_GLOBAL gu64_t *)(ca.comm + ((long)group * 2 + nxt) * (NB * H) + tid + 512 * i);
            gu64_t x[KP];
            unsigned spins = dead ? (1u << 22) : 0u;
            if (tid < NB * H) {
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int i = 0; i < KP; ++i) x[i] = __hip_atomic_load(slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < KP; ++i) ready = ready && (unsigned)(x[i] >> 32) == ca.epoch0 + (unsigned)(t + 1);
                    if (ready) break;
                    if (++spins > (1u << 22)) { dead = true; coop_report(ca.err, COOP_ERR_FWD_WAIT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#ifdef GRU_STAMPS
                ph[4] += spins;
#endif
                GRU_STAMP(2);
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    int gw, gk;
                    coop_fwd_who<H, NB>(tid + 512 * i, gw, gk);
                    unsigned short tt[3];
                    split3(dead ? NAN : __uint_as_float((unsigned)x[i]), tt);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) hp[nxt][pl][gw][gk] = tt[pl];
                }
            }
            coop_lds_barrier();
            GRU_STAMP(3);
        }
    }
#ifdef GRU_STAMPS
    if (ca.stamps != nullptr && (tid == 0 || tid == 511))
        for (int i = 0; i < 5; ++i) ca.stamps[(blockIdx.x * 2 + (tid != 0)) * 8 + i] = ph[i];
#endif
    if (a.hlast != nullptr && q < NB && n0 + q < a.N) a.hlast[(long)(n0 + q) * H + j] = hprev;
}

// Backward twin of gru_fwd_coop_kernel: member m keeps the SAME 3 U rows of W_hh (its U units x 3 gates) in
// registers, now one COLUMN j' per thread (thread (j', half): 96 rows), forms its partial W_hh^T dGH for all H
// columns and the members exchange the U-column pieces the others own.
//   comm: [groups][2][G (sender)][NB][H] granules, zeroed before the launch.
template <int H, int NB> __global__ __launch_bounds__(512) void gru_bwd_coop_kernel(GruCoopArgs ca)
{
    using C = CoopCfg<H>;
    constexpr int U = C::U, G = C::G, HALVES = C::HALVES;
    static_assert(3 * U / HALVES == 96, "96 weights per thread");
    static_assert(NB * U <= 512, "one elementwise thread per (window, unit)");
    __shared__ __attribute__((aligned(16))) float dgs[NB][3 * U];          // this member's dGH rows (gate, unit)
    __shared__ float part[HALVES][NB][H];
    const GruArgs &a = ca.g;
    const int T = a.T;
    int group, member;
    coop_who<G>(ca.groups, ca.xcd_map, group, member);
    const int tid = threadIdx.x;
    const int jc = tid & (H - 1), half = tid / H;             // column jc, rows half*96 .. +96 of the member's 3 U
    const int n0 = group * NB;

    f32x2 w[48];
#pragma unroll
    for (int i = 0; i < 96; ++i) {
        const int lr = half * 96 + i;                          // local row = gate*U + unit
        w[i / 2][i % 2] = a.whh[(long)((lr / U) * H + member * U + (lr % U)) * H + jc];
    }
    // elementwise role: thread (es, eu) for tid < NB*U
    const int es = tid / U, eu = tid - es * U;
    const int ej = member * U + eu;
    const int en = n0 + es;
    const bool ew = tid < NB * U;
    const bool emine = ew && en < a.N;
    float carry = 0.f;
    if (emine) {                                               // zero junk row T of dGH
        float *zr = a.dgh + ((long)en * (T + 1) + T) * 3 * H;
        zr[ej] = 0.f; zr[H + ej] = 0.f; zr[2 * H + ej] = 0.f;
    }
    // the saved activations of step t-1 are requested while step t runs (same reason as in the forward kernel)
    float p_dout = 0.f, p_r = 0.f, p_z = 0.f, p_c = 0.f, p_hn = 0.f, p_hp = 0.f;
    auto request = [&](int t) {
        const long row = (long)en * T + t;
        const float *gs = a.gates + row * 3 * H;
        p_dout = a.dout[row * H + ej];
        p_r = gs[ej]; p_z = gs[H + ej]; p_c = gs[2 * H + ej];
        p_hn = a.hn[row * H + ej];
        p_hp = a.hall[((long)en * (T + 1) + t) * H + ej];
    };
    if (emine) request(T - 1);
    coop_weights_ready(w);
    bool dead = false;
#ifdef GRU_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = T - 1; t >= 0; --t) {
        const int par = t & 1;
        const unsigned epoch = ca.epoch0 + (unsigned)(T - t);              // 1, 2, ...
        float keep = 0.f;
        if (ew) {
            float dpr = 0.f, dpz = 0.f, dhn = 0.f;
            if (emine) {
                const long row = (long)en * T + t;
                const float dh = p_dout + carry;
                const float r = p_r, z = p_z, c = p_c;
                const float hnv = p_hn;
                const float hp_ = p_hp;
                const float dc = dh * (1.f - z);
                const float dz = dh * (hp_ - c);
                const float dpn = dc * (1.f - c * c);
                dpr = dpn * hnv * r * (1.f - r);
                dpz = dz * z * (1.f - z);
                dhn = dpn * r;
                keep = dh * z;
                float *gi = a.dgi + row * 3 * H;
                gi[ej] = dpr; gi[H + ej] = dpz; gi[2 * H + ej] = dpn;
                float *gh = a.dgh + ((long)en * (T + 1) + t) * 3 * H;
                gh[ej] = dpr; gh[H + ej] = dpz; gh[2 * H + ej] = dhn;
            }
            dgs[es][eu] = dpr; dgs[es][U + eu] = dpz; dgs[es][2 * U + eu] = dhn;
            // behind the stores: a request in front of them made the compiler wait for it (vmcnt) where a store reused a register
            if (emine && t > 0) request(t - 1);
        }
        coop_lds_barrier();
        GRU_STAMP(0);
        // partial[jc] over this thread's 96 rows, all NB windows
        f32x2 acc[NB];                                         // even / odd rows
#pragma unroll
        for (int s = 0; s < NB; ++s) acc[s] = f32x2{0.f, 0.f};
#pragma unroll
        for (int i4 = 0; i4 < GRU_ABL_B; ++i4) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                const float4 d4 = *reinterpret_cast<const float4 *>(&dgs[s][half * 96 + 4 * i4]);
                acc[s] = pk_fma(w[2 * i4 + 1], pk_hi(d4), pk_fma(w[2 * i4], pk_lo(d4), acc[s]));
            }
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) part[half][s][jc] = acc[s].x + acc[s].y;
        coop_lds_barrier();
        GRU_STAMP(1);
        auto column = [&](int s, int k) {
            float v = part[0][s][k];
#pragma unroll
            for (int hh = 1; hh < HALVES; ++hh) v += part[hh][s][k];
            return v;
        };
        // publish the columns other members own (this member's own columns stay in LDS)
        if (t > 0) {
            for (int idx = tid; idx < NB * H; idx += 512) {
                const int s = idx / H, k = idx - s * H;
                if (k / U == member) continue;
                const float v = column(s, k);
                COOP_GLOBAL gu64_t *slot =
                    (COOP_GLOBAL gu64_t *)(ca.comm + ((((long)group * 2 + par) * G + member) * NB + s) * H + k);
                __hip_atomic_store(slot, ((gu64_t)epoch << 32) | (gu64_t)__float_as_uint(dead ? NAN : v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            GRU_STAMP(2);
            if (ew) {
                float sum = keep + column(es, ej);
                // the partners' pieces are polled together: one L2 round trip per attempt, not G - 1 in a row
                COOP_GLOBAL gu64_t *slot[G - 1];
#pragma unroll
                for (int d = 1; d < G; ++d) {
                    const int src = (member + d) & (G - 1);
                    slot[d - 1] = (COOP_GLOBAL gu64_t *)(ca.comm + ((((long)group * 2 + par) * G + src) * NB + es) * H + ej);
                }
                gu64_t x[G - 1];
                unsigned spins = dead ? (1u << 22) : 0u;
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int d = 0; d < G - 1; ++d) x[d] = __hip_atomic_load(slot[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int d = 0; d < G - 1; ++d) ready = ready && (unsigned)(x[d] >> 32) == epoch;
                    if (ready) break;
                    if (++spins > (1u << 22)) { dead = true; coop_report(ca.err, COOP_ERR_BWD_WAIT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int d = 0; d < G - 1; ++d) sum += __uint_as_float((unsigned)x[d]);
                carry = dead ? NAN : sum;
#ifdef GRU_STAMPS
                ph[4] += spins;
#endif
            }
            GRU_STAMP(3);
            // no barrier here: dgs is rewritten before the next step's first barrier, part after it, and every thread has
            // finished reading both when it gets there; `dead` stays with the thread that timed out (its carry is poisoned)
        }
    }
#ifdef GRU_STAMPS
    if (ca.stamps != nullptr && (tid == 0 || tid == 511))
        for (int i = 0; i < 5; ++i) ca.stamps[(blockIdx.x * 2 + (tid != 0)) * 8 + i] = ph[i];
#endif
}

template <int H, int NB> __global__ __launch_bounds__(512) void gru_bwd_mfma_kernel(GruCoopArgs ca)
{
    using C = CoopCfg<H>;
    constexpr int U = C::U, G = C::G;
    constexpr int K3 = 3 * U, LDK = K3 + 8;                                 // K = the member's 3U rows of W_hh, in steps of 32
    constexpr int TW = H / 16 / 8;                                          // column tiles per wave: 4 (H = 512), 2 (H = 256)
    static_assert(K3 % 32 == 0 && TW * 8 * 16 == H && TW * (K3 / 32) == 12, "twelve (tile, K step) operands per wave, all of K each");
    static_assert(NB <= 8 && NB * U <= 512, "one elementwise thread per (window, unit)");
    __shared__ __attribute__((aligned(16))) unsigned short dgp[3][8][LDK];  // the three bf16 terms of this member's dGH rows
    __shared__ __attribute__((aligned(16))) unsigned short zrow[LDK];
    __shared__ float part[2][8][H];                                          // [row half of the tile][window][column]
    const GruArgs &a = ca.g;
    const int T = a.T;
    int group, member;
    coop_who<G>(ca.groups, ca.xcd_map, group, member);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = group * NB;

    // B operands: B[k = local row (gate * U + unit)][n = column jc] = W_hh[gate * H + member * U + unit][jc]; wave w: column
    // tiles TW w .. TW w + TW - 1, every K step (no K split: the tile's sums are complete in its accumulators)
    MfmaFrag bw[TW][K3 / 32][3];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int jcol = (wave * TW + i) * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < K3 / 32; ++ks) {
            unsigned short t[8][3];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int lr = ks * 32 + 8 * (lane >> 4) + e;
                split3(a.whh[(long)((lr / U) * H + member * U + (lr % U)) * H + jcol], t[e]);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int d = 0; d < 4; ++d) bw[i][ks][pl].d[d] = (unsigned)t[2 * d][pl] | ((unsigned)t[2 * d + 1][pl] << 16);
        }
    }
    for (int idx = tid; idx < 3 * 8 * LDK; idx += 512) (&dgp[0][0][0])[idx] = 0;
    for (int kk = tid; kk < LDK; kk += 512) zrow[kk] = 0;
    __syncthreads();
    const int arow = lane & 15, akg = lane >> 4;
    const bool upper = arow >= 8;
    const char *const planes = reinterpret_cast<const char *>(&dgp[0][0][0]);
    const char *const pa1 = planes + (((upper ? 1 : 0) * 8 + (arow & 7)) * LDK + 8 * akg) * 2;      // [d0 | d1]
    const char *const pa2 = planes + (((upper ? 2 : 0) * 8 + (arow & 7)) * LDK + 8 * akg) * 2;      // [d0 | d2]
    const char *const pa3 = upper ? reinterpret_cast<const char *>(zrow) + 16 * akg : planes + ((2 * 8 + arow) * LDK + 8 * akg) * 2;   // [d2 | 0]
    // elementwise role: thread (es, eu) for tid < NB*U
    const int es = tid / U, eu = tid - es * U;
    const int ej = member * U + eu;
    const int en = n0 + es;
    const bool ew = tid < NB * U;
    const bool emine = ew && en < a.N;
    float carry = 0.f;
    if (emine) {                                               // zero junk row T of dGH
        float *zr = a.dgh + ((long)en * (T + 1) + T) * 3 * H;
        zr[ej] = 0.f; zr[H + ej] = 0.f; zr[2 * H + ej] = 0.f;
    }
    // the saved activations of step t-1 are requested while step t runs (same reason as in the forward kernel)
    float p_dout = 0.f, p_r = 0.f, p_z = 0.f, p_c = 0.f, p_hn = 0.f, p_hp = 0.f;
    auto request = [&](int t) {
        const long row = (long)en * T + t;
        const float *gs = a.gates + row * 3 * H;
        p_dout = a.dout[row * H + ej];
        p_r = gs[ej]; p_z = gs[H + ej]; p_c = gs[2 * H + ej];
        p_hn = a.hn[row * H + ej];
        p_hp = a.hall[((long)en * (T + 1) + t) * H + ej];
    };
    if (emine) request(T - 1);
    bool dead = false;
#ifdef GRU_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = T - 1; t >= 0; --t) {
        const int par = t & 1;
        const unsigned epoch = ca.epoch0 + (unsigned)(T - t);              // 1, 2, ...
        float keep = 0.f;
        if (ew) {
            float dpr = 0.f, dpz = 0.f, dhn = 0.f;
            if (emine) {
                const long row = (long)en * T + t;
                const float dh = p_dout + carry;
                const float r = p_r, z = p_z, c = p_c;
                const float hnv = p_hn;
                const float hp_ = p_hp;
                const float dc = dh * (1.f - z);
                const float dz = dh * (hp_ - c);
                const float dpn = dc * (1.f - c * c);
                dpr = dpn * hnv * r * (1.f - r);
                dpz = dz * z * (1.f - z);
                dhn = dpn * r;
                keep = dh * z;
                float *gi = a.dgi + row * 3 * H;
                gi[ej] = dpr; gi[H + ej] = dpz; gi[2 * H + ej] = dpn;
                float *gh = a.dgh + ((long)en * (T + 1) + t) * 3 * H;
                gh[ej] = dpr; gh[H + ej] = dpz; gh[2 * H + ej] = dhn;
            }
            {
                unsigned short t0[3], t1[3], t2[3];
                split3(dpr, t0); split3(dpz, t1); split3(dhn, t2);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) { dgp[pl][es][eu] = t0[pl]; dgp[pl][es][U + eu] = t1[pl]; dgp[pl][es][2 * U + eu] = t2[pl]; }
            }
            // behind the stores: a request in front of them made the compiler wait for it (vmcnt) where a store reused a register
            if (emine && t > 0) request(t - 1);
        }
        coop_lds_barrier();
        GRU_STAMP(0);
        // out[window][column] = sum over the member's rows: four products per K step (see gru_fwd_mfma_kernel), four column
        // tiles per wave sharing the A fragments
        {
            f32x4_t acc[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < K3 / 32; ++ks) {
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t *>(pa1 + ks * 64);
                const bf16x8_t a2 = *reinterpret_cast<const bf16x8_t *>(pa2 + ks * 64);
                const bf16x8_t a3 = *reinterpret_cast<const bf16x8_t *>(pa3 + (upper ? 0 : ks * 64));
#pragma unroll
                for (int i = 0; i < TW; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8_t, bw[i][ks][0]), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8_t, bw[i][ks][1]), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8_t, bw[i][ks][2]), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, __builtin_bit_cast(bf16x8_t, bw[i][ks][0]), acc[i], 0, 0, 0);
                }
            }
            // lane: column lane % 16 of the tile, rows 4 (lane / 16) .. + 3: rows 0-7 = the windows, rows 8-15 = the second half
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                const int col = (wave * TW + i) * 16 + (lane & 15);
#pragma unroll
                for (int e = 0; e < 4; ++e) part[lane >> 5][4 * ((lane >> 4) & 1) + e][col] = acc[i][e];
            }
        }
        coop_lds_barrier();
        GRU_STAMP(1);
        auto column = [&](int s, int k) { return part[0][s][k] + part[1][s][k]; };
        // publish the columns other members own (this member's own columns stay in LDS)
        if (t > 0) {
            for (int idx = tid; idx < NB * H; idx += 512) {
                const int s = idx / H, k = idx - s * H;
                if (k / U == member) continue;
                const float v = column(s, k);
                COOP_GLOBAL gu64_t *slot =
                    (COOP_GLOBAL gu64_t *)(ca.comm + ((((long)group * 2 + par) * G + member) * NB + s) * H + k);
                __hip_atomic_store(slot, ((gu64_t)epoch << 32) | (gu64_t)__float_as_uint(dead ? NAN : v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            GRU_STAMP(2);
            if (ew) {
                float sum = keep + column(es, ej);
                // the partners' pieces are polled together: one L2 round trip per attempt, not G - 1 in a row
                COOP_GLOBAL gu64_t *slot[G - 1];
#pragma unroll
                for (int d = 1; d < G; ++d) {
                    const int src = (member + d) & (G - 1);
                    slot[d - 1] = (COOP_GLOBAL gu64_t *)(ca.comm + ((((long)group * 2 + par) * G + src) * NB + es) * H + ej);
                }
                gu64_t x[G - 1];
                unsigned spins = dead ? (1u << 22) : 0u;
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int d = 0; d < G - 1; ++d) x[d] = __hip_atomic_load(slot[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int d = 0; d < G - 1; ++d) ready = ready && (unsigned)(x[d] >> 32) == epoch;
                    if (ready) break;
                    if (++spins > (1u << 22)) { dead = true; coop_report(ca.err, COOP_ERR_BWD_WAIT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int d = 0; d < G - 1; ++d) sum += __uint_as_float((unsigned)x[d]);
                carry = dead ? NAN : sum;
#ifdef GRU_STAMPS
                ph[4] += spins;
#endif
            }
            GRU_STAMP(3);
            // no barrier here: dgs is rewritten before the next step's first barrier, part after it, and every thread has
            // finished reading both when it gets there; `dead` stays with the thread that timed out (its carry is poisoned)
        }
    }
#ifdef GRU_STAMPS
    if (ca.stamps != nullptr && (tid == 0 || tid == 511))
        for (int i = 0; i < 5; ++i) ca.stamps[(blockIdx.x * 2 + (tid != 0)) * 8 + i] = ph[i];
#endif
}

#ifdef GRU_STAMPS
static int gru_print_stamps(const char *what, const unsigned long long *stamps, int nblk, int H, int nb, int T, hipStream_t st)
{
    static unsigned long long host[512 * 8];
    static int calls = 0;
    CPC_CHECK_HIP(hipStreamSynchronize(st));
    CPC_CHECK_HIP(hipMemcpy(host, stamps, sizeof(host), hipMemcpyDeviceToHost));
    if (++calls % 7 != 0) return CPC_OK;
    for (int who = 0; who < 2; ++who) {
        double sum[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < nblk; ++b)
            for (int i = 0; i < 5; ++i) {
                const double v = (double)host[(b * 2 + who) * 8 + i];
                sum[i] += v; mx[i] = std::max(mx[i], v);
            }
        fprintf(stderr, "gru stamps H=%d nb=%d thread %3d, us per step (mean, slowest workgroup) %s: %.2f (%.2f) | %.2f (%.2f) | %.2f (%.2f) | %.2f (%.2f); "
                "failed polls per step %.2f (%.2f)\n", H, nb, who ? 511 : 0, what, sum[0] / nblk / T * 0.01, mx[0] / T * 0.01, sum[1] / nblk / T * 0.01,
                mx[1] / T * 0.01, sum[2] / nblk / T * 0.01, mx[2] / T * 0.01, sum[3] / nblk / T * 0.01, mx[3] / T * 0.01, sum[4] / nblk / T, mx[4] / T);
    }
    return CPC_OK;
}
#endif

// The matrix-pipe form of the step.  Measured (profiles/r03_gru_stamps.txt): H = 512 with 8 windows per group -- forward 1.47 ->
// 1.23 ms, backward 2.03 -> 1.37 ms per CPC-large step; H = 256 with 2 windows per group (CPC-small: a 16-row tile is 3/4
// padding) -- forward 0.28 -> 0.38 ms (the K parts' partial tiles cost a barrier and LDS round trip more than the FMAs they
// replace), backward 0.30 -> 0.275 ms (no K split there).  CPC_GRU_NO_MFMA=1 / CPC_GRU_MFMA_ALL=1: never / always.
static bool mfma_wanted(int H, int nb, bool backward)
{
    static const bool off = getenv("CPC_GRU_NO_MFMA") != nullptr, all = getenv("CPC_GRU_MFMA_ALL") != nullptr;
    if (off) return false;
    return all || (H == 512 && nb == 8) || (backward && H == 256 && nb >= 2);
}
template <int H, int NB> static bool mfma_fits_fwd() { static const bool f = coop_fits(gru_fwd_mfma_kernel<H, NB>, 1, 1); return f; }
template <int H, int NB> static bool mfma_fits_bwd() { static const bool f = coop_fits(gru_bwd_mfma_kernel<H, NB>, 1, 1); return f; }

template <int H> static void launch_coop_fwd(int nb, dim3 grid, hipStream_t st, const GruCoopArgs &ca)
{
    coop_count_launch();
    const bool mm = mfma_wanted(H, nb, false);
    if (nb == 1) { if (mm && mfma_fits_fwd<H, 1>()) hipLaunchKernelGGL((gru_fwd_mfma_kernel<H, 1>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_fwd_coop_kernel<H, 1>), grid, dim3(512), 0, st, ca); }
    else if (nb == 2) { if (mm && mfma_fits_fwd<H, 2>()) hipLaunchKernelGGL((gru_fwd_mfma_kernel<H, 2>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_fwd_coop_kernel<H, 2>), grid, dim3(512), 0, st, ca); }
    else if (nb == 4) { if (mm && mfma_fits_fwd<H, 4>()) hipLaunchKernelGGL((gru_fwd_mfma_kernel<H, 4>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_fwd_coop_kernel<H, 4>), grid, dim3(512), 0, st, ca); }
    else { if (mm && mfma_fits_fwd<H, 8>()) hipLaunchKernelGGL((gru_fwd_mfma_kernel<H, 8>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_fwd_coop_kernel<H, 8>), grid, dim3(512), 0, st, ca); }
}
template <int H> static void launch_coop_bwd(int nb, dim3 grid, hipStream_t st, const GruCoopArgs &ca)
{
    coop_count_launch();
    const bool mm = mfma_wanted(H, nb, true);
    if (nb == 1) { if (mm && mfma_fits_bwd<H, 1>()) hipLaunchKernelGGL((gru_bwd_mfma_kernel<H, 1>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_bwd_coop_kernel<H, 1>), grid, dim3(512), 0, st, ca); }
    else if (nb == 2) { if (mm && mfma_fits_bwd<H, 2>()) hipLaunchKernelGGL((gru_bwd_mfma_kernel<H, 2>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_bwd_coop_kernel<H, 2>), grid, dim3(512), 0, st, ca); }
    else if (nb == 4) { if (mm && mfma_fits_bwd<H, 4>()) hipLaunchKernelGGL((gru_bwd_mfma_kernel<H, 4>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_bwd_coop_kernel<H, 4>), grid, dim3(512), 0, st, ca); }
    else { if (mm && mfma_fits_bwd<H, 8>()) hipLaunchKernelGGL((gru_bwd_mfma_kernel<H, 8>), grid, dim3(512), 0, st, ca); else hipLaunchKernelGGL((gru_bwd_coop_kernel<H, 8>), grid, dim3(512), 0, st, ca); }
}

// does the cooperative kernel fit a CU, and the grid the chip?  (cached per instance: the occupancy query is not free)
template <int H> static bool coop_fwd_fits(int nb, unsigned grid, int n_cus)
{
    static int ok[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};        // 0 unknown, 1 fits a CU, -1 does not
    if (ok[nb] == 0)
        ok[nb] = (nb == 1 ? coop_fits(gru_fwd_coop_kernel<H, 1>, 1, 1) : nb == 2 ? coop_fits(gru_fwd_coop_kernel<H, 2>, 1, 1)
                  : nb == 4 ? coop_fits(gru_fwd_coop_kernel<H, 4>, 1, 1) : coop_fits(gru_fwd_coop_kernel<H, 8>, 1, 1)) ? 1 : -1;
    return ok[nb] == 1 && (int)grid <= n_cus;
}
template <int H> static bool coop_bwd_fits(int nb, unsigned grid, int n_cus)
{
    static int ok[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ok[nb] == 0)
        ok[nb] = (nb == 1 ? coop_fits(gru_bwd_coop_kernel<H, 1>, 1, 1) : nb == 2 ? coop_fits(gru_bwd_coop_kernel<H, 2>, 1, 1)
                  : nb == 4 ? coop_fits(gru_bwd_coop_kernel<H, 4>, 1, 1) : coop_fits(gru_bwd_coop_kernel<H, 8>, 1, 1)) ? 1 : -1;
    return ok[nb] == 1 && (int)grid <= n_cus;
}

// ------------------------------------------------------------------------------------------------
struct GruLayout {
    int N, T, Din, H, layers;
    // saved, per layer
    float *gates[8], *hn[8], *hall[8], *outl[8];
    size_t saved_bytes;
    // scratch
    float *gi, *dgi, *dgh, *dxa, *dxb, *wt_l[8], *cs, *tn, *tn2;
    float *dgi_l[8], *dgh_l[8];            // layers 1..: gate gradients of their own (deferred tail: the side stream still reads them)
    size_t tn2_bytes;
    float4 *wpack;
    unsigned long long *comm;
    size_t comm_bytes;
    size_t tn_bytes, scratch_bytes;
};

static int gru_layout(GruLayout &g, int N, int T, int Din, int H, int layers, void *saved, void *scratch)
{
    CPC_REQUIRE(N > 0 && T > 0 && Din > 0, "gru: bad shape n=%d t=%d in=%d", N, T, Din);
    CPC_REQUIRE(H % 4 == 0 && H >= 4 && H <= 1024, "gru: hidden %d must be a multiple of 4 and <= 1024", H);
    CPC_REQUIRE(layers >= 1 && layers <= 8, "gru: 1..8 layers supported (got %d)", layers);
    g.N = N; g.T = T; g.Din = Din; g.H = H; g.layers = layers;
    Carver sv(saved);
    for (int l = 0; l < layers; ++l) {
        g.gates[l] = sv.take<float>((size_t)N * T * 3 * H);
        g.hn[l] = sv.take<float>((size_t)N * T * H);
        g.hall[l] = sv.take<float>((size_t)N * (T + 1) * H);
        g.outl[l] = (l + 1 < layers) ? sv.take<float>((size_t)N * T * H) : nullptr;
    }
    g.saved_bytes = sv.used();
    Carver sc(scratch);
    const int dmax = std::max(Din, H);
    g.gi = sc.take<float>((size_t)N * T * 3 * H);
    g.dgi = g.gi;                                     // forward's GI and backward's dGI never coexist
    g.dgh = sc.take<float>((size_t)N * (T + 1) * 3 * H);
    g.dxa = sc.take<float>((size_t)N * T * dmax);
    g.dxb = sc.take<float>((size_t)N * T * dmax);
    for (int l = 0; l < layers; ++l) g.wt_l[l] = sc.take<float>((size_t)3 * H * dmax);        // W_ih^T of every layer (backward)
    g.wpack = sc.take<float4>((size_t)3 * H * H / 4);
    g.cs = sc.take<float>(colsum_rows_scratch_bytes(3 * H) / sizeof(float));
    // granules of the cooperative kernels: backward [groups][2][G][NB][H], groups*NB < N + 8 windows, G <= 16
    g.comm_bytes = coop_comm_bytes(H, N);
    g.comm = sc.take<unsigned long long>(g.comm_bytes / sizeof(unsigned long long));
    g.tn_bytes = std::max(gemm_tn_scratch_bytes(3 * H, H, (long)N * (T + 1)), gemm_tn_scratch_bytes(3 * H, dmax, (long)N * T));
    g.tn_bytes = std::max(g.tn_bytes, gemm_tn_scratch_bytes(3 * H, Din, (long)N * T));
    // the same room serves an ordered K split of the projections (GI = X W_ih^T, dX = dGI W_ih) when they have few tiles
    g.tn_bytes = std::max(g.tn_bytes, std::max(gemm_nt_scratch_bytes((long)N * T, 3 * H, dmax), gemm_nt_scratch_bytes((long)N * T, dmax, 3 * H)));
    g.tn = sc.take<float>(g.tn_bytes / sizeof(float));
    // (the input-gradient product's K split when the weight-gradient products run beside it on the side stream: gru_backward, defer_tail)
    g.tn2_bytes = gemm_nt_scratch_bytes((long)N * T, dmax, 3 * H);
    g.tn2 = sc.take<float>(g.tn2_bytes / sizeof(float));
    g.dgi_l[0] = g.dgi; g.dgh_l[0] = g.dgh;
    for (int l = 1; l < layers; ++l) {
        g.dgi_l[l] = sc.take<float>((size_t)N * T * 3 * H);
        g.dgh_l[l] = sc.take<float>((size_t)N * (T + 1) * 3 * H);
    }
    g.scratch_bytes = sc.used();
    return CPC_OK;
}

static int gru_forward(const float *x, const float *const *prm, const float *h0, float *out, float *h_last, void *saved,
                       void *scratch, int N, int T, int Din, int H, int layers, hipStream_t st)
{
    GruLayout g;
    CPC_TRY(gru_layout(g, N, T, Din, H, layers, saved, scratch));
    const int hp = std::max(64, (int)cdiv(H, 64) * 64);
    const int kq = std::max(1, std::min(1024 / hp, H / 4));
    const float *xin = x;
    int din = Din;
    for (int l = 0; l < layers; ++l) {
        const float *w_ih = prm[4 * l], *w_hh = prm[4 * l + 1], *b_ih = prm[4 * l + 2], *b_hh = prm[4 * l + 3];
        RowMap none{};
        none.splitk_scratch = g.tn; none.splitk_bytes = g.tn_bytes;
        CPC_TRY(gemm_nt(xin, din, w_ih, din, g.gi, 3L * H, b_ih, (long)N * T, 3 * H, din, none, st));
        GruArgs a{};
        a.gi = g.gi; a.wpack = g.wpack; a.bhh = b_hh;
        a.h0 = h0 ? h0 + (size_t)l * N * H : nullptr;
        a.out = (l + 1 < layers) ? g.outl[l] : out;
        a.hall = g.hall[l]; a.gates = g.gates[l]; a.hn = g.hn[l];
        a.hlast = h_last ? h_last + (size_t)l * N * H : nullptr;
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq; a.whh = w_hh;
        static const bool coop_off = getenv("CPC_GRU_STREAM") != nullptr;
        static const int n_cus = coop_cu_count();
        // the cooperative kernel needs every workgroup resident at once (1 per CU)
        int G = 0;
        int nb = (coop_off || !coop_allowed()) ? 0 : coop_windows_per_group(H, N, n_cus, &G);
        if (nb != 0 && !(H == 256 ? coop_fwd_fits<256>(nb, (unsigned)(cdiv(N, nb) * G), n_cus) : coop_fwd_fits<512>(nb, (unsigned)(cdiv(N, nb) * G), n_cus)))
            nb = 0;                             // not resident all at once: the streaming kernel has no such requirement
        if (nb != 0) {
            GruCoopArgs ca{};
            ca.g = a; ca.groups = (int)cdiv(N, nb);
            ca.xcd_map = (ca.groups % 8 == 0) ? 1 : 0;
            ca.err = coop_error_word(); ca.fault = coop_fault_injection();
            CPC_TRY(coop_comm_acquire(sizeof(gu64_t) * (size_t)ca.groups * 2 * nb * H, T, st, &ca.comm, &ca.epoch0));
#ifdef GRU_STAMPS
            static unsigned long long *stamps = nullptr;
            if (stamps == nullptr) CPC_CHECK_HIP(hipMalloc(&stamps, 512 * 8 * sizeof(unsigned long long)));
            ca.stamps = stamps;
#endif
            {
                ProfScope prof(PROF_GRU_FWD, st);
                const dim3 grid((unsigned)(ca.groups * G));
                if (H == 256) launch_coop_fwd<256>(nb, grid, st, ca);
                else launch_coop_fwd<512>(nb, grid, st, ca);
            }
#ifdef GRU_STAMPS
            CPC_TRY(gru_print_stamps("fwd: math | (barrier+) gates+publish | wait | lds+barrier", stamps, ca.groups * G, H, nb, T, st));
#endif
        } else {
            hipLaunchKernelGGL(gru_pack_fwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H);      // (the streaming kernel's weight layout only)
            CPC_CHECK_LAUNCH("gru_pack_fwd_kernel");
            ProfScope prof(PROF_GRU_FWD, st);
            const size_t lds = sizeof(float) * (cdiv(H, 4) * 4 + (size_t)kq * 3 * hp);
            hipLaunchKernelGGL(gru_fwd_kernel, dim3((unsigned)N), dim3(kq * hp), lds, st, a);
        }
        CPC_CHECK_LAUNCH("gru_fwd_kernel");
        xin = a.out;
        din = H;
    }
    return CPC_OK;
}

// defer_tail: the weight gradients of every layer (nothing on `st` needs them before the optimiser) are produced on the library's
// side stream: layer l's beside the recurrent kernel of layer l - 1 (latency-bound: the matrix pipe is idle), layer 0's beside what
// the caller enqueues next (the encoder's backward: its normalisation / reduction kernels leave the matrix pipe idle for ~0.3 ms per
// step).  Each layer keeps its gate gradients in a buffer of its own for that.  cpc_side_tail_join makes a stream wait for them
static int gru_backward(const float *x, const float *const *prm, const float *dout, void *saved, void *scratch, float *dx,
                        float *const *grads, int N, int T, int Din, int H, int layers, hipStream_t st, bool defer_tail = false)
{
    GruLayout g;
    CPC_TRY(gru_layout(g, N, T, Din, H, layers, saved, scratch));
    const int hp = std::max(64, (int)cdiv(H, 64) * 64);
    const int kq = std::max(1, std::min(1024 / hp, H / 4));
    const float *dcur = dout;
    // W_ih^T of every layer that has an input gradient, in front of the first recurrent kernel: the transposes depend on the weights
    // only, and a small kernel queued BEHIND a recurrent kernel starts while the deferred criterion sum / the weight-gradient
    // products hold the chip on the side stream -- seen at 212 us (3 MB) on the critical path of CPC-large, 5 us alone
    for (int l = layers - 1; l >= 0; --l)
        if (l > 0 || dx != nullptr) CPC_TRY(transpose2d(prm[4 * l], g.wt_l[l], 3 * H, (l == 0) ? Din : H, st));
    for (int l = layers - 1; l >= 0; --l) {
        const float *w_hh = prm[4 * l + 1];
        const float *xin = (l == 0) ? x : g.outl[l - 1];
        const int din = (l == 0) ? Din : H;
        GruArgs a{};
        a.wpack = g.wpack; a.hall = g.hall[l]; a.gates = g.gates[l]; a.hn = g.hn[l];
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq; a.whh = w_hh;
        // (deferred tail: every layer's gate gradients stay where they are until the side stream has used them)
        float *const dgi = defer_tail ? g.dgi_l[l] : g.dgi, *const dgh = defer_tail ? g.dgh_l[l] : g.dgh;
        a.dout = dcur; a.dgi = dgi; a.dgh = dgh;
        CPC_TRY(infonce_deferred_mark(st));       // (see infonce_deferred_start below)
        static const bool coop_off = getenv("CPC_GRU_STREAM") != nullptr;
        static const int n_cus = coop_cu_count();
        int G = 0;
        int nb = (coop_off || !coop_allowed()) ? 0 : coop_windows_per_group(H, N, n_cus, &G);
        if (nb != 0 && !(H == 256 ? coop_bwd_fits<256>(nb, (unsigned)(cdiv(N, nb) * G), n_cus) : coop_bwd_fits<512>(nb, (unsigned)(cdiv(N, nb) * G), n_cus)))
            nb = 0;
        if (nb != 0) {
            GruCoopArgs ca{};
            ca.g = a; ca.groups = (int)cdiv(N, nb);
            ca.xcd_map = (ca.groups % 8 == 0) ? 1 : 0;
            ca.err = coop_error_word(); ca.fault = coop_fault_injection();
            CPC_TRY(coop_comm_acquire(sizeof(gu64_t) * (size_t)ca.groups * 2 * G * nb * H, T, st, &ca.comm, &ca.epoch0));
#ifdef GRU_STAMPS
            static unsigned long long *stamps = nullptr;
            if (stamps == nullptr) CPC_CHECK_HIP(hipMalloc(&stamps, 512 * 8 * sizeof(unsigned long long)));
            ca.stamps = stamps;
#endif
            {
                ProfScope prof(PROF_GRU_BWD, st);
                const dim3 grid((unsigned)(ca.groups * G));
                if (H == 256) launch_coop_bwd<256>(nb, grid, st, ca);
                else launch_coop_bwd<512>(nb, grid, st, ca);
            }
#ifdef GRU_STAMPS
            CPC_TRY(gru_print_stamps("bwd: gates | math | publish | wait", stamps, ca.groups * G, H, nb, T, st));
#endif
        } else {
            hipLaunchKernelGGL(gru_pack_bwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H);
            CPC_CHECK_LAUNCH("gru_pack_bwd_kernel");
            ProfScope prof(PROF_GRU_BWD, st);
            const size_t lds = sizeof(float) * ((size_t)3 * H + (size_t)kq * hp);
            hipLaunchKernelGGL(gru_bwd_kernel, dim3((unsigned)N), dim3(kq * hp), lds, st, a);
        }
        CPC_CHECK_LAUNCH("gru_bwd_kernel");
        CPC_TRY(infonce_deferred_start(st));      // (no-op unless a deferred criterion backward is waiting to run beside this)

        hipStream_t wst = st;
        const bool tail = defer_tail;
        if (tail) CPC_TRY(side_tail_begin(st, &wst));
        // dW_hh[g][k] = sum_{n,t} dGH[n,t][g] * h_{t-1}[n][k]   (hall row t is h_{t-1}; row T of dGH is zero)
        CPC_TRY(gemm_tn(dgh, 3L * H, g.hall[l], H, grads[4 * l + 1], H, 3 * H, H, (long)N * (T + 1), g.tn, g.tn_bytes, 0, 0, wst));
        CPC_TRY(colsum_rows(dgh, 3L * H, (long)N * (T + 1), 3 * H, grads[4 * l + 3], g.cs, wst));
        // dW_ih[g][k] = sum dGI[n,t][g] * x[n,t][k]
        CPC_TRY(gemm_tn(dgi, 3L * H, xin, din, grads[4 * l], din, 3 * H, din, (long)N * T, g.tn, g.tn_bytes, 0, 0, wst));
        CPC_TRY(colsum_rows(dgi, 3L * H, (long)N * T, 3 * H, grads[4 * l + 2], g.cs, wst));
        if (tail) CPC_TRY(side_tail_end());
        // dX = dGI . W_ih
        float *dxl = (l == 0) ? dx : ((l % 2) ? g.dxa : g.dxb);
        if (dxl != nullptr) {
            RowMap none{};
            if (tail) { none.splitk_scratch = g.tn2; none.splitk_bytes = g.tn2_bytes; }        // (g.tn is the side stream's now)
            else { none.splitk_scratch = g.tn; none.splitk_bytes = g.tn_bytes; }
            CPC_TRY(gemm_nt(dgi, 3L * H, g.wt_l[l], 3L * H, dxl, din, nullptr, (long)N * T, din, 3 * H, none, st));
        }
        dcur = dxl;
    }
    return CPC_OK;
}

}  // namespace cpc

extern "C" size_t cpc_gru_saved_bytes(int n, int t, int dim_in, int hidden, int layers)
{
    cpc::GruLayout g;
    if (cpc::gru_layout(g, n, t, dim_in, hidden, layers, nullptr, nullptr) != CPC_OK) return 0;
    return g.saved_bytes;
}

extern "C" size_t cpc_gru_scratch_bytes(int n, int t, int dim_in, int hidden, int layers)
{
    cpc::GruLayout g;
    if (cpc::gru_layout(g, n, t, dim_in, hidden, layers, nullptr, nullptr) != CPC_OK) return 0;
    return g.scratch_bytes;
}

extern "C" int cpc_gru_forward(const float *x, const float *const *params, const float *h0, float *out, float *h_last,
                               void *saved, void *scratch, int n, int t, int dim_in, int hidden, int layers, cpc_stream_t stream)
{
    CPC_TRY(cpc::coop_error_take("cpc_gru_forward"));      // a time-out of an earlier cooperative launch surfaces here
    return cpc::gru_forward(x, params, h0, out, h_last, saved, scratch, n, t, dim_in, hidden, layers, static_cast<hipStream_t>(stream));
}

extern "C" int cpc_gru_backward(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                cpc_stream_t stream)
{
    cpc::coop_count_backward_call();
    CPC_TRY(cpc::coop_error_take("cpc_gru_backward"));      // a time-out of an earlier cooperative launch surfaces here
    return cpc::gru_backward(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                             static_cast<hipStream_t>(stream));
}

extern "C" int cpc_gru_backward_deferred(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                         float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                         cpc_stream_t stream)
{
    cpc::coop_count_backward_call();
    CPC_TRY(cpc::coop_error_take("cpc_gru_backward_deferred"));
    return cpc::gru_backward(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                             static_cast<hipStream_t>(stream), true);
}

extern "C" int cpc_side_tail_join(cpc_stream_t stream) { return cpc::side_tail_join(static_cast<hipStream_t>(stream)); }
extern "C" int cpc_side_tail_wait(cpc_stream_t stream) { return cpc::side_tail_wait(static_cast<hipStream_t>(stream)); }
