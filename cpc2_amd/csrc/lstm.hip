// CPCAR (mode="LSTM", the reference's default arMode) on gfx950.  Reference: /root/reference/cpc/model.py:158-207
// (:180-183) -> torch.nn.LSTM (batch_first, gate order i, f, g, o):
//     i = sigmoid(W_ii x + b_ii + W_hi h + b_hi)      f = sigmoid(W_if x + b_if + W_hf h + b_hf)
//     g = tanh   (W_ig x + b_ig + W_hg h + b_hg)      o = sigmoid(W_io x + b_io + W_ho h + b_ho)
//     c' = f * c + i * g                               h' = o * tanh(c')
//
// mode="RNN" (model.py:174-176 -> torch.nn.RNN, tanh):  h' = tanh(W_ih x + b_ih + W_hh h + b_hh) shares every kernel
// here with G = 1 gate block instead of 4.
//
// Same shape as the streaming GRU path (gru.hip): per layer one GEMM for all input projections
// GI = X W_ih^T + b_ih, then ONE persistent kernel for the T sequential steps -- a workgroup owns a window for the
// whole sequence, thread (j, q) owns hidden unit j and K slice q, h lives in LDS, W_hh (re-laid out so that lanes
// read consecutive float4s) is streamed from L2 every step.  Backward mirrors it (BPTT), then GEMMs give dW_hh,
// dW_ih and dX.  Both bias gradients are column sums of the same pre-activation gradient.
#include "common.h"
#include "coop.h"

#include <algorithm>
#include <cstdlib>

namespace cpc {

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// W_hh [G*H][H] -> wf[(k4*G + g)*H + j] = W[g*H + j][4*k4 .. 4*k4+3]   (forward: thread j, all k)
__global__ void lstm_pack_fwd_kernel(const float *w, float4 *wf, int H, int G)
{
    const int total = G * H * (H / 4);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx % H;
        const int g = (idx / H) % G;
        const int k4 = idx / (G * H);
        const float *src = w + (long)(g * H + j) * H + 4 * k4;
        wf[idx] = make_float4(src[0], src[1], src[2], src[3]);
    }
}

// W_hh [G*H][H] -> wb[g4*H + j] = (W[4*g4][j], .., W[4*g4+3][j])   (backward: thread j = column)
__global__ void lstm_pack_bwd_kernel(const float *w, float4 *wb, int H, int G)
{
    const int total = G * H / 4 * H;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx % H;
        const int g4 = idx / H;
        wb[idx] = make_float4(w[(long)(4 * g4) * H + j], w[(long)(4 * g4 + 1) * H + j], w[(long)(4 * g4 + 2) * H + j],
                              w[(long)(4 * g4 + 3) * H + j]);
    }
}

struct LstmArgs {
    const float *gi;      // [N*T][G*H]  input projections incl. b_ih
    const float4 *wpack;  // packed W_hh (streaming kernels)
    const float *whh;     // [G*H][H] as stored (cooperative kernels)
    const float *bhh;     // [G*H]
    const float *h0, *c0; // [N][H] or null
    float *out;           // [N][T][H]
    float *hall;          // [N][T+1][H]  row 0 = h0, row t+1 = h_t
    float *call;          // [N][T+1][H]  row 0 = c0, row t+1 = c_t            (LSTM only)
    float *gates;         // [N*T][4H]    i, f, g, o after the non-linearity    (LSTM only)
    float *hlast, *clast; // [N][H] or null
    int N, T, H;
    int hp, kq;           // threads = kq * hp: hp = H rounded up to 64, kq = K-split factor
    // backward
    const float *dout;    // [N][T][H]
    float *dgi;           // [N*T][G*H]
    float *dgh;           // [N][T+1][G*H], row T zero (same values as dgi, laid out for the W_hh gradient)
};

// G = 4: LSTM, G = 1: tanh RNN
template <int G>
__global__ void lstm_fwd_kernel(LstmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hs[H] | red[kq][G][hp]
    const int H = a.H, T = a.T, hp = a.hp, kq = a.kq;
    float *hs = smem;
    float *red = smem + ((H + 3) / 4) * 4;
    const int j = threadIdx.x % hp, q = threadIdx.x / hp;
    const bool act = j < H;
    const int n = blockIdx.x;
    const int k4_per = (H / 4 + kq - 1) / kq;
    const int k4_lo = q * k4_per, k4_hi = min(H / 4, k4_lo + k4_per);

    float hprev = 0.f, cprev = 0.f;
    float bh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bh[g] = 0.f;
    if (act && q == 0) {
        hprev = a.h0 != nullptr ? a.h0[(long)n * H + j] : 0.f;
        hs[j] = hprev;
        a.hall[((long)n * (T + 1)) * H + j] = hprev;
        if (G == 4) {
            cprev = a.c0 != nullptr ? a.c0[(long)n * H + j] : 0.f;
            a.call[((long)n * (T + 1)) * H + j] = cprev;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) bh[g] = a.bhh[g * H + j];
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        float acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = 0.f;
        if (act) {
            const float4 *wp = a.wpack + j;
#pragma unroll 4
            for (int k4 = k4_lo; k4 < k4_hi; ++k4) {
                const float4 h4 = reinterpret_cast<const float4 *>(hs)[k4];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float4 w4 = wp[(long)(k4 * G + g) * H];
                    acc[g] = fmaf(w4.x, h4.x, fmaf(w4.y, h4.y, fmaf(w4.z, h4.z, fmaf(w4.w, h4.w, acc[g]))));
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) red[(q * G + g) * hp + j] = acc[g];
        }
        __syncthreads();                       // partial sums visible; nobody reads hs any more
        if (act && q == 0) {
            float pre[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                pre[g] = bh[g];
                for (int qq = 0; qq < kq; ++qq) pre[g] += red[(qq * G + g) * hp + j];
            }
            const long row = (long)n * T + t;
            const float *gin = a.gi + row * G * H;
            float hv;
            if (G == 4) {
                const float ig = sigm(gin[j] + pre[0]);
                const float fg = sigm(gin[H + j] + pre[G > 1 ? 1 : 0]);
                const float gg = tanhf(gin[2 * H + j] + pre[G > 2 ? 2 : 0]);
                const float og = sigm(gin[3 * H + j] + pre[G > 3 ? 3 : 0]);
                const float cv = fg * cprev + ig * gg;
                hv = og * tanhf(cv);
                float *gs = a.gates + row * 4 * H;
                gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
                a.call[((long)n * (T + 1) + t + 1) * H + j] = cv;
                cprev = cv;
            } else {
                hv = tanhf(gin[j] + pre[0]);
            }
            a.out[row * H + j] = hv;
            a.hall[((long)n * (T + 1) + t + 1) * H + j] = hv;
            hs[j] = hv;
            hprev = hv;
        }
        __syncthreads();
    }
    if (act && q == 0) {
        if (a.hlast != nullptr) a.hlast[(long)n * H + j] = hprev;
        if (G == 4 && a.clast != nullptr) a.clast[(long)n * H + j] = cprev;
    }
}

template <int G>
__global__ void lstm_bwd_kernel(LstmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // dg[G*H] | red[kq][hp]
    const int H = a.H, T = a.T, hp = a.hp, kq = a.kq;
    float *dg = smem;
    float *red = smem + G * H;
    const int j = threadIdx.x % hp, q = threadIdx.x / hp;
    const bool act = j < H;
    const int n = blockIdx.x;
    const int g4_total = G * H / 4;
    const int g4_per = (g4_total + kq - 1) / kq;
    const int g4_lo = q * g4_per, g4_hi = min(g4_total, g4_lo + g4_per);

    float carry_h = 0.f, carry_c = 0.f;
    if (act && q == 0) {                                         // zero junk row T of dGH
        float *zr = a.dgh + ((long)n * (T + 1) + T) * G * H;
#pragma unroll
        for (int g = 0; g < G; ++g) zr[g * H + j] = 0.f;
    }
    for (int t = T - 1; t >= 0; --t) {
        if (act && q == 0) {
            const long row = (long)n * T + t;
            const float dh = a.dout[row * H + j] + carry_h;
            float dp[G];
            if (G == 4) {
                const float *gs = a.gates + row * 4 * H;
                const float ig = gs[j], fg = gs[H + j], gg = gs[2 * H + j], og = gs[3 * H + j];
                const float cv = a.call[((long)n * (T + 1) + t + 1) * H + j];
                const float cp = a.call[((long)n * (T + 1) + t) * H + j];
                const float tc = tanhf(cv);
                const float dcv = dh * og * (1.f - tc * tc) + carry_c;
                dp[0] = dcv * gg * ig * (1.f - ig);
                dp[G > 1 ? 1 : 0] = dcv * cp * fg * (1.f - fg);
                dp[G > 2 ? 2 : 0] = dcv * ig * (1.f - gg * gg);
                dp[G > 3 ? 3 : 0] = dh * tc * og * (1.f - og);
                carry_c = dcv * fg;
            } else {
                const float hv = a.hall[((long)n * (T + 1) + t + 1) * H + j];
                dp[0] = dh * (1.f - hv * hv);
            }
            float *gi = a.dgi + row * G * H;
            float *gh = a.dgh + ((long)n * (T + 1) + t) * G * H;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                gi[g * H + j] = dp[g];
                gh[g * H + j] = dp[g];
                dg[g * H + j] = dp[g];
            }
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
            float sum = 0.f;
            for (int qq = 0; qq < kq; ++qq) sum += red[qq * hp + j];
            carry_h = sum;                       // dh_{t-1} = W_hh^T dG_t
        }
        // dg is rewritten only after every thread passed the barrier above; red is read again only after the next
        // two barriers
    }
}

// ------------------------------------------------------------------------------------------------
// On-chip LSTM recurrence for H = 256 and 512 (same scheme as gru.hip's cooperative kernels, see coop.h): a group of
// G workgroups shares NB windows and W_hh never leaves the register file -- 4 gates x 32 columns = 128 weights per
// thread.  The cell state of (window q, unit j) stays in the register of the lane that finishes that pair.
struct LstmCoopArgs {
    LstmArgs g;
    gu64_t *comm;          // granules (coop_comm_acquire: the library's own buffer): fwd [groups][2][G][U][NB] (coop_fwd_slot), bwd [groups][2][G][NB][H]
    unsigned epoch0;       // this launch's epochs are epoch0 + 1 .. epoch0 + T: no granule of an earlier launch carries one of them
    int groups, xcd_map;
    int *err;              // host-visible error word (coop.h), or nullptr
    int fault;             // tests: member 0 of group 0 withholds its publish of step 1
};

template <int H, int NB> __global__ __launch_bounds__(512) void lstm_fwd_coop_kernel(LstmCoopArgs ca)
{
    using C = CoopCfg<H>;
    constexpr int QS = C::QS, U = C::U, G = C::G;
    constexpr int KP = NB * H / 512 > 0 ? NB * H / 512 : 1;          // granules gathered per thread and step
    static_assert(NB <= QS, "one finishing lane per window");
    __shared__ __attribute__((aligned(16))) float hs[2][NB][C::LDH];
    const LstmArgs &a = ca.g;
    const int T = a.T;
    int group, member;
    coop_who<G>(ca.groups, ca.xcd_map, group, member);
    const int tid = threadIdx.x;
    const int q = tid & (QS - 1), u = tid / QS;
    const int j = member * U + u;
    const int n0 = group * NB;

    f32x2 w[4][16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i4 = 0; i4 < 8; ++i4) {
            const float4 v = *reinterpret_cast<const float4 *>(a.whh + (long)(g * H + j) * H + q * 32 + 4 * i4);
            w[g][2 * i4] = pk_lo(v); w[g][2 * i4 + 1] = pk_hi(v);
        }
    const float bh0 = a.bhh[j], bh1 = a.bhh[H + j], bh2 = a.bhh[2 * H + j], bh3 = a.bhh[3 * H + j];

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
    float cprev = 0.f;
    float gin0 = 0.f, gin1 = 0.f, gin2 = 0.f, gin3 = 0.f;
    if (mine) {
        cprev = a.c0 != nullptr ? a.c0[(long)ns * H + j] : 0.f;
        a.call[((long)ns * (T + 1)) * H + j] = cprev;
        const float *gp = a.gi + (long)ns * T * 4 * H;
        gin0 = gp[j]; gin1 = gp[H + j]; gin2 = gp[2 * H + j]; gin3 = gp[3 * H + j];
    }
    bool dead = false;
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // the input projections of step t were requested one step ago; the next step's are requested now, so that no
        // L2/HBM round trip sits on the serial path of a time step
        const float gi0 = gin0, gi1 = gin1, gi2 = gin2, gi3 = gin3;
        if (mine && t + 1 < T) {
            const float *gp = a.gi + ((long)ns * T + t + 1) * 4 * H;
            gin0 = gp[j]; gin1 = gp[H + j]; gin2 = gp[2 * H + j]; gin3 = gp[3 * H + j];
        }
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;          // the pre-activations of window q, for its finishing lane
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            f32x2 a2[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // even / odd columns
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                const float4 h4 = *reinterpret_cast<const float4 *>(&hs[cur][s][q * 36 + 4 * i4]);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    a2[g] = pk_fma(w[g][2 * i4 + 1], pk_hi(h4), pk_fma(w[g][2 * i4], pk_lo(h4), a2[g]));
            }
            float acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = coop_group_sum<QS>(a2[g].x + a2[g].y);
            if (s == q) { g0 = acc[0]; g1 = acc[1]; g2 = acc[2]; g3 = acc[3]; }
        }
        if (q < NB) {
            const float ig = sigm(gi0 + g0 + bh0);
            const float fg = sigm(gi1 + g1 + bh1);
            const float gg = tanhf(gi2 + g2 + bh2);
            const float og = sigm(gi3 + g3 + bh3);
            const float cv = fg * cprev + ig * gg;
            float hv = og * tanhf(cv);
            cprev = cv;
            if (dead) hv = NAN;
            // publish first (also for padding windows, so that every granule of the epoch gets written): the other
            // members wait for this store, nobody waits for the saved activations below
            COOP_GLOBAL gu64_t *slot = (COOP_GLOBAL gu64_t *)(ca.comm + coop_fwd_slot<H, NB>(group, nxt, member, u, q));
            if (!(ca.fault && group == 0 && member == 0 && t == 1))
                __hip_atomic_store(slot, ((gu64_t)(ca.epoch0 + (unsigned)(t + 1)) << 32) | (gu64_t)__float_as_uint(mine ? hv : 0.f),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine) {
                const long row = (long)ns * T + t;
                float *gs = a.gates + row * 4 * H;
                gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
                a.out[row * H + j] = hv;
                a.hall[((long)ns * (T + 1) + t + 1) * H + j] = hv;
                a.call[((long)ns * (T + 1) + t + 1) * H + j] = cv;
            }
        }
        // gather the whole new h (all members) into the other LDS buffer; a thread's KP granules are polled together
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
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int idx = tid + 512 * i;
                    int gw, gk;
                    coop_fwd_who<H, NB>(idx, gw, gk);
                    hs[nxt][gw][coop_pad(gk)] = dead ? NAN : __uint_as_float((unsigned)x[i]);
                }
            }
            coop_lds_barrier();            // `dead` stays with the thread that timed out: what it gathered is poisoned above
        }
    }
    if (mine) {
        // hall row T was written by this same lane
        if (a.hlast != nullptr) a.hlast[(long)ns * H + j] = a.hall[((long)ns * (T + 1) + T) * H + j];
        if (a.clast != nullptr) a.clast[(long)ns * H + j] = cprev;
    }
}

// Backward twin: member m keeps the SAME 4 U rows of W_hh (its U units x 4 gates) in registers, one COLUMN j' per
// thread (thread (j', half): 128 rows), forms its partial W_hh^T dG for all H columns and the members exchange the
// U-column pieces the others own.   comm: [groups][2][G (sender)][NB][H] granules, zeroed before the launch.
template <int H, int NB> __global__ __launch_bounds__(512) void lstm_bwd_coop_kernel(LstmCoopArgs ca)
{
    using C = CoopCfg<H>;
    constexpr int U = C::U, G = C::G, HALVES = C::HALVES;
    constexpr int RW = 4 * U / HALVES;                                      // rows per thread
    static_assert(RW == 128, "128 weights per thread");
    static_assert(NB * U <= 512, "one elementwise thread per (window, unit)");
    __shared__ __attribute__((aligned(16))) float dgs[NB][4 * U];          // this member's dG rows (gate, unit)
    __shared__ float part[HALVES][NB][H];
    const LstmArgs &a = ca.g;
    const int T = a.T;
    int group, member;
    coop_who<G>(ca.groups, ca.xcd_map, group, member);
    const int tid = threadIdx.x;
    const int jc = tid & (H - 1), half = tid / H;             // column jc, rows half*RW .. +RW of the member's 4 U
    const int n0 = group * NB;

    f32x2 w[RW / 2];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int lr = half * RW + i;                          // local row = gate*U + unit
        w[i / 2][i % 2] = a.whh[(long)((lr / U) * H + member * U + (lr % U)) * H + jc];
    }
    // elementwise role: thread (es, eu) for tid < NB*U
    const int es = tid / U, eu = tid - es * U;
    const int ej = member * U + eu;
    const int en = n0 + es;
    const bool ew = tid < NB * U;
    const bool emine = ew && en < a.N;
    float carry = 0.f, carry_c = 0.f;
    if (emine) {                                               // zero junk row T of dGH
        float *zr = a.dgh + ((long)en * (T + 1) + T) * 4 * H;
        zr[ej] = 0.f; zr[H + ej] = 0.f; zr[2 * H + ej] = 0.f; zr[3 * H + ej] = 0.f;
    }
    // the saved activations of step t-1 are requested while step t runs (same reason as in the forward kernel)
    float p_dout = 0.f, p_i = 0.f, p_f = 0.f, p_g = 0.f, p_o = 0.f, p_c = 0.f, p_cp = 0.f;
    auto request = [&](int t) {
        const long row = (long)en * T + t;
        const float *gs = a.gates + row * 4 * H;
        p_dout = a.dout[row * H + ej];
        p_i = gs[ej]; p_f = gs[H + ej]; p_g = gs[2 * H + ej]; p_o = gs[3 * H + ej];
        p_cp = a.call[((long)en * (T + 1) + t) * H + ej];
    };
    if (emine) {
        p_c = a.call[((long)en * (T + 1) + T) * H + ej];
        request(T - 1);
    }
    coop_weights_ready(w);
    bool dead = false;
    for (int t = T - 1; t >= 0; --t) {
        const int par = t & 1;
        const unsigned epoch = ca.epoch0 + (unsigned)(T - t);              // 1, 2, ...
        if (ew) {
            float dpi = 0.f, dpf = 0.f, dpg = 0.f, dpo = 0.f;
            if (emine) {
                const long row = (long)en * T + t;
                const float dh = p_dout + carry;
                const float ig = p_i, fg = p_f, gg = p_g, og = p_o, cv = p_c, cp = p_cp;
                p_c = cp;                                       // c_{t-1} is the next step's c_t
                const float tc = tanhf(cv);
                const float dcv = dh * og * (1.f - tc * tc) + carry_c;
                dpi = dcv * gg * ig * (1.f - ig);
                dpf = dcv * cp * fg * (1.f - fg);
                dpg = dcv * ig * (1.f - gg * gg);
                dpo = dh * tc * og * (1.f - og);
                carry_c = dcv * fg;
                float *gi = a.dgi + row * 4 * H;
                gi[ej] = dpi; gi[H + ej] = dpf; gi[2 * H + ej] = dpg; gi[3 * H + ej] = dpo;
                float *gh = a.dgh + ((long)en * (T + 1) + t) * 4 * H;
                gh[ej] = dpi; gh[H + ej] = dpf; gh[2 * H + ej] = dpg; gh[3 * H + ej] = dpo;
            }
            dgs[es][eu] = dpi; dgs[es][U + eu] = dpf; dgs[es][2 * U + eu] = dpg; dgs[es][3 * U + eu] = dpo;
            if (emine && t > 0) request(t - 1);                 // behind the stores (see gru_bwd_coop_kernel)
        }
        coop_lds_barrier();
        if (t == 0) break;                                      // dh_{-1} is not needed
        // partial[jc] over this thread's RW rows, all NB windows
        f32x2 acc[NB];                                         // even / odd rows
#pragma unroll
        for (int s = 0; s < NB; ++s) acc[s] = f32x2{0.f, 0.f};
#pragma unroll
        for (int i4 = 0; i4 < RW / 4; ++i4) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                const float4 d4 = *reinterpret_cast<const float4 *>(&dgs[s][half * RW + 4 * i4]);
                acc[s] = pk_fma(w[2 * i4 + 1], pk_hi(d4), pk_fma(w[2 * i4], pk_lo(d4), acc[s]));
            }
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) part[half][s][jc] = acc[s].x + acc[s].y;
        coop_lds_barrier();
        auto column = [&](int s, int k) {
            float v = part[0][s][k];
#pragma unroll
            for (int hh = 1; hh < HALVES; ++hh) v += part[hh][s][k];
            return v;
        };
        // publish the columns other members own (this member's own columns stay in LDS)
        for (int idx = tid; idx < NB * H; idx += 512) {
            const int s = idx / H, k = idx - s * H;
            if (k / U == member) continue;
            const float v = column(s, k);
            COOP_GLOBAL gu64_t *slot =
                (COOP_GLOBAL gu64_t *)(ca.comm + ((((long)group * 2 + par) * G + member) * NB + s) * H + k);
            __hip_atomic_store(slot, ((gu64_t)epoch << 32) | (gu64_t)__float_as_uint(dead ? NAN : v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ew) {
            float sum = column(es, ej);
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
        }
        // no barrier here: dgs is rewritten before the next step's first barrier, part after it, and every thread has
        // finished reading both when it gets there; `dead` stays with the thread that timed out (its carry is poisoned)
    }
}

template <int H> static bool lstm_coop_fwd_fits(int nb, unsigned grid, int n_cus)
{
    static int ok[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};        // 0 unknown, 1 fits a CU, -1 does not
    if (ok[nb] == 0)
        ok[nb] = (nb == 1 ? coop_fits(lstm_fwd_coop_kernel<H, 1>, 1, 1) : nb == 2 ? coop_fits(lstm_fwd_coop_kernel<H, 2>, 1, 1)
                  : nb == 4 ? coop_fits(lstm_fwd_coop_kernel<H, 4>, 1, 1) : coop_fits(lstm_fwd_coop_kernel<H, 8>, 1, 1)) ? 1 : -1;
    return ok[nb] == 1 && (int)grid <= n_cus;
}
template <int H> static bool lstm_coop_bwd_fits(int nb, unsigned grid, int n_cus)
{
    static int ok[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ok[nb] == 0)
        ok[nb] = (nb == 1 ? coop_fits(lstm_bwd_coop_kernel<H, 1>, 1, 1) : nb == 2 ? coop_fits(lstm_bwd_coop_kernel<H, 2>, 1, 1)
                  : nb == 4 ? coop_fits(lstm_bwd_coop_kernel<H, 4>, 1, 1) : coop_fits(lstm_bwd_coop_kernel<H, 8>, 1, 1)) ? 1 : -1;
    return ok[nb] == 1 && (int)grid <= n_cus;
}
template <int H> static void launch_lstm_coop_fwd(int nb, dim3 grid, hipStream_t st, const LstmCoopArgs &ca)
{
    coop_count_launch();
    if (nb == 1) hipLaunchKernelGGL((lstm_fwd_coop_kernel<H, 1>), grid, dim3(512), 0, st, ca);
    else if (nb == 2) hipLaunchKernelGGL((lstm_fwd_coop_kernel<H, 2>), grid, dim3(512), 0, st, ca);
    else if (nb == 4) hipLaunchKernelGGL((lstm_fwd_coop_kernel<H, 4>), grid, dim3(512), 0, st, ca);
    else hipLaunchKernelGGL((lstm_fwd_coop_kernel<H, 8>), grid, dim3(512), 0, st, ca);
}
template <int H> static void launch_lstm_coop_bwd(int nb, dim3 grid, hipStream_t st, const LstmCoopArgs &ca)
{
    coop_count_launch();
    if (nb == 1) hipLaunchKernelGGL((lstm_bwd_coop_kernel<H, 1>), grid, dim3(512), 0, st, ca);
    else if (nb == 2) hipLaunchKernelGGL((lstm_bwd_coop_kernel<H, 2>), grid, dim3(512), 0, st, ca);
    else if (nb == 4) hipLaunchKernelGGL((lstm_bwd_coop_kernel<H, 4>), grid, dim3(512), 0, st, ca);
    else hipLaunchKernelGGL((lstm_bwd_coop_kernel<H, 8>), grid, dim3(512), 0, st, ca);
}

struct LstmLayout {
    int N, T, Din, H, layers, G;
    // saved, per layer
    float *gates[8], *hall[8], *call[8], *outl[8];
    size_t saved_bytes;
    // scratch
    float *gi, *dgi, *dgh, *dxa, *dxb, *wt_l[8], *cs, *tn, *tn2;
    float *dgi_l[8], *dgh_l[8];            // layers 1..: gate gradients of their own (deferred tail: the side stream still reads them)
    size_t tn2_bytes;
    float4 *wpack;
    gu64_t *comm;
    size_t comm_bytes;
    size_t tn_bytes, scratch_bytes;
};

int lstm_layout(LstmLayout &g, int G, int N, int T, int Din, int H, int layers, void *saved, void *scratch)
{
    const char *who = G == 4 ? "lstm" : "rnn";
    CPC_REQUIRE(N > 0 && T > 0 && Din > 0, "%s: bad shape n=%d t=%d in=%d", who, N, T, Din);
    CPC_REQUIRE(H % 4 == 0 && H >= 4 && H <= 1024, "%s: hidden %d must be a multiple of 4 and <= 1024", who, H);
    CPC_REQUIRE(layers >= 1 && layers <= 8, "%s: 1..8 layers supported (got %d)", who, layers);
    g.N = N; g.T = T; g.Din = Din; g.H = H; g.layers = layers; g.G = G;
    Carver sv(saved);
    for (int l = 0; l < layers; ++l) {
        g.gates[l] = G == 4 ? sv.take<float>((size_t)N * T * 4 * H) : nullptr;
        g.hall[l] = sv.take<float>((size_t)N * (T + 1) * H);
        g.call[l] = G == 4 ? sv.take<float>((size_t)N * (T + 1) * H) : nullptr;
        g.outl[l] = (l + 1 < layers) ? sv.take<float>((size_t)N * T * H) : nullptr;
    }
    g.saved_bytes = sv.used();
    Carver sc(scratch);
    const int dmax = std::max(Din, H);
    g.gi = sc.take<float>((size_t)N * T * G * H);
    g.dgi = g.gi;                                     // forward's GI and backward's dGI never coexist
    g.dgh = sc.take<float>((size_t)N * (T + 1) * G * H);
    g.dxa = sc.take<float>((size_t)N * T * dmax);
    g.dxb = sc.take<float>((size_t)N * T * dmax);
    for (int l = 0; l < layers; ++l) g.wt_l[l] = sc.take<float>((size_t)G * H * dmax);        // W_ih^T of every layer (backward)
    g.wpack = sc.take<float4>((size_t)G * H * H / 4);
    g.cs = sc.take<float>(colsum_rows_scratch_bytes(G * H) / sizeof(float));
    g.comm_bytes = G == 4 ? coop_comm_bytes(H, N) : 256;
    g.comm = sc.take<gu64_t>(g.comm_bytes / sizeof(gu64_t));
    g.tn_bytes = std::max(gemm_tn_scratch_bytes(G * H, H, (long)N * (T + 1)), gemm_tn_scratch_bytes(G * H, dmax, (long)N * T));
    g.tn_bytes = std::max(g.tn_bytes, gemm_tn_scratch_bytes(G * H, Din, (long)N * T));
    // the same room serves an ordered K split of the projections when they have few tiles
    g.tn_bytes = std::max(g.tn_bytes, std::max(gemm_nt_scratch_bytes((long)N * T, G * H, dmax), gemm_nt_scratch_bytes((long)N * T, dmax, G * H)));
    g.tn = sc.take<float>(g.tn_bytes / sizeof(float));
    g.tn2_bytes = gemm_nt_scratch_bytes((long)N * T, dmax, G * H);      // (the input-gradient product's K split beside a deferred tail)
    g.tn2 = sc.take<float>(g.tn2_bytes / sizeof(float));
    g.dgi_l[0] = g.dgi; g.dgh_l[0] = g.dgh;
    for (int l = 1; l < layers; ++l) {
        g.dgi_l[l] = sc.take<float>((size_t)N * T * G * H);
        g.dgh_l[l] = sc.take<float>((size_t)N * (T + 1) * G * H);
    }
    g.scratch_bytes = sc.used();
    return CPC_OK;
}

// cooperative kernels exist for the LSTM at H = 256 / 512 (CPC_LSTM_STREAM forces the streaming ones)
int lstm_coop_windows(int G, int H, int N, int *members)
{
    static const bool coop_off = getenv("CPC_LSTM_STREAM") != nullptr;
    static const int n_cus = coop_cu_count();
    if (G != 4 || coop_off || !coop_allowed()) return 0;
    return coop_windows_per_group(H, N, n_cus, members);
}

void lstm_threads(int H, int G, int &hp, int &kq)
{
    hp = std::max(64, (int)cdiv(H, 64) * 64);
    kq = std::max(1, std::min(1024 / hp, G * H / 4));
    kq = std::min(kq, H / 4);
}

}  // namespace

template <int G>
static int lstm_forward(const float *x, const float *const *prm, const float *h0, const float *c0, float *out, float *h_last,
                        float *c_last, void *saved, void *scratch, int N, int T, int Din, int H, int layers, hipStream_t st)
{
    LstmLayout g;
    CPC_TRY(lstm_layout(g, G, N, T, Din, H, layers, saved, scratch));
    int hp, kq;
    lstm_threads(H, G, hp, kq);
    const float *xin = x;
    int din = Din;
    for (int l = 0; l < layers; ++l) {
        const float *w_ih = prm[4 * l], *w_hh = prm[4 * l + 1], *b_ih = prm[4 * l + 2], *b_hh = prm[4 * l + 3];
        RowMap none{};
        none.splitk_scratch = g.tn; none.splitk_bytes = g.tn_bytes;
        CPC_TRY(gemm_nt(xin, din, w_ih, din, g.gi, (long)G * H, b_ih, (long)N * T, G * H, din, none, st));
        LstmArgs a{};
        a.gi = g.gi; a.wpack = g.wpack; a.whh = w_hh; a.bhh = b_hh;
        a.h0 = h0 ? h0 + (size_t)l * N * H : nullptr;
        a.c0 = c0 ? c0 + (size_t)l * N * H : nullptr;
        a.out = (l + 1 < layers) ? g.outl[l] : out;
        a.hall = g.hall[l]; a.call = g.call[l]; a.gates = g.gates[l];
        a.hlast = h_last ? h_last + (size_t)l * N * H : nullptr;
        a.clast = c_last ? c_last + (size_t)l * N * H : nullptr;
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq;
        int members = 0;
        int nb = lstm_coop_windows(G, H, N, &members);
        if (nb != 0 && !(H == 256 ? lstm_coop_fwd_fits<256>(nb, (unsigned)(cdiv(N, nb) * members), coop_cu_count())
                                  : lstm_coop_fwd_fits<512>(nb, (unsigned)(cdiv(N, nb) * members), coop_cu_count())))
            nb = 0;                             // not resident all at once: the streaming kernel has no such requirement
        if (nb != 0) {
            LstmCoopArgs ca{};
            ca.g = a; ca.groups = (int)cdiv(N, nb);
            ca.xcd_map = (ca.groups % 8 == 0) ? 1 : 0;
            ca.err = coop_error_word(); ca.fault = coop_fault_injection();
            CPC_TRY(coop_comm_acquire(sizeof(gu64_t) * (size_t)ca.groups * 2 * nb * H, T, st, &ca.comm, &ca.epoch0));
            ProfScope prof(PROF_GRU_FWD, st);
            const dim3 grid((unsigned)(ca.groups * members));
            if (H == 256) launch_lstm_coop_fwd<256>(nb, grid, st, ca);
            else launch_lstm_coop_fwd<512>(nb, grid, st, ca);
        } else {
            hipLaunchKernelGGL(lstm_pack_fwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H, G);
            CPC_CHECK_LAUNCH("lstm_pack_fwd_kernel");
            ProfScope prof(PROF_GRU_FWD, st);
            const size_t lds = sizeof(float) * (cdiv(H, 4) * 4 + (size_t)kq * G * hp);
            hipLaunchKernelGGL(lstm_fwd_kernel<G>, dim3((unsigned)N), dim3(kq * hp), lds, st, a);
        }
        CPC_CHECK_LAUNCH("lstm_fwd_kernel");
        xin = a.out;
        din = H;
    }
    return CPC_OK;
}

template <int G>
static int lstm_backward(const float *x, const float *const *prm, const float *dout, void *saved, void *scratch, float *dx,
                         float *const *grads, int N, int T, int Din, int H, int layers, hipStream_t st, bool defer_tail = false)
{
    LstmLayout g;
    CPC_TRY(lstm_layout(g, G, N, T, Din, H, layers, saved, scratch));
    int hp, kq;
    lstm_threads(H, G, hp, kq);
    const float *dcur = dout;
    // W_ih^T of every layer that has an input gradient, in front of the first recurrent kernel: the transposes depend on the weights
    // only, and a small kernel queued BEHIND a recurrent kernel starts while the deferred criterion sum / the weight-gradient
    // products hold the chip on the side stream -- seen at 212 us (3 MB) on the critical path of CPC-large, 5 us alone
    for (int l = layers - 1; l >= 0; --l)
        if (l > 0 || dx != nullptr) CPC_TRY(transpose2d(prm[4 * l], g.wt_l[l], G * H, (l == 0) ? Din : H, st));
    for (int l = layers - 1; l >= 0; --l) {
        const float *w_hh = prm[4 * l + 1];
        const float *xin = (l == 0) ? x : g.outl[l - 1];
        const int din = (l == 0) ? Din : H;
        LstmArgs a{};
        a.wpack = g.wpack; a.whh = w_hh; a.hall = g.hall[l]; a.call = g.call[l]; a.gates = g.gates[l];
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq;
        // (deferred tail: every layer's gate gradients stay where they are until the side stream has used them)
        float *const dgi = defer_tail ? g.dgi_l[l] : g.dgi, *const dgh = defer_tail ? g.dgh_l[l] : g.dgh;
        a.dout = dcur; a.dgi = dgi; a.dgh = dgh;
        CPC_TRY(infonce_deferred_mark(st));       // (see infonce_deferred_start below)
        int members = 0;
        int nb = lstm_coop_windows(G, H, N, &members);
        if (nb != 0 && !(H == 256 ? lstm_coop_bwd_fits<256>(nb, (unsigned)(cdiv(N, nb) * members), coop_cu_count())
                                  : lstm_coop_bwd_fits<512>(nb, (unsigned)(cdiv(N, nb) * members), coop_cu_count())))
            nb = 0;
        if (nb != 0) {
            LstmCoopArgs ca{};
            ca.g = a; ca.groups = (int)cdiv(N, nb);
            ca.xcd_map = (ca.groups % 8 == 0) ? 1 : 0;
            ca.err = coop_error_word(); ca.fault = coop_fault_injection();
            CPC_TRY(coop_comm_acquire(sizeof(gu64_t) * (size_t)ca.groups * 2 * members * nb * H, T, st, &ca.comm, &ca.epoch0));
            ProfScope prof(PROF_GRU_BWD, st);
            const dim3 grid((unsigned)(ca.groups * members));
            if (H == 256) launch_lstm_coop_bwd<256>(nb, grid, st, ca);
            else launch_lstm_coop_bwd<512>(nb, grid, st, ca);
        } else {
            hipLaunchKernelGGL(lstm_pack_bwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H, G);
            CPC_CHECK_LAUNCH("lstm_pack_bwd_kernel");
            ProfScope prof(PROF_GRU_BWD, st);
            const size_t lds = sizeof(float) * ((size_t)G * H + (size_t)kq * hp);
            hipLaunchKernelGGL(lstm_bwd_kernel<G>, dim3((unsigned)N), dim3(kq * hp), lds, st, a);
        }
        CPC_CHECK_LAUNCH("lstm_bwd_kernel");
        CPC_TRY(infonce_deferred_start(st));      // (no-op unless a deferred criterion backward is waiting to run beside this)
        const int GH = G * H;
        // (defer_tail: layer 0's weight gradients on the library's side stream, as in gru_backward)
        hipStream_t wst = st;
        const bool tail = defer_tail;
        if (tail) CPC_TRY(side_tail_begin(st, &wst));
        // dW_hh[g][k] = sum_{n,t} dG[n,t][g] * h_{t-1}[n][k]   (hall row t is h_{t-1}; row T of dGH is zero)
        CPC_TRY(gemm_tn(dgh, GH, g.hall[l], H, grads[4 * l + 1], H, GH, H, (long)N * (T + 1), g.tn, g.tn_bytes, 0, 0, wst));
        CPC_TRY(colsum_rows(dgh, GH, (long)N * (T + 1), GH, grads[4 * l + 3], g.cs, wst));
        // dW_ih[g][k] = sum dG[n,t][g] * x[n,t][k]
        CPC_TRY(gemm_tn(dgi, GH, xin, din, grads[4 * l], din, GH, din, (long)N * T, g.tn, g.tn_bytes, 0, 0, wst));
        CPC_TRY(colsum_rows(dgi, GH, (long)N * T, GH, grads[4 * l + 2], g.cs, wst));
        if (tail) CPC_TRY(side_tail_end());
        // dX = dG . W_ih
        float *dxl = (l == 0) ? dx : ((l % 2) ? g.dxa : g.dxb);
        if (dxl != nullptr) {
            RowMap none{};
            if (tail) { none.splitk_scratch = g.tn2; none.splitk_bytes = g.tn2_bytes; }        // (g.tn is the side stream's now)
            else { none.splitk_scratch = g.tn; none.splitk_bytes = g.tn_bytes; }
            CPC_TRY(gemm_nt(dgi, GH, g.wt_l[l], GH, dxl, din, nullptr, (long)N * T, din, GH, none, st));
        }
        dcur = dxl;
    }
    return CPC_OK;
}

}  // namespace cpc

extern "C" size_t cpc_lstm_saved_bytes(int n, int t, int dim_in, int hidden, int layers)
{
    cpc::LstmLayout g;
    if (cpc::lstm_layout(g, 4, n, t, dim_in, hidden, layers, nullptr, nullptr) != CPC_OK) return 0;
    return g.saved_bytes;
}

extern "C" size_t cpc_lstm_scratch_bytes(int n, int t, int dim_in, int hidden, int layers)
{
    cpc::LstmLayout g;
    if (cpc::lstm_layout(g, 4, n, t, dim_in, hidden, layers, nullptr, nullptr) != CPC_OK) return 0;
    return g.scratch_bytes;
}

extern "C" int cpc_lstm_forward(const float *x, const float *const *params, const float *h0, const float *c0, float *out,
                                float *h_last, float *c_last, void *saved, void *scratch, int n, int t, int dim_in, int hidden,
                                int layers, cpc_stream_t stream)
{
    CPC_TRY(cpc::coop_error_take("cpc_lstm_forward"));      // a time-out of an earlier cooperative launch surfaces here
    return cpc::lstm_forward<4>(x, params, h0, c0, out, h_last, c_last, saved, scratch, n, t, dim_in, hidden, layers,
                                static_cast<hipStream_t>(stream));
}

extern "C" int cpc_lstm_backward(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                 float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                 cpc_stream_t stream)
{
    cpc::coop_count_backward_call();
    CPC_TRY(cpc::coop_error_take("cpc_lstm_backward"));      // a time-out of an earlier cooperative launch surfaces here
    return cpc::lstm_backward<4>(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int cpc_lstm_backward_deferred(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                          float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                          cpc_stream_t stream)
{
    cpc::coop_count_backward_call();
    CPC_TRY(cpc::coop_error_take("cpc_lstm_backward_deferred"));
    return cpc::lstm_backward<4>(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                                 static_cast<hipStream_t>(stream), true);
}

extern "C" size_t cpc_rnn_saved_bytes(int n, int t, int dim_in, int hidden, int layers)
{
    cpc::LstmLayout g;
    if (cpc::lstm_layout(g, 1, n, t, dim_in, hidden, layers, nullptr, nullptr) != CPC_OK) return 0;
    return g.saved_bytes;
}

extern "C" size_t cpc_rnn_scratch_bytes(int n, int t, int dim_in, int hidden, int layers)
{
    cpc::LstmLayout g;
    if (cpc::lstm_layout(g, 1, n, t, dim_in, hidden, layers, nullptr, nullptr) != CPC_OK) return 0;
    return g.scratch_bytes;
}

extern "C" int cpc_rnn_forward(const float *x, const float *const *params, const float *h0, float *out, float *h_last,
                               void *saved, void *scratch, int n, int t, int dim_in, int hidden, int layers,
                               cpc_stream_t stream)
{
    CPC_TRY(cpc::coop_error_take("cpc_rnn_forward"));      // a time-out of an earlier cooperative launch surfaces here
    return cpc::lstm_forward<1>(x, params, h0, nullptr, out, h_last, nullptr, saved, scratch, n, t, dim_in, hidden, layers,
                                static_cast<hipStream_t>(stream));
}

extern "C" int cpc_rnn_backward(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                cpc_stream_t stream)
{
    cpc::coop_count_backward_call();
    CPC_TRY(cpc::coop_error_take("cpc_rnn_backward"));      // a time-out of an earlier cooperative launch surfaces here
    return cpc::lstm_backward<1>(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                                 static_cast<hipStream_t>(stream));
}
