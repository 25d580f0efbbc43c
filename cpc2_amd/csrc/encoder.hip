// CPCEncoder on gfx950: conv0 + ChannelNorm + ReLU fused (HBM-bound), conv1..4 as implicit GEMMs (gemm_f32.hip:
// f32 products on the bf16 matrix pipe) over channel-last activations, ChannelNorm(+ReLU) row kernels, and the
// matching backward.
//
// Reference: /root/reference/cpc/model.py:63-108 (CPCEncoder), :27-60 (ChannelNorm).
//
// Layouts (H = hidden, N = windows, L[i] = length after layer i-1, L[0] = samples):
//   Y_i  (i=0..3)  output of layer i = input of layer i+1:  [N][R_i][H], R_i = s_{i+1} * (L[i+2] + 2);
//                  data row of position l is row l + p_{i+1}; the other rows are zero (conv padding).
//                  Output frame t of layer i+1 then reads the k*H CONTIGUOUS floats at row t*s, so the
//                  conv is a GEMM whose A rows overlap: A(m) = Y_i + m*s*H, m = n*Rv + t, Rv = L[i+2]+2
//                  (rows t >= L[i+2] of a sample are junk: never multiplied by the split kernels, which tile each
//                  sample's valid rows separately, and ignored downstream).
//   Xh_i (i=1..4)  normalised pre-activation xhat: [N*Rv_i][H] (conv output, normalised in place)
//   dU_i           gradient wrt conv output, SHIFTED by one row: row n*Rv + t + 1; rows 0 and > L of every
//                  sample are zero.  Backward-data (k == 2s) then is ONE GEMM with A(m) = dU + m*H (rows t_hi-1
//                  and t_hi, 2H contiguous floats) and the s phases side by side in N; weight-gradient a TN
//                  GEMM over m with A(m) = dU + (m+1)*H, B(m) = Y_{i-1} + m*s*H.
#include "common.h"
#include "rowcfg.h"

#include <algorithm>

namespace cpc {

constexpr int C0_TB = 64;     // conv0: output rows per block tile
constexpr int C0_K = 10, C0_S = 5, C0_P = 3;
#ifndef C0_AHEAD_N
#define C0_AHEAD_N 1
#endif
constexpr int C0_AHEAD = C0_AHEAD_N;   // conv0_bwd: dy rows requested ahead per lane group (measured 1..4, see the kernel)

struct Conv0Args {
    const float *x;        // windows 0 .. n_split - 1: [n_split][L0]
    const float *x2;       // windows n_split .. N - 1: [N - n_split][L0]  (the caller's past and future batches: train.py:99's cat without the copy)
    int n_split;
    const float *w;        // [H][10]
    const float *b;        // [H]
    const float *gamma;    // [H]
    const float *beta;     // [H]
    float *y;              // Y0 [N][R0][H]
    float *stats;          // [N*L1][2] mean, rstd
    int N, L0, L1, R0, halo;
    float eps;
    // backward only
    const float *dy;       // [N][L1][H]
    float *part;           // [slots][13][H]
    int tiles_per_sample, n_tiles;
#ifdef CPC_C0_DBG
    unsigned *dbg;         // [blocks][8]: wrapping sums of the bit patterns a block LOADED (parameters, dy rows, statistics, xs);
                           // waves whose HW_ID changed between entry and exit (context switch), wave 0's run time in 10 ns
                           // ticks and its HW_ID at entry / exit
#endif
};

__device__ __forceinline__ const float *conv0_window(const Conv0Args &a, long n)
{
    return n < a.n_split ? a.x + n * a.L0 : a.x2 + (n - a.n_split) * a.L0;
}

template <int H> __device__ __forceinline__ void conv0_load_segment(float *xs, const Conv0Args &a, int n, int t0)
{
    // xs[i] = x[n][5*t0 - 3 + i], i < 5*TB + 5, zero outside [0, L0)
    const float *xn = conv0_window(a, n);
    for (int i = threadIdx.x; i < C0_S * C0_TB + C0_K - C0_S; i += blockDim.x) {
        const int pos = C0_S * t0 - C0_P + i;
        xs[i] = (pos >= 0 && pos < a.L0) ? xn[pos] : 0.f;
    }
}

template <int H> __global__ __launch_bounds__(256) void conv0_fwd_kernel(Conv0Args a)
{
    using Cfg = RowCfg<H>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL, RPW = Cfg::RPW;
    __shared__ float xs[C0_S * C0_TB + 8];

    const int tile = blockIdx.x;
    const int n = tile / a.tiles_per_sample;
    const int t0 = (tile - n * a.tiles_per_sample) * C0_TB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gi = lane / G;

    conv0_load_segment<H>(xs, a, n, t0);

    float wreg[VPL][4][C0_K], breg[VPL][4], gam[VPL][4], bet[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (v * G + gl) * 4 + e;
#pragma unroll
            for (int j = 0; j < C0_K; ++j) wreg[v][e][j] = a.w[c * C0_K + j];
            breg[v][e] = a.b[c];
            gam[v][e] = a.gamma[c];
            bet[v][e] = a.beta[c];
        }

    // zero the halo rows of this sample (conv padding of layer 1)
    if (t0 == 0 || t0 + C0_TB >= a.L1) {
        const int lo0 = 0, hi0 = a.halo;                    // front halo
        const int lo1 = a.halo + a.L1, hi1 = a.R0;          // back halo
        const int nfront = (t0 == 0) ? (hi0 - lo0) : 0;
        const int nback = (t0 + C0_TB >= a.L1) ? (hi1 - lo1) : 0;
        const int total4 = (nfront + nback) * (H / 4);
        for (int i = threadIdx.x; i < total4; i += blockDim.x) {
            int row = i / (H / 4);
            const int c4 = i - row * (H / 4);
            row = row < nfront ? lo0 + row : lo1 + (row - nfront);
            reinterpret_cast<float4 *>(a.y + ((long)n * a.R0 + row) * H)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    constexpr int ROWS_PER_PASS = 4 * RPW;
    for (int it = 0; it < C0_TB / ROWS_PER_PASS; ++it) {
        const int slot = it * ROWS_PER_PASS + wave * RPW + gi;
        const int t = t0 + slot;
        float xr[C0_K];
#pragma unroll
        for (int j = 0; j < C0_K; ++j) xr[j] = xs[C0_S * slot + j];

        float u[VPL][4];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = breg[v][e];
#pragma unroll
                for (int j = 0; j < C0_K; ++j) acc = fmaf(wreg[v][e][j], xr[j], acc);
                u[v][e] = acc;
                s += acc;
            }
        const float mean = group_sum<G>(s) * (1.f / H);
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u[v][e] -= mean;
                ss = fmaf(u[v][e], u[v][e], ss);
            }
        const float rstd = rsqrtf(group_sum<G>(ss) * (1.f / (H - 1)) + a.eps);
        if (t < a.L1) {
            float *yrow = a.y + ((long)n * a.R0 + a.halo + t) * H;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                float4 o;
                o.x = fmaxf(fmaf(u[v][0] * rstd, gam[v][0], bet[v][0]), 0.f);
                o.y = fmaxf(fmaf(u[v][1] * rstd, gam[v][1], bet[v][1]), 0.f);
                o.z = fmaxf(fmaf(u[v][2] * rstd, gam[v][2], bet[v][2]), 0.f);
                o.w = fmaxf(fmaf(u[v][3] * rstd, gam[v][3], bet[v][3]), 0.f);
                reinterpret_cast<float4 *>(yrow)[v * G + gl] = o;
            }
            if (gl == 0) {
                a.stats[((long)n * a.L1 + t) * 2 + 0] = mean;
                a.stats[((long)n * a.L1 + t) * 2 + 1] = rstd;
            }
        }
    }
}

// Backward of relu(norm(conv0(x))): recomputes the conv (10 MACs per output) instead of saving it.
// Persistent blocks stride over tiles; every lane group keeps its 13*4*VPL partial sums in registers
// and writes them once: part[slot][q][c], q = 0..9 dW tap, 10 db, 11 dgamma, 12 dbeta.
template <int H> __global__ __launch_bounds__(256) void conv0_bwd_kernel(Conv0Args a)
{
    using Cfg = RowCfg<H>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL, RPW = Cfg::RPW;
    __shared__ float xs[C0_S * C0_TB + 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gi = lane / G;

    // Hidden 512 (two float4 per lane): the forward weights from LDS ([tap][channel]: a lane's four channels of a tap are one
    // 16-byte read) instead of 80 registers per lane -- 212 VGPRs, two waves per SIMD instead of one: 0.50 -> 0.40 ms at CPC-large.
    // Up to hidden 256 the registers win (three waves either way, and 0.165 against 0.17-0.18 ms with the reads in the loop).
    constexpr bool WL = H >= 512;
    __shared__ __attribute__((aligned(16))) float wl[WL ? C0_K : 1][WL ? H : 4];
    float wreg[WL ? 1 : VPL][4][C0_K];
    if constexpr (WL)
        for (int i = threadIdx.x; i < C0_K * H; i += blockDim.x) wl[i % C0_K][i / C0_K] = a.w[i];
    float breg[VPL][4], gam[VPL][4], bet[VPL][4];
    float dwacc[VPL][4][C0_K], dbacc[VPL][4], dgacc[VPL][4], dbeacc[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (v * G + gl) * 4 + e;
#pragma unroll
            for (int j = 0; j < C0_K; ++j) {
                if constexpr (!WL) wreg[v][e][j] = a.w[c * C0_K + j];
                dwacc[v][e][j] = 0.f;
            }
            breg[v][e] = a.b[c];
            gam[v][e] = a.gamma[c];
            bet[v][e] = a.beta[c];
            dbacc[v][e] = dgacc[v][e] = dbeacc[v][e] = 0.f;
        }

#ifdef CPC_C0_DBG
    __shared__ unsigned dbg_s[8];
    if (threadIdx.x < 8) dbg_s[threadIdx.x] = 0u;
    const unsigned hwid0 = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
    unsigned dbg_dy = 0u, dbg_st = 0u;
    {
        unsigned pp = 0u;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < C0_K; ++j) pp += __float_as_uint(wreg[v][e][j]);
                pp += __float_as_uint(breg[v][e]) + __float_as_uint(gam[v][e]) + __float_as_uint(bet[v][e]);
            }
        __syncthreads();
        atomicAdd(&dbg_s[0], pp);
    }
#endif
    constexpr int ROWS_PER_PASS = 4 * RPW;
    // the dy rows and the statistics of passes it+1 .. it+C0_AHEAD are in flight while pass it is worked on: with two or three
    // waves per SIMD nothing else hides the HBM round trip of a load that is consumed right away.  Measured at hidden 256
    // (profiles/r03_conv0_bwd_ahead.txt): what pays is three waves per SIMD, not depth -- the pass loop left to the
    // compiler's unroller took 192-214 VGPRs (two waves) and 180 us; held at one ring per trip (#pragma unroll 1) it takes
    // <= 160 VGPRs and 158 us with 1 or 3 rows ahead, while 2 or 4 ahead (175-243 VGPRs, two waves) take 234-242 us.
    constexpr int NPASS = C0_TB / ROWS_PER_PASS;
    constexpr int AHEAD = NPASS < C0_AHEAD ? NPASS : C0_AHEAD;
    float4 gq[AHEAD][VPL];
    float mq[AHEAD], rq[AHEAD];
    auto request = [&](int n, int t, int ring) {
        const bool ok = t < a.L1;
        const long row = (long)n * a.L1 + (ok ? t : 0);
        mq[ring] = a.stats[row * 2 + 0];
        rq[ring] = ok ? a.stats[row * 2 + 1] : 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) gq[ring][v] = reinterpret_cast<const float4 *>(a.dy + row * H)[v * G + gl];
    };
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int n = tile / a.tiles_per_sample;
        const int t0 = (tile - n * a.tiles_per_sample) * C0_TB;
#pragma unroll
        for (int d = 0; d < AHEAD; ++d) request(n, t0 + d * ROWS_PER_PASS + wave * RPW + gi, d);
        __syncthreads();
        conv0_load_segment<H>(xs, a, n, t0);
        __syncthreads();
#pragma unroll 1
        for (int it0 = 0; it0 < NPASS; it0 += AHEAD)
#pragma unroll
        for (int ring = 0; ring < AHEAD; ++ring) {
            const int it = it0 + ring;
            if (NPASS % AHEAD != 0 && it >= NPASS) break;
            const int slot = it * ROWS_PER_PASS + wave * RPW + gi;
            const int t = t0 + slot;
            const bool valid = t < a.L1;
            float xr[C0_K];
#pragma unroll
            for (int j = 0; j < C0_K; ++j) xr[j] = xs[C0_S * slot + j];
            const float mean = mq[ring], rstd = rq[ring];       // rstd = 0 on rows past the end: du = 0 there
            float4 gcur[VPL];
#pragma unroll
            for (int v = 0; v < VPL; ++v) gcur[v] = valid ? gq[ring][v] : make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef CPC_C0_DBG
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                dbg_dy += __float_as_uint(gcur[v].x) + __float_as_uint(gcur[v].y) + __float_as_uint(gcur[v].z) + __float_as_uint(gcur[v].w);
            if (gl == 0) dbg_st += __float_as_uint(mean) + __float_as_uint(rstd);
#endif
            if (it + AHEAD < NPASS) request(n, t + AHEAD * ROWS_PER_PASS, ring);
            float xh[VPL][4], gx[VPL][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const float gy[4] = {gcur[v].x, gcur[v].y, gcur[v].z, gcur[v].w};
                float cacc[4] = {breg[v][0], breg[v][1], breg[v][2], breg[v][3]};
                if constexpr (WL) {
                    unsigned woff = (unsigned)(v * G + gl) * 16u;
                    asm volatile("" : "+v"(woff));         // (opaque per pass: the reads stay in the loop instead of in registers)
#pragma unroll
                    for (int j = 0; j < C0_K; ++j) {
                        const float4 w4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(&wl[j][0]) + woff);
                        cacc[0] = fmaf(w4.x, xr[j], cacc[0]); cacc[1] = fmaf(w4.y, xr[j], cacc[1]);
                        cacc[2] = fmaf(w4.z, xr[j], cacc[2]); cacc[3] = fmaf(w4.w, xr[j], cacc[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < C0_K; ++j) cacc[e] = fmaf(wreg[v][e][j], xr[j], cacc[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float acc = cacc[e];
                    const float xhat = (acc - mean) * rstd;
                    const float act = fmaf(xhat, gam[v][e], bet[v][e]);
                    const float g = act > 0.f ? gy[e] : 0.f;
                    dbeacc[v][e] += g;
                    dgacc[v][e] = fmaf(g, xhat, dgacc[v][e]);
                    xh[v][e] = xhat;
                    gx[v][e] = g * gam[v][e];
                    s1 += gx[v][e];
                    s2 = fmaf(gx[v][e], xhat, s2);
                }
            }
            s1 = group_sum<G>(s1) * (1.f / H);
            s2 = group_sum<G>(s2) * (1.f / (H - 1));
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float du = rstd * (gx[v][e] - s1 - xh[v][e] * s2);   // 0 when !valid (rstd = 0)
                    dbacc[v][e] += du;
#pragma unroll
                    for (int j = 0; j < C0_K; ++j) dwacc[v][e][j] = fmaf(du, xr[j], dwacc[v][e][j]);
                }
        }
    }

    // block-level sum of the 4*RPW lane groups' partials, one row of part[] per block; the groups take turns on one
    // 13 H float LDS row (a [groups][13 H] array would cost a workgroup per CU)
    __shared__ float red[13 * H];
    for (int turn = 0; turn < 4 * RPW; ++turn) {
        if (turn == wave * RPW + gi) {
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = (v * G + gl) * 4 + e;
#pragma unroll
                    for (int j = 0; j < C0_K; ++j) red[j * H + c] = (turn ? red[j * H + c] : 0.f) + dwacc[v][e][j];
                    red[10 * H + c] = (turn ? red[10 * H + c] : 0.f) + dbacc[v][e];
                    red[11 * H + c] = (turn ? red[11 * H + c] : 0.f) + dgacc[v][e];
                    red[12 * H + c] = (turn ? red[12 * H + c] : 0.f) + dbeacc[v][e];
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < 13 * H; c += 256) a.part[(long)blockIdx.x * 13 * H + c] = red[c];
#ifdef CPC_C0_DBG
    atomicAdd(&dbg_s[1], dbg_dy);
    atomicAdd(&dbg_s[2], dbg_st);
    {
        // is the staged signal still what was loaded?  (LDS words changed behind the block's back: another workgroup's stores)
        const int last = blockIdx.x < a.n_tiles ? blockIdx.x + ((a.n_tiles - 1 - blockIdx.x) / gridDim.x) * gridDim.x : -1;
        if (last >= 0) {
            const int n = last / a.tiles_per_sample, t0 = (last - n * a.tiles_per_sample) * C0_TB;
            const float *xn = conv0_window(a, n);
            for (int i = threadIdx.x; i < C0_S * C0_TB + C0_K - C0_S; i += blockDim.x) {
                const int pos = C0_S * t0 - C0_P + i;
                const float want = (pos >= 0 && pos < a.L0) ? xn[pos] : 0.f;
                if (__float_as_uint(want) != __float_as_uint(xs[i])) { atomicAdd(&dbg_s[4], 1u); atomicMax(&dbg_s[3], (unsigned)i); }
            }
        }
    }
    {
        const unsigned hwid1 = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        if (threadIdx.x == 0) {
            dbg_s[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t_in);
            dbg_s[6] = hwid0;
            dbg_s[7] = hwid1;
        }
    }
    __syncthreads();
    if (threadIdx.x < 8 && a.dbg != nullptr) a.dbg[blockIdx.x * 8 + threadIdx.x] = dbg_s[threadIdx.x];
#endif
}

// sums[13][H] -> conv0.weight grad [H][1][10], bias grad, norm weight/bias grads
__global__ void conv0_finalize_kernel(const float *sums, float *dw, float *db, float *dgamma, float *dbeta, int H)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    for (int j = 0; j < C0_K; ++j) dw[c * C0_K + j] = sums[j * H + c];
    db[c] = sums[10 * H + c];
    dgamma[c] = sums[11 * H + c];
    dbeta[c] = sums[12 * H + c];
}

// ------------------------------------------------------------------------------------------------
struct NormArgs {
    float *u;              // fwd: conv output [N*Rv][H], overwritten by xhat.  bwd: xhat (read)
    const float *gamma, *beta;
    float *rstd;           // [N*Rv]
    float *y;              // fwd: next layer input [N][Rnext][H] (or z)
    int N, Lout, Rv, Rnext, halo;
    float eps;
    // backward
    const float *dy;       // [N][Lout][H]
    float *du;             // [N*Rv + 1][H], shifted by one row
    float *part;           // [slots][3][H]: dgamma, dbeta, dbias
    // fwd: the conv output as the partial products of a K split (nslabs > 0): row m is the sum, in slab order, of
    // slabs[s * slab_stride + m * H ..] -- the sum the split's own reduction pass would have written to u
    const float *slabs;
    int nslabs;
    long slab_stride;
};

// 16 bytes of row m of the conv output: from u, or summed over the K split's slabs in their order (bit-identical to the
// reduction pass: 0 + s0 + s1 + ...)
__device__ __forceinline__ float4 norm_in4(const NormArgs &a, long m, int H, int c4)
{
    if (a.nslabs == 0) return reinterpret_cast<const float4 *>(a.u + m * H)[c4];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sl = 0; sl < a.nslabs; ++sl) {
        const float4 v = reinterpret_cast<const float4 *>(a.slabs + sl * a.slab_stride + m * H)[c4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    return acc;
}

// (sample n, row r inside it) of signal row `base + off`, R rows per sample.  `base` is the same for the whole workgroup (a function
// of blockIdx and loop counters): ITS division is one scalar-unit sequence per tile; the lanes add their small offset and carry.
// Rounds 1-5 wrote `row / R` per row and lane: a 64-bit division by a run-time value is ~70 VALU instructions -- a third of
// conv0_fwd_pl_kernel's vector work (87 M of them per launch, the kernel's time at four cycles each: profiles/r06_a_kernel_counters.txt).
struct RowBase { long n; int r; };
__device__ __forceinline__ RowBase row_base(long base, int R)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)base >> 32));
    const long b = (long)(((unsigned long long)hi << 32) | lo);
    RowBase q;
    q.n = b / R;
    q.r = (int)(b - q.n * R);
    return q;
}
__device__ __forceinline__ void row_at(const RowBase &q, int off, int R, long &n, int &r)
{
    n = q.n;
    r = q.r + off;
    while (r >= R) { r -= R; ++n; }          // (off is a few rows: at most one trip unless the samples are shorter than a tile)
}

template <int H> __global__ __launch_bounds__(256) void norm_fwd_kernel(NormArgs a)
{
    using Cfg = RowCfg<H>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL, RPW = Cfg::RPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gi = lane / G;
    float gam[VPL][4], bet[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gam[v][e] = a.gamma[(v * G + gl) * 4 + e];
            bet[v][e] = a.beta[(v * G + gl) * 4 + e];
        }
    const long total = (long)a.N * a.Rnext;
    const long stride = (long)gridDim.x * 4 * RPW;
    for (long base = (long)blockIdx.x * 4 * RPW; base < total; base += stride) {
        const long row = base + wave * RPW + gi;          // row of Y
        const bool in_range = row < total;
        long nl; int rr;
        row_at(row_base(base, a.Rnext), wave * RPW + gi, a.Rnext, nl, rr);
        const int n = in_range ? (int)nl : 0;
        const int t = in_range ? rr - a.halo : -1;
        const bool valid = in_range && t >= 0 && t < a.Lout;
        const long m = (long)n * a.Rv + (valid ? t : 0);
        float4 x4[VPL];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            x4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) x4[v] = norm_in4(a, m, H, v * G + gl);
            s += (x4[v].x + x4[v].y) + (x4[v].z + x4[v].w);
        }
        const float mean = group_sum<G>(s) * (1.f / H);
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            x4[v].x -= mean; x4[v].y -= mean; x4[v].z -= mean; x4[v].w -= mean;
            ss = fmaf(x4[v].x, x4[v].x, ss); ss = fmaf(x4[v].y, x4[v].y, ss);
            ss = fmaf(x4[v].z, x4[v].z, ss); ss = fmaf(x4[v].w, x4[v].w, ss);
        }
        const float rstd = rsqrtf(group_sum<G>(ss) * (1.f / (H - 1)) + a.eps);
        if (!in_range) continue;
        float *yrow = a.y + row * H;
        if (valid) {
            if (gl == 0) a.rstd[m] = rstd;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                float4 xh, o;
                xh.x = x4[v].x * rstd; xh.y = x4[v].y * rstd; xh.z = x4[v].z * rstd; xh.w = x4[v].w * rstd;
                reinterpret_cast<float4 *>(a.u + m * H)[v * G + gl] = xh;
                o.x = fmaxf(fmaf(xh.x, gam[v][0], bet[v][0]), 0.f);
                o.y = fmaxf(fmaf(xh.y, gam[v][1], bet[v][1]), 0.f);
                o.z = fmaxf(fmaf(xh.z, gam[v][2], bet[v][2]), 0.f);
                o.w = fmaxf(fmaf(xh.w, gam[v][3], bet[v][3]), 0.f);
                reinterpret_cast<float4 *>(yrow)[v * G + gl] = o;
            }
        } else {
#pragma unroll
            for (int v = 0; v < VPL; ++v) reinterpret_cast<float4 *>(yrow)[v * G + gl] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

template <int H> __global__ __launch_bounds__(256) void norm_bwd_kernel(NormArgs a)
{
    using Cfg = RowCfg<H>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL, RPW = Cfg::RPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gi = lane / G;
    float gam[VPL][4], bet[VPL][4], dg[VPL][4], dbe[VPL][4], dbi[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gam[v][e] = a.gamma[(v * G + gl) * 4 + e];
            bet[v][e] = a.beta[(v * G + gl) * 4 + e];
            dg[v][e] = dbe[v][e] = dbi[v][e] = 0.f;
        }
    const long total = (long)a.N * a.Rv + 2;                 // rows of dU, + one zero row: the weight-gradient GEMM
                                                             // (A(m) = dU + (m+1) H) and backward-data (rows m, m+1) read it
    const long stride = (long)gridDim.x * 4 * RPW;
    for (long base = (long)blockIdx.x * 4 * RPW; base < total; base += stride) {
        const long row = base + wave * RPW + gi;
        const bool in_range = row < total;
        long nl; int rr;
        row_at(row_base(base, a.Rv), wave * RPW + gi, a.Rv, nl, rr);
        const int n = in_range ? (int)nl : 0;
        const int t = in_range ? rr - 1 : -1;
        const bool valid = in_range && n < a.N && t >= 0 && t < a.Lout;
        const long m = (long)n * a.Rv + (valid ? t : 0);
        const float rstd = valid ? a.rstd[m] : 0.f;
        float xh[VPL][4], gx[VPL][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = x4;
            if (valid) {
                x4 = reinterpret_cast<const float4 *>(a.u + m * H)[v * G + gl];
                g4 = reinterpret_cast<const float4 *>(a.dy + ((long)n * a.Lout + t) * H)[v * G + gl];
            }
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float act = fmaf(xv[e], gam[v][e], bet[v][e]);
                const float g = (valid && act > 0.f) ? gv[e] : 0.f;
                dbe[v][e] += g;
                dg[v][e] = fmaf(g, xv[e], dg[v][e]);
                xh[v][e] = xv[e];
                gx[v][e] = g * gam[v][e];
                s1 += gx[v][e];
                s2 = fmaf(gx[v][e], xv[e], s2);
            }
        }
        s1 = group_sum<G>(s1) * (1.f / H);
        s2 = group_sum<G>(s2) * (1.f / (H - 1));
        if (!in_range) continue;
        float *durow = a.du + row * H;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = rstd * (gx[v][e] - s1 - xh[v][e] * s2);
                dbi[v][e] += o[e];
            }
            reinterpret_cast<float4 *>(durow)[v * G + gl] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    // block-level sum of the 4*RPW lane groups' partials, one row of part[] per block
    __shared__ float red[4 * RPW][3 * H];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (v * G + gl) * 4 + e;
            red[wave * RPW + gi][c] = dg[v][e];
            red[wave * RPW + gi][H + c] = dbe[v][e];
            red[wave * RPW + gi][2 * H + c] = dbi[v][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * H; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * RPW; ++i) t += red[i][c];
        a.part[(long)blockIdx.x * 3 * H + c] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Producers of chunked bf16 planes (gemm_planes.hip, PlanesOperand in common.h) for H = 256 / 512.  A block turns 16
// consecutive signal rows into whole cache lines of the three planes: the row kernels keep their lane layout (a lane
// owns 4 channels of a row), drop the split values into an LDS tile [plane][row][channel] (rows padded by 32 bytes:
// the flush reads the same 32-byte column of different rows), and the flush copies 16-byte pieces so that the
// (16 / s) rows of a phase of a chunk -- contiguous in the plane -- leave through neighbouring lanes.
struct PlaneOut { unsigned short *p; long plane; int sshift; long rts; };
#define CPC_DISPATCH_HP(H, ...)                                 \
    switch (H) {                                                \
    case 256: { constexpr int HH = 256; __VA_ARGS__; } break;  \
    case 512: { constexpr int HH = 512; __VA_ARGS__; } break;  \
    default: break;                                             \
    }

template <int H> struct PlaneTile {
    static constexpr int ROWB = 2 * H + 32;
    static constexpr int BYTES = 3 * 16 * ROWB;
};

__device__ __forceinline__ void split4_planes(const float (&a)[4], uint2 (&w)[3])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    float lo0 = a[0], hi0 = a[1], lo1 = a[2], hi1 = a[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        f2 p0 = {lo0, hi0}, p1 = {lo1, hi1};
        const uint32_t k0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(p0, b2));
        const uint32_t k1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, b2));
        w[t] = make_uint2(k0, k1);
        lo0 -= __uint_as_float(k0 << 16); hi0 -= __uint_as_float(k0 & 0xffff0000u);
        lo1 -= __uint_as_float(k1 << 16); hi1 -= __uint_as_float(k1 & 0xffff0000u);
    }
}

// float4 number f4 (channels 4 f4 .. 4 f4 + 3) of tile row `row16`
template <int H> __device__ __forceinline__ void tile_put(char *tile, int row16, int f4, const float (&a)[4])
{
    uint2 w[3];
    split4_planes(a, w);
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<uint2 *>(tile + (t * 16 + row16) * PlaneTile<H>::ROWB + f4 * 8) = w[t];
}

// the tile holds signal rows G0 .. G0 + 15 (G0 % 16 == 0); all 256 threads
template <int H> __device__ __forceinline__ void tile_flush(const char *tile, const PlaneOut &o, long G0)
{
    const int sh = o.sshift, qn = 16 >> sh;
    for (int u = threadIdx.x; u < (H / 16) * 32; u += 256) {
        const int c = u >> 5, w = u & 31, half = w & 1, k = w >> 1;
        const int phi = k >> (4 - sh), qi = k & (qn - 1), row = (qi << sh) + phi;
        const long chunk = (long)((c << sh) + phi) * o.rts + (G0 >> sh) + qi;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint4 v = *reinterpret_cast<const uint4 *>(tile + (t * 16 + row) * PlaneTile<H>::ROWB + c * 32 + half * 16);
            *reinterpret_cast<uint4 *>(o.p + t * o.plane + chunk * 16 + half * 8) = v;
        }
    }
}

// conv0 + norm + relu -> Y0 as planes.  One block per 64 consecutive rows of Y0 (halo and slack rows included: zeros).
// (n_groups groups of 64 rows, walked by gridDim.x persistent blocks: the 52 parameter registers of a lane are loaded once per block
//  instead of once per 64 rows, and the samples of the next group are requested before the current one is computed)
template <int H> __global__ __launch_bounds__(256) void conv0_fwd_pl_kernel(Conv0Args a, PlaneOut o, long n_groups)
{
    using Cfg = RowCfg<H>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL;
    static_assert(Cfg::RPW == 1, "plane producers: one row per wave pass");
    __shared__ float xs[64 * C0_K];
    __shared__ __attribute__((aligned(16))) char tile[PlaneTile<H>::BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane;

    // three samples per thread and group (64 rows x 10 taps = 640 = 2.5 x 256), requested together
    auto fetch = [&](long grp, float (&v)[3]) {
        const RowBase fq = row_base(grp * 64, a.R0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int i = threadIdx.x + 256 * q;
            v[q] = 0.f;
            if (i < 64 * C0_K && grp < n_groups) {
                const int slot = i / C0_K, j = i - slot * C0_K;
                long n; int rr;
                row_at(fq, slot, a.R0, n, rr);
                const int t = rr - a.halo;
                const int pos = C0_S * t - C0_P + j;
                if (n < a.N && t >= 0 && t < a.L1 && pos >= 0 && pos < a.L0) v[q] = conv0_window(a, n)[pos];
            }
        }
    };
    float xnext[3];
    fetch(blockIdx.x, xnext);
    float wreg[VPL][4][C0_K], breg[VPL][4], gam[VPL][4], bet[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (v * G + gl) * 4 + e;
#pragma unroll
            for (int j = 0; j < C0_K; ++j) wreg[v][e][j] = a.w[c * C0_K + j];
            breg[v][e] = a.b[c];
            gam[v][e] = a.gamma[c];
            bet[v][e] = a.beta[c];
        }
    for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long G0 = grp * 64;
    const RowBase gq = row_base(G0, a.R0);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (threadIdx.x + 256 * q < 64 * C0_K) xs[threadIdx.x + 256 * q] = xnext[q];
    __syncthreads();
    fetch(grp + gridDim.x, xnext);               // (in flight under this group's arithmetic)
    for (int sub = 0; sub < 4; ++sub) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r16 = pass * 4 + wave, slot = sub * 16 + r16;
            long n; int rr;
            row_at(gq, slot, a.R0, n, rr);
            const int t = rr - a.halo;
            const bool valid = n < a.N && t >= 0 && t < a.L1;
            float xr[C0_K];
#pragma unroll
            for (int j = 0; j < C0_K; ++j) xr[j] = xs[C0_K * slot + j];
            float u[VPL][4];
            float sm = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float acc = breg[v][e];
#pragma unroll
                    for (int j = 0; j < C0_K; ++j) acc = fmaf(wreg[v][e][j], xr[j], acc);
                    u[v][e] = acc;
                    sm += acc;
                }
            const float mean = group_sum<G>(sm) * (1.f / H);
            float ss = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u[v][e] -= mean;
                    ss = fmaf(u[v][e], u[v][e], ss);
                }
            const float rstd = rsqrtf(group_sum<G>(ss) * (1.f / (H - 1)) + a.eps);
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = valid ? fmaxf(fmaf(u[v][e] * rstd, gam[v][e], bet[v][e]), 0.f) : 0.f;
                tile_put<H>(tile, r16, v * G + gl, y);
            }
            if (valid && gl == 0) {
                a.stats[(n * a.L1 + t) * 2 + 0] = mean;
                a.stats[(n * a.L1 + t) * 2 + 1] = rstd;
            }
        }
        __syncthreads();
        tile_flush<H>(tile, o, G0 + sub * 16);
        __syncthreads();
    }
    }
}

// Rows of a layer's input planes that no output frame lands in (a sample's halo rows, the slack behind the last sample): zeros.
// (norm_fwd_pl_kernel writes them itself; the product's fused epilogue -- gemm_nt_planes with a PlanesNormOut -- writes frames only.)
__global__ void zero_plane_rows_kernel(PlaneOut o, long total_rows, int N, long rows_next, int halo, int lout, int chunks)
{
    // only the rows that are no frame are enumerated: rows_next - lout per sample (halo in front, the rest behind), then the slack
    const long q = (long)blockIdx.x * blockDim.y + threadIdx.y;
    const long hr = rows_next - lout, in_samples = (long)N * hr;
    long r;
    if (q < in_samples) {
        const long n = q / hr;
        const int j = (int)(q - n * hr);
        r = n * rows_next + (j < halo ? j : lout + j);
    } else {
        r = (long)N * rows_next + (q - in_samples);
        if (r >= total_rows) return;
    }
    const long smask = (1L << o.sshift) - 1;
    for (int i = threadIdx.x; i < 3 * chunks * 2; i += blockDim.x) {
        const int pl = i / (chunks * 2), rem = i - pl * chunks * 2, c = rem >> 1, half = rem & 1;
        const long chunk = (((long)c << o.sshift) + (r & smask)) * o.rts + (r >> o.sshift);
        *reinterpret_cast<uint4 *>(o.p + pl * o.plane + chunk * 16 + half * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ChannelNorm + ReLU of layers 1..3 -> xhat (f32, in place) and the next layer's input as planes; tiles of 16 rows of Y
template <int H> __global__ __launch_bounds__(256) void norm_fwd_pl_kernel(NormArgs a, PlaneOut o, long n_tiles)
{
    using Cfg = RowCfg<H>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL;
    static_assert(Cfg::RPW == 1, "plane producers: one row per wave pass");
    __shared__ __attribute__((aligned(16))) char tile[PlaneTile<H>::BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane;
    float gam[VPL][4], bet[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gam[v][e] = a.gamma[(v * G + gl) * 4 + e];
            bet[v][e] = a.beta[(v * G + gl) * 4 + e];
        }
    for (long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const RowBase tq = row_base(tl * 16, a.Rnext);
        // the four rows of this wave are requested together: a row is a load -> use chain
        float4 x4[4][VPL];
        long mrow[4];
        bool ok[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            long n; int rr;
            row_at(tq, pass * 4 + wave, a.Rnext, n, rr);
            const int t = rr - a.halo;
            ok[pass] = n < a.N && t >= 0 && t < a.Lout;
            mrow[pass] = n * a.Rv + (ok[pass] ? t : 0);
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                x4[pass][v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[pass]) x4[pass][v] = norm_in4(a, mrow[pass], H, v * G + gl);
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            float sm = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) sm += (x4[pass][v].x + x4[pass][v].y) + (x4[pass][v].z + x4[pass][v].w);
            const float mean = group_sum<G>(sm) * (1.f / H);
            float ss = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                float4 &q = x4[pass][v];
                q.x -= mean; q.y -= mean; q.z -= mean; q.w -= mean;
                ss = fmaf(q.x, q.x, ss); ss = fmaf(q.y, q.y, ss); ss = fmaf(q.z, q.z, ss); ss = fmaf(q.w, q.w, ss);
            }
            const float rstd = rsqrtf(group_sum<G>(ss) * (1.f / (H - 1)) + a.eps);
            if (ok[pass] && gl == 0) a.rstd[mrow[pass]] = rstd;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const float4 q = x4[pass][v];
                const float xh[4] = {q.x * rstd, q.y * rstd, q.z * rstd, q.w * rstd};
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = ok[pass] ? fmaxf(fmaf(xh[e], gam[v][e], bet[v][e]), 0.f) : 0.f;
                if (ok[pass]) reinterpret_cast<float4 *>(a.u + mrow[pass] * H)[v * G + gl] = make_float4(xh[0], xh[1], xh[2], xh[3]);
                tile_put<H>(tile, pass * 4 + wave, v * G + gl, y);
            }
        }
        __syncthreads();
        tile_flush<H>(tile, o, tl * 16);
        __syncthreads();
    }
}

// backward of ChannelNorm + ReLU -> dU as planes (rows shifted by one, zero rows at the sample borders and in the slack)
template <int H> __global__ __launch_bounds__(256) void norm_bwd_pl_kernel(NormArgs a, PlaneOut o, long n_tiles)
{
    using Cfg = RowCfg<H>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL;
    static_assert(Cfg::RPW == 1, "plane producers: one row per wave pass");
    __shared__ __attribute__((aligned(16))) char tile[PlaneTile<H>::BYTES > 4 * 3 * H * 4 ? PlaneTile<H>::BYTES : 4 * 3 * H * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane;
    float gam[VPL][4], bet[VPL][4], dg[VPL][4], dbe[VPL][4], dbi[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gam[v][e] = a.gamma[(v * G + gl) * 4 + e];
            bet[v][e] = a.beta[(v * G + gl) * 4 + e];
            dg[v][e] = dbe[v][e] = dbi[v][e] = 0.f;
        }
    for (long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const RowBase tq = row_base(tl * 16, a.Rv);
        float4 x4[4][VPL], g4[4][VPL];
        float rs[4];
        bool ok[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            long n; int rr;
            row_at(tq, pass * 4 + wave, a.Rv, n, rr);
            const int t = rr - 1;
            ok[pass] = n < a.N && t >= 0 && t < a.Lout;
            const long m = n * a.Rv + (ok[pass] ? t : 0);
            rs[pass] = ok[pass] ? a.rstd[m] : 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                x4[pass][v] = make_float4(0.f, 0.f, 0.f, 0.f);
                g4[pass][v] = x4[pass][v];
                if (ok[pass]) {
                    x4[pass][v] = reinterpret_cast<const float4 *>(a.u + m * H)[v * G + gl];
                    g4[pass][v] = reinterpret_cast<const float4 *>(a.dy + (n * a.Lout + t) * H)[v * G + gl];
                }
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            float xh[VPL][4], gx[VPL][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const float xv[4] = {x4[pass][v].x, x4[pass][v].y, x4[pass][v].z, x4[pass][v].w};
                const float gv[4] = {g4[pass][v].x, g4[pass][v].y, g4[pass][v].z, g4[pass][v].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float act = fmaf(xv[e], gam[v][e], bet[v][e]);
                    const float g = (ok[pass] && act > 0.f) ? gv[e] : 0.f;
                    dbe[v][e] += g;
                    dg[v][e] = fmaf(g, xv[e], dg[v][e]);
                    xh[v][e] = xv[e];
                    gx[v][e] = g * gam[v][e];
                    s1 += gx[v][e];
                    s2 = fmaf(gx[v][e], xv[e], s2);
                }
            }
            s1 = group_sum<G>(s1) * (1.f / H);
            s2 = group_sum<G>(s2) * (1.f / (H - 1));
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                float du[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    du[e] = rs[pass] * (gx[v][e] - s1 - xh[v][e] * s2);
                    dbi[v][e] += du[e];
                }
                tile_put<H>(tile, pass * 4 + wave, v * G + gl, du);
            }
        }
        __syncthreads();
        tile_flush<H>(tile, o, tl * 16);
        __syncthreads();
    }
    // block-level sum of the four waves' partials, one row of part[] per block (the tile's LDS is free now)
    float *red = reinterpret_cast<float *>(tile);            // [4][3 H]
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (v * G + gl) * 4 + e;
            red[wave * 3 * H + c] = dg[v][e];
            red[wave * 3 * H + H + c] = dbe[v][e];
            red[wave * 3 * H + 2 * H + c] = dbi[v][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * H; c += 256)
        a.part[(long)blockIdx.x * 3 * H + c] = (red[c] + red[3 * H + c]) + (red[6 * H + c] + red[9 * H + c]);
}

// Backward data, boundary rows.  The plane-fed product covers the virtual rows t_hi = 0 .. L_out - 1 of a sample; what
// is left is t_hi = L_out, whose own dU row is the zero border row, so only dU(L_out - 1) contributes:
//   dY[n][L_in - p + j][ci] = sum_co dU(n, L_out - 1)[co] * W[co][ci][j + s],  j < p
// -- N * p rows of H outputs, a dot product of length H each: VALU work on the planes (p0 + p1 + p2 is the f32 value,
// exactly).  One block per (sample, j); thread = ci (+ 256 per pass); the weights come from the backward-data planes
// (row j*H + ci, the 16 co of chunk c at K step 2c: consecutive ci are consecutive 32-byte chunks).
__device__ __forceinline__ float bf16_up(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__global__ __launch_bounds__(256) void bwd_edge_kernel(const unsigned short *dup, long duplane, long durts, const unsigned short *wdp,
                                                       long wplane, float *dy, int H, int s, int p, int Rv, int Lout, int Lin)
{
    // one workgroup per (sample n, phase j, 64 output channels); thread = (channel, quarter of the H / 16 chunks of the
    // reduction): every load of a thread is issued before the first is used (a thread that walks all of K alone spends the
    // kernel waiting for one dependent load after the other: 12 us per launch for 67 MFLOP), the four partial sums of a
    // channel meet in LDS and are added in a fixed order
    extern __shared__ float du[];                           // [H] + [4][64]
    float *red = du + H;
    const int quarters = H / 64;
    const int cq = blockIdx.x % quarters, nj = blockIdx.x / quarters;
    const int n = nj / p, j = nj - n * p;
    const long R = (long)n * Rv + Lout;                      // dU(L_out - 1) is stored one row down
    for (int c = threadIdx.x; c < H; c += 256) {
        const long off = ((long)(c >> 4) * durts + R) * 16 + (c & 15);
        du[c] = (bf16_up(dup[off]) + bf16_up(dup[duplane + off])) + bf16_up(dup[2 * duplane + off]);
    }
    __syncthreads();
    const int rows = s * H;
    const int ci = cq * 64 + (threadIdx.x & 63), kq = threadIdx.x >> 6;
    const int per = H / 64;                                  // chunks per thread (H / 16 chunks, 4 quarters)
    float acc = 0.f;
    for (int c0 = 0; c0 < per; c0 += 4) {
        uint4 w[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = kq * per + c0 + u;
            const long off = ((long)(2 * c) * rows + (long)j * H + ci) * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                w[u][2 * pl] = *reinterpret_cast<const uint4 *>(wdp + pl * wplane + off);
                w[u][2 * pl + 1] = *reinterpret_cast<const uint4 *>(wdp + pl * wplane + off + 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = kq * per + c0 + u;
            unsigned short w0[16], w1[16], w2[16];
            *reinterpret_cast<uint4 *>(w0) = w[u][0]; *reinterpret_cast<uint4 *>(w0 + 8) = w[u][1];
            *reinterpret_cast<uint4 *>(w1) = w[u][2]; *reinterpret_cast<uint4 *>(w1 + 8) = w[u][3];
            *reinterpret_cast<uint4 *>(w2) = w[u][4]; *reinterpret_cast<uint4 *>(w2 + 8) = w[u][5];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc = fmaf(du[c * 16 + e], (bf16_up(w0[e]) + bf16_up(w1[e])) + bf16_up(w2[e]), acc);
        }
    }
    red[kq * 64 + (threadIdx.x & 63)] = acc;
    __syncthreads();
    if (kq == 0) {
        const int l = threadIdx.x;
        dy[((long)n * Lin + Lin - p + j) * H + ci] = ((red[l] + red[64 + l]) + red[128 + l]) + red[192 + l];
    }
}

// Conv1d weights as the plane-fed GEMMs' B operands (gemm_planes.hip), one launch for all layers:
// blockIdx.z = 0: forward operand, rows co, K order (chunk c of ci, tap 0, s, 1, s + 1, ...);
// blockIdx.z = 1: backward-data operand, rows (phase j, ci), K order (chunk c of co, dU row t_hi - 1 then t_hi).
// Both are [K / 16][rows][16] per plane, planes H * H * k elements apart.
struct PermutePlanes { const float *w[4]; unsigned short *wf[4]; unsigned short *wd[4]; int k[4]; int s[4]; int H; };
__global__ void permute_conv_planes_kernel(PermutePlanes a)
{
    const int li = blockIdx.y, H = a.H, k = a.k[li], s = a.s[li];
    const float *w = a.w[li];
    const long total = (long)H * H * k;
    unsigned short *dst = blockIdx.z == 0 ? a.wf[li] : a.wd[li];
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 15);
        float v;
        if (blockIdx.z == 0) {
            const int co = (int)((idx >> 4) % H), ks = (int)((idx >> 4) / H);
            const int c = ks / k, jj = ks - c * k;
            const int j = k > 1 ? (jj >> 1) + (jj & 1) * s : 0;
            v = w[((long)co * H + c * 16 + e) * k + j];
        } else {
            const int rows = s * H;
            const int n = (int)((idx >> 4) % rows), ks = (int)((idx >> 4) / rows);
            const int c = ks >> 1, first = !(ks & 1);
            const int jph = n / H, ci = n - jph * H;
            v = w[((long)(c * 16 + e) * H + ci) * k + (first ? jph + s : jph)];
        }
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f2 pr = {v, 0.f};
            const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, b2));
            dst[t * total + idx] = (unsigned short)(pk & 0xffffu);
            v -= __uint_as_float(pk << 16);
        }
    }
}

// Every Conv1d weight re-layout of a step in one launch (blockIdx.y = layer - 1, blockIdx.z = 0 forward operand
// wf[co][j*H + ci], 1 backward-data operand bd[j][ci][kk] -- see rowops.hip permute_conv_*), H x H x k each.
struct PermuteAll { const float *w[4]; float *wf[4]; float *wd[4]; int k[4]; int s[4]; int H; };
__global__ void permute_conv_all_kernel(PermuteAll a)
{
    const int li = blockIdx.y, H = a.H, k = a.k[li], s = a.s[li];
    const float *w = a.w[li];
    const long total = (long)H * H * k;
    if (blockIdx.z == 0) {
        float *wr = a.wf[li];
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int co = (int)(idx / ((long)H * k));
            const int rem = (int)(idx - (long)co * H * k);
            const int j = rem / H, ci = rem - j * H;
            wr[idx] = w[((long)co * H + ci) * k + j];
        }
    } else {
        float *bd = a.wd[li];
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int kk = (int)(idx % (2 * H));
            const long r = idx / (2 * H);
            const int ci = (int)(r % H);
            const int j = (int)(r / H);
            const int co = kk < H ? kk : kk - H;
            const int tap = kk < H ? j + s : j;
            bd[idx] = w[((long)co * H + ci) * k + tap];
        }
    }
}

struct ZeroTails { float *p[4]; int n[4]; };
__global__ void zero_tails_kernel(ZeroTails z)
{
    for (int i = threadIdx.x; i < z.n[blockIdx.x]; i += blockDim.x) z.p[blockIdx.x][i] = 0.f;
}

// ------------------------------------------------------------------------------------------------
struct EncLayout {
    int H, N;
    int L[6];        // L[0] = samples, L[i+1] = frames after layer i
    int Rv[5];       // virtual GEMM rows per sample of layer i (i >= 1): L[i+1] + 2
    int R[4];        // rows per sample of Y_i (i = 0..3): s_{i+1} * Rv[i+1]
    // saved
    float *Y[4];
    float *Xh[5];    // index 1..4
    float *rstd[5];  // index 1..4
    float *stats0;
    float *Wd[5];    // index 1..4: backward-data operand of layer i, [s*H][2H] (laid out by the forward pass)
    size_t saved_bytes;
    // scratch
    float *Wf[5];    // index 1..4: forward GEMM operand of layer i, [H][k*H]
    float *dYa, *dYb, *dU, *part, *sums, *cs, *tn;
    size_t tn_bytes;
    float *part_l[5], *cs_l[5], *tn_l[5];      // per-layer copies for the deferred backward
    size_t tn_l_bytes[5];
    size_t scratch_bytes;
    // plane-fed GEMMs (H = 256, 512): Y_i, dU and the weights live as chunked bf16 planes instead of f32
    int planes;
    unsigned short *Yp[4];  long Yplane[4], Yrts[4];     // saved: input of layer i + 1, s_{i+1} phases
    unsigned short *Wdp[5];                               // saved: backward-data operand of layer i
    unsigned short *Wfp[5];                               // scratch: forward operand of layer i
    unsigned short *dUp;    long dUplane, dUrows;         // scratch: dU of the current layer (largest: layer 1)
};

constexpr int NORM_BWD_BLOCKS = 2048;   // 8 per CU: the row loop is a load -> use chain, only occupancy hides its latency
constexpr int CONV0_BWD_BLOCKS = 768;

static bool supported_hidden(int H) { return H == 32 || H == 64 || H == 128 || H == 256 || H == 512; }
static int log2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
// the encoder's GEMMs run on pre-split planes when the hidden size is made of 256-column tiles (CPC_NO_PLANES=1: A/B switch)
static bool use_planes(int H)
{
    static const bool off = getenv("CPC_NO_PLANES") != nullptr;
    return !off && H % 256 == 0;
}

static int enc_layout(EncLayout &e, int N, int length, int H, void *saved, void *scratch)
{
    CPC_REQUIRE(supported_hidden(H), "encoder: hidden size %d not supported (32, 64, 128, 256, 512)", H);
    CPC_REQUIRE(N > 0 && length >= 400, "encoder: need n_windows > 0 and length >= 400 (got %d, %d)", N, length);
    e.H = H; e.N = N;
    e.L[0] = length;
    for (int i = 0; i < 5; ++i) {
        e.L[i + 1] = (e.L[i] + 2 * kConv[i].p - kConv[i].k) / kConv[i].s + 1;
        CPC_REQUIRE(e.L[i + 1] >= 1, "encoder: input too short");
    }
    for (int i = 1; i < 5; ++i) e.Rv[i] = e.L[i + 1] + 2;
    e.Rv[0] = 0;
    for (int i = 0; i < 4; ++i) e.R[i] = kConv[i + 1].s * e.Rv[i + 1];

    e.planes = use_planes(H) ? 1 : 0;
    Carver sv(saved);
    for (int i = 0; i < 4; ++i) {
        e.Y[i] = nullptr; e.Yp[i] = nullptr;
        if (!e.planes) { e.Y[i] = sv.take<float>(((size_t)N * e.R[i] + kConv[i + 1].k) * H); continue; }   // + slack rows
        // rows of a phase: the weight-gradient product reads whole 32-row steps of virtual rows (+ a tap beyond)
        const int s = kConv[i + 1].s;
        e.Yrts[i] = (cdiv((long)N * e.Rv[i + 1], 32) * 32 + 4 + 15) / 16 * 16;
        e.Yplane[i] = (long)(H / 16) * s * e.Yrts[i] * 16;
        e.Yp[i] = sv.take<unsigned short>((size_t)3 * e.Yplane[i]);
    }
    e.Xh[0] = nullptr; e.rstd[0] = nullptr;
    for (int i = 1; i < 5; ++i) {
        e.Xh[i] = sv.take<float>((size_t)N * e.Rv[i] * H);
        e.rstd[i] = sv.take<float>((size_t)N * e.Rv[i]);
    }
    e.stats0 = sv.take<float>((size_t)N * e.L[1] * 2);
    e.Wd[0] = nullptr;
    for (int i = 1; i < 5; ++i) {
        e.Wd[i] = nullptr; e.Wdp[i] = nullptr;
        if (e.planes) e.Wdp[i] = sv.take<unsigned short>((size_t)3 * kConv[i].k * H * H);
        else e.Wd[i] = sv.take<float>((size_t)kConv[i].k * H * H);
    }
    e.saved_bytes = sv.used();

    Carver sc(scratch);
    e.Wf[0] = nullptr;
    for (int i = 1; i < 5; ++i) {
        e.Wf[i] = nullptr; e.Wfp[i] = nullptr;
        if (e.planes) e.Wfp[i] = sc.take<unsigned short>((size_t)3 * kConv[i].k * H * H);
        else e.Wf[i] = sc.take<float>((size_t)kConv[i].k * H * H);
    }
    e.dYa = sc.take<float>((size_t)N * e.L[1] * H);
    e.dYb = sc.take<float>((size_t)N * e.L[2] * H);
    e.dU = nullptr; e.dUp = nullptr; e.dUrows = 0; e.dUplane = 0;
    if (e.planes) {
        e.dUrows = (cdiv((long)N * e.Rv[1], 32) * 32 + 4 + 15) / 16 * 16;
        e.dUplane = (long)(H / 16) * e.dUrows * 16;
        e.dUp = sc.take<unsigned short>((size_t)3 * e.dUplane);
    } else {
        e.dU = sc.take<float>(((size_t)N * e.Rv[1] + 2) * H);
    }
    const size_t part_floats = std::max((size_t)CONV0_BWD_BLOCKS * 13 * H, (size_t)NORM_BWD_BLOCKS * 3 * H);
    e.part = sc.take<float>(part_floats);
    e.sums = sc.take<float>((size_t)13 * H);
    e.cs = sc.take<float>(colsum_split_scratch_bytes(13 * H) / sizeof(float));
    e.tn_bytes = 0;
    for (int i = 1; i < 5; ++i) {
        e.tn_bytes = std::max(e.tn_bytes, std::max(gemm_tn_scratch_bytes(H, kConv[i].k * H, (long)N * e.Rv[i]),
                                                    gemm_nt_scratch_bytes((long)N * e.Rv[i], H, kConv[i].k * H)));   // + forward K split
        if (e.planes)
            e.tn_bytes = std::max(e.tn_bytes, std::max(gemm_tn_planes_scratch_bytes(H, kConv[i].k * H, (long)N * e.Rv[i]),
                                                        gemm_nt_planes_scratch_bytes((long)N * e.L[i + 1], H, kConv[i].k * H, (long)N * e.Rv[i])));
    }
    e.tn = sc.take<float>(e.tn_bytes / sizeof(float));
    // deferred form of the backward (encoder_backward, defer_small): each layer's partial sums and weight-gradient slabs stay
    // where they are until the side stream has summed them, so every layer has its own
    for (int i = 1; i < 5; ++i) {
        e.part_l[i] = nullptr; e.cs_l[i] = nullptr; e.tn_l[i] = nullptr; e.tn_l_bytes[i] = 0;
        if (e.planes) {
            e.part_l[i] = sc.take<float>((size_t)NORM_BWD_BLOCKS * 3 * H);
            e.cs_l[i] = sc.take<float>(colsum_split_scratch_bytes(3 * H) / sizeof(float));
            e.tn_l_bytes[i] = gemm_tn_planes_scratch_bytes(H, kConv[i].k * H, (long)N * e.Rv[i]);
            e.tn_l[i] = sc.take<float>(e.tn_l_bytes[i] / sizeof(float));
        }
    }
    e.scratch_bytes = sc.used();
    return CPC_OK;
}

// x2 / n_first: windows n_first .. N - 1 come from x2 (nullptr: all N from x)
static int encoder_forward(const float *x, const float *const *prm, float *z, void *saved, void *scratch, int N,
                           int length, int H, float eps, hipStream_t st, const float *x2 = nullptr, int n_first = 0)
{
    CPC_REQUIRE(x2 == nullptr || (n_first > 0 && n_first < N), "encoder: the first batch must hold 1 .. n_windows - 1 windows (got %d of %d)", n_first, N);
    EncLayout e;
    CPC_TRY(enc_layout(e, N, length, H, saved, scratch));

    // layer 0: fused conv + norm + relu straight from the waveform
    Conv0Args c0{};
    c0.x = x; c0.x2 = x2; c0.n_split = x2 != nullptr ? n_first : N;
    c0.w = prm[0]; c0.b = prm[1]; c0.gamma = prm[2]; c0.beta = prm[3];
    c0.y = e.Y[0]; c0.stats = e.stats0;
    c0.N = N; c0.L0 = e.L[0]; c0.L1 = e.L[1]; c0.R0 = e.R[0]; c0.halo = kConv[1].p; c0.eps = eps;
    c0.tiles_per_sample = (int)cdiv(e.L[1], C0_TB);
    c0.n_tiles = N * c0.tiles_per_sample;
    for (int i = 1; i < 5; ++i) CPC_REQUIRE(kConv[i].k == 2 * kConv[i].s, "encoder: layer %d needs kernel == 2 * stride", i);
    if (e.planes) {
        ProfScope prof(PROF_CONV0_FWD, st);
        const PlaneOut o{e.Yp[0], e.Yplane[0], log2i(kConv[1].s), e.Yrts[0]};
        const long rows = (long)kConv[1].s * e.Yrts[0];               // every row of the planes is written (halo, slack: zeros)
        // (4096 persistent blocks: 187 -> 176 us at CPC-small against one block per group; 1024 / 2048: 179-180)
        const long groups = rows / 64;
        const unsigned blocks = (unsigned)std::min<long>(groups, 4096);
        CPC_DISPATCH_HP(H, hipLaunchKernelGGL(conv0_fwd_pl_kernel<HH>, dim3(blocks), dim3(256), 0, st, c0, o, groups));
    } else {
        ProfScope prof(PROF_CONV0_FWD, st);
        CPC_DISPATCH_H(H, hipLaunchKernelGGL(conv0_fwd_kernel<HH>, dim3(c0.n_tiles), dim3(256), 0, st, c0));
    }
    CPC_CHECK_LAUNCH("conv0_fwd_kernel");

    if (e.planes) {
        PermutePlanes pa{};
        pa.H = H;
        for (int i = 1; i < 5; ++i) {
            pa.w[i - 1] = prm[4 * i]; pa.wf[i - 1] = e.Wfp[i]; pa.wd[i - 1] = e.Wdp[i]; pa.k[i - 1] = kConv[i].k; pa.s[i - 1] = kConv[i].s;
        }
        hipLaunchKernelGGL(permute_conv_planes_kernel, dim3(256, 4, 2), dim3(256), 0, st, pa);
        CPC_CHECK_LAUNCH("permute_conv_planes_kernel");
    } else {
        // slack rows after each Y_{i-1}: read by junk GEMM rows and by the weight-gradient GEMM -> must be finite
        ZeroTails zt{};
        for (int i = 1; i < 5; ++i) { zt.p[i - 1] = e.Y[i - 1] + (size_t)N * e.R[i - 1] * H; zt.n[i - 1] = kConv[i].k * H; }
        hipLaunchKernelGGL(zero_tails_kernel, dim3(4), dim3(256), 0, st, zt);
        CPC_CHECK_LAUNCH("zero_tails_kernel");
        PermuteAll pa{};
        pa.H = H;
        for (int i = 1; i < 5; ++i) {
            pa.w[i - 1] = prm[4 * i]; pa.wf[i - 1] = e.Wf[i]; pa.wd[i - 1] = e.Wd[i]; pa.k[i - 1] = kConv[i].k; pa.s[i - 1] = kConv[i].s;
        }
        hipLaunchKernelGGL(permute_conv_all_kernel, dim3(256, 4, 2), dim3(256), 0, st, pa);
        CPC_CHECK_LAUNCH("permute_conv_all_kernel");
    }
    for (int i = 1; i < 5; ++i) {
        const int k = kConv[i].k, s = kConv[i].s;
        int left_slabs = 0;
        if (e.planes) {
            // GEMM row m = (sample, frame) over the valid frames only; output row = virtual row of Xh
            const PlanesOperand A{e.Yp[i - 1], e.Yplane[i - 1], log2i(k), log2i(s), e.Yrts[i - 1], e.L[i + 1], (long)e.Rv[i]};
            const PlanesOperand B{e.Wfp[i], (long)k * H * H, 0, 0, (long)H, 0, 0};
            RowMap out{};
            out.enabled = 1; out.rv = e.L[i + 1]; out.out_stride = 1; out.out_off = 0; out.l_max = e.Rv[i]; out.rows_out = e.Rv[i];
            out.splitk_scratch = e.tn; out.splitk_bytes = e.tn_bytes;
            if (i < 4 && gemm_nt_planes_norm_ok((long)N * e.L[i + 1], H, k * H)) {
                // ChannelNorm + ReLU + split in the product's epilogue (hidden 256, products that do not split K: conv1, conv2 at the
                // training shapes): the tile holds whole rows, so y never goes to memory and back
                const PlaneOut o{e.Yp[i], e.Yplane[i], log2i(kConv[i + 1].s), e.Yrts[i]};
                const PlanesNormOut nf{prm[4 * i + 2], prm[4 * i + 3], eps, e.rstd[i], o.p, o.plane, o.sshift, o.rts, (long)e.R[i], kConv[i + 1].p};
                const long rows = (long)kConv[i + 1].s * e.Yrts[i];
                const long zrows = (long)N * (e.R[i] - e.L[i + 1]) + (rows - (long)N * e.R[i]);      // rows that are no frame
                hipLaunchKernelGGL(zero_plane_rows_kernel, dim3((unsigned)cdiv(zrows, 4)), dim3(64, 4), 0, st, o, rows, N, (long)e.R[i],
                                   kConv[i + 1].p, e.L[i + 1], H / 16);
                CPC_CHECK_LAUNCH("zero_plane_rows_kernel");
                CPC_TRY(gemm_nt_planes(A, B, e.Xh[i], H, prm[4 * i + 1], (long)N * e.L[i + 1], H, k * H, out, st, &nf));
                continue;
            }
            // (a K split leaves its partial products where they are: the norm kernel below sums them as it reads its rows)
            CPC_TRY(gemm_nt_planes(A, B, e.Xh[i], H, prm[4 * i + 1], (long)N * e.L[i + 1], H, k * H, out, st, nullptr, &left_slabs));
        } else {
            RowMap vrows{};                                  // output rows = virtual rows; rows t >= L of a sample are junk
            vrows.seg_rows = e.Rv[i]; vrows.seg_valid = e.L[i + 1];
            vrows.splitk_scratch = e.tn; vrows.splitk_bytes = e.tn_bytes;
            CPC_TRY(gemm_nt(e.Y[i - 1], (long)s * H, e.Wf[i], (long)k * H, e.Xh[i], H, prm[4 * i + 1], (long)N * e.Rv[i], H,
                            k * H, vrows, st));
        }
        NormArgs na{};
        na.u = e.Xh[i]; na.gamma = prm[4 * i + 2]; na.beta = prm[4 * i + 3]; na.rstd = e.rstd[i];
        na.N = N; na.Lout = e.L[i + 1]; na.Rv = e.Rv[i]; na.eps = eps;
        if (left_slabs > 0) { na.slabs = e.tn; na.nslabs = left_slabs; na.slab_stride = (long)N * e.Rv[i] * H; }
        if (i < 4) { na.y = e.Y[i]; na.Rnext = e.R[i]; na.halo = kConv[i + 1].p; }
        else { na.y = z; na.Rnext = e.L[5]; na.halo = 0; }
        if (e.planes && i < 4) {
            const PlaneOut o{e.Yp[i], e.Yplane[i], log2i(kConv[i + 1].s), e.Yrts[i]};
            const long tiles = (long)kConv[i + 1].s * e.Yrts[i] / 16;
            CPC_DISPATCH_HP(H, hipLaunchKernelGGL(norm_fwd_pl_kernel<HH>, dim3((unsigned)std::min<long>(tiles, 2048)), dim3(256), 0, st, na, o, tiles));
        } else {
            const long rows = (long)N * na.Rnext;
            const int rpb = 4 * (64 / std::min(64, H / 4));
            const int blocks = (int)std::min<long>(cdiv(rows, rpb), 4096);
            CPC_DISPATCH_H(H, hipLaunchKernelGGL(norm_fwd_kernel<HH>, dim3(blocks), dim3(256), 0, st, na));
        }
        CPC_CHECK_LAUNCH("norm_fwd_kernel");
    }
    return CPC_OK;
}

// defer_small: the passes of layers 1-4 that only produce parameter gradients from what a big kernel has left behind -- the two-stage
// column sums of dgamma / dbeta / dbias and the sum of the weight-gradient product's K-split slabs, eight to ten launches of 5-10 us
// that nothing on `st` needs -- run on the library's side stream (side_tail_*), each layer with buffers of its own
static int encoder_backward(const float *x, const float *const *prm, const float *dz, void *saved, void *scratch,
                            float *const *grads, int N, int length, int H, float eps, hipStream_t st, bool defer_small = false,
                            const float *x2 = nullptr, int n_first = 0)
{
    CPC_REQUIRE(x2 == nullptr || (n_first > 0 && n_first < N), "encoder: the first batch must hold 1 .. n_windows - 1 windows (got %d of %d)", n_first, N);
    EncLayout e;
    CPC_TRY(enc_layout(e, N, length, H, saved, scratch));

    const float *dy = dz;                  // [N][L[i+1]][H] of the current layer
    int left[5] = {0, 0, 0, 0, 0};         // deferred form: K-split slabs of the layers' weight-gradient products still to be summed
    for (int i = 4; i >= 1; --i) {
        const int k = kConv[i].k, s = kConv[i].s, p = kConv[i].p;
        // norm + relu backward -> dU_i (shifted rows), partial sums for dgamma, dbeta, dbias
        NormArgs na{};
        na.u = e.Xh[i]; na.gamma = prm[4 * i + 2]; na.beta = prm[4 * i + 3]; na.rstd = e.rstd[i];
        na.N = N; na.Lout = e.L[i + 1]; na.Rv = e.Rv[i]; na.eps = eps;
        na.dy = dy; na.du = e.dU; na.part = e.part;
        float *dprev = (i % 2 == 0) ? e.dYb : e.dYa;         // i=4 -> dYb, 3 -> dYa, 2 -> dYb, 1 -> dYa
        const bool side = defer_small && e.planes;
        if (side) na.part = e.part_l[i];
        if (e.planes) {
            // dU of layer i as planes: rows of the layer, + the zero rows the weight-gradient product's last step reads
            const long durows = (cdiv((long)N * e.Rv[i], 32) * 32 + 4 + 15) / 16 * 16;
            const PlaneOut o{e.dUp, e.dUplane, 0, e.dUrows};
            CPC_DISPATCH_HP(H, hipLaunchKernelGGL(norm_bwd_pl_kernel<HH>, dim3(NORM_BWD_BLOCKS), dim3(256), 0, st, na, o, durows / 16));
            CPC_CHECK_LAUNCH("norm_bwd_pl_kernel");
            if (!side) CPC_TRY(colsum_split(na.part, NORM_BWD_BLOCKS, 3L * H, 3 * H, grads[4 * i + 2], grads[4 * i + 3], grads[4 * i + 1], H, e.cs, st));
            // weight gradient over the virtual rows (dU is zero on a sample's border rows): dW[co][j*H+ci] = sum_m dU(m+1)[co] Y(m s + j)[ci]
            const PlanesTNOperand TA{e.dUp, e.dUplane, 0, e.dUrows, 1, H};
            const PlanesTNOperand TB{e.Yp[i - 1], e.Yplane[i - 1], log2i(s), e.Yrts[i - 1], 0, H};
            CPC_TRY(gemm_tn_planes(TA, TB, grads[4 * i], 0, H, k * H, (long)N * e.Rv[i], side ? e.tn_l[i] : e.tn, side ? e.tn_l_bytes[i] : e.tn_bytes,
                                   H, k, st, side ? &left[i] : nullptr));
            if (side && i == 1) {
                // everything the layers have left behind -- their partial sums and K-split slabs, each in a buffer of its own -- is
                // finished on the side stream from here on, under the last backward-data product and conv0's backward.  ONE fork: an
                // event on the caller's stream costs it ~6 us of idle time (one per pass made the passes' move a wash)
                hipStream_t wst = st;
                CPC_TRY(side_tail_begin(st, &wst));
                for (int j = 4; j >= 1; --j) {
                    CPC_TRY(colsum_split(e.part_l[j], NORM_BWD_BLOCKS, 3L * H, 3 * H, grads[4 * j + 2], grads[4 * j + 3], grads[4 * j + 1], H, e.cs_l[j], wst));
                    if (left[j] > 0) CPC_TRY(planes_tn_reduce(e.tn_l[j], left[j], H, kConv[j].k * H, grads[4 * j], 0, H, kConv[j].k, wst));
                }
                CPC_TRY(side_tail_end());
            }
            // backward data.  Lengths that divide by the stride (training windows): rows t_hi = 0 .. L_out - 1 of every sample
            // in whole tiles, and a small kernel for the boundary row t_hi = L_out; otherwise all L_out + 1 rows in the product
            // (the trailing input rows no output frame reads get a zero gradient)
            const PlanesOperand B{e.Wdp[i], (long)k * H * H, 0, 0, (long)s * H, 0, 0};
            const bool even = e.L[i] == s * e.L[i + 1];
            const int rows = even ? e.L[i + 1] : e.L[i + 1] + 1;
            if (!even) CPC_CHECK_HIP(hipMemsetAsync(dprev, 0, sizeof(float) * (size_t)N * e.L[i] * H, st));
            RowMap map{};
            map.enabled = 1; map.rv = rows; map.out_stride = s; map.out_off = -p;
            map.l_max = e.L[i]; map.rows_out = e.L[i]; map.col_rows = H;
            const PlanesOperand A{e.dUp, e.dUplane, 1, 0, e.dUrows, rows, (long)e.Rv[i]};
            CPC_TRY(gemm_nt_planes(A, B, dprev, H, nullptr, (long)N * rows, s * H, 2 * H, map, st));
            if (even) {
                hipLaunchKernelGGL(bwd_edge_kernel, dim3((unsigned)(N * p * (H / 64))), dim3(256), sizeof(float) * (H + 256), st, e.dUp, e.dUplane, e.dUrows,
                                   e.Wdp[i], (long)k * H * H, dprev, H, s, p, e.Rv[i], e.L[i + 1], e.L[i]);
                CPC_CHECK_LAUNCH("bwd_edge_kernel");
            }
            dy = dprev;
            continue;
        }
        CPC_DISPATCH_H(H, hipLaunchKernelGGL(norm_bwd_kernel<HH>, dim3(NORM_BWD_BLOCKS), dim3(256), 0, st, na));
        CPC_CHECK_LAUNCH("norm_bwd_kernel");
        // part[block][dgamma | dbeta | dbias] -> the three gradients
        CPC_TRY(colsum_split(e.part, NORM_BWD_BLOCKS, 3L * H, 3 * H, grads[4 * i + 2], grads[4 * i + 3], grads[4 * i + 1], H, e.cs, st));

        // weight gradient: dW[co][j*H+ci] = sum_m dU(m)[co] * Y_{i-1}(m)[j*H+ci]
        CPC_TRY(gemm_tn(e.dU + H, H, e.Y[i - 1], (long)s * H, grads[4 * i], 0, H, k * H, (long)N * e.Rv[i], e.tn,
                        e.tn_bytes, H, k, st));

        // backward data: dY_{i-1}[n][t_hi*s + j - p] = [dU(t_hi-1), dU(t_hi)] . Bd[j] for the s phases j -- ONE GEMM
        // with the phases side by side in N (Bd is [s*H][2H]); output column j*H + ci of virtual row t_hi is
        // element ci of data row t_hi*s - p + j, i.e. the s*H outputs of a row are contiguous in dY_{i-1}
        // (Bd = e.Wd[i] was laid out by the forward pass)
        {
            RowMap map{};
            map.enabled = 1; map.rv = e.Rv[i]; map.out_stride = s; map.out_off = -p;
            map.l_max = e.L[i]; map.rows_out = e.L[i]; map.col_rows = H;
            // worth it when it saves a round of 128 x 256 tiles on the chip (2 workgroups per CU): 4104 -> 4096 tiles
            // for conv1, 1028 -> 1024 for conv2, 516 -> 512 for conv3 at the 20480-sample window
            const long col_blocks = std::max(1, s * H / 256);
            const long rounds_full = cdiv(cdiv((long)N * e.Rv[i], 128) * col_blocks, 512);
            const long rounds_seg = cdiv((long)N * cdiv(e.L[i + 1], 128) * col_blocks, 512);
            if (e.L[i] == s * e.L[i + 1] && rounds_seg < rounds_full) {
                // virtual rows t_hi = 0 .. L_out produce outputs, row L_out + 1 none.  L_out + 1 rows per sample do not
                // tile (1025 = 8 x 128 + 1): the main product takes rows 0 .. L_out - 1 of every sample (whole tiles,
                // no junk row multiplied), a second one with a single row per sample the boundary row t_hi = L_out
                map.seg_rows = e.Rv[i]; map.seg_valid = e.L[i + 1];
                CPC_TRY(gemm_nt(e.dU, H, e.Wd[i], 2L * H, dprev, H, nullptr, (long)N * e.Rv[i], s * H, 2 * H, map, st));
                RowMap edge{};
                edge.enabled = 1; edge.rv = 1; edge.out_stride = s; edge.out_off = e.L[i + 1] * s - p;
                edge.l_max = e.L[i]; edge.rows_out = e.L[i]; edge.col_rows = H;
                CPC_TRY(gemm_nt(e.dU + (size_t)e.L[i + 1] * H, (long)e.Rv[i] * H, e.Wd[i], 2L * H, dprev, H, nullptr, N, s * H,
                                2 * H, edge, st));
            } else {
                CPC_TRY(gemm_nt(e.dU, H, e.Wd[i], 2L * H, dprev, H, nullptr, (long)N * e.Rv[i], s * H, 2 * H, map, st));
            }
        }
        dy = dprev;
    }

    // layer 0
    Conv0Args c0{};
    c0.x = x; c0.x2 = x2; c0.n_split = x2 != nullptr ? n_first : N;
    c0.w = prm[0]; c0.b = prm[1]; c0.gamma = prm[2]; c0.beta = prm[3];
    c0.stats = e.stats0; c0.dy = dy; c0.part = e.part;
    c0.N = N; c0.L0 = e.L[0]; c0.L1 = e.L[1]; c0.R0 = e.R[0]; c0.halo = kConv[1].p; c0.eps = eps;
    c0.tiles_per_sample = (int)cdiv(e.L[1], C0_TB);
    c0.n_tiles = N * c0.tiles_per_sample;
#ifdef CPC_C0_DBG
    // (diagnostic build, tools/dp_conv0_probe.py: the head of the weight-gradient scratch, idle by now)
    c0.dbg = e.tn_bytes >= (size_t)CONV0_BWD_BLOCKS * 32 ? reinterpret_cast<unsigned *>(e.tn) : nullptr;
#endif
    {
        ProfScope prof(PROF_CONV0_BWD, st);
        CPC_DISPATCH_H(H, hipLaunchKernelGGL(conv0_bwd_kernel<HH>, dim3(CONV0_BWD_BLOCKS), dim3(256), 0, st, c0));
    }
    CPC_CHECK_LAUNCH("conv0_bwd_kernel");
    CPC_TRY(colsum_split(e.part, CONV0_BWD_BLOCKS, 13L * H, 13 * H, e.sums, e.sums + 13 * H, e.sums + 13 * H, 13 * H, e.cs, st));
    hipLaunchKernelGGL(conv0_finalize_kernel, dim3((unsigned)cdiv(H, 64)), dim3(64), 0, st, e.sums, grads[0], grads[1],
                       grads[2], grads[3], H);
    CPC_CHECK_LAUNCH("conv0_finalize_kernel");
    return CPC_OK;
}

}  // namespace cpc

extern "C" int cpc_encoder_frames(int length)
{
    int l = length;
    for (int i = 0; i < 5; ++i) l = (l + 2 * cpc::kConv[i].p - cpc::kConv[i].k) / cpc::kConv[i].s + 1;
    return l;
}

extern "C" size_t cpc_encoder_saved_bytes(int n_windows, int length, int hidden)
{
    cpc::EncLayout e;
    if (cpc::enc_layout(e, n_windows, length, hidden, nullptr, nullptr) != CPC_OK) return 0;
    return e.saved_bytes;
}

extern "C" size_t cpc_encoder_scratch_bytes(int n_windows, int length, int hidden)
{
    cpc::EncLayout e;
    if (cpc::enc_layout(e, n_windows, length, hidden, nullptr, nullptr) != CPC_OK) return 0;
    return e.scratch_bytes;
}

// inspection (tests): where layer `layer`'s saved state sits inside `saved` -- see cpc2_hip.h
extern "C" int cpc_encoder_saved_layout(int n_windows, int length, int hidden, int layer, long *out)
{
    cpc::EncLayout e;
    if (layer < 0 || layer > 4 || out == nullptr) { cpc::set_error("cpc_encoder_saved_layout: layer must be 0..4 and out non-null"); return CPC_ERR_INVALID; }
    char *const base = reinterpret_cast<char *>(uintptr_t(1) << 20);          // never dereferenced: offsets only
    CPC_TRY(cpc::enc_layout(e, n_windows, length, hidden, base, nullptr));
    for (int i = 0; i < 10; ++i) out[i] = 0;
    out[0] = out[1] = out[4] = -1;
    out[3] = e.L[layer + 1];
    if (layer >= 1) {
        out[0] = (long)(reinterpret_cast<const char *>(e.Xh[layer]) - base);
        out[1] = (long)(reinterpret_cast<const char *>(e.rstd[layer]) - base);
        out[2] = e.Rv[layer];
    }
    if (e.planes && layer < 4) {
        out[4] = (long)(reinterpret_cast<const char *>(e.Yp[layer]) - base);
        out[5] = e.Yplane[layer];
        out[6] = e.Yrts[layer];
        out[7] = cpc::log2i(cpc::kConv[layer + 1].s);
        out[8] = e.R[layer];
        out[9] = cpc::kConv[layer + 1].p;
    }
    return CPC_OK;
}

extern "C" int cpc_encoder_forward(const float *x, const float *const *params, float *z, void *saved, void *scratch,
                                   int n_windows, int length, int hidden, float eps, cpc_stream_t stream)
{
    return cpc::encoder_forward(x, params, z, saved, scratch, n_windows, length, hidden, eps, static_cast<hipStream_t>(stream));
}

extern "C" int cpc_encoder_backward(const float *x, const float *const *params, const float *dz, void *saved, void *scratch,
                                    float *const *grads, int n_windows, int length, int hidden, float eps, cpc_stream_t stream)
{
    return cpc::encoder_backward(x, params, dz, saved, scratch, grads, n_windows, length, hidden, eps,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int cpc_encoder_forward2(const float *x_first, const float *x_rest, int n_first, const float *const *params, float *z, void *saved,
                                    void *scratch, int n_windows, int length, int hidden, float eps, cpc_stream_t stream)
{
    return cpc::encoder_forward(x_first, params, z, saved, scratch, n_windows, length, hidden, eps, static_cast<hipStream_t>(stream), x_rest, n_first);
}

extern "C" int cpc_encoder_backward2(const float *x_first, const float *x_rest, int n_first, const float *const *params, const float *dz,
                                     void *saved, void *scratch, float *const *grads, int n_windows, int length, int hidden, float eps,
                                     int deferred, cpc_stream_t stream)
{
    return cpc::encoder_backward(x_first, params, dz, saved, scratch, grads, n_windows, length, hidden, eps,
                                 static_cast<hipStream_t>(stream), deferred != 0, x_rest, n_first);
}

extern "C" int cpc_encoder_backward_deferred(const float *x, const float *const *params, const float *dz, void *saved, void *scratch,
                                             float *const *grads, int n_windows, int length, int hidden, float eps, cpc_stream_t stream)
{
    return cpc::encoder_backward(x, params, dz, saved, scratch, grads, n_windows, length, hidden, eps,
                                 static_cast<hipStream_t>(stream), true);
}
