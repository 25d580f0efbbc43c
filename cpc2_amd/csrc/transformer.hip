// Light causal transformer used as autoregressive network (arMode="transformer", BASELINE config 4).
// Reference: /root/reference/cpc/transformers.py -- ScaledDotProductAttention :10-70 (causal mask, relative
// positions via the "skew" trick :61-66), MultiHeadAttention :73-104 (8 heads, no biases), FFNetwork :107-116
// (dff 2048, ReLU, dropout), TransformerLayer :119-134:
//        y   = LN1(x + Wo . MHA(x))            out = LN2(Wl (y + FFN(y)) + bl)
//
// GEMM-shaped parts (QKV/Wo/FFN/last_linear, forward and both gradients) run on the f32-MFMA GEMMs of
// gemm_f32.hip; this file adds the fused causal attention with relative-position bias (forward keeps only the
// attention probabilities; backward recomputes nothing and keeps dK/dV/dKrelpos in registers), LayerNorm row
// kernels fused with the residual add, and the ReLU+dropout elementwise pair.
//
// Dropout (p = 0.1 in training mode, transformers.py:16,112) uses a counter-based hash keyed by (seed, element
// index): statistically equivalent to torch's, not stream-identical (parity is checked with p = 0).
#include "common.h"
#include "rowcfg.h"

#include <algorithm>
#include <functional>
#include <vector>
#include <cmath>

namespace cpc {

constexpr int TR_HEADS = 8;          // transformers.py:120 (nheads=8)
constexpr int TR_QT = 32;            // query rows per attention tile

// ------------------------------------------------------------------------------------------------ LayerNorm
struct LnArgs {
    const float *a, *b;        // input = a (+ b)
    const float *w, *bias;     // [D]
    float *y, *xhat, *rstd;
    long rows;
    float eps;
    const float *dy;           // backward
    float *dx, *part;          // part[slot][2][D]: dw, db
};

template <int D> __global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs a)
{
    using Cfg = RowCfg<D>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL, RPW = Cfg::RPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gi = lane / G;
    float4 w4[VPL], b4[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        w4[v] = reinterpret_cast<const float4 *>(a.w)[v * G + gl];
        b4[v] = reinterpret_cast<const float4 *>(a.bias)[v * G + gl];
    }
    const long stride = (long)gridDim.x * 4 * RPW;
    for (long base = (long)blockIdx.x * 4 * RPW; base < a.rows; base += stride) {
        const long row = base + wave * RPW + gi;
        const bool ok = row < a.rows;
        float4 x4[VPL];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            x4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                x4[v] = reinterpret_cast<const float4 *>(a.a + row * D)[v * G + gl];
                if (a.b != nullptr) {
                    const float4 r = reinterpret_cast<const float4 *>(a.b + row * D)[v * G + gl];
                    x4[v].x += r.x; x4[v].y += r.y; x4[v].z += r.z; x4[v].w += r.w;
                }
            }
            s += (x4[v].x + x4[v].y) + (x4[v].z + x4[v].w);
        }
        const float mean = group_sum<G>(s) * (1.f / D);
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            x4[v].x -= mean; x4[v].y -= mean; x4[v].z -= mean; x4[v].w -= mean;
            ss = fmaf(x4[v].x, x4[v].x, ss); ss = fmaf(x4[v].y, x4[v].y, ss);
            ss = fmaf(x4[v].z, x4[v].z, ss); ss = fmaf(x4[v].w, x4[v].w, ss);
        }
        const float rstd = rsqrtf(group_sum<G>(ss) * (1.f / D) + a.eps);     // biased variance (nn.LayerNorm)
        if (!ok) continue;
        if (gl == 0) a.rstd[row] = rstd;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            float4 xh, o;
            xh.x = x4[v].x * rstd; xh.y = x4[v].y * rstd; xh.z = x4[v].z * rstd; xh.w = x4[v].w * rstd;
            o.x = fmaf(xh.x, w4[v].x, b4[v].x); o.y = fmaf(xh.y, w4[v].y, b4[v].y);
            o.z = fmaf(xh.z, w4[v].z, b4[v].z); o.w = fmaf(xh.w, w4[v].w, b4[v].w);
            reinterpret_cast<float4 *>(a.xhat + row * D)[v * G + gl] = xh;
            reinterpret_cast<float4 *>(a.y + row * D)[v * G + gl] = o;
        }
    }
}

constexpr int LN_BWD_BLOCKS = 256;

template <int D> __global__ __launch_bounds__(256) void ln_bwd_kernel(LnArgs a)
{
    using Cfg = RowCfg<D>;
    constexpr int G = Cfg::G, VPL = Cfg::VPL, RPW = Cfg::RPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gi = lane / G;
    float4 w4[VPL];
    float dw[VPL][4], db[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        w4[v] = reinterpret_cast<const float4 *>(a.w)[v * G + gl];
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[v][e] = db[v][e] = 0.f;
    }
    const long stride = (long)gridDim.x * 4 * RPW;
    for (long base = (long)blockIdx.x * 4 * RPW; base < a.rows; base += stride) {
        const long row = base + wave * RPW + gi;
        const bool ok = row < a.rows;
        float xh[VPL][4], g[VPL][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f), d4 = x4;
            if (ok) {
                x4 = reinterpret_cast<const float4 *>(a.xhat + row * D)[v * G + gl];
                d4 = reinterpret_cast<const float4 *>(a.dy + row * D)[v * G + gl];
            }
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
            const float wv[4] = {w4[v].x, w4[v].y, w4[v].z, w4[v].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                db[v][e] += dv[e];
                dw[v][e] = fmaf(dv[e], xv[e], dw[v][e]);
                xh[v][e] = xv[e];
                g[v][e] = dv[e] * wv[e];
                s1 += g[v][e];
                s2 = fmaf(g[v][e], xv[e], s2);
            }
        }
        s1 = group_sum<G>(s1) * (1.f / D);
        s2 = group_sum<G>(s2) * (1.f / D);
        if (!ok) continue;
        const float rstd = a.rstd[row];
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            reinterpret_cast<float4 *>(a.dx + row * D)[v * G + gl] =
                make_float4(rstd * (g[v][0] - s1 - xh[v][0] * s2), rstd * (g[v][1] - s1 - xh[v][1] * s2),
                            rstd * (g[v][2] - s1 - xh[v][2] * s2), rstd * (g[v][3] - s1 - xh[v][3] * s2));
    }
    const int slot = (blockIdx.x * 4 + wave) * RPW + gi;
    float *pp = a.part + (long)slot * 2 * D;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pp[(v * G + gl) * 4 + e] = dw[v][e];
            pp[D + (v * G + gl) * 4 + e] = db[v][e];
        }
}

// ------------------------------------------------------------------------------------------------ elementwise
__global__ void add2_kernel(float *out, const float *a, const float *b, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
        reinterpret_cast<float4 *>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

// multi-classifier head (transformers.py:137-158): t[(r, k)][:] = y[r][:] + f[r][k*D + :]  and its adjoint
__global__ void add_bcast_kernel(float *t, const float *y, const float *f, long rows, int nc, int d4)
{
    const long n4 = rows * nc * d4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ((long)nc * d4);
        const int c = (int)(i % d4);
        const float4 a = reinterpret_cast<const float4 *>(y)[r * d4 + c], b = reinterpret_cast<const float4 *>(f)[i];
        reinterpret_cast<float4 *>(t)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

__global__ void sum_classifiers_kernel(float *dy, const float *dt, long rows, int nc, int d4)
{
    const long n4 = rows * d4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / d4;
        const int c = (int)(i % d4);
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nc; ++k) {
            const float4 v = reinterpret_cast<const float4 *>(dt)[(r * nc + k) * d4 + c];
            s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        }
        reinterpret_cast<float4 *>(dy)[i] = s4;
    }
}

static int launch_add2(float *out, const float *a, const float *b, long n, hipStream_t st)
{
    const long n4 = n / 4;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)std::min<long>(cdiv(n4, 256), 4096)), dim3(256), 0, st, out, a, b, n4);
    CPC_CHECK_LAUNCH("add2_kernel");
    return CPC_OK;
}

// ------------------------------------------------------------------------------------------------ attention
struct AttnArgs {
    const float *qkv;      // [N*S][3D]  rows (n,s): Q | K | V, head h at columns h*dk
    const float *krel;     // [dk][SS] or null
    float *probs;          // [N*heads*chunks][SS][SS]  softmax output (before dropout)
    float *ctx;            // [N*S][D]   heads merged
    int N, S, D, dk, SS, chunks;
    uint64_t seed;
    uint32_t thresh;
    float scale, inv_sqrt_dk;
    // backward
    const float *dctx;     // [N*S][D]
    float *dqkv;           // [N*S][3D]
    float *dkrel_part;     // [N*heads*chunks][dk][SS]
    unsigned long long *stamps;   // -DAT_STAMPS builds (probes): [workgroup][wave][8] s_memrealtime at the phase boundaries
};

// one workgroup per (sequence chunk of one head, 32-row query tile)
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dk = a.dk, SS = a.SS, ldk = dk + 1, ldp = SS + 1;
    float *Ks = smem;                    // [SS][ldk]
    float *Vs = Ks + SS * ldk;           // [SS][ldk]
    float *Rs = Vs + SS * ldk;           // [dk][ldp]
    float *Qs = Rs + dk * ldp;           // [TR_QT][ldk]
    float *Ps = Qs + TR_QT * ldk;        // [TR_QT][ldp]

    const int tiles = (SS + TR_QT - 1) / TR_QT;
    const int cid = blockIdx.x / tiles, qt = blockIdx.x - cid * tiles;     // cid = (n*heads + h)*chunks + c
    const int c = cid % a.chunks, nh = cid / a.chunks;
    const int h = nh % TR_HEADS, n = nh / TR_HEADS;
    const long row0 = (long)n * a.S + (long)c * SS;                          // first row of this chunk in qkv
    const int tid = threadIdx.x;

    for (int i = tid; i < SS * dk; i += 256) {
        const int j = i / dk, d = i - j * dk;
        const float *src = a.qkv + (row0 + j) * 3 * a.D + h * dk + d;
        Ks[j * ldk + d] = src[a.D];
        Vs[j * ldk + d] = src[2 * a.D];
    }
    if (a.krel != nullptr)
        for (int i = tid; i < dk * SS; i += 256) Rs[(i / SS) * ldp + (i % SS)] = a.krel[i];
    for (int i = tid; i < TR_QT * dk; i += 256) {
        const int r = i / dk, d = i - r * dk;
        const int ig = qt * TR_QT + r;
        Qs[r * ldk + d] = ig < SS ? a.qkv[(row0 + ig) * 3 * a.D + h * dk + d] : 0.f;
    }
    __syncthreads();

    const int ti = tid >> 3, tj = tid & 7;
    const int ig = qt * TR_QT + ti;
    float sc[16];
    float mx = -INFINITY;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int j = tj + 8 * cc;
        float s = -INFINITY;
        if (j < SS && j <= ig && ig < SS) {
            float acc = 0.f;
            for (int d = 0; d < dk; ++d) acc = fmaf(Qs[ti * ldk + d], Ks[j * ldk + d], acc);
            if (a.krel != nullptr) {
                const int m = SS - 1 - (ig - j);                          // transformers.py:61-66
                for (int d = 0; d < dk; ++d) acc = fmaf(Qs[ti * ldk + d], Rs[d * ldp + m], acc);
            }
            s = acc * a.inv_sqrt_dk;
        }
        sc[cc] = s;
        mx = fmaxf(mx, s);
    }
    for (int off = 4; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        sc[cc] = (sc[cc] == -INFINITY) ? 0.f : expf(sc[cc] - mx);
        sum += sc[cc];
    }
    for (int off = 4; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int j = tj + 8 * cc;
        if (j < SS) {
            const float p = sc[cc] * inv;
            float pd = 0.f;
            if (ig < SS) {
                const long idx = ((long)cid * SS + ig) * SS + j;
                a.probs[idx] = p;
                pd = p * drop_mul(a.seed, (uint64_t)idx, a.thresh, a.scale);
            }
            Ps[ti * ldp + j] = pd;
        }
    }
    __syncthreads();
    if (ig < SS) {
        for (int d = tj; d < dk; d += 8) {
            float acc = 0.f;
            for (int j = 0; j <= ig; ++j) acc = fmaf(Ps[ti * ldp + j], Vs[j * ldk + d], acc);
            a.ctx[(row0 + ig) * a.D + h * dk + d] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA form of the attention for head size 32 (d_model = 256) and sizeSeq <= 128: one workgroup per (sequence chunk,
// head), wave w owns query rows 32w .. 32w+31.  f32 MFMA 32x32x2 throughout (exact f32 products).  K, Krelpos^T and V^T
// sit in LDS with the reduction index contiguous, so every operand fragment is a 16-byte read (lane (r, h) takes
// k = 8q + 4h + 0..3 of its row for q = 0..3; A and B use the same k order, which is all the product needs).
//   S  = Q K^T                       causal: only column tiles jt <= w
//   R  = Q Krelpos                   [32 x SS]; S[i][j] += R[i][SS-1-(i-j)]  (transformers.py:61-66), gathered through LDS
//   P  = softmax(S / sqrt(dk) + mask);  probs <- P;  Pd = dropout(P)
//   ctx = Pd V                       Pd staged through LDS to become the A operand
constexpr int AT_LD = 36;            // floats per Ks / RsT row (32 + pad)
constexpr int AT_LP = 132;           // floats per VsT / Ps row (128 + pad)
constexpr size_t AT_FWD_LDS = sizeof(float) * (2 * 128 * AT_LD + 32 * AT_LP + 4 * 32 * AT_LP);

typedef float at_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void at_mfma4(at_f32x16 &acc, const float4 &a4, const float4 &b4)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
}
__device__ __forceinline__ int at_row(int e, int h2) { return (e & 3) + 8 * (e >> 2) + 4 * h2; }   // C/D row of register e
__device__ __forceinline__ void at_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

#ifdef AT_STAMPS
#define AT_STAMP(i) do { if ((threadIdx.x & 63) == 0) a.stamps[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AT_STAMP(i) do { } while (0)
#endif
// Softmax of the wave's 32 query rows over the column tiles jt <= w, the probabilities to memory and their dropped-out copy to the
// wave's LDS tile.  In sweeps over the 16 rows a lane holds -- row maxima, exponentials + row sums, normalisation + stores -- with
// the tile loop outside and no per-element branch: with one wave per SIMD (121 KB of LDS per workgroup) nothing hides the latency
// of a dependent chain or the bubble of a branch, and the first form of this code (a branch per element for the tile bound, the
// sequence bound and the dropout switch, five crossbar hops per reduction) took 18 us of the wave's 41 (stamps: -DAT_STAMPS).
// Base-2 exponentials: the scores are scaled by log2(e) / sqrt(dk) in the one multiplication they get anyway.
// FULL: sizeSeq == 128 (no bound checks); DROP: dropout on.
template <bool FULL, bool DROP>
__device__ __forceinline__ void at_softmax_store(const AttnArgs &a, at_f32x16 (&sc)[4], float *Pw, int cid, int w, int r32, int h2)
{
    const int SS = a.SS;
    const float c2 = a.inv_sqrt_dk * 1.44269504088896341f;
    float mx[16], sm[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = 32 * w + at_row(e, h2);
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int j = 32 * jt + r32;
            const bool ok = jt <= w && j <= i && (FULL || i < SS);
            sc[jt][e] = ok ? sc[jt][e] * c2 : -INFINITY;
            m = fmaxf(m, sc[jt][e]);
        }
        mx[e] = m;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        mx[e] = group_max32(mx[e]);
        if (!FULL) mx[e] = mx[e] == -INFINITY ? 0.f : mx[e];      // (a row past the sequence: every exponential below is exp2(-inf) = 0)
        sm[e] = 0.f;
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
        if (jt <= w) {                                             // (wave uniform)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                sc[jt][e] = __builtin_amdgcn_exp2f(sc[jt][e] - mx[e]);
                sm[e] += sc[jt][e];
            }
        }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float t = group_sum<32>(sm[e]);
        float r = __builtin_amdgcn_rcpf(t);
        r = r * (2.f - t * r);                                      // one Newton step: the quotient is as good as a division's
        sm[e] = t > 0.f ? r : 0.f;
    }
    AT_STAMP(7);
    float *const pb = a.probs + (long)cid * SS * SS;                // (uniform)
    const uint64_t ib = (uint64_t)cid * SS * SS;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
        if (jt <= w) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int il = at_row(e, h2), i = 32 * w + il, j = 32 * jt + r32;
                const float p = sc[jt][e] * sm[e];
                const int off = i * SS + j;
                float pd = DROP ? p * drop_mul(a.seed, ib + (uint64_t)off, a.thresh, a.scale) : p;
                if (FULL) pb[off] = p;
                else if (i < SS && j < SS) pb[off] = p;
                else pd = 0.f;
                Pw[il * AT_LP + j] = pd;
            }
        }
}

__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ks = smem;                        // [128][AT_LD]   K rows
    float *RsT = Ks + 128 * AT_LD;           // [128][AT_LD]   RsT[m][d] = Krelpos[d][m]
    float *VsT = RsT + 128 * AT_LD;          // [32][AT_LP]    VsT[d][j] = V[j][d]
    float *Ps = VsT + 32 * AT_LP;            // [4][32][AT_LP] per-wave staging (R tiles, then Pd)
    const int SS = a.SS;
    const int cid = blockIdx.x;              // (n*heads + head)*chunks + c
    const int c = cid % a.chunks, nh = cid / a.chunks;
    const int head = nh % TR_HEADS, n = nh / TR_HEADS;
    const long row0 = (long)n * a.S + (long)c * SS;
    const int tid = threadIdx.x;
    AT_STAMP(0);
    const int lane = tid & 63, w = tid >> 6, r32 = lane & 31, h2 = lane >> 5;
    float *Pw = Ps + w * 32 * AT_LP;
    const int iq = 32 * w + r32;                                 // this lane's A-operand row (requested first: in flight under the staging)
    float4 qf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        qf[q] = iq < SS ? *reinterpret_cast<const float4 *>(a.qkv + (row0 + iq) * 3 * a.D + head * 32 + 8 * q + 4 * h2)
                        : make_float4(0.f, 0.f, 0.f, 0.f);

    {   // staging: every load requested before the first LDS store (see attn_bwd_mfma_kernel)
        float4 kv[4], vv[4], r4[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it, j = i >> 3, d4 = (i & 7) * 4;
            kv[it] = make_float4(0.f, 0.f, 0.f, 0.f); vv[it] = kv[it];
            if (j < SS) {
                const float *src = a.qkv + (row0 + j) * 3 * a.D + head * 32 + d4;
                kv[it] = *reinterpret_cast<const float4 *>(src + a.D);
                vv[it] = *reinterpret_cast<const float4 *>(src + 2 * a.D);
            }
        }
        if (a.krel != nullptr) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {                     // thread: position m, four head channels (one 16-byte LDS store)
                const int i = tid + 256 * it, d4 = (i >> 7) * 4, m = i & 127;
                r4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < SS) r4[it] = make_float4(a.krel[d4 * SS + m], a.krel[(d4 + 1) * SS + m], a.krel[(d4 + 2) * SS + m], a.krel[(d4 + 3) * SS + m]);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it, j = i >> 3, d4 = (i & 7) * 4;
            *reinterpret_cast<float4 *>(&Ks[j * AT_LD + d4]) = kv[it];
            VsT[(d4 + 0) * AT_LP + j] = vv[it].x; VsT[(d4 + 1) * AT_LP + j] = vv[it].y;
            VsT[(d4 + 2) * AT_LP + j] = vv[it].z; VsT[(d4 + 3) * AT_LP + j] = vv[it].w;
        }
        if (a.krel != nullptr) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int i = tid + 256 * it, d4 = (i >> 7) * 4, m = i & 127;
                *reinterpret_cast<float4 *>(&RsT[m * AT_LD + d4]) = r4[it];
            }
        }
    }
    __syncthreads();
    AT_STAMP(1);

    at_f32x16 sc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[jt][e] = 0.f;
        if (jt <= w) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                at_mfma4(sc[jt], qf[q], *reinterpret_cast<const float4 *>(&Ks[(32 * jt + r32) * AT_LD + 8 * q + 4 * h2]));
        }
    }
    AT_STAMP(2);
    if (a.krel != nullptr) {
        const int m_min = SS - 32 * w - 32 > 0 ? SS - 32 * w - 32 : 0;
        for (int mt = m_min >> 5; mt <= (SS - 1) >> 5; ++mt) {
            at_f32x16 racc;
#pragma unroll
            for (int e = 0; e < 16; ++e) racc[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                at_mfma4(racc, qf[q], *reinterpret_cast<const float4 *>(&RsT[(32 * mt + r32) * AT_LD + 8 * q + 4 * h2]));
#pragma unroll
            for (int e = 0; e < 16; ++e) Pw[at_row(e, h2) * AT_LP + 32 * mt + r32] = racc[e];
        }
        at_wave_sync();
        // (every read unconditional, at a clamped position, and selected afterwards: sixteen reads of a tile are then in flight
        //  together instead of one branch and one wait per element)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            if (jt <= w) {
                float rv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int il = at_row(e, h2), i = 32 * w + il, j = 32 * jt + r32;
                    rv[e] = Pw[il * AT_LP + min(SS - 1 - i + j, AT_LP - 1)];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = 32 * w + at_row(e, h2), j = 32 * jt + r32;
                    sc[jt][e] += (j <= i && i < SS) ? rv[e] : 0.f;
                }
            }
        at_wave_sync();                                          // Pw is reused for Pd below
    }
    AT_STAMP(3);
    // softmax over j <= i (a row lives in one 32-lane half, over the tiles jt <= w): at_softmax_store
    if (SS == 128) {
        if (a.thresh != 0u) at_softmax_store<true, true>(a, sc, Pw, cid, w, r32, h2);
        else at_softmax_store<true, false>(a, sc, Pw, cid, w, r32, h2);
    } else {
        if (a.thresh != 0u) at_softmax_store<false, true>(a, sc, Pw, cid, w, r32, h2);
        else at_softmax_store<false, false>(a, sc, Pw, cid, w, r32, h2);
    }
    at_wave_sync();
    AT_STAMP(4);
    // ctx = Pd V over j < 32 (w + 1)
    at_f32x16 cx;
#pragma unroll
    for (int e = 0; e < 16; ++e) cx[e] = 0.f;
    for (int kq = 0; kq < 4 * (w + 1); ++kq)
        at_mfma4(cx, *reinterpret_cast<const float4 *>(&Pw[r32 * AT_LP + 8 * kq + 4 * h2]),
                 *reinterpret_cast<const float4 *>(&VsT[r32 * AT_LP + 8 * kq + 4 * h2]));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = 32 * w + at_row(e, h2);
        if (i < SS) a.ctx[(row0 + i) * a.D + head * 32 + r32] = cx[e];
    }
    AT_STAMP(5);
#ifdef AT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AT_STAMP(6);
#endif
}

// Backward twin (head size 32, sizeSeq <= 128), same workgroup / wave decomposition.  Wave w owns query rows 32w.. for
// the row-wise quantities (dPd, dS, dQ) and key rows / relative positions 32w.. for the column-wise ones (dV, dK,
// dKrelpos).  With M the dropout multiplier, Pd = P.M (P is read back from the forward pass):
//   dPd = dO V^T;  dP = dPd.M;  dS = P.(dP - rowsum(dP.P)) / sqrt(dk)
//   dV  = Pd^T dO;   dK = dS^T Q;   dQ = dS K + T Krelpos^T;   dKrelpos^T = T^T Q,   T[i][m] = dS[i][m - (SS-1-i)]
// Operands whose reduction index is the LDS row (Pd^T, dS^T, T^T, T) are read 4 bytes at a time (consecutive lanes hit
// consecutive addresses); everything else is a 16-byte fragment read.
constexpr size_t AT_BWD_LDS = sizeof(float) * (4 * 32 * AT_LP + 128 * AT_LD + 128 * AT_LP);

__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *QT = smem;                        // [32][AT_LP]   QT[d][i]
    float *KT = QT + 32 * AT_LP;             // [32][AT_LP]   KT[d][j]
    float *OT = KT + 32 * AT_LP;             // [32][AT_LP]   dO^T[d][i]
    float *Rk = OT + 32 * AT_LP;             // [32][AT_LP]   Krelpos[d][m]
    float *Vs = Rk + 32 * AT_LP;             // [128][AT_LD]  V rows
    float *PS = Vs + 128 * AT_LD;            // [128][AT_LP]  Pd, then dS
    const int SS = a.SS;
    const int cid = blockIdx.x;
    const int c = cid % a.chunks, nh = cid / a.chunks;
    const int head = nh % TR_HEADS, n = nh / TR_HEADS;
    const long row0 = (long)n * a.S + (long)c * SS;
    const int tid = threadIdx.x;
    const bool rel = a.krel != nullptr;

    AT_STAMP(0);
    // staging: every load of the thread requested before the first LDS store (a `for (i = tid; ...; i += 256)` loop is not unrolled --
    // its trip count depends on tid -- and then waits for the loads of one iteration before it requests the next: four, resp. sixteen,
    // round trips to memory in a row, 7.5 of this workgroup's 37 us)
    {
        float4 q4[4], k4[4], v4[4], o4[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it, j = i >> 3, d4 = (i & 7) * 4;
            q4[it] = make_float4(0.f, 0.f, 0.f, 0.f); k4[it] = q4[it]; v4[it] = q4[it]; o4[it] = q4[it];
            if (j < SS) {
                const float *src = a.qkv + (row0 + j) * 3 * a.D + head * 32 + d4;
                q4[it] = *reinterpret_cast<const float4 *>(src);
                k4[it] = *reinterpret_cast<const float4 *>(src + a.D);
                v4[it] = *reinterpret_cast<const float4 *>(src + 2 * a.D);
                o4[it] = *reinterpret_cast<const float4 *>(a.dctx + (row0 + j) * a.D + head * 32 + d4);
            }
        }
        float rk[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int i = tid + 256 * it, d = i >> 7, m = i & 127;
            rk[it] = (rel && m < SS) ? a.krel[d * SS + m] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it, j = i >> 3, d4 = (i & 7) * 4;
            *reinterpret_cast<float4 *>(&Vs[j * AT_LD + d4]) = v4[it];
            QT[(d4 + 0) * AT_LP + j] = q4[it].x; QT[(d4 + 1) * AT_LP + j] = q4[it].y; QT[(d4 + 2) * AT_LP + j] = q4[it].z; QT[(d4 + 3) * AT_LP + j] = q4[it].w;
            KT[(d4 + 0) * AT_LP + j] = k4[it].x; KT[(d4 + 1) * AT_LP + j] = k4[it].y; KT[(d4 + 2) * AT_LP + j] = k4[it].z; KT[(d4 + 3) * AT_LP + j] = k4[it].w;
            OT[(d4 + 0) * AT_LP + j] = o4[it].x; OT[(d4 + 1) * AT_LP + j] = o4[it].y; OT[(d4 + 2) * AT_LP + j] = o4[it].z; OT[(d4 + 3) * AT_LP + j] = o4[it].w;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int i = tid + 256 * it, d = i >> 7, m = i & 127;
            Rk[d * AT_LP + m] = rk[it];
        }
    }

    const int lane = tid & 63, w = tid >> 6, r32 = lane & 31, h2 = lane >> 5;
    // ---- phase 1: dPd = dO V^T for this wave's query rows; Pd -> PS; dS kept in registers
    const int iq = 32 * w + r32;
    float4 of[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        of[q] = iq < SS ? *reinterpret_cast<const float4 *>(a.dctx + (row0 + iq) * a.D + head * 32 + 8 * q + 4 * h2)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    // the probabilities of this wave's rows, requested before anything needs them (the first form read each one inside the branch that
    // used it: 64 exposed round trips to memory, 33 of the workgroup's 60 us -- stamps, -DAT_STAMPS); masked positions read element 0
    float pv[4][16];
    const float *const pb = a.probs + (long)cid * SS * SS;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) pv[jt][e] = 0.f;
        if (jt <= w) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = 32 * w + at_row(e, h2), j = 32 * jt + r32;
                pv[jt][e] = pb[(j <= i && i < SS) ? i * SS + j : 0];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (the staging only: __syncthreads would also wait for the loads above)
    AT_STAMP(1);
    at_f32x16 ds[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) ds[jt][e] = 0.f;
        if (jt <= w) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                at_mfma4(ds[jt], of[q], *reinterpret_cast<const float4 *>(&Vs[(32 * jt + r32) * AT_LD + 8 * q + 4 * h2]));
        }
    }
    {
        float rs[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) rs[e] = 0.f;
        const uint64_t ib = (uint64_t)cid * SS * SS;
        const bool drop = a.thresh != 0u;                        // (uniform)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            if (jt <= w) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = 32 * w + at_row(e, h2), j = 32 * jt + r32;
                    const bool ok = j <= i && i < SS;
                    const float mul = drop ? drop_mul(a.seed, ib + (uint64_t)(i * SS + j), a.thresh, a.scale) : 1.f;
                    const float pr = ok ? pv[jt][e] : 0.f;
                    const float da = ok ? ds[jt][e] * mul : 0.f;          // d loss / d P (through the dropout)
                    PS[i * AT_LP + j] = pr * mul;
                    pv[jt][e] = pr;
                    ds[jt][e] = da;
                    rs[e] = fmaf(da, pr, rs[e]);
                }
            }
#pragma unroll
        for (int e = 0; e < 16; ++e) rs[e] = group_sum<32>(rs[e]);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) ds[jt][e] = pv[jt][e] * (ds[jt][e] - rs[e]) * a.inv_sqrt_dk;      // 0 where masked
    }
    __syncthreads();
    AT_STAMP(2);
    // ---- phase 2: dV[j][d] = sum_{i >= j} Pd[i][j] dO[i][d] for key rows j = 32w + r32
    const int jk = 32 * w + r32;
    {
        at_f32x16 dv;
#pragma unroll
        for (int e = 0; e < 16; ++e) dv[e] = 0.f;
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) {
            if (kq < 4 * w) continue;                                  // (wave uniform)
            const int i0 = 8 * kq + 4 * h2;
            const float4 a4 = make_float4(PS[(i0 + 0) * AT_LP + jk], PS[(i0 + 1) * AT_LP + jk], PS[(i0 + 2) * AT_LP + jk], PS[(i0 + 3) * AT_LP + jk]);
            at_mfma4(dv, a4, *reinterpret_cast<const float4 *>(&OT[r32 * AT_LP + i0]));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = 32 * w + at_row(e, h2);
            if (j < SS) a.dqkv[(row0 + j) * 3 * a.D + 2 * a.D + head * 32 + r32] = dv[e];
        }
    }
    __syncthreads();
    AT_STAMP(3);
    // ---- phase 3: dS -> PS
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
        if (jt <= w) {
#pragma unroll
            for (int e = 0; e < 16; ++e) PS[(32 * w + at_row(e, h2)) * AT_LP + 32 * jt + r32] = ds[jt][e];
        }
    __syncthreads();
    AT_STAMP(4);
    // ---- phase 4a: dK[j][d] = sum_{i >= j} dS[i][j] Q[i][d]
    {
        at_f32x16 dkk;
#pragma unroll
        for (int e = 0; e < 16; ++e) dkk[e] = 0.f;
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) {
            if (kq < 4 * w) continue;                                  // (wave uniform)
            const int i0 = 8 * kq + 4 * h2;
            const float4 a4 = make_float4(PS[(i0 + 0) * AT_LP + jk], PS[(i0 + 1) * AT_LP + jk], PS[(i0 + 2) * AT_LP + jk], PS[(i0 + 3) * AT_LP + jk]);
            at_mfma4(dkk, a4, *reinterpret_cast<const float4 *>(&QT[r32 * AT_LP + i0]));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = 32 * w + at_row(e, h2);
            if (j < SS) a.dqkv[(row0 + j) * 3 * a.D + a.D + head * 32 + r32] = dkk[e];
        }
    }
    AT_STAMP(5);
    // ---- phase 4b: dQ[i][d] = sum_{j <= i} dS[i][j] K[j][d]  (+ sum_m T[i][m] Krelpos[d][m])
    {
        at_f32x16 dq;
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[e] = 0.f;
        for (int kq = 0; kq < 4 * (w + 1); ++kq)
            at_mfma4(dq, *reinterpret_cast<const float4 *>(&PS[iq * AT_LP + 8 * kq + 4 * h2]),
                     *reinterpret_cast<const float4 *>(&KT[r32 * AT_LP + 8 * kq + 4 * h2]));
        if (rel) {
            // T[i][m] = dS[i][m - (SS-1-i)] for 0 <= m - (SS-1-i) <= i, else 0; m ranges over [SS-1-i, SS-1]
            const int m_min = SS - 32 * w - 32 > 0 ? SS - 32 * w - 32 : 0;
            const int sh = SS - 1 - iq;                           // this lane's row shift
#pragma unroll
            for (int kq = 0; kq < 16; ++kq) {
                if (kq < (m_min >> 3) || kq > ((SS - 1) >> 3)) continue;      // (wave uniform)
                const int mb = 8 * kq + 4 * h2;
                float tv[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    // (read at a clamped position, selected afterwards: no branch per element.  j <= iq is m < SS)
                    const int m = mb + x, j = m - sh;
                    const float v = PS[iq * AT_LP + max(j, 0)];
                    tv[x] = (m >= sh && m < SS && iq < SS) ? v : 0.f;
                }
                at_mfma4(dq, make_float4(tv[0], tv[1], tv[2], tv[3]), *reinterpret_cast<const float4 *>(&Rk[r32 * AT_LP + mb]));
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = 32 * w + at_row(e, h2);
            if (i < SS) a.dqkv[(row0 + i) * 3 * a.D + head * 32 + r32] = dq[e];
        }
    }
    AT_STAMP(6);
    // ---- phase 4c: dKrelpos^T[m][d] = sum_i T[i][m] Q[i][d] for m = 32w + r32
    if (rel) {
        at_f32x16 dr;
#pragma unroll
        for (int e = 0; e < 16; ++e) dr[e] = 0.f;
        const int mk = 32 * w + r32;
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) {
            const int i0 = 8 * kq + 4 * h2;
            float tv[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int i = i0 + x, j = mk - (SS - 1 - i);            // (j <= i is mk < SS)
                const float v = PS[i * AT_LP + max(j, 0)];
                tv[x] = (j >= 0 && i < SS && mk < SS) ? v : 0.f;
            }
            at_mfma4(dr, make_float4(tv[0], tv[1], tv[2], tv[3]), *reinterpret_cast<const float4 *>(&QT[r32 * AT_LP + i0]));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 32 * w + at_row(e, h2);
            if (m < SS) a.dkrel_part[((long)cid * 32 + r32) * SS + m] = dr[e];
        }
    }
    AT_STAMP(7);
}

// one workgroup per sequence chunk of one head; loops over its query tiles.  dK, dV and dKrelpos accumulate
// in registers: thread (j = tid/2, half) owns K/V row j, thread (m = tid%128, half) owns Krelpos column m.
template <int DKH>   // dk / 2
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dk = a.dk, SS = a.SS, ldk = dk + 1, ldp = SS + 1;
    float *Qs = smem;                    // [SS][ldk]
    float *Ks = Qs + SS * ldk;
    float *Vs = Ks + SS * ldk;
    float *Rs = Vs + SS * ldk;           // [dk][ldp]
    float *Ts = Rs + dk * ldp;           // [TR_QT][ldp]   P*mask, then dS
    float *Os = Ts + TR_QT * ldp;        // [TR_QT][ldk]   dO tile

    const int cid = blockIdx.x;
    const int c = cid % a.chunks, nh = cid / a.chunks;
    const int h = nh % TR_HEADS, n = nh / TR_HEADS;
    const long row0 = (long)n * a.S + (long)c * SS;
    const int tid = threadIdx.x;

    for (int i = tid; i < SS * dk; i += 256) {
        const int j = i / dk, d = i - j * dk;
        const float *src = a.qkv + (row0 + j) * 3 * a.D + h * dk + d;
        Qs[j * ldk + d] = src[0];
        Ks[j * ldk + d] = src[a.D];
        Vs[j * ldk + d] = src[2 * a.D];
    }
    if (a.krel != nullptr)
        for (int i = tid; i < dk * SS; i += 256) Rs[(i / SS) * ldp + (i % SS)] = a.krel[i];

    float dKacc[DKH], dVacc[DKH], dRacc[DKH];
#pragma unroll
    for (int e = 0; e < DKH; ++e) dKacc[e] = dVacc[e] = dRacc[e] = 0.f;
    const int oj = tid >> 1, ohalf = tid & 1;          // owner of K/V row oj, channels ohalf*DKH ..
    const int om = tid & 127, omh = tid >> 7;          // owner of Krelpos column om, channels omh*DKH ..
    const int ti = tid >> 3, tj = tid & 7;

    const int tiles = (SS + TR_QT - 1) / TR_QT;
    for (int qt = 0; qt < tiles; ++qt) {
        __syncthreads();                                // previous tile fully consumed (and staging done)
        for (int i = tid; i < TR_QT * dk; i += 256) {
            const int r = i / dk, d = i - r * dk;
            const int ig = qt * TR_QT + r;
            Os[r * ldk + d] = ig < SS ? a.dctx[(row0 + ig) * a.D + h * dk + d] : 0.f;
        }
        __syncthreads();
        const int ig = qt * TR_QT + ti;
        float pa[16], da[16];
        float rs = 0.f;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int j = tj + 8 * cc;
            pa[cc] = 0.f; da[cc] = 0.f;
            if (j < SS && j <= ig && ig < SS) {
                const long idx = ((long)cid * SS + ig) * SS + j;
                const float p = a.probs[idx];
                const float mul = drop_mul(a.seed, (uint64_t)idx, a.thresh, a.scale);
                float dp = 0.f;
                for (int d = 0; d < dk; ++d) dp = fmaf(Os[ti * ldk + d], Vs[j * ldk + d], dp);
                pa[cc] = p;
                da[cc] = dp * mul;                      // d loss / d A (through the dropout)
                rs = fmaf(da[cc], p, rs);
                Ts[ti * ldp + j] = p * mul;             // dropped probabilities, for dV
            } else if (j < SS) {
                Ts[ti * ldp + j] = 0.f;
            }
        }
        for (int off = 4; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
        __syncthreads();
        // dV[j][d] += sum_i Pdrop[i][j] * dO[i][d]
        if (oj < SS) {
            for (int r = 0; r < TR_QT; ++r) {
                const float p = Ts[r * ldp + oj];
#pragma unroll
                for (int e = 0; e < DKH; ++e) dVacc[e] = fmaf(p, Os[r * ldk + ohalf * DKH + e], dVacc[e]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int j = tj + 8 * cc;
            if (j < SS) Ts[ti * ldp + j] = pa[cc] * (da[cc] - rs) * a.inv_sqrt_dk;      // dS (0 where masked)
        }
        __syncthreads();
        // dQ[i][d] = sum_j dS[i][j] (K[j][d] + Krel[d][SS-1-(i-j)])
        if (ig < SS) {
            for (int d = tj; d < dk; d += 8) {
                float acc = 0.f;
                for (int j = 0; j <= ig; ++j) {
                    float kv = Ks[j * ldk + d];
                    if (a.krel != nullptr) kv += Rs[d * ldp + SS - 1 - (ig - j)];
                    acc = fmaf(Ts[ti * ldp + j], kv, acc);
                }
                a.dqkv[(row0 + ig) * 3 * a.D + h * dk + d] = acc;
            }
        }
        // dK[j][d] += sum_i dS[i][j] Q[i][d]
        if (oj < SS) {
            for (int r = 0; r < TR_QT; ++r) {
                const int i2 = qt * TR_QT + r;
                if (i2 >= SS) break;
                const float ds = Ts[r * ldp + oj];
#pragma unroll
                for (int e = 0; e < DKH; ++e) dKacc[e] = fmaf(ds, Qs[i2 * ldk + ohalf * DKH + e], dKacc[e]);
            }
        }
        // dKrel[d][m] += sum_{i: j = i-(SS-1-m) >= 0} dS[i][j] Q[i][d]
        if (a.krel != nullptr && om < SS) {
            for (int r = 0; r < TR_QT; ++r) {
                const int i2 = qt * TR_QT + r;
                if (i2 >= SS) break;
                const int j = i2 - (SS - 1 - om);
                if (j < 0) continue;
                const float ds = Ts[r * ldp + j];
#pragma unroll
                for (int e = 0; e < DKH; ++e) dRacc[e] = fmaf(ds, Qs[i2 * ldk + omh * DKH + e], dRacc[e]);
            }
        }
    }
    if (oj < SS) {
#pragma unroll
        for (int e = 0; e < DKH; ++e) {
            float *dst = a.dqkv + (row0 + oj) * 3 * a.D + h * dk + ohalf * DKH + e;
            dst[a.D] = dKacc[e];
            dst[2 * a.D] = dVacc[e];
        }
    }
    if (a.krel != nullptr && om < SS) {
#pragma unroll
        for (int e = 0; e < DKH; ++e) a.dkrel_part[((long)cid * dk + omh * DKH + e) * SS + om] = dRacc[e];
    }
}

// ------------------------------------------------------------------------------------------------ orchestration
// per-layer parameter order of the C ABI
enum { P_WQ = 0, P_WK, P_WV, P_WO, P_KREL, P_LN1W, P_LN1B, P_W1, P_B1, P_W2, P_B2, P_WL, P_BL, P_LN2W, P_LN2B, P_COUNT };
constexpr int TR_DFF = 2048;        // transformers.py:119 (dff=2048)

struct TrLayout {
    int N, S, D, Dout, SS, layers, dk, chunks, nc;
    long rows;
    // saved per layer
    float *qkv[4], *probs[4], *ctx[4], *xh1[4], *rstd1[4], *y[4], *hdrop[4], *t[4], *xh2[4], *rstd2[4], *xout[4];
    size_t saved_bytes;
    // scratch
    float *wqkv, *wt, *o, *u, *da, *db, *dc, *dh, *dqkv, *part, *krel_part, *tn, *cs;
    float *da2, *db2, *part2, *tn2;      // second homes of buffers the backward writes twice (transformer_backward, defer_tail)
    size_t tn_bytes, scratch_bytes, lds_fwd, lds_bwd;
};

static int tr_layout(TrLayout &L, int N, int S, int D, int Dout, int SS, int layers, int nc, void *saved, void *scratch)
{
    CPC_REQUIRE(nc >= 1 && nc <= 64, "transformer: 1..64 classifiers in the last layer's head (got %d)", nc);
    CPC_REQUIRE(nc == 1 || layers == 1 || D == Dout, "transformer: stacked layers need dmodel == dout");
    CPC_REQUIRE(supported_row_width(D) && supported_row_width(Dout), "transformer: model dims %d/%d not supported", D, Dout);
    CPC_REQUIRE(layers >= 1 && layers <= 4, "transformer: 1..4 layers supported (got %d)", layers);
    CPC_REQUIRE(layers == 1 || D == Dout, "transformer: stacked layers need dmodel == dout");
    CPC_REQUIRE(N > 0 && S > 0 && SS > 0 && SS <= 128, "transformer: need 0 < sizeSeq <= 128 (got %d)", SS);
    CPC_REQUIRE(S % SS == 0, "transformer: sequence length %d must be a multiple of sizeSeq %d", S, SS);
    L.N = N; L.S = S; L.D = D; L.Dout = Dout; L.SS = SS; L.layers = layers; L.dk = D / TR_HEADS; L.chunks = S / SS; L.nc = nc;
    L.rows = (long)N * S;
    const size_t R = (size_t)L.rows;
    const size_t nchunk = (size_t)N * TR_HEADS * L.chunks;
    Carver sv(saved);
    for (int l = 0; l < layers; ++l) {
        L.qkv[l] = sv.take<float>(R * 3 * D);
        L.probs[l] = sv.take<float>(nchunk * SS * SS);
        L.ctx[l] = sv.take<float>(R * D);
        L.xh1[l] = sv.take<float>(R * D);
        L.rstd1[l] = sv.take<float>(R);
        L.y[l] = sv.take<float>(R * D);
        L.hdrop[l] = sv.take<float>(R * TR_DFF);
        const size_t k = (l + 1 == layers) ? (size_t)nc : 1;       // the multi-classifier head is the last layer only
        L.t[l] = sv.take<float>(R * k * D);
        L.xh2[l] = sv.take<float>(R * k * Dout);
        L.rstd2[l] = sv.take<float>(R * k);
        L.xout[l] = (l + 1 < layers) ? sv.take<float>(R * Dout) : nullptr;
    }
    L.saved_bytes = sv.used();
    Carver sc(scratch);
    const int dmax = std::max(D, Dout);
    L.wqkv = sc.take<float>((size_t)3 * D * D);
    L.wt = sc.take<float>(std::max((size_t)TR_DFF * dmax * nc, (size_t)3 * D * D));
    L.o = sc.take<float>(R * dmax);
    L.u = sc.take<float>(R * dmax * nc);
    L.da = sc.take<float>(R * dmax * nc);
    L.db = sc.take<float>(R * dmax * nc);
    L.dc = sc.take<float>(R * dmax);
    L.dh = sc.take<float>(R * TR_DFF);
    L.dqkv = sc.take<float>(R * 3 * D);
    L.part = sc.take<float>((size_t)LN_BWD_BLOCKS * 4 * rows_per_wave(std::min(D, Dout)) * 2 * dmax);
    L.krel_part = sc.take<float>(nchunk * L.dk * SS);
    L.cs = sc.take<float>(colsum_rows_scratch_bytes(std::max(TR_DFF, nc * D)) / sizeof(float));
    L.tn_bytes = std::max(gemm_tn_scratch_bytes(TR_DFF, dmax, L.rows), gemm_tn_scratch_bytes(dmax * nc, TR_DFF, L.rows));
    L.tn_bytes = std::max(L.tn_bytes, gemm_tn_scratch_bytes(dmax, dmax, L.rows * nc));
    L.tn_bytes = std::max(L.tn_bytes, gemm_tn_scratch_bytes(3 * D, D, L.rows));                // (Q | K | V weight gradients as one product)
    // the same room serves an ordered K split of the layer's products when they have few tiles
    L.tn_bytes = std::max(L.tn_bytes, std::max(gemm_nt_scratch_bytes(L.rows * nc, dmax, TR_DFF), gemm_nt_scratch_bytes(L.rows, dmax, 3 * dmax)));
    L.tn = sc.take<float>(L.tn_bytes / sizeof(float));
    L.da2 = sc.take<float>(R * dmax * nc);
    L.db2 = sc.take<float>(R * dmax * nc);
    L.part2 = sc.take<float>((size_t)LN_BWD_BLOCKS * 4 * rows_per_wave(std::min(D, Dout)) * 2 * dmax);
    L.tn2 = sc.take<float>(L.tn_bytes / sizeof(float));
    L.scratch_bytes = sc.used();
    const size_t ldk = L.dk + 1, ldp = SS + 1;
    L.lds_fwd = sizeof(float) * (2 * SS * ldk + L.dk * ldp + TR_QT * ldk + TR_QT * ldp);
    L.lds_bwd = sizeof(float) * (3 * SS * ldk + L.dk * ldp + TR_QT * ldp + TR_QT * ldk);
    CPC_REQUIRE(L.lds_bwd <= 160 * 1024, "transformer: attention tile needs %zu B of LDS", L.lds_bwd);
    return CPC_OK;
}

template <typename K> static int allow_lds_tr(K kern, size_t bytes)
{
    if (bytes > 64 * 1024)
        CPC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return CPC_OK;
}

static int launch_ln_fwd(const float *a, const float *b, const float *w, const float *bias, float *y, float *xhat, float *rstd,
                         long rows, int D, float eps, hipStream_t st)
{
    LnArgs la{};
    la.a = a; la.b = b; la.w = w; la.bias = bias; la.y = y; la.xhat = xhat; la.rstd = rstd; la.rows = rows; la.eps = eps;
    const int rpb = 4 * rows_per_wave(D);
    const int blocks = (int)std::min<long>(cdiv(rows, rpb), 4096);
    CPC_DISPATCH_H(D, hipLaunchKernelGGL(ln_fwd_kernel<HH>, dim3(blocks), dim3(256), 0, st, la));
    CPC_CHECK_LAUNCH("ln_fwd_kernel");
    return CPC_OK;
}

// dgamma / dbeta of a LayerNorm from the partial sums its backward kernel left
static int ln_bwd_sums(const float *part, int D, float *dw, float *db, hipStream_t st)
{
    const long slots = (long)LN_BWD_BLOCKS * 4 * rows_per_wave(D);
    CPC_TRY(colsum(part, slots, 2L * D, D, dw, st));
    CPC_TRY(colsum(part + D, slots, 2L * D, D, db, st));
    return CPC_OK;
}

static int launch_ln_bwd(const float *dy, const float *xhat, const float *rstd, const float *w, float *dx, float *dw, float *db,
                         float *part, long rows, int D, hipStream_t st)
{
    LnArgs la{};
    la.dy = dy; la.xhat = const_cast<float *>(xhat); la.rstd = const_cast<float *>(rstd); la.w = w; la.dx = dx; la.part = part;
    la.rows = rows;
    CPC_DISPATCH_H(D, hipLaunchKernelGGL(ln_bwd_kernel<HH>, dim3(LN_BWD_BLOCKS), dim3(256), 0, st, la));
    CPC_CHECK_LAUNCH("ln_bwd_kernel");
    if (dw == nullptr) return CPC_OK;            // the caller sums the partials later (ln_bwd_sums)
    return ln_bwd_sums(part, D, dw, db, st);
}

static uint32_t drop_thresh(float p) { return p <= 0.f ? 0u : (uint32_t)std::min(4294967295.0, (double)p * 4294967296.0); }

static int transformer_forward(const float *x, const float *const *prm, float *out, void *saved, void *scratch, int N, int S,
                               int D, int Dout, int SS, int layers, int nc, float p_drop, uint64_t seed, hipStream_t st)
{
    TrLayout L;
    CPC_TRY(tr_layout(L, N, S, D, Dout, SS, layers, nc, saved, scratch));
    const long R = L.rows;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thresh = drop_thresh(p_drop);
    RowMap none{};
    none.splitk_scratch = L.tn; none.splitk_bytes = L.tn_bytes;
    const float *xin = x;
    for (int l = 0; l < layers; ++l) {
        const float *const *p = prm + (size_t)l * P_COUNT;
        const uint64_t lseed = seed + 0x1000ull * (uint64_t)l;
        // Q | K | V = x [Wq; Wk; Wv]^T                                     (transformers.py:99-102)
        // (one product when the three weights sit back to back -- FlatAdam's buffer keeps them in registration order)
        if (p[P_WK] == p[P_WQ] + (size_t)D * D && p[P_WV] == p[P_WK] + (size_t)D * D) {
            CPC_TRY(gemm_nt(xin, D, p[P_WQ], D, L.qkv[l], 3L * D, nullptr, R, 3 * D, D, none, st));
        } else {
            for (int i = 0; i < 3; ++i)
                CPC_TRY(gemm_nt(xin, D, p[P_WQ + i], D, L.qkv[l] + (size_t)i * D, 3L * D, nullptr, R, D, D, none, st));
        }
        AttnArgs aa{};
        aa.qkv = L.qkv[l]; aa.krel = p[P_KREL]; aa.probs = L.probs[l]; aa.ctx = L.ctx[l];
        aa.N = N; aa.S = S; aa.D = D; aa.dk = L.dk; aa.SS = SS; aa.chunks = L.chunks;
        aa.seed = lseed; aa.thresh = thresh; aa.scale = scale; aa.inv_sqrt_dk = 1.f / std::sqrt((float)L.dk);
        static const bool attn_valu = getenv("CPC_ATTN_VALU") != nullptr;       // the VALU kernels, for A/B tests
        if (L.dk == 32 && !attn_valu) {
            CPC_TRY(allow_lds_tr(attn_fwd_mfma_kernel, AT_FWD_LDS));
#ifdef AT_STAMPS
            static unsigned long long *stamps = nullptr;
            const int nwg = N * TR_HEADS * L.chunks;
            if (stamps == nullptr) CPC_CHECK_HIP(hipMalloc(&stamps, 65536 * 32 * sizeof(unsigned long long)));
            aa.stamps = stamps;
#endif
            hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3((unsigned)(N * TR_HEADS * L.chunks)), dim3(256), AT_FWD_LDS, st, aa);
            CPC_CHECK_LAUNCH("attn_fwd_mfma_kernel");
#ifdef AT_STAMPS
            {
                static std::vector<unsigned long long> host(65536 * 32);
                CPC_CHECK_HIP(hipStreamSynchronize(st));
                const int nb = std::min(nwg, 65536);
                CPC_CHECK_HIP(hipMemcpy(host.data(), stamps, (size_t)nb * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                for (int w = 0; w < 4; ++w) {
                    double ph[6] = {0, 0, 0, 0, 0, 0};
                    for (int b = 0; b < nb; ++b)
                        for (int i = 0; i < 6; ++i) ph[i] += (double)(host[((size_t)b * 4 + w) * 8 + i + 1] - host[((size_t)b * 4 + w) * 8 + i]);
                    double sw = 0;
                    for (int b = 0; b < nb; ++b) sw += (double)(host[((size_t)b * 4 + w) * 8 + 7] - host[((size_t)b * 4 + w) * 8 + 3]);
                    fprintf(stderr, "attn fwd stamps wave %d (us): load %.2f | q+S %.2f | R+skew %.2f | softmax %.2f (max, exp, sum sweeps %.2f) | PV + ctx %.2f | drain %.2f\n", w,
                            ph[0] / nb * 0.01, ph[1] / nb * 0.01, ph[2] / nb * 0.01, ph[3] / nb * 0.01, sw / nb * 0.01, ph[4] / nb * 0.01, ph[5] / nb * 0.01);
                }
                unsigned long long first = ~0ull, last = 0;
                for (int b = 0; b < nb; ++b) { first = std::min(first, host[(size_t)b * 32]); for (int w = 0; w < 4; ++w) last = std::max(last, host[((size_t)b * 4 + w) * 8 + 6]); }
                fprintf(stderr, "attn fwd: %d workgroups, span %.1f us\n", nb, (double)(last - first) * 0.01);
            }
#endif
        } else {
            CPC_TRY(allow_lds_tr(attn_fwd_kernel, L.lds_fwd));
            const int tiles = (SS + TR_QT - 1) / TR_QT;
            hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(N * TR_HEADS * L.chunks * tiles)), dim3(256), L.lds_fwd, st, aa);
            CPC_CHECK_LAUNCH("attn_fwd_kernel");
        }
        CPC_TRY(gemm_nt(L.ctx[l], D, p[P_WO], D, L.o, D, nullptr, R, D, D, none, st));                       // Wo (:104)
        CPC_TRY(launch_ln_fwd(xin, L.o, p[P_LN1W], p[P_LN1B], L.y[l], L.xh1[l], L.rstd1[l], R, D, 1e-5f, st));  // :133
        {   // lin1 (:116) with ReLU + dropout (:112-116) in the product's epilogue (the separate pass read and wrote 134 MB at C4's shape)
            RowMap act = none;
            act.epi = EPI_RELU_DROPOUT; act.epi_seed = lseed ^ 0xFFull; act.epi_thresh = thresh; act.epi_scale = scale;
            act.splitk_scratch = nullptr; act.splitk_bytes = 0;
            CPC_TRY(gemm_nt(L.y[l], D, p[P_W1], D, L.hdrop[l], TR_DFF, p[P_B1], R, TR_DFF, D, act, st));
        }
        const int k = (l + 1 == layers) ? nc : 1;          // classifiers of this layer's head: lin2 is [k*D][dff]
        CPC_TRY(gemm_nt(L.hdrop[l], TR_DFF, p[P_W2], TR_DFF, L.u, (long)k * D, p[P_B2], R, k * D, TR_DFF, none, st));   // lin2
        if (k == 1) {
            CPC_TRY(launch_add2(L.t[l], L.y[l], L.u, R * D, st));                                               // y + FFN(y) (:134)
        } else {                                                                                                 // (:156-158)
            hipLaunchKernelGGL(add_bcast_kernel, dim3(4096), dim3(256), 0, st, L.t[l], L.y[l], L.u, R, k, D / 4);
            CPC_CHECK_LAUNCH("add_bcast_kernel");
        }
        CPC_TRY(gemm_nt(L.t[l], D, p[P_WL], D, L.u, Dout, p[P_BL], R * k, Dout, D, none, st));                   // last_linear
        float *yo = (l + 1 < layers) ? L.xout[l] : out;
        CPC_TRY(launch_ln_fwd(L.u, nullptr, p[P_LN2W], p[P_LN2B], yo, L.xh2[l], L.rstd2[l], R * k, Dout, 1e-5f, st));
        xin = yo;
    }
    return CPC_OK;
}

static int transformer_backward(const float *x, const float *const *prm, const float *dout, void *saved, void *scratch, float *dx,
                                float *const *grads, int N, int S, int D, int Dout, int SS, int layers, int nc, float p_drop,
                                uint64_t seed, hipStream_t st, bool defer_tail = false)
{
    TrLayout L;
    CPC_TRY(tr_layout(L, N, S, D, Dout, SS, layers, nc, saved, scratch));
    const long R = L.rows;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thresh = drop_thresh(p_drop);
    RowMap none{};
    none.splitk_scratch = L.tn; none.splitk_bytes = L.tn_bytes;
    const float *dcur = dout;
    // defer_tail: everything that only FINISHES a parameter gradient of the last layer handled (layer 0) -- seven weight-gradient
    // products, their bias sums, the LayerNorms' and Krelpos' column sums: ~0.6 ms of a CPC-transformer step that nothing on `st`
    // needs -- is collected and launched on the library's side stream at the end (ONE fork), where it runs under the encoder's
    // backward.  The two buffers the pass writes twice (da, db) and the LayerNorm partials get second homes, the input-gradient
    // products their own K-split room.
    std::vector<std::function<int(hipStream_t)>> tail;
    for (int l = layers - 1; l >= 0; --l) {
        const float *const *p = prm + (size_t)l * P_COUNT;
        float *const *g = grads + (size_t)l * P_COUNT;
        const float *xin = (l == 0) ? x : L.xout[l - 1];
        const uint64_t lseed = seed + 0x1000ull * (uint64_t)l;
        const int k = (l + 1 == layers) ? nc : 1;          // classifiers of this layer's head
        const bool dt = defer_tail && l == 0 && nc == 1;
        float *const da2 = dt ? L.da2 : L.da, *const db2 = dt ? L.db2 : L.db, *const part2 = dt ? L.part2 : L.part;
        if (dt) { none.splitk_scratch = L.tn2; }            // (L.tn is the side stream's from here on)
        auto W = [&](std::function<int(hipStream_t)> fn) -> int {
            if (dt) { tail.push_back(std::move(fn)); return CPC_OK; }
            return fn(st);
        };
        // LN2
        if (l == layers - 1) CPC_TRY(infonce_deferred_mark(st));      // (see infonce_deferred_start below)
        CPC_TRY(launch_ln_bwd(dcur, L.xh2[l], L.rstd2[l], p[P_LN2W], L.da, nullptr, nullptr, L.part, R * k, Dout, st));       // da = du
        // a deferred criterion backward waiting to run (its dz sum and predictor weight gradients) starts here, beside this layer's
        // backward, as it does beside the recurrent kernels (no-op when nothing is pending)
        if (l == layers - 1) CPC_TRY(infonce_deferred_start(st));
        CPC_TRY(W([=](hipStream_t ws) -> int { return ln_bwd_sums(L.part, Dout, g[P_LN2W], g[P_LN2B], ws); }));
        // last_linear: u = t Wl^T + bl
        CPC_TRY(W([=](hipStream_t ws) -> int {
            CPC_TRY(gemm_tn(L.da, Dout, L.t[l], D, g[P_WL], D, Dout, D, R * k, L.tn, L.tn_bytes, 0, 0, ws));
            return colsum_rows(L.da, Dout, R * k, Dout, g[P_BL], L.cs, ws);
        }));
        CPC_TRY(transpose2d(p[P_WL], L.wt, Dout, D, st));                                            // [D][Dout]
        CPC_TRY(gemm_nt(L.da, Dout, L.wt, Dout, L.db, D, nullptr, R * k, D, Dout, none, st));        // db = dt [R*k][D] = df [R][k*D]
        // lin2: f = h W2^T + b2
        CPC_TRY(W([=](hipStream_t ws) -> int {
            CPC_TRY(gemm_tn(L.db, (long)k * D, L.hdrop[l], TR_DFF, g[P_W2], TR_DFF, k * D, TR_DFF, R, L.tn, L.tn_bytes, 0, 0, ws));
            return colsum_rows(L.db, (long)k * D, R, k * D, g[P_B2], L.cs, ws);
        }));
        CPC_TRY(transpose2d(p[P_W2], L.wt, k * D, TR_DFF, st));                                      // [dff][k*D]
        {   // (the adjoint of ReLU + dropout in the epilogue: h holds relu(.) * mask * scale, an element carries gradient iff it is > 0)
            RowMap gate = none;
            gate.epi = EPI_GATE; gate.epi_gate = L.hdrop[l]; gate.epi_scale = scale;
            gate.splitk_scratch = nullptr; gate.splitk_bytes = 0;
            CPC_TRY(gemm_nt(L.db, (long)k * D, L.wt, (long)k * D, L.dh, TR_DFF, nullptr, R, TR_DFF, k * D, gate, st));
        }
        // lin1: h = y W1^T + b1
        CPC_TRY(W([=](hipStream_t ws) -> int {
            CPC_TRY(gemm_tn(L.dh, TR_DFF, L.y[l], D, g[P_W1], D, TR_DFF, D, R, L.tn, L.tn_bytes, 0, 0, ws));
            return colsum_rows(L.dh, TR_DFF, R, TR_DFF, g[P_B1], L.cs, ws);
        }));
        CPC_TRY(transpose2d(p[P_W1], L.wt, TR_DFF, D, st));                                          // [D][dff]
        CPC_TRY(gemm_nt(L.dh, TR_DFF, L.wt, TR_DFF, L.dc, D, nullptr, R, D, TR_DFF, none, st));      // dc = dy_b
        if (k > 1) {                                                                                  // dy_a = sum over classifiers of dt
            hipLaunchKernelGGL(sum_classifiers_kernel, dim3(2048), dim3(256), 0, st, L.da, L.db, R, k, D / 4);
            CPC_CHECK_LAUNCH("sum_classifiers_kernel");
            CPC_TRY(launch_add2(L.dc, L.dc, L.da, R * D, st));
        } else {
            CPC_TRY(launch_add2(L.dc, L.dc, L.db, R * D, st));                                        // dy = dy_a + dy_b
        }
        // LN1: y = LN(x + o)
        CPC_TRY(launch_ln_bwd(L.dc, L.xh1[l], L.rstd1[l], p[P_LN1W], da2, nullptr, nullptr, part2, R, D, st));             // da2 = d(x+o)
        CPC_TRY(W([=](hipStream_t ws) -> int { return ln_bwd_sums(part2, D, g[P_LN1W], g[P_LN1B], ws); }));
        // Wo: o = ctx Wo^T
        CPC_TRY(W([=](hipStream_t ws) -> int { return gemm_tn(da2, D, L.ctx[l], D, g[P_WO], D, D, D, R, L.tn, L.tn_bytes, 0, 0, ws); }));
        CPC_TRY(transpose2d(p[P_WO], L.wt, D, D, st));
        CPC_TRY(gemm_nt(da2, D, L.wt, D, db2, D, nullptr, R, D, D, none, st));                       // db2 = dctx
        // attention
        AttnArgs aa{};
        aa.qkv = L.qkv[l]; aa.krel = p[P_KREL]; aa.probs = L.probs[l];
        aa.N = N; aa.S = S; aa.D = D; aa.dk = L.dk; aa.SS = SS; aa.chunks = L.chunks;
        aa.seed = lseed; aa.thresh = thresh; aa.scale = scale; aa.inv_sqrt_dk = 1.f / std::sqrt((float)L.dk);
        aa.dctx = db2; aa.dqkv = L.dqkv; aa.dkrel_part = L.krel_part;
        const int nchunk = N * TR_HEADS * L.chunks;
        int status = CPC_OK;
        static const bool attn_valu = getenv("CPC_ATTN_VALU") != nullptr;
        if (L.dk == 32 && !attn_valu) {
            status = allow_lds_tr(attn_bwd_mfma_kernel, AT_BWD_LDS);
#ifdef AT_STAMPS
            static unsigned long long *bstamps = nullptr;
            if (bstamps == nullptr) CPC_CHECK_HIP(hipMalloc(&bstamps, 65536 * 32 * sizeof(unsigned long long)));
            aa.stamps = bstamps;
#endif
            if (status == CPC_OK) hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3((unsigned)nchunk), dim3(256), AT_BWD_LDS, st, aa);
#ifdef AT_STAMPS
            {
                static std::vector<unsigned long long> host(65536 * 32);
                CPC_CHECK_HIP(hipStreamSynchronize(st));
                const int nb = std::min(nchunk, 65536);
                CPC_CHECK_HIP(hipMemcpy(host.data(), bstamps, (size_t)nb * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                for (int w = 0; w < 4; ++w) {
                    double ph[7] = {0, 0, 0, 0, 0, 0, 0};
                    for (int b = 0; b < nb; ++b)
                        for (int i = 0; i < 7; ++i) ph[i] += (double)(host[((size_t)b * 4 + w) * 8 + i + 1] - host[((size_t)b * 4 + w) * 8 + i]);
                    fprintf(stderr, "attn bwd stamps wave %d (us): load %.2f | dPd + dS %.2f | dV %.2f | dS -> LDS %.2f | dK %.2f | dQ %.2f | dKrel %.2f\n", w,
                            ph[0] / nb * 0.01, ph[1] / nb * 0.01, ph[2] / nb * 0.01, ph[3] / nb * 0.01, ph[4] / nb * 0.01, ph[5] / nb * 0.01, ph[6] / nb * 0.01);
                }
                unsigned long long first = ~0ull, last = 0;
                for (int b = 0; b < nb; ++b) { first = std::min(first, host[(size_t)b * 32]); for (int w = 0; w < 4; ++w) last = std::max(last, host[((size_t)b * 4 + w) * 8 + 7]); }
                fprintf(stderr, "attn bwd: %d workgroups, span %.1f us\n", nb, (double)(last - first) * 0.01);
            }
#endif
        } else
        switch (L.dk / 2) {
#define TR_CASE(X) case X: status = allow_lds_tr(attn_bwd_kernel<X>, L.lds_bwd); \
        if (status == CPC_OK) hipLaunchKernelGGL(attn_bwd_kernel<X>, dim3((unsigned)nchunk), dim3(256), L.lds_bwd, st, aa); break;
            TR_CASE(2) TR_CASE(4) TR_CASE(8) TR_CASE(16) TR_CASE(32)
#undef TR_CASE
        default: set_error("transformer: head size %d not supported", L.dk); return CPC_ERR_INVALID;
        }
        CPC_TRY(status);
        CPC_CHECK_LAUNCH("attn_bwd_kernel");
        if (p[P_KREL] != nullptr)
            CPC_TRY(W([=](hipStream_t ws) -> int { return colsum(L.krel_part, nchunk, (long)L.dk * SS, L.dk * SS, g[P_KREL], ws); }));
        // Q/K/V projections
        CPC_TRY(W([=](hipStream_t ws) -> int {
            if (g[P_WK] == g[P_WQ] + (size_t)D * D && g[P_WV] == g[P_WK] + (size_t)D * D)               // (back to back: one product)
                return gemm_tn(L.dqkv, 3L * D, xin, D, g[P_WQ], D, 3 * D, D, R, L.tn, L.tn_bytes, 0, 0, ws);
            for (int i = 0; i < 3; ++i)
                CPC_TRY(gemm_tn(L.dqkv + (size_t)i * D, 3L * D, xin, D, g[P_WQ + i], D, D, D, R, L.tn, L.tn_bytes, 0, 0, ws));
            return (int)CPC_OK;
        }));
        float *dxl = (l == 0) ? dx : L.dc;
        if (dxl != nullptr) {
            const float *wqkv = p[P_WQ];
            if (!(p[P_WK] == p[P_WQ] + (size_t)D * D && p[P_WV] == p[P_WK] + (size_t)D * D)) {
                for (int i = 0; i < 3; ++i) CPC_CHECK_HIP(hipMemcpyAsync(L.wqkv + (size_t)i * D * D, p[P_WQ + i], sizeof(float) * D * D,
                                                                         hipMemcpyDeviceToDevice, st));
                wqkv = L.wqkv;
            }
            CPC_TRY(transpose2d(wqkv, L.wt, 3 * D, D, st));                                           // [D][3D]
            CPC_TRY(gemm_nt(L.dqkv, 3L * D, L.wt, 3L * D, L.u, D, nullptr, R, D, 3 * D, none, st));
            CPC_TRY(launch_add2(dxl, L.u, da2, R * D, st));                                            // + residual path
        }
        dcur = dxl;
    }
    if (!tail.empty()) {
        hipStream_t ws = st;
        CPC_TRY(side_tail_begin(st, &ws));
        for (auto &fn : tail) CPC_TRY(fn(ws));
        CPC_TRY(side_tail_end());
    }
    return CPC_OK;
}

}  // namespace cpc

extern "C" int cpc_transformer_param_count(void) { return cpc::P_COUNT; }

extern "C" size_t cpc_transformer_saved_bytes(int n, int s, int d_model, int d_out, int size_seq, int layers, int n_classifiers)
{
    cpc::TrLayout L;
    if (cpc::tr_layout(L, n, s, d_model, d_out, size_seq, layers, n_classifiers, nullptr, nullptr) != CPC_OK) return 0;
    return L.saved_bytes;
}

extern "C" size_t cpc_transformer_scratch_bytes(int n, int s, int d_model, int d_out, int size_seq, int layers, int n_classifiers)
{
    cpc::TrLayout L;
    if (cpc::tr_layout(L, n, s, d_model, d_out, size_seq, layers, n_classifiers, nullptr, nullptr) != CPC_OK) return 0;
    return L.scratch_bytes;
}

extern "C" int cpc_transformer_forward(const float *x, const float *const *params, float *out, void *saved, void *scratch, int n,
                                       int s, int d_model, int d_out, int size_seq, int layers, int n_classifiers, float dropout_p,
                                       unsigned long long seed, cpc_stream_t stream)
{
    return cpc::transformer_forward(x, params, out, saved, scratch, n, s, d_model, d_out, size_seq, layers, n_classifiers, dropout_p, seed,
                                    static_cast<hipStream_t>(stream));
}

extern "C" int cpc_transformer_backward(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                        float *dx, float *const *grads, int n, int s, int d_model, int d_out, int size_seq,
                                        int layers, int n_classifiers, float dropout_p, unsigned long long seed, cpc_stream_t stream)
{
    return cpc::transformer_backward(x, params, dout, saved, scratch, dx, grads, n, s, d_model, d_out, size_seq, layers, n_classifiers, dropout_p,
                                     seed, static_cast<hipStream_t>(stream));
}

extern "C" int cpc_transformer_backward_deferred(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                                 float *dx, float *const *grads, int n, int s, int d_model, int d_out, int size_seq,
                                                 int layers, int n_classifiers, float dropout_p, unsigned long long seed,
                                                 cpc_stream_t stream)
{
    return cpc::transformer_backward(x, params, dout, saved, scratch, dx, grads, n, s, d_model, d_out, size_seq, layers, n_classifiers, dropout_p,
                                     seed, static_cast<hipStream_t>(stream), true);
}
