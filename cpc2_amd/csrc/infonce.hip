// CPCUnsupersivedCriterion with linear predictors on gfx950.
// Reference: /root/reference/cpc/criterion/criterion.py:329-363 (forward), :291-302 (getPrediction),
// :237-286 (sampleClean), :152-173 (PredictionNetwork.forward).
//
// The reference materialises, for each of the K prediction steps, a [b, 1+Nneg, W, H] candidate tensor
// (11.8 GB at b=64) and a same-size product.  Here one workgroup owns one (window b, frame t): it keeps
// the K predictions P_k = W_k c_t in LDS, GATHERS candidate rows of z (L2/MALL resident) into LDS in
// chunks, scores all K x candidates with the f32 MFMA and does the cross-entropy in place; only the
// logits (for backward), K partial losses and K hit flags leave the kernel.
//
// Candidate list of a workgroup: 16 "positive-tile" rows z[b][t+1 .. t+16] (the positive of step k is
// column k of that tile -- computed by the same MFMA chain as the negatives, so a negative that happens
// to be the positive frame ties EXACTLY, as in the reference) followed by the Nneg gathered negatives.
#include "common.h"

#include <algorithm>

namespace cpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NCE_ROWS = 16;   // MFMA M: predictions padded to 16 rows
constexpr int NCE_POS = 16;    // positive-tile columns

template <int H> struct NceCfg {
    static constexpr int NC = (H <= 256) ? 64 : 32;   // candidates per LDS chunk
    static constexpr int LD = H + 4;                   // padded LDS row (floats)
};

struct NceArgs {
    const float *P;        // [b*T][K*H]  predictions
    const float *z;        // [b*T][H]
    const int32_t *ext;    // [b][Nneg][W]
    const float *weights;  // [b*W] or null
    float *logits;         // [b*W][K][Nneg+1]
    float *lse;            // [b*W][K]
    float *lossp;          // [b*W][K]   w * CE
    float *hit;            // [b*W][K]   1 if argmax == 0
    int b, T, W, K, Nneg;
    int lw;                // LDS logits/dS row length (floats), multiple of 4
    // backward
    const float *dloss;    // [K]
    float *dP;             // [b*T][K*H]
    float *dz;             // [b*T][H]  (atomics)
    float inv_count;       // 1 / (b*W)
};

// gathers candidates [c0, c0+NC) of the workgroup's list into Cs (zero rows where the slot is empty)
template <int H> __device__ __forceinline__ void nce_gather(float *Cs, int *rowidx, const NceArgs &a, int bb, int t, int c0)
{
    constexpr int NC = NceCfg<H>::NC, LD = NceCfg<H>::LD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < NC; i += 4) {
        const int g = c0 + i;                    // global candidate index
        long row = -1;
        if (g < NCE_POS) {
            const int tt = t + 1 + g;
            if (g < a.K && tt < a.T) row = (long)bb * a.T + tt;
        } else if (g - NCE_POS < a.Nneg) {
            row = a.ext[((long)bb * a.Nneg + (g - NCE_POS)) * a.W + t];
        }
        if (lane == 0 && rowidx != nullptr) rowidx[i] = (int)row;
        for (int d4 = lane; d4 < H / 4; d4 += 64) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= 0) v = reinterpret_cast<const float4 *>(a.z + row * H)[d4];
            *reinterpret_cast<float4 *>(&Cs[i * LD + 4 * d4]) = v;
        }
    }
}

template <int H> __device__ __forceinline__ void nce_load_p(float *Ps, const NceArgs &a, long bt_row)
{
    constexpr int LD = NceCfg<H>::LD;
    const float *src = a.P + bt_row * a.K * H;
    for (int i = threadIdx.x; i < NCE_ROWS * (H / 4); i += blockDim.x) {
        const int k = i / (H / 4), d4 = i - k * (H / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < a.K) v = reinterpret_cast<const float4 *>(src + (long)k * H)[d4];
        *reinterpret_cast<float4 *>(&Ps[k * LD + 4 * d4]) = v;
    }
}

template <int H> __global__ __launch_bounds__(256) void infonce_fwd_kernel(NceArgs a)
{
    constexpr int NC = NceCfg<H>::NC, LD = NceCfg<H>::LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                          // [16][LD]
    float *Cs = Ps + NCE_ROWS * LD;            // [NC][LD]
    float *Ls = Cs + NC * LD;                  // [16][lw]   logits by global candidate index

    const int bt = blockIdx.x;
    const int bb = bt / a.W, t = bt - bb * a.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_h = 1.f / H;

    nce_load_p<H>(Ps, a, (long)bb * a.T + t);
    const int ncand = NCE_POS + a.Nneg;
    for (int c0 = 0; c0 < ncand; c0 += NC) {
        __syncthreads();                       // Cs free (and Ps visible on the first pass)
        nce_gather<H>(Cs, nullptr, a, bb, t, c0);
        __syncthreads();
        if (wave < NC / 16 && c0 + wave * 16 < ncand) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *pa = &Ps[(lane & 15) * LD + 4 * (lane >> 4)];
            const float *pb = &Cs[(wave * 16 + (lane & 15)) * LD + 4 * (lane >> 4)];
#pragma unroll 4
            for (int kk = 0; kk < H / 16; ++kk) {
                const float4 av = *reinterpret_cast<const float4 *>(pa + 16 * kk);
                const float4 bv = *reinterpret_cast<const float4 *>(pb + 16 * kk);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
            }
            // acc[r] = <P_k, cand_g>, k = 4*(lane>>4) + r, g = c0 + 16*wave + (lane&15)
            const int g = c0 + wave * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * (lane >> 4) + r;
                if (k < a.K && g < ncand) Ls[k * a.lw + g] = acc[r] * inv_h;
            }
        }
    }
    __syncthreads();

    // cross-entropy vs class 0 (the positive = column k of the positive tile), criterion.py:345-357
    const float wgt = a.weights != nullptr ? a.weights[bt] : 1.f;
    for (int k = wave; k < a.K; k += 4) {
        const float *lrow = Ls + k * a.lw;
        const float pos = lrow[k];
        float mneg = -INFINITY;
        for (int j = lane; j < a.Nneg; j += 64) mneg = fmaxf(mneg, lrow[NCE_POS + j]);
        for (int off = 32; off > 0; off >>= 1) mneg = fmaxf(mneg, __shfl_xor(mneg, off, 64));
        const float m = fmaxf(mneg, pos);
        float se = 0.f;
        for (int j = lane; j < a.Nneg; j += 64) se += expf(lrow[NCE_POS + j] - m);
        for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
        se += expf(pos - m);
        const float lse = m + logf(se);
        float *lg = a.logits + ((long)bt * a.K + k) * (a.Nneg + 1);
        for (int j = lane; j < a.Nneg; j += 64) lg[1 + j] = lrow[NCE_POS + j];
        if (lane == 0) {
            lg[0] = pos;
            a.lse[(long)bt * a.K + k] = lse;
            a.lossp[(long)bt * a.K + k] = wgt * (lse - pos);
            a.hit[(long)bt * a.K + k] = pos >= mneg ? 1.f : 0.f;    // first-index-wins argmax
        }
    }
}

// losses[k] = sum_i lossp[i][k] / count ; acc[k] = sum_i hit[i][k] / count     (one workgroup per output)
__global__ void infonce_reduce_kernel(const float *lossp, const float *hit, long rows, int K, float inv_count, float *losses, float *acc)
{
    __shared__ float red[256];
    const int k = blockIdx.x;
    const float *src = k < K ? lossp : hit;
    const int kk = k < K ? k : k - K;
    float s = 0.f;
    for (long r = threadIdx.x; r < rows; r += blockDim.x) s += src[r * K + kk];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) (k < K ? losses : acc)[kk] = red[0] * inv_count;
}

// Backward: grid over b*T; workgroups with t >= W only zero their dP row.
template <int H> __global__ __launch_bounds__(256) void infonce_bwd_kernel(NceArgs a)
{
    constexpr int NC = NceCfg<H>::NC, LD = NceCfg<H>::LD;
    constexpr int DP_TILES = (H / 16 + 3) / 4;       // 16-wide dP column tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                          // [16][LD]
    float *Cs = Ps + NCE_ROWS * LD;            // [NC][LD]
    float *dS = Cs + NC * LD;                  // [16][lw]   d loss / d <P_k, cand_g>
    int *rowidx = reinterpret_cast<int *>(dS + NCE_ROWS * a.lw);   // [NC]

    const int bb = blockIdx.x / a.T, t = blockIdx.x - bb * a.T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *dprow = a.dP + (long)blockIdx.x * a.K * H;
    if (t >= a.W) {
        for (int i = threadIdx.x; i < a.K * H / 4; i += blockDim.x)
            reinterpret_cast<float4 *>(dprow)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const long bt = (long)bb * a.W + t;
    const int ncand = NCE_POS + a.Nneg;
    const float inv_h = 1.f / H;
    const float wgt = (a.weights != nullptr ? a.weights[bt] : 1.f) * a.inv_count;

    nce_load_p<H>(Ps, a, (long)bb * a.T + t);
    // dS[k][g]: softmax - onehot, scaled by upstream grad, weight, 1/count and 1/H
    for (int i = threadIdx.x; i < NCE_ROWS * a.lw; i += blockDim.x) {
        const int k = i / a.lw, g = i - k * a.lw;
        float v = 0.f;
        if (k < a.K) {
            const float coef = a.dloss[k] * wgt * inv_h;
            const float *lg = a.logits + (bt * a.K + k) * (a.Nneg + 1);
            const float l = a.lse[bt * a.K + k];
            if (g < NCE_POS) {
                if (g == k) v = coef * (expf(lg[0] - l) - 1.f);
            } else if (g - NCE_POS < a.Nneg) {
                v = coef * expf(lg[1 + g - NCE_POS] - l);
            }
        }
        dS[i] = v;
    }

    f32x4 dp[DP_TILES];
#pragma unroll
    for (int i = 0; i < DP_TILES; ++i) dp[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < ncand; c0 += NC) {
        __syncthreads();
        nce_gather<H>(Cs, rowidx, a, bb, t, c0);
        __syncthreads();

        // dP[k][d] += sum_g dS[k][g] * cand_g[d]      (16x16x4: M = k, N = d tile, K = candidates)
#pragma unroll
        for (int i = 0; i < DP_TILES; ++i) {
            const int dt = wave + 4 * i;
            if (dt < H / 16) {
                for (int q = 0; q < NC / 16; ++q) {
                    const float4 av = *reinterpret_cast<const float4 *>(&dS[(lane & 15) * a.lw + c0 + 16 * q + 4 * (lane >> 4)]);
                    const float *pb = &Cs[(16 * q + 4 * (lane >> 4)) * LD + dt * 16 + (lane & 15)];
                    dp[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, pb[0], dp[i], 0, 0, 0);
                    dp[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, pb[LD], dp[i], 0, 0, 0);
                    dp[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, pb[2 * LD], dp[i], 0, 0, 0);
                    dp[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, pb[3 * LD], dp[i], 0, 0, 0);
                }
            }
        }

        // dCand_g[d] = sum_k dS[k][g] * P_k[d] -> atomicAdd into dz[row_g]   (32x32x2: M = cand, N = d, K = k)
        constexpr int NT = (NC / 32) * (H / 32);
        for (int tt = wave; tt < NT; tt += 4) {
            const int ct = tt % (NC / 32), dt = tt / (NC / 32);
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kp = 0; kp < NCE_ROWS / 2; ++kp) {
                const int kr = 2 * kp + (lane >> 5);
                const float av = dS[kr * a.lw + c0 + ct * 32 + (lane & 31)];
                const float bv = Ps[kr * LD + dt * 32 + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int row = rowidx[ci];
                if (row >= 0) atomicAdd(a.dz + (long)row * H + dt * 32 + (lane & 31), acc[e]);
            }
        }
    }

    // dp[i][r] = dP[k = 4*(lane>>4) + r][d = dt*16 + (lane&15)]
#pragma unroll
    for (int i = 0; i < DP_TILES; ++i) {
        const int dt = wave + 4 * i;
        if (dt < H / 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * (lane >> 4) + r;
                if (k < a.K) dprow[(long)k * H + dt * 16 + (lane & 15)] = dp[i][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct NceLayout {
    int b, T, K, W, Har, Henc, Nneg, lw;
    float *P, *logits, *lse;            // saved
    size_t saved_bytes;
    float *lossp, *hit, *dP, *wt, *tn;  // scratch
    size_t tn_bytes, scratch_bytes;
    size_t lds_fwd, lds_bwd;
};

static bool nce_supported(int H) { return H == 32 || H == 64 || H == 128 || H == 256 || H == 512; }

static int nce_layout(NceLayout &l, int b, int T, int K, int Har, int Henc, int Nneg, void *saved, void *scratch)
{
    CPC_REQUIRE(nce_supported(Henc), "infonce: encoder dim %d not supported (32, 64, 128, 256, 512)", Henc);
    CPC_REQUIRE(b > 0 && K >= 1 && K <= 16 && T > K && Nneg >= 1 && Har >= 1,
                "infonce: bad shape b=%d T=%d K=%d (1..16) dim_ar=%d n_neg=%d", b, T, K, Har, Nneg);
    l.b = b; l.T = T; l.K = K; l.W = T - K; l.Har = Har; l.Henc = Henc; l.Nneg = Nneg;
    const int nc = Henc <= 256 ? 64 : 32;
    l.lw = (int)cdiv(NCE_POS + Nneg, nc) * nc + 4;
    Carver sv(saved);
    l.P = sv.take<float>((size_t)b * T * K * Henc);
    l.logits = sv.take<float>((size_t)b * l.W * K * (Nneg + 1));
    l.lse = sv.take<float>((size_t)b * l.W * K);
    l.saved_bytes = sv.used();
    Carver sc(scratch);
    l.lossp = sc.take<float>((size_t)b * l.W * K);
    l.hit = sc.take<float>((size_t)b * l.W * K);
    l.dP = sc.take<float>((size_t)b * T * K * Henc);
    l.wt = sc.take<float>((size_t)K * Henc * Har);
    l.tn_bytes = gemm_tn_scratch_bytes(K * Henc, Har, (long)b * T);
    l.tn = sc.take<float>(l.tn_bytes / sizeof(float));
    l.scratch_bytes = sc.used();
    const size_t ld = Henc + 4;
    l.lds_fwd = sizeof(float) * ((NCE_ROWS + nc) * ld + (size_t)NCE_ROWS * l.lw);
    l.lds_bwd = l.lds_fwd + sizeof(int) * nc;
    CPC_REQUIRE(l.lds_bwd <= 160 * 1024, "infonce: n_neg=%d needs %zu B of LDS (> 160 KiB)", Nneg, l.lds_bwd);
    return CPC_OK;
}

template <typename Kern> static int allow_lds(Kern kern, size_t bytes)
{
    if (bytes > 64 * 1024)
        CPC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return CPC_OK;
}

#define NCE_DISPATCH(H, ...)                                    \
    switch (H) {                                                \
    case 32: { constexpr int HH = 32; __VA_ARGS__; } break;    \
    case 64: { constexpr int HH = 64; __VA_ARGS__; } break;    \
    case 128: { constexpr int HH = 128; __VA_ARGS__; } break;  \
    case 256: { constexpr int HH = 256; __VA_ARGS__; } break;  \
    case 512: { constexpr int HH = 512; __VA_ARGS__; } break;  \
    default: break;                                             \
    }

static int infonce_forward(const float *c, const float *z, const float *wpred, const int32_t *ext, const float *weights,
                           float *losses, float *acc, void *saved, void *scratch, int b, int T, int K, int Har, int Henc,
                           int Nneg, hipStream_t st)
{
    NceLayout l;
    CPC_TRY(nce_layout(l, b, T, K, Har, Henc, Nneg, saved, scratch));
    RowMap none{};
    // all K predictors in one GEMM: P[(b,t)][k*Henc + e] = sum_a c[b,t,a] * W_k[e][a]     (criterion.py:163)
    CPC_TRY(gemm_nt(c, Har, wpred, Har, l.P, (long)K * Henc, nullptr, (long)b * T, K * Henc, Har, none, st));
    NceArgs a{};
    a.P = l.P; a.z = z; a.ext = ext; a.weights = weights; a.logits = l.logits; a.lse = l.lse; a.lossp = l.lossp; a.hit = l.hit;
    a.b = b; a.T = T; a.W = l.W; a.K = K; a.Nneg = Nneg; a.lw = l.lw; a.inv_count = 1.f / ((float)b * l.W);
    int status = CPC_OK;
    {
        ProfScope prof(PROF_NCE_FWD, st);
        NCE_DISPATCH(Henc, {
            status = allow_lds(infonce_fwd_kernel<HH>, l.lds_fwd);
            if (status == CPC_OK) hipLaunchKernelGGL(infonce_fwd_kernel<HH>, dim3((unsigned)(b * l.W)), dim3(256), l.lds_fwd, st, a);
        });
    }
    CPC_TRY(status);
    CPC_CHECK_LAUNCH("infonce_fwd_kernel");
    hipLaunchKernelGGL(infonce_reduce_kernel, dim3(2 * K), dim3(256), 0, st, l.lossp, l.hit, (long)b * l.W, K, a.inv_count, losses, acc);
    CPC_CHECK_LAUNCH("infonce_reduce_kernel");
    return CPC_OK;
}

static int infonce_backward(const float *c, const float *z, const float *wpred, const int32_t *ext, const float *weights,
                            const float *dlosses, void *saved, void *scratch, float *dc, float *dz, float *dwpred, int b, int T,
                            int K, int Har, int Henc, int Nneg, hipStream_t st)
{
    NceLayout l;
    CPC_TRY(nce_layout(l, b, T, K, Har, Henc, Nneg, saved, scratch));
    CPC_CHECK_HIP(hipMemsetAsync(dz, 0, sizeof(float) * (size_t)b * T * Henc, st));
    NceArgs a{};
    a.P = l.P; a.z = z; a.ext = ext; a.weights = weights; a.logits = l.logits; a.lse = l.lse;
    a.b = b; a.T = T; a.W = l.W; a.K = K; a.Nneg = Nneg; a.lw = l.lw; a.inv_count = 1.f / ((float)b * l.W);
    a.dloss = dlosses; a.dP = l.dP; a.dz = dz;
    int status = CPC_OK;
    {
        ProfScope prof(PROF_NCE_BWD, st);
        NCE_DISPATCH(Henc, {
            status = allow_lds(infonce_bwd_kernel<HH>, l.lds_bwd);
            if (status == CPC_OK) hipLaunchKernelGGL(infonce_bwd_kernel<HH>, dim3((unsigned)(b * T)), dim3(256), l.lds_bwd, st, a);
        });
    }
    CPC_TRY(status);
    CPC_CHECK_LAUNCH("infonce_bwd_kernel");
    // dc = dP . W  (rows t >= W of dP are zero)
    CPC_TRY(transpose2d(wpred, l.wt, K * Henc, Har, st));                        // [Har][K*Henc]
    RowMap none{};
    CPC_TRY(gemm_nt(l.dP, (long)K * Henc, l.wt, (long)K * Henc, dc, Har, nullptr, (long)b * T, Har, K * Henc, none, st));
    // dW_k[e][a] = sum_{b,t} dP[(b,t)][k*Henc + e] * c[b,t,a]
    CPC_TRY(gemm_tn(l.dP, (long)K * Henc, c, Har, dwpred, Har, K * Henc, Har, (long)b * T, l.tn, l.tn_bytes, 0, 0, st));
    return CPC_OK;
}

}  // namespace cpc

extern "C" size_t cpc_infonce_saved_bytes(int b, int t, int k, int dim_ar, int dim_enc, int n_neg)
{
    cpc::NceLayout l;
    if (cpc::nce_layout(l, b, t, k, dim_ar, dim_enc, n_neg, nullptr, nullptr) != CPC_OK) return 0;
    return l.saved_bytes;
}

extern "C" size_t cpc_infonce_scratch_bytes(int b, int t, int k, int dim_ar, int dim_enc, int n_neg)
{
    cpc::NceLayout l;
    if (cpc::nce_layout(l, b, t, k, dim_ar, dim_enc, n_neg, nullptr, nullptr) != CPC_OK) return 0;
    return l.scratch_bytes;
}

extern "C" int cpc_infonce_forward(const float *c, const float *z, const float *wpred, const int32_t *ext_idx, const float *weights,
                                   float *losses, float *acc, void *saved, void *scratch, int b, int t, int k, int dim_ar,
                                   int dim_enc, int n_neg, cpc_stream_t stream)
{
    return cpc::infonce_forward(c, z, wpred, ext_idx, weights, losses, acc, saved, scratch, b, t, k, dim_ar, dim_enc, n_neg,
                                static_cast<hipStream_t>(stream));
}

extern "C" int cpc_infonce_backward(const float *c, const float *z, const float *wpred, const int32_t *ext_idx, const float *weights,
                                    const float *dlosses, void *saved, void *scratch, float *dc, float *dz, float *dwpred, int b,
                                    int t, int k, int dim_ar, int dim_enc, int n_neg, cpc_stream_t stream)
{
    return cpc::infonce_backward(c, z, wpred, ext_idx, weights, dlosses, saved, scratch, dc, dz, dwpred, b, t, k, dim_ar, dim_enc,
                                 n_neg, static_cast<hipStream_t>(stream));
}
