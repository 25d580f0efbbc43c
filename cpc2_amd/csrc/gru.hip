// CPCAR (mode="GRU") on gfx950.  Reference: /root/reference/cpc/model.py:158-207 -> torch.nn.GRU
// (batch_first, gate order r,z,n):
//     r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)        z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//     n = tanh(W_in x + b_in + r * (W_hn h + b_hn))      h' = (1 - z) * n + z * h
//
// Forward per layer: one MFMA GEMM for all input projections GI = X W_ih^T + b_ih, then ONE
// persistent kernel for the T sequential steps.  Windows are independent, so each workgroup owns
// GRU_NB windows for the whole sequence (no inter-workgroup synchronisation); thread j owns hidden
// unit j, keeps h in LDS and streams W_hh (re-laid out so that lanes read consecutive float4s)
// from L2 every step.  Backward mirrors it (BPTT), then three GEMMs give dW_hh, dW_ih and dX.
#include "common.h"

#include <algorithm>

namespace cpc {

constexpr int GRU_NB = 2;   // windows per workgroup

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
    const float *bhh;     // [3H]
    const float *h0;      // [N][H] or null
    float *out;           // [N][T][H]
    float *hall;          // [N][T+1][H]  row 0 = h0, row t+1 = h_t
    float *gates;         // [N*T][3H]    r, z, n
    float *hn;            // [N*T][H]     W_hn h + b_hn
    float *hlast;         // [N][H] or null
    int N, T, H;
    // backward
    const float *dout;    // [N][T][H]
    float *dgi;           // [N*T][3H]
    float *dgh;           // [N][T+1][3H], row T zero
};

__global__ void gru_fwd_kernel(GruArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hs[GRU_NB][H]
    const int H = a.H, T = a.T;
    const int j = threadIdx.x;
    const bool act = j < H;
    const int n0 = blockIdx.x * GRU_NB;
    float hprev[GRU_NB];
#pragma unroll
    for (int s = 0; s < GRU_NB; ++s) {
        const int n = n0 + s;
        hprev[s] = (act && n < a.N && a.h0 != nullptr) ? a.h0[(long)n * H + j] : 0.f;
        if (act) {
            smem[s * H + j] = hprev[s];
            if (n < a.N) a.hall[((long)n * (T + 1)) * H + j] = hprev[s];
        }
    }
    float bh[3] = {0.f, 0.f, 0.f};
    if (act) { bh[0] = a.bhh[j]; bh[1] = a.bhh[H + j]; bh[2] = a.bhh[2 * H + j]; }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        float acc[GRU_NB][3];
#pragma unroll
        for (int s = 0; s < GRU_NB; ++s) acc[s][0] = acc[s][1] = acc[s][2] = 0.f;
        if (act) {
            const float4 *wp = a.wpack + j;
            for (int k4 = 0; k4 < H / 4; ++k4) {
                const float4 wr = wp[(long)(k4 * 3 + 0) * H];
                const float4 wz = wp[(long)(k4 * 3 + 1) * H];
                const float4 wn = wp[(long)(k4 * 3 + 2) * H];
#pragma unroll
                for (int s = 0; s < GRU_NB; ++s) {
                    const float4 h4 = reinterpret_cast<const float4 *>(smem + s * H)[k4];
                    acc[s][0] = fmaf(wr.x, h4.x, fmaf(wr.y, h4.y, fmaf(wr.z, h4.z, fmaf(wr.w, h4.w, acc[s][0]))));
                    acc[s][1] = fmaf(wz.x, h4.x, fmaf(wz.y, h4.y, fmaf(wz.z, h4.z, fmaf(wz.w, h4.w, acc[s][1]))));
                    acc[s][2] = fmaf(wn.x, h4.x, fmaf(wn.y, h4.y, fmaf(wn.z, h4.z, fmaf(wn.w, h4.w, acc[s][2]))));
                }
            }
        }
        float hnew[GRU_NB];
#pragma unroll
        for (int s = 0; s < GRU_NB; ++s) {
            const int n = n0 + s;
            hnew[s] = 0.f;
            if (act && n < a.N) {
                const long row = (long)n * T + t;
                const float *g = a.gi + row * 3 * H;
                const float ghn = acc[s][2] + bh[2];
                const float r = sigmoidf_(g[j] + acc[s][0] + bh[0]);
                const float z = sigmoidf_(g[H + j] + acc[s][1] + bh[1]);
                const float c = tanhf(g[2 * H + j] + r * ghn);
                const float hv = (1.f - z) * c + z * hprev[s];
                hnew[s] = hv;
                float *gs = a.gates + row * 3 * H;
                gs[j] = r; gs[H + j] = z; gs[2 * H + j] = c;
                a.hn[row * H + j] = ghn;
                a.out[row * H + j] = hv;
                a.hall[((long)n * (T + 1) + t + 1) * H + j] = hv;
            }
        }
        __syncthreads();            // every thread is done reading the old h
        if (act) {
#pragma unroll
            for (int s = 0; s < GRU_NB; ++s) { smem[s * H + j] = hnew[s]; hprev[s] = hnew[s]; }
        }
        __syncthreads();
    }
    if (act && a.hlast != nullptr) {
#pragma unroll
        for (int s = 0; s < GRU_NB; ++s)
            if (n0 + s < a.N) a.hlast[(long)(n0 + s) * H + j] = hprev[s];
    }
}

__global__ void gru_bwd_kernel(GruArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // dgh[GRU_NB][3H]
    const int H = a.H, T = a.T;
    const int j = threadIdx.x;
    const bool act = j < H;
    const int n0 = blockIdx.x * GRU_NB;
    float carry[GRU_NB];
#pragma unroll
    for (int s = 0; s < GRU_NB; ++s) {
        carry[s] = 0.f;
        const int n = n0 + s;
        if (act && n < a.N) {                                   // zero junk row T of dGH
            float *zr = a.dgh + ((long)n * (T + 1) + T) * 3 * H;
            zr[j] = 0.f; zr[H + j] = 0.f; zr[2 * H + j] = 0.f;
        }
    }
    for (int t = T - 1; t >= 0; --t) {
        float keep[GRU_NB];
#pragma unroll
        for (int s = 0; s < GRU_NB; ++s) {
            const int n = n0 + s;
            float dpr = 0.f, dpz = 0.f, dhn = 0.f;
            keep[s] = 0.f;
            if (act && n < a.N) {
                const long row = (long)n * T + t;
                const float dh = a.dout[row * H + j] + carry[s];
                const float *gs = a.gates + row * 3 * H;
                const float r = gs[j], z = gs[H + j], c = gs[2 * H + j];
                const float hnv = a.hn[row * H + j];
                const float hp = a.hall[((long)n * (T + 1) + t) * H + j];
                const float dc = dh * (1.f - z);
                const float dz = dh * (hp - c);
                const float dpn = dc * (1.f - c * c);
                dpr = dpn * hnv * r * (1.f - r);
                dpz = dz * z * (1.f - z);
                dhn = dpn * r;
                keep[s] = dh * z;
                float *gi = a.dgi + row * 3 * H;
                gi[j] = dpr; gi[H + j] = dpz; gi[2 * H + j] = dpn;
                float *gh = a.dgh + ((long)n * (T + 1) + t) * 3 * H;
                gh[j] = dpr; gh[H + j] = dpz; gh[2 * H + j] = dhn;
            }
            if (act) {
                smem[s * 3 * H + j] = dpr;
                smem[s * 3 * H + H + j] = dpz;
                smem[s * 3 * H + 2 * H + j] = dhn;
            }
        }
        __syncthreads();
        if (act) {
            float acc[GRU_NB];
#pragma unroll
            for (int s = 0; s < GRU_NB; ++s) acc[s] = 0.f;
            const float4 *wp = a.wpack + j;
            for (int g4 = 0; g4 < 3 * H / 4; ++g4) {
                const float4 w4 = wp[(long)g4 * H];
#pragma unroll
                for (int s = 0; s < GRU_NB; ++s) {
                    const float4 d4 = reinterpret_cast<const float4 *>(smem + s * 3 * H)[g4];
                    acc[s] = fmaf(w4.x, d4.x, fmaf(w4.y, d4.y, fmaf(w4.z, d4.z, fmaf(w4.w, d4.w, acc[s]))));
                }
            }
#pragma unroll
            for (int s = 0; s < GRU_NB; ++s) carry[s] = keep[s] + acc[s];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
struct GruLayout {
    int N, T, Din, H, layers;
    // saved, per layer
    float *gates[8], *hn[8], *hall[8], *outl[8];
    size_t saved_bytes;
    // scratch
    float *gi, *dgi, *dgh, *dxa, *dxb, *wt, *cs, *tn;
    float4 *wpack;
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
    g.wt = sc.take<float>((size_t)3 * H * dmax);
    g.wpack = sc.take<float4>((size_t)3 * H * H / 4);
    g.cs = sc.take<float>(colsum_rows_scratch_bytes(3 * H) / sizeof(float));
    g.tn_bytes = std::max(gemm_tn_scratch_bytes(3 * H, H, (long)N * (T + 1)), gemm_tn_scratch_bytes(3 * H, dmax, (long)N * T));
    g.tn_bytes = std::max(g.tn_bytes, gemm_tn_scratch_bytes(3 * H, Din, (long)N * T));
    g.tn = sc.take<float>(g.tn_bytes / sizeof(float));
    g.scratch_bytes = sc.used();
    return CPC_OK;
}

static int gru_forward(const float *x, const float *const *prm, const float *h0, float *out, float *h_last, void *saved,
                       void *scratch, int N, int T, int Din, int H, int layers, hipStream_t st)
{
    GruLayout g;
    CPC_TRY(gru_layout(g, N, T, Din, H, layers, saved, scratch));
    const int threads = std::max(64, (int)cdiv(H, 64) * 64);
    const float *xin = x;
    int din = Din;
    for (int l = 0; l < layers; ++l) {
        const float *w_ih = prm[4 * l], *w_hh = prm[4 * l + 1], *b_ih = prm[4 * l + 2], *b_hh = prm[4 * l + 3];
        RowMap none{};
        CPC_TRY(gemm_nt(xin, din, w_ih, din, g.gi, 3L * H, b_ih, (long)N * T, 3 * H, din, none, st));
        hipLaunchKernelGGL(gru_pack_fwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H);
        CPC_CHECK_LAUNCH("gru_pack_fwd_kernel");
        GruArgs a{};
        a.gi = g.gi; a.wpack = g.wpack; a.bhh = b_hh;
        a.h0 = h0 ? h0 + (size_t)l * N * H : nullptr;
        a.out = (l + 1 < layers) ? g.outl[l] : out;
        a.hall = g.hall[l]; a.gates = g.gates[l]; a.hn = g.hn[l];
        a.hlast = h_last ? h_last + (size_t)l * N * H : nullptr;
        a.N = N; a.T = T; a.H = H;
        {
            ProfScope prof(PROF_GRU_FWD, st);
            hipLaunchKernelGGL(gru_fwd_kernel, dim3((unsigned)cdiv(N, GRU_NB)), dim3(threads), sizeof(float) * GRU_NB * H, st, a);
        }
        CPC_CHECK_LAUNCH("gru_fwd_kernel");
        xin = a.out;
        din = H;
    }
    return CPC_OK;
}

static int gru_backward(const float *x, const float *const *prm, const float *dout, void *saved, void *scratch, float *dx,
                        float *const *grads, int N, int T, int Din, int H, int layers, hipStream_t st)
{
    GruLayout g;
    CPC_TRY(gru_layout(g, N, T, Din, H, layers, saved, scratch));
    const int threads = std::max(64, (int)cdiv(H, 64) * 64);
    const float *dcur = dout;
    for (int l = layers - 1; l >= 0; --l) {
        const float *w_ih = prm[4 * l], *w_hh = prm[4 * l + 1];
        const float *xin = (l == 0) ? x : g.outl[l - 1];
        const int din = (l == 0) ? Din : H;
        hipLaunchKernelGGL(gru_pack_bwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H);
        CPC_CHECK_LAUNCH("gru_pack_bwd_kernel");
        GruArgs a{};
        a.wpack = g.wpack; a.hall = g.hall[l]; a.gates = g.gates[l]; a.hn = g.hn[l];
        a.N = N; a.T = T; a.H = H;
        a.dout = dcur; a.dgi = g.dgi; a.dgh = g.dgh;
        {
            ProfScope prof(PROF_GRU_BWD, st);
            hipLaunchKernelGGL(gru_bwd_kernel, dim3((unsigned)cdiv(N, GRU_NB)), dim3(threads), sizeof(float) * GRU_NB * 3 * H, st, a);
        }
        CPC_CHECK_LAUNCH("gru_bwd_kernel");

        // dW_hh[g][k] = sum_{n,t} dGH[n,t][g] * h_{t-1}[n][k]   (hall row t is h_{t-1}; row T of dGH is zero)
        CPC_TRY(gemm_tn(g.dgh, 3L * H, g.hall[l], H, grads[4 * l + 1], H, 3 * H, H, (long)N * (T + 1), g.tn, g.tn_bytes, 0, 0, st));
        CPC_TRY(colsum_rows(g.dgh, 3L * H, (long)N * (T + 1), 3 * H, grads[4 * l + 3], g.cs, st));
        // dW_ih[g][k] = sum dGI[n,t][g] * x[n,t][k]
        CPC_TRY(gemm_tn(g.dgi, 3L * H, xin, din, grads[4 * l], din, 3 * H, din, (long)N * T, g.tn, g.tn_bytes, 0, 0, st));
        CPC_TRY(colsum_rows(g.dgi, 3L * H, (long)N * T, 3 * H, grads[4 * l + 2], g.cs, st));
        // dX = dGI . W_ih
        float *dxl = (l == 0) ? dx : ((l % 2) ? g.dxa : g.dxb);
        if (dxl != nullptr) {
            CPC_TRY(transpose2d(w_ih, g.wt, 3 * H, din, st));                      // [din][3H]
            RowMap none{};
            CPC_TRY(gemm_nt(g.dgi, 3L * H, g.wt, 3L * H, dxl, din, nullptr, (long)N * T, din, 3 * H, none, st));
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
    return cpc::gru_forward(x, params, h0, out, h_last, saved, scratch, n, t, dim_in, hidden, layers, static_cast<hipStream_t>(stream));
}

extern "C" int cpc_gru_backward(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                cpc_stream_t stream)
{
    return cpc::gru_backward(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                             static_cast<hipStream_t>(stream));
}
