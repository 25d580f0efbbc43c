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

#include <algorithm>

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
    const float4 *wpack;  // packed W_hh
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

struct LstmLayout {
    int N, T, Din, H, layers, G;
    // saved, per layer
    float *gates[8], *hall[8], *call[8], *outl[8];
    size_t saved_bytes;
    // scratch
    float *gi, *dgi, *dgh, *dxa, *dxb, *wt, *cs, *tn;
    float4 *wpack;
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
    g.wt = sc.take<float>((size_t)G * H * dmax);
    g.wpack = sc.take<float4>((size_t)G * H * H / 4);
    g.cs = sc.take<float>(colsum_rows_scratch_bytes(G * H) / sizeof(float));
    g.tn_bytes = std::max(gemm_tn_scratch_bytes(G * H, H, (long)N * (T + 1)), gemm_tn_scratch_bytes(G * H, dmax, (long)N * T));
    g.tn_bytes = std::max(g.tn_bytes, gemm_tn_scratch_bytes(G * H, Din, (long)N * T));
    g.tn = sc.take<float>(g.tn_bytes / sizeof(float));
    g.scratch_bytes = sc.used();
    return CPC_OK;
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
        CPC_TRY(gemm_nt(xin, din, w_ih, din, g.gi, (long)G * H, b_ih, (long)N * T, G * H, din, none, st));
        hipLaunchKernelGGL(lstm_pack_fwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H, G);
        CPC_CHECK_LAUNCH("lstm_pack_fwd_kernel");
        LstmArgs a{};
        a.gi = g.gi; a.wpack = g.wpack; a.bhh = b_hh;
        a.h0 = h0 ? h0 + (size_t)l * N * H : nullptr;
        a.c0 = c0 ? c0 + (size_t)l * N * H : nullptr;
        a.out = (l + 1 < layers) ? g.outl[l] : out;
        a.hall = g.hall[l]; a.call = g.call[l]; a.gates = g.gates[l];
        a.hlast = h_last ? h_last + (size_t)l * N * H : nullptr;
        a.clast = c_last ? c_last + (size_t)l * N * H : nullptr;
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq;
        const size_t lds = sizeof(float) * (cdiv(H, 4) * 4 + (size_t)kq * G * hp);
        hipLaunchKernelGGL(lstm_fwd_kernel<G>, dim3((unsigned)N), dim3(kq * hp), lds, st, a);
        CPC_CHECK_LAUNCH("lstm_fwd_kernel");
        xin = a.out;
        din = H;
    }
    return CPC_OK;
}

template <int G>
static int lstm_backward(const float *x, const float *const *prm, const float *dout, void *saved, void *scratch, float *dx,
                         float *const *grads, int N, int T, int Din, int H, int layers, hipStream_t st)
{
    LstmLayout g;
    CPC_TRY(lstm_layout(g, G, N, T, Din, H, layers, saved, scratch));
    int hp, kq;
    lstm_threads(H, G, hp, kq);
    const float *dcur = dout;
    for (int l = layers - 1; l >= 0; --l) {
        const float *w_ih = prm[4 * l], *w_hh = prm[4 * l + 1];
        const float *xin = (l == 0) ? x : g.outl[l - 1];
        const int din = (l == 0) ? Din : H;
        hipLaunchKernelGGL(lstm_pack_bwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H, G);
        CPC_CHECK_LAUNCH("lstm_pack_bwd_kernel");
        LstmArgs a{};
        a.wpack = g.wpack; a.hall = g.hall[l]; a.call = g.call[l]; a.gates = g.gates[l];
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq;
        a.dout = dcur; a.dgi = g.dgi; a.dgh = g.dgh;
        const size_t lds = sizeof(float) * ((size_t)G * H + (size_t)kq * hp);
        hipLaunchKernelGGL(lstm_bwd_kernel<G>, dim3((unsigned)N), dim3(kq * hp), lds, st, a);
        CPC_CHECK_LAUNCH("lstm_bwd_kernel");
        const int GH = G * H;
        // dW_hh[g][k] = sum_{n,t} dG[n,t][g] * h_{t-1}[n][k]   (hall row t is h_{t-1}; row T of dGH is zero)
        CPC_TRY(gemm_tn(g.dgh, GH, g.hall[l], H, grads[4 * l + 1], H, GH, H, (long)N * (T + 1), g.tn, g.tn_bytes, 0, 0, st));
        CPC_TRY(colsum_rows(g.dgh, GH, (long)N * (T + 1), GH, grads[4 * l + 3], g.cs, st));
        // dW_ih[g][k] = sum dG[n,t][g] * x[n,t][k]
        CPC_TRY(gemm_tn(g.dgi, GH, xin, din, grads[4 * l], din, GH, din, (long)N * T, g.tn, g.tn_bytes, 0, 0, st));
        CPC_TRY(colsum_rows(g.dgi, GH, (long)N * T, GH, grads[4 * l + 2], g.cs, st));
        // dX = dG . W_ih
        float *dxl = (l == 0) ? dx : ((l % 2) ? g.dxa : g.dxb);
        if (dxl != nullptr) {
            CPC_TRY(transpose2d(w_ih, g.wt, GH, din, st));                         // [din][G*H]
            RowMap none{};
            CPC_TRY(gemm_nt(g.dgi, GH, g.wt, GH, dxl, din, nullptr, (long)N * T, din, GH, none, st));
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
    return cpc::lstm_forward<4>(x, params, h0, c0, out, h_last, c_last, saved, scratch, n, t, dim_in, hidden, layers,
                                static_cast<hipStream_t>(stream));
}

extern "C" int cpc_lstm_backward(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                 float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                 cpc_stream_t stream)
{
    return cpc::lstm_backward<4>(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                                 static_cast<hipStream_t>(stream));
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
    return cpc::lstm_forward<1>(x, params, h0, nullptr, out, h_last, nullptr, saved, scratch, n, t, dim_in, hidden, layers,
                                static_cast<hipStream_t>(stream));
}

extern "C" int cpc_rnn_backward(const float *x, const float *const *params, const float *dout, void *saved, void *scratch,
                                float *dx, float *const *grads, int n, int t, int dim_in, int hidden, int layers,
                                cpc_stream_t stream)
{
    return cpc::lstm_backward<1>(x, params, dout, saved, scratch, dx, grads, n, t, dim_in, hidden, layers,
                                 static_cast<hipStream_t>(stream));
}
