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

#include <algorithm>

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
    const int hp = std::max(64, (int)cdiv(H, 64) * 64);
    const int kq = std::max(1, std::min(1024 / hp, H / 4));
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
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq;
        {
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

static int gru_backward(const float *x, const float *const *prm, const float *dout, void *saved, void *scratch, float *dx,
                        float *const *grads, int N, int T, int Din, int H, int layers, hipStream_t st)
{
    GruLayout g;
    CPC_TRY(gru_layout(g, N, T, Din, H, layers, saved, scratch));
    const int hp = std::max(64, (int)cdiv(H, 64) * 64);
    const int kq = std::max(1, std::min(1024 / hp, H / 4));
    const float *dcur = dout;
    for (int l = layers - 1; l >= 0; --l) {
        const float *w_ih = prm[4 * l], *w_hh = prm[4 * l + 1];
        const float *xin = (l == 0) ? x : g.outl[l - 1];
        const int din = (l == 0) ? Din : H;
        hipLaunchKernelGGL(gru_pack_bwd_kernel, dim3(256), dim3(256), 0, st, w_hh, g.wpack, H);
        CPC_CHECK_LAUNCH("gru_pack_bwd_kernel");
        GruArgs a{};
        a.wpack = g.wpack; a.hall = g.hall[l]; a.gates = g.gates[l]; a.hn = g.hn[l];
        a.N = N; a.T = T; a.H = H; a.hp = hp; a.kq = kq;
        a.dout = dcur; a.dgi = g.dgi; a.dgh = g.dgh;
        {
            ProfScope prof(PROF_GRU_BWD, st);
            const size_t lds = sizeof(float) * ((size_t)3 * H + (size_t)kq * hp);
            hipLaunchKernelGGL(gru_bwd_kernel, dim3((unsigned)N), dim3(kq * hp), lds, st, a);
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
