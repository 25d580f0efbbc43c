// Small HBM-bound helpers: error state, column sums, weight re-layouts, channel-first ChannelNorm
// (standalone module API), fused flat Adam.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace cpc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- in-situ kernel timing (bench.py)
// When enabled, launchers bracket selected kernels with hipEvents on the launch stream; bench.py reads
// the summed durations after its timed region.  Disabled (the default) it costs one relaxed load.
static std::atomic<int> g_prof_on{0};
struct ProfRec { int slot; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static const char *kProfNames[PROF_SLOTS] = {"gemm_nt", "gemm_tn", "infonce_fwd", "infonce_bwd", "gru_fwd", "gru_bwd",
                                             "conv0_fwd", "conv0_bwd"};

ProfScope::ProfScope(int slot, hipStream_t st) : slot_(slot), st_(st), active_(false)
{
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) return;
    active_ = hipEventRecord(a_, st_) == hipSuccess;
}

ProfScope::~ProfScope()
{
    if (!active_) return;
    (void)hipEventRecord(b_, st_);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({slot_, a_, b_});
}

// ---------------------------------------------------------------- column sums
// block (64 columns x 16 row lanes): coalesced along columns, rows strided over threadIdx.y, LDS tree over y
__global__ void colsum_kernel(const float *part, long rows, long ld, int width, float *out)
{
    __shared__ float red[16][65];
    const int c = blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    if (c < width) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // 4 independent chains: loads of a thread overlap
        long r = threadIdx.y;
        for (; r + 48 < rows; r += 64) {
            s0 += part[r * ld + c];
            s1 += part[(r + 16) * ld + c];
            s2 += part[(r + 32) * ld + c];
            s3 += part[(r + 48) * ld + c];
        }
        for (; r < rows; r += 16) s0 += part[r * ld + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < width) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
        out[c] = t;
    }
}

int colsum(const float *part, long rows, long ld, int width, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(width, 64)), dim3(64, 16), 0, st, part, rows, ld, width, out);
    CPC_CHECK_LAUNCH("colsum_kernel");
    return CPC_OK;
}

constexpr int CS_BLOCKS = 256;

// stage 1: block b sums rows b, b+CS_BLOCKS, ... ; threads over columns (coalesced)
__global__ void colsum_rows_stage1(const float *a, long ld, long rows, int width, float *part)
{
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float s = 0.f;
        for (long r = blockIdx.x; r < rows; r += gridDim.x) s += a[r * ld + c];
        part[(long)blockIdx.x * width + c] = s;
    }
}

size_t colsum_rows_scratch_bytes(int width) { return align_up((size_t)CS_BLOCKS * width * sizeof(float), 256); }

int colsum_rows(const float *a, long ld, long rows, int width, float *out, void *scratch, hipStream_t st)
{
    float *part = static_cast<float *>(scratch);
    hipLaunchKernelGGL(colsum_rows_stage1, dim3(CS_BLOCKS), dim3(256), 0, st, a, ld, rows, width, part);
    CPC_CHECK_LAUNCH("colsum_rows_stage1");
    return colsum(part, CS_BLOCKS, width, width, out, st);
}

// ---------------------------------------------------------------- weight re-layouts
// Conv1d weight w[co][ci][j] -> wr[co][j*cin + ci]  (the B operand of the implicit GEMM)
__global__ void permute_conv_fwd_kernel(const float *w, float *wr, int cout, int cin, int k)
{
    const long total = (long)cout * cin * k;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx / ((long)cin * k));
        const int rem = (int)(idx - (long)co * cin * k);
        const int j = rem / cin, ci = rem - j * cin;
        wr[idx] = w[((long)co * cin + ci) * k + j];
    }
}

int permute_conv_fwd(const float *w, float *wr, int cout, int cin, int k, hipStream_t st)
{
    const long total = (long)cout * cin * k;
    hipLaunchKernelGGL(permute_conv_fwd_kernel, dim3((unsigned)std::min<long>(cdiv(total, 256), 2048)), dim3(256), 0, st, w, wr, cout, cin, k);
    CPC_CHECK_LAUNCH("permute_conv_fwd_kernel");
    return CPC_OK;
}

// Backward-data operand, kernel k == 2*s: for phase j in [0,s):
//   bd[j][ci][kk] = w[kk][ci][j+s]        for kk <  cout   (pairs with du[t_hi-1])
//                 = w[kk-cout][ci][j]      for kk >= cout   (pairs with du[t_hi])
__global__ void permute_conv_dgrad_kernel(const float *w, float *bd, int cout, int cin, int k, int s)
{
    const long total = (long)s * cin * 2 * cout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(idx % (2 * cout));
        const long r = idx / (2 * cout);
        const int ci = (int)(r % cin);
        const int j = (int)(r / cin);
        const int co = kk < cout ? kk : kk - cout;
        const int tap = kk < cout ? j + s : j;
        bd[idx] = w[((long)co * cin + ci) * k + tap];
    }
}

int permute_conv_dgrad(const float *w, float *bd, int cout, int cin, int k, int s, hipStream_t st)
{
    CPC_REQUIRE(k == 2 * s, "permute_conv_dgrad needs kernel == 2*stride (got k=%d s=%d)", k, s);
    const long total = (long)s * cin * 2 * cout;
    hipLaunchKernelGGL(permute_conv_dgrad_kernel, dim3((unsigned)std::min<long>(cdiv(total, 256), 2048)), dim3(256), 0, st, w, bd, cout, cin, k, s);
    CPC_CHECK_LAUNCH("permute_conv_dgrad_kernel");
    return CPC_OK;
}

__global__ void transpose2d_kernel(const float *a, float *at, int rows, int cols)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? a[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows) at[(long)c * rows + r] = tile[threadIdx.x][i];
    }
}

int transpose2d(const float *a, float *at, int rows, int cols, hipStream_t st)
{
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), dim3(32, 8), 0, st, a, at, rows, cols);
    CPC_CHECK_LAUNCH("transpose2d_kernel");
    return CPC_OK;
}

// ---------------------------------------------------------------- ChannelNorm, channel-first x[N,C,L]
// One thread per (n,l) column; lanes run along l so every access is coalesced.
__global__ void channelnorm_cf_fwd_kernel(const float *x, const float *w, const float *b, float *y, float *rstd_save,
                                          int N, int C, int L, float eps)
{
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= (long)N * L) return;
    const int n = (int)(col / L), l = (int)(col - (long)n * L);
    const float *xp = x + (long)n * C * L + l;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += xp[(long)c * L];
    const float mean = s / C;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) { const float d = xp[(long)c * L] - mean; ss += d * d; }
    const float rstd = rsqrtf(ss / (C - 1) + eps);
    rstd_save[col] = rstd;
    float *yp = y + (long)n * C * L + l;
    for (int c = 0; c < C; ++c) {
        float v = (xp[(long)c * L] - mean) * rstd;
        if (w != nullptr) v = v * w[c] + b[c];
        yp[(long)c * L] = v;
    }
}

// dx per column; dw/db via atomics on [C] (dw, db zeroed by the launcher)
__global__ void channelnorm_cf_bwd_kernel(const float *x, const float *w, const float *dy, const float *rstd_save,
                                          float *dx, float *dw, float *db, int N, int C, int L)
{
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = col < (long)N * L;
    const int n = ok ? (int)(col / L) : 0, l = ok ? (int)(col - (long)n * L) : 0;
    const float *xp = x + (long)n * C * L + l;
    const float *gp = dy + (long)n * C * L + l;
    float mean = 0.f, rstd = 0.f, s1 = 0.f, s2 = 0.f;
    if (ok) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += xp[(long)c * L];
        mean = s / C;
        rstd = rstd_save[col];
        for (int c = 0; c < C; ++c) {
            const float xh = (xp[(long)c * L] - mean) * rstd;
            const float g = gp[(long)c * L] * (w != nullptr ? w[c] : 1.f);
            s1 += g;
            s2 += g * xh;
        }
    }
    for (int c = 0; c < C; ++c) {
        float gy = 0.f, xh = 0.f;
        if (ok) {
            xh = (xp[(long)c * L] - mean) * rstd;
            gy = gp[(long)c * L];
            const float g = gy * (w != nullptr ? w[c] : 1.f);
            dx[(long)n * C * L + (long)c * L + l] = rstd * (g - s1 / C - xh * s2 / (C - 1));
        }
        if (dw != nullptr) {
            // wave-level reduction, one atomic per wave per channel
            float a = gy * xh, bsum = gy;
            for (int off = 32; off > 0; off >>= 1) {
                a += __shfl_down(a, off, 64);
                bsum += __shfl_down(bsum, off, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&dw[c], a);
                atomicAdd(&db[c], bsum);
            }
        }
    }
}

// ---------------------------------------------------------------- window gather (feeder)
// out[i][0..L) = audio[offsets[i] .. offsets[i]+L): the training windows are contiguous slices of ONE flat,
// HBM-resident audio buffer (dataset.py:308-321 slices the same way on the host)
__global__ void window_gather_kernel(const float *audio, long total, const long *offsets, float *out, int b, int L)
{
    const int i = blockIdx.y;
    const long off = offsets[i];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < L; k += gridDim.x * blockDim.x) {
        const long src = off + k;
        out[(long)i * L + k] = (src >= 0 && src < total) ? audio[src] : 0.f;
    }
}

// ---------------------------------------------------------------- Adam
__global__ void adam_kernel(float *p, const float *g, float *m, float *v, long n, float lr_c1, float rsqrt_c2,
                            float beta1, float beta2, float eps, float grad_scale)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * rsqrt_c2 + eps;
        p[i] -= lr_c1 * (mi / denom);
    }
}

}  // namespace cpc

extern "C" int cpc_version(void) { return 100; }

extern "C" int cpc_prof_enable(int on)
{
    cpc::g_prof_on.store(on ? 1 : 0);
    return CPC_OK;
}

// Sums (and releases) the finished records of kernel class `name`; synchronises on their events.
extern "C" int cpc_prof_read(const char *name, double *total_ms, long *count)
{
    int slot = -1;
    for (int i = 0; i < cpc::PROF_SLOTS; ++i)
        if (std::strcmp(name, cpc::kProfNames[i]) == 0) slot = i;
    CPC_REQUIRE(slot >= 0 && total_ms != nullptr && count != nullptr, "cpc_prof_read: unknown kernel class '%s'", name);
    std::lock_guard<std::mutex> lk(cpc::g_prof_mu);
    double tot = 0.0;
    long n = 0;
    std::vector<cpc::ProfRec> keep;
    for (auto &r : cpc::g_prof) {
        if (r.slot != slot) { keep.push_back(r); continue; }
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; ++n; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    cpc::g_prof.swap(keep);
    *total_ms = tot;
    *count = n;
    return CPC_OK;
}
extern "C" const char *cpc_last_error(void) { return cpc::g_err; }

extern "C" int cpc_channelnorm_forward(const float *x, const float *w, const float *b, float *y, float *rstd_save,
                                       int N, int C, int L, float eps, cpc_stream_t stream)
{
    CPC_REQUIRE(N > 0 && C > 1 && L > 0, "channelnorm: bad shape N=%d C=%d L=%d", N, C, L);
    const long cols = (long)N * L;
    hipLaunchKernelGGL(cpc::channelnorm_cf_fwd_kernel, dim3((unsigned)cpc::cdiv(cols, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, w, b, y, rstd_save, N, C, L, eps);
    CPC_CHECK_LAUNCH("channelnorm_cf_fwd_kernel");
    return CPC_OK;
}

extern "C" int cpc_channelnorm_backward(const float *x, const float *w, const float *dy, const float *rstd_save, float *dx,
                                        float *dw, float *db, int N, int C, int L, float eps, cpc_stream_t stream)
{
    (void)eps;
    CPC_REQUIRE(N > 0 && C > 1 && L > 0, "channelnorm: bad shape N=%d C=%d L=%d", N, C, L);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dw != nullptr) {
        CPC_CHECK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * C, st));
        CPC_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * C, st));
    }
    const long cols = (long)N * L;
    hipLaunchKernelGGL(cpc::channelnorm_cf_bwd_kernel, dim3((unsigned)cpc::cdiv(cols, 256)), dim3(256), 0, st, x, w, dy,
                       rstd_save, dx, dw, db, N, C, L);
    CPC_CHECK_LAUNCH("channelnorm_cf_bwd_kernel");
    return CPC_OK;
}

extern "C" int cpc_window_gather(const float *audio, long total_samples, const long *offsets, float *out, int batch, int window,
                                 cpc_stream_t stream)
{
    CPC_REQUIRE(audio != nullptr && offsets != nullptr && out != nullptr && batch > 0 && window > 0 && total_samples > 0,
                "window_gather: bad arguments (batch=%d window=%d)", batch, window);
    hipLaunchKernelGGL(cpc::window_gather_kernel, dim3((unsigned)std::min<long>(cpc::cdiv(window, 256), 64), (unsigned)batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), audio, total_samples, offsets, out, batch, window);
    CPC_CHECK_LAUNCH("window_gather_kernel");
    return CPC_OK;
}

extern "C" int cpc_adam_step(float *p, const float *g, float *m, float *v, long n, int step, float lr, float beta1,
                             float beta2, float eps, float grad_scale, cpc_stream_t stream)
{
    CPC_REQUIRE(n > 0 && step >= 1, "adam: bad n=%ld step=%d", n, step);
    const double c1 = 1.0 - std::pow((double)beta1, step);
    const double c2 = 1.0 - std::pow((double)beta2, step);
    const float lr_c1 = (float)(lr / c1);
    const float rsqrt_c2 = (float)(1.0 / std::sqrt(c2));
    const long blocks = std::min<long>(cpc::cdiv(n, 256), 4096);
    hipLaunchKernelGGL(cpc::adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n,
                       lr_c1, rsqrt_c2, beta1, beta2, eps, grad_scale);
    CPC_CHECK_LAUNCH("adam_kernel");
    return CPC_OK;
}
