// Small HBM-bound helpers: error state, column sums, weight re-layouts, channel-first ChannelNorm
// (standalone module API), fused flat Adam.
#include "common.h"
#include "coop.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
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
                                             "conv0_fwd", "conv0_bwd", "gemm_planes_nt", "gemm_planes_tn", "side_wait"};

ProfScope::ProfScope(int slot, hipStream_t st, bool attached) : attached_(attached), slot_(slot), st_(st), active_(false)
{
    const int on = g_prof_on.load(std::memory_order_relaxed);
    if (on == 0 || (on == 2 && slot != PROF_PLANES_NT && slot != PROF_SIDE_WAIT) || (on == 3 && slot != PROF_GEMM_NT && slot != PROF_SIDE_WAIT)) return;
    if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) return;
    active_ = attached_ ? true : hipEventRecord(a_, st_) == hipSuccess;
}

void ProfScope::cancel()
{
    if (!active_) return;
    active_ = false;
    (void)hipEventDestroy(a_);
    (void)hipEventDestroy(b_);
}

ProfScope::~ProfScope()
{
    if (!active_) return;
    if (!attached_) (void)hipEventRecord(b_, st_);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({slot_, a_, b_});
}

// ---------------------------------------------------------------- asynchronous error word (coop.h)
static int *g_err_host = nullptr, *g_err_dev = nullptr;
static std::once_flag g_err_once;
int *coop_error_word()
{
    std::call_once(g_err_once, [] {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_err_host = static_cast<int *>(h);
        g_err_dev = static_cast<int *>(d);
        *g_err_host = 0;
    });
    return g_err_dev;
}
// ---- granule memory of the cooperative recurrent kernels (coop.h)
struct CoopComm { gu64_t *p = nullptr; size_t bytes = 0; unsigned next = 0; unsigned long used = 0; };
constexpr size_t COOP_COMM_STREAMS = 8;     // buffers kept: one per (device, stream) that launched a recurrent kernel lately
static std::mutex g_coop_comm_mu;
static std::map<std::pair<int, hipStream_t>, CoopComm> g_coop_comm;
int coop_comm_buffers()
{
    std::lock_guard<std::mutex> lk(g_coop_comm_mu);
    return (int)g_coop_comm.size();
}
int coop_comm_acquire(size_t bytes, int T, hipStream_t st, gu64_t **comm, unsigned *epoch0)
{
    std::mutex &mu = g_coop_comm_mu;
    std::map<std::pair<int, hipStream_t>, CoopComm> &bufs = g_coop_comm;
    int dev = 0;
    CPC_CHECK_HIP(hipGetDevice(&dev));
    static unsigned long tick = 0;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(dev, st);
    if (bufs.find(key) == bufs.end() && bufs.size() >= COOP_COMM_STREAMS) {
        // a process that keeps making streams (a stream per request, say) would otherwise keep a buffer for each of them for good:
        // the one unused for longest goes (hipFree waits for the device: nothing still polls it)
        auto old = bufs.begin();
        for (auto it = bufs.begin(); it != bufs.end(); ++it)
            if (it->second.used < old->second.used) old = it;
        if (old->second.p != nullptr) {
            int cur = dev;
            if (old->first.first != cur) CPC_CHECK_HIP(hipSetDevice(old->first.first));
            const hipError_t fe = hipFree(old->second.p);
            if (old->first.first != cur) CPC_CHECK_HIP(hipSetDevice(cur));
            CPC_CHECK_HIP(fe);
        }
        bufs.erase(old);
    }
    CoopComm &c = bufs[key];
    c.used = ++tick;
    if (c.bytes < bytes) {
        // (rare: the first launch on this stream, or a larger shape.  hipFree waits for the device, so nothing still polls the old one)
        if (c.p != nullptr) CPC_CHECK_HIP(hipFree(c.p));
        c.p = nullptr; c.bytes = 0;
        const size_t want = align_up(bytes, (size_t)1 << 20);
        CPC_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&c.p), want));
        CPC_CHECK_HIP(hipMemsetAsync(c.p, 0, want, st));
        c.bytes = want; c.next = 0;
        const char *v = getenv("CPC_COOP_EPOCH_START");        // tests: a new buffer starts counting here (the wrap is 2^32 launches away otherwise)
        if (v != nullptr) c.next = (unsigned)strtoul(v, nullptr, 0);
    }
    if ((unsigned long long)c.next + (unsigned)T + 2ull > 0xFFFFFFF0ull) {       // the 32-bit epoch would wrap: start over
        CPC_CHECK_HIP(hipMemsetAsync(c.p, 0, c.bytes, st));
        c.next = 0;
    }
    *comm = c.p;
    *epoch0 = c.next;
    c.next += (unsigned)T + 1u;
    return CPC_OK;
}

// cooperative policy of the process (cpc_coop_set_policy): 0 = cooperative kernels wherever they fit, 1 = streaming kernels only
static std::atomic<int> g_coop_policy{0};
bool coop_allowed() { return g_coop_policy.load(std::memory_order_relaxed) == 0; }
static std::atomic<long> g_coop_launches{0};
void coop_count_launch() { g_coop_launches.fetch_add(1, std::memory_order_relaxed); }
long coop_launches() { return g_coop_launches.load(std::memory_order_relaxed); }
static std::atomic<long> g_rec_bwd_calls{0};
void coop_count_backward_call() { g_rec_bwd_calls.fetch_add(1, std::memory_order_relaxed); }

int coop_error_take(const char *where)
{
    if (g_err_host == nullptr) return CPC_OK;
    const int code = __atomic_exchange_n(g_err_host, 0, __ATOMIC_ACQ_REL);
    if (code == 0) return CPC_OK;
    if (code == COOP_ERR_NONFINITE_GRAD) {
        set_error("%s: the Adam step met non-finite gradient elements and skipped THOSE elements (their parameters and moments are "
                  "unchanged); the finite elements of the same step were applied and the step count advanced -- a partial update, "
                  "unlike torch.optim.Adam, which would have propagated the NaN: reload the last checkpoint or continue knowingly", where);
        return CPC_ERR_HIP;
    }
    if (code == COOP_ERR_BAD_INDEX) {
        set_error("%s: the criterion was given negative-sample indices outside [0, batch * frames); they were replaced by row 0 "
                  "(no out-of-bounds gather took place) -- the loss of that step is meaningless", where);
        return CPC_ERR_HIP;
    }
    // whatever kept the workgroups from being resident together (another tenant on the device) is likely to still be there: from
    // here on this process takes the streaming kernels, which have no such requirement (cpc_coop_set_policy(0) restores the default)
    g_coop_policy.store(1, std::memory_order_relaxed);
    set_error("%s: a cooperative recurrent kernel (%s pass) gave up waiting for the other workgroups of its group -- its "
              "workgroups were not all resident at once (another kernel on the device?); its outputs are NaN (the fused Adam "
              "step skips non-finite gradient elements: the parameters are intact).  This process now uses the streaming "
              "(non-cooperative) recurrent kernels; cpc_coop_set_policy(0) restores the default", where,
              code == COOP_ERR_FWD_WAIT ? "forward" : "backward");
    return CPC_ERR_HIP;
}
int coop_fault_injection()
{
    const char *v = getenv("CPC_COOP_FAULT");       // read every time: tests switch it on and off
    return v != nullptr && v[0] == '1';
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

// Two-stage column sums for tall partial-sum matrices: stage 1 spreads the rows over COLSUM_SPLIT x (width/64)
// blocks, stage 2 adds the COLSUM_SPLIT partial rows and scatters column c to outs[c / seg][c % seg] (the
// gamma / beta / bias gradients of a layer leave one partial matrix for three different tensors).
__global__ void colsum_stage1_kernel(const float *part, long rows, long ld, int width, float *scratch)
{
    __shared__ float red[16][65];
    const int c = blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    if (c < width) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const long step = (long)gridDim.y * 16;
        long r = (long)blockIdx.y * 16 + threadIdx.y;
        for (; r + 3 * step < rows; r += 4 * step) {
            s0 += part[r * ld + c];
            s1 += part[(r + step) * ld + c];
            s2 += part[(r + 2 * step) * ld + c];
            s3 += part[(r + 3 * step) * ld + c];
        }
        for (; r < rows; r += step) s0 += part[r * ld + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < width) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
        scratch[(long)blockIdx.y * width + c] = t;
    }
}

__global__ void colsum_stage2_kernel(const float *scratch, int split, int width, float *out0, float *out1, float *out2, int seg)
{
    // thread (x, y): column blockIdx.x*64 + x, partial rows y, y + 4, ...; the 4 row groups meet in LDS
    __shared__ float red[4][65];
    const int c = blockIdx.x * 64 + threadIdx.x;
    float t = 0.f;
    if (c < width)
        for (int i = threadIdx.y; i < split; i += 4) t += scratch[(long)i * width + c];
    red[threadIdx.y][threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.y == 0 && c < width) {
        t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const int which = c / seg, off = c - which * seg;
        float *out = which == 0 ? out0 : (which == 1 ? out1 : out2);
        out[off] = t;
    }
}

size_t colsum_split_scratch_bytes(int width) { return align_up((size_t)COLSUM_SPLIT * width * sizeof(float), 256); }

int colsum_split(const float *part, long rows, long ld, int width, float *out0, float *out1, float *out2, int seg,
                 void *scratch, hipStream_t st)
{
    CPC_REQUIRE(seg > 0 && width <= 3 * seg, "colsum_split: width %d does not fit three segments of %d", width, seg);
    float *sc = static_cast<float *>(scratch);
    hipLaunchKernelGGL(colsum_stage1_kernel, dim3((unsigned)cdiv(width, 64), COLSUM_SPLIT), dim3(64, 16), 0, st, part, rows, ld, width, sc);
    CPC_CHECK_LAUNCH("colsum_stage1_kernel");
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((unsigned)cdiv(width, 64)), dim3(64, 4), 0, st, sc, COLSUM_SPLIT, width, out0, out1, out2, seg);
    CPC_CHECK_LAUNCH("colsum_stage2_kernel");
    return CPC_OK;
}

constexpr int CS_BLOCKS = 1024;

// stage 1: block b sums rows b, b+CS_BLOCKS, ... ; threads over columns (coalesced), four independent chains
__global__ void colsum_rows_stage1(const float *a, long ld, long rows, int width, float *part)
{
    const long step = gridDim.x;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long r = blockIdx.x;
        for (; r + 3 * step < rows; r += 4 * step) {
            s0 += a[r * ld + c];
            s1 += a[(r + step) * ld + c];
            s2 += a[(r + 2 * step) * ld + c];
            s3 += a[(r + 3 * step) * ld + c];
        }
        for (; r < rows; r += step) s0 += a[r * ld + c];
        part[(long)blockIdx.x * width + c] = (s0 + s1) + (s2 + s3);
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

// ---------------------------------------------------------------- transposes
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
// A gradient element that is not finite (a cooperative recurrent kernel that timed out poisons its outputs with NaN, and
// through the loss every gradient; with data parallelism the all-reduce carries it to every rank) is NOT applied: parameter
// and moments of that element stay as they are and the asynchronous error word is set, so the next cpc_async_error_check /
// recurrent entry point reports it -- the weights are still the ones of the last good step.
__global__ void adam_kernel(float *p, const float *g, float *m, float *v, long n, float lr_c1, float rsqrt_c2,
                            float beta1, float beta2, float eps, float grad_scale, int *err)
{
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        if (!(fabsf(gi) <= 3.4028234e38f)) { bad = true; continue; }
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * rsqrt_c2 + eps;
        p[i] -= lr_c1 * (mi / denom);
    }
    if (bad) coop_report(err, COOP_ERR_NONFINITE_GRAD);
}

// ---------------------------------------------------------------- streams that run BESIDE a given stream
// A HIP stream is served by one of a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default, per priority); the runtime hands a new
// stream the least-used queue AT THAT MOMENT, and two streams on one queue run one after the other -- whatever their events say.
// Which queue a stream of this library's got used to be an accident of creation order: in a process with a process group
// (torch's stream pools, RCCL's stream) the side stream could land on the training stream's own queue, and the ~1.1 ms of
// deferred work per step then ran IN FRONT of the backward pass instead of beside it (5.86 against 4.9 ms per step, round 5's
// unexplained record; profiles/r06_process_group_queues.md).  A stream made here is TESTED: a kernel that spins for 200 us on
// each stream to avoid, a one-wave kernel on the candidate -- if the candidate's finishes while the spin is still running the two
// are on different queues.  Candidates that fail are kept alive (they hold their queue's use count) until one passes.
// Sharing a queue serialises ANY two streams (measured pair by pair: profiles/r06_process_group_queues.md).
__global__ void queue_probe_spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();                 // 100 MHz, constant
    for (int i = 0; i < (1 << 24) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}
__global__ void queue_probe_tag_kernel() {}

static std::mutex g_streams_mu;
static std::vector<std::pair<int, hipStream_t>> g_apart;       // (device, stream) of every stream made by stream_create_apart
static std::atomic<long> g_apart_fail{0};

// 1: kernels of `b` run beside kernels of `a`; 0: behind them (same hardware queue, or a blocking stream beside the null stream)
static int streams_overlap_locked(hipStream_t a, hipStream_t b, int *overlap)
{
    hipEvent_t spun = nullptr, tagged = nullptr;
    CPC_CHECK_HIP(hipEventCreateWithFlags(&spun, hipEventDisableTiming));
    CPC_CHECK_HIP(hipEventCreateWithFlags(&tagged, hipEventDisableTiming));
    hipLaunchKernelGGL(queue_probe_spin_kernel, dim3(1), dim3(64), 0, a, 20000LL);
    hipError_t e = hipEventRecord(spun, a);
    if (e == hipSuccess) { hipLaunchKernelGGL(queue_probe_tag_kernel, dim3(1), dim3(64), 0, b); e = hipEventRecord(tagged, b); }
    if (e == hipSuccess) e = hipEventSynchronize(tagged);
    if (e == hipSuccess) {
        const hipError_t q = hipEventQuery(spun);
        *overlap = q == hipErrorNotReady ? 1 : 0;
        if (q != hipErrorNotReady && q != hipSuccess) e = q;
    }
    (void)hipGetLastError();                               // (hipErrorNotReady is not an error)
    if (e == hipSuccess) e = hipEventSynchronize(spun);
    (void)hipEventDestroy(spun);
    (void)hipEventDestroy(tagged);
    CPC_CHECK_HIP(e);
    return CPC_OK;
}

int stream_create_apart(const hipStream_t *avoid, int n_avoid, hipStream_t *out)
{
    int dev = 0;
    CPC_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_streams_mu);
    // (only the streams the caller names: with four hardware queues the library's three streams -- side, sampler, data-parallel
    //  helper -- can each sit apart from the training stream, not all four from each other; nothing of theirs needs that)
    std::vector<hipStream_t> others(avoid, avoid + n_avoid);
    std::vector<hipStream_t> rejected;
    hipStream_t cand = nullptr;
    bool ok = false;
    for (int attempt = 0; attempt < 12 && !ok; ++attempt) {
        CPC_CHECK_HIP(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
        ok = true;
        for (hipStream_t o : others) {
            int overlap = 0;
            CPC_TRY(streams_overlap_locked(o, cand, &overlap));
            if (!overlap) { ok = false; break; }
        }
        if (!ok) rejected.push_back(cand);
    }
    if (!ok) {                       // (a saturated device can fail every test: the last candidate serves, and the count says so)
        g_apart_fail.fetch_add(1, std::memory_order_relaxed);
        rejected.pop_back();
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    g_apart.emplace_back(dev, cand);
    *out = cand;
    return CPC_OK;
}

}  // namespace cpc

extern "C" int cpc_stream_create_apart(const cpc_stream_t *avoid, int n_avoid, cpc_stream_t *out)
{
    CPC_REQUIRE(out != nullptr && n_avoid >= 0 && (n_avoid == 0 || avoid != nullptr), "cpc_stream_create_apart: bad arguments");
    std::vector<hipStream_t> av(n_avoid);
    for (int i = 0; i < n_avoid; ++i) av[i] = static_cast<hipStream_t>(avoid[i]);
    hipStream_t st = nullptr;
    CPC_TRY(cpc::stream_create_apart(av.data(), n_avoid, &st));
    *out = st;
    return CPC_OK;
}

extern "C" int cpc_streams_overlap(cpc_stream_t a, cpc_stream_t b)
{
    std::lock_guard<std::mutex> lk(cpc::g_streams_mu);
    int overlap = 0;
    const int rc = cpc::streams_overlap_locked(static_cast<hipStream_t>(a), static_cast<hipStream_t>(b), &overlap);
    return rc != CPC_OK ? rc : overlap;
}

// a kernel that keeps one wave of `stream` busy for `ticks` of the 100 MHz clock (diagnostics: bench.py tests whether RCCL's stream,
// which it cannot name, runs beside the training stream)
extern "C" int cpc_stream_spin(cpc_stream_t stream, long ticks)
{
    CPC_REQUIRE(ticks > 0 && ticks <= 100000000L, "cpc_stream_spin: ticks out of range");
    hipLaunchKernelGGL(cpc::queue_probe_spin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (long long)ticks);
    CPC_CHECK_LAUNCH("queue_probe_spin_kernel");
    return CPC_OK;
}

extern "C" long cpc_stream_apart_failures(void) { return cpc::g_apart_fail.load(std::memory_order_relaxed); }

extern "C" int cpc_version(void) { return 106; }     // 106: round 6 (cpc_stream_create_apart, cpc_negidx_wait_on, ...); 105: round 5 (cpc_encoder_forward2 / backward2, cpc_coop_set_policy, cpc_side_tail_wait, ...)

extern "C" int cpc_prof_enable(int on)
{
    cpc::g_prof_on.store(on == 2 ? 2 : (on ? 1 : 0));
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
                       lr_c1, rsqrt_c2, beta1, beta2, eps, grad_scale, cpc::coop_error_word());
    CPC_CHECK_LAUNCH("adam_kernel");
    return CPC_OK;
}

extern "C" int cpc_async_error_check(cpc_stream_t stream)
{
    CPC_CHECK_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return cpc::coop_error_take("cpc_async_error_check");
}

// Cooperative recurrent launches (GRU / LSTM kernels that need every workgroup resident) issued by this process so far: the
// data-parallel glue asserts its collectives are ordered behind them (cpc2_amd/train.py, DataParallelContext.attach)
extern "C" long cpc_coop_launches(void) { return cpc::coop_launches(); }
extern "C" int cpc_coop_comm_buffers(void) { return cpc::coop_comm_buffers(); }
extern "C" long cpc_recurrent_backward_calls(void) { return cpc::g_rec_bwd_calls.load(std::memory_order_relaxed); }
extern "C" int cpc_coop_set_policy(int policy)
{
    if (policy < 0) return cpc::g_coop_policy.load(std::memory_order_relaxed);
    return cpc::g_coop_policy.exchange(policy != 0 ? 1 : 0, std::memory_order_relaxed);
}
