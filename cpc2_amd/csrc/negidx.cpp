// Host-side negative-index sampler, bit-exact with the reference's CPU path.
// Reference: /root/reference/cpc/criterion/criterion.py:247-266 (two torch.randint calls on the CPU
// generator = 32-bit MT19937, one output per element in order, value = out % range + low;
// batchIdx first, then seqIdx; extIdx = (seqIdx + t) % T + batchIdx * T with t fastest).
//
// The generator state is kept in torch's bookkeeping form (left / next) so that it can be seeded from,
// and written back to, torch.get_rng_state() -- the criterion then consumes the SAME stream the
// reference would.  Whole 624-word blocks are twisted and tempered in bulk.
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/cpc2_hip.h"

namespace cpc { void set_error(const char *fmt, ...); }

struct cpc_mt19937 {
    uint32_t mt[624];
    int left;   // torch: twist when --left == 0
    int next;
    std::vector<uint32_t> tmp;
    // ONE worker thread per generator, started by the first asynchronous call and parked on a condition variable between jobs
    // (until round 5 every call started a thread and -- for the device forms -- created and destroyed a HIP stream: the stream's
    // hardware queue was then whatever the runtime's least-used one was at that moment, in a process with a process group the
    // training stream's own; profiles/r06_process_group_queues.md).  At most one job in flight; every other entry point waits.
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool busy = false, quit = false;
    int worker_status = 0;
    // device forms: a stream of the worker's own (created apart from the training stream's hardware queue) and the event that
    // marks "step i + 1's words / indices are on the device"
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int stream_device = -1;
    bool device_pending = false;      // `done` has been recorded and nobody has waited for it yet
};

namespace {
constexpr int N = 624, M = 397;

inline void twist(uint32_t *mt)
{
    auto mix = [](uint32_t u, uint32_t v) { return (u & 0x80000000u) | (v & 0x7fffffffu); };
    auto mag = [](uint32_t y) { return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); };
    int i = 0;
    for (; i < N - M; ++i) mt[i] = mt[i + M] ^ mag(mix(mt[i], mt[i + 1]));
    for (; i < N - 1; ++i) mt[i] = mt[i + M - N] ^ mag(mix(mt[i], mt[i + 1]));
    mt[N - 1] = mt[M - 1] ^ mag(mix(mt[N - 1], mt[0]));
}

inline uint32_t temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// n raw outputs into dst, same stream as torch's mt19937::operator()
void draw(cpc_mt19937 *g, uint32_t *dst, size_t n)
{
    size_t pos = 0;
    while (pos < n) {
        if (g->left == 1) {          // next draw would twist
            twist(g->mt);
            g->left = N + 1;
            g->next = 0;
        }
        size_t take = (size_t)(g->left - 1);
        if (take > n - pos) take = n - pos;
        const uint32_t *src = g->mt + g->next;
        for (size_t i = 0; i < take; ++i) dst[pos + i] = temper(src[i]);
        g->next += (int)take;
        g->left -= (int)take;
        pos += take;
    }
}
}  // namespace

extern "C" cpc_mt19937 *cpc_mt_create(uint32_t seed)
{
    cpc_mt19937 *g = new (std::nothrow) cpc_mt19937();
    if (g != nullptr) cpc_mt_seed(g, seed);
    return g;
}

// (cpc_stream_create_apart lives in the device half of the library; a host-only build of this file -- the sanitizer test -- links without it)
extern "C" int cpc_stream_create_apart(const cpc_stream_t *avoid, int n_avoid, cpc_stream_t *out) __attribute__((weak));

namespace {
void worker_loop(cpc_mt19937 *g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    for (;;) {
        g->cv.wait(lk, [g] { return g->quit || (g->busy && g->job); });
        if (g->quit) return;
        std::function<int()> job = std::move(g->job);
        g->job = nullptr;
        lk.unlock();
        const int st = job();
        lk.lock();
        g->worker_status = st;
        g->busy = false;
        g->cv.notify_all();
    }
}

// wait for the job in flight (if any); returns its status once
int join_job(cpc_mt19937 *g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [g] { return !g->busy; });
    const int st = g->worker_status;
    g->worker_status = 0;
    return st;
}

// wait for the job in flight without taking its status (the next cpc_negidx_wait still reports it)
void wait_idle(cpc_mt19937 *g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [g] { return !g->busy; });
}

void submit(cpc_mt19937 *g, std::function<int()> job)
{
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [g] { return !g->busy; });
    if (!g->worker.joinable()) g->worker = std::thread(worker_loop, g);
    g->job = std::move(job);
    g->busy = true;
    g->cv.notify_all();
}

// the worker's stream and event on `device` (worker thread only)
hipError_t worker_stream(cpc_mt19937 *g, int device, cpc_stream_t caller_stream)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess || (g->stream != nullptr && g->stream_device == device)) return e;
    if (g->stream != nullptr) { (void)hipStreamDestroy(g->stream); g->stream = nullptr; }
    if (g->done != nullptr) { (void)hipEventDestroy(g->done); g->done = nullptr; }
    if (cpc_stream_create_apart != nullptr) {
        cpc_stream_t st = nullptr;
        if (cpc_stream_create_apart(&caller_stream, 1, &st) != CPC_OK) return hipErrorUnknown;
        g->stream = static_cast<hipStream_t>(st);
    } else {
        e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
    g->stream_device = device;
    return e;
}
}  // namespace

extern "C" void cpc_mt_destroy(cpc_mt19937 *g)
{
    if (g == nullptr) return;
    (void)join_job(g);
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
        g->cv.notify_all();
    }
    if (g->worker.joinable()) g->worker.join();
    if (g->stream != nullptr) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    if (g->done != nullptr) (void)hipEventDestroy(g->done);
    delete g;
}

extern "C" int cpc_mt_seed(cpc_mt19937 *g, uint32_t seed)
{
    if (g == nullptr) { cpc::set_error("cpc_mt_seed: null generator"); return CPC_ERR_INVALID; }
    (void)join_job(g);
    g->mt[0] = seed;
    for (int i = 1; i < N; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->left = 1;
    g->next = 0;
    return CPC_OK;
}

extern "C" int cpc_mt_get_state(const cpc_mt19937 *g, uint32_t *mt624, int *left, int *next)
{
    if (g == nullptr || mt624 == nullptr) { cpc::set_error("cpc_mt_get_state: null argument"); return CPC_ERR_INVALID; }
    wait_idle(const_cast<cpc_mt19937 *>(g));               // (a draw in flight owns the generator until it is done)
    std::memcpy(mt624, g->mt, sizeof(g->mt));
    if (left) *left = g->left;
    if (next) *next = g->next;
    return CPC_OK;
}

extern "C" int cpc_mt_set_state(cpc_mt19937 *g, const uint32_t *mt624, int left, int next)
{
    if (g == nullptr || mt624 == nullptr || left < 1 || left > N || next < 0 || next > N || (left > 1 && next + left - 1 != N)) {
        cpc::set_error("cpc_mt_set_state: invalid state (left=%d next=%d)", left, next);
        return CPC_ERR_INVALID;
    }
    wait_idle(g);
    std::memcpy(g->mt, mt624, sizeof(g->mt));
    g->left = left;
    g->next = next;
    return CPC_OK;
}

namespace {
// a % d for 32-bit a, exact: quotient estimate in double (53-bit mantissa, error <= 1) + one correction.
// Branch-free and auto-vectorisable, unlike the hardware 32-bit divide.
struct FastMod {
    uint32_t d;
    double inv;
    explicit FastMod(uint32_t dd) : d(dd), inv(1.0 / (double)dd) {}
    inline uint32_t operator()(uint32_t a) const
    {
        const uint32_t q = (uint32_t)((double)a * inv);
        int64_t r = (int64_t)a - (int64_t)q * d;
        r += (r < 0) ? (int64_t)d : 0;
        r -= (r >= (int64_t)d) ? (int64_t)d : 0;
        return (uint32_t)r;
    }
};

int sample_impl(cpc_mt19937 *g, int batch, int seq_len, int window, int n_neg, int time_major, int32_t *ext,
                int64_t *batch_idx_opt, int64_t *seq_idx_opt)
{
    const size_t n = (size_t)n_neg * window * batch;
    g->tmp.resize(2 * n);
    uint32_t *raw_b = g->tmp.data(), *raw_s = raw_b + n;
    draw(g, raw_b, n);      // batchIdx stream first (criterion.py:247-250)
    draw(g, raw_s, n);      // then seqIdx (criterion.py:253-256)
    const FastMod mod_b((uint32_t)batch), mod_s((uint32_t)(seq_len - 1));
    const uint32_t T = (uint32_t)seq_len;
    // in place: raw_b <- batchIdx * T, raw_s <- seqIdx (two flat, vectorisable passes)
    for (size_t i = 0; i < n; ++i) raw_b[i] = mod_b(raw_b[i]);
    for (size_t i = 0; i < n; ++i) raw_s[i] = mod_s(raw_s[i]) + 1u;
    if (batch_idx_opt) for (size_t i = 0; i < n; ++i) batch_idx_opt[i] = (int64_t)raw_b[i];
    if (seq_idx_opt) for (size_t i = 0; i < n; ++i) seq_idx_opt[i] = (int64_t)raw_s[i];
    // draw order: i = (bb*n_neg + nn)*window + t   (criterion.py:259-263)
    size_t i = 0;
    for (size_t row = 0; row < (size_t)n_neg * batch; ++row) {
        const size_t bb = row / (size_t)n_neg, nn = row % (size_t)n_neg;
        int32_t *dst = time_major ? ext + bb * (size_t)window * n_neg + nn : ext + i;
        const size_t step = time_major ? (size_t)n_neg : 1;
        for (uint32_t t = 0; t < (uint32_t)window; ++t, ++i) {
            uint32_t seq = raw_s[i] + t;
            seq -= (seq >= T) ? T : 0u;                      // seqIdx <= T-1, t <= T-1 -> one wrap at most
            dst[t * step] = (int32_t)(seq + raw_b[i] * T);
        }
    }
    return CPC_OK;
}

int check_args(cpc_mt19937 *g, int batch, int seq_len, int window, int n_neg, const int32_t *ext)
{
    if (g == nullptr || ext == nullptr || batch < 1 || seq_len < 2 || window < 1 || window > seq_len || n_neg < 1) {
        cpc::set_error("cpc_negidx_sample_host: bad arguments (batch=%d seq_len=%d window=%d n_neg=%d)", batch, seq_len, window, n_neg);
        return CPC_ERR_INVALID;
    }
    if ((long long)batch * seq_len > 2147483647LL) {
        cpc::set_error("cpc_negidx_sample_host: batch*seq_len exceeds int32");
        return CPC_ERR_INVALID;
    }
    return CPC_OK;
}
}  // namespace

extern "C" int cpc_negidx_wait(cpc_mt19937 *g)
{
    if (g == nullptr) { cpc::set_error("cpc_negidx_wait: null generator"); return CPC_ERR_INVALID; }
    const int st = join_job(g);
    if (g->device_pending) {          // host-blocking form: the device has the words when this returns
        g->device_pending = false;
        if (hipEventSynchronize(g->done) != hipSuccess && st == CPC_OK) { cpc::set_error("cpc_negidx_wait: event wait failed"); return CPC_ERR_HIP; }
    }
    return st;
}

// the worker's stream (NULL before its first device job), for diagnostics
extern "C" int cpc_negidx_stream(cpc_mt19937 *g, cpc_stream_t *out)
{
    if (g == nullptr || out == nullptr) { cpc::set_error("cpc_negidx_stream: null argument"); return CPC_ERR_INVALID; }
    (void)join_job(g);
    *out = g->stream;
    return CPC_OK;
}

// The same hand-over WITHOUT blocking the host on the device: waits for the worker's host part (the draw and the enqueue of
// copy + expansion), then makes `stream` wait for the event behind them.
extern "C" int cpc_negidx_wait_on(cpc_mt19937 *g, cpc_stream_t stream)
{
    if (g == nullptr) { cpc::set_error("cpc_negidx_wait_on: null generator"); return CPC_ERR_INVALID; }
    const int st = join_job(g);
    if (g->device_pending) {
        g->device_pending = false;
        if (hipStreamWaitEvent(static_cast<hipStream_t>(stream), g->done, 0) != hipSuccess && st == CPC_OK) {
            cpc::set_error("cpc_negidx_wait_on: hipStreamWaitEvent failed");
            return CPC_ERR_HIP;
        }
    }
    return st;
}

extern "C" int cpc_negidx_sample_host(cpc_mt19937 *g, int batch, int seq_len, int window, int n_neg, int time_major,
                                      int32_t *ext_idx_host, int64_t *batch_idx_host_opt, int64_t *seq_idx_host_opt)
{
    const int st = check_args(g, batch, seq_len, window, n_neg, ext_idx_host);
    if (st != CPC_OK) return st;
    cpc_negidx_wait(g);      // an asynchronous sample (if any) owns the generator until it is done
    return sample_impl(g, batch, seq_len, window, n_neg, time_major, ext_idx_host, batch_idx_host_opt, seq_idx_host_opt);
}

// Raw generator outputs (tempered 32-bit words, the values torch.randint reduces modulo its range): the
// sequential, bit-exact part of the sampler.  cpc_negidx_expand (device) turns 2*n of them into extIdx.
extern "C" int cpc_mt_draw_host(cpc_mt19937 *g, uint32_t *raw_host, size_t n)
{
    if (g == nullptr || raw_host == nullptr) { cpc::set_error("cpc_mt_draw_host: null argument"); return CPC_ERR_INVALID; }
    cpc_negidx_wait(g);
    draw(g, raw_host, n);
    return CPC_OK;
}

extern "C" int cpc_mt_draw_host_async(cpc_mt19937 *g, uint32_t *raw_host, size_t n)
{
    if (g == nullptr || raw_host == nullptr) { cpc::set_error("cpc_mt_draw_host_async: null argument"); return CPC_ERR_INVALID; }
    cpc_negidx_wait(g);
    submit(g, [=] { draw(g, raw_host, n); return (int)CPC_OK; });
    return CPC_OK;
}

// (cpc_negidx_expand lives in the device half of the library; a host-only build of this file -- the sanitizer test -- links without it)
extern "C" int cpc_negidx_expand(const uint32_t *raw, int32_t *ext_idx, int batch, int seq_len, int window, int n_neg,
                                 cpc_stream_t stream) __attribute__((weak));

// draw on the worker thread, upload from the (pinned) staging buffer on the worker's stream and -- ext_dev != nullptr -- expand
// there too; `done` is recorded behind it: cpc_negidx_wait (host) / cpc_negidx_wait_on (a stream) hand the result over
static int device_job(cpc_mt19937 *g, uint32_t *raw_host, uint32_t *raw_dev, int32_t *ext_dev, size_t n, int device,
                      int batch, int seq_len, int window, int n_neg, cpc_stream_t caller_stream, const char *who)
{
    draw(g, raw_host, n);
    hipError_t e = worker_stream(g, device, caller_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(raw_dev, raw_host, n * sizeof(uint32_t), hipMemcpyHostToDevice, g->stream);
    int rc = CPC_OK;
    if (e == hipSuccess && ext_dev != nullptr) rc = cpc_negidx_expand(raw_dev, ext_dev, batch, seq_len, window, n_neg, g->stream);
    if (e == hipSuccess) e = hipEventRecord(g->done, g->stream);
    if (e == hipSuccess) g->device_pending = true;
    if (e != hipSuccess) cpc::set_error("%s: %s", who, hipGetErrorString(e));
    return e != hipSuccess ? (int)CPC_ERR_HIP : rc;
}

extern "C" int cpc_mt_draw_device_async(cpc_mt19937 *g, uint32_t *raw_host, uint32_t *raw_dev, size_t n, int device,
                                        cpc_stream_t caller_stream)
{
    if (g == nullptr || raw_host == nullptr || raw_dev == nullptr) { cpc::set_error("cpc_mt_draw_device_async: null argument"); return CPC_ERR_INVALID; }
    cpc_negidx_wait(g);
    submit(g, [=] { return device_job(g, raw_host, raw_dev, nullptr, n, device, 0, 0, 0, 0, caller_stream, "cpc_mt_draw_device_async"); });
    return CPC_OK;
}

// The same, and the worker also EXPANDS the words into extIdx on its stream (cpc_negidx_expand): step i + 1's index tensor is
// complete on the device before step i has ended, and nothing of the sampler is left on the training stream.
extern "C" int cpc_mt_draw_expand_device_async(cpc_mt19937 *g, uint32_t *raw_host, uint32_t *raw_dev, int32_t *ext_dev, int device,
                                               int batch, int seq_len, int window, int n_neg, cpc_stream_t caller_stream)
{
    if (g == nullptr || raw_host == nullptr || raw_dev == nullptr || ext_dev == nullptr || batch < 1 || seq_len < 2 || window < 1 || n_neg < 1) {
        cpc::set_error("cpc_mt_draw_expand_device_async: bad argument");
        return CPC_ERR_INVALID;
    }
    if (cpc_negidx_expand == nullptr) { cpc::set_error("cpc_mt_draw_expand_device_async: built without the device kernels"); return CPC_ERR_HIP; }
    const size_t n = 2 * (size_t)batch * n_neg * window;
    cpc_negidx_wait(g);
    submit(g, [=] { return device_job(g, raw_host, raw_dev, ext_dev, n, device, batch, seq_len, window, n_neg, caller_stream,
                                      "cpc_mt_draw_expand_device_async"); });
    return CPC_OK;
}

// The same draw ahead, preceded ON THE WORKER by a repositioning of the generator: state (mt624, left, next) is restored and
// `skip_words` outputs are generated and dropped.  For a draw ahead whose words were used only in part (the call that followed was
// smaller: its 2 n words are a PREFIX of the 2 n' drawn, the stream being one sequence whatever it is cut into): the generator
// has to stand behind the words that WERE consumed before the next draw, and the caller does not wait for that.
extern "C" int cpc_mt_redraw_expand_device_async(cpc_mt19937 *g, const uint32_t *restore_mt624, int restore_left, int restore_next,
                                                 size_t skip_words, uint32_t *raw_host, uint32_t *raw_dev, int32_t *ext_dev, int device,
                                                 int batch, int seq_len, int window, int n_neg, cpc_stream_t caller_stream)
{
    if (g == nullptr || restore_mt624 == nullptr || raw_host == nullptr || raw_dev == nullptr || ext_dev == nullptr || batch < 1 || seq_len < 2 ||
        window < 1 || n_neg < 1 || restore_left < 1 || restore_left > N || restore_next < 0 || restore_next > N ||
        (restore_left > 1 && restore_next + restore_left - 1 != N)) {
        cpc::set_error("cpc_mt_redraw_expand_device_async: bad argument");
        return CPC_ERR_INVALID;
    }
    if (cpc_negidx_expand == nullptr) { cpc::set_error("cpc_mt_redraw_expand_device_async: built without the device kernels"); return CPC_ERR_HIP; }
    const size_t n = 2 * (size_t)batch * n_neg * window;
    std::vector<uint32_t> st(restore_mt624, restore_mt624 + N);
    cpc_negidx_wait(g);
    submit(g, [=] {
        std::memcpy(g->mt, st.data(), sizeof(g->mt));
        g->left = restore_left;
        g->next = restore_next;
        g->tmp.resize(4096);
        for (size_t done = 0; done < skip_words;) {
            const size_t take = std::min<size_t>(4096, skip_words - done);
            draw(g, g->tmp.data(), take);
            done += take;
        }
        return device_job(g, raw_host, raw_dev, ext_dev, n, device, batch, seq_len, window, n_neg, caller_stream, "cpc_mt_redraw_expand_device_async");
    });
    return CPC_OK;
}

// Same as cpc_negidx_sample_host but on the worker thread: returns at once, ext_idx_host is valid after
// cpc_negidx_wait(g).  Lets the host draw step i+1's indices while the GPU is busy with step i.
extern "C" int cpc_negidx_sample_host_async(cpc_mt19937 *g, int batch, int seq_len, int window, int n_neg, int time_major,
                                            int32_t *ext_idx_host)
{
    const int st = check_args(g, batch, seq_len, window, n_neg, ext_idx_host);
    if (st != CPC_OK) return st;
    cpc_negidx_wait(g);
    submit(g, [=] { return sample_impl(g, batch, seq_len, window, n_neg, time_major, ext_idx_host, nullptr, nullptr); });
    return CPC_OK;
}
