// Internal helpers shared by the HIP translation units of libcpc2_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/cpc2_hip.h"

namespace cpc {

void set_error(const char *fmt, ...);

#define CPC_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            cpc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return CPC_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define CPC_CHECK_LAUNCH(name)                                                           \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            cpc::set_error("launch of %s failed: %s", name, hipGetErrorString(_e));      \
            return CPC_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define CPC_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            cpc::set_error(__VA_ARGS__);                                                 \
            return CPC_ERR_INVALID;                                                      \
        }                                                                                \
    } while (0)

#define CPC_TRY(expr)                                                                    \
    do {                                                                                 \
        int _s = (expr);                                                                 \
        if (_s != CPC_OK) return _s;                                                     \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided buffer (256-byte aligned carves).
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T> T *take(size_t count) {
        off = align_up(off, 256);
        T *p = reinterpret_cast<T *>(base ? base + off : nullptr);
        off += count * sizeof(T);
        return p;
    }
    size_t used() const { return align_up(off, 256); }
};

// ---- optional in-situ kernel timing (see cpc_prof_* in cpc2_hip.h) ----
enum ProfSlot { PROF_GEMM_NT = 0, PROF_GEMM_TN, PROF_NCE_FWD, PROF_NCE_BWD, PROF_GRU_FWD, PROF_GRU_BWD, PROF_CONV0_FWD,
                PROF_CONV0_BWD, PROF_PLANES_NT, PROF_PLANES_TN,
                PROF_SIDE_WAIT,      // a stream held at a join with the library's side stream (events around the wait: idle time, not a kernel)
                PROF_SLOTS };
class ProfScope {
public:
    // attached = true: the scope records nothing itself -- the ONE kernel launched inside takes start() / stop() as the events
    // of its own dispatch (hipExtLaunchKernelGGL): no barrier packets on the stream, which cost ~6 us of idle time each
    ProfScope(int slot, hipStream_t st, bool attached = false);
    ~ProfScope();
    hipEvent_t start() const { return active_ ? a_ : nullptr; }
    hipEvent_t stop() const { return active_ ? b_ : nullptr; }
    // an attached scope whose kernel was NOT launched after all (an error path): its events were never recorded -- drop them
    // instead of queueing a record cpc_prof_read would call hipEventElapsedTime on
    void cancel();
private:
    bool attached_;
    int slot_;
    hipStream_t st_;
    hipEvent_t a_, b_;
    bool active_;
};

// ---- encoder geometry (model.py:85-94) ----
struct ConvGeom { int k, s, p; };
static const ConvGeom kConv[5] = {{10, 5, 3}, {8, 4, 2}, {4, 2, 1}, {4, 2, 1}, {4, 2, 1}};

// ---- dropout: counter-based hash keyed by (seed, element index) -- the same mask wherever it is recomputed ----
#ifdef __HIPCC__
// (32-bit arithmetic throughout: the murmur3 finaliser over the index, keyed by two words derived from the seed -- those are wave
//  uniform, i.e. scalar-unit work.  The splitmix64 finaliser used until round 4 cost three 64-bit multiplies per element: a third
//  of the FFN activation's product once the mask moved into its epilogue.)
__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t hash32(uint64_t seed, uint64_t idx)
{
    const uint32_t s1 = fmix32((uint32_t)seed ^ 0x9E3779B9u), s2 = fmix32((uint32_t)(seed >> 32) + 0x7F4A7C15u + s1);
    return fmix32((((uint32_t)idx ^ s1) * 0x9E3779B1u) + ((uint32_t)(idx >> 32) ^ s2) * 0x85EBCA77u + s2);
}
// multiplier applied to a kept/dropped element: 1/(1-p) or 0 (thresh = p * 2^32; thresh == 0 -> always 1)
__device__ __forceinline__ float drop_mul(uint64_t seed, uint64_t idx, uint32_t thresh, float scale)
{
    return (thresh == 0u || hash32(seed, idx) >= thresh) ? scale : 0.f;
}
#endif

// ---- row mapping of a GEMM epilogue: virtual row m -> output row (or skip) ----
struct RowMap {
    int enabled;      // 0: crow = m
    int rv;           // virtual rows per group (sample)
    int out_stride;   // l = t*out_stride + out_off
    int out_off;
    int l_max;        // valid l in [0, l_max)
    int rows_out;     // output rows per group
    int col_rows;     // > 0: output column n belongs to row l + n / col_rows (the s phases of a backward-data
                      // product side by side: N = s * col_rows, ldc = col_rows); multiple of 32.  0: off
    // independent of `enabled`: M is made of segments of seg_rows virtual rows of which only the first seg_valid
    // need computing (a sample's trailing junk rows); the split kernels then tile each segment separately
    int seg_rows, seg_valid;
    // optional: room for the partial products of a K split (few output tiles, long K).  With it the partial sums go to
    // slabs that a second kernel adds in a fixed order; without it (or when too small) they are added to C with fp32
    // atomics, in whatever order the workgroups arrive.  Size: gemm_nt_scratch_bytes(M, N, K)
    void *splitk_scratch;
    size_t splitk_bytes;
    // elementwise work fused into the store of a DENSE output (no row map, ldc == N, no K split; gemm_nt refuses otherwise):
    //   EPI_RELU_DROPOUT: c = v > 0 ? v * drop_mul(seed, m * N + n) : 0      (the FFN's activation, transformers.py:112-116)
    //   EPI_GATE:         c = gate[m * N + n] > 0 ? v * scale : 0            (its adjoint: gate = the activation's output)
    int epi;
    unsigned long long epi_seed;
    unsigned epi_thresh;
    float epi_scale;
    const float *epi_gate;
};
enum { EPI_NONE = 0, EPI_RELU_DROPOUT = 1, EPI_GATE = 2 };

// ---- internal launchers (defined in gemm_f32.hip / rowops.hip) ----
int gemm_nt(const float *A, long lda, const float *B, long ldb, float *C, long ldc, const float *bias,
            long M, int N, int K, const RowMap &map, hipStream_t st);
size_t gemm_nt_scratch_bytes(long M, int N, int K);      // upper bound of what RowMap::splitk_scratch can need
size_t gemm_tn_scratch_bytes(int M, int N, long R);
// conv_cin > 0: C is a Conv1d weight [M][conv_cin][conv_k] and column j*conv_cin+ci goes to [ci][j]
int gemm_tn(const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N, long R,
            void *scratch, size_t scratch_bytes, int conv_cin, int conv_k, hipStream_t st);

// out[c] = sum_r part[r*ld + c]
int colsum(const float *part, long rows, long ld, int width, float *out, hipStream_t st);
// two-stage form for tall partial matrices; column c goes to outs[c / seg][c % seg] (up to three outputs)
constexpr int COLSUM_SPLIT = 32;
size_t colsum_split_scratch_bytes(int width);
int colsum_split(const float *part, long rows, long ld, int width, float *out0, float *out1, float *out2, int seg,
                 void *scratch, hipStream_t st);
// column sums of a [rows][width] matrix with row stride ld (two-stage; scratch >= colsum_scratch_bytes)
size_t colsum_rows_scratch_bytes(int width);
int colsum_rows(const float *a, long ld, long rows, int width, float *out, void *scratch, hipStream_t st);

int transpose2d(const float *a, float *at, int rows, int cols, hipStream_t st);                      // at[c][r]=a[r][c]

// ---- operands stored as their three bf16 terms (gemm_planes.hip) ----
// Chunked planes: the 16 consecutive K elements (channels) 16c .. 16c+15 of signal row R are 32 contiguous bytes at chunk
// index (c * s + R % s) * rts + R / s of every plane (s = 1 << sshift: the stride of the Conv1d that reads the signal; its
// phases are stored apart so that the rows t*s + j of consecutive output frames t are consecutive chunks).
// K step ks of a GEMM over it is chunk c = ks >> kshift of tap j = (jj >> 1) + (jj & 1) * s, jj = ks & (k - 1), k = 1 << kshift
// taps (k = 2 s, or 1): chunk-major, the taps j and j + s -- which read the same signal rows one apart -- next to each other.
// kshift = 0 for an operand whose K is one run of chunks (weights [N][K] stored in that K order, plain matrices).
// GEMM row m starts at signal row s * ((m / segv) * seg_q + m % segv).
struct PlanesOperand {
    const unsigned short *p;
    long plane;          // elements between planes
    int kshift;          // log2(taps)
    int sshift;          // log2(s)
    long rts;            // chunks per (c, phase): rows_total / s
    int segv;            // GEMM rows per segment (sample); <= 0: one segment
    long seg_q;          // row distance between segments, in units of s rows
};
int split_planes(const float *x, long ld, long rows, int cols, unsigned short *planes, long plane, int sshift, long rts,
                 hipStream_t st);
bool gemm_nt_planes_ok(long M, int N, int K);
// ChannelNorm + ReLU + the three-term split fused into the product's epilogue (N == 256: a tile holds whole rows of the output,
// no K split): C receives xhat = (y - mean) * rstd instead of y, `rstd` the per-row statistic, and relu(gamma * xhat + beta)
// leaves as the NEXT layer's input planes (row of (sample n, frame t) = n * rows_next + halo + t; other rows are not written).
struct PlanesNormOut {
    const float *gamma, *beta;
    float eps;
    float *rstd;                 // [rows of C]
    unsigned short *p;           // planes of the next layer's input (layout: PlanesOperand)
    long plane;
    int sshift;
    long rts;
    long rows_next;
    int halo;
};
bool gemm_nt_planes_norm_ok(long M, int N, int K);
// left_slabs (optional): a K split then leaves its partial products in RowMap::splitk_scratch (slab s at s * out_rows * N floats,
// bias in slab 0) and reports their number instead of summing them into C -- for a consumer that sums as it reads; 0: C is written
int gemm_nt_planes(const PlanesOperand &A, const PlanesOperand &B, float *C, long ldc, const float *bias, long M, int N, int K,
                   const RowMap &map, hipStream_t st, const PlanesNormOut *norm = nullptr, int *left_slabs = nullptr);
// weight-gradient form: C[i][j] = sum_{r < R} X(r, i) * Y(r, j); column x of an operand is channel x % C of tap tap0 + x / C,
// i.e. element x % C of signal row r * s + tap.  Rows R .. round_up(R, 32) - 1 of A must be ZERO and those of B finite.
struct PlanesTNOperand {
    const unsigned short *p;
    long plane;
    int sshift;
    long rts;
    int tap0;            // first tap (a row shift for s = 1)
    int C;               // channels per tap (multiple of 32)
};
bool gemm_tn_planes_ok(int M, int N, long R);
size_t gemm_tn_planes_scratch_bytes(int M, int N, long R);
size_t gemm_nt_planes_scratch_bytes(long M, int N, int K, long out_rows);   // what RowMap::splitk_scratch needs (0: no K split)
// left_slabs (optional): the K split's slabs are left in `scratch` and their number reported instead of being summed into C -- the
// caller runs planes_tn_reduce when and where it wants (the deferred encoder backward: on the side stream, at the end)
int gemm_tn_planes(const PlanesTNOperand &A, const PlanesTNOperand &B, float *C, long ldc, int M, int N, long R, void *scratch,
                   size_t scratch_bytes, int conv_cin, int conv_k, hipStream_t st, int *left_slabs = nullptr);
int planes_tn_reduce(const float *slabs, int S, int M, int N, float *C, long ldc, int conv_cin, int conv_k, hipStream_t st);

int gemm_mode();        // gemm_f32.hip: 0 exact bf16 split (six products), 1 f32 MFMA, 2 three products (opt-in, cpc_gemm_set_mode)

// infonce.hip: start what a deferred criterion backward (cpc_infonce_backward_deferred) left to do, on the library's side
// stream, ordered behind what `st` holds now; no-op when nothing is pending.  Called by the context networks' backward entry
// points right behind their first kernel.
int infonce_deferred_mark(hipStream_t st);      // the point of `st` the side stream waits for (first call after the backward wins)
int infonce_deferred_start(hipStream_t st);
// work of a backward entry point that nothing on its stream needs (weight gradients): on the library's side stream, joined later
int side_tail_begin(hipStream_t st, hipStream_t *side_stream);
int side_tail_end();
int side_tail_join(hipStream_t st);
int side_tail_wait(hipStream_t st);       // `st` waits for the tail, which stays pending (a later side_tail_join still joins)
// rowops.hip: a non-blocking stream of default priority that demonstrably runs BESIDE avoid[0 .. n) (not on their hardware queues)
int stream_create_apart(const hipStream_t *avoid, int n_avoid, hipStream_t *out);

}  // namespace cpc
