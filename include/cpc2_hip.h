/*
 * cpc2_hip.h -- C ABI of the MI355X-native CPC training hot path (libcpc2_hip.so).
 *
 * The reference (MarvinLvn/CPC2) has no FFI/operator registry on this path: its boundary is
 * the Python nn.Module API of cpc/model.py and cpc/criterion/criterion.py, under which it
 * dispatches PyTorch ATen ops.  This library is what sits UNDER that boundary here: each entry
 * point replaces the ATen call sequence of one reference function (cited per function as
 * file:line relative to /root/reference).  The Python modules under cpc2_amd/ keep the
 * reference's class names,
 * constructor/forward signatures and state-dict keys and call these through ctypes.
 *
 * Conventions
 *   - plain C, no torch types; all pointers are DEVICE pointers unless named *_host.
 *   - the caller allocates every buffer (inputs, outputs, saved-for-backward workspace,
 *     scratch); sizes come from the *_bytes() queries; the library never frees or retains.
 *   - all work is enqueued on `stream` (a hipStream_t); no host synchronisation.  One exception to
 *     "on `stream`": cpc_infonce_backward* build their reference lists on a stream owned by the
 *     library (one per device), forked from and joined to `stream` with events inside the call.
 *   - every function returns 0 on success, a negative cpc_status otherwise;
 *     cpc_last_error() gives a thread-local message.
 *   - fp32 everywhere (the reference's arithmetic); activations are CHANNEL-LAST
 *     ([windows][frames][channels]) inside the library.
 *   - hidden sizes supported by the fused row kernels: 32, 64, 128, 256, 512.
 */
#ifndef CPC2_HIP_H
#define CPC2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *cpc_stream_t; /* hipStream_t */

enum cpc_status {
    CPC_OK = 0,
    CPC_ERR_INVALID = -1, /* bad argument / unsupported shape */
    CPC_ERR_HIP = -2,     /* a HIP runtime call or kernel launch failed */
    CPC_ERR_WORKSPACE = -3 /* workspace / scratch too small */
};

int cpc_version(void);          /* 100 x major + minor; 105 = the entry points of round 5 (cpc_encoder_forward2 / backward2, cpc_coop_set_policy,
                                  * cpc_recurrent_backward_calls, cpc_side_tail_wait) */
const char *cpc_last_error(void);

/* In-situ kernel timing for bench.py: when enabled, the launchers bracket each launch of the named
 * kernel class with hipEvents on the launch stream.  cpc_prof_read sums and releases the finished
 * records of one class: "gemm_planes_nt", "gemm_planes_tn" (the plane-fed products of the encoder), "gemm_nt", "gemm_tn"
 * (the split-in-kernel products), "infonce_fwd", "infonce_bwd", "gru_fwd", "gru_bwd", "conv0_fwd", "conv0_bwd" ("gru_*"
 * times whichever recurrent kernel runs: GRU, LSTM or RNN).  on = 1: every class; on = 2: "gemm_planes_nt" only, on = 3:
 * "gemm_nt" only (the events cost about 0.1 ms per step when every class is timed); 0 (default): off. */
/* Asynchronous errors of the cooperative recurrent kernels (GRU / LSTM at hidden 256 / 512: workgroups that exchange
 * the hidden state through L2 and need to be resident all at once).  Every wait inside them is bounded; a wave whose wait
 * runs out poisons its outputs with NaN AND records a code in a host-visible word.  cpc_gru_* / cpc_lstm_* / cpc_rnn_*
 * return CPC_ERR_HIP (message in cpc_last_error) at their NEXT call when the word is set; cpc_async_error_check
 * synchronises `stream`, then reports and clears it.  Reporting a time-out also switches the process to the streaming
 * recurrent kernels (cpc_coop_set_policy(1)): the step that timed out is lost (NaN loss; cpc_adam_step skips non-finite gradient
 * elements, so the parameters are intact), the following ones do not depend on co-residency.  CPC_COOP_FAULT=1 in the environment
 * (tests) makes one member withhold one publish so that its group times out. */
int cpc_async_error_check(cpc_stream_t stream);

/* Streams that run BESIDE a given stream.  A HIP stream is served by one of a few hardware queues (4 by default per priority) that
 * the runtime assigns by use count at creation, and two streams on one queue run one after the other.  cpc_stream_create_apart
 * returns a non-blocking stream of default priority whose kernels were OBSERVED to run beside those of every avoid[i] (the
 * caller's stream(s); NULL = the null stream is a valid entry): a 200 us spin kernel on the one, a one-wave kernel on the
 * candidate, who finishes first.  (With four queues at most three streams can be apart from a given one AND from each other.)  Blocks the host until avoid[i] has
 * drained (once).  The library's own side stream and the sampler's worker stream are made this way; cpc2_amd.train's
 * data-parallel helper stream too.  cpc_streams_overlap(a, b): the same test on two given streams, 1 = beside, 0 = behind, < 0
 * error.  cpc_stream_apart_failures: streams handed out although every candidate failed the test (a saturated device). */
int cpc_stream_create_apart(const cpc_stream_t *avoid, int n_avoid, cpc_stream_t *out);
int cpc_streams_overlap(cpc_stream_t a, cpc_stream_t b);
long cpc_stream_apart_failures(void);
int cpc_stream_spin(cpc_stream_t stream, long ticks);      /* one wave of `stream` busy for `ticks` of the 100 MHz clock (diagnostics) */
/* the library's side stream on the current device (created on first use, apart from `caller`), for diagnostics */
int cpc_side_stream(cpc_stream_t caller, cpc_stream_t *out);
/* Cooperative recurrent launches (the GRU / LSTM kernels at hidden 256 / 512 that need every workgroup resident at once)
 * issued by this process so far.  The data-parallel glue (train.py:523-527's role) checks with it that a gradient all-reduce
 * is never issued between a step's forward and backward recurrent launches. */
long cpc_coop_launches(void);
/* Granule buffers the cooperative kernels' launches hold right now: one per (device, stream) that launched one lately, at most
 * eight -- the one unused for longest is freed when a ninth stream comes (diagnostics / tests). */
int cpc_coop_comm_buffers(void);
/* Calls of the recurrent backward entry points (cpc_gru_backward*, cpc_lstm_backward*, cpc_rnn_backward; cooperative or streaming
 * kernels alike) by this process so far: what "the recurrent backward of this step has been issued" is decided by. */
long cpc_recurrent_backward_calls(void);
/* Process-wide policy for those kernels: 0 (default) = cooperative wherever they fit, 1 = the streaming (non-cooperative) kernels
 * only; returns the previous policy, policy < 0 only queries.  The cooperative kernels assume that nothing else holds CUs while
 * they run.  One rank per GPU with the library's own gradient exchange keeps that by stream order (cpc2_amd/train.py,
 * DataParallelContext); the reference's arrangement -- DistributedDataParallel around model and criterion on an RCCL process
 * group, cpc/train.py:523-527 -- does not: its criterion bucket is all-reduced (an RCCL kernel on the same CUs) while the
 * recurrent backward runs.  cpcStep selects policy 1 when it is handed such a wrapper (sticky for the process). */
int cpc_coop_set_policy(int policy);
int cpc_prof_enable(int on);
int cpc_prof_read(const char *name, double *total_ms, long *count);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 GEMMs.  Replace the ATen addmm/linear calls behind nn.Linear / nn.GRU input
 * projections (criterion.py:144-146,163; model.py:196) and, through the encoder entry points,
 * the cuDNN convolutions of model.py:81-107.
 *   nt:  C[M,N] = A[M,K] . B[N,K]^T (+ bias[N])          (lda/ldb/ldc in elements)
 *   tn:  C[M,N] = sum_r A[r,M]^T . B[r,N], r < R   (split over R; scratch from the query)
 * f32 in, f32 out, f32 accumulation.  Mode 0 (default) multiplies on the bf16 matrix pipe after an
 * exact three-term bf16 split of every operand (six partial products, error <= that of the f32
 * MFMA path, tests/test_gpu_parity.py::test_gemm_split_accuracy); mode 1 uses the f32 MFMA
 * (v_mfma_f32_32x32x2_f32); mode 2 (opt-in, never the default) makes the encoder's plane-fed convolution products at
 * hidden 256 / 512 multiply only a0 b0 + a0 b1 + a1 b0 of the split (16 bits of product mantissa, f32 accumulation: half the
 * matrix work; TF32 -- what cuDNN gives the reference's convolutions on its own GPUs by default -- keeps 10 bits), everything
 * else as mode 0.  cpc_gemm_set_mode returns the previous mode; other values only query.  The mode is PROCESS-WIDE (atomic, default
 * 0) and the only setting the library keeps between calls: it selects the arithmetic of a whole run (the tests' yardstick, the
 * benchmark's labelled entry) and has to reach the backward pass, which autograd runs on a thread of its own -- a per-thread mode
 * does not (tried in round 4).  Select it before the first call of a run, not concurrently with one.
 * cpc_gemm_nt splits K over workgroups when the output has few tiles and then adds the partial sums with
 * fp32 atomics (C is zeroed first); the module entry points below lend scratch for an ordered reduction instead.
 * ------------------------------------------------------------------------------------------ */
int cpc_gemm_set_mode(int mode);
int cpc_gemm_nt(const float *A, long lda, const float *B, long ldb, float *C, long ldc,
                const float *bias, int M, int N, int K, cpc_stream_t stream);
size_t cpc_gemm_tn_scratch_bytes(int M, int N, long R);
int cpc_gemm_tn(const float *A, long lda, const float *B, long ldb, float *C, long ldc,
                int M, int N, long R, void *scratch, size_t scratch_bytes, cpc_stream_t stream);

/* The same products with operands stored as their three bf16 terms ("planes": x = p0 + p1 + p2, p0 = bf16(x),
 * p1 = bf16(x - p0), p2 = bf16(x - p0 - p1); bf16 stored as 16-bit words, planes `plane_stride` elements apart).
 * The encoder's kernels write activations and weights in this form so that the GEMM tiles go global -> LDS by
 * LDS-DMA with no arithmetic besides the MFMAs; these two entries expose the format for tests and probes.
 * Layout of a plane: 16-element chunks; columns 16c .. 16c+15 of row R at chunk (c * s + R % s) * rows_per_phase + R / s,
 * s = 1 << stride_log2 (the stride of the Conv1d that will read the rows; 0 for a plain matrix).
 *   cpc_split_planes:   x[rows][ld] (f32, `cols` columns, cols % 16 == 0) -> planes[3][plane_stride]
 *   cpc_gemm_nt_planes: C[M,N] = A . B^T (+ bias).  K step ks (16 elements) of A is chunk c = ks >> a_taps_log2 of tap
 *                       j = (jj >> 1) + (jj & 1) * s, jj = ks & (k - 1) -- a Conv1d with k = 2 s taps over a C-channel signal:
 *                       K = k * C in the order (chunk, tap 0, s, 1, s + 1, ...); a plain matrix: a_taps_log2 = 0 --;
 *                       GEMM row m starts at signal row s * ((m / a_seg_rows) * a_seg_q + m % a_seg_rows)
 *                       (a_seg_rows <= 0: one segment).
 *                       B: planes of a plain [N][K] matrix in the same K order (stride_log2 0, rows_per_phase N).
 *                       N % 256 == 0, K % 32 == 0, K >= 64. */
int cpc_split_planes(const float *x, long ld, long rows, int cols, void *planes, long plane_stride, int stride_log2,
                     long rows_per_phase, cpc_stream_t stream);
int cpc_gemm_nt_planes(const void *a_planes, long a_plane_stride, int a_taps_log2, int a_stride_log2,
                       long a_rows_per_phase, int a_seg_rows, long a_seg_q, const void *b_planes, long b_plane_stride,
                       float *C, long ldc, const float *bias, long M, int N, int K, cpc_stream_t stream);
/*   cpc_gemm_tn_planes: C[M,N] = sum_{r<R} X(r, .)^T Y(r, .)  (weight gradients).  Column x of an operand is channel
 *                       x % channels of signal row r * s + tap + x / channels (s = 1 << stride_log2) of its planes.
 *                       M % 256 == 0, N % 256 == 0, R >= 64; rows R .. round_up(R, 32) - 1 of A must be zero, of B finite.
 *                       The partial sums of the row slabs are added in a fixed order (bitwise reproducible). */
size_t cpc_gemm_tn_planes_scratch_bytes(int M, int N, long R);
int cpc_gemm_tn_planes(const void *a_planes, long a_plane_stride, int a_stride_log2, long a_rows_per_phase, int a_tap,
                       int a_channels, const void *b_planes, long b_plane_stride, int b_stride_log2,
                       long b_rows_per_phase, int b_tap, int b_channels, float *C, long ldc, int M, int N, long R,
                       void *scratch, size_t scratch_bytes, cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ChannelNorm on a channel-FIRST tensor x[N,C,L] (standalone module API, model.py:27-60):
 * per (n,l) statistics over C, UNBIASED variance, y = (x-mean)*rsqrt(var+eps)*w[c]+b[c].
 * w/b may be NULL (affine=False).  rstd_save[N*L] is written by forward, read by backward.
 * ------------------------------------------------------------------------------------------ */
int cpc_channelnorm_forward(const float *x, const float *w, const float *b, float *y,
                            float *rstd_save, int N, int C, int L, float eps, cpc_stream_t stream);
int cpc_channelnorm_backward(const float *x, const float *w, const float *dy, const float *rstd_save,
                             float *dx, float *dw, float *db, int N, int C, int L, float eps,
                             cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CPCEncoder (model.py:63-108): five strided Conv1d (k,s,p) = (10,5,3),(8,4,2),(4,2,1)x3, each
 * followed by ChannelNorm (model.py:52-60) and ReLU.
 *   x        [n_windows, 1, length]                     raw waveform
 *   params   20 pointers in state-dict order: for i in 0..4:
 *              conv{i}.weight [H, Cin, k], conv{i}.bias [H],
 *              batchNorm{i}.weight [1,H,1], batchNorm{i}.bias [1,H,1]
 *   z        [n_windows, frames, H] channel-last output (== reference output.permute(0,2,1),
 *            i.e. exactly what CPCModel.forward hands on, model.py:382)
 *   saved    activations kept for backward (cpc_encoder_saved_bytes)
 *   scratch  temporaries (cpc_encoder_scratch_bytes; same query serves forward and backward)
 *   grads    20 pointers, same order/shapes as params; overwritten (not accumulated)
 * ------------------------------------------------------------------------------------------ */
int cpc_encoder_frames(int length);
size_t cpc_encoder_saved_bytes(int n_windows, int length, int hidden);
size_t cpc_encoder_scratch_bytes(int n_windows, int length, int hidden);
int cpc_encoder_forward(const float *x, const float *const *params, float *z, void *saved,
                        void *scratch, int n_windows, int length, int hidden, float eps,
                        cpc_stream_t stream);
int cpc_encoder_backward(const float *x, const float *const *params, const float *dz, void *saved,
                         void *scratch, float *const *grads, int n_windows, int length, int hidden,
                         float eps, cpc_stream_t stream);
/* Deferred form of the same backward: the small passes of conv1-4 that only finish parameter gradients (the column sums of
 * dgamma / dbeta / dbias, the sum of each weight-gradient product's K-split slabs) run on a stream of the library's; the gradients
 * of conv1-4 and their norms may then not be read (nor x, saved, scratch reused) until cpc_side_tail_join(stream') -- see
 * cpc_gru_backward_deferred.  conv0's gradients are complete on `stream` as before. */
int cpc_encoder_backward_deferred(const float *x, const float *const *params, const float *dz, void *saved,
                         void *scratch, float *const *grads, int n_windows, int length, int hidden,
                         float eps, cpc_stream_t stream);
/* The same encoder over TWO input batches without concatenating them: windows 0 .. n_first - 1 are x_first [n_first, 1, length],
 * windows n_first .. n_windows - 1 are x_rest [n_windows - n_first, 1, length] -- train.py:99's cat([past, future]) as two pointers
 * (only the first layer reads the waveform).  Everything else as cpc_encoder_forward / cpc_encoder_backward; deferred != 0 selects
 * the deferred form of the backward (cpc_encoder_backward_deferred). */
int cpc_encoder_forward2(const float *x_first, const float *x_rest, int n_first, const float *const *params, float *z, void *saved,
                         void *scratch, int n_windows, int length, int hidden, float eps, cpc_stream_t stream);
int cpc_encoder_backward2(const float *x_first, const float *x_rest, int n_first, const float *const *params, const float *dz,
                          void *saved, void *scratch, float *const *grads, int n_windows, int length, int hidden, float eps,
                          int deferred, cpc_stream_t stream);
/* Inspection (tests only; the layout of `saved` is otherwise private): what the forward pass keeps of layer 0..4 --
 * the ChannelNorm of model.py:52-60 as (xhat, rstd) (layers 1..4) and, at hidden 256 / 512, the layer's ReLU'd output as the
 * next layer's input planes (layers 0..3).  out[10]:
 *   [0] byte offset of xhat [n_windows * out[2]][hidden] f32 (row n * out[2] + t, t < out[3] valid frames; -1: layer 0),
 *   [1] byte offset of rstd [n_windows * out[2]] f32 (-1: layer 0), [2] rows per window, [3] frames per window,
 *   [4] byte offset of the three bf16 planes of the layer's output (-1: stored as f32, or layer 4), [5] elements per plane,
 *   [6] rows per phase, [7] log2 of the reading stride s; channel ch of frame t of window n is element
 *   ((((ch / 16) << [7]) + (R & (s - 1))) * [6] + (R >> [7])) * 16 + ch % 16 with R = n * [8] + [9] + t. */
int cpc_encoder_saved_layout(int n_windows, int length, int hidden, int layer, long *out);

/* ------------------------------------------------------------------------------------------
 * CPCAR with mode="GRU" (model.py:158-207 -> torch.nn.GRU, batch_first, gate order r,z,n).
 *   x       [n, t, dim_in]
 *   params  4 pointers per layer: weight_ih_l{k} [3H, in], weight_hh_l{k} [3H, H],
 *           bias_ih_l{k} [3H], bias_hh_l{k} [3H]
 *   h0      [layers, n, H] or NULL (zeros)          (model.py:196, keepHidden :197-201)
 *   out     [n, t, H];  h_last [layers, n, H] or NULL
 *   backward: dout [n,t,H] -> dx [n,t,dim_in] (may be NULL), grads (4 per layer, overwritten)
 * ------------------------------------------------------------------------------------------ */
size_t cpc_gru_saved_bytes(int n, int t, int dim_in, int hidden, int layers);
size_t cpc_gru_scratch_bytes(int n, int t, int dim_in, int hidden, int layers);
int cpc_gru_forward(const float *x, const float *const *params, const float *h0, float *out,
                    float *h_last, void *saved, void *scratch, int n, int t, int dim_in, int hidden,
                    int layers, cpc_stream_t stream);
int cpc_gru_backward(const float *x, const float *const *params, const float *dout, void *saved,
                     void *scratch, float *dx, float *const *grads, int n, int t, int dim_in,
                     int hidden, int layers, cpc_stream_t stream);
/* Deferred form of the same backward: on return `dx` is ordered on `stream`; the parameter gradients of every layer (weight_ih,
 * weight_hh and the two biases: two weight-gradient products and two column sums per layer that nothing in a backward pass needs
 * before the optimiser) are produced on a stream of the library's, beside the next layer's recurrent kernel and what the caller
 * enqueues on `stream` next, and NOTHING may read them (nor reuse x, saved, scratch, dout) until cpc_side_tail_join(stream') has been called for the
 * stream' that will -- it makes stream' wait for them (a no-op when nothing is pending; one such tail per device).  For callers
 * that write gradients in place and read them only at the end of the backward pass (cpc2_amd: FlatAdam's flat buffer). */
int cpc_gru_backward_deferred(const float *x, const float *const *params, const float *dout, void *saved,
                              void *scratch, float *dx, float *const *grads, int n, int t, int dim_in,
                              int hidden, int layers, cpc_stream_t stream);
int cpc_side_tail_join(cpc_stream_t stream);
/* `stream` waits for the same work, which STAYS pending: for a stream that only consumes the finished gradients (the data-parallel
 * exchange's helper stream) while the caller's stream goes on; the buffers of the deferred calls are released by cpc_side_tail_join. */
int cpc_side_tail_wait(cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CPCAR with mode="LSTM" (model.py:171-173 -> torch.nn.LSTM, batch_first, gate order i,f,g,o;
 * the default arMode of this fork, cpc/cpc_default_config.py).  Same conventions as the GRU:
 *   params  4 pointers per layer: weight_ih_l{k} [4H, in], weight_hh_l{k} [4H, H],
 *           bias_ih_l{k} [4H], bias_hh_l{k} [4H]
 *   h0, c0  [layers, n, H] or NULL (zeros);  h_last, c_last [layers, n, H] or NULL
 *           (keepHidden keeps the (h, c) tuple, model.py:197-199)
 * ------------------------------------------------------------------------------------------ */
size_t cpc_lstm_saved_bytes(int n, int t, int dim_in, int hidden, int layers);
size_t cpc_lstm_scratch_bytes(int n, int t, int dim_in, int hidden, int layers);
int cpc_lstm_forward(const float *x, const float *const *params, const float *h0, const float *c0,
                     float *out, float *h_last, float *c_last, void *saved, void *scratch, int n,
                     int t, int dim_in, int hidden, int layers, cpc_stream_t stream);
int cpc_lstm_backward(const float *x, const float *const *params, const float *dout, void *saved,
                      void *scratch, float *dx, float *const *grads, int n, int t, int dim_in,
                      int hidden, int layers, cpc_stream_t stream);
/* Deferred form: as cpc_gru_backward_deferred (layer 0's weight gradients on the library's stream until cpc_side_tail_join). */
int cpc_lstm_backward_deferred(const float *x, const float *const *params, const float *dout, void *saved,
                      void *scratch, float *dx, float *const *grads, int n, int t, int dim_in,
                      int hidden, int layers, cpc_stream_t stream);

/* CPCAR with mode="RNN" (model.py:174-176 -> torch.nn.RNN, tanh): weight_ih [H, in], weight_hh [H, H],
 * bias_ih [H], bias_hh [H]; arguments as for the GRU. */
size_t cpc_rnn_saved_bytes(int n, int t, int dim_in, int hidden, int layers);
size_t cpc_rnn_scratch_bytes(int n, int t, int dim_in, int hidden, int layers);
int cpc_rnn_forward(const float *x, const float *const *params, const float *h0, float *out,
                    float *h_last, void *saved, void *scratch, int n, int t, int dim_in, int hidden,
                    int layers, cpc_stream_t stream);
int cpc_rnn_backward(const float *x, const float *const *params, const float *dout, void *saved,
                     void *scratch, float *dx, float *const *grads, int n, int t, int dim_in,
                     int hidden, int layers, cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Transformer autoregressive network (arMode="transformer", cpc/transformers.py:10-134,176-187):
 * `layers` TransformerLayers, each   y = LN1(x + Wo.MHA(x)),  out = LN2(Wl (y + FFN(y)) + bl),
 * 8 heads, causal mask, relative-position bias q_i.Krelpos[:, S-1-(i-j)], dff = 2048, ReLU.
 *   x       [n, s, d_model];  out [n, s, d_out];  s must be a multiple of size_seq (<= 128): longer
 *           inputs are attended in independent blocks of size_seq (transformers.py:38-50)
 *   params  cpc_transformer_param_count() = 15 pointers per layer, in THIS order:
 *           Wq, Wk, Wv, Wo [d,d]; Krelpos [d/8, size_seq] (NULL: no relative positions);
 *           ln_multihead.weight, .bias [d]; lin1.weight [2048,d], lin1.bias [2048];
 *           lin2.weight [d,2048], lin2.bias [d]; last_linear.weight [d_out,d], .bias [d_out];
 *           ln_ffnetwork.weight, .bias [d_out]
 *   dropout_p  0 in eval mode; in training the masks come from a counter-based hash of (seed, index):
 *           pass the same (dropout_p, seed) to backward.
 *   n_classifiers  1: plain TransformerLayers.  k > 1: the LAST layer is the multi-classifier head of
 *           the multi-head predictor (MultiClassifierTransformerHead, transformers.py:137-158):
 *           lin2.weight is [k*d, 2048], lin2.bias [k*d], and
 *           out[n, s, c, :] = LN2(Wl (y + FFN(y)[c*d : (c+1)*d]) + bl), out is [n, s, k, d_out].
 *   backward: dout -> dx (may be NULL), grads (same order/shapes as params, overwritten).
 * ------------------------------------------------------------------------------------------ */
int cpc_transformer_param_count(void);
size_t cpc_transformer_saved_bytes(int n, int s, int d_model, int d_out, int size_seq, int layers, int n_classifiers);
size_t cpc_transformer_scratch_bytes(int n, int s, int d_model, int d_out, int size_seq, int layers, int n_classifiers);
int cpc_transformer_forward(const float *x, const float *const *params, float *out, void *saved,
                            void *scratch, int n, int s, int d_model, int d_out, int size_seq, int layers,
                            int n_classifiers, float dropout_p, unsigned long long seed, cpc_stream_t stream);
int cpc_transformer_backward(const float *x, const float *const *params, const float *dout, void *saved,
                             void *scratch, float *dx, float *const *grads, int n, int s, int d_model,
                             int d_out, int size_seq, int layers, int n_classifiers, float dropout_p,
                             unsigned long long seed, cpc_stream_t stream);
/* Deferred form (see cpc_gru_backward_deferred): with one classifier, every parameter gradient of layer 0 -- seven weight-gradient
 * products with their bias sums, the LayerNorms' and Krelpos' column sums -- is produced on the library's stream after this call
 * has returned; `dx` and the gradients of layers 1.. are ordered on `stream`.  cpc_side_tail_join before anything reads them. */
int cpc_transformer_backward_deferred(const float *x, const float *const *params, const float *dout, void *saved,
                             void *scratch, float *dx, float *const *grads, int n, int s, int d_model,
                             int d_out, int size_seq, int layers, int n_classifiers, float dropout_p,
                             unsigned long long seed, cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Negative-index sampler of CPCUnsupersivedCriterion.sampleClean (criterion.py:247-266), HOST
 * side, bit-exact with torch's CPU generator: 32-bit MT19937, one draw per element,
 * batchIdx = draw % batch (all n first), then seqIdx = draw % (T-1) + 1;
 * extIdx[(bb*n_neg + nn)*W + t] = (seqIdx + t) % T + batchIdx*T   (draw / reference order).
 * time_major != 0 stores the SAME values as [b][W][n_neg] (the negatives of one (b,t) contiguous):
 * the layout cpc_infonce_* consume.  batch_idx/seq_idx outputs (optional) stay in draw order.
 * State interop: mt[624], left, next as in torch's CPUGeneratorImpl legacy state.
 * ------------------------------------------------------------------------------------------ */
typedef struct cpc_mt19937 cpc_mt19937;
cpc_mt19937 *cpc_mt_create(uint32_t seed);
void cpc_mt_destroy(cpc_mt19937 *g);
int cpc_mt_seed(cpc_mt19937 *g, uint32_t seed);
int cpc_mt_get_state(const cpc_mt19937 *g, uint32_t *mt624, int *left, int *next);
int cpc_mt_set_state(cpc_mt19937 *g, const uint32_t *mt624, int left, int next);
int cpc_negidx_sample_host(cpc_mt19937 *g, int batch, int seq_len, int window, int n_neg,
                           int time_major, int32_t *ext_idx_host, int64_t *batch_idx_host_opt,
                           int64_t *seq_idx_host_opt);
/* The same draw on a worker thread (at most one in flight per generator): returns at once;
 * ext_idx_host is valid after cpc_negidx_wait(g).  Every other cpc_mt_* / cpc_negidx_* call on g
 * waits for it first, so the stream stays sequential. */
int cpc_negidx_sample_host_async(cpc_mt19937 *g, int batch, int seq_len, int window, int n_neg,
                                 int time_major, int32_t *ext_idx_host);
int cpc_negidx_wait(cpc_mt19937 *g);
/* Split form used by the training loop: the host only produces the raw generator words (the sequential,
 * torch-bit-exact part; 2*n of them, n = batch*n_neg*window: batchIdx stream then seqIdx stream), the
 * device reduces them (% batch, % (T-1) + 1), applies the time offset and writes the time-major extIdx.
 * Integer-exact: cpc_negidx_expand(raw) == cpc_negidx_sample_host(time_major = 1). */
int cpc_mt_draw_host(cpc_mt19937 *g, uint32_t *raw_host, size_t n);
int cpc_mt_draw_host_async(cpc_mt19937 *g, uint32_t *raw_host, size_t n);
/* ... and uploaded by the worker on its own stream (raw_host pinned): raw_dev holds the words once
 * cpc_negidx_wait returns, so no copy sits on the training stream. */
int cpc_mt_draw_device_async(cpc_mt19937 *g, uint32_t *raw_host, uint32_t *raw_dev, size_t n, int device,
                             cpc_stream_t caller_stream);
/* ... and expanded there too (cpc_negidx_expand on the worker's stream): the two torch.randint calls AND the index arithmetic of
 * criterion.py:247-266 for step i + 1, done while step i runs.  The worker is ONE thread per generator with ONE stream, made on its
 * first device job apart from `caller_stream`'s hardware queue (cpc_stream_create_apart); an event is recorded behind the job.
 * ext_dev [batch * window * n_neg] may be read by the host's device work after cpc_negidx_wait (blocks the host until the device
 * has it) or, without blocking the host on the device, by everything enqueued on `stream` after cpc_negidx_wait_on(g, stream). */
int cpc_mt_draw_expand_device_async(cpc_mt19937 *g, uint32_t *raw_host, uint32_t *raw_dev, int32_t *ext_dev, int device,
                                    int batch, int seq_len, int window, int n_neg, cpc_stream_t caller_stream);
/* ... preceded on the worker by a repositioning of the generator: the state (mt624, left, next) is restored and skip_words outputs
 * are generated and dropped.  For a draw ahead of which a SMALLER call used only a prefix (its 2 n words are the first 2 n of the
 * stream whatever it is cut into): the generator has to stand behind the consumed words before the next draw, and the caller does
 * not wait for that. */
int cpc_mt_redraw_expand_device_async(cpc_mt19937 *g, const uint32_t *restore_mt624, int restore_left, int restore_next,
                                      size_t skip_words, uint32_t *raw_host, uint32_t *raw_dev, int32_t *ext_dev, int device,
                                      int batch, int seq_len, int window, int n_neg, cpc_stream_t caller_stream);
int cpc_negidx_wait_on(cpc_mt19937 *g, cpc_stream_t stream);
int cpc_negidx_stream(cpc_mt19937 *g, cpc_stream_t *out);      /* the worker's stream (NULL before its first device job) */
int cpc_negidx_expand(const uint32_t *raw, int32_t *ext_idx, int batch, int seq_len, int window, int n_neg,
                      cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CPCUnsupersivedCriterion.forward with linear predictors (criterion.py:329-363, 291-302,
 * 237-286, PredictionNetwork.forward :152-173): K predictions W_k c_t, scored by dot product / H
 * against 1 positive z[b, t+k] + n_neg gathered negatives, cross-entropy vs class 0.
 * The gathered candidate tensors of the reference are never materialised.
 *   c        [b, T, dim_ar]   context (only t < W = T-K is used)
 *   z        [b, T, dim_enc]  encoder targets
 *   wpred    [K, dim_enc, dim_ar] packed nn.Linear weights (predictors.k.weight)
 *   ext_idx  [b, W, n_neg] int32 rows of z.view(b*T, dim_enc): cpc_negidx_sample_host with
 *            time_major = 1 (same values as the reference's [b, n_neg, W] order, transposed)
 *   weights  [b*W] per-sample loss weights or NULL (ones)        (criterion.py:334-340)
 *   losses   [K]  mean_i(w_i * CE_i);  acc [K] = #(argmax == 0) / (b*W)
 * backward: dlosses [K] upstream gradient -> dc [b,T,dim_ar], dz [b,T,dim_enc],
 *           dwpred [K,dim_enc,dim_ar]  (all overwritten; dz is summed per row in a fixed order, so it
 *           is identical from run to run; CPC_NCE_ATOMIC=1 in the environment selects fp32 atomics)
 * ------------------------------------------------------------------------------------------ */
size_t cpc_infonce_saved_bytes(int b, int t, int k, int dim_ar, int dim_enc, int n_neg);
size_t cpc_infonce_scratch_bytes(int b, int t, int k, int dim_ar, int dim_enc, int n_neg);
/* byte offset, inside `saved`, of the logits the forward pass leaves there: float [b, W, K, 1 + n_neg], candidate 0 = the
 * positive, the negatives in slot order (cpc_infonce_perm_offset) -- what getPrediction (criterion.py:291-302) returns once
 * brought back to the caller's order and permuted to K tensors [b, 1 + n_neg, W] */
size_t cpc_infonce_logits_offset(int b, int t, int k, int dim_ar, int dim_enc, int n_neg);
/* The forward pass visits a (b, t)'s negatives sorted by z-row block and leaves the logits in THAT order (element 1 + g of a
 * row = the negative in slot g: a tile's columns are then consecutive floats).  byte offset, inside `saved`, of the
 * permutation: uint16 [b, W, n_neg], perm[g] = the caller's number (its position in ext_idx) of the negative in slot g */
size_t cpc_infonce_perm_offset(int b, int t, int k, int dim_ar, int dim_enc, int n_neg);
/* (ext_idx is range-checked on the device before anything gathers with it: an index outside [0, b * t) is replaced by 0 and
 *  reported by the next cpc_async_error_check(stream) -- criterion.py:264-268's gather has no such check.) */
int cpc_infonce_forward(const float *c, const float *z, const float *wpred, const int32_t *ext_idx,
                        const float *weights, float *losses, float *acc, void *saved, void *scratch,
                        int b, int t, int k, int dim_ar, int dim_enc, int n_neg, cpc_stream_t stream);
int cpc_infonce_backward(const float *c, const float *z, const float *wpred, const int32_t *ext_idx,
                         const float *weights, const float *dlosses, void *saved, void *scratch,
                         float *dc, float *dz, float *dwpred, int b, int t, int k, int dim_ar,
                         int dim_enc, int n_neg, cpc_stream_t stream);
/* Deferred form of the same backward.  On return only `dc` -- what the context network's backward (cpc/model.py:158-207 under
 * autograd) needs next -- is ordered on `stream`; `dz` and `dwpred` are produced on a stream of the library's, beside whatever
 * the caller enqueues on `stream` afterwards, and NOTHING may read them (nor reuse c, z, ext_idx, saved, scratch) until
 * cpc_infonce_join(stream') has been called for the stream' that will: it makes stream' wait for them (a no-op when nothing is
 * pending; one deferred backward may be pending per device).  The context network's backward is latency-bound and leaves the
 * chip idle; the criterion's dz is a memory-bound sum over ~1 GB -- together they take the time of the longer one. */
int cpc_infonce_backward_deferred(const float *c, const float *z, const float *wpred, const int32_t *ext_idx,
                                  const float *weights, const float *dlosses, void *saved, void *scratch,
                                  float *dc, float *dz, float *dwpred, int b, int t, int k, int dim_ar,
                                  int dim_enc, int n_neg, cpc_stream_t stream);
int cpc_infonce_join(cpc_stream_t stream);
/* The same criterion when the caller hands over ONLY the W = t - k context frames it uses: c and dc are [b, t - k, dim_ar] (z stays
 * [b, t, dim_enc]).  criterion.py:296 slices `cFeature[:, :windowSize]` itself; a causal context network that keeps no state across
 * calls need not compute the k frames behind it (cpc2_amd.train.cpcStep runs it on W steps and calls these).  deferred != 0: the
 * deferred form of the backward (cpc_infonce_backward_deferred). */
int cpc_infonce_forward_cw(const float *c, const float *z, const float *wpred, const int32_t *ext_idx,
                           const float *weights, float *losses, float *acc, void *saved, void *scratch,
                           int b, int t, int k, int dim_ar, int dim_enc, int n_neg, cpc_stream_t stream);
int cpc_infonce_backward_cw(const float *c, const float *z, const float *wpred, const int32_t *ext_idx,
                            const float *weights, const float *dlosses, void *saved, void *scratch,
                            float *dc, float *dz, float *dwpred, int b, int t, int k, int dim_ar,
                            int dim_enc, int n_neg, int deferred, cpc_stream_t stream);

/* The same criterion when the K predictions come from predictor MODULES instead of linear maps
 * (rnnMode="transformer": criterion.py:136-143; the predictions are produced by cpc_transformer_*):
 *   pred   K pointers, each [b, W, dim_enc] (W = t - k): output of predictor k on c[:, :W]
 *   dpred  K pointers, same shapes (overwritten);  saved/scratch sizes: cpc_infonce_*_bytes with dim_ar = dim_enc */
int cpc_infonce_forward_pred(const float *const *pred, const float *z, const int32_t *ext_idx, const float *weights,
                             float *losses, float *acc, void *saved, void *scratch, int b, int t, int k,
                             int dim_enc, int n_neg, cpc_stream_t stream);
int cpc_infonce_backward_pred(const float *const *pred, const float *z, const int32_t *ext_idx,
                              const float *weights, const float *dlosses, void *saved, void *scratch,
                              float *const *dpred, float *dz, int b, int t, int k, int dim_enc, int n_neg,
                              cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Native FLAC reader for the window feeder (replaces torchaudio.load at cpc/dataset.py:411-437 and
 * cpc/feature_loader.py:343; the image ships no audio library).  HOST side.  Output is float32
 * [channels][total_samples] scaled to [-1, 1) like torchaudio.load; every decode is checked against
 * the MD5 of the unencoded audio stored in STREAMINFO (md5_ok, and an error if it differs).
 * ------------------------------------------------------------------------------------------ */
int cpc_flac_info(const char *path, int *sample_rate, int *channels, int *bits_per_sample,
                  long *total_samples);
int cpc_flac_decode_f32(const char *path, float *out_host, long capacity_floats, int *md5_ok);

/* Window feeder (cpc/dataset.py:308-321 __getitem__ slicing): out[i][0..window) =
 * audio[offsets[i] .. offsets[i]+window) from one flat device-resident audio buffer (zeros outside it).
 * audio, offsets (int64) and out are DEVICE pointers. */
int cpc_window_gather(const float *audio, long total_samples, const long *offsets, float *out, int batch,
                      int window, cpc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Adam on one flat fp32 buffer (torch.optim.Adam as built at train.py:477-479: no weight decay,
 * no amsgrad).  g is multiplied by grad_scale first (1/world_size after an all-reduce SUM).
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * An element whose gradient is not finite is left alone (p, m, v unchanged) and the asynchronous error word is set: the
 * next cpc_async_error_check(stream) returns CPC_ERR_HIP.  A cooperative recurrent kernel that timed out therefore cannot
 * poison the weights.  This is a PARTIAL update when only some elements are non-finite (the finite ones are applied, the step
 * count advances) and a deviation from the reference, whose torch.optim.Adam (train.py:477-479) propagates the NaN into the
 * weights; recovery = reload the last checkpoint, or continue knowingly.
 * ------------------------------------------------------------------------------------------ */
int cpc_adam_step(float *p, const float *g, float *m, float *v, long n, int step, float lr,
                  float beta1, float beta2, float eps, float grad_scale, cpc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CPC2_HIP_H */
