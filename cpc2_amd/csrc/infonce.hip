// CPCUnsupersivedCriterion with linear predictors on gfx950.
// Reference: /root/reference/cpc/criterion/criterion.py:329-363 (forward), :291-302 (getPrediction),
// :237-286 (sampleClean), :152-173 (PredictionNetwork.forward).
//
// The reference materialises, for each of the K prediction steps, a [b, 1+Nneg, W, H] candidate tensor
// (11.8 GB at b=64) and a same-size product.  Here one WAVE owns one (window b, frame t): it keeps the K
// predictions P_k = W_k c_t in registers, streams the gathered candidate rows of z (8.4 MB, L2/MALL
// resident) straight into the MFMA operand, scores all K x candidates with the f32 MFMA and does the
// cross-entropy in registers; only the logits (for backward), K partial losses and K hit flags leave it.
//
// Candidate list of a workgroup: 16 "positive-tile" rows z[b][t+1 .. t+16] (the positive of step k is
// column k of that tile -- computed by the same MFMA chain as the negatives, so a negative that happens
// to be the positive frame ties EXACTLY, as in the reference) followed by the Nneg gathered negatives.
#include "common.h"
#include <hip/hip_ext.h>
#include "coop.h"
#include "ldsdma.h"
#include "rowcfg.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace cpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NCE_ROWS = 16;   // MFMA M: predictions padded to 16 rows
constexpr int NCE_POS = 16;    // positive-tile columns

struct NceArgs {
    // predictions of step k for (bb, t): Pk[k] + (bb*p_rows + t)*p_stride  (H floats).
    //   linear predictors : Pk[k] = P + k*H, p_stride = K*H, p_rows = T   (one GEMM output [b*T][K*H])
    //   module predictors : Pk[k] = output of predictor k [b][W][H], p_stride = H, p_rows = W
    const float *Pk[16];
    float *dPk[16];        // same addressing, gradients
    long p_stride;
    int p_rows;
    int p_packed;          // the K prediction rows of a (b, t) are K*H consecutive floats (linear predictors: one GEMM output)
    const float *z;        // [b*T][H]
    const int32_t *ext;    // [b][W][Nneg]  TIME-MAJOR index layout (the negatives of one (b,t) are contiguous); inside the
                           // library: sorted by z-row block (nce_block_sort_kernel), slot g holds the caller's negative perm[g]
    const unsigned short *perm;   // [b][W][Nneg]: the caller's negative number of slot g (logits keep the caller's order)
    const float *weights;  // [b*W] or null
    float *logits;         // [b*W][K][Nneg+1]: element 0 the positive, element 1 + g the negative in SLOT g of the block-sorted
                           // list (not the caller's number perm[g]): the columns of a tile are then consecutive floats of a row
    float *lse;            // [b*W][K]
    float *lossp;          // [b*W][K]   w * CE
    float *hit;            // [b*W][K]   1 if argmax == 0
    int b, T, W, K, Nneg;
    int lw;                // LDS dS row length (floats), multiple of 4
    // backward
    const float *dloss;    // [K]
    float *dz;             // [b*T][H]  (atomics; unused when vbuf is set)
    float *vbuf;           // [b*W][K + Nneg][H] or null: every candidate's dz contribution stored once (see below)
    float *ds_buf;         // [b*W][17][lw]: dS and the candidate rows, handed from infonce_bwd_kernel to the dz kernels
    float inv_count;       // 1 / (b*W)
    unsigned long long *stamps;   // diagnostic build (CPC_NCE_STAMP=1): per wave, s_memtime at five points of the forward kernel
};

// The gather table z (b*T rows) is larger than one XCD's L2 (8.4 MB against 4 MB at the benchmark shape), and a (b,t)'s
// negatives are spread uniformly over it: every wave walking ITS negatives in the order they were drawn makes the chip's
// working set the whole table, and about half of the gathered rows miss L2.  Sorted by z-row BLOCK (blocks of ~2 MB), all
// waves -- they start together and do the same amount of work per candidate -- are in the same block at the same time:
// the working set is a block or two.  One wave per (b,t): stable counting sort of its Nneg indices by block, and the
// permutation (the logits keep the caller's candidate order; losses and gradients do not depend on the order).
// Also the ONE place where the caller's indices are looked at before anything gathers with them (criterion.py:264-268's gather;
// SURVEY section 5's range check): every kernel of both passes walks `sorted`, so an index outside [0, nrows) is replaced by 0
// here -- no out-of-bounds LDS-DMA gather can follow -- and reported through the asynchronous error word (cpc_async_error_check).
__global__ __launch_bounds__(64) void nce_block_sort_kernel(const int32_t *ext, int32_t *sorted, unsigned short *perm, int Nneg,
                                                            int rows_per_block, int nblk, int nrows, int *err)
{
    const long base = (long)blockIdx.x * Nneg;
    const int lane = threadIdx.x;
    int pos = 0;
    bool bad = false;
    for (int blk = 0; blk < nblk; ++blk)
        for (int c = 0; c < Nneg; c += 64) {
            const int j = c + lane;
            int row = j < Nneg ? ext[base + j] : -1;
            if (j < Nneg && (unsigned)row >= (unsigned)nrows) { bad = true; row = 0; }
            const bool mine = j < Nneg && min(row / rows_per_block, nblk - 1) == blk;
            const unsigned long long m = __ballot(mine);
            if (mine) {
                const int at = pos + __popcll(m & ((1ull << lane) - 1ull));
                sorted[base + at] = row;
                perm[base + at] = (unsigned short)j;
            }
            pos += __popcll(m);
        }
    if (bad) coop_report(err, COOP_ERR_BAD_INDEX);
}

// z row of candidate g of (bb, t): g < 16 -> positive tile (row t+1+g, only g < K), else negative g-16
__device__ __forceinline__ long nce_row(const NceArgs &a, int bb, int t, int g)
{
#if defined(NCE_DBG) && (NCE_DBG & 1)
    return (g < NCE_POS || g - NCE_POS < a.Nneg) ? (long)(g & 15) : -1;       // probe: every candidate from 16 hot rows
#endif
    if (g < NCE_POS) return (g < a.K && t + 1 + g < a.T) ? (long)bb * a.T + t + 1 + g : -1;
    const int j = g - NCE_POS;
    return j < a.Nneg ? (long)a.ext[((long)bb * a.W + t) * a.Nneg + j] : -1;
}

// Forward: ONE WAVE per (b,t), no LDS, no barriers.  The wave keeps its K predictions (the MFMA A
// operand, P[k][16kk + 4q ..]) in registers and walks the candidate tiles; each lane streams the 16-byte
// pieces of "its" candidate row straight from L2 into the MFMA B operand (lane (r,q): candidate r of the
// tile, k-slice q), so a gathered row is read exactly once and never staged.  Logits of a tile land as
// acc[e] = <P_{4q+e}, cand_r> / H; the cross-entropy is an online (max, sum-exp) over tiles, merged across
// the 16 lanes of a row group at the end.
template <int H> __global__ __launch_bounds__(64) void infonce_fwd_kernel(NceArgs a)
{
    constexpr int KK = H / 16;
    const int bt = blockIdx.x;
    const int bb = bt / a.W, t = bt - bb * a.W;
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const float inv_h = 1.f / H;

    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
    if (a.stamps) st0 = __builtin_amdgcn_s_memtime();
    float4 areg[KK];
    {
        const float *prow = a.Pk[r < a.K ? r : 0] + ((long)bb * a.p_rows + t) * a.p_stride + 4 * q;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            areg[kk] = r < a.K ? *reinterpret_cast<const float4 *>(prow + 16 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ntiles = 1 + (a.Nneg + 15) / 16;
    float pos[4] = {0.f, 0.f, 0.f, 0.f};
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, s[4] = {0.f, 0.f, 0.f, 0.f};

    // the wave's negative rows go to LDS first, in one coalesced round trip: a
    // tile's row loads then issue at once instead of behind an index load of their own (two dependent L2 round trips per
    // tile were the kernel's critical path)
    extern __shared__ __attribute__((aligned(16))) int fwd_lds[];
    int *lrow = fwd_lds;                                                             // [Nneg]
    for (int j = lane; j < a.Nneg; j += 64) lrow[j] = a.ext[(long)bt * a.Nneg + j];
    __syncthreads();                           // one wave
    auto cand_row = [&](int g) -> long {       // z row of candidate g: the positive tile, then the negatives
#if defined(NCE_DBG) && (NCE_DBG & 1)
        return (g < NCE_POS || g - NCE_POS < a.Nneg) ? (long)(g & 15) : -1;
#endif
        if (g < NCE_POS) return (g < a.K && t + 1 + g < a.T) ? (long)bb * a.T + t + 1 + g : -1;
        return g - NCE_POS < a.Nneg ? (long)lrow[g - NCE_POS] : -1;
    };

    // (rows of padding candidates are clamped to row 0, not zeroed: their columns are never used, and an unconditional
    // load keeps the loop free of branches -- hipcc then waits for the CURRENT tile's loads only, not for the prefetch)
    if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st1 = __builtin_amdgcn_s_memtime(); }
    long row = max(cand_row(r), 0L);
    float4 bcur[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) bcur[kk] = *reinterpret_cast<const float4 *>(a.z + row * H + 4 * q + 16 * kk);

    if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st2 = __builtin_amdgcn_s_memtime(); }
    for (int tile = 0; tile < ntiles; ++tile) {
        // prefetch the next tile's rows while this one is multiplied
        float4 bnext[KK];
        const long nrow = max(cand_row(16 * (tile + 1) + r), 0L);        // (past the last tile: row 0, unused)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) bnext[kk] = *reinterpret_cast<const float4 *>(a.z + nrow * H + 4 * q + 16 * kk);

        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#if defined(NCE_DBG) && (NCE_DBG & 2)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) acc[kk & 3] += bcur[kk].x + bcur[kk].y + bcur[kk].z + bcur[kk].w;   // probe: no MFMA
#else
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[kk].x, bcur[kk].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[kk].y, bcur[kk].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[kk].z, bcur[kk].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[kk].w, bcur[kk].w, acc, 0, 0, 0);
        }
#endif
        // the prefetched rows are taken over HERE, before this tile's logits are stored: vmcnt counts stores too, and a
        // wait for the rows placed behind the stores would sit out a store round trip per tile
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) bcur[kk] = bnext[kk];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tile == 0) {
            // positive of step k = column k of this tile: lane 16q + (4q+e) of row group q holds it
#pragma unroll
            for (int e = 0; e < 4; ++e) pos[e] = __shfl(acc[e], 16 * q + 4 * q + e, 64) * inv_h;
        } else {
            const int j = 16 * (tile - 1) + r;              // negative slot of this lane's column
            if (j < a.Nneg) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * q + e;
                    const float x = acc[e] * inv_h;
                    if (k < a.K) a.logits[((long)bt * a.K + k) * (a.Nneg + 1) + 1 + j] = x;       // 16 lanes: 64 consecutive bytes
                    const float mn = fmaxf(m[e], x);
                    s[e] = s[e] * expf(m[e] - mn) + expf(x - mn);    // m = -inf: s = 0, exp(-inf) = 0
                    m[e] = mn;
                }
            }
        }
    }

    if (a.stamps) st3 = __builtin_amdgcn_s_memtime();
    // merge (m, s) over the 16 lanes of the row group; lanes that saw no candidate carry (-inf, 0)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const float m2 = __shfl_xor(m[e], off, 64), s2 = __shfl_xor(s[e], off, 64);
            const float mn = fmaxf(m[e], m2);
            const float f1 = (m[e] == -INFINITY) ? 0.f : expf(m[e] - mn);
            const float f2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
            s[e] = s[e] * f1 + s2 * f2;
            m[e] = mn;
        }
    }
    if (r == 0) {
        const float wgt = a.weights != nullptr ? a.weights[bt] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * q + e;
            if (k < a.K) {
                const float mx = fmaxf(m[e], pos[e]);
                const float se = s[e] * expf(m[e] - mx) + expf(pos[e] - mx);
                const float lse = mx + logf(se);
                a.logits[((long)bt * a.K + k) * (a.Nneg + 1)] = pos[e];
                a.lse[(long)bt * a.K + k] = lse;
                a.lossp[(long)bt * a.K + k] = wgt * (lse - pos[e]);
                a.hit[(long)bt * a.K + k] = pos[e] >= m[e] ? 1.f : 0.f;       // first-index-wins argmax
            }
        }
    }
    if (a.stamps && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *o = a.stamps + (long)bt * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = __builtin_amdgcn_s_memtime(); o[5] = __builtin_amdgcn_s_memrealtime();
    }
}

// The same forward pass for H = 256 / 512 as a STREAM through LDS.  Lanes that each pull 16-byte pieces of "their" row (the
// kernel above) make every load instruction touch 16 half-used cache lines, and a wave that starts by loading its own P
// rows idles its SIMD for microseconds; measured, that kernel spends a third of its time multiplying.  Here:
//   * one global_load_lds_dwordx4 moves one KiB of whole row pieces (KC = 128 floats of two rows) into LDS, the rows chosen
//     per half wave; an ELEMENT of the stream = 16 rows x KC floats = 8 such pieces, a piece 32 bytes past a KiB multiple
//     from the previous one (fragment reads -- lane (n, q): 16 bytes at row n, floats 4q + 16kk -- at most 2-way conflicts);
//   * persistent waves, TWO per SIMD (an LDS-DMA instruction holds a wave's issue port for ~100 cycles: one wave requests
//     while the other multiplies): a wave walks (b,t) = blockIdx, + gridDim, ... and its stream is P(b,t) [the K prediction
//     rows, read from LDS into the A-operand registers], the candidate tiles, P of the NEXT (b,t) with its index list, ...:
//     element e + 1 is requested before element e is used (s_waitcnt vmcnt(n_{e+1}): everything older than the newest
//     request -- the logits stored meanwhile included -- has landed), two LDS slots per wave;
//   * nothing else loads: no prologue, no staging registers.
constexpr int NCE_KC = 128;                                  // floats of a row per element
constexpr int NCE_PIECE = 1024 + 32;                         // bytes between pieces in LDS
constexpr int NCE_PP = 16 * NCE_KC * 4 / 1024;               // pieces per element (8)
constexpr int NCE_SLOT = NCE_PP * NCE_PIECE;
constexpr int NCE_LIST = 1024;                               // bytes per index list in LDS: the rows (int) of <= 256 negatives
template <int H> __global__ __launch_bounds__(64, 2) void infonce_fwd_dma_kernel(NceArgs a, int n_bt)
{
    static_assert(H % NCE_KC == 0 && NCE_KC == 128, "two rows per piece");
    constexpr int HS = H / NCE_KC;                           // elements per candidate tile (and per P)
    constexpr int FR = NCE_KC / 16;                          // fragments (16 bytes per lane) per element
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const float inv_h = 1.f / H, sc2 = 1.4426950408889634f / H;
    extern __shared__ __attribute__((aligned(1024))) char dma_lds[];
    // the two index lists first, then the two slots: a slot's address minus the largest K-slice offset (glds16x8: M0 = address
    // - offset) must not go below zero -- the hardware drops an LDS-DMA whose M0 is negative (tools/scratch/ldsdma_offset_probe.hip)
    static_assert(2 * NCE_LIST >= (HS - 1) * NCE_KC * 4, "slot addresses stay above the K-slice offsets");
    const char *lists = dma_lds;                                               // [2][NCE_LIST]
    const unsigned lds0 = lds_addr(dma_lds) + 2 * NCE_LIST;
    const int ntiles = 1 + (a.Nneg + 15) / 16;
    unsigned wslot = 0;                                                        // slot the next request fills
    unsigned rslot = 0;                                                        // slot of the element to use next
    const int up = lane >> 5;                                                  // which of a piece's two rows this lane moves
    const unsigned vcol = (lane & 31) * 16;

    // ---- requests: NCE_PP LDS-DMA instructions each (+ 1 for an index list), all eight pieces of an element in one statement
    // (glds16x8).  A lane's byte offsets into z for the eight pieces of a tile -- row of candidate 2 i + up, column vcol -- are
    // formed once per tile (vr_*); the K slice of a request is the instruction's offset field.
    const unsigned zmax = (unsigned)(a.b * a.T - 1);
    auto rows_of_list = [&](const char *list, int tile, u32x2_t (&pr)[NCE_PP / 2]) {   // negatives 16 (tile - 1) + ..: issue only
        const unsigned la = lds_addr(list) + (unsigned)(16 * (tile - 1) + up) * 4u;
        pr[0] = lds_read2<0, 2>(la);   pr[1] = lds_read2<4, 6>(la);   pr[2] = lds_read2<8, 10>(la);   pr[3] = lds_read2<12, 14>(la);
    };
    auto rows_to_offsets = [&](const u32x2_t (&pr)[NCE_PP / 2], unsigned (&vr)[NCE_PP]) {   // after the reads' wait
#pragma unroll
#if defined(NCE_DBG) && (NCE_DBG & 8)
        for (int i = 0; i < NCE_PP; ++i) vr[i] = (min(pr[i >> 1][i & 1], zmax) & 2047u) * (unsigned)(H * 4) + vcol;   // probe: a 2 MiB table (L2-resident)
#elif defined(NCE_DBG) && (NCE_DBG & 16)
        for (int i = 0; i < NCE_PP; ++i) vr[i] = (min(pr[i >> 1][i & 1], zmax) & 15u) * (unsigned)(H * 4) + vcol;     // probe: 16 hot rows
#else
        for (int i = 0; i < NCE_PP; ++i) vr[i] = min(pr[i >> 1][i & 1], zmax) * (unsigned)(H * 4) + vcol;   // (slots past Nneg: any valid row)
#endif
    };
    auto req_rows = [&](const unsigned (&vr)[NCE_PP], int half) {
        const unsigned dst = lds0 + wslot;
#if defined(NCE_ABL) && (NCE_ABL & 8)
        half = 0;                                           // probe: every K slice re-reads slice 0 (a gather table of half the size)
#endif
        if (half == 0) glds16x8<NCE_PIECE, 0>(dst, vr, a.z);
        else if (half == 1) glds16x8<NCE_PIECE, NCE_KC * 4>(dst, vr, a.z);
        else if (half == 2) glds16x8<NCE_PIECE, 2 * NCE_KC * 4>(dst, vr, a.z);
        else glds16x8<NCE_PIECE, 3 * NCE_KC * 4>(dst, vr, a.z);
        wslot ^= NCE_SLOT;
    };
    static_assert(HS <= 4, "K slices through the offset field");
    auto req_p = [&](int bt, int half) {                   // the K prediction rows of (b,t) (rows >= K: row 0, unused)
        const int bb = bt / a.W, t = bt - bb * a.W;
        const unsigned dst = lds0 + wslot;
        if (a.p_packed) {
            // one tensor [b * p_rows][K * H]: row k of (b, t) is K-row number (bb p_rows + t) K + k of it
#if defined(NCE_DBG) && (NCE_DBG & 32)
            const unsigned r0 = (unsigned)(((bb * a.p_rows + t) & 63) * a.K);            // probe: the predictions of 64 (b,t) only (no P traffic)
#else
            const unsigned r0 = (unsigned)((bb * a.p_rows + t) * a.K);
#endif
            unsigned vp[NCE_PP];
#pragma unroll
            for (int i = 0; i < NCE_PP; ++i) vp[i] = (r0 + (2 * i + up < a.K ? 2 * i + up : 0)) * (unsigned)(H * 4) + vcol;
            // Non-temporal loads here (-DNCE_P_NT) take the kernel ALONE from 118 to 113 us -- the read-once rows no longer push z
            // out of L2 -- but the backward pass re-reads P, finds it gone from the caches and loses 0.03 ms: net loss in the step.
#ifdef NCE_P_NT
            constexpr bool PNT = true;
#else
            constexpr bool PNT = false;
#endif
            if (half == 0) glds16x8<NCE_PIECE, 0, PNT>(dst, vp, a.Pk[0]);
            else if (half == 1) glds16x8<NCE_PIECE, NCE_KC * 4, PNT>(dst, vp, a.Pk[0]);
            else if (half == 2) glds16x8<NCE_PIECE, 2 * NCE_KC * 4, PNT>(dst, vp, a.Pk[0]);
            else glds16x8<NCE_PIECE, 3 * NCE_KC * 4, PNT>(dst, vp, a.Pk[0]);
        } else {
            const long off = ((long)bb * a.p_rows + t) * a.p_stride + half * NCE_KC;
#pragma unroll
            for (int i = 0; i < NCE_PP; ++i) {
                const float *p0 = a.Pk[2 * i < a.K ? 2 * i : 0] + off, *p1 = a.Pk[2 * i + 1 < a.K ? 2 * i + 1 : 0] + off;
                glds16_addr(dst + i * NCE_PIECE, reinterpret_cast<const char *>(up ? p1 : p0) + vcol);   // (the K predictions may be K tensors)
            }
        }
        wslot ^= NCE_SLOT;
    };
    auto req_list = [&](int bt, int par) {                 // rows of (b,t)'s negatives, in slot order
        const unsigned dst = lds_addr(lists) + par * NCE_LIST;       // (the piece reads up to 1 KiB past a short list: inside `saved`)
        glds16(dst, lane * 16, reinterpret_cast<const char *>(a.ext) + (long)bt * a.Nneg * 4);
    };
    auto wait_newest = [&](int n) {                        // everything older than the newest n DMA instructions has landed
        if (n == NCE_PP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == NCE_PP + 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(NCE_PP == 8, "the waits above are written for 8 pieces");
    // lane (n = r, q): row n = piece n / 2, row n % 2 of it; floats 4q + 16kk
    const unsigned fbase = lds0 + (r >> 1) * NCE_PIECE + (r & 1) * (NCE_KC * 4) + q * 16;
    auto read_frags = [&](frag_t (&f)[FR]) {
        const unsigned fa = fbase + rslot;
        f[0] = lds_read16<0>(fa);     f[1] = lds_read16<64>(fa);    f[2] = lds_read16<128>(fa);   f[3] = lds_read16<192>(fa);
        f[4] = lds_read16<256>(fa);   f[5] = lds_read16<320>(fa);   f[6] = lds_read16<384>(fa);   f[7] = lds_read16<448>(fa);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                     :
                     : "memory");
        rslot ^= NCE_SLOT;
    };
    // the same with the row indices of the NEXT tile requested alongside (one wait for both)
    auto read_frags_rows = [&](frag_t (&f)[FR], const char *list, int tile_next, unsigned (&vr)[NCE_PP]) {
        const unsigned fa = fbase + rslot;
        u32x2_t pr[NCE_PP / 2];
        rows_of_list(list, tile_next, pr);
        f[0] = lds_read16<0>(fa);     f[1] = lds_read16<64>(fa);    f[2] = lds_read16<128>(fa);   f[3] = lds_read16<192>(fa);
        f[4] = lds_read16<256>(fa);   f[5] = lds_read16<320>(fa);   f[6] = lds_read16<384>(fa);   f[7] = lds_read16<448>(fa);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(pr[0]), "+v"(pr[1]),
                       "+v"(pr[2]), "+v"(pr[3])
                     :
                     : "memory");
        rows_to_offsets(pr, vr);
        rslot ^= NCE_SLOT;
    };

    float4 areg[HS * FR];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int bt = blockIdx.x, par = 0;
    if (bt >= n_bt) return;
#if defined(NCE_STAGGER)
    {
        // probe: the second wave of a SIMD (odd wave slot: HW_ID bits 3:0) starts NCE_STAGGER x 64 cycles late, so that one wave of
        // the pair requests while the other multiplies from the first element on
        const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        if (hwid & 1u) __builtin_amdgcn_s_sleep(NCE_STAGGER);
    }
#endif
    req_p(bt, 0);
    req_list(bt, 0);
    for (; bt < n_bt; bt += gridDim.x, par ^= 1) {
        const int bb = bt / a.W, t = bt - bb * a.W;
        const int bt_next = bt + gridDim.x;
        // ---- the P elements (the first one travelled with the index list): into the A-operand registers
        // byte offsets of the rows of the tile whose K slices are being requested; replaced by the next tile's (read from the
        // list beside the fragments of K slice HS - 2) once the last slice of the current one has been requested
        unsigned vr[NCE_PP];
#pragma unroll
        for (int i = 0; i < NCE_PP; ++i)                    // tile 0 = the positives z[b][t+1 ..]: no list needed
            vr[i] = (unsigned)(2 * i + up < a.K ? bb * a.T + t + 1 + 2 * i + up : 0) * (unsigned)(H * 4) + vcol;
#pragma unroll
        for (int half = 0; half < HS; ++half) {
            if (half + 1 < HS) req_p(bt, half + 1);
            else req_rows(vr, 0);
            wait_newest(NCE_PP);
            frag_t f[FR];
            read_frags(f);
#pragma unroll
            for (int kk = 0; kk < FR; ++kk) areg[half * FR + kk] = __builtin_bit_cast(float4, f[kk]);
        }
        // lists[par] has landed (it is older than the P elements' successors)
        const char *const list = lists + par * NCE_LIST;
        // Cross-entropy in base 2 on the raw accumulators (v_exp_f32 / v_log_f32 are single instructions, the natural-base
        // library forms ~15 each): score x = acc / H, x2 = x log2(e).
        float posacc[4] = {0.f, 0.f, 0.f, 0.f};              // raw accumulators of the positives
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sm[4] = {0.f, 0.f, 0.f, 0.f};       // base 2: running max, sum
        float macc[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};                                    // raw running max (argmax)
        // logits row of step k = 4q + e: lrow0 + e * lstep, past the positive's slot (rows >= K are never stored)
        float *const lrow0 = a.logits + ((long)bt * a.K + 4 * q) * (a.Nneg + 1) + 1;
        const int lstep = a.Nneg + 1;
        for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll
            for (int half = 0; half < HS; ++half) {
                // request the next element of the stream, then take this one
                int newest = NCE_PP;
                if (half + 1 < HS) {
                    req_rows(vr, half + 1);
                } else if (tile + 1 < ntiles) {
                    req_rows(vr, 0);                                        // (the rows of tile + 1 by now)
                } else if (bt_next < n_bt) {
                    req_p(bt_next, 0);
                    req_list(bt_next, par ^ 1);
                    newest = NCE_PP + 1;
                } else {
                    newest = 0;
                }
                wait_newest(newest);
                frag_t bf[FR];
                if (half == HS - 2) read_frags_rows(bf, list, min(tile + 1, ntiles - 1), vr);   // + the rows of tile + 1 (unused after the last tile)
                else read_frags(bf);
#pragma unroll
                for (int kk = 0; kk < FR; kk += 2) {
#if defined(NCE_ABL) && (NCE_ABL & 4)
                    acc0[0] += __builtin_bit_cast(f32x4, bf[kk])[0]; acc1[1] += __builtin_bit_cast(f32x4, bf[kk + 1])[1];
                    continue;
#endif
                    const float4 p0 = areg[half * FR + kk], p1 = areg[half * FR + kk + 1];
                    const f32x4 b0 = __builtin_bit_cast(f32x4, bf[kk]), b1 = __builtin_bit_cast(f32x4, bf[kk + 1]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.x, b0[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.x, b1[0], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.y, b0[1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.y, b1[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.z, b0[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.z, b1[2], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.w, b0[3], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.w, b1[3], acc1, 0, 0, 0);
                }
            }
            // ---- a candidate tile is complete
            f32x4 acc = acc0 + acc1;
            acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1 = acc0;
            if (tile == 0) {
                // positive of step k = column k of this tile: lane 16q + (4q+e) of row group q holds it
#pragma unroll
                for (int e = 0; e < 4; ++e) posacc[e] = __shfl(acc[e], 16 * q + 4 * q + e, 64);
            } else {
                const int j = 16 * (tile - 1) + r;              // negative slot of this lane's column
                if (j < a.Nneg) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#if !(defined(NCE_ABL) && (NCE_ABL & 1))
                        // slot order: the 16 lanes of a row group write 64 consecutive bytes (in the caller's order these were
                        // 64 scattered dwords per instruction -- 206 MB of write traffic for 46 MB of logits, and every store had
                        // to retire before the next element's wait could pass: 17 % of the kernel)
#ifdef NCE_LOGITS_NT                                        /* (as NCE_P_NT: the backward pass reads the logits back) */
                        if (4 * q + e < a.K) __builtin_nontemporal_store(acc[e] * inv_h, lrow0 + e * lstep + j);
#else
                        if (4 * q + e < a.K) lrow0[e * lstep + j] = acc[e] * inv_h;
#endif
#endif
#if defined(NCE_ABL) && (NCE_ABL & 2)
                        m[e] = fmaxf(m[e], acc[e]);
#else
                        const float x2 = acc[e] * sc2;
                        const float mn = fmaxf(m[e], x2);
                        sm[e] = sm[e] * __builtin_amdgcn_exp2f(m[e] - mn) + __builtin_amdgcn_exp2f(x2 - mn);   // m = -inf: 0 * 0 + ..
                        m[e] = mn;
                        macc[e] = fmaxf(macc[e], acc[e]);
#endif
                    }
                }
            }
        }
        // merge over the 16 lanes of a row group (data-parallel primitives): the max first, ONE rescale, then the sum.
        // Lanes that saw no candidate carry (-inf, 0).
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float mx = m[e], ma = macc[e];
            mx = fmaxf(mx, row_dpp<0xB1>(mx)); mx = fmaxf(mx, row_dpp<0x4E>(mx)); mx = fmaxf(mx, row_dpp<0x141>(mx)); mx = fmaxf(mx, row_dpp<0x140>(mx));
            ma = fmaxf(ma, row_dpp<0xB1>(ma)); ma = fmaxf(ma, row_dpp<0x4E>(ma)); ma = fmaxf(ma, row_dpp<0x141>(ma)); ma = fmaxf(ma, row_dpp<0x140>(ma));
            float sv = m[e] == -INFINITY ? 0.f : sm[e] * __builtin_amdgcn_exp2f(m[e] - mx);
            sv += row_dpp<0xB1>(sv); sv += row_dpp<0x4E>(sv); sv += row_dpp<0x141>(sv); sv += row_dpp<0x140>(sv);
            m[e] = mx; sm[e] = sv; macc[e] = ma;
        }
        if (r == 0) {
            typedef const __attribute__((address_space(4))) float *const_float_p;        // (a scalar load: a vector load here would
            const float wgt = a.weights != nullptr ? ((const_float_p)(unsigned long long)a.weights)[bt] : 1.f;   //  count in vmcnt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * q + e;
                if (k < a.K) {
                    const float pos = posacc[e] * inv_h, pos2 = posacc[e] * sc2;
                    const float mx = fmaxf(m[e], pos2);
                    const float se = sm[e] * __builtin_amdgcn_exp2f(m[e] - mx) + __builtin_amdgcn_exp2f(pos2 - mx);
                    const float lse = (mx + __builtin_amdgcn_logf(se)) * 0.693147180559945f;
                    lrow0[e * lstep - 1] = pos;
                    a.lse[(long)bt * a.K + k] = lse;
                    a.lossp[(long)bt * a.K + k] = wgt * (lse - pos);
                    a.hit[(long)bt * a.K + k] = posacc[e] >= macc[e] ? 1.f : 0.f;  // first-index-wins argmax (raw: exact ties tie)
                }
            }
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
    // eight loads requested before the first is used (a plain `for (r = tid; r < rows; r += 256) s += ...` waits for every load
    // before it requests the next: 29 round trips in a row at the headline shape, 12-33 us for 360 KB); fixed order per thread
    float s = 0.f;
    for (long r0 = threadIdx.x; r0 < rows; r0 += 8L * blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long r = r0 + (long)u * blockDim.x;
            v[u] = src[(r < rows ? r : rows - 1) * K + kk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (r0 + (long)u * blockDim.x < rows) ? v[u] : 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) (k < K ? losses : acc)[kk] = red[0] * inv_count;
}

// Backward: ONE WAVE per (b,t) of the b*T grid (t >= W only zeroes its dP row).  LDS (private to the wave)
// holds dS[16][lw] = d loss / d <P_k, cand_g> and the candidates' z-row indices.
//   dP^T[d][k] += sum_g cand_g[d] * dS[k][g]    16x16x4, A = candidate rows streamed from L2 as float4s: lane
//                 (i, q) reads cand_{4s+q}[64T + 4i ..+3] and feeds the 4 row-interleaved tiles d = 64T + 4i + e
//   dz[row_g][d] += sum_k dS[k][g] * P_k[d]     32x32x2.  The b*W*(K + Nneg) contribution rows are either added to dz
//                 with fp32 atomics (128-byte segments; the atomic units' rate then IS the kernel's time) or, when
//                 a.vbuf is set, stored once to vbuf[(b,t)][candidate][H] and summed per target row by
//                 infonce_dz_gather_kernel from counting-sorted reference lists: plain streaming traffic, and a
//                 summation order that does not change from run to run.
template <int H> __global__ __launch_bounds__(64) void infonce_bwd_kernel(NceArgs a)
{
    constexpr int DG = (H + 63) / 64;          // groups of 4 interleaved 16-row d tiles (H = 32: half a group)
    constexpr int DT32 = H / 32;               // 32-wide d tiles of the dz product
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dS = smem;                                              // [16][lw]
    int *rowidx = reinterpret_cast<int *>(dS + NCE_ROWS * a.lw);   // [lw]

    const int bb = blockIdx.x / a.p_rows, t = blockIdx.x - bb * a.p_rows;
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    if (t >= a.W) {            // only reached when p_rows == T: zero the unused rows of the GEMM-shaped dP
        for (int k = 0; k < a.K; ++k) {
            float *row = a.dPk[k] + ((long)bb * a.p_rows + t) * a.p_stride;
            for (int i = lane; i < H / 4; i += 64) reinterpret_cast<float4 *>(row)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const long bt = (long)bb * a.W + t;
    const int ncand = NCE_POS + a.Nneg;
    const int npad = a.lw - 4;                                     // multiple of 32 >= ncand
    const float inv_h = 1.f / H;
    const float wgt = (a.weights != nullptr ? a.weights[bt] : 1.f) * a.inv_count;

    for (int g = lane; g < a.lw; g += 64) rowidx[g] = g < ncand ? (int)nce_row(a, bb, t, g) : -1;
    // dS[k][g]: one prediction step k at a time (wave-uniform), lanes over the candidates: the logits row of (b,t,k)
    // is read coalesced and nothing is divided
    for (int k = 0; k < NCE_ROWS; ++k) {
        float coef = 0.f, l = 0.f;
        const float *lg = a.logits;
        if (k < a.K) {
            coef = a.dloss[k] * wgt * inv_h;
            lg = a.logits + (bt * a.K + k) * (a.Nneg + 1);
            l = a.lse[bt * a.K + k];
        }
        for (int g = lane; g < a.lw; g += 64) {
            float v = 0.f;
            if (k < a.K) {
                if (g < NCE_POS) {
                    if (g == k) v = coef * (expf(lg[0] - l) - 1.f);
                } else if (g - NCE_POS < a.Nneg) {
                    v = coef * expf(lg[1 + g - NCE_POS] - l);
                }
            }
            dS[k * a.lw + g] = v;
        }
    }
    __syncthreads();                           // one wave: orders the LDS writes before the reads below

    // ---- dP ------------------------------------------------------------------------------------
    f32x4 dp[DG][4];
#pragma unroll
    for (int T4 = 0; T4 < DG; ++T4)
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[T4][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < npad / 4; s0 += 4) {          // npad is a multiple of 32 -> npad/4 of 8
        int rows4[4];
        float bv4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = 4 * (s0 + u) + q;
            rows4[u] = rowidx[g];
            bv4[u] = dS[r * a.lw + g];
        }
        float4 av[4][DG];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *src = a.z + (long)(rows4[u] >= 0 ? rows4[u] : 0) * H + 4 * r;
#pragma unroll
            for (int T4 = 0; T4 < DG; ++T4) {
                av[u][T4] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rows4[u] >= 0 && 64 * T4 + 4 * r < H) av[u][T4] = *reinterpret_cast<const float4 *>(src + 64 * T4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int T4 = 0; T4 < DG; ++T4) {
                dp[T4][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][T4].x, bv4[u], dp[T4][0], 0, 0, 0);
                dp[T4][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][T4].y, bv4[u], dp[T4][1], 0, 0, 0);
                dp[T4][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][T4].z, bv4[u], dp[T4][2], 0, 0, 0);
                dp[T4][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][T4].w, bv4[u], dp[T4][3], 0, 0, 0);
            }
    }
    // dp[T4][e][reg] = dP[k = r][d = 64*T4 + 4*(4q + reg) + e]  -> 16 consecutive d per (T4, lane)
    if (r < a.K) {
        float *dprow = a.dPk[r] + ((long)bb * a.p_rows + t) * a.p_stride;
#pragma unroll
        for (int T4 = 0; T4 < DG; ++T4)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (64 * T4 + 16 * q + 4 * reg < H)
                    *reinterpret_cast<float4 *>(dprow + 64 * T4 + 16 * q + 4 * reg) =
                        make_float4(dp[T4][0][reg], dp[T4][1][reg], dp[T4][2][reg], dp[T4][3][reg]);
    }

    // ---- dz ------------------------------------------------------------------------------------
    const int r32 = lane & 31, h = lane >> 5;
    const long prow_off = ((long)bb * a.p_rows + t) * a.p_stride;
    if (a.vbuf != nullptr) {
        // the contribution rows are formed and stored by infonce_dz_store_kernel (its own launch: this kernel's register
        // budget allows two waves per SIMD, too few to overlap the row gather above with 1 GB of stores)
        float *dsg = a.ds_buf + bt * (long)(NCE_ROWS + 1) * a.lw;
        for (int i = lane; i < NCE_ROWS * a.lw; i += 64) dsg[i] = dS[i];
        for (int g = lane; g < a.lw; g += 64) reinterpret_cast<int *>(dsg)[NCE_ROWS * a.lw + g] = rowidx[g];
        return;
    }
    for (int dt = 0; dt < DT32; ++dt) {
        float bvals[NCE_ROWS / 2];
#pragma unroll
        for (int kp = 0; kp < NCE_ROWS / 2; ++kp) {
            const int k = 2 * kp + h;
            bvals[kp] = k < a.K ? a.Pk[k][prow_off + dt * 32 + r32] : 0.f;
        }
        for (int ct = 0; ct < npad / 32; ++ct) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kp = 0; kp < NCE_ROWS / 2; ++kp) {
                const float av = dS[(2 * kp + h) * a.lw + ct * 32 + r32];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvals[kp], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int row = rowidx[ci];
                if (row >= 0) atomicAdd(a.dz + (long)row * H + dt * 32 + r32, acc[e]);
            }
        }
    }
}

// Second half of the backward pass when the dz contributions are stored (a.vbuf): one wave per (b,t) reloads its
// dS[16][lw] and candidate rows (written by infonce_bwd_kernel) and forms  V[cand][d] = sum_k dS[k][cand] * P_k[d]
// with the 32x32x2 MFMA, candidate tiles outermost so that the H/32 128-byte pieces of a row are stored back to back.
template <int H> __global__ __launch_bounds__(64) void infonce_dz_store_kernel(NceArgs a)
{
    constexpr int DT32 = H / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dS = smem;                                              // [16][lw] then lw row indices
    const int *rowidx = reinterpret_cast<const int *>(dS + NCE_ROWS * a.lw);
    const int bb = blockIdx.x / a.W, t = blockIdx.x - bb * a.W;
    const long bt = blockIdx.x;
    const int lane = threadIdx.x, r32 = lane & 31, h = lane >> 5;
    const int npad = a.lw - 4;
    const float4 *src = reinterpret_cast<const float4 *>(a.ds_buf + bt * (long)(NCE_ROWS + 1) * a.lw);
    for (int i = lane; i < (NCE_ROWS + 1) * a.lw / 4; i += 64) reinterpret_cast<float4 *>(dS)[i] = src[i];
    const long prow_off = ((long)bb * a.p_rows + t) * a.p_stride;
    float bvals[DT32][NCE_ROWS / 2];
#pragma unroll
    for (int dt = 0; dt < DT32; ++dt)
#pragma unroll
        for (int kp = 0; kp < NCE_ROWS / 2; ++kp) {
            const int k = 2 * kp + h;
            bvals[dt][kp] = k < a.K ? a.Pk[k][prow_off + dt * 32 + r32] : 0.f;
        }
    __syncthreads();
    float *vb = a.vbuf + bt * (long)(a.K + a.Nneg) * H + r32;
    for (int ct = 0; ct < npad / 32; ++ct) {
        float av[NCE_ROWS / 2];
#pragma unroll
        for (int kp = 0; kp < NCE_ROWS / 2; ++kp) av[kp] = dS[(2 * kp + h) * a.lw + ct * 32 + r32];
        int voff[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            voff[e] = rowidx[ci] >= 0 ? (ci < NCE_POS ? ci : ci - NCE_POS + a.K) * H : -1;
        }
#pragma unroll
        for (int dt = 0; dt < DT32; ++dt) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kp = 0; kp < NCE_ROWS / 2; ++kp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp], bvals[dt][kp], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (voff[e] >= 0) vb[voff[e] + dt * 32] = acc[e];
        }
    }
}

// Backward at encoder width 256 / 512 with stored contributions: both halves above in ONE workgroup per (b,t), H/128 waves
// that each own 128 of the H channels -- half the accumulators and operands of the one-wave form, so more waves per SIMD
// stay resident and the candidate-row gathers of some overlap the 1 GB of contribution-row stores of others.
//   phase 0  dS[16][lw] and the candidates' z rows into LDS.  ONE round of global loads: the index lists and the K logits
//            rows are requested together;
//            under load a dependent global load costs microseconds
//   phase 1  dP[k][d] (as infonce_bwd_kernel, d in this wave's 128 channels), the rows of the next 16 candidates
//            requested before the current 16 are multiplied; phase 2 runs INSIDE phase 1's loop, group by group
//   phase 2  V[cand][d] = sum_k dS[k][cand] * P_k[d]  16x16x4: M = d, so a lane holds four consecutive channels of one
//            candidate; 16 candidates x 64 channels go through a wave-private LDS tile and leave as 256-byte row pieces,
//            16 bytes per lane (the one-dword-per-lane form of infonce_dz_store_kernel stored at 3.9 TB/s).  The
//            contribution buffer keeps 16 rows for the positive slots, so no store is conditional (rows of unused slots
//            are written as zeros and never read).  The row stores are non-temporal: 1.2 GB of write-once data otherwise
//            sweeps the gathered z rows out of L2 (PMC: 928 MB fetched per launch for 970 MB of gathers).
// NKK = ceil(K / 4).  Needs Nneg % 16 == 0, lw <= 320 and K * (Nneg + 1) floats within the staging tiles.
constexpr int NCE_SROW = 68;               // staging tile row: 64 channels + 4 floats
template <int H, int NKK> __global__ __launch_bounds__(H / 2) void infonce_bwd_fused_kernel(NceArgs a)
{
    constexpr int NW = H / 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dS = smem;                                              // [16][lw]
    int *rowidx = reinterpret_cast<int *>(dS + NCE_ROWS * a.lw);   // [lw]
    float *stage0 = reinterpret_cast<float *>(rowidx + a.lw);      // [NW][16][NCE_SROW]; phase 0: the K logits rows
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    float *stage = stage0 + wave * 16 * NCE_SROW;

    const int bb = blockIdx.x / a.p_rows, t = blockIdx.x - bb * a.p_rows;
    if (t >= a.W) {            // only reached when p_rows == T: zero the unused rows of the GEMM-shaped dP
        for (int k = 0; k < a.K; ++k) {
            float *row = a.dPk[k] + ((long)bb * a.p_rows + t) * a.p_stride;
            for (int i = threadIdx.x; i < H / 4; i += 64 * NW) reinterpret_cast<float4 *>(row)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const long bt = (long)bb * a.W + t;
    const int ncand = NCE_POS + a.Nneg;
    const int npad = a.lw - 4;                                     // multiple of 32 >= ncand
    const float inv_h = 1.f / H;
    const float wgt = (a.weights != nullptr ? a.weights[bt] : 1.f) * a.inv_count;
    unsigned long long *stamp = a.stamps != nullptr && threadIdx.x == 0 ? a.stamps + bt * 8 : nullptr;
    if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();

    // ---- phase 0 -------------------------------------------------------------------------------
    // (loads are unconditional, from clamped addresses, and selected afterwards: a load under a lane condition becomes a
    //  branch with its own wait)
    constexpr int GI = 5;                                          // candidate slots per lane: lw <= 320
    constexpr int KW = NCE_ROWS / NW;                              // prediction steps per wave: k = wave + NW * kk
    const int lrow = a.Nneg + 1;
    int gi[GI];                                                    // where candidate g = lane + 64 i sits in a logits row
#pragma unroll
    for (int i = 0; i < GI; ++i) {
        const int j = lane + 64 * i - NCE_POS;
        gi[i] = j >= 0 && j < a.Nneg ? 1 + j : 0;                  // (slot k < 16, the positive of step k: element 0)
    }
    float coef[KW], lsek[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
        const int kc = min(wave + NW * kk, a.K - 1);
        coef[kk] = a.dloss[kc] * wgt * inv_h;
        lsek[kk] = a.lse[bt * a.K + kc];
    }
    {
        const float *lg = a.logits + bt * a.K * lrow;              // K rows, contiguous
        const int n = a.K * lrow;
        constexpr int NT = 16 * NCE_SROW / 64;                    // the staging tiles hold NT floats per thread (host: n fits)
        float tmp[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) tmp[i] = lg[min((int)threadIdx.x + 64 * NW * i, n - 1)];
        for (int g = threadIdx.x; g < a.lw; g += 64 * NW) {
            const int j = min(max(g - NCE_POS, 0), a.Nneg - 1);
            const int neg = a.ext[bt * a.Nneg + j];
            const int pos = (g < a.K && t + 1 + g < a.T) ? bb * a.T + t + 1 + g : -1;
            rowidx[g] = g < NCE_POS ? pos : (g < ncand ? neg : -1);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if ((int)threadIdx.x + 64 * NW * i < n) stage0[threadIdx.x + 64 * NW * i] = tmp[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
        const int k = wave + NW * kk, kc = min(k, a.K - 1);
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int g = lane + 64 * i;
            const float e = expf(stage0[kc * lrow + gi[i]] - lsek[kk]);
            float v = g >= NCE_POS ? (g < ncand ? coef[kk] * e : 0.f) : (g == k ? coef[kk] * (e - 1.f) : 0.f);
            if (k >= a.K) v = 0.f;
            if (g < a.lw) dS[k * a.lw + g] = v;
        }
    }
    __syncthreads();
    if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();

    // ---- phases 1 and 2, one loop over groups of 16 candidates: the rows of group i + 1 are requested, the contribution rows
    // of group i are formed and stored (no gathered row needed), THEN the gathered rows of group i are multiplied into dP --
    // the gather's latency passes under the stores and the two kinds of traffic overlap inside a wave.
    const int d0 = 128 * wave;
    const long prow_off = ((long)bb * a.p_rows + t) * a.p_stride + d0;
    float pa[8][NKK];                                              // phase 2's A: P_{4kk + q}[d0 + 16*tile + r]
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int k = 4 * kk + q;
        const float *pk = a.Pk[min(k, a.K - 1)] + prow_off + r;
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) {
            const float v = pk[16 * tile];
            pa[tile][kk] = k < a.K ? v : 0.f;
        }
    }
    // row `cand` of the (b,t)'s NCE_POS + Nneg contribution rows; this lane: rows 4j + q of a tile, its 16-byte piece r
    float *vb = a.vbuf + (bt * (long)ncand + q) * H + d0 + 4 * r;
    // (a macro, not a lambda: inside a lambda the staging tile's address space is lost and its arrays go to scratch memory)
#define NCE_ROWS_GROUP(C0)                                                                                                     \
    {                                                                                                                          \
        const int c0_ = (C0);                                                                                                  \
        float bv[NKK];                                                                                                         \
        _Pragma("unroll") for (int kk = 0; kk < NKK; ++kk) bv[kk] = dS[(4 * kk + q) * a.lw + c0_ + r];                        \
        _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                               \
            f32x4 acc[4];                                                                                                      \
            _Pragma("unroll") for (int tl = 0; tl < 4; ++tl) acc[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};                            \
            _Pragma("unroll") for (int kk = 0; kk < NKK; ++kk)                                                                 \
                _Pragma("unroll") for (int tl = 0; tl < 4; ++tl)                                                               \
                    acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * half + tl][kk], bv[kk], acc[tl], 0, 0, 0);           \
            _Pragma("unroll") for (int tl = 0; tl < 4; ++tl)                                                                   \
                *reinterpret_cast<f32x4 *>(stage + r * NCE_SROW + 16 * tl + 4 * q) = acc[tl];                                  \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                             \
            __builtin_amdgcn_wave_barrier();                                                                                   \
            float4 v[4];                                                                                                       \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                      \
                v[j] = *reinterpret_cast<const float4 *>(stage + (4 * j + q) * NCE_SROW + 4 * r);                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                      \
                __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v[j]),                                                   \
                                            reinterpret_cast<f32x4 *>(vb + (long)(c0_ + 4 * j) * H + 64 * half));              \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                             \
            __builtin_amdgcn_wave_barrier();                                                                                   \
        }                                                                                                                      \
    }
    {
        f32x4 dp[2][4];
#pragma unroll
        for (int T4 = 0; T4 < 2; ++T4)
#pragma unroll
            for (int e = 0; e < 4; ++e) dp[T4][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 av[2][4][2];
        float bv4[2][4];
        auto request = [&](int buf, int s0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = 4 * (s0 + u) + q;
                const int row = rowidx[g];
                bv4[buf][u] = dS[r * a.lw + g];
                const float *src = a.z + (long)(row >= 0 ? row : 0) * H + d0 + 4 * r;      // (row 0 for the padding: dS is 0 there)
                av[buf][u][0] = *reinterpret_cast<const float4 *>(src);
                av[buf][u][1] = *reinterpret_cast<const float4 *>(src + 64);
            }
        };
        auto multiply = [&](int buf) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int T4 = 0; T4 < 2; ++T4) {
                    dp[T4][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][u][T4].x, bv4[buf][u], dp[T4][0], 0, 0, 0);
                    dp[T4][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][u][T4].y, bv4[buf][u], dp[T4][1], 0, 0, 0);
                    dp[T4][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][u][T4].z, bv4[buf][u], dp[T4][2], 0, 0, 0);
                    dp[T4][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][u][T4].w, bv4[buf][u], dp[T4][3], 0, 0, 0);
                }
        };
        // (measured: with ONE row buffer and three waves per SIMD the kernel takes the same 378 us -- it moves 1.2 GB of stores and
        //  0.6 GB of L2 misses at the fabric's ~4.8 TB/s either way)
        const int nb = npad / 16;                                  // even (npad is a multiple of 32); ncand / 16 or one more
        request(0, 0);
        for (int ib = 0; ib < nb; ib += 2) {
            request(1, 4 * (ib + 1));
            NCE_ROWS_GROUP(16 * ib)
            multiply(0);
            if (ib + 2 < nb) request(0, 4 * (ib + 2));
            if (16 * (ib + 1) < ncand) NCE_ROWS_GROUP(16 * (ib + 1))
            multiply(1);
        }
        // dp[T4][e][reg] = dP[k = r][d = d0 + 64*T4 + 4*(4q + reg) + e]
        if (r < a.K) {
            float *dprow = a.dPk[r] + ((long)bb * a.p_rows + t) * a.p_stride + d0;
#pragma unroll
            for (int T4 = 0; T4 < 2; ++T4)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    *reinterpret_cast<float4 *>(dprow + 64 * T4 + 16 * q + 4 * reg) =
                        make_float4(dp[T4][0][reg], dp[T4][1][reg], dp[T4][2][reg], dp[T4][3][reg]);
        }
    }
#undef NCE_ROWS_GROUP
    if (stamp) stamp[2] = __builtin_amdgcn_s_memtime();
    if (stamp) { stamp[3] = __builtin_amdgcn_s_memtime(); stamp[4] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- reference lists: which (b, t, negative) triples point at z row r (counting sort of ext by value) -------------
__global__ void nce_hist_kernel(const int32_t *ext, long n, int *counts)
{
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x)
        atomicAdd(counts + ext[o], 1);
}

// offsets[r] = sum of counts[0..r), offsets[rows] = total; one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void nce_scan_kernel(const int *counts, int rows, int *offsets)
{
    __shared__ int part[1024];
    const int per = (rows + 1023) / 1024;
    const int lo = min(rows, (int)threadIdx.x * per), hi = min(rows, lo + per);
    int sum = 0;
    for (int r = lo; r < hi; ++r) sum += counts[r];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int r = lo; r < hi; ++r) { offsets[r] = run; run += counts[r]; }
    if (threadIdx.x == 1023) offsets[rows] = part[1023];
}

__global__ void nce_fill_kernel(const int32_t *ext, long n, const int *offsets, int *cursor, int *entries)
{
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
        const int r = ext[o];
        entries[offsets[r] + atomicAdd(cursor + r, 1)] = (int)o;
    }
}

// the fill order depends on the atomics' arrival order: sort every row's list (one wave per row, rank sort in LDS) so
// that the sum below always runs in the same order.  Lists longer than NCE_SORT_MAX stay as filled (correct, but the
// order of their sum may then vary from run to run).
constexpr int NCE_SORT_MAX = 512;
__global__ __launch_bounds__(256) void nce_sort_kernel(const int *offsets, int *entries, int rows)
{
    __shared__ int buf[4][NCE_SORT_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const int beg = offsets[r], n = offsets[r + 1] - beg;
    if (n < 2 || n > NCE_SORT_MAX) return;
    for (int i = lane; i < n; i += 64) buf[wave][i] = entries[beg + i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (int i = lane; i < n; i += 64) {
        const int v = buf[wave][i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += buf[wave][j] < v;          // entries are distinct
        entries[beg + rank] = v;
    }
}

// (read-once data: a non-temporal load leaves the caches to what is read again)
__device__ __forceinline__ float4 nt_load4(const float4 *p)
{
    return __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)));
}

// dz[r][:] = sum of the stored contributions that point at z row r = (bb, t'): the positives of steps k = 0..K-1
// come from (bb, t' - 1 - k), the negatives from the row's reference list.  One wave per row, 16 bytes per lane.
#ifdef NCE_GATHER_WIDE                       /* (A/B build: the 68-register kernel of rounds 1-3, which cannot sit beside the matrix-pipe GRU backward) */
#define NCE_GATHER_BOUNDS __launch_bounds__(256)
#else
#define NCE_GATHER_BOUNDS __launch_bounds__(256, 8)
#endif
template <int H> __global__ NCE_GATHER_BOUNDS void infonce_dz_gather_kernel(const float *vbuf, const int *offsets,
                                                                                   const int *entries, float *dz, int b, int T,
                                                                                   int W, int K, int Nneg, int pos_rows)
{
    constexpr int C4 = H / 4;                        // float4 per row
    constexpr int PER = (C4 + 63) / 64;              // per lane
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (a wave walks rows blockIdx.x * 4 + wave, + 4 gridDim.x, ...: the launcher may cap the grid so that only a few waves per CU
    //  stream beside a recurrent kernel, whose hand-off latency grows with the streaming waves on its CU)
    for (int r = blockIdx.x * 4 + wave; r < b * T; r += gridDim.x * 4) {
    const int bb = r / T, tp = r - bb * T;
    const int vrows = pos_rows + Nneg;              // pos_rows = K (one-wave store kernel) or 16 (fused kernel)
    const float4 *v4 = reinterpret_cast<const float4 *>(vbuf);
    float4 acc[PER];
#pragma unroll
    for (int c = 0; c < PER; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_row = [&](long vrow) {
#pragma unroll
        for (int c = 0; c < PER; ++c) {
            if (lane + 64 * c < C4) {
                const float4 v = nt_load4(v4 + vrow * C4 + lane + 64 * c);
                acc[c].x += v.x; acc[c].y += v.y; acc[c].z += v.z; acc[c].w += v.w;
            }
        }
    };
    // newest rows first: the lists are sorted by (b,t) and the contribution rows were written in that order, so the tail of
    // every list is what the memory-side cache (256 MB of the 1.1 GB) still holds when this kernel starts
    const int beg = offsets[r], end = offsets[r + 1];
    int e = end;
    for (; e - 8 >= beg; e -= 8) {                   // eight rows in flight
        // (row numbers as 32-bit values -- b * W * (16 + Nneg) rows < 2^31 -- and the kernel held to 64 registers: the matrix-pipe
        //  recurrent backward it is meant to run beside leaves exactly 64 per SIMD free, 2 x 222 of 512)
        unsigned src[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned o = (unsigned)entries[e - 1 - i];
            src[i] = (o / (unsigned)Nneg) * (unsigned)vrows + (unsigned)pos_rows + o % (unsigned)Nneg;
        }
        float4 v[8][PER];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c < PER; ++c)
                v[i][c] = lane + 64 * c < C4 ? nt_load4(v4 + (long)src[i] * C4 + lane + 64 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c < PER; ++c) { acc[c].x += v[i][c].x; acc[c].y += v[i][c].y; acc[c].z += v[i][c].z; acc[c].w += v[i][c].w; }
    }
    for (; e > beg; --e) {
        const int o = entries[e - 1];
        add_row((long)(o / Nneg) * vrows + pos_rows + o % Nneg);
    }
    for (int k = 0; k < K; ++k) {
        const int t = tp - 1 - k;
        if (t >= 0 && t < W) add_row(((long)bb * W + t) * vrows + k);
    }
#pragma unroll
    for (int c = 0; c < PER; ++c)
        if (lane + 64 * c < C4) reinterpret_cast<float4 *>(dz)[(long)r * C4 + lane + 64 * c] = acc[c];
    }
}

// raw[0..n) -> batchIdx = raw % b, raw[n..2n) -> seqIdx = raw % (T-1) + 1 (draw order i = (bb*Nneg + nn)*W + t);
// ext[(bb*W + t)*Nneg + nn] = (seqIdx + t) mod T + batchIdx*T      (criterion.py:247-266, integer-exact)
__global__ void negidx_expand_kernel(const uint32_t *raw, int32_t *ext, int b, int T, int W, int Nneg)
{
    const long n = (long)b * Nneg * W;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
        const int nn = (int)(o % Nneg);
        const long bt = o / Nneg;
        const int t = (int)(bt % W), bb = (int)(bt / W);
        const long i = ((long)bb * Nneg + nn) * W + t;
        const uint32_t bi = raw[i] % (uint32_t)b;
        uint32_t seq = raw[n + i] % (uint32_t)(T - 1) + 1u + (uint32_t)t;
        if (seq >= (uint32_t)T) seq -= (uint32_t)T;
        ext[o] = (int32_t)(seq + bi * (uint32_t)T);
    }
}

// ------------------------------------------------------------------------------------------------
struct NceLayout {
    int b, T, K, W, Har, Henc, Nneg, lw;
    int Pr;                             // frames per window the context (and P, dP, dc) holds: T, or W when the caller hands over c[:, :W]
    float *P, *logits, *lse;            // saved
    int32_t *ext_sorted;                // saved: the negatives of every (b,t) sorted by z-row block
    unsigned short *perm;               // saved: slot -> the caller's negative number
    int nblk, rows_per_block;
    size_t saved_bytes;
    float *lossp, *hit, *dP, *wt, *tn, *tn_late;  // scratch
    float *vbuf;                        // scratch: [b*W][K + Nneg][Henc] contribution rows of the dz product
    float *ds_buf;                      // scratch: [b*W][17][lw]
    int *counts, *offsets, *entries;    // scratch: reference lists (counts doubles as the fill cursor)
    size_t tn_bytes, scratch_bytes;
    size_t lds_fwd, lds_bwd, lds_bwd_fused;
};

static bool nce_supported(int H) { return H == 32 || H == 64 || H == 128 || H == 256 || H == 512; }

static int nce_layout(NceLayout &l, int b, int T, int K, int Har, int Henc, int Nneg, void *saved, void *scratch)
{
    CPC_REQUIRE(nce_supported(Henc), "infonce: encoder dim %d not supported (32, 64, 128, 256, 512)", Henc);
    CPC_REQUIRE(b > 0 && K >= 1 && K <= 16 && T > K && Nneg >= 1 && Har >= 1,
                "infonce: bad shape b=%d T=%d K=%d (1..16) dim_ar=%d n_neg=%d", b, T, K, Har, Nneg);
    l.b = b; l.T = T; l.K = K; l.W = T - K; l.Har = Har; l.Henc = Henc; l.Nneg = Nneg; l.Pr = T;
    l.lw = (int)cdiv(NCE_POS + Nneg, 32) * 32 + 4;
    Carver sv(saved);
    l.P = sv.take<float>((size_t)b * T * K * Henc);
    l.logits = sv.take<float>((size_t)b * l.W * K * (Nneg + 1));
    l.lse = sv.take<float>((size_t)b * l.W * K);
    l.ext_sorted = sv.take<int32_t>((size_t)b * l.W * Nneg);
    l.perm = sv.take<unsigned short>((size_t)b * l.W * Nneg + 1024);     // (+ slack: the streaming kernel reads whole KiB pieces)
    // blocks of about 2 MB of z rows (an XCD's L2 holds 4 MB); CPC_NCE_NOSORT=1: one block = the drawn order
    static const bool nosort = getenv("CPC_NCE_NOSORT") != nullptr;
    const size_t zbytes = sizeof(float) * (size_t)b * T * Henc;
    l.nblk = nosort ? 1 : (int)std::min<size_t>(16, std::max<size_t>(1, (zbytes + (1u << 21) - 1) >> 21));
    l.rows_per_block = (int)cdiv((long)b * T, l.nblk);
    l.saved_bytes = sv.used();
    Carver sc(scratch);
    l.lossp = sc.take<float>((size_t)b * l.W * K);
    l.hit = sc.take<float>((size_t)b * l.W * K);
    l.dP = sc.take<float>((size_t)b * T * K * Henc);
    l.wt = sc.take<float>((size_t)K * Henc * Har);
    l.tn_bytes = std::max(gemm_tn_scratch_bytes(K * Henc, Har, (long)b * T), gemm_nt_scratch_bytes((long)b * T, Har, K * Henc));
    l.tn = sc.take<float>(l.tn_bytes / sizeof(float));
    l.tn_late = sc.take<float>(l.tn_bytes / sizeof(float));   // the weight gradient's slabs when it runs beside dc's K split (deferred form)
    l.counts = sc.take<int>((size_t)b * T);
    l.offsets = sc.take<int>((size_t)b * T + 1);
    l.entries = sc.take<int>((size_t)b * l.W * Nneg);
    l.vbuf = sc.take<float>((size_t)b * l.W * (NCE_POS + Nneg) * Henc);
    l.ds_buf = sc.take<float>((size_t)b * l.W * (NCE_ROWS + 1) * l.lw);
    l.scratch_bytes = sc.used();
    l.lds_fwd = align_up(sizeof(int) * (size_t)Nneg + sizeof(unsigned short) * (size_t)Nneg, 16);
    l.lds_bwd = sizeof(float) * (size_t)NCE_ROWS * l.lw + sizeof(int) * (size_t)l.lw;
    CPC_REQUIRE(l.lds_bwd <= 64 * 1024, "infonce: n_neg=%d needs %zu B of LDS (> 64 KiB)", Nneg, l.lds_bwd);
    l.lds_bwd_fused = l.lds_bwd + sizeof(float) * (size_t)(Henc / 128) * 16 * NCE_SROW;
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

static void nce_common(NceArgs &a, const NceLayout &l, const float *z, const int32_t *ext, const float *weights)
{
    (void)ext;                                  // the kernels walk the block-sorted copy (forward: nce_sort_negatives)
    a.z = z; a.ext = l.ext_sorted; a.perm = l.perm; a.weights = weights; a.logits = l.logits; a.lse = l.lse; a.lossp = l.lossp; a.hit = l.hit;
    a.b = l.b; a.T = l.T; a.W = l.W; a.K = l.K; a.Nneg = l.Nneg; a.lw = l.lw; a.inv_count = 1.f / ((float)l.b * l.W);
}

static int nce_sort_negatives(const NceLayout &l, const int32_t *ext, hipStream_t st)
{
    CPC_REQUIRE(l.Nneg <= 65535, "infonce: at most 65535 negatives (got %d)", l.Nneg);
    hipLaunchKernelGGL(nce_block_sort_kernel, dim3((unsigned)(l.b * l.W)), dim3(64), 0, st, ext, l.ext_sorted, l.perm, l.Nneg,
                       l.rows_per_block, l.nblk, l.b * l.T, coop_error_word());
    CPC_CHECK_LAUNCH("nce_block_sort_kernel");
    return CPC_OK;
}

static int nce_launch_fwd(NceArgs &a, const NceLayout &l, float *losses, float *acc, hipStream_t st)
{
    static const bool stamp = getenv("CPC_NCE_STAMP") != nullptr;
    static unsigned long long *stamps = nullptr;
    const long nw = (long)l.b * l.W;
    if (stamp && nw <= 65536) {
        if (stamps == nullptr) CPC_CHECK_HIP(hipMalloc(&stamps, 65536 * 8 * sizeof(unsigned long long)));
        a.stamps = stamps;
    }
    int status = CPC_OK;
    {
        static const bool no_dma = getenv("CPC_NCE_NO_DMA") != nullptr;           // A/B switch: the register-gather kernel
        const bool dma = !no_dma && (l.Henc == 256 || l.Henc == 512) && a.stamps == nullptr && l.Nneg % 8 == 0 && l.Nneg <= 256 && a.perm != nullptr;
        ProfScope prof(PROF_NCE_FWD, st, dma);      // (the stream kernel takes the timing events with its own dispatch: common.h)
        const hipEvent_t e0 = prof.start(), e1 = prof.stop();
        if (dma) {
            // persistent waves, two per SIMD: 8 workgroups of one wave per CU (LDS: two 8.25 KiB slots + two index lists each)
            const size_t lds = 2 * NCE_SLOT + 2 * NCE_LIST;
            static int n_cus = 0;
            if (n_cus == 0) {
                int dev = 0;
                CPC_CHECK_HIP(hipGetDevice(&dev));
                CPC_CHECK_HIP(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev));
            }
            const int n_bt = l.b * l.W;
            const unsigned grid = (unsigned)std::min(n_bt, 8 * n_cus);
            if (l.Henc == 256) {
                status = allow_lds(infonce_fwd_dma_kernel<256>, lds);
                if (status == CPC_OK && e0 != nullptr) hipExtLaunchKernelGGL(infonce_fwd_dma_kernel<256>, dim3(grid), dim3(64), lds, st, e0, e1, 0, a, n_bt);
                else if (status == CPC_OK) hipLaunchKernelGGL(infonce_fwd_dma_kernel<256>, dim3(grid), dim3(64), lds, st, a, n_bt);
            } else {
                status = allow_lds(infonce_fwd_dma_kernel<512>, lds);
                if (status == CPC_OK && e0 != nullptr) hipExtLaunchKernelGGL(infonce_fwd_dma_kernel<512>, dim3(grid), dim3(64), lds, st, e0, e1, 0, a, n_bt);
                else if (status == CPC_OK) hipLaunchKernelGGL(infonce_fwd_dma_kernel<512>, dim3(grid), dim3(64), lds, st, a, n_bt);
            }
        } else {
            NCE_DISPATCH(l.Henc, {
                status = allow_lds(infonce_fwd_kernel<HH>, l.lds_fwd);
                if (status == CPC_OK) hipLaunchKernelGGL(infonce_fwd_kernel<HH>, dim3((unsigned)(l.b * l.W)), dim3(64), l.lds_fwd, st, a);
            });
        }
        if (status != CPC_OK) prof.cancel();        // (nothing was launched: an attached scope's events were never recorded)
    }
    CPC_TRY(status);
    CPC_CHECK_LAUNCH("infonce_fwd_kernel");
    if (a.stamps != nullptr) {
        static std::vector<unsigned long long> h(65536 * 8);
        CPC_CHECK_HIP(hipStreamSynchronize(st));
        CPC_CHECK_HIP(hipMemcpy(h.data(), stamps, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double p = 0, f = 0, lp = 0, e = 0;
        unsigned long long r0 = ~0ull, r1 = 0;
        for (long i = 0; i < nw; ++i) {
            p += (double)(h[i * 8 + 1] - h[i * 8]); f += (double)(h[i * 8 + 2] - h[i * 8 + 1]); lp += (double)(h[i * 8 + 3] - h[i * 8 + 2]);
            e += (double)(h[i * 8 + 4] - h[i * 8 + 3]);
            r0 = std::min(r0, h[i * 8 + 5]); r1 = std::max(r1, h[i * 8 + 5]);
        }
        fprintf(stderr, "infonce_fwd stamps (cycles per wave): P loads %.0f, first tile rows %.0f, tile loop %.0f, epilogue %.0f; last - first wave exit %.1f us\n",
                p / nw, f / nw, lp / nw, e / nw, (double)(r1 - r0) * 0.01);
    }
    hipLaunchKernelGGL(infonce_reduce_kernel, dim3(2 * l.K), dim3(256), 0, st, l.lossp, l.hit, (long)l.b * l.W, l.K, a.inv_count, losses, acc);
    CPC_CHECK_LAUNCH("infonce_reduce_kernel");
    return CPC_OK;
}

// The reference lists depend on the indices only, the kernels before the gather do not need them: they are built
// on a stream of the library's own (one per device, created on first use), forked from and joined to the caller's
// stream with events -- no host synchronisation, and nothing of it outlives the call on the caller's stream.
struct NceSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    // deferred form of the backward (cpc_infonce_backward_deferred): `mid` = the fused backward kernel has run on the caller's
    // stream, `late` = dz and the predictor weight gradients are complete on the side stream; `pending` until cpc_infonce_join
    hipEvent_t mid = nullptr, late = nullptr;
    hipEvent_t tail_fork = nullptr, tail = nullptr;     // side_tail_*: work of another backward entry point finishing on this stream
    bool tail_pending = false;
    std::atomic<bool> pending{false};
    // what is left to launch of the pending backward (infonce_deferred_start): it is started by the NEXT backward entry point
    // called on the caller's stream (the context network's, right behind its first kernel) or, failing that, by the join
    bool marked = false, started = false, late_fused = false;     // marked: `mid` recorded
    NceLayout l;
    float *dz = nullptr, *dwpred = nullptr;
    const float *c = nullptr;
};

static int nce_side(NceSide **out, hipStream_t caller)
{
    static std::mutex mu;
    static std::map<int, NceSide> sides;
    int dev = 0;
    CPC_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    NceSide &sd = sides[dev];
    if (sd.stream == nullptr) {
        // DEFAULT priority, deliberately.  A lowest-priority stream looked right for work that runs beside the caller's, and costs
        // nothing in a single-process run -- but in a process that has also initialised RCCL every kernel of the step ran ~45 %
        // slower (7.7 against 5.3 ms per step with one rank; found by bisection, profiles/r03_dist_priority_bisect.txt).
        // ... and on a hardware queue of its own: tested against the caller's stream (stream_create_apart, rowops.hip)
        CPC_TRY(stream_create_apart(&caller, 1, &sd.stream));
        CPC_CHECK_HIP(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
        CPC_CHECK_HIP(hipEventCreateWithFlags(&sd.join, hipEventDisableTiming));
        CPC_CHECK_HIP(hipEventCreateWithFlags(&sd.mid, hipEventDisableTiming));
        CPC_CHECK_HIP(hipEventCreateWithFlags(&sd.late, hipEventDisableTiming));
        CPC_CHECK_HIP(hipEventCreateWithFlags(&sd.tail_fork, hipEventDisableTiming));
        CPC_CHECK_HIP(hipEventCreateWithFlags(&sd.tail, hipEventDisableTiming));
    }
    *out = &sd;
    return CPC_OK;
}

// dz of the criterion: CPC_NCE_ATOMIC in the environment selects the fp32-atomic form, the default stores every
// contribution once and sums per target row (see infonce_bwd_kernel)
static int nce_launch_gather(const NceLayout &l, float *dz, bool fused, hipStream_t st)
{
    const int rows = l.b * l.T;
    // (one workgroup per four rows.  Fewer, persistent workgroups -- so that only a few waves per CU stream beside the recurrent
    //  backward -- were measured again in round 4: 256 / 512 / 1024 workgroups cost CPC-large +1.2 / +0.35 / 0 ms per step and
    //  CPC-small 0 .. +0.15, as in round 3)
    const unsigned wgs = (unsigned)cdiv(rows, 4);
    NCE_DISPATCH(l.Henc, hipLaunchKernelGGL(infonce_dz_gather_kernel<HH>, dim3(wgs), dim3(256), 0, st, l.vbuf,
                                             l.offsets, l.entries, dz, l.b, l.T, l.W, l.K, l.Nneg, fused ? NCE_POS : l.K));
    CPC_CHECK_LAUNCH("infonce_dz_gather_kernel");
    return CPC_OK;
}

// `late` (deferred form): the sum of the contribution rows into dz is NOT launched; *late is left pointing at the side record
// (the reference lists are queued on its stream) and the caller launches nce_launch_gather there when it wants it to start
static int nce_launch_bwd(NceArgs &a, const NceLayout &l, float *dz, hipStream_t st, NceSide **late = nullptr)
{
    static const bool atomic_dz = getenv("CPC_NCE_ATOMIC") != nullptr;
    const long n = (long)l.b * l.W * l.Nneg;
    const int rows = l.b * l.T;
    a.dz = dz;
    a.vbuf = atomic_dz ? nullptr : l.vbuf;
    a.ds_buf = l.ds_buf;
    static const bool stamp = getenv("CPC_NCE_STAMP") != nullptr;
    if (stamp && (long)l.b * l.W <= 65536) {
        static unsigned long long *stamps = nullptr;
        if (stamps == nullptr) CPC_CHECK_HIP(hipMalloc(&stamps, 65536 * 8 * sizeof(unsigned long long)));
        a.stamps = stamps;
    }
    ProfScope prof(PROF_NCE_BWD, st);
    NceSide *side = nullptr;
    if (atomic_dz) {
        CPC_CHECK_HIP(hipMemsetAsync(dz, 0, sizeof(float) * (size_t)rows * l.Henc, st));
    } else {
        CPC_TRY(nce_side(&side, st));
        CPC_CHECK_HIP(hipEventRecord(side->fork, st));
        CPC_CHECK_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
        const unsigned blocks = (unsigned)std::min<long>(cdiv(n, 256), 4096);
        CPC_CHECK_HIP(hipMemsetAsync(l.counts, 0, sizeof(int) * (size_t)rows, side->stream));
        hipLaunchKernelGGL(nce_hist_kernel, dim3(blocks), dim3(256), 0, side->stream, a.ext, n, l.counts);
        hipLaunchKernelGGL(nce_scan_kernel, dim3(1), dim3(1024), 0, side->stream, l.counts, rows, l.offsets);
        CPC_CHECK_HIP(hipMemsetAsync(l.counts, 0, sizeof(int) * (size_t)rows, side->stream));
        hipLaunchKernelGGL(nce_fill_kernel, dim3(blocks), dim3(256), 0, side->stream, a.ext, n, l.offsets, l.counts, l.entries);
        hipLaunchKernelGGL(nce_sort_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, side->stream, l.offsets, l.entries, rows);
        CPC_CHECK_LAUNCH("infonce reference lists");
        CPC_CHECK_HIP(hipEventRecord(side->join, side->stream));
    }
    int status = CPC_OK;
    static const bool no_fused = getenv("CPC_NCE_NO_FUSED_BWD") != nullptr;       // A/B switch: the two one-wave kernels
    const bool fused = !atomic_dz && !no_fused && (l.Henc == 256 || l.Henc == 512) && a.perm != nullptr && l.lw <= 320 &&
                       l.Nneg % 16 == 0 && (size_t)l.K * (l.Nneg + 1) <= (size_t)(l.Henc / 128) * 16 * NCE_SROW;
    if (fused) {
        const unsigned grid = (unsigned)(l.b * a.p_rows);
#define NCE_FUSED(HH, NKK)                                                                                                  \
    {                                                                                                                       \
        status = allow_lds(infonce_bwd_fused_kernel<HH, NKK>, l.lds_bwd_fused);                                             \
        if (status == CPC_OK) hipLaunchKernelGGL((infonce_bwd_fused_kernel<HH, NKK>), dim3(grid), dim3(HH / 2), l.lds_bwd_fused, st, a); \
    }
        const int nkk = (l.K + 3) / 4;
        if (l.Henc == 256) {
            if (nkk == 1) NCE_FUSED(256, 1) else if (nkk == 2) NCE_FUSED(256, 2) else if (nkk == 3) NCE_FUSED(256, 3) else NCE_FUSED(256, 4)
        } else {
            if (nkk == 1) NCE_FUSED(512, 1) else if (nkk == 2) NCE_FUSED(512, 2) else if (nkk == 3) NCE_FUSED(512, 3) else NCE_FUSED(512, 4)
        }
#undef NCE_FUSED
    } else {
        NCE_DISPATCH(l.Henc, {
            status = allow_lds(infonce_bwd_kernel<HH>, l.lds_bwd);
            if (status == CPC_OK) hipLaunchKernelGGL(infonce_bwd_kernel<HH>, dim3((unsigned)(l.b * a.p_rows)), dim3(64), l.lds_bwd, st, a);
        });
    }
    CPC_TRY(status);
    CPC_CHECK_LAUNCH("infonce_bwd_kernel");
    if (fused && a.stamps != nullptr) {
        const long nw = (long)l.b * l.W;
        std::vector<unsigned long long> h(nw * 8);
        CPC_CHECK_HIP(hipStreamSynchronize(st));
        CPC_CHECK_HIP(hipMemcpy(h.data(), a.stamps, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double p0 = 0, p1 = 0, p2 = 0;
        unsigned long long r0 = ~0ull, r1 = 0;
        for (long i = 0; i < nw; ++i) {
            p0 += (double)(h[i * 8 + 1] - h[i * 8]); p1 += (double)(h[i * 8 + 2] - h[i * 8 + 1]); p2 += (double)(h[i * 8 + 3] - h[i * 8 + 2]);
            r0 = std::min(r0, h[i * 8 + 4]); r1 = std::max(r1, h[i * 8 + 4]);
        }
        fprintf(stderr, "infonce_bwd_fused stamps (cycles per workgroup): dS %.0f, dP %.0f, contribution rows %.0f; last - first exit %.1f us\n",
                p0 / nw, p1 / nw, p2 / nw, (double)(r1 - r0) * 0.01);
    }
    if (!atomic_dz) {
        if (!fused) {
            NCE_DISPATCH(l.Henc, hipLaunchKernelGGL(infonce_dz_store_kernel<HH>, dim3((unsigned)(l.b * l.W)), dim3(64), l.lds_bwd, st, a));
            CPC_CHECK_LAUNCH("infonce_dz_store_kernel");
        }
        if (late != nullptr) {                  // the caller launches the sum on the side stream (nce_launch_gather), later
            *late = side;
            side->late_fused = fused;
            return CPC_OK;
        }
        CPC_CHECK_HIP(hipStreamWaitEvent(st, side->join, 0));
        CPC_TRY(nce_launch_gather(l, dz, fused, st));
    }
    return CPC_OK;
}

// c_frames: frames per window of c (and of dc): T (criterion.py:296 slices c[:, :W] itself), or W = T - K when the caller hands over
// that slice (cpcStep: the context network then only runs the W steps whose output is used)
static int infonce_forward(const float *c, const float *z, const float *wpred, const int32_t *ext, const float *weights,
                           float *losses, float *acc, void *saved, void *scratch, int b, int T, int K, int Har, int Henc,
                           int Nneg, hipStream_t st, int c_frames = 0)
{
    NceLayout l;
    CPC_TRY(nce_layout(l, b, T, K, Har, Henc, Nneg, saved, scratch));
    if (c_frames != 0) {
        CPC_REQUIRE(c_frames == T || c_frames == l.W, "infonce: the context holds %d frames per window (expected %d or %d)", c_frames, T, l.W);
        l.Pr = c_frames;
    }
    RowMap none{};
    // all K predictors in one GEMM: P[(b,t)][k*Henc + e] = sum_a c[b,t,a] * W_k[e][a]     (criterion.py:163)
    CPC_TRY(gemm_nt(c, Har, wpred, Har, l.P, (long)K * Henc, nullptr, (long)b * l.Pr, K * Henc, Har, none, st));
    CPC_TRY(nce_sort_negatives(l, ext, st));
    NceArgs a{};
    nce_common(a, l, z, ext, weights);
    for (int k = 0; k < K; ++k) a.Pk[k] = l.P + (size_t)k * Henc;
    a.p_stride = (long)K * Henc; a.p_rows = l.Pr;
    a.p_packed = (size_t)b * T * K * Henc * sizeof(float) < (1ull << 32) ? 1 : 0;      // (32-bit byte offsets in the streaming kernel)
    return nce_launch_fwd(a, l, losses, acc, st);
}

static int infonce_backward(const float *c, const float *z, const float *wpred, const int32_t *ext, const float *weights,
                            const float *dlosses, void *saved, void *scratch, float *dc, float *dz, float *dwpred, int b, int T,
                            int K, int Har, int Henc, int Nneg, hipStream_t st, bool defer, int c_frames = 0)
{
    NceLayout l;
    CPC_TRY(nce_layout(l, b, T, K, Har, Henc, Nneg, saved, scratch));
    if (c_frames != 0) {
        CPC_REQUIRE(c_frames == T || c_frames == l.W, "infonce: the context holds %d frames per window (expected %d or %d)", c_frames, T, l.W);
        l.Pr = c_frames;
    }
    NceArgs a{};
    nce_common(a, l, z, ext, weights);
    for (int k = 0; k < K; ++k) { a.Pk[k] = l.P + (size_t)k * Henc; a.dPk[k] = l.dP + (size_t)k * Henc; }
    a.p_stride = (long)K * Henc; a.p_rows = l.Pr;
    a.dloss = dlosses;
    // Deferred form: only dc -- what the context network's backward waits for -- is produced on `st`.  dz (a memory-bound sum
    // over ~1 GB of contribution rows) and the predictors' weight gradients run on the side stream, beside whatever the caller
    // queues on `st` next (the recurrent backward: latency-bound, the chip mostly idle), until cpc_infonce_join.
    NceSide *late = nullptr;
    CPC_TRY(nce_launch_bwd(a, l, dz, st, defer ? &late : nullptr));
    // dc = dP . W  (rows t >= W of dP are zero)
    CPC_TRY(transpose2d(wpred, l.wt, K * Henc, Har, st));                        // [Har][K*Henc]
    RowMap none{};
    none.splitk_scratch = l.tn; none.splitk_bytes = l.tn_bytes;                  // few tiles, K = 12 H: ordered K split
    CPC_TRY(gemm_nt(l.dP, (long)K * Henc, l.wt, (long)K * Henc, dc, Har, nullptr, (long)b * l.Pr, Har, K * Henc, none, st));
    // dW_k[e][a] = sum_{b,t} dP[(b,t)][k*Henc + e] * c[b,t,a]
    if (late != nullptr) {
        // nothing is queued on the side stream yet: queued here, beside the product above, it took the chip from it (85 -> 280 us)
        // and then stood in the way of the small kernels in front of the recurrent backward instead of filling its idle time
        late->l = l; late->dz = dz; late->dwpred = dwpred; late->c = c;
        late->started = false; late->marked = false;
        late->pending.store(true);
    } else {
        CPC_TRY(gemm_tn(l.dP, (long)K * Henc, c, Har, dwpred, Har, K * Henc, Har, (long)b * l.Pr, l.tn, l.tn_bytes, 0, 0, st));
    }
    return CPC_OK;
}

// Queue the rest of a pending deferred backward on the side stream (no-op when nothing is pending or it has been started),
// ordered behind what `st` held at infonce_deferred_mark (or holds now, if that was not called).  The recurrent backward entry
// points mark in front of their first kernel and start behind it: both run at once, and the recurrent kernel -- which needs
// every workgroup resident -- is dispatched first; the sum and the weight gradient take what the chip has left.
int infonce_deferred_mark(hipStream_t st)
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, st));
    if (!side->pending.load() || side->started || side->marked) return CPC_OK;
    CPC_CHECK_HIP(hipEventRecord(side->mid, st));
    side->marked = true;
    return CPC_OK;
}

int infonce_deferred_start(hipStream_t st)
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, st));
    if (!side->pending.load() || side->started) return CPC_OK;
    const NceLayout &l = side->l;
    CPC_TRY(infonce_deferred_mark(st));
    CPC_CHECK_HIP(hipStreamWaitEvent(side->stream, side->mid, 0));
    CPC_TRY(nce_launch_gather(l, side->dz, side->late_fused, side->stream));
    CPC_TRY(gemm_tn(l.dP, (long)l.K * l.Henc, side->c, l.Har, side->dwpred, l.Har, l.K * l.Henc, l.Har, (long)l.b * l.Pr, l.tn_late, l.tn_bytes, 0,
                    0, side->stream));
    CPC_CHECK_HIP(hipEventRecord(side->late, side->stream));
    side->started = true;
    return CPC_OK;
}

// ---- "tail" work of a backward entry point on the library's side stream: ordered behind what `st` holds now (and behind whatever
// the side stream already has queued); whoever reads its results waits at side_tail_join
int side_tail_begin(hipStream_t st, hipStream_t *side_stream)
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, st));
    CPC_CHECK_HIP(hipEventRecord(side->tail_fork, st));
    CPC_CHECK_HIP(hipStreamWaitEvent(side->stream, side->tail_fork, 0));
    *side_stream = side->stream;
    return CPC_OK;
}
int side_tail_end()
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, nullptr));
    CPC_CHECK_HIP(hipEventRecord(side->tail, side->stream));
    side->tail_pending = true;
    return CPC_OK;
}
int side_tail_join(hipStream_t st)
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, st));
    if (side->tail_pending) {
        ProfScope held(PROF_SIDE_WAIT, st);          // (bench.py: how long `st` stands still here)
        CPC_CHECK_HIP(hipStreamWaitEvent(st, side->tail, 0));
        side->tail_pending = false;
    }
    return CPC_OK;
}

int side_stream_peek(hipStream_t caller, hipStream_t *out)
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, caller));
    *out = side->stream;
    return CPC_OK;
}

// the same wait WITHOUT taking the tail off the books: a helper stream (the data-parallel exchange's) orders itself behind the tail
// while the caller's stream goes on; whoever reuses the tail's buffers still joins with side_tail_join
int side_tail_wait(hipStream_t st)
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, st));
    if (side->tail_pending) CPC_CHECK_HIP(hipStreamWaitEvent(st, side->tail, 0));
    return CPC_OK;
}

// every stream that is going to touch dz / dwpred of a deferred backward waits here (no-op when nothing is pending)
static int infonce_join(hipStream_t st)
{
    NceSide *side = nullptr;
    CPC_TRY(nce_side(&side, st));
    if (side->pending.load()) {
        CPC_TRY(infonce_deferred_start(st));
        {
            ProfScope held(PROF_SIDE_WAIT, st);
            CPC_CHECK_HIP(hipStreamWaitEvent(st, side->late, 0));
        }
        side->pending.store(false);
    }
    return CPC_OK;
}

// ---- predictions supplied by the caller (non-linear predictor modules): pred[k] is [b][W][Henc]
static int infonce_forward_pred(const float *const *pred, const float *z, const int32_t *ext, const float *weights, float *losses,
                                float *acc, void *saved, void *scratch, int b, int T, int K, int Henc, int Nneg, hipStream_t st)
{
    NceLayout l;
    CPC_TRY(nce_layout(l, b, T, K, Henc, Henc, Nneg, saved, scratch));
    CPC_TRY(nce_sort_negatives(l, ext, st));
    NceArgs a{};
    nce_common(a, l, z, ext, weights);
    for (int k = 0; k < K; ++k) a.Pk[k] = pred[k];
    a.p_stride = Henc; a.p_rows = l.W;
    return nce_launch_fwd(a, l, losses, acc, st);
}

static int infonce_backward_pred(const float *const *pred, const float *z, const int32_t *ext, const float *weights,
                                 const float *dlosses, void *saved, void *scratch, float *const *dpred, float *dz, int b, int T, int K,
                                 int Henc, int Nneg, hipStream_t st)
{
    NceLayout l;
    CPC_TRY(nce_layout(l, b, T, K, Henc, Henc, Nneg, saved, scratch));
    NceArgs a{};
    nce_common(a, l, z, ext, weights);
    for (int k = 0; k < K; ++k) { a.Pk[k] = pred[k]; a.dPk[k] = dpred[k]; }
    a.p_stride = Henc; a.p_rows = l.W;
    a.dloss = dlosses;
    return nce_launch_bwd(a, l, dz, st);
}

}  // namespace cpc

extern "C" int cpc_negidx_expand(const uint32_t *raw, int32_t *ext_idx, int batch, int seq_len, int window, int n_neg,
                                 cpc_stream_t stream)
{
    CPC_REQUIRE(raw != nullptr && ext_idx != nullptr && batch >= 1 && seq_len >= 2 && window >= 1 && window <= seq_len && n_neg >= 1 &&
                    (long long)batch * seq_len <= 2147483647LL,
                "cpc_negidx_expand: bad arguments (batch=%d seq_len=%d window=%d n_neg=%d)", batch, seq_len, window, n_neg);
    const long n = (long)batch * n_neg * window;
    hipLaunchKernelGGL(cpc::negidx_expand_kernel, dim3((unsigned)std::min<long>(cpc::cdiv(n, 256), 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), raw, ext_idx, batch, seq_len, window, n_neg);
    CPC_CHECK_LAUNCH("negidx_expand_kernel");
    return CPC_OK;
}

extern "C" size_t cpc_infonce_saved_bytes(int b, int t, int k, int dim_ar, int dim_enc, int n_neg)
{
    cpc::NceLayout l;
    if (cpc::nce_layout(l, b, t, k, dim_ar, dim_enc, n_neg, nullptr, nullptr) != CPC_OK) return 0;
    return l.saved_bytes;
}

extern "C" size_t cpc_infonce_logits_offset(int b, int t, int k, int dim_ar, int dim_enc, int n_neg)
{
    cpc::NceLayout l;
    char *const base = reinterpret_cast<char *>(4096);   // a fake base, never dereferenced: the layout only does arithmetic
    if (cpc::nce_layout(l, b, t, k, dim_ar, dim_enc, n_neg, base, nullptr) != CPC_OK) return (size_t)-1;
    return (size_t)(reinterpret_cast<char *>(l.logits) - base);
}

extern "C" size_t cpc_infonce_perm_offset(int b, int t, int k, int dim_ar, int dim_enc, int n_neg)
{
    cpc::NceLayout l;
    char *const base = reinterpret_cast<char *>(4096);
    if (cpc::nce_layout(l, b, t, k, dim_ar, dim_enc, n_neg, base, nullptr) != CPC_OK) return (size_t)-1;
    return (size_t)(reinterpret_cast<char *>(l.perm) - base);
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
                                 n_neg, static_cast<hipStream_t>(stream), false);
}

extern "C" int cpc_infonce_backward_deferred(const float *c, const float *z, const float *wpred, const int32_t *ext_idx,
                                             const float *weights, const float *dlosses, void *saved, void *scratch, float *dc,
                                             float *dz, float *dwpred, int b, int t, int k, int dim_ar, int dim_enc, int n_neg,
                                             cpc_stream_t stream)
{
    return cpc::infonce_backward(c, z, wpred, ext_idx, weights, dlosses, saved, scratch, dc, dz, dwpred, b, t, k, dim_ar, dim_enc,
                                 n_neg, static_cast<hipStream_t>(stream), true);
}

extern "C" int cpc_infonce_forward_cw(const float *c, const float *z, const float *wpred, const int32_t *ext_idx, const float *weights,
                                      float *losses, float *acc, void *saved, void *scratch, int b, int t, int k, int dim_ar,
                                      int dim_enc, int n_neg, cpc_stream_t stream)
{
    return cpc::infonce_forward(c, z, wpred, ext_idx, weights, losses, acc, saved, scratch, b, t, k, dim_ar, dim_enc, n_neg,
                                static_cast<hipStream_t>(stream), t - k);
}

extern "C" int cpc_infonce_backward_cw(const float *c, const float *z, const float *wpred, const int32_t *ext_idx, const float *weights,
                                       const float *dlosses, void *saved, void *scratch, float *dc, float *dz, float *dwpred, int b,
                                       int t, int k, int dim_ar, int dim_enc, int n_neg, int deferred, cpc_stream_t stream)
{
    return cpc::infonce_backward(c, z, wpred, ext_idx, weights, dlosses, saved, scratch, dc, dz, dwpred, b, t, k, dim_ar, dim_enc,
                                 n_neg, static_cast<hipStream_t>(stream), deferred != 0, t - k);
}

extern "C" int cpc_infonce_join(cpc_stream_t stream) { return cpc::infonce_join(static_cast<hipStream_t>(stream)); }
extern "C" int cpc_side_stream(cpc_stream_t caller, cpc_stream_t *out)
{
    CPC_REQUIRE(out != nullptr, "cpc_side_stream: null output");
    hipStream_t st = nullptr;
    CPC_TRY(cpc::side_stream_peek(static_cast<hipStream_t>(caller), &st));
    *out = st;
    return CPC_OK;
}

extern "C" int cpc_infonce_forward_pred(const float *const *pred, const float *z, const int32_t *ext_idx, const float *weights,
                                        float *losses, float *acc, void *saved, void *scratch, int b, int t, int k, int dim_enc,
                                        int n_neg, cpc_stream_t stream)
{
    return cpc::infonce_forward_pred(pred, z, ext_idx, weights, losses, acc, saved, scratch, b, t, k, dim_enc, n_neg,
                                     static_cast<hipStream_t>(stream));
}

extern "C" int cpc_infonce_backward_pred(const float *const *pred, const float *z, const int32_t *ext_idx, const float *weights,
                                         const float *dlosses, void *saved, void *scratch, float *const *dpred, float *dz, int b,
                                         int t, int k, int dim_enc, int n_neg, cpc_stream_t stream)
{
    return cpc::infonce_backward_pred(pred, z, ext_idx, weights, dlosses, saved, scratch, dpred, dz, b, t, k, dim_enc, n_neg,
                                      static_cast<hipStream_t>(stream));
}
