// fp32 GEMMs for gfx950.  Two kernel families compute the same products:
//   * gemm_*_x6_kernel (default): f32 operands are split exactly into three bf16 terms while they are staged
//     into LDS and multiplied on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate) as six
//     partial products accumulated in f32 -- f32-GEMM accuracy (see the comment at gemm_nt_x6_kernel);
//   * gemm_*_kernel: the f32 MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD); unaligned shapes, and the
//     yardstick the split kernels' error is tested against (cpc_gemm_set_mode(1)).
//
//   gemm_nt : C[map(m)][n] = sum_k A[m*lda + k] * B[n*ldb + k] (+ bias[n])
//             A rows may OVERLAP (lda < K): that is how the strided Conv1d layers run as
//             implicit GEMMs over channel-last activations (see encoder.hip).
//   gemm_tn : C[i][j] = sum_r A[r*lda + i] * B[r*ldb + j]   (weight gradients; split over r,
//             partial slabs reduced by a second kernel -> bitwise reproducible, no atomics)
//
// Block tile 128x128, 256 threads = 4 waves (2x2), wave tile 64x64 = 2x2 MFMA tiles of 32x32,
// K step 32.  Operands are staged global -> registers -> LDS (prefetch of the next K tile is
// in flight while the current one is multiplied); LDS rows are padded to 36 floats so the
// ds_read_b128 fragment reads are bank-conflict free.
#include "common.h"
#include <atomic>

#include <algorithm>
#include <type_traits>

namespace cpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// 0: bf16x6 split kernels (default); 1: native f32-MFMA kernels (the accuracy yardstick; also CPC_GEMM_NATIVE_F32=1)
// Process-wide, deliberately: the one setting the library keeps between calls.  (Round 4 made it thread-local for a day -- and the
// backward pass stopped seeing it: autograd runs backward on its own worker thread, so a mode selected on the thread that calls
// forward never reached the weight-gradient products.  A mode is a property of a RUN -- the accuracy yardstick of the tests, the
// labelled three-product entry of bench.py -- not of a call; it is an atomic so that a concurrent reader sees one value or the other.)
static std::atomic<int> g_gemm_mode{getenv("CPC_GEMM_NATIVE_F32") != nullptr ? 1 : 0};

constexpr int BN = 128, BK = 32;
constexpr int BM = 128;          // TN kernel tile; the NT kernel derives its own from MI
constexpr int LDS_NT = BK + 4;    // 36 floats per LDS row (NT: k contiguous)
constexpr int LDS_TN = BM + 4;    // 132 floats per LDS row (TN: i/j contiguous)

struct GemmNTArgs {
    const float *A; long lda;
    const float *B; long ldb;
    float *C; long ldc;
    const float *bias;
    long M; int N; int K;
    RowMap map;
    int aligned;   // K%4==0, lda%4==0, ldb%4==0, bases 16-B aligned
    int kchunk;    // K range per blockIdx.y (multiple of BK); gridDim.y > 1: partial products are atomically added
    // segmented rows (x6 kernels): M = segments of seg_rows rows of which the first seg_valid are computed, tiles
    // never straddle a segment (the junk virtual rows of a sample are then never multiplied).  seg_rows = 0: off
    int seg_rows, seg_valid;
    float *slabs;      // K split with ordered reduction: partial product of blockIdx.y goes to slabs + blockIdx.y * M * N
    int xcd_remap;     // split kernels: column panels of a row tile on one XCD (row tile count % 8 == 0, several panels)
    int vec_out;       // x6 kernels: dense output in 16-byte pieces through LDS (no row map, no atomics; N, ldc % 4 == 0)
    int dbg;           // -DX6_PROBE builds only: 1 no epilogue, 2 no split arithmetic, 4 no MFMAs, 8 no loads after the first K step
};

// acc[i][j][e] is C[m][n], m = m0 + wm*32*MI + i*32 + (e&3) + 8*(e>>2) + 4h, n = n0 + wn*64 + j*32 + r32
// (the C/D layout of every 32x32 MFMA on gfx950, f32- and bf16-input alike)
template <int MI, int NJ>
__device__ __forceinline__ void nt_epilogue(const GemmNTArgs &p, f32x16 (&acc)[MI][NJ], long m0, long m_end, int n0, int wm, int wn, int r32, int h)
{
    float bias_v[NJ];
    int ncol[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        ncol[j] = n0 + wn * 32 * NJ + j * 32 + r32;
        bias_v[j] = (p.bias != nullptr && ncol[j] < p.N && blockIdx.y == 0) ? p.bias[ncol[j]] : 0.f;
    }
    // column block of each 32-wide MFMA tile (backward-data phases side by side): 32 | col_rows
    int jrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) jrow[j] = (p.map.enabled && p.map.col_rows > 0) ? (n0 + wn * 32 * NJ + j * 32) / p.map.col_rows : 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long m = m0 + wm * 32 * MI + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m >= m_end) continue;
            long crow = m;
            long l = 0;
            if (p.map.enabled) {
                const long g = m / p.map.rv;
                const int t = (int)(m - g * p.map.rv);
                l = (long)t * p.map.out_stride + p.map.out_off;
                crow = g * p.map.rows_out + l;
            }
            float *crowp = p.slabs != nullptr ? p.slabs + ((long)blockIdx.y * p.M + crow) * p.N : p.C + crow * p.ldc;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (p.map.enabled && (l + jrow[j] < 0 || l + jrow[j] >= p.map.l_max)) continue;
                if (ncol[j] < p.N) {
                    if (gridDim.y > 1 && p.slabs == nullptr) atomicAdd(&crowp[ncol[j]], acc[i][j][e] + bias_v[j]);
                    else crowp[ncol[j]] = acc[i][j][e] + bias_v[j];
                }
            }
        }
    }
}

// Dense output tiles through LDS: the wave's 32-row blocks are staged one at a time in its own corner of the (by now idle)
// operand tiles and leave as 16-byte stores of whole row segments -- 8 or 16 store instructions per 32-row block instead of 32
// four-byte ones per MFMA tile.  With K = 256 the four-byte form was HALF of the NT kernel's time (predictor product 108 us, 59
// without its epilogue; the FFN's first product 126 / 41: profiles/r04_x6_ladder.txt).
// The wave's tile starts at (row0, col0); rows < row_end and columns < ncols (% 4 == 0) are stored; NT: non-temporal stores.
// one 32-row block of the wave's tile: registers -> the wave's staging rows -> memory
template <int NJ>
__device__ __forceinline__ void store_block_staged(const f32x16 (&acc)[NJ], const float (&bias_v)[NJ], float *stg, int lane, float *outb, long ldo,
                                                   long row0, long row_end, int col0, int ncols, bool nt, const RowMap *epi)
{
    constexpr int W = 32 * NJ;              // columns of a wave
    constexpr int LDW = W + 4;              // floats per staged row
    constexpr int LPR = W / 4;              // lanes per row of 16-byte pieces
    constexpr int RPI = 64 / LPR;           // rows per store instruction
    const int r32 = lane & 31, h = lane >> 5;
    const int c4 = (lane % LPR) * 4, rr = lane / LPR;
    const int col = col0 + c4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) stg[((e & 3) + 8 * (e >> 2) + 4 * h) * LDW + j * 32 + r32] = acc[j][e] + bias_v[j];
    if (epi == nullptr) {
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + rr;
            const long m = row0 + row;
            const float4 v = *reinterpret_cast<const float4 *>(stg + row * LDW + c4);
            if (m < row_end && col < ncols) {
                if (nt) __builtin_nontemporal_store(__builtin_bit_cast(f32x4_t, v), reinterpret_cast<f32x4_t *>(outb + m * ldo + col));
                else *reinterpret_cast<float4 *>(outb + m * ldo + col) = v;
            }
        }
        return;
    }
    // with elementwise work (a real loop: nothing here indexes the accumulators)
#pragma unroll 1
    for (int it = 0; it < 32 / RPI; ++it) {
        const int row = it * RPI + rr;
        const long m = row0 + row;
        float4 v = *reinterpret_cast<const float4 *>(stg + row * LDW + c4);
        if (m < row_end && col < ncols) {
            if (epi->epi == EPI_RELU_DROPOUT) {
                const uint64_t idx = (uint64_t)(m * ldo + col);
                v.x = v.x > 0.f ? v.x * drop_mul(epi->epi_seed, idx, epi->epi_thresh, epi->epi_scale) : 0.f;
                v.y = v.y > 0.f ? v.y * drop_mul(epi->epi_seed, idx + 1, epi->epi_thresh, epi->epi_scale) : 0.f;
                v.z = v.z > 0.f ? v.z * drop_mul(epi->epi_seed, idx + 2, epi->epi_thresh, epi->epi_scale) : 0.f;
                v.w = v.w > 0.f ? v.w * drop_mul(epi->epi_seed, idx + 3, epi->epi_thresh, epi->epi_scale) : 0.f;
            } else {
                const float4 g = *reinterpret_cast<const float4 *>(epi->epi_gate + m * ldo + col);
                v.x = g.x > 0.f ? v.x * epi->epi_scale : 0.f;
                v.y = g.y > 0.f ? v.y * epi->epi_scale : 0.f;
                v.z = g.z > 0.f ? v.z * epi->epi_scale : 0.f;
                v.w = g.w > 0.f ? v.w * epi->epi_scale : 0.f;
            }
            if (nt) __builtin_nontemporal_store(__builtin_bit_cast(f32x4_t, v), reinterpret_cast<f32x4_t *>(outb + m * ldo + col));
            else *reinterpret_cast<float4 *>(outb + m * ldo + col) = v;
        }
    }
}

template <int MI, int NJ>
__device__ __forceinline__ void store_tile_staged(f32x16 (&acc)[MI][NJ], char *lds, int wave, int lane, float *outb, long ldo, long row0, long row_end,
                                                  int col0, int ncols, const float *bias, bool nt, const RowMap *epi = nullptr)
{
    static_assert(MI == 1 || MI == 2, "row blocks are spelled out so that the accumulators are never indexed by a loop variable");
    float *const stg = reinterpret_cast<float *>(lds) + wave * 32 * (32 * NJ + 4);
    float bias_v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ncol = col0 + j * 32 + (lane & 31);
        bias_v[j] = (bias != nullptr && ncol < ncols) ? bias[ncol] : 0.f;
    }
    store_block_staged<NJ>(acc[0], bias_v, stg, lane, outb, ldo, row0, row_end, col0, ncols, nt, epi);
    if constexpr (MI == 2) store_block_staged<NJ>(acc[1], bias_v, stg, lane, outb, ldo, row0 + 32, row_end, col0, ncols, nt, epi);
}

template <int MI, int NJ>
__device__ __forceinline__ void nt_epilogue_staged(const GemmNTArgs &p, f32x16 (&acc)[MI][NJ], char *lds, long m0, long m_end, int n0, int wave,
                                                   int wm, int wn, int lane)
{
    float *const outb = p.slabs != nullptr ? p.slabs + (long)blockIdx.y * p.M * p.N : p.C;
    store_tile_staged<MI, NJ>(acc, lds, wave, lane, outb, p.slabs != nullptr ? p.N : p.ldc, m0 + wm * 32 * MI, m_end, n0 + wn * 32 * NJ, p.N,
                              blockIdx.y == 0 ? p.bias : nullptr, p.vec_out == 2, p.map.epi != EPI_NONE ? &p.map : nullptr);
}

// 4 consecutive elements k..k+3 of row `row` (k < K or zero).  Rows beyond row_max are CLAMPED, not zeroed:
// for the output dimensions (M, N) such rows only feed outputs the epilogue discards.
// ALIGNED (extent % 4 == 0, 16-byte aligned rows) and !KTAIL: one unconditional 16-byte load -- no branch and
// no select, so a K step's 8 loads issue back to back and are waited for only where the LDS write needs them.
template <bool ALIGNED, bool KTAIL>
__device__ __forceinline__ float4 ld4(const float *base, long row, long row_max, long ld, int k, int K)
{
    const long rc = row < row_max ? row : row_max - 1;
    const float *r = base + rc * ld;
    if (ALIGNED) {
        if (!KTAIL) return *reinterpret_cast<const float4 *>(r + k);
        const int kc = k < K ? k : K - 4;
        float4 v = *reinterpret_cast<const float4 *>(r + kc);
        if (k >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) v.x = r[k];
        if (k + 1 < K) v.y = r[k + 1];
        if (k + 2 < K) v.z = r[k + 2];
        if (k + 3 < K) v.w = r[k + 3];
        return v;
    }
}

// MI = 32-row MFMA tiles per wave along M: block tile (64*MI) x 128.  MI = 2 is the default 128x128 tile;
// MI = 1 (64x128) is picked when the 128-row grid would leave CUs with a single workgroup.
template <bool ALIGNED, int MI> __global__ __launch_bounds__(256, 3) void gemm_nt_kernel(GemmNTArgs p)
{
    constexpr int BM = 64 * MI;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_NT];
    float *As = lds;                  // [BM][LDS_NT]
    float *Bs = lds + BM * LDS_NT;    // [BN][LDS_NT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    // 1-D grid, N tile fastest: the blocks that share an A panel are dispatched back to back
    const int tiles_n = (p.N + BN - 1) / BN;
    const long m0 = (long)(blockIdx.x / tiles_n) * BM;
    const int n0 = (int)(blockIdx.x % tiles_n) * BN;

    // loader: 2*MI float4 of A and 4 of B per thread; slot = tid + 256*q -> row = slot/8, c4 = slot%8
    const int lrow = tid >> 3;        // + 32*q
    const int lc4 = (tid & 7) * 4;
    constexpr int QA = 2 * MI;
    float4 ra[QA], rb[4];

    const int kbeg = blockIdx.y * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk_full = (kend - kbeg) / BK;    // K tiles that need no tail handling
    auto load_tiles = [&](int kt) {
        const int k0 = kbeg + kt * BK;
        if (ALIGNED && kt < nk_full) {
#pragma unroll
            for (int q = 0; q < QA; ++q) ra[q] = ld4<ALIGNED, false>(p.A, m0 + lrow + 32 * q, p.M, p.lda, k0 + lc4, kend);
#pragma unroll
            for (int q = 0; q < 4; ++q) rb[q] = ld4<ALIGNED, false>(p.B, n0 + lrow + 32 * q, p.N, p.ldb, k0 + lc4, kend);
        } else {
#pragma unroll
            for (int q = 0; q < QA; ++q) ra[q] = ld4<ALIGNED, true>(p.A, m0 + lrow + 32 * q, p.M, p.lda, k0 + lc4, kend);
#pragma unroll
            for (int q = 0; q < 4; ++q) rb[q] = ld4<ALIGNED, true>(p.B, n0 + lrow + 32 * q, p.N, p.ldb, k0 + lc4, kend);
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int q = 0; q < QA; ++q) *reinterpret_cast<float4 *>(&As[(lrow + 32 * q) * LDS_NT + lc4]) = ra[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(&Bs[(lrow + 32 * q) * LDS_NT + lc4]) = rb[q];
        __syncthreads();
        if (kt + 1 < nk) load_tiles(kt + 1);

        // fragments of slice kk+1 are read while slice kk is multiplied (two register sets).
        // lane (r32, h) takes k = kk*8 + 4h + {0,1,2,3}; MFMA step e pairs k-slot (h, e) of A with the same k of B,
        // so any k permutation shared by both operands is valid.
        float4 fa[2][MI], fb[2][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[0][i] = *reinterpret_cast<const float4 *>(&As[(wm * 32 * MI + i * 32 + r32) * LDS_NT + h * 4]);
#pragma unroll
        for (int i = 0; i < 2; ++i) fb[0][i] = *reinterpret_cast<const float4 *>(&Bs[(wn * 64 + i * 32 + r32) * LDS_NT + h * 4]);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa[nxt][i] = *reinterpret_cast<const float4 *>(&As[(wm * 32 * MI + i * 32 + r32) * LDS_NT + (kk + 1) * 8 + h * 4]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fb[nxt][i] = *reinterpret_cast<const float4 *>(&Bs[(wn * 64 + i * 32 + r32) * LDS_NT + (kk + 1) * 8 + h * 4]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fb[cur][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fb[cur][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, fb[cur][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, fb[cur][j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    nt_epilogue<MI, 2>(p, acc, m0, p.M, n0, wm, wn, r32, h);
}

// ------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe (16x the f32 MFMA rate on gfx950): every f32 operand is split
// EXACTLY into three bf16 terms x = x0 + x1 + x2 (round-to-nearest residuals: |x - x0 - x1 - x2| <= 2^-27 |x|)
// while it is staged into LDS, and a.b is accumulated in f32 from the six products whose weight is >= 2^-18:
// a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 (the dropped a1b2, a2b1, a2b2 are <= 2^-26 |a||b|, below the rounding of
// one f32 multiply).  bf16 x bf16 is exact in f32, so the result carries f32-GEMM accuracy (tests compare both
// kernels with an fp64 product), at 6 bf16 MFMAs per 16 k instead of 8 f32 MFMAs of 1/16 the rate.
//
// LDS image per operand: three planes [rows][32 k] bf16 (64-byte rows, unpadded); the 16-byte chunk c of row r
// lives at chunk c ^ ((r >> 2) & 3), which makes both the ds_write_b128 of the loader (4 lanes per row) and the
// ds_read_b128 of the fragments (lane = row, 8 consecutive k) bank-conflict free.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b)
{
    f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32, RNE
}

__device__ __forceinline__ void split2(float a, float b, uint32_t &x0, uint32_t &x1, uint32_t &x2)
{
    x0 = pk_bf16(a, b);
    a -= __uint_as_float(x0 << 16);
    b -= __uint_as_float(x0 & 0xffff0000u);
    x1 = pk_bf16(a, b);
    a -= __uint_as_float(x1 << 16);
    b -= __uint_as_float(x1 & 0xffff0000u);
    x2 = pk_bf16(a, b);
}

// 8 consecutive k of one row -> one 16-byte chunk in each of the three planes
__device__ __forceinline__ void split8_store(const float4 &u, const float4 &v, char *plane0, int plane_bytes, int off)
{
    uint4 w0, w1, w2;
    split2(u.x, u.y, w0.x, w1.x, w2.x);
    split2(u.z, u.w, w0.y, w1.y, w2.y);
    split2(v.x, v.y, w0.z, w1.z, w2.z);
    split2(v.z, v.w, w0.w, w1.w, w2.w);
    *reinterpret_cast<uint4 *>(plane0 + off) = w0;
    *reinterpret_cast<uint4 *>(plane0 + plane_bytes + off) = w1;
    *reinterpret_cast<uint4 *>(plane0 + 2 * plane_bytes + off) = w2;
}

// (-DX6_PROBE, dbg & 2: the same stores without the split's arithmetic)
__device__ __forceinline__ void raw8_store(const float4 &u, const float4 &v, char *plane0, int plane_bytes, int off)
{
    uint4 w0 = make_uint4(__float_as_uint(u.x), __float_as_uint(u.y), __float_as_uint(v.x), __float_as_uint(v.y));
    uint4 w1 = make_uint4(__float_as_uint(u.z), __float_as_uint(u.w), __float_as_uint(v.z), __float_as_uint(v.w));
    *reinterpret_cast<uint4 *>(plane0 + off) = w0;
    *reinterpret_cast<uint4 *>(plane0 + plane_bytes + off) = w1;
    *reinterpret_cast<uint4 *>(plane0 + 2 * plane_bytes + off) = w0;
}

__device__ __forceinline__ int x6_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// MI x NJ 32x32 MFMA tiles per wave, waves 2 x 2: block tile (64 MI) x (64 NJ).  (2, 2): 128 x 128, three workgroups
// per CU; (2, 4): 128 x 256 -- for N = 256 the A panel is then read, split and staged once -- two per CU; (1, 2)
// for small problems.
template <int MI, int NJ, int DBG = 0> __global__ __launch_bounds__(256, NJ == 4 ? 2 : 3) void gemm_nt_x6_kernel(GemmNTArgs p)
{
    constexpr int BM = 64 * MI, BNX = 64 * NJ;
    constexpr int PA = BM * 64, PB = BNX * 64;             // bytes per plane
    __shared__ __attribute__((aligned(16))) char lds[3 * (PA + PB)];
    char *As = lds;                   // [3][BM][64 B]
    char *Bs = lds + 3 * PA;          // [3][BNX][64 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    const int tiles_n = (p.N + BNX - 1) / BNX;
    long mt = blockIdx.x / tiles_n;
    int n0 = (int)(blockIdx.x % tiles_n) * BNX;
    if (p.xcd_remap) {
        // workgroups go to the 8 XCDs round robin: give the column panels of one row tile to ONE XCD, back to back, so that
        // the A rows they share are fetched into one L2 once instead of into tiles_n of them
        const long j = blockIdx.x >> 3;
        mt = (j / tiles_n) * 8 + (blockIdx.x & 7);
        n0 = (int)(j % tiles_n) * BNX;
    }
    long m0 = mt * BM, m_end = p.M;
    if (p.seg_rows > 0) {
        const int tps = (p.seg_valid + BM - 1) / BM;       // tiles per segment
        const long g = mt / tps;
        m0 = g * p.seg_rows + (mt - g * tps) * BM;
        m_end = g * p.seg_rows + p.seg_valid;
    }

    // loader: thread = (row tid/4 [+64 q], 8 consecutive k at (tid%4)*8): two float4 per row, MI rows of A, NJ of B
    const int lrow = tid >> 2;
    const int lchunk = tid & 3;
    const int lk = lchunk * 8;
    float4 ra[MI][2], rb[NJ][2];

    const int kbeg = blockIdx.y * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk_full = (kend - kbeg) / BK;
    auto load_tiles = [&](int kt) {
        const int k0 = kbeg + kt * BK + lk;
        if (kt < nk_full) {
#pragma unroll
            for (int q = 0; q < MI; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) ra[q][c] = ld4<true, false>(p.A, m0 + lrow + 64 * q, p.M, p.lda, k0 + 4 * c, kend);
#pragma unroll
            for (int q = 0; q < NJ; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) rb[q][c] = ld4<true, false>(p.B, n0 + lrow + 64 * q, p.N, p.ldb, k0 + 4 * c, kend);
        } else {
#pragma unroll
            for (int q = 0; q < MI; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) ra[q][c] = ld4<true, true>(p.A, m0 + lrow + 64 * q, p.M, p.lda, k0 + 4 * c, kend);
#pragma unroll
            for (int q = 0; q < NJ; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) rb[q][c] = ld4<true, true>(p.B, n0 + lrow + 64 * q, p.N, p.ldb, k0 + 4 * c, kend);
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    constexpr int dbg = DBG;
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr ((dbg & 2) != 0) {
#pragma unroll
            for (int q = 0; q < MI; ++q) raw8_store(ra[q][0], ra[q][1], As, PA, x6_off(lrow + 64 * q, lchunk));
#pragma unroll
            for (int q = 0; q < NJ; ++q) raw8_store(rb[q][0], rb[q][1], Bs, PB, x6_off(lrow + 64 * q, lchunk));
        } else {
#pragma unroll
        for (int q = 0; q < MI; ++q) split8_store(ra[q][0], ra[q][1], As, PA, x6_off(lrow + 64 * q, lchunk));
#pragma unroll
        for (int q = 0; q < NJ; ++q) split8_store(rb[q][0], rb[q][1], Bs, PB, x6_off(lrow + 64 * q, lchunk));
        }
        __syncthreads();
        if (kt + 1 < nk && !(dbg & 8)) load_tiles(kt + 1);

#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            // lane (r32, h) holds k = kk*16 + 8h .. +7 of its row: chunk 2 kk + h
            bf16x8_t fa[MI][3];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    fa[i][t] = *reinterpret_cast<const bf16x8_t *>(As + t * PA + x6_off(wm * 32 * MI + i * 32 + r32, 2 * kk + h));
#pragma unroll
            for (int jp = 0; jp < NJ; jp += 2) {           // two column tiles at a time: bounds the fragment registers
                bf16x8_t fb[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        fb[j][t] = *reinterpret_cast<const bf16x8_t *>(Bs + t * PB + x6_off(wn * 32 * NJ + (jp + j) * 32 + r32, 2 * kk + h));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x16 &c = acc[i][jp + j];
                        if constexpr ((dbg & 4) != 0) { c[0] += (float)fa[i][0][0] + (float)fb[j][1][0] + (float)fa[i][2][1] + (float)fb[j][2][0] + (float)fa[i][1][0] + (float)fb[j][0][0]; continue; }
                        // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
    if constexpr ((dbg & 1) != 0) { if (acc[0][0][0] == 123.456f) p.C[0] = acc[MI - 1][NJ - 1][3]; return; }
    static_assert(4 * 32 * (32 * NJ + 4) * 4 <= 3 * (PA + PB), "the staged epilogue lives in the operand tiles");
    if (p.vec_out) nt_epilogue_staged<MI, NJ>(p, acc, lds, m0, m_end, n0, wave, wm, wn, lane);      // (behind the loop's last barrier)
    else nt_epilogue<MI, NJ>(p, acc, m0, m_end, n0, wm, wn, r32, h);
}


// ------------------------------------------------------------------------------------------------
// LDS image of one 16-k stage of the pipelined kernels (gemm_tn_x6p_kernel; the NT twin is tools/experiments/gemm_nt_x6p_kernel.inc):
// three planes, each [2 halves of the 16 k][rows][16 bytes]: lane (r32, h)'s fragment (8 consecutive k) is chunk r32 of half h --
// the 32 lanes of a half read 512 contiguous bytes (every 16-lane group of a ds_read_b128 its own 16 bank quads); the loader's
// stores stay conflict free because the second half starts 128 bytes off a multiple of 256.
template <int ROWS> __device__ __forceinline__ int x6p_off(int row, int half) { return half * (ROWS * 16 + 128) + row * 16; }
template <int ROWS> constexpr int x6p_plane() { return 2 * ROWS * 16 + 128; }


__global__ void gemm_tn_reduce_kernel(const float *slab, int S, int M, int N, float *C, long ldc, int conv_cin, int conv_k);

// A/B switch: CPC_GEMM_X6_OLD=1 selects the two-barrier TN kernel and the K-split rule of rounds 1-5
static bool x6_pipelined()
{
    static const bool old_kernels = getenv("CPC_GEMM_X6_OLD") != nullptr;
    return !old_kernels;
}

// few tiles but a long K (e.g. dC = dP . W, K = 12 H): K is split over blockIdx.y
// (Round 6, tools/x6_sweep.py: below K = 2048 a split costs more than the idle CUs it fills -- 7424 x 256 x 768: 34 us whole
//  against 44 in four parts; 8192 x 256 x 256: 13.5 against 27 in two; 7424 x 512 x 1536: 83 against 88 -- above it the parts stay
//  >= 512 k long.  CPC_GEMM_X6_OLD=1 keeps the rule of rounds 1-5 for A/B runs.)
static int nt_splits(long blocks, int K)
{
    if (!x6_pipelined()) {
        if (blocks >= 2 * 256 || K < 8 * BK) return 1;
        return (int)std::max<long>(1, std::min<long>(cdiv(3 * 256, blocks), K / (4 * BK)));
    }
    if (blocks >= 2 * 256 || K < 2048) return 1;
    return (int)std::max<long>(1, std::min<long>(cdiv(3 * 256, blocks), K / 512));
}

size_t gemm_nt_scratch_bytes(long M, int N, int K)
{
    // whichever tile the launcher picks: 64- or 128-row tiles, 128-column panels
    const long s1 = nt_splits(cdiv(M, 64) * cdiv(N, 128), K), s2 = nt_splits(cdiv(M, 128) * cdiv(N, 128), K);
    const long s = std::max(s1, s2);
    return s > 1 ? align_up(sizeof(float) * (size_t)(s + 1) * M * N, 256) : 0;
}

// RowMap::epi as a pass of its own, for the kernels whose epilogue does not do it (f32-MFMA yardstick, unaligned shapes)
__global__ void epi_pass_kernel(float *c, long n, RowMap map)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = c[i];
        if (map.epi == EPI_RELU_DROPOUT) c[i] = v > 0.f ? v * drop_mul(map.epi_seed, (uint64_t)i, map.epi_thresh, map.epi_scale) : 0.f;
        else c[i] = map.epi_gate[i] > 0.f ? v * map.epi_scale : 0.f;
    }
}

int gemm_nt(const float *A, long lda, const float *B, long ldb, float *C, long ldc, const float *bias,
            long M, int N, int K, const RowMap &map, hipStream_t st)
{
    CPC_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem M=%ld N=%d K=%d", M, N, K);
    GemmNTArgs a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.map = map;
    a.aligned = (K % 4 == 0) && (K >= 4) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) % 16 == 0);
    a.dbg = 0;
#ifdef X6_PROBE
    if (getenv("X6_DBG")) a.dbg = atoi(getenv("X6_DBG"));
#endif
    const bool native = g_gemm_mode.load() == 1;
    const bool split_kernels = a.aligned && !native;
    // segmented row tiling (split kernels): never multiply the junk rows at the end of a sample
    a.seg_rows = 0; a.seg_valid = 0;
    if (split_kernels && map.seg_rows > 0 && map.seg_valid > 0 && M % map.seg_rows == 0) { a.seg_rows = map.seg_rows; a.seg_valid = map.seg_valid; }
    auto m_tiles = [&](int bm) { return a.seg_rows > 0 ? (M / a.seg_rows) * cdiv(a.seg_valid, bm) : cdiv(M, bm); };
    // tile choice: 128 x 256 (A read, split and staged once per 256 columns; two workgroups per CU) when N is made of
    // 256-column panels and the grid still fills the chip twice over; else 128 x 128 (three per CU); 64 x 128 when
    // even that leaves CUs with fewer than two workgroups
    int mi = 2, nj = 2;
    if (split_kernels && N % 256 == 0 && m_tiles(128) * (N / 256) >= 2 * 256) {
        // ... unless the 128 x 128 tiling fills its rounds so much better that it wins anyway (e.g. the predictor product,
        // 768 tiles on 512 places = two rounds at 75 %, against 1536 on 768 = two full rounds).  Measured full-grid rates:
        // 165 (128 x 256) against 150 (128 x 128) TFLOP/s; with ONE column panel the wide tile also reads A once.
        auto fill = [](long blocks, long places) { return (double)blocks / (double)(cdiv(blocks, places) * places); };
        const double wide = 165.0 * fill(m_tiles(128) * (N / 256), 2 * 256), narrow = 150.0 * fill(m_tiles(128) * (N / 128), 3 * 256);
        static const bool wide_always = getenv("CPC_GEMM_WIDE_ALWAYS") != nullptr;        // A/B switch
        nj = (N == 256 || wide >= narrow || wide_always) ? 4 : 2;
    } else if (a.aligned && m_tiles(128) * cdiv(N, BN) < 2 * 256) mi = 1;
    // (tile / split sweeps of tools/x6_sweep.py: CPC_GEMM_TILE="mi,nj", CPC_GEMM_SPLITS=n)
    static const char *tile_env = getenv("CPC_GEMM_TILE"), *splits_env = getenv("CPC_GEMM_SPLITS");
    if (tile_env != nullptr && split_kernels) {
        if (sscanf(tile_env, "%d,%d", &mi, &nj) != 2 || !((mi == 1 && nj == 2) || (mi == 2 && (nj == 2 || nj == 4)))) { mi = 2; nj = 2; }
    }
    const long blocks = m_tiles(64 * mi) * cdiv(N, 64 * nj);
    CPC_REQUIRE(blocks <= 2147483647L, "gemm_nt: grid too large (%ld blocks)", blocks);
    // few tiles but a long K (e.g. dC = dP . W, K = 12 H): split K over blockIdx.y, partial products are
    // atomically added into a zeroed C (dense, unmapped outputs only)
    int splits = 1;
    if (a.aligned && !map.enabled && ldc == N && map.epi == EPI_NONE) splits = splits_env != nullptr ? std::max(1, atoi(splits_env)) : nt_splits(blocks, K);
    a.kchunk = (int)(cdiv(cdiv(K, splits), BK) * BK);
    splits = (int)cdiv(K, a.kchunk);
    static const bool no_remap = getenv("CPC_GEMM_NO_XCD") != nullptr;
    a.xcd_remap = (split_kernels && !no_remap && cdiv(N, 64 * nj) > 1 && m_tiles(64 * mi) % 8 == 0) ? 1 : 0;
    // the partial products go to slabs summed in a fixed order when the caller lent the room, else straight into a zeroed
    // C with atomics
    a.slabs = nullptr;
    if (splits > 1 && map.splitk_scratch != nullptr && map.splitk_bytes >= sizeof(float) * (size_t)splits * M * N &&
        reinterpret_cast<uintptr_t>(map.splitk_scratch) % 16 == 0 && M <= 2147483647L)
        a.slabs = static_cast<float *>(map.splitk_scratch);
    if (splits > 1 && a.slabs == nullptr) CPC_CHECK_HIP(hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N, st));
    // dense outputs leave in 16-byte pieces through LDS (nt_epilogue_staged); non-temporal once the output is larger than the L2s
    // together (it is read back from the memory side either way: measured -0.03 .. -0.05 ms per step on the transformer / large
    // configurations, nothing at CPC-small)
    a.vec_out = (!map.enabled && (splits == 1 || a.slabs != nullptr) && N % 4 == 0 && ldc % 4 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0)
                    ? ((size_t)M * N * sizeof(float) >= (32u << 20) ? 2 : 1) : 0;
    CPC_REQUIRE(map.epi == EPI_NONE || (!map.enabled && splits == 1 && ldc == N),
                "gemm_nt: a fused elementwise epilogue needs a dense output (M=%ld N=%d K=%d ldc=%ld)", M, N, K, ldc);
    const bool epi_pass = map.epi != EPI_NONE && !(a.vec_out != 0 && split_kernels);     // (only the staged store does it in place)
    dim3 grid((unsigned)blocks, (unsigned)splits);
    static const bool log_shapes = getenv("CPC_GEMM_LOG") != nullptr;        // tools/x6_shapes.py: one line per launch
    if (log_shapes) fprintf(stderr, "cpc_gemm nt M=%ld N=%d K=%d lda=%ld ldb=%ld ldc=%ld tile=%dx%d grid=%ld splits=%d map=%d epi=%d kernel=%s\n", M, N, K, lda,
                            ldb, ldc, 64 * mi, 64 * nj, blocks, splits, map.enabled, map.epi, split_kernels ? "x6" : "f32");
    ProfScope prof(PROF_GEMM_NT, st);
    if (!a.aligned) hipLaunchKernelGGL((gemm_nt_kernel<false, 2>), grid, dim3(256), 0, st, a);
    else if (native && mi == 1) hipLaunchKernelGGL((gemm_nt_kernel<true, 1>), grid, dim3(256), 0, st, a);
    else if (native) hipLaunchKernelGGL((gemm_nt_kernel<true, 2>), grid, dim3(256), 0, st, a);
#ifdef X6_PROBE
#define X6_CASE(D) else if (a.dbg == D && nj == 4) hipLaunchKernelGGL((gemm_nt_x6_kernel<2, 4, D>), grid, dim3(256), 0, st, a); \
    else if (a.dbg == D && mi == 1) hipLaunchKernelGGL((gemm_nt_x6_kernel<1, 2, D>), grid, dim3(256), 0, st, a); \
    else if (a.dbg == D) hipLaunchKernelGGL((gemm_nt_x6_kernel<2, 2, D>), grid, dim3(256), 0, st, a);
    X6_CASE(1) X6_CASE(2) X6_CASE(4) X6_CASE(8) X6_CASE(3) X6_CASE(6) X6_CASE(14) X6_CASE(15)
#endif
    else if (nj == 4) hipLaunchKernelGGL((gemm_nt_x6_kernel<2, 4>), grid, dim3(256), 0, st, a);
    else if (mi == 1) hipLaunchKernelGGL((gemm_nt_x6_kernel<1, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_nt_x6_kernel<2, 2>), grid, dim3(256), 0, st, a);
    CPC_CHECK_LAUNCH("gemm_nt_kernel");
    if (epi_pass) {
        hipLaunchKernelGGL(epi_pass_kernel, dim3((unsigned)std::min<long>(cdiv(M * N, 256), 4096)), dim3(256), 0, st, C, M * N, map);
        CPC_CHECK_LAUNCH("epi_pass_kernel");
    }
    if (a.slabs != nullptr) {
        const long total = M * N;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)std::min<long>(cdiv(total, 256), 2048)), dim3(256), 0, st, a.slabs,
                           splits, (int)M, N, C, ldc, 0, 0);
        CPC_CHECK_LAUNCH("gemm_nt split-K reduce");
    }
    return CPC_OK;
}

// ------------------------------------------------------------------------------------------------
struct GemmTNArgs {
    const float *A; long lda;
    const float *B; long ldb;
    float *slab;       // [S][M][N]
    int M, N;
    long R, chunk;     // rows per split (multiple of BK)
    int aligned;
    int xcd_remap;     // split kernel: all tiles of a row slab on one XCD (slab count % 8 == 0)
};

template <bool ALIGNED> __global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs p)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDS_TN];
    float *As = lds;                   // [BK][LDS_TN]  (row r, column i)
    float *Bs = lds + BK * LDS_TN;     // [BK][LDS_TN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    const int i0 = blockIdx.y * BM;
    const int j0 = blockIdx.x * BN;
    const long rbeg = (long)blockIdx.z * p.chunk;
    long rend = rbeg + p.chunk;
    if (rend > p.R) rend = p.R;

    // loader: tile [32 rows][128 cols] = 1024 float4; slot = tid + 256*q -> row = slot/32, c4 = slot%32
    const int lr = tid >> 5;           // + 8*q
    const int lc = (tid & 31) * 4;
    float4 ra[4], rb[4];
    // rows r are the REDUCTION index: rows >= rend must contribute zero (tail tile only); columns beyond
    // M / N are clamped (they only feed outputs that are never stored).
    const int ca = ALIGNED ? min(i0 + lc, p.M - 4) : i0 + lc;
    const int cb = ALIGNED ? min(j0 + lc, p.N - 4) : j0 + lc;
    auto load_tiles = [&](long r0) {
        const bool full = r0 + BK <= rend;              // uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long r = r0 + lr + 8 * q;
            const long rc = (full || r < rend) ? r : rend - 1;
            float4 va, vb;
            if (ALIGNED) {
                va = *reinterpret_cast<const float4 *>(p.A + rc * p.lda + ca);
                vb = *reinterpret_cast<const float4 *>(p.B + rc * p.ldb + cb);
            } else {
                va = ld4<false, true>(p.A + i0 + lc, rc, rend, p.lda, 0, p.M - (i0 + lc));
                vb = ld4<false, true>(p.B + j0 + lc, rc, rend, p.ldb, 0, p.N - (j0 + lc));
            }
            ra[q] = va;
            rb[q] = vb;
        }
        return full;
    };
    // zeroing of the tail tile's out-of-range rows is applied when the registers are written to LDS
    // (after the MFMAs of the previous tile), so the loads above stay free of dependent selects.

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (rbeg < rend) {
        bool full = load_tiles(rbeg);
        for (long r0 = rbeg; r0 < rend; r0 += BK) {
            if (!full) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + lr + 8 * q >= rend) {
                        ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                        rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<float4 *>(&As[(lr + 8 * q) * LDS_TN + lc]) = ra[q];
                *reinterpret_cast<float4 *>(&Bs[(lr + 8 * q) * LDS_TN + lc]) = rb[q];
            }
            __syncthreads();
            if (r0 + BK < rend) full = load_tiles(r0 + BK);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                // MFMA tile (i_t, j_t) owns rows wm*64 + 2*r + i_t and columns wn*64 + 2*c + j_t (interleaved),
                // so one ds_read_b64 per operand feeds both tiles of that operand.
                const float2 a = *reinterpret_cast<const float2 *>(&As[(2 * kk + h) * LDS_TN + wm * 64 + 2 * r32]);
                const float2 b = *reinterpret_cast<const float2 *>(&Bs[(2 * kk + h) * LDS_TN + wn * 64 + 2 * r32]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // slab[z][i][j]: i = i0 + wm*64 + 2*((e&3) + 8*(e>>2) + 4h) + i_t ; j = j0 + wn*64 + 2*r32 + j_t
    float *slab = p.slab + (long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * h) + it;
            if (i >= p.M) continue;
            const int j = j0 + wn * 64 + 2 * r32;
            float *dst = slab + (long)i * p.N + j;
            if (j + 1 < p.N && (p.N % 2 == 0)) {
                *reinterpret_cast<float2 *>(dst) = make_float2(acc[it][0][e], acc[it][1][e]);
            } else {
                if (j < p.N) dst[0] = acc[it][0][e];
                if (j + 1 < p.N) dst[1] = acc[it][1][e];
            }
        }
}

// TN product on the bf16 pipe (same three-term split as gemm_nt_x6_kernel).  The reduction index r is the ROW
// of both operands, so the loader transposes while it splits: a thread takes an 8(r) x 4(column) micro tile
// (8 row loads of 16 bytes), and writes, for each of its 4 columns, the 8 consecutive r as one 16-byte chunk per
// plane -> the LDS image is [column][32 r] bf16, the layout (and swizzle) the NT kernel reads its fragments from.
__global__ __launch_bounds__(256, 3) void gemm_tn_x6_kernel(GemmTNArgs p)
{
    constexpr int PL = BM * 64;                            // bytes per plane (BM == BN)
    __shared__ __attribute__((aligned(16))) char lds[6 * PL];
    char *As = lds;                   // [3][BM][64 B]
    char *Bs = lds + 3 * PL;          // [3][BN][64 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        // workgroups go to the 8 XCDs round robin in dispatch order (x fastest): give ALL tiles of one row slab to one XCD,
        // so that the slab's A and B rows are fetched into one L2 instead of into eight
        const unsigned b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned tiles = gridDim.x * gridDim.y, j = b >> 3, tile = j % tiles;
        bz = (int)((j / tiles) * 8 + (b & 7));
        by = (int)(tile / gridDim.x);
        bx = (int)(tile % gridDim.x);
    }
    const int i0 = by * BM;
    const int j0 = bx * BN;
    const long rbeg = (long)bz * p.chunk;
    long rend = rbeg + p.chunk;
    if (rend > p.R) rend = p.R;

    // waves 0,1 stage A, waves 2,3 stage B: micro tile u -> rows 8*(u&3) .. +7, columns 4*(u>>2) .. +3
    const bool isA = tid < 128;                                // wave uniform
    const int u = tid & 127;
    const int rg = u & 3, cg = u >> 2;
    const float *src = isA ? p.A : p.B;
    const long ld = isA ? p.lda : p.ldb;
    const int ncols = isA ? p.M : p.N;
    const int col = min((isA ? i0 : j0) + 4 * cg, ncols - 4);  // clamped columns only feed outputs never stored
    char *dst = isA ? As : Bs;
    float4 rv[8];
    auto load_tiles = [&](long r0) {
        const bool full = r0 + BK <= rend;              // uniform
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const long r = r0 + 8 * rg + d;
            const long rc = (full || r < rend) ? r : rend - 1;
            rv[d] = *reinterpret_cast<const float4 *>(src + rc * ld + col);
        }
        return full;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (rbeg < rend) {
        bool full = load_tiles(rbeg);
        for (long r0 = rbeg; r0 < rend; r0 += BK) {
            if (!full) {
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    if (r0 + 8 * rg + d >= rend) rv[d] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            split8_store(make_float4(rv[0].x, rv[1].x, rv[2].x, rv[3].x), make_float4(rv[4].x, rv[5].x, rv[6].x, rv[7].x),
                         dst, PL, x6_off(4 * cg + 0, rg));
            split8_store(make_float4(rv[0].y, rv[1].y, rv[2].y, rv[3].y), make_float4(rv[4].y, rv[5].y, rv[6].y, rv[7].y),
                         dst, PL, x6_off(4 * cg + 1, rg));
            split8_store(make_float4(rv[0].z, rv[1].z, rv[2].z, rv[3].z), make_float4(rv[4].z, rv[5].z, rv[6].z, rv[7].z),
                         dst, PL, x6_off(4 * cg + 2, rg));
            split8_store(make_float4(rv[0].w, rv[1].w, rv[2].w, rv[3].w), make_float4(rv[4].w, rv[5].w, rv[6].w, rv[7].w),
                         dst, PL, x6_off(4 * cg + 3, rg));
            __syncthreads();
            if (r0 + BK < rend) full = load_tiles(r0 + BK);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8_t fa[2][3], fb[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        fa[i][t] = *reinterpret_cast<const bf16x8_t *>(As + t * PL + x6_off(wm * 64 + i * 32 + r32, 2 * kk + h));
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        fb[j][t] = *reinterpret_cast<const bf16x8_t *>(Bs + t * PL + x6_off(wn * 64 + j * 32 + r32, 2 * kk + h));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
    }

    // slab[z][i][j]: i = i0 + wm*64 + it*32 + (e&3) + 8*(e>>2) + 4h ; j = j0 + wn*64 + jt*32 + r32
    float *slab = p.slab + (long)bz * p.M * p.N;
    // (16-byte pieces through LDS as in the NT kernel were measured here too: no difference -- tools/scratch/ab_tn_out.sh)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm * 64 + it * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i >= p.M) continue;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int j = j0 + wn * 64 + jt * 32 + r32;
                if (j < p.N) slab[(long)i * p.N + j] = acc[it][jt][e];
            }
        }
}

// The TN product as a SOFTWARE PIPELINE (round 6).  gemm_tn_x6_kernel stores a step of 32 rows into LDS, waits at a barrier,
// multiplies, waits again.  Here a stage is 16 reduction rows (one MFMA's depth), LDS holds TWO stages and every wave splits and
// stores stage s + 1 while the matrix pipe works through the MFMAs it has issued for stage s (sched_group_barrier interleaves
// them: no branch inside a stage, clamped stage numbers instead); raw operands arrive two stages ahead in two register sets; ONE
// barrier per stage.  5-9 % faster than the two-barrier kernel on the large weight-gradient products (predictor 3072 x 256 over
// 7424 rows: 83 against 91 us; CPC-large 6144 x 512: 287 against 305), equal on the small ones (profiles/r06_x6_sweep.md).
// The loader transposes while it splits: waves 0, 1 stage A, waves 2, 3 stage B; a thread takes an 8 (r) x 2 (columns) micro tile
// (eight 8-byte loads; a wave's load covers 256 contiguous bytes of two rows) and writes, per column, its 8 consecutive r as one
// 16-byte chunk per plane.
__global__ __launch_bounds__(256, 3) void gemm_tn_x6p_kernel(GemmTNArgs p)
{
    constexpr int PL = x6p_plane<BM>();                    // bytes per plane of one stage (BM == BN)
    constexpr int STAGE = 6 * PL;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned tiles = gridDim.x * gridDim.y, j = b >> 3, tile = j % tiles;
        bz = (int)((j / tiles) * 8 + (b & 7));
        by = (int)(tile / gridDim.x);
        bx = (int)(tile % gridDim.x);
    }
    const int i0 = by * BM;
    const int j0 = bx * BN;
    const long rbeg = (long)bz * p.chunk;
    long rend = rbeg + p.chunk;
    if (rend > p.R) rend = p.R;

    const bool isA = tid < 128;                                // wave uniform
    const int u = tid & 127;
    const int rg = u & 1, cg = u >> 1;
    const long ld = isA ? p.lda : p.ldb;
    const int ncols = isA ? p.M : p.N;
    const int col = min((isA ? i0 : j0) + 2 * cg, ncols - 2);  // clamped columns only feed outputs never stored
    const float *src = (isA ? p.A : p.B) + col;
    const int plane_off = isA ? 0 : 3 * PL;
    float2 rv[2][8];                                           // two register sets: stage s lives in set s & 1

    const int nk = rbeg < rend ? (int)((rend - rbeg + 15) / 16) : 0;
    auto load_stage = [&](int st, auto SET, auto KT) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        constexpr bool kt = decltype(KT)::value;
        const long r0 = rbeg + (long)min(st, nk - 1) * 16 + 8 * rg;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            if (!kt) rv[set][d] = *reinterpret_cast<const float2 *>(src + (r0 + d) * ld);
            else {
                const float2 v = *reinterpret_cast<const float2 *>(src + min(r0 + d, rend - 1) * ld);
                rv[set][d] = r0 + d < rend ? v : make_float2(0.f, 0.f);     // rows beyond the range contribute zero
            }
        }
    };
    auto store_stage = [&](auto SET, char *buf) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const float2 *r = rv[set];
        split8_store(make_float4(r[0].x, r[1].x, r[2].x, r[3].x), make_float4(r[4].x, r[5].x, r[6].x, r[7].x), buf + plane_off, PL,
                     x6p_off<BM>(2 * cg, rg));
        split8_store(make_float4(r[0].y, r[1].y, r[2].y, r[3].y), make_float4(r[4].y, r[5].y, r[6].y, r[7].y), buf + plane_off, PL,
                     x6p_off<BM>(2 * cg + 1, rg));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto stage = [&](int st, auto PAR, auto KT) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value;
        const char *As = lds + par * STAGE, *Bs = As + 3 * PL;
        bf16x8_t fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) fa[i][t] = *reinterpret_cast<const bf16x8_t *>(As + t * PL + x6p_off<BM>(wm * 64 + i * 32 + r32, h));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t) fb[j][t] = *reinterpret_cast<const bf16x8_t *>(Bs + t * PL + x6p_off<BM>(wn * 64 + j * 32 + r32, h));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
            }
        store_stage(std::integral_constant<int, par ^ 1>{}, lds + (par ^ 1) * STAGE);
        load_stage(st + 3, std::integral_constant<int, par ^ 1>{}, KT);
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
            if (g % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
            if (g % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
        }
        __syncthreads();
    };
    auto run = [&](auto KT) __attribute__((always_inline)) {
        load_stage(0, I0{}, KT);
        load_stage(1, I1{}, KT);
        store_stage(I0{}, lds);
        load_stage(2, I0{}, KT);
        __syncthreads();
        int st = 0;
        for (; st + 1 < nk; st += 2) {
            stage(st, I0{}, KT);
            stage(st + 1, I1{}, KT);
        }
        if (st < nk) stage(st, I0{}, KT);
    };
    if (nk > 0) {
        if ((rend - rbeg) % 16 != 0) run(std::true_type{}); else run(std::false_type{});
    }

    // slab[z][i][j]: i = i0 + wm*64 + it*32 + (e&3) + 8*(e>>2) + 4h ; j = j0 + wn*64 + jt*32 + r32
    float *slab = p.slab + (long)bz * p.M * p.N;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm * 64 + it * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i >= p.M) continue;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int j = j0 + wn * 64 + jt * 32 + r32;
                if (j < p.N) slab[(long)i * p.N + j] = acc[it][jt][e];
            }
        }
}

// out = sum over slabs; optional Conv1d weight re-layout (column jj*cin+ci -> [ci][jj])
__global__ void gemm_tn_reduce_kernel(const float *slab, int S, int M, int N, float *C, long ldc, int conv_cin, int conv_k)
{
    const long total = (long)M * N;
    if (N % 4 == 0 && conv_cin % 4 == 0 && reinterpret_cast<uintptr_t>(slab) % 16 == 0) {
        // four outputs per thread, 16-byte loads, four slabs in flight (a latency-bound pass); slab order fixed
        const long total4 = total / 4;
        const float4 *s4 = reinterpret_cast<const float4 *>(slab);
        for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int z = 0;
            for (; z + 4 <= S; z += 4) {
                const float4 v0 = s4[(long)z * total4 + i4], v1 = s4[(long)(z + 1) * total4 + i4];
                const float4 v2 = s4[(long)(z + 2) * total4 + i4], v3 = s4[(long)(z + 3) * total4 + i4];
                acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
                acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
                acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
                acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
            }
            for (; z < S; ++z) {
                const float4 v = s4[(long)z * total4 + i4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            const long idx = i4 * 4;
            const int i = (int)(idx / N), j = (int)(idx - (long)i * N);
            if (conv_cin > 0) {                             // conv_cin % 4 == 0: the four elements share the tap
                const int jj = j / conv_cin, ci = j - jj * conv_cin;
                float *dst = C + (long)i * conv_cin * conv_k + (long)ci * conv_k + jj;
                dst[0] = acc.x; dst[conv_k] = acc.y; dst[2 * conv_k] = acc.z; dst[3 * conv_k] = acc.w;
            } else {
                float *dst = C + (long)i * ldc + j;
                dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w;
            }
        }
        return;
    }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += slab[(long)z * total + idx];
        const int i = (int)(idx / N), j = (int)(idx - (long)i * N);
        if (conv_cin > 0) {
            const int jj = j / conv_cin, ci = j - jj * conv_cin;
            C[(long)i * conv_cin * conv_k + (long)ci * conv_k + jj] = s;
        } else {
            C[(long)i * ldc + j] = s;
        }
    }
}

static int tn_splits(int M, int N, long R, long *chunk_out)
{
    const long tiles = cdiv(M, BM) * cdiv(N, BN);
    long S = 768 / tiles;                     // one full wave of 3 workgroups per CU (256 CUs)
    static const char *splits_env = getenv("CPC_GEMM_TN_SPLITS");     // (tools/x6_sweep.py)
    if (splits_env != nullptr) S = std::max(1, atoi(splits_env));
    const long max_s = cdiv(R, 4 * BK);      // at least 128 rows per split
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    long chunk = cdiv(cdiv(R, S), BK) * BK;
    S = cdiv(R, chunk);
    *chunk_out = chunk;
    return (int)S;
}

// (Round 4 measured the large weight-gradient products of CPC-large -- 16384 x 1536 x 512, 8192 x 6144 x 512 -- through the
//  plane-fed kernel instead, operands split into planes by two streaming passes first: 222 / 344 us against 171 / 332 us for this
//  family alone, i.e. no gain: 12-48 output tiles of 256 x 256 leave a quarter of the chip idle or cost XCD locality.  Their 380 us
//  each INSIDE the step is contention with the criterion's sum running beside them, not this kernel: DESIGN.md.)
size_t gemm_tn_scratch_bytes(int M, int N, long R)
{
    long chunk;
    const int S = tn_splits(M, N, R, &chunk);
    return align_up((size_t)S * M * N * sizeof(float), 256);
}

int gemm_tn(const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N, long R,
            void *scratch, size_t scratch_bytes, int conv_cin, int conv_k, hipStream_t st)
{
    CPC_REQUIRE(M > 0 && N > 0 && R > 0, "gemm_tn: empty problem M=%d N=%d R=%ld", M, N, R);
    GemmTNArgs a;
    long chunk;
    const int S = tn_splits(M, N, R, &chunk);
    if ((size_t)S * M * N * sizeof(float) > scratch_bytes) {
        set_error("gemm_tn: scratch too small (%zu < %zu)", scratch_bytes, (size_t)S * M * N * sizeof(float));
        return CPC_ERR_WORKSPACE;
    }
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.slab = static_cast<float *>(scratch);
    a.M = M; a.N = N; a.R = R; a.chunk = chunk;
    a.aligned = (M % 4 == 0) && (N % 4 == 0) && (M >= 4) && (N >= 4) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) % 16 == 0);
    dim3 grid((unsigned)cdiv(N, BN), (unsigned)cdiv(M, BM), (unsigned)S);
    static const bool no_remap = getenv("CPC_GEMM_NO_XCD") != nullptr;
    a.xcd_remap = (!no_remap && S % 8 == 0 && grid.x * grid.y > 1) ? 1 : 0;
    static const bool log_shapes = getenv("CPC_GEMM_LOG") != nullptr;
    if (log_shapes) fprintf(stderr, "cpc_gemm tn M=%d N=%d R=%ld lda=%ld ldb=%ld ldc=%ld tile=128x128 grid=%ld splits=%d chunk=%ld kernel=%s\n", M, N, R, lda, ldb,
                            ldc, cdiv(N, BN) * cdiv(M, BM), S, chunk, a.aligned ? "x6" : "f32");
    ProfScope prof(PROF_GEMM_TN, st);
    const bool native = g_gemm_mode.load() == 1;
    if (a.aligned && !native && x6_pipelined()) hipLaunchKernelGGL(gemm_tn_x6p_kernel, grid, dim3(256), 0, st, a);
    else if (a.aligned && !native) hipLaunchKernelGGL(gemm_tn_x6_kernel, grid, dim3(256), 0, st, a);
    else if (a.aligned) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, st, a);
    CPC_CHECK_LAUNCH("gemm_tn_kernel");
    const long total = (long)M * N;
    int blocks = (int)(cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256));
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.slab, S, M, N, C, ldc, conv_cin, conv_k);
    CPC_CHECK_LAUNCH("gemm_tn_reduce_kernel");
    return CPC_OK;
}

int gemm_set_mode(int mode)
{
    const int prev = g_gemm_mode.load();
    if (mode == 0 || mode == 1 || mode == 2) g_gemm_mode.store(mode);
    return prev;
}

}  // namespace cpc

// ------------------------------------------------------------------------------------------------
namespace cpc { int gemm_mode() { return g_gemm_mode.load(); } }
extern "C" int cpc_gemm_set_mode(int mode) { return cpc::gemm_set_mode(mode); }

extern "C" int cpc_gemm_nt(const float *A, long lda, const float *B, long ldb, float *C, long ldc,
                           const float *bias, int M, int N, int K, cpc_stream_t stream)
{
    cpc::RowMap map{};
    return cpc::gemm_nt(A, lda, B, ldb, C, ldc, bias, M, N, K, map, static_cast<hipStream_t>(stream));
}

extern "C" size_t cpc_gemm_tn_scratch_bytes(int M, int N, long R) { return cpc::gemm_tn_scratch_bytes(M, N, R); }

extern "C" int cpc_gemm_tn(const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N,
                           long R, void *scratch, size_t scratch_bytes, cpc_stream_t stream)
{
    return cpc::gemm_tn(A, lda, B, ldb, C, ldc, M, N, R, scratch, scratch_bytes, 0, 0, static_cast<hipStream_t>(stream));
}
