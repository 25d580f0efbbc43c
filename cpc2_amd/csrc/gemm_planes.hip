// f32 GEMMs on the bf16 matrix pipe whose operands arrive ALREADY split into their three bf16 terms ("planes").
//
// The split x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), round to nearest) is exact to
// 2^-27 |x| (see gemm_f32.hip), so whoever produces an activation or lays out a weight can store it as the three planes
// without losing anything, and the product a.b = a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 then needs no arithmetic besides
// its MFMAs: the tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4), never through registers.
//
// Plane layout (PlanesOperand, common.h): 16-element chunks, chunk-major.  The 16 consecutive K elements (channels)
// 16c .. 16c+15 of signal row R are 32 contiguous bytes at chunk index (c * s + R % s) * rts + R / s: consecutive rows
// (of the same phase R % s, for the input of a stride-s Conv1d) are consecutive chunks, so the 32 rows x 16 k piece an
// MFMA operand is made of is ONE contiguous KiB whatever the row overlap of the implicit GEMM -- an LDS-DMA instruction
// then fetches whole cache lines (with row-major planes each lane would touch its own line and use a quarter of it).
//
//   gemm_nt_planes : C[map(m)][n] = sum_k A(m, k) * B(n, k) (+ bias[n]),  k = 16 ks + kl, ks -> (chunk c, tap j)
//
// Tile 256 x 256, 512 threads = 8 waves (2 x 4, wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32), K step 16:
// a stage is 48 pieces of 1 KiB = (operand, plane, 32-row block): piece q holds, at byte lane * 16, the 8 consecutive k
// (k half = lane >> 5) of row (lane & 31) of its block -- the operand layout of v_mfma_f32_32x32x16_bf16, so a fragment
// read is ds_read_b128 at lane * 16 + constant (conflict free by construction).  Three stages (144 KiB) form a ring.
// Schedule of stage t per wave (one s_barrier per stage, no point where the matrix pipe has nothing queued):
//   row blocks 0, 1, 2: the A fragments of block i + 1 are read while block i is multiplied;
//   after block 2: s_waitcnt vmcnt (the wave's own pieces of stage t + 1 have landed, stage t + 2 stays in flight),
//     s_barrier (=> stage t + 1 landed for everyone, everyone has read all of stage t), request stage t + 3 into
//     stage t's slot, read the B fragments and the first A fragments of stage t + 1 (second register set);
//   row block 3 is multiplied while those reads and the new requests are under way.
// All LDS traffic is issued from inline asm: hipcc drains vmcnt to 0 before any LDS read it can see behind an LDS-DMA.
#include "common.h"
#include <hip/hip_ext.h>
#include "ldsdma.h"
#include "rowcfg.h"

#include <algorithm>

namespace cpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;
// (frag_t of ldsdma.h: one MFMA operand (8 bf16) as four dwords: hipcc handles a bf16x8 value that crosses a branch element by element)

constexpr int PT_BM = 256, PT_BN = 256, PT_BK = 16;
constexpr int PT_PIECE = 1024;                    // bytes: 32 rows x 16 k x bf16
constexpr int PT_STAGE = 48 * PT_PIECE;           // A: 3 planes x 8 blocks, B: 3 planes x 8 blocks
constexpr int PT_RING = 3;
constexpr int PT_LDS = PT_RING * PT_STAGE;        // 147456 bytes
// "Pair" form of the NT kernel's K loop (a k = 2s convolution, forward or backward data): the taps j and j + s of a channel
// chunk read the SAME rows of the signal one row apart, so the A tile of the pair is staged ONCE -- 257 rows: 8 blocks + the row
// behind the tile as a ninth block -- and the second tap's fragments are read one row further down.  A double stage = A (27
// pieces) + B of tap j (24) + B of tap j + s (24) = 75 pieces instead of 96: an LDS-DMA piece costs the SIMD's matrix pipe
// ~60 cycles whichever wave issues it (measured: profiles/r04_planes_levers.md), so requests per MFMA are what bounds the loop.
constexpr int PP_A = 27 * PT_PIECE;
constexpr int PP_DS = 75 * PT_PIECE;              // 76800 bytes; ring of two double stages
constexpr int PP_LDS = 2 * PP_DS;                 // 153600 bytes

struct PlanesSide {                               // device view of a PlanesOperand
    const bf16_t *p; long plane;
    int kshift, sshift; long rts;
    int segv; long seg_q;
};

// weight-gradient form (TN): C[i][j] = sum_r X(r, i) * Y(r, j); column x of an operand is channel x % C of tap
// tap0 + x / C, i.e. element (x % C) of signal row r * s + tap (s = 1 << sshift)
struct PlanesTNSide { const bf16_t *p; long plane; int sshift; long rts; int tap0, C; };

struct PlanesNTArgs {
    PlanesSide A, B;
    PlanesTNSide TA, TB;                          // TN kernel only
    long R;                                       // TN: reduction rows (multiple of 32); blockIdx.z takes rchunk of them
    long rchunk;
    float *C; long ldc;
    const float *bias;
    long M; int N; int K;
    RowMap map;
    int tiles_m, tiles_n, xcd_remap;
    int kchunk;                                   // K range per blockIdx.y (multiple of 32)
    float *slabs;                                 // K split: partial product of blockIdx.y -> slabs + blockIdx.y * slab_rows * N
    long slab_rows;                               // output rows (after the row map)
    PlanesNormOut norm;                           // NORM kernels only (N == 256, no K split)
    int dbg;                                      // probes: 1 no loads after the prologue, 2 no MFMAs, 8 clock stamps, 16 half the A requests, 32 a third fewer fragment reads
    unsigned long long *stamps;                   // dbg & 8: [workgroup][8] = memtime, memrealtime at loop start and end, ...
};

// chunk index of K step ks, row q = 0
__device__ __forceinline__ long chunk0(const PlanesSide &o, int ks)
{
    // K order: chunk-major, and within a chunk the taps j and j + s next to each other (0, s, 1, s + 1, ...): they read the
    // same rows of the signal one apart, so the second one finds them in L2
    const int c = ks >> o.kshift, jj = ks - (c << o.kshift);
    const int j = (jj >> 1) + ((jj & 1) << o.sshift);
    const int smask = (1 << o.sshift) - 1;
    return (long)((c << o.sshift) + (j & smask)) * o.rts + (j >> o.sshift);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// NT: the piece holds the operand layout itself, one ds_read_b128 at lane * 16.  TN: the piece holds [4 r][2 chunks][4 r][16 x]
// (r = reduction row, x = output row / column) and the operand is read transposed, two ds_read_b64_tr_b16 (4 r each).
template <bool TN, int OFF> __device__ __forceinline__ frag_t lds_frag(unsigned addr)
{
    if constexpr (!TN) {
        frag_t v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
        return v;
    } else {
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                     : "=&v"(lo), "=&v"(hi)
                     : "v"(addr), "n"(OFF), "n"(OFF + 256)
                     : "memory");
        frag_t v = {lo[0], lo[1], hi[0], hi[1]};
        return v;
    }
}
__device__ __forceinline__ void lds_wait3(frag_t (&f)[3])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
}
__device__ __forceinline__ void lds_wait9(frag_t (&f)[3], frag_t (&g)[2][3])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[0][2]), "+v"(g[1][0]), "+v"(g[1][1]),
                   "+v"(g[1][2])
                 :
                 : "memory");
}

// product q (0..5, smallest terms first) of row block fragments a with column block fragments b, into c
template <int Q> __device__ __forceinline__ void mma1(f32x16 &c, const frag_t (&a)[3], const frag_t (&b)[3])
{
    constexpr int ia = Q == 0 ? 2 : Q == 1 ? 0 : Q == 2 ? 1 : Q == 3 ? 1 : 0;
    constexpr int ib = Q == 0 ? 0 : Q == 1 ? 2 : Q == 2 ? 1 : Q == 3 ? 0 : Q == 4 ? 1 : 0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ia]), __builtin_bit_cast(bf16x8_t, b[ib]), c, 0, 0, 0);
}
// the two column tiles alternate (each accumulator chain gets a slot of slack)
template <int Q> __device__ __forceinline__ void mma2(f32x16 &c0, f32x16 &c1, const frag_t (&a)[3], const frag_t (&b)[2][3])
{
    mma1<Q>(c0, a, b[0]);
    mma1<Q>(c1, a, b[1]);
}
// TERMS = 6: the exact product.  TERMS = 3 (cpc_gemm_set_mode(2), opt-in): a0 b0 + a0 b1 + a1 b0 only -- 16 bits of product
// mantissa instead of 24 (TF32, what the reference's convolutions get on its own GPUs by default, keeps 10); the planes are moved
// as before, the three small products are not multiplied.
template <int TERMS> __device__ __forceinline__ void mma6(f32x16 &c0, f32x16 &c1, const frag_t (&a)[3], const frag_t (&b)[2][3], bool skip)
{
    if (skip) {
        asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]));
        return;
    }
    if constexpr (TERMS == 6) { mma2<0>(c0, c1, a, b); mma2<1>(c0, c1, a, b); mma2<2>(c0, c1, a, b); }
    else asm volatile("" ::"v"(a[2]), "v"(b[0][2]), "v"(b[1][2]));
    mma2<3>(c0, c1, a, b); mma2<4>(c0, c1, a, b); mma2<5>(c0, c1, a, b);
}

template <bool TN, int BLK> __device__ __forceinline__ void read_a(frag_t (&a)[3], unsigned fa)
{
    a[0] = lds_frag<TN, BLK * PT_PIECE>(fa);
    a[1] = lds_frag<TN, (8 + BLK) * PT_PIECE>(fa);
    a[2] = lds_frag<TN, (16 + BLK) * PT_PIECE>(fa);
}
template <bool TN> __device__ __forceinline__ void read_b(frag_t (&b)[2][3], unsigned fb)
{
    b[0][0] = lds_frag<TN, 0>(fb);        b[0][1] = lds_frag<TN, 8 * PT_PIECE>(fb);    b[0][2] = lds_frag<TN, 16 * PT_PIECE>(fb);
    b[1][0] = lds_frag<TN, PT_PIECE>(fb); b[1][1] = lds_frag<TN, 9 * PT_PIECE>(fb);    b[1][2] = lds_frag<TN, 17 * PT_PIECE>(fb);
}

// four f32 -> their three bf16 terms, packed as 4 bf16 per term (round to nearest; exact to 2^-27, see gemm_f32.hip)
__device__ __forceinline__ void split4_terms(const float (&a)[4], uint2 (&w)[3])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    float lo0 = a[0], hi0 = a[1], lo1 = a[2], hi1 = a[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        f2 p0 = {lo0, hi0}, p1 = {lo1, hi1};
        const uint32_t k0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(p0, b2));
        const uint32_t k1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, b2));
        w[t] = make_uint2(k0, k1);
        lo0 -= __uint_as_float(k0 << 16); hi0 -= __uint_as_float(k0 & 0xffff0000u);
        lo1 -= __uint_as_float(k1 << 16); hi1 -= __uint_as_float(k1 & 0xffff0000u);
    }
}

template <int BLK> __device__ __forceinline__ void read_a_pair(frag_t (&a)[3], unsigned fa)
{
    a[0] = lds_frag<false, BLK * PT_PIECE>(fa);
    a[1] = lds_frag<false, (9 + BLK) * PT_PIECE>(fa);
    a[2] = lds_frag<false, (18 + BLK) * PT_PIECE>(fa);
}

template <int DBG, bool TN, int TERMS = 6, bool NORM = false, bool PAIR = false> __global__ __launch_bounds__(512, 2) void gemm_planes_kernel(PlanesNTArgs p)
{
    static_assert(!(PAIR && TN), "the pair form is an NT form");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const unsigned long long rentry = (p.dbg & 8) ? __builtin_amdgcn_s_memrealtime() : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r32 = lane & 31, h = lane >> 5;

    long m0;
    int n0, ks0, nst;
    unsigned a_vo, b_vo;                                                // LDS-DMA: byte offset of this lane's 16 bytes
    long rbeg = 0;
    unsigned slab_z = blockIdx.z;
    if constexpr (!TN) {
        long mt = blockIdx.x / p.tiles_n;
        int nt = (int)(blockIdx.x % p.tiles_n);
        if (p.xcd_remap) {
            // workgroups go to the 8 XCDs round robin: the column tiles of one row tile run on ONE XCD, back to back
            const long j = blockIdx.x >> 3;
            mt = (j / p.tiles_n) * 8 + (blockIdx.x & 7);
            nt = (int)(j % p.tiles_n);
        }
        m0 = mt * PT_BM;
        n0 = nt * PT_BN;
        ks0 = blockIdx.y * (p.kchunk / PT_BK);
        nst = (min(p.K, (int)(blockIdx.y + 1) * p.kchunk) - (int)blockIdx.y * p.kchunk) / PT_BK;      // even, >= 2
        // wave w fills row block w of A and of B, all three planes (pieces w, 8 + w, ... 40 + w)
        const long ma = min(m0 + wave * 32 + r32, p.M - 1);             // rows past the end feed discarded outputs only
        const long aseg = ma / p.A.segv;
        a_vo = (unsigned)(32 * (aseg * p.A.seg_q + (ma - aseg * p.A.segv)) + 16 * h);
        const long nb = min((long)n0 + wave * 32 + r32, (long)p.N - 1);
        const long bseg = nb / p.B.segv;
        b_vo = (unsigned)(32 * (bseg * p.B.seg_q + (nb - bseg * p.B.segv)) + 16 * h);
    } else {
        // blockIdx.x: column tile (j), .y: row tile (i), .z: slab of rchunk reduction rows
        unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
        if (p.xcd_remap) {
            // workgroups go to the 8 XCDs round robin in dispatch order (x fastest): ALL tiles of a slab on ONE XCD, so that
            // the slab's rows of both operands are fetched into one L2 instead of eight (the column tiles of a conv weight
            // gradient are its taps: they read the same dU rows, and taps j, j + s the same signal rows)
            const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            const unsigned tiles = gridDim.x * gridDim.y, i = lin >> 3, tile = i % tiles;
            bz = (i / tiles) * 8 + (lin & 7);
            by = tile / gridDim.x;
            bx = tile % gridDim.x;
        }
        m0 = (long)by * PT_BM;
        n0 = (int)bx * PT_BN;
        ks0 = 0;
        rbeg = (long)bz * p.rchunk;
        slab_z = bz;
        if (rbeg >= p.R) return;                                        // (a slab slot past the last slab: the grid's z is padded to a multiple of 8)
        nst = (int)((min(p.R, rbeg + p.rchunk) - rbeg) / PT_BK);        // even, >= 2
        // wave w fills the 32 columns (two chunks) 32 w .. 32 w + 31 of both tiles; lane -> (4-row group, chunk, row, half)
        const int rowp = 4 * (lane >> 4) + ((lane >> 1) & 3), ch = (lane >> 3) & 1, half = lane & 1;
        auto vo = [&](const PlanesTNSide &o, long x0) {
            const int tap = o.tap0 + (int)(x0 / o.C), c = (int)(x0 % o.C) / 16 + 2 * wave + ch;
            const int smask = (1 << o.sshift) - 1;
            return (unsigned)(32 * ((long)((c << o.sshift) + (tap & smask)) * o.rts + (tap >> o.sshift) + rowp) + 16 * half);
        };
        a_vo = vo(p.TA, m0);
        b_vo = vo(p.TB, n0);
    }
    const unsigned lds0 = (unsigned)(unsigned long long)lds;
    constexpr bool noload = (DBG & 1) != 0, nomma = (DBG & 2) != 0;     // probes only
    // probes with wrong numbers and valid timing: what a lever could win AT MOST, before building it.  DBG & 4: the A pieces of the
    // odd stages are not requested (the L2 -> LDS traffic of a kernel that stages the A operand of a k = 2s convolution once per
    // tap pair); DBG & 8: a third of the fragment reads is skipped (the LDS read traffic of 128 x 128 wave tiles)
    constexpr bool half_a = (DBG & 4) != 0, fewer_reads = (DBG & 8) != 0;
    // stage t -> ring slot at byte offset `slot`: six pieces per wave, piece i of the stage requested by issue1<i>
    struct Src { unsigned dst; const char *ab, *bb; };
    auto stage_src = [&](int t, unsigned slot) {
        Src q;
        q.dst = lds0 + slot + wave * PT_PIECE;
        if constexpr (!TN) {
            q.ab = reinterpret_cast<const char *>(p.A.p) + 32 * chunk0(p.A, ks0 + t);
            q.bb = reinterpret_cast<const char *>(p.B.p) + 32 * chunk0(p.B, ks0 + t);
        } else {
            q.ab = reinterpret_cast<const char *>(p.TA.p) + 32 * (rbeg + (long)t * PT_BK);
            q.bb = reinterpret_cast<const char *>(p.TB.p) + 32 * (rbeg + (long)t * PT_BK);
        }
        return q;
    };
    auto issue1 = [&](const Src &q, int i) {
        const long pla = TN ? p.TA.plane : p.A.plane, plb = TN ? p.TB.plane : p.B.plane;
        if (i < 3) glds16(q.dst + i * 8 * PT_PIECE, a_vo, q.ab + 2 * i * pla);
        else glds16(q.dst + i * 8 * PT_PIECE, b_vo, q.bb + 2 * (i - 3) * plb);
    };
    auto issue = [&](int t, unsigned slot) {
        const Src q = stage_src(t, slot);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (!(half_a && (t & 1) && i < 3)) issue1(q, i);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if constexpr (PAIR) {
        // ---- pair form: double stage d = K steps 2d (tap j) and 2d + 1 (tap j + s) of a chunk; ring of two slots
        const int nds = nst >> 1;
        const unsigned fa_b = lds0 + lane * 16 + wr * 4 * PT_PIECE;                 // A fragments, tap j
        const unsigned fs_b = fa_b + (r32 == 31 ? PT_PIECE - 31 * 16 : 16);         // tap j + s: one row down (row 31 -> row 0 of the next block)
        const unsigned fj_b = lds0 + lane * 16 + PP_A + wc * 2 * PT_PIECE;          // B fragments of tap j; tap j + s: + 24 pieces
        // the ninth A block: every lane the row behind the tile (row m0 + 255 of this sample, + 1), 16 bytes of it
        const long m8 = m0 + PT_BM - 1, seg8 = m8 / p.A.segv;
        const unsigned a_vo8 = (unsigned)(32 * (seg8 * p.A.seg_q + (m8 - seg8 * p.A.segv) + 1) + 16 * h);
        struct SrcP { unsigned dst; const char *ab, *b0, *b1; };
        auto src_pair = [&](int d, unsigned slot) {
            SrcP q;
            q.dst = lds0 + slot + wave * PT_PIECE;
            q.ab = reinterpret_cast<const char *>(p.A.p) + 32 * chunk0(p.A, ks0 + 2 * d);
            q.b0 = reinterpret_cast<const char *>(p.B.p) + 32 * chunk0(p.B, ks0 + 2 * d);
            q.b1 = reinterpret_cast<const char *>(p.B.p) + 32 * chunk0(p.B, ks0 + 2 * d + 1);
            return q;
        };
        // piece i of a double stage as wave `wave` requests it: 0-2 A planes, 3-5 B of tap j, 6-8 B of tap j + s, 9 (waves 0-2) the ninth A block of plane `wave`
        auto issue_p = [&](const SrcP &q, int i) {
            if (i < 3) glds16(q.dst + i * 9 * PT_PIECE, a_vo, q.ab + 2 * i * p.A.plane);
            else if (i < 6) glds16(q.dst + PP_A + (i - 3) * 8 * PT_PIECE, b_vo, q.b0 + 2 * (i - 3) * p.B.plane);
            else if (i < 9) glds16(q.dst + PP_A + (24 + (i - 6) * 8) * PT_PIECE, b_vo, q.b1 + 2 * (i - 6) * p.B.plane);
            else if (wave < 3) glds16(lds0 + (q.dst - lds0 - wave * PT_PIECE) + (wave * 9 + 8) * PT_PIECE, a_vo8, q.ab + 2 * wave * p.A.plane);
        };
        {
            const SrcP q0 = src_pair(0, 0);
#pragma unroll
            for (int i = 0; i < 10; ++i) issue_p(q0, i);
        }
        if (nds > 1) {
            const SrcP q1 = src_pair(1, PP_DS);
#pragma unroll
            for (int i = 0; i < 10; ++i) issue_p(q1, i);
            if (wave < 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        frag_t b0[2][3], b1[2][3], ax[3], ay[3];
        read_b<false>(b0, fj_b);
        read_a_pair<0>(ax, fa_b);
        lds_wait9(ax, b0);
        unsigned slot = 0;
#define PT_SB __builtin_amdgcn_sched_barrier(0)
        // one half of a double stage: `fa` addresses its A fragments (tap j: fa_b + slot, tap j + s: fs_b + slot), bc its B
        // fragments; `fan` / `fbn` address the NEXT half's first A block and its B fragments (-> ax, bn).  last: the barrier that
        // ends the double stage sits in front of row block 3.
        // Requests: the slot of double stage d is free behind that barrier and has to be full at the next one, a whole double stage
        // later.  A piece costs the SIMD ~60 cycles of matrix-pipe time wherever it is issued, but ten in a row also stop the wave
        // that issues them for longer than its partner has MFMAs queued: they are spread -- the A pieces of double stage d + 2 between
        // the MFMAs of this row block 3, the B pieces two at a time behind row blocks 0, 1, 2 of the next half (d + 1's first one).
        auto half = [&](int d, unsigned fa, unsigned fan, unsigned fbn, frag_t (&bc)[2][3], frag_t (&bn)[2][3], const bool last) {
            // first half of double stage d >= 1: the B pieces of double stage d + 1 (its A pieces went out in front of this half)
            const bool breq = !last && d >= 1 && d + 1 < nds && !noload;
            const SrcP qb = src_pair(breq ? d + 1 : 0, slot == 0 ? PP_DS : 0);
            read_a_pair<1>(ay, fa);
            PT_SB;
            mma6<TERMS>(acc[0][0], acc[0][1], ax, bc, nomma);
            PT_SB;
            if (breq) { issue_p(qb, 3); issue_p(qb, 4); } PT_SB;
            lds_wait3(ay);
            read_a_pair<2>(ax, fa);
            PT_SB;
            mma6<TERMS>(acc[1][0], acc[1][1], ay, bc, nomma);
            PT_SB;
            if (breq) { issue_p(qb, 5); issue_p(qb, 6); } PT_SB;
            lds_wait3(ax);
            read_a_pair<3>(ay, fa);
            PT_SB;
            mma6<TERMS>(acc[2][0], acc[2][1], ax, bc, nomma);
            PT_SB;
            if (breq) { issue_p(qb, 7); issue_p(qb, 8); } PT_SB;
            lds_wait3(ay);
            bool req = false;
            SrcP q = src_pair(0, slot);
            if (last) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of double stage d + 1 (all it has in flight)
                __builtin_amdgcn_s_barrier();                         // d + 1 has landed for everyone; everyone has read all of d
                req = d + 2 < nds && !noload;
                if (req) q = src_pair(d + 2, slot);
            }
            if (!nomma && TERMS == 6) { mma2<0>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
            bn[0][0] = lds_frag<false, 0>(fbn); bn[0][1] = lds_frag<false, 8 * PT_PIECE>(fbn); bn[0][2] = lds_frag<false, 16 * PT_PIECE>(fbn); PT_SB;
            if (!nomma && TERMS == 6) { mma2<1>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
            bn[1][0] = lds_frag<false, PT_PIECE>(fbn); bn[1][1] = lds_frag<false, 9 * PT_PIECE>(fbn); bn[1][2] = lds_frag<false, 17 * PT_PIECE>(fbn); PT_SB;
            if (!nomma && TERMS == 6) { mma2<2>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
            read_a_pair<0>(ax, fan); PT_SB;
            if (req) { issue_p(q, 0); issue_p(q, 9); } PT_SB;
            if (!nomma) { mma2<3>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
            if (req) { issue_p(q, 1); } PT_SB;
            if (!nomma) { mma2<4>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
            if (req) { issue_p(q, 2); } PT_SB;
            if (!nomma) { mma2<5>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
            lds_wait9(ax, bn);
        };
#undef PT_SB
        unsigned long long c0 = 0, r0 = 0;
        if (p.dbg & 8) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
        for (int d = 0; d < nds; ++d) {
            const unsigned nslot = slot == 0 ? PP_DS : 0;
            half(d, fa_b + slot, fs_b + slot, fj_b + slot + 24 * PT_PIECE, b0, b1, false);
            half(d, fs_b + slot, fa_b + nslot, fj_b + nslot, b1, b0, true);
            slot = nslot;
        }
        if ((p.dbg & 8) && tid == 0) {
            p.stamps[blockIdx.x * 8 + 0] = c0; p.stamps[blockIdx.x * 8 + 1] = r0;
            p.stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
            p.stamps[blockIdx.x * 8 + 4] = rentry;
        }
    } else {
    // fragment addresses inside a stage: A piece (plane, block wr * 4 + i), B piece 24 + (plane, block wc * 2 + j)
    const unsigned flane = TN ? h * 512 + ((lane >> 4) & 1) * 128 + ((lane & 15) >> 2) * 32 + (lane & 3) * 8 : lane * 16;
    const unsigned fa0 = lds0 + flane + wr * 4 * PT_PIECE;
    const unsigned fb0 = lds0 + flane + (24 + wc * 2) * PT_PIECE;

    // ---- prologue: stages 0, 1, 2 requested; stage 0's fragments in registers
    issue(0, 0);
    issue(1, PT_STAGE);
    if (nst > 2) {
        issue(2, 2 * PT_STAGE);
        if constexpr (half_a) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
        if constexpr (half_a) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    frag_t b0[2][3], b1[2][3], ax[3], ay[3];
    read_b<TN>(b0, fb0);
    read_a<TN, 0>(ax, fa0);
    lds_wait9(ax, b0);

    // one stage; bc: B fragments of this stage, bn: of the next one; ax holds block 0 on entry (and on exit)
    //
    // (Round 4 measured two schedule variants here and removed them again: waves 4-7 requesting half a stage later than their SIMD
    //  partners, and a static s_setprio 1 for waves 4-7 -- no effect in any arm, profiles/r04_planes_levers.md: an LDS-DMA piece
    //  costs the SIMD's matrix pipe ~60 cycles whichever wave issues it.)
    unsigned slot = 0;                                                   // ring slot of stage t (byte offset)
#define PT_SB __builtin_amdgcn_sched_barrier(0)
    auto stage = [&](int t, frag_t (&bc)[2][3], frag_t (&bn)[2][3], const bool odd) {
        const unsigned fa = fa0 + slot;
        const unsigned nslot = slot == (PT_RING - 1) * PT_STAGE ? 0 : slot + PT_STAGE;
        read_a<TN, 1>(ay, fa);
        PT_SB;
        mma6<TERMS>(acc[0][0], acc[0][1], ax, bc, nomma);
        PT_SB;
        lds_wait3(ay);
        read_a<TN, 2>(ax, fa);
        PT_SB;
        mma6<TERMS>(acc[1][0], acc[1][1], ay, bc, nomma);
        PT_SB;
        lds_wait3(ax);
        if constexpr (!fewer_reads) read_a<TN, 3>(ay, fa);
        PT_SB;
        mma6<TERMS>(acc[2][0], acc[2][1], ax, bc, nomma);
        PT_SB;
        lds_wait3(ay);                          // every LDS read of stage t by this wave is done
        // (no branch may enclose an asm LDS read: hipcc would copy its destination registers at the join, before the wait)
        if (t + 2 < nst && !noload) {
            if (half_a && odd) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");       // (stage t + 2 stays in flight: odd like t, three pieces)
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // stage t + 1 has landed for everyone; everyone has read all of stage t
        // row block 3, with the requests for stage t + 3 and the reads of stage t + 1's first fragments BETWEEN its MFMAs:
        // the eight waves leave the barrier together, and an LDS-DMA piece holds a wave's issue for ~100 cycles
        const bool req = t + 3 < nst && !noload;
        const Src q = stage_src(req ? t + 3 : t, slot);
        const unsigned fbn = fb0 + nslot, fan = fa0 + nslot;
        if (!nomma && TERMS == 6) { mma2<0>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
        bn[0][0] = lds_frag<TN, 0>(fbn); bn[0][1] = lds_frag<TN, 8 * PT_PIECE>(fbn); bn[0][2] = lds_frag<TN, 16 * PT_PIECE>(fbn); PT_SB;
        if (!nomma && TERMS == 6) { mma2<1>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
        if constexpr (!fewer_reads) { bn[1][0] = lds_frag<TN, PT_PIECE>(fbn); bn[1][1] = lds_frag<TN, 9 * PT_PIECE>(fbn); bn[1][2] = lds_frag<TN, 17 * PT_PIECE>(fbn); } PT_SB;
        if (!nomma && TERMS == 6) { mma2<2>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
        read_a<TN, 0>(ax, fan); PT_SB;
        if (req && !(half_a && !odd)) { issue1(q, 0); issue1(q, 1); } PT_SB;      // (stage t + 3 is odd when t is even)
        if (!nomma) { mma2<3>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
        if (req) { if (!(half_a && !odd)) issue1(q, 2); issue1(q, 3); } PT_SB;
        if (!nomma) { mma2<4>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
        if (req) { issue1(q, 4); issue1(q, 5); } PT_SB;
        if (!nomma) { mma2<5>(acc[3][0], acc[3][1], ay, bc); } PT_SB;
        lds_wait9(ax, bn);
        slot = nslot;
    };
#undef PT_SB
    unsigned long long c0 = 0, r0 = 0;
    if (p.dbg & 8) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int t = 0; t < nst; t += 2) {
        stage(t, b0, b1, false);
        stage(t + 1, b1, b0, true);
    }
    if ((p.dbg & 8) && tid == 0) {
        p.stamps[blockIdx.x * 8 + 0] = c0; p.stamps[blockIdx.x * 8 + 1] = r0;
        p.stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        p.stamps[blockIdx.x * 8 + 4] = rentry;
    }

    }

    // ---- epilogue: acc[i][j][e] is C[m][n], m = m0 + wr*128 + i*32 + (e&3) + 8*(e>>2) + 4h, n = n0 + wc*64 + j*32 + r32.
    // The wave's 128 x 64 tile goes through LDS (its own 16 KiB, two passes of 64 rows) so that every lane stores 16 bytes
    // of a row: 16 store instructions per lane instead of 128.
    __builtin_amdgcn_s_barrier();                      // everyone has left the ring
    float *const stg = reinterpret_cast<float *>(lds) + wave * 4096;
    const int nw = n0 + wc * 64;                       // first column of the wave
    const int c4 = (lane & 15) * 4;
    float bias_v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_v[j] = (!TN && p.bias != nullptr && blockIdx.y == 0) ? p.bias[nw + j * 32 + r32] : 0.f;
    const int jrow = (p.map.enabled && p.map.col_rows > 0) ? nw / p.map.col_rows : 0;
    const long rv = p.map.enabled ? p.map.rv : (1L << 62);
    float *const out = TN ? p.slabs + (long)slab_z * p.M * p.N : p.slabs != nullptr ? p.slabs + (long)blockIdx.y * p.slab_rows * p.N : p.C;
    const long ldo = (TN || p.slabs != nullptr) ? p.N : p.ldc;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    stg[(i2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 64 + j * 32 + r32] = acc[pass * 2 + i2][j][e] + bias_v[j];
        // rows of this pass: m = mrow + 4 it + (lane >> 4); (group g, row t) follow by carrying (no division per row)
        const long mrow = m0 + wr * 128 + pass * 64 + (lane >> 4);
        long g = mrow / rv, t = mrow - g * rv;
        if constexpr (NORM) {
            // ---- ChannelNorm + ReLU + split of whole rows: a row's 256 columns are with the four waves wc = 0..3 of this wr
            float *const rsum = reinterpret_cast<float *>(lds + 8 * 16384);        // [wr][row of the pass][wc]
            float *const rss = rsum + 2 * 64 * 4;
            const int rloc = wr * 64 + (lane >> 4);                                // + 4 it
            float4 v[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) v[it] = *reinterpret_cast<const float4 *>(stg + (it * 4 + (lane >> 4)) * 64 + c4);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const float sm = group_sum<16>((v[it].x + v[it].y) + (v[it].z + v[it].w));
                if ((lane & 15) == 0) rsum[(rloc + 4 * it) * 4 + wc] = sm;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            float mean[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const float4 q4 = *reinterpret_cast<const float4 *>(rsum + (rloc + 4 * it) * 4);
                mean[it] = ((q4.x + q4.y) + (q4.z + q4.w)) * (1.f / PT_BN);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                float4 &q = v[it];
                q.x -= mean[it]; q.y -= mean[it]; q.z -= mean[it]; q.w -= mean[it];
                const float ss = group_sum<16>(fmaf(q.x, q.x, fmaf(q.y, q.y, fmaf(q.z, q.z, q.w * q.w))));
                if ((lane & 15) == 0) rss[(rloc + 4 * it) * 4 + wc] = ss;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const float4 gm = *reinterpret_cast<const float4 *>(p.norm.gamma + nw + c4);
            const float4 bt = *reinterpret_cast<const float4 *>(p.norm.beta + nw + c4);
            const int sh = p.norm.sshift;
            const long smask = (1L << sh) - 1;
            const long chunk_c = (long)(nw >> 4);                              // first of the wave's four 16-channel chunks
            // The next layer's planes leave through a wave-private LDS tile (the wave's staging area: its rows are in registers by
            // now), 32 rows at a time: [plane][chunk][row][32 bytes], chunks 33 rows apart (the four chunks of a row land in
            // different banks).  A lane then stores 16 bytes of (plane, chunk, row lane / 2): the rows of a chunk are consecutive
            // in memory (every s-th one), so a store instruction writes runs of 512 bytes and more -- as 8-byte stores straight
            // from the registers every instruction wrote sixteen 32-byte pieces, and the epilogue took 28 us per tile instead
            // of 18 without them (profiles/r04_planes_levers.md).
            uint2 *const ptile = reinterpret_cast<uint2 *>(stg);
            constexpr int PCH = 33 * 4;                                        // uint2 per (plane, chunk)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) {
                    const int it = hf * 8 + i8;
                    const float4 q4 = *reinterpret_cast<const float4 *>(rss + (rloc + 4 * it) * 4);
                    const float rstd = rsqrtf(((q4.x + q4.y) + (q4.z + q4.w)) * (1.f / (PT_BN - 1)) + p.norm.eps);
                    const long m = mrow + 4 * it;
                    const long crow = g * p.map.rows_out + t;             // (forward map: out_stride 1, out_off 0)
                    const float4 xh = make_float4(v[it].x * rstd, v[it].y * rstd, v[it].z * rstd, v[it].w * rstd);
                    if (m < p.M) {
                        *reinterpret_cast<float4 *>(out + crow * ldo + nw + c4) = xh;
                        if ((lane & 15) == 0 && wc == 0) p.norm.rstd[crow] = rstd;
                    }
                    const float y[4] = {fmaxf(fmaf(xh.x, gm.x, bt.x), 0.f), fmaxf(fmaf(xh.y, gm.y, bt.y), 0.f),
                                        fmaxf(fmaf(xh.z, gm.z, bt.z), 0.f), fmaxf(fmaf(xh.w, gm.w, bt.w), 0.f)};
                    uint2 w3[3];
                    split4_terms(y, w3);
                    if (i8 == 0) asm volatile("" ::: "memory");
                    const int prow = i8 * 4 + (lane >> 4), pch = (lane & 15) >> 2, pq = lane & 3;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) ptile[(pl * 4 + pch) * PCH + prow * 4 + pq] = w3[pl];
                    t += 4;
                    while (t >= rv) { t -= rv; ++g; }
                }
                // flush: row lane / 2 of the 32, half lane & 1; twelve (plane, chunk) blocks of one KiB
                const long mf = m0 + wr * 128 + pass * 64 + hf * 32 + (lane >> 1);
                const long gf = mf / rv, tf = mf - gf * rv;
                const long R = gf * p.norm.rows_next + p.norm.halo + tf;
                const long rowchunk = (R & smask) * p.norm.rts + (R >> sh);
                if (mf < p.M && !(p.dbg & 64)) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        const int pl = k >> 2, pch = k & 3;
                        const uint4 u = *reinterpret_cast<const uint4 *>(ptile + k * PCH + (lane >> 1) * 4 + 2 * (lane & 1));
                        const long chunk = (((chunk_c + pch) << sh)) * p.norm.rts + rowchunk;
                        *reinterpret_cast<uint4 *>(p.norm.p + pl * p.norm.plane + chunk * 16 + 8 * (lane & 1)) = u;
                    }
                }
                asm volatile("" ::: "memory");       // (the tile is the staging area under another type: keep the next writes behind these reads)
            }
            continue;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const float4 v = *reinterpret_cast<const float4 *>(stg + (it * 4 + (lane >> 4)) * 64 + c4);
            const long m = mrow + 4 * it;
            long crow = m, l = 0;
            if (p.map.enabled) {
                l = t * p.map.out_stride + p.map.out_off;
                crow = g * p.map.rows_out + l;
            }
            const bool ok = m < p.M && (!p.map.enabled || (l + jrow >= 0 && l + jrow < p.map.l_max));
            if (ok) *reinterpret_cast<float4 *>(out + crow * ldo + nw + c4) = v;
            t += 4;
            while (t >= rv) { t -= rv; ++g; }
        }
    }
    if (p.dbg & 8) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) p.stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    }
}

// x[rows][ld] (f32, `cols` columns used, cols % 16 == 0) -> chunked planes (layout: PlanesOperand).
// One workgroup per tile of 32 rows x 128 columns: the tile is read in whole rows (512 contiguous bytes), split, turned through LDS
// and written per (plane, chunk) as the 32 rows x 32 bytes that are CONTIGUOUS in the chunk-major layout (1 KiB runs for a plain
// matrix, 1 KiB / s for the input of a stride-s convolution).  (The first version let every thread store its own half chunk:
// neighbouring threads were rts * 32 bytes apart, and a 250 MB pass took 160 us instead of 50.)
constexpr int SP_ROWS = 32, SP_COLS = 128;
__global__ __launch_bounds__(256) void split_planes_kernel(const float *x, long ld, long rows, int cols, bf16_t *planes, long plane, int sshift,
                                                           long rts)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) uint2 tile[3][SP_COLS / 16][SP_ROWS + 1][4];  // [plane][chunk][row][4 x 8 bytes]; + 1: the chunks start in different banks
    const int col_tiles = (cols + SP_COLS - 1) / SP_COLS;
    const long n_tiles = ((rows + SP_ROWS - 1) / SP_ROWS) * col_tiles;
    const long smask = (1L << sshift) - 1;
    for (long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const long R0 = (tl / col_tiles) * SP_ROWS;
        const int c0 = (int)(tl % col_tiles) * SP_COLS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = q * 256 + threadIdx.x, r = idx >> 5, c4 = idx & 31;            // 32 float4 per tile row
            const long R = R0 + r;
            const int col = c0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R < rows && col < cols) v = *reinterpret_cast<const float4 *>(x + R * ld + col);
            float lo0 = v.x, hi0 = v.y, lo1 = v.z, hi1 = v.w;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                f2 p0 = {lo0, hi0}, p1 = {lo1, hi1};
                const uint32_t k0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(p0, b2));
                const uint32_t k1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, b2));
                tile[t][c4 >> 2][r][c4 & 3] = make_uint2(k0, k1);
                lo0 -= __uint_as_float(k0 << 16); hi0 -= __uint_as_float(k0 & 0xffff0000u);
                lo1 -= __uint_as_float(k1 << 16); hi1 -= __uint_as_float(k1 & 0xffff0000u);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int piece = q * 256 + threadIdx.x;                 // 3 planes x 8 chunks x 64 half rows
            const int pc = piece >> 6, j = piece & 63, r = j >> 1, half = j & 1;
            const int t = pc >> 3, ch = pc & 7;
            const long R = R0 + r;
            const int c = (c0 >> 4) + ch;
            if (R < rows && c * 16 < cols) {
                const long chunk = (((long)c << sshift) + (R & smask)) * rts + (R >> sshift);
                *reinterpret_cast<uint4 *>(planes + t * plane + chunk * 16 + half * 8) =
                    *reinterpret_cast<const uint4 *>(&tile[t][ch][r][2 * half]);
            }
        }
        __syncthreads();
    }
}

int split_planes(const float *x, long ld, long rows, int cols, bf16_t *planes, long plane, int sshift, long rts, hipStream_t st)
{
    CPC_REQUIRE(rows > 0 && cols > 0 && cols % 16 == 0 && ld % 4 == 0 && plane % 8 == 0 && sshift >= 0 && sshift < 8 &&
                    rts >= cdiv(rows, 1L << sshift) && plane >= (long)(cols / 16) * (rts << sshift) * 16 &&
                    reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(planes) % 16 == 0,
                "split_planes: bad arguments (rows=%ld cols=%d ld=%ld plane=%ld sshift=%d rts=%ld)", rows, cols, ld, plane, sshift, rts);
    const long tiles = cdiv(rows, SP_ROWS) * cdiv(cols, SP_COLS);
    const long blocks = std::min<long>(tiles, 16384);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ld, rows, cols, planes, plane, sshift, rts);
    CPC_CHECK_LAUNCH("split_planes_kernel");
    return CPC_OK;
}

__global__ void planes_tn_reduce_kernel(const float *slab, int S, int M, int N, float *C, long ldc, int conv_cin, int conv_k);

bool gemm_nt_planes_ok(long M, int N, int K)
{
    return N % PT_BN == 0 && K % (2 * PT_BK) == 0 && K >= 4 * PT_BK && M >= 1;
}

// few tiles and a K long enough: K is split over blockIdx.y so that every CU gets a workgroup (>= 16 stages per split)
static int nt_planes_splits(long M, int N, int K)
{
    const long tiles = cdiv(M, PT_BM) * (N / PT_BN);
    if (tiles > 160) return 1;
    int sp = (int)std::min<long>(256 / tiles, K / 256);
    while (sp > 1 && K % (32 * sp) != 0) --sp;
    return std::max(sp, 1);
}

size_t gemm_nt_planes_scratch_bytes(long M, int N, int K, long out_rows)
{
    const int sp = nt_planes_splits(M, N, K);
    return sp > 1 ? align_up((size_t)sp * out_rows * N * sizeof(float), 256) : 0;
}

static int side_of(const PlanesOperand &o, PlanesSide &s, const char *name)
{
    CPC_REQUIRE(o.p != nullptr && reinterpret_cast<uintptr_t>(o.p) % 32 == 0 && o.plane % 16 == 0 && o.plane > 0 &&
                    2 * o.plane < (1L << 31) && o.sshift >= 0 && o.sshift < 8 && o.kshift >= 0 && o.kshift <= 8 && o.rts > 0,
                "gemm_nt_planes: bad %s operand", name);
    s.p = o.p; s.plane = o.plane; s.kshift = o.kshift; s.sshift = o.sshift; s.rts = o.rts;
    s.segv = o.segv > 0 ? o.segv : 0x7fffffff; s.seg_q = o.seg_q;
    return CPC_OK;
}

bool gemm_nt_planes_norm_ok(long M, int N, int K)
{
    static const bool off = getenv("CPC_NO_NORM_FUSION") != nullptr;          // A/B switch
    return !off && gemm_mode() != 2 && gemm_nt_planes_ok(M, N, K) && N == PT_BN && nt_planes_splits(M, N, K) == 1;   // (mode 2: its own kernels)
}

// one instantiation: dynamic LDS attribute (once), launch
template <int DBG, bool TN, int TERMS, bool NORM, bool PAIR> static int launch_planes(dim3 grid, const PlanesNTArgs &a, hipStream_t st,
                                                                                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr)
{
    constexpr int bytes = PAIR ? PP_LDS : PT_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        CPC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<DBG, TN, TERMS, NORM, PAIR>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        attr_set = true;
    }
    if (ev_start != nullptr)       // (in-situ timing: the events of the dispatch itself, no barrier packets on the stream)
        hipExtLaunchKernelGGL((gemm_planes_kernel<DBG, TN, TERMS, NORM, PAIR>), grid, dim3(512), bytes, st, ev_start, ev_stop, 0, a);
    else
        hipLaunchKernelGGL((gemm_planes_kernel<DBG, TN, TERMS, NORM, PAIR>), grid, dim3(512), bytes, st, a);
    return CPC_OK;
}

int gemm_nt_planes(const PlanesOperand &A, const PlanesOperand &B, float *C, long ldc, const float *bias, long M, int N, int K,
                   const RowMap &map, hipStream_t st, const PlanesNormOut *norm, int *left_slabs)
{
    if (left_slabs != nullptr) *left_slabs = 0;
    CPC_REQUIRE(gemm_nt_planes_ok(M, N, K), "gemm_nt_planes: shape M=%ld N=%d K=%d not supported", M, N, K);
    CPC_REQUIRE(map.epi == EPI_NONE, "gemm_nt_planes: no fused elementwise epilogue in this kernel");
    CPC_REQUIRE(norm == nullptr || (gemm_nt_planes_norm_ok(M, N, K) && map.enabled && map.out_stride == 1 && map.out_off == 0 &&
                                    map.col_rows == 0 && norm->gamma && norm->beta && norm->rstd && norm->p && gemm_mode() != 2 &&
                                    reinterpret_cast<uintptr_t>(norm->p) % 8 == 0 && norm->plane % 4 == 0),
                "gemm_nt_planes: the fused norm needs N == 256, no K split and a forward row map");
    PlanesNTArgs a{};
    CPC_TRY(side_of(A, a.A, "A"));
    CPC_TRY(side_of(B, a.B, "B"));
    a.C = C; a.ldc = ldc; a.bias = bias; a.M = M; a.N = N; a.K = K; a.map = map;
    if (norm != nullptr) a.norm = *norm;
    a.tiles_m = (int)cdiv(M, PT_BM); a.tiles_n = N / PT_BN;
    static const bool no_remap = getenv("CPC_GEMM_NO_XCD") != nullptr;
    a.xcd_remap = (!no_remap && a.tiles_n > 1 && a.tiles_m % 8 == 0) ? 1 : 0;
    a.kchunk = K; a.slabs = nullptr; a.slab_rows = 0;
    // few tiles and a K long enough: split K over blockIdx.y (one workgroup per CU), partial products to slabs that a
    // second kernel adds in a fixed order -- when the caller lent the room (RowMap::splitk_scratch)
    int splits = 1;
    {
        const long out_rows = map.enabled ? cdiv(M, map.rv) * map.rows_out : M;
        const int sp = nt_planes_splits(M, N, K);
        if (norm == nullptr && sp > 1 && map.splitk_scratch != nullptr && (!map.enabled || map.col_rows == 0) &&
            reinterpret_cast<uintptr_t>(map.splitk_scratch) % 16 == 0 && (size_t)sp * out_rows * N * sizeof(float) <= map.splitk_bytes) {
            splits = sp;
            a.kchunk = K / sp;
            a.slabs = static_cast<float *>(map.splitk_scratch);
            a.slab_rows = out_rows;
        }
    }
    static const int dbg_env = getenv("CPC_PLANES_DBG") ? atoi(getenv("CPC_PLANES_DBG")) : 0;
    a.dbg = dbg_env;
    static unsigned long long *stamps = nullptr;
    if (a.dbg & 8) {
        if (stamps == nullptr) CPC_CHECK_HIP(hipMalloc(&stamps, 65536 * 8 * sizeof(unsigned long long)));
        a.stamps = stamps;
    }
    const long blocks = (long)a.tiles_m * a.tiles_n;
    // the pair form (A staged once per tap pair): an operand with tap pairs whose tiles do not straddle samples
    static const bool no_pair = getenv("CPC_PLANES_NO_PAIR") != nullptr;
    const bool pair = !no_pair && A.kshift >= 1 && a.A.segv % PT_BM == 0 && M % PT_BM == 0 && (a.kchunk / PT_BK) % 2 == 0;
    const dim3 grid((unsigned)blocks, (unsigned)splits);
    ProfScope prof(PROF_PLANES_NT, st, true);
    const hipEvent_t e0 = prof.start(), e1 = prof.stop();
    int rc = CPC_OK;
    const int mode3 = gemm_mode() == 2;
    const int sel = (a.dbg & 48) ? ((a.dbg & 48) >> 2) : (a.dbg & 3);                 // DBG template value of the probes
    if (pair) {
        if (sel == 1) rc = launch_planes<1, false, 6, false, true>(grid, a, st, e0, e1);
        else if (sel == 2) rc = launch_planes<2, false, 6, false, true>(grid, a, st, e0, e1);
        else if (norm != nullptr) rc = launch_planes<0, false, 6, true, true>(grid, a, st, e0, e1);
        else if (mode3) rc = launch_planes<0, false, 3, false, true>(grid, a, st, e0, e1);
        else rc = launch_planes<0, false, 6, false, true>(grid, a, st, e0, e1);
    } else {
        if (sel == 4) rc = launch_planes<4, false, 6, false, false>(grid, a, st, e0, e1);
        else if (sel == 8) rc = launch_planes<8, false, 6, false, false>(grid, a, st, e0, e1);
        else if (sel == 12) rc = launch_planes<12, false, 6, false, false>(grid, a, st, e0, e1);
        else if (sel == 1) rc = launch_planes<1, false, 6, false, false>(grid, a, st, e0, e1);
        else if (sel == 2) rc = launch_planes<2, false, 6, false, false>(grid, a, st, e0, e1);
        else if (norm != nullptr) rc = launch_planes<0, false, 6, true, false>(grid, a, st, e0, e1);
        else if (mode3) rc = launch_planes<0, false, 3, false, false>(grid, a, st, e0, e1);
        else rc = launch_planes<0, false, 6, false, false>(grid, a, st, e0, e1);
    }
    if (rc != CPC_OK) { prof.cancel(); return rc; }
    CPC_CHECK_LAUNCH("gemm_planes_kernel (nt)");
    if (splits > 1 && left_slabs != nullptr && ldc == N) {
        *left_slabs = splits;
    } else if (splits > 1) {
        const long total = a.slab_rows * N;
        hipLaunchKernelGGL(planes_tn_reduce_kernel, dim3((unsigned)std::min<long>(cdiv(total / 4, 256), 4096)), dim3(256), 0, st, a.slabs, splits,
                           (int)a.slab_rows, N, C, ldc, 0, 0);
        CPC_CHECK_LAUNCH("planes split-K reduce");
    }
    if (a.dbg & 8) {
        static unsigned long long host[65536 * 8];
        CPC_CHECK_HIP(hipStreamSynchronize(st));
        const long nb = std::min<long>(blocks, 65536);
        CPC_CHECK_HIP(hipMemcpy(host, stamps, nb * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double cyc = 0, real = 0, pro = 0, epi = 0;
        unsigned long long first = ~0ull, last = 0;
        for (long i = 0; i < nb; ++i) {
            cyc += (double)(host[i * 8 + 2] - host[i * 8]); real += (double)(host[i * 8 + 3] - host[i * 8 + 1]);
            pro += (double)(host[i * 8 + 1] - host[i * 8 + 4]); epi += (double)(host[i * 8 + 5] - host[i * 8 + 3]);
            first = std::min(first, host[i * 8 + 4]); last = std::max(last, host[i * 8 + 5]);
        }
        fprintf(stderr, "planes stamps: %ld tiles, loop %.0f cycles = %.2f us per tile, clock %.3f GHz; prologue %.2f us, epilogue %.2f us, kernel span %.1f us\n",
                nb, cyc / nb, real / nb * 0.01, cyc / real * 0.1, pro / nb * 0.01, epi / nb * 0.01, (double)(last - first) * 0.01);
    }
    return CPC_OK;
}

// out = sum over slabs, in slab order (bitwise reproducible); optional Conv1d weight re-layout (column jj*cin+ci -> [ci][jj])
// (four output elements per thread, 16-byte loads, four slabs in flight: the slabs are a few tens of MB that the producing
//  kernel has just written -- this is a latency-bound pass, not a bandwidth-bound one)
__global__ void planes_tn_reduce_kernel(const float *slab, int S, int M, int N, float *C, long ldc, int conv_cin, int conv_k)
{
    const long total4 = (long)M * N / 4;                       // N % 256 == 0
    const float4 *s4 = reinterpret_cast<const float4 *>(slab);
    for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int z = 0;
        for (; z + 4 <= S; z += 4) {                            // fixed order: slab 0, 1, 2, ... (bitwise reproducible)
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
        if (conv_cin > 0) {                                     // conv_cin % 4 == 0: the four elements share the tap
            const int jj = j / conv_cin, ci = j - jj * conv_cin;
            float *dst = C + (long)i * conv_cin * conv_k + (long)ci * conv_k + jj;
            dst[0] = acc.x; dst[conv_k] = acc.y; dst[2 * conv_k] = acc.z; dst[3 * conv_k] = acc.w;
        } else {
            float *dst = C + (long)i * ldc + j;
            dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w;
        }
    }
}

// slabs of `chunk` reduction rows: as many as give every CU one workgroup in ONE round.  *slots_out = the grid's z: the slab count
// padded to a multiple of 8, so that all tiles of a slab share an XCD (gemm_planes_kernel, xcd_remap) whatever the count -- the
// padding slots' workgroups leave at once.  (Rounds 2-5 looked for a chunk whose slab count WAS a multiple of 8: conv3's weight
// gradient then ran as 4 x 40 = 160 workgroups of 52 stages on 256 CUs, conv4's as 192 of 22, conv2's as 224 of 74; now 244 of
// 34, 232 of 18, 252 of 66.  CPC_PLANES_TN_OLD_SPLITS=1 keeps the old rule for A/B runs.)
static int tn_planes_splits(int M, int N, long R, long *chunk_out, int *slots_out = nullptr)
{
    const long tiles = (long)(M / PT_BM) * (N / PT_BN);
    const long Rp = cdiv(R, 32) * 32;
    long S = std::max<long>(1, (256 + tiles / 2) / tiles);            // one workgroup per CU, one round
    S = std::min(S, std::max<long>(1, Rp / 128));
    long chunk = cdiv(cdiv(Rp, S), 32) * 32;
    static const bool old_rule = getenv("CPC_PLANES_TN_OLD_SPLITS") != nullptr;
    if (old_rule) {
        if (S >= 8 && tiles > 1)
            for (long c = chunk; c <= chunk + 32 * 16; c += 32)
                if (cdiv(Rp, c) % 8 == 0) { chunk = c; break; }
        *chunk_out = chunk;
        if (slots_out != nullptr) *slots_out = (int)cdiv(Rp, chunk);
        return (int)cdiv(Rp, chunk);
    }
    const long slabs = cdiv(Rp, chunk);
    *chunk_out = chunk;
    if (slots_out != nullptr) *slots_out = (int)((slabs >= 8 && tiles > 1) ? cdiv(slabs, 8) * 8 : slabs);
    return (int)slabs;
}

bool gemm_tn_planes_ok(int M, int N, long R) { return M % PT_BM == 0 && N % PT_BN == 0 && R >= 64; }

size_t gemm_tn_planes_scratch_bytes(int M, int N, long R)
{
    long chunk;
    const int S = tn_planes_splits(M, N, R, &chunk);
    return align_up((size_t)S * M * N * sizeof(float), 256);
}

int gemm_tn_planes(const PlanesTNOperand &A, const PlanesTNOperand &B, float *C, long ldc, int M, int N, long R, void *scratch,
                   size_t scratch_bytes, int conv_cin, int conv_k, hipStream_t st, int *left_slabs)
{
    if (left_slabs != nullptr) *left_slabs = 0;
    CPC_REQUIRE(gemm_tn_planes_ok(M, N, R), "gemm_tn_planes: shape M=%d N=%d R=%ld not supported", M, N, R);
    auto bad = [](const PlanesTNOperand &o) {
        return o.p == nullptr || reinterpret_cast<uintptr_t>(o.p) % 32 != 0 || o.plane % 16 != 0 || o.plane <= 0 ||
               2 * o.plane >= (1L << 31) || o.sshift < 0 || o.sshift > 7 || o.rts <= 0 || o.C % 32 != 0 || o.C <= 0 || o.tap0 < 0;
    };
    CPC_REQUIRE(!bad(A) && !bad(B), "gemm_tn_planes: bad operand");
    long chunk;
    int slots = 0;
    const int S = tn_planes_splits(M, N, R, &chunk, &slots);
    if ((size_t)S * M * N * sizeof(float) > scratch_bytes) {
        set_error("gemm_tn_planes: scratch too small (%zu < %zu)", scratch_bytes, (size_t)S * M * N * sizeof(float));
        return CPC_ERR_WORKSPACE;
    }
    PlanesNTArgs a{};
    a.TA = PlanesTNSide{A.p, A.plane, A.sshift, A.rts, A.tap0, A.C};
    a.TB = PlanesTNSide{B.p, B.plane, B.sshift, B.rts, B.tap0, B.C};
    a.R = cdiv(R, 32) * 32;            // rows R .. of A are zero (caller), of B finite
    a.rchunk = chunk;
    a.M = M; a.N = N; a.K = 0; a.slabs = static_cast<float *>(scratch);
    a.dbg = 0;
    static const bool no_remap = getenv("CPC_GEMM_NO_XCD") != nullptr;
    a.xcd_remap = (!no_remap && slots % 8 == 0 && (M / PT_BM) * (N / PT_BN) > 1) ? 1 : 0;
    {
        ProfScope prof(PROF_PLANES_TN, st);
        const dim3 grid((unsigned)(N / PT_BN), (unsigned)(M / PT_BM), (unsigned)slots);
        const int rc = gemm_mode() == 2 ? launch_planes<0, true, 3, false, false>(grid, a, st) : launch_planes<0, true, 6, false, false>(grid, a, st);
        if (rc != CPC_OK) return rc;
    }
    CPC_CHECK_LAUNCH("gemm_planes_kernel (tn)");
    if (left_slabs != nullptr) {           // the caller sums the slabs later (planes_tn_reduce)
        *left_slabs = S;
        return CPC_OK;
    }
    return planes_tn_reduce(a.slabs, S, M, N, C, ldc, conv_cin, conv_k, st);
}

int planes_tn_reduce(const float *slabs, int S, int M, int N, float *C, long ldc, int conv_cin, int conv_k, hipStream_t st)
{
    const long total = (long)M * N;
    hipLaunchKernelGGL(planes_tn_reduce_kernel, dim3((unsigned)std::min<long>(cdiv(total / 4, 256), 2048)), dim3(256), 0, st, slabs, S, M, N, C,
                       ldc, conv_cin, conv_k);
    CPC_CHECK_LAUNCH("planes_tn_reduce_kernel");
    return CPC_OK;
}

}  // namespace cpc

// ------------------------------------------------------------------------------------------------
extern "C" int cpc_split_planes(const float *x, long ld, long rows, int cols, void *planes, long plane_stride, int stride_log2,
                                long rows_per_phase, cpc_stream_t stream)
{
    return cpc::split_planes(x, ld, rows, cols, static_cast<cpc::bf16_t *>(planes), plane_stride, stride_log2, rows_per_phase,
                             static_cast<hipStream_t>(stream));
}

extern "C" int cpc_gemm_nt_planes(const void *a_planes, long a_plane_stride, int a_taps_log2, int a_stride_log2,
                                  long a_rows_per_phase, int a_seg_rows, long a_seg_q, const void *b_planes, long b_plane_stride,
                                  float *C, long ldc, const float *bias, long M, int N, int K, cpc_stream_t stream)
{
    cpc::PlanesOperand A{static_cast<const cpc::bf16_t *>(a_planes), a_plane_stride, a_taps_log2, a_stride_log2, a_rows_per_phase,
                         a_seg_rows, a_seg_q};
    cpc::PlanesOperand B{static_cast<const cpc::bf16_t *>(b_planes), b_plane_stride, 0, 0, N, 0, 0};
    cpc::RowMap map{};
    return cpc::gemm_nt_planes(A, B, C, ldc, bias, M, N, K, map, static_cast<hipStream_t>(stream));
}

extern "C" size_t cpc_gemm_tn_planes_scratch_bytes(int M, int N, long R) { return cpc::gemm_tn_planes_scratch_bytes(M, N, R); }

extern "C" int cpc_gemm_tn_planes(const void *a_planes, long a_plane_stride, int a_stride_log2, long a_rows_per_phase, int a_tap,
                                  int a_channels, const void *b_planes, long b_plane_stride, int b_stride_log2,
                                  long b_rows_per_phase, int b_tap, int b_channels, float *C, long ldc, int M, int N, long R,
                                  void *scratch, size_t scratch_bytes, cpc_stream_t stream)
{
    cpc::PlanesTNOperand A{static_cast<const cpc::bf16_t *>(a_planes), a_plane_stride, a_stride_log2, a_rows_per_phase, a_tap, a_channels};
    cpc::PlanesTNOperand B{static_cast<const cpc::bf16_t *>(b_planes), b_plane_stride, b_stride_log2, b_rows_per_phase, b_tap, b_channels};
    return cpc::gemm_tn_planes(A, B, C, ldc, M, N, R, scratch, scratch_bytes, 0, 0, static_cast<hipStream_t>(stream));
}
