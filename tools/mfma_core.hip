// What does the multiply core of a split GEMM reach by itself?  One workgroup (4 waves) per CU, LDS pre-filled with
// the three bf16 planes of a 256 x 32 A tile and a 128 x 32 B tile (two stages), each wave multiplies its 128 x 64
// sub tile: per K step 2 x (12 + 6 ds_read_b128, 48 MFMAs).  Variants: fragment reads up front or prefetched one
// slice ahead, with or without the per-K-step barrier.   hipcc -O3 --offload-arch=gfx950 tools/mfma_core.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int MI = 4, NJ = 2, BM = 256, BN = 128;
constexpr int PA = BM * 64, PB = BN * 64, STAGE = 3 * (PA + PB);
__device__ __forceinline__ int off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16(f32x4_t &dst, const float *ptr)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait2(f32x4_t &a, f32x4_t &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
__device__ __forceinline__ float4 f4(const f32x4_t &v) { return make_float4(v[0], v[1], v[2], v[3]); }
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ void split2(float a, float b, uint32_t &x0, uint32_t &x1, uint32_t &x2)
{
    x0 = pk(a, b); a -= __uint_as_float(x0 << 16); b -= __uint_as_float(x0 & 0xffff0000u);
    x1 = pk(a, b); a -= __uint_as_float(x1 << 16); b -= __uint_as_float(x1 & 0xffff0000u);
    x2 = pk(a, b);
}
__device__ __forceinline__ void split8_store(const float4 &u, const float4 &v, char *plane0, int plane_bytes, int o)
{
    uint4 w0, w1, w2;
    split2(u.x, u.y, w0.x, w1.x, w2.x); split2(u.z, u.w, w0.y, w1.y, w2.y);
    split2(v.x, v.y, w0.z, w1.z, w2.z); split2(v.z, v.w, w0.w, w1.w, w2.w);
    *reinterpret_cast<uint4 *>(plane0 + o) = w0;
    *reinterpret_cast<uint4 *>(plane0 + plane_bytes + o) = w1;
    *reinterpret_cast<uint4 *>(plane0 + 2 * plane_bytes + o) = w2;
}

// VARIANT 4: + split and LDS stores of the next tile from registers (slices 2..7); 5: + its global loads (slices 0, 1)
template <int VARIANT> __global__ __launch_bounds__(256, 1) void k(float *out, int ksteps, const float *src = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 2 * STAGE / 4; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = 0x3f803f80u ^ (i * 2654435761u & 0x007f007fu);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, h = lane >> 5;
    f32x16 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    for (int kt = 0; kt < ksteps; ++kt) {
        const char *As = lds + (kt & 1) * STAGE, *Bs = As + 3 * PA;
        if (VARIANT == 0 || VARIANT == 2) {                 // all fragments of a kk first
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fa[MI][3], fb[NJ][3];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int t = 0; t < 3; ++t) fa[i][t] = *reinterpret_cast<const bf16x8 *>(As + t * PA + off(wm * 128 + i * 32 + r32, 2 * kk + h));
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int t = 0; t < 3; ++t) fb[j][t] = *reinterpret_cast<const bf16x8 *>(Bs + t * PB + off(wn * 64 + j * 32 + r32, 2 * kk + h));
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[t]], fb[j][TB[t]], acc[i][j], 0, 0, 0);
            }
        } else {                                            // slices (kk, i), fragments of the next slice requested first
            bf16x8 fa[2][3], fb[2][NJ][3];
            auto read_a = [&](int i, int kk, bf16x8 (&f)[3]) {
#pragma unroll
                for (int t = 0; t < 3; ++t) f[t] = *reinterpret_cast<const bf16x8 *>(As + t * PA + off(wm * 128 + i * 32 + r32, 2 * kk + h));
            };
            auto read_b = [&](int kk, bf16x8 (&f)[NJ][3]) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int t = 0; t < 3; ++t) f[j][t] = *reinterpret_cast<const bf16x8 *>(Bs + t * PB + off(wn * 64 + j * 32 + r32, 2 * kk + h));
            };
            read_b(0, fb[0]);
            read_a(0, 0, fa[0]);
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int kk = sl >> 2, i = sl & 3;
                if (i < 3) read_a(i + 1, kk, fa[(sl + 1) & 1]);
                else if (kk == 0) { read_a(0, 1, fa[(sl + 1) & 1]); read_b(1, fb[1]); }
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sl & 1][TA[t]], fb[kk][j][TB[t]], acc[i][j], 0, 0, 0);
                if (VARIANT == 1 || VARIANT == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (VARIANT == 0 || VARIANT == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VARIANT> void run(float *out, const char *name)
{
    hipFuncSetAttribute((const void *)k<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ksteps = 4096;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<VARIANT>, dim3(256), dim3(256), 2 * STAGE, 0, out, ksteps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 256.0 * ksteps * (2.0 * BM * BN * 32) * 6;
    printf("%-44s %.2f ms  %.0f TFLOP/s (bf16)  = %.0f TF of f32 products\n", name, best, flops / best / 1e9, flops / best / 1e9 / 6);
}

template <int VARIANT, int NLD = 6, int NST = 6> __global__ __launch_bounds__(256, 1) void k2(float *out, int ksteps, const float *src, long lda)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 2 * STAGE / 4; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = 0x3f803f80u ^ (i * 2654435761u & 0x007f007fu);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, h = lane >> 5;
    const int lrow = tid >> 2, lchunk = tid & 3;
    f32x16 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int NU = MI + NJ;
    float4 rg[2][NU][2];
    for (int s2 = 0; s2 < 2; ++s2) for (int u = 0; u < NU; ++u) for (int c = 0; c < 2; ++c) rg[s2][u][c] = make_float4(tid * 1e-3f + u, 1.f + c, 2.f, 3.f + s2);
    // VARIANT 5: every workgroup streams its own rows (HBM); 6: all workgroups read the same 3 MB (L2 hits);
    // 7: each workgroup re-reads one 48 KB tile (L1/TCP hits)
    const float *base = src + ((long)(VARIANT == 5 ? blockIdx.x : 0) * 384 + lrow) * lda + lchunk * 8;
    auto step = [&](int kt, int stage, float4 (&ldr)[NU][2], float4 (&str)[NU][2]) {
        const char *As = lds + stage * STAGE, *Bs = As + 3 * PA;
        char *Aw = lds + (stage ^ 1) * STAGE, *Bw = Aw + 3 * PA;
        bf16x8 fa[2][3], fb[2][NJ][3];
        auto read_a = [&](int i, int kk, bf16x8 (&f)[3]) {
#pragma unroll
            for (int t = 0; t < 3; ++t) f[t] = *reinterpret_cast<const bf16x8 *>(As + t * PA + off(wm * 128 + i * 32 + r32, 2 * kk + h));
        };
        auto read_b = [&](int kk, bf16x8 (&f)[NJ][3]) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 3; ++t) f[j][t] = *reinterpret_cast<const bf16x8 *>(Bs + t * PB + off(wn * 64 + j * 32 + r32, 2 * kk + h));
        };
        read_b(0, fb[0]);
        read_a(0, 0, fa[0]);
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int kk = sl >> 2, i = sl & 3;
            if (i < 3) read_a(i + 1, kk, fa[(sl + 1) & 1]);
            else if (kk == 0) { read_a(0, 1, fa[(sl + 1) & 1]); read_b(1, fb[1]); }
            if (sl < 2) {
                if (VARIANT >= 5) {
#pragma unroll
                    for (int u = 0; u < NU; ++u)
                        if (u < NLD && (u < 3) == (sl == 0)) {
#pragma unroll
                            for (int c = 0; c < 2; ++c) ldr[u][c] = *reinterpret_cast<const float4 *>(base + (long)64 * u * lda + (VARIANT == 7 ? 0 : (kt + 2) * 32 % 2048) + 4 * c);
                        }
                }
            } else {
                const int u = sl - 2;
                if (u >= NST) {}
                else if (u < MI) split8_store(str[u][0], str[u][1], Aw, PA, off(lrow + 64 * u, lchunk));
                else split8_store(str[u][0], str[u][1], Bw, PB, off(lrow + 64 * (u - MI), lchunk));
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sl & 1][TA[t]], fb[kk][j][TB[t]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int kt = 0; kt < ksteps; kt += 2) {
        step(kt, 0, rg[0], rg[1]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        step(kt + 1, 1, rg[1], rg[0]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (VARIANT == 4) {                                     // keep the register tiles from being loop invariant
#pragma unroll
            for (int u = 0; u < NU; ++u) { rg[0][u][0].x += 1.f; rg[1][u][1].y += 1.f; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VARIANT, int NLD = 6, int NST = 6> __global__ __launch_bounds__(256, 1) void k3(float *out, int ksteps, const float *src, long lda)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 2 * STAGE / 4; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = 0x3f803f80u ^ (i * 2654435761u & 0x007f007fu);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, h = lane >> 5;
    const int lrow = tid >> 2, lchunk = tid & 3;
    f32x16 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int NU = MI + NJ;
    f32x4_t rg[2][NU][2];
    for (int s2 = 0; s2 < 2; ++s2) for (int u = 0; u < NU; ++u) for (int c = 0; c < 2; ++c) rg[s2][u][c] = (f32x4_t){tid * 1e-3f + u, 1.f + c, 2.f, 3.f + s2};
    // VARIANT 5: every workgroup streams its own rows (HBM); 6: all workgroups read the same 3 MB (L2 hits);
    // 7: each workgroup re-reads one 48 KB tile (L1/TCP hits)
    const float *base = src + ((long)(VARIANT == 5 ? blockIdx.x : 0) * 384 + lrow) * lda + lchunk * 8;
    auto step = [&](int kt, int stage, f32x4_t (&ldr)[NU][2], f32x4_t (&str)[NU][2]) {
        const char *As = lds + stage * STAGE, *Bs = As + 3 * PA;
        char *Aw = lds + (stage ^ 1) * STAGE, *Bw = Aw + 3 * PA;
        bf16x8 fa[2][3], fb[2][NJ][3];
        auto read_a = [&](int i, int kk, bf16x8 (&f)[3]) {
#pragma unroll
            for (int t = 0; t < 3; ++t) f[t] = *reinterpret_cast<const bf16x8 *>(As + t * PA + off(wm * 128 + i * 32 + r32, 2 * kk + h));
        };
        auto read_b = [&](int kk, bf16x8 (&f)[NJ][3]) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 3; ++t) f[j][t] = *reinterpret_cast<const bf16x8 *>(Bs + t * PB + off(wn * 64 + j * 32 + r32, 2 * kk + h));
        };
        read_b(0, fb[0]);
        read_a(0, 0, fa[0]);
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int kk = sl >> 2, i = sl & 3;
            if (i < 3) read_a(i + 1, kk, fa[(sl + 1) & 1]);
            else if (kk == 0) { read_a(0, 1, fa[(sl + 1) & 1]); read_b(1, fb[1]); }
            if (sl < 2) {
                if (VARIANT >= 5) {
#pragma unroll
                    for (int u = 0; u < NU; ++u)
                        if (u < NLD && (u < 3) == (sl == 0)) {
#pragma unroll
                            for (int c = 0; c < 2; ++c) gload16(ldr[u][c], base + (long)64 * u * lda + (kt + 2) * 32 % 2048 + 4 * c);
                        }
                }
            } else {
                const int u = sl - 2;
                if (u < NST) {
                    vm_wait2<2 * NLD>(str[u][0], str[u][1]);          // everything older than this step's loads has landed
                    if (u < MI) split8_store(f4(str[u][0]), f4(str[u][1]), Aw, PA, off(lrow + 64 * u, lchunk));
                    else split8_store(f4(str[u][0]), f4(str[u][1]), Bw, PB, off(lrow + 64 * (u - MI), lchunk));
                }
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sl & 1][TA[t]], fb[kk][j][TB[t]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int kt = 0; kt < ksteps; kt += 2) {
        step(kt, 0, rg[0], rg[1]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        step(kt + 1, 1, rg[1], rg[0]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VARIANT, int NLD = 6, int NST = 6> void run2(float *out, const float *src, const char *name)
{
    hipFuncSetAttribute((const void *)k2<VARIANT, NLD, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ksteps = 4096;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k2<VARIANT, NLD, NST>), dim3(256), dim3(256), 2 * STAGE, 0, out, ksteps, src, 2048L);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 256.0 * ksteps * (2.0 * BM * BN * 32) * 6;
    printf("%-44s %.2f ms  %.0f TFLOP/s (bf16)  = %.0f TF of f32 products\n", name, best, flops / best / 1e9, flops / best / 1e9 / 6);
}
template <int VARIANT, int NLD = 6, int NST = 6> void run3(float *out, const float *src, const char *name)
{
    hipFuncSetAttribute((const void *)k3<VARIANT, NLD, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ksteps = 4096;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k3<VARIANT, NLD, NST>), dim3(256), dim3(256), 2 * STAGE, 0, out, ksteps, src, 2048L);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 256.0 * ksteps * (2.0 * BM * BN * 32) * 6;
    printf("%-44s %.2f ms  %.0f TFLOP/s (bf16)  = %.0f TF of f32 products\n", name, best, flops / best / 1e9, flops / best / 1e9 / 6);
}

int main()
{
    float *out; hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0>(out, "reads up front, barrier per K step");
    run<2>(out, "reads up front, no barrier");
    run<1>(out, "sliced + prefetched reads, barrier");
    run<3>(out, "sliced + prefetched reads, no barrier");
    float *src; hipMalloc(&src, (size_t)256 * 384 * 2048 * 4 + (1 << 20)); hipMemset(src, 0, (size_t)256 * 384 * 2048 * 4 + (1 << 20));
    run2<4>(out, src, "sliced + split/stores of the next tile");
    run2<5>(out, src, "sliced + split/stores + global loads (HBM)");
    run2<6>(out, src, "sliced + split/stores + global loads (L2)");
    run2<7>(out, src, "sliced + split/stores + global loads (L1)");
    run2<6, 4, 4>(out, src, "4 of 6 units loaded + split per step (L2)");
    run2<6, 2, 2>(out, src, "2 of 6 units loaded + split per step (L2)");
    run2<5, 4, 4>(out, src, "4 of 6 units loaded + split per step (HBM)");
    run2<6, 6, 0>(out, src, "6 units loaded, none split (L2)");
    run3<6>(out, src, "asm loads + explicit vmcnt waits (L2)");
    run3<5>(out, src, "asm loads + explicit vmcnt waits (HBM)");
    return 0;
}
