// Same-wave fillers: one wave per SIMD issues bf16 MFMAs; between them it places VALU ops, LDS reads (consumed a
// whole iteration later), LDS writes or global loads.  What does each cost on top of the bare MFMA stream?
// (Other-wave work does not overlap with an MFMA stream on this device: tools/mfma_overlap.hip.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return (s & 0x007f007fu) | 0x3f003f00u; }
__device__ inline bf16x8 frag(uint32_t &s) { uint4 v = {rnd(s), rnd(s), rnd(s), rnd(s)}; return __builtin_bit_cast(bf16x8, v); }

// per iteration: 24 MFMAs and NV VALU, NR ds_read_b128, NW ds_write_b128, NG global_load_dwordx4, spread evenly
template <int NV, int NR, int NW, int NG> __global__ __launch_bounds__(256) void k(float *out, const float4 *src, int iters)
{
    __shared__ float4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = frag(s); b[i] = frag(s); }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane * 1e-3f + i;
    float4 rd[NR > 0 ? NR : 1], rg[NG > 0 ? NG : 1];
    for (int i = 0; i < (NR > 0 ? NR : 1); ++i) rd[i] = make_float4(0, 0, 0, 0);
    for (int i = 0; i < (NG > 0 ? NG : 1); ++i) rg[i] = make_float4(0, 0, 0, 0);
    float keep = 0.f;
    const float4 *g = src + (size_t)blockIdx.x * 65536 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        // consume what was requested one iteration ago
        for (int i = 0; i < NR; ++i) keep += rd[i].x;
        for (int i = 0; i < NG; ++i) keep += rg[i].y;
        int nv = 0, nr = 0, nw = 0, ng = 0;
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[(m >> 1) & 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m >> 1) & 1], b[m & 1], acc[(m >> 1) & 1][m & 1], 0, 0, 0);
            while (nv * 24 < (m + 1) * NV) { x[nv & 7] = fmaf(x[nv & 7], 1.0001f, 0.5f); ++nv; }
            while (nr * 24 < (m + 1) * NR) { rd[nr] = lds[(lane + nr * 64 + (it & 7) * 8) & 2047]; ++nr; }
            while (nw * 24 < (m + 1) * NW) { lds[(lane + nw * 64) & 2047] = make_float4(x[0], x[1], x[2], x[3]); ++nw; }
            while (ng * 24 < (m + 1) * NG) { rg[ng] = g[((it * NG + ng) * 256) & 65535]; ++ng; }
        }
    }
    float sum = keep;
    for (int i = 0; i < 8; ++i) sum += x[i];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = sum + lds[lane].x;
}

template <int NV, int NR, int NW, int NG> void run(float *out, const float4 *src)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NR, NW, NG>), dim3(256), dim3(256), 0, 0, out, src, 20000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("per 24 MFMAs: %3d VALU %2d ds_read_b128 %2d ds_write_b128 %2d global_load_dwordx4 : %.2f ms\\n", NV, NR, NW, NG, best);
}

int main()
{
    float *out; float4 *src;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&src, (size_t)256 * 65536 * 16 + 4096 * 16);
    hipMemset(src, 0, (size_t)256 * 65536 * 16 + 4096 * 16);
    run<0, 0, 0, 0>(out, src);
    run<48, 0, 0, 0>(out, src);
    run<96, 0, 0, 0>(out, src);
    run<0, 6, 0, 0>(out, src);
    run<0, 12, 0, 0>(out, src);
    run<0, 0, 3, 0>(out, src);
    run<0, 0, 6, 0>(out, src);
    run<0, 0, 0, 3>(out, src);
    run<0, 0, 0, 6>(out, src);
    run<60, 9, 5, 3>(out, src);
    return 0;
}
