// Calibration: what does a bare f32 MFMA loop deliver on THIS device (clock under load included)?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same for v_mfma_f32_16x16x4_f32 (the similarity kernel's instruction) and v_mfma_f32_4x4x1_16B_f32 (sixteen independent
// 4 x 4 x 1 blocks: K = 12 predictions would fill three 4-row groups with no 12 -> 16 padding -- IF it issues at the f32 rate)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE> __global__ __launch_bounds__(256) void k2(float *out, int iters, float a0, float b0)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SHAPE == 16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        }
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE> static void run2(float *out, const char *name, double flops_per_instr)
{
    const int blocks = 256 * 3, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k2<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * 4 * iters * 8;
        printf("%s rep %d: %.3f ms  %.1f TFLOP/s  (%.2f ns per instruction per SIMD-resident wave set: %.1f instructions/us/SIMD)\n", name, rep, ms,
               instr * flops_per_instr / ms / 1e9, ms * 1e6 / (instr / 1024.0), instr / 1024.0 / (ms * 1e3));
    }
}

int main()
{
    float *out;
    const int blocks = 256 * 3, iters = 20000;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 /*waves*/ * iters * 4 * (32.0 * 32 * 2 * 2);
        printf("rep %d: %.3f ms  %.1f TFLOP/s\n", rep, ms, flops / ms / 1e9);
    }
    run2<16>(out, "v_mfma_f32_16x16x4_f32    ", 2048.0);
    run2<4>(out, "v_mfma_f32_4x4x1_16B_f32  ", 512.0);
    return 0;
}
