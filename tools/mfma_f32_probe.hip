// Round 5: issue rate of v_mfma_f32_16x16x4_f32 as the similarity kernel uses it (chains of dependent accumulations), by number of
// independent accumulators per wave and waves per SIMD; in-kernel clock = d s_memtime / d s_memrealtime (100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_probe.hip -o /tmp/mfma_f32_probe && /tmp/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC> __global__ __launch_bounds__(64) void k(float *out, unsigned long long *stamps, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f + 1.f, b = 2.f - threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            asm volatile("" : "+v"(a), "+v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x < 4096) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC> static void run(int wps, float *out, unsigned long long *st)
{
    const int iters = 2048 / NACC, grid = 256 * 4 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int t = 0; t < 4; ++t) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC>), dim3(grid), dim3(64), 0, 0, out, st, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long h[2];
    (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    const double n_mfma = 16.0 * NACC * iters;                 // per wave
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    printf("%d accumulators, %d wave(s)/SIMD: %7.1f us; %.1f cycles per MFMA per SIMD; clock %.2f GHz; %.1f TFLOP/s\n", NACC, wps, best * 1e3,
           (double)h[0] / (n_mfma * wps), ghz, n_mfma * grid * 2048.0 / (best * 1e-3) / 1e12);
}

int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&st, 4096 * 16);
    for (int wps : {1, 2, 3, 4}) { run<1>(wps, out, st); run<2>(wps, out, st); run<4>(wps, out, st); }
    return 0;
}
