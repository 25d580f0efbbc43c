// Calibration: bare bf16 MFMA loops on pseudo-random operands, 32x32x16 vs 16x16x32 (same 64x64 output per wave),
// 1..3 waves per SIMD: what the matrix pipe delivers on THIS device with the clock it holds under load.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_bf16_shapes.hip -o /tmp/mfma_bf16 && /tmp/mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef FULLRANGE
// sign, 7 mantissa bits and 3 exponent bits random (values 2^-7 .. 2): what activations look like to the multipliers
__device__ inline uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; const uint32_t e = (s >> 9) & 0x03800380u; return (s & 0x007f007fu) | (0x3c003c00u + e) | ((s >> 3) & 0x80008000u); }
#else
__device__ inline uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return (s & 0x007f007fu) | 0x3f003f00u | ((s >> 3) & 0x80008000u); }
#endif
__device__ inline bf16x8 frag(uint32_t &s) { uint4 v = {rnd(s), rnd(s), rnd(s), rnd(s)}; return __builtin_bit_cast(bf16x8, v); }

template <int SHAPE> __global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *stamps)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frag(s); b[i] = frag(s); }
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * r], b[j + 2 * r], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) sum += acc[i][j][e];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

static unsigned long long *stamps;
template <int SHAPE> void run(float *out, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, stamps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    // per wave and iteration: 64x64 outputs x 32 k
    const double flops = (double)blocks * 4 * iters * (64.0 * 64 * 32 * 2);
    unsigned long long h[2];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("shape %s, %d waves/SIMD: %.2f ms  %.0f TFLOP/s  clock %.3f GHz\n", SHAPE == 32 ? "32x32x16" : "16x16x32", blocks_per_cu, best, flops / best / 1e9, (double)h[0] / (double)h[1] * 0.1);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
    hipMalloc(&stamps, 256 * 3 * 2 * sizeof(unsigned long long));
    for (int w = 1; w <= 3; ++w) { run<32>(out, w); run<16>(out, w); }
    return 0;
}
