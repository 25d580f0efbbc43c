// Same-wave fillers: one wave per SIMD issues MFMAs with F independent VALU ops (or LDS ops) after each one.
// Compare with tools/mfma_overlap.hip (other-wave work does not overlap on this device).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return (s & 0x007f007fu) | 0x3f003f00u; }
__device__ inline bf16x8 frag(uint32_t &s) { uint4 v = {rnd(s), rnd(s), rnd(s), rnd(s)}; return __builtin_bit_cast(bf16x8, v); }

template <int F, int KIND> __global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = frag(s); b[i] = frag(s); }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane * 1e-3f + i;
    float4 v = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (KIND == 0) {
#pragma unroll
                        for (int f = 0; f < F; ++f) x[(r * 4 + i * 2 + j + f) & 7] = fmaf(x[(r * 4 + i * 2 + j + f) & 7], 1.0001f, 0.5f);
                    } else if (KIND == 1) {
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            const float4 t = *reinterpret_cast<const float4 *>(&lds[((lane + (r * 4 + i * 2 + j + f) * 64 + it) & 2047) * 4]);
                            v.x += t.x;
                        }
                    } else {
#pragma unroll
                        for (int f = 0; f < F; ++f)
                            *reinterpret_cast<float4 *>(&lds[((lane + (r * 4 + i * 2 + j + f) * 64) & 2047) * 4]) = make_float4(x[0], x[1], x[2], x[3]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (KIND == 0) __builtin_amdgcn_sched_group_barrier(0x002, F, 0);
                    else if (KIND == 1) { __builtin_amdgcn_sched_group_barrier(0x100, F, 0); __builtin_amdgcn_sched_group_barrier(0x002, F, 0); }
                    else __builtin_amdgcn_sched_group_barrier(0x200, F, 0);
                }
    }
    float sum = v.x;
    for (int i = 0; i < 8; ++i) sum += x[i];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = sum + lds[lane];
}

template <int F, int KIND> void run(float *out, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<F, KIND>), dim3(256), dim3(256), 0, 0, out, 20000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-14s %d per MFMA: %.2f ms\n", name, F, best);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0, 0>(out, "VALU"); run<2, 0>(out, "VALU"); run<4, 0>(out, "VALU"); run<6, 0>(out, "VALU"); run<8, 0>(out, "VALU");
    run<1, 1>(out, "ds_read_b128"); run<2, 1>(out, "ds_read_b128");
    run<1, 2>(out, "ds_write_b128");
    return 0;
}
