// Does other work on a SIMD overlap with a wave that issues bf16 MFMAs back to back?  8 waves per workgroup, one per CU:
// waves 0-3 run an MFMA loop (one per SIMD), waves 4-7 a VALU / LDS-read / global-load loop.  Times each role alone
// and both together.  hipcc -O3 --offload-arch=gfx950 tools/mfma_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return (s & 0x007f007fu) | 0x3f003f00u; }
__device__ inline bf16x8 frag(uint32_t &s) { uint4 v = {rnd(s), rnd(s), rnd(s), rnd(s)}; return __builtin_bit_cast(bf16x8, v); }

// mode bit 0: MFMA waves work; bit 1: other waves work; kind: 0 VALU fma chains, 1 LDS reads, 2 global loads
__global__ __launch_bounds__(512) void k(float *out, const float *src, int iters, int mode, int kind)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 1e-3f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sum = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
            bf16x8 a[2], b[2];
            for (int i = 0; i < 2; ++i) { a[i] = frag(s); b[i] = frag(s); }
            f32x16 acc[2][2];
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        }
    } else if (mode & 2) {
        if (kind == 0) {
            float x[8];
            for (int i = 0; i < 8; ++i) x[i] = lane * 1e-3f + i;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 12; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);     // 96 VALU per iteration (vs 24 MFMA)
            }
            for (int i = 0; i < 8; ++i) sum += x[i];
        } else if (kind == 1) {
            float4 v = make_float4(0, 0, 0, 0);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 12; ++r) {
                    const float4 t = *reinterpret_cast<const float4 *>(&lds[((lane + r * 64 + it) & 2047) * 4]);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
            }
            sum = v.x + v.y + v.z + v.w;
        } else {
            float4 v = make_float4(0, 0, 0, 0);
            const float4 *g = reinterpret_cast<const float4 *>(src) + (size_t)blockIdx.x * 65536;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 t = g[(lane + (wave - 4) * 64 + r * 256 + it * 1024) & 65535];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
            }
            sum = v.x + v.y + v.z + v.w;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main()
{
    float *out, *src;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&src, (size_t)256 * 65536 * 16);
    hipMemset(src, 0, (size_t)256 * 65536 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[3] = {"VALU fma", "ds_read_b128", "global_load_dwordx4"};
    for (int kind = 0; kind < 3; ++kind)
        for (int mode = 1; mode <= 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, src, iters, mode, kind);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%-20s mode %d (%s): %.2f ms\n", names[kind], mode, mode == 1 ? "MFMA only" : mode == 2 ? "other only" : "both", best);
        }
    return 0;
}
