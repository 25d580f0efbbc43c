// How the offset field of global_load_lds_dwordx4 and M0 combine on gfx950 (tools: csrc/ldsdma.h, glds16x8).
//   hipcc -O2 --offload-arch=gfx950 tools/scratch/ldsdma_offset_probe.hip -o /tmp/ldsdma_offset_probe && /tmp/ldsdma_offset_probe
// For LDS targets `base` and instruction offsets `off`, M0 = base - off (the compensation glds16x8 uses): where do the 1024
// bytes land, and which source bytes are they?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OFF> __global__ void probe(const float *src, float *out, unsigned base)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = -1.f;
    __syncthreads();
    const unsigned m0v = (unsigned)(unsigned long long)lds + base - OFF;
    const unsigned vo = threadIdx.x * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%4\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "s"(m0v), "v"(vo), "s"(src), "n"(OFF) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += 64) out[i] = lds[i];
}
template <int OFF> void run(const float *src, float *out, unsigned base)
{
    static float h[8192];
    hipLaunchKernelGGL(probe<OFF>, dim3(1), dim3(64), 32768, 0, src, out, base);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int first = -1, count = 0;
    for (int i = 0; i < 8192; ++i) if (h[i] >= 0.f) { if (first < 0) first = i; ++count; }
    printf("base %5u  offset %4d  (M0 = %6d): %3d floats written, first at byte %6d holding source byte %6.0f\n", base, OFF, (int)base - OFF, count,
           first * 4, first >= 0 ? h[first] * 4 : -1.f);
}
int main()
{
    float *src, *out;
    static float hs[16384];
    for (int i = 0; i < 16384; ++i) hs[i] = (float)i;
    hipMalloc(&src, sizeof(hs)); hipMalloc(&out, 8192 * 4);
    hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
    for (unsigned base : {0u, 1024u, 4224u, 8448u, 12672u}) { run<0>(src, out, base); run<512>(src, out, base); run<1024>(src, out, base); run<1536>(src, out, base); }
    return 0;
}
