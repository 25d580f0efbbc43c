// Does a produce -> consume pair of kernels run faster when the intermediate buffer is small enough to stay in the
// memory-side cache (256 MB Infinity Cache) and is reused chunk after chunk?  Total traffic fixed at 1 GiB written +
// 1 GiB read; the buffer is S MiB, the pair is launched 1024/S times on it.
//   hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void produce(float4 *buf, long n4, float v)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        buf[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
}
__global__ void consume(const float4 *buf, long n4, float *out)
{
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = buf[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}

int main()
{
    const long total = 1L << 30;
    float4 *buf; float *out;
    hipMalloc(&buf, total); hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mib : {1024, 512, 256, 192, 128, 96, 64, 32, 16}) {
        const long bytes = (long)mib << 20, n4 = bytes / 16;
        const int reps = (int)(total / bytes);
        float best = 1e9f;
        for (int trial = 0; trial < 4; ++trial) {
            hipEventRecord(a);
            for (int r = 0; r < reps; ++r) {
                produce<<<2048, 256>>>(buf, n4, (float)r);
                consume<<<2048, 256>>>(buf, n4, out);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("buffer %5d MiB x %3d chunks: %.3f ms for 1 GiB written + 1 GiB read  (%.2f TB/s)\n", mib, reps, best,
               2.0 * total / best / 1e9);
    }
    return 0;
}
