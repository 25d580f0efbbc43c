// Which HSA queue does a HIP stream get, and does a CU-masked stream overlap with the null stream?
// hipcc --offload-arch=gfx950 -O2 -o stream_queue_probe stream_queue_probe.hip ; rocprofv3 --kernel-trace -- ./stream_queue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long cycles, int *out) { long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (out) out[0] = 1; }
__global__ void tag_a(int *o) { if (o) o[0] = 1; }   // distinct names so that the trace tells the streams apart
__global__ void tag_b(int *o) { if (o) o[0] = 1; }
__global__ void tag_c(int *o) { if (o) o[0] = 1; }
__global__ void tag_d(int *o) { if (o) o[0] = 1; }
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int words = (p.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> mask(words, 0xffffffffu);
    printf("CUs %d mask words %d\n", p.multiProcessorCount, words);
    std::vector<hipStream_t> plain(10);
    for (auto &s : plain) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t m1, m2;
    CK(hipExtStreamCreateWithCUMask(&m1, words, mask.data()));
    CK(hipExtStreamCreateWithCUMask(&m2, words, mask.data()));
    unsigned f = 99; CK(hipStreamGetFlags(m1, &f)); printf("masked stream flags %u (hipStreamNonBlocking = %u)\n", f, hipStreamNonBlocking);
    int pr = 99; CK(hipStreamGetPriority(m1, &pr)); printf("masked stream priority %d\n", pr);
    // one tagged kernel per stream: the trace's Queue_Id column answers the mapping
    hipLaunchKernelGGL(tag_a, dim3(1), dim3(64), 0, 0, nullptr);
    for (auto &s : plain) hipLaunchKernelGGL(tag_b, dim3(1), dim3(64), 0, s, nullptr);
    hipLaunchKernelGGL(tag_c, dim3(1), dim3(64), 0, m1, nullptr);
    hipLaunchKernelGGL(tag_d, dim3(1), dim3(64), 0, m2, nullptr);
    CK(hipDeviceSynchronize());
    // overlap: 20 ms spin on the null stream, then a tiny kernel on the masked stream; when does the tiny one finish?
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); printf("priority range: least %d greatest %d\n", lo, hi);
    hipStream_t ph, pl; CK(hipStreamCreateWithPriority(&ph, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&pl, hipStreamNonBlocking, lo));
    hipLaunchKernelGGL(tag_d, dim3(1), dim3(64), 0, ph, nullptr);
    hipLaunchKernelGGL(tag_d, dim3(2), dim3(64), 0, pl, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<hipStream_t> all = {m1, ph, pl};
    for (auto s : plain) all.push_back(s);
    for (size_t which = 0; which < all.size(); ++which) {
        hipStream_t other = all[which];
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 2000000LL, nullptr);
        CK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(tag_c, dim3(1), dim3(64 + which), 0, other, nullptr);       // (block size = 64 + index: readable in the trace)
        CK(hipEventRecord(e2, other));
        CK(hipEventSynchronize(e2));
        hipError_t q = hipEventQuery(e1);
        printf("%s %zu: tiny kernel done while the null stream's spin is %s\n", which == 0 ? "masked" : which == 1 ? "high priority" : which == 2 ? "low priority" : "plain non-blocking", which,
               q == hipErrorNotReady ? "STILL RUNNING (overlaps)" : "already over (serialised)");
        CK(hipDeviceSynchronize());
    }
    return 0;
}
