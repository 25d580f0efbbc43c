// Round 5: can the similarity kernel's candidate stream go STRAIGHT into the MFMA B operand (registers, no LDS-DMA) and overlap
// with the f32 MFMAs?  The LDS-DMA kernel of rounds 2-4 runs at MFMA time + stream time (profiles/r05_nce_ladder.txt): a
// global_load_lds instruction holds the SIMD's issue for ~100 cycles per KiB, whichever wave issues it.
//
// One persistent wave per slot; item = one (b,t): a P tile (16 rows x H floats, contiguous) + TILES candidate tiles of 16 gathered
// rows of an R-row table.  Lane (r, q) loads the 16-byte pieces [16 kk + 4 q .. +3] of ITS row (kk = 0 .. H/16-1) and feeds them to
// v_mfma_f32_16x16x4f32 unchanged (the K permutation is the same for A and B).  Software pipeline: the pieces of tile j + 1 are
// requested before tile j is multiplied.
//   hipcc -O3 --offload-arch=gfx950 tools/nce_stream_probe.hip -o /tmp/nce_stream_probe && /tmp/nce_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 256, KK = H / 16, TILES = 9, NBT = 7424;

// CH = chunks per candidate tile (1: a whole tile of 16 rows x 256 floats = 64 registers per lane in flight; 2: half tiles, 32).
// The chunk c + 1 is requested, THEN chunk c is multiplied (sched_barrier: hipcc otherwise sinks the loads between the MFMAs and
// waits for each at once).
template <int MFMA, int WPS, int CH> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void stream_kernel(const float *z, const float *P, const int *idx, float *out, int n_bt, unsigned row_mask)
{
    constexpr int CK = KK / CH;                       // float4 pieces per chunk and lane
    constexpr int NCH = TILES * CH;
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    float4 areg[KK], buf[2][CK];
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    for (int bt = blockIdx.x; bt < n_bt; bt += gridDim.x) {
        int rows[TILES];
#pragma unroll
        for (int j = 0; j < TILES; ++j) rows[j] = idx[(long)bt * TILES * 16 + j * 16 + r] & row_mask;
        const float *prow = P + ((long)bt * 16 + r) * H + 4 * q;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) areg[kk] = *reinterpret_cast<const float4 *>(prow + 16 * kk);
        {
            const float *zr = z + (long)rows[0] * H + 4 * q;
#pragma unroll
            for (int kk = 0; kk < CK; ++kk) buf[0][kk] = *reinterpret_cast<const float4 *>(zr + 16 * kk);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (MFMA != 2 && c + 1 < NCH) {
                const int jn = (c + 1) / CH, hn = (c + 1) % CH;
                const float *zn = z + (long)rows[jn] * H + 4 * q + hn * CK * 16;
#pragma unroll
                for (int kk = 0; kk < CK; ++kk) buf[(c + 1) & 1][kk] = *reinterpret_cast<const float4 *>(zn + 16 * kk);
            }
            if (MFMA == 2) {                         // (opaque to the optimiser: the nine tiles' identical chains must not be merged)
#pragma unroll
                for (int kk = 0; kk < CK; ++kk) asm volatile("" : "+v"(buf[0][kk].x), "+v"(buf[0][kk].y), "+v"(buf[0][kk].z), "+v"(buf[0][kk].w));
            }
            __builtin_amdgcn_sched_barrier(0);
            const int h = c % CH;
            if (MFMA) {
#pragma unroll
                for (int kk = 0; kk < CK; kk += 2) {
                    const float4 a0 = areg[h * CK + kk], a1 = areg[h * CK + kk + 1], b0 = buf[MFMA == 2 ? 0 : (c & 1)][kk], b1 = buf[MFMA == 2 ? 0 : (c & 1)][kk + 1];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < CK; ++kk) { acc0[kk & 3] += buf[c & 1][kk].x + buf[c & 1][kk].w; acc1[kk & 3] += areg[h * CK + kk].y; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (h == CH - 1) { tot += acc0 + acc1; acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1 = acc0; }
        }
    }
    if (tot[0] + tot[1] + tot[2] + tot[3] == 123.456f) out[blockIdx.x * 64 + lane] = tot[0];
}

template <int MFMA, int WPS, int CH> static float run(const float *z, const float *P, const int *idx, float *out, int grid, unsigned mask)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int trial = 0; trial < 5; ++trial) {
        hipEventRecord(a);
        hipLaunchKernelGGL((stream_kernel<MFMA, WPS, CH>), dim3(grid), dim3(64), 0, 0, z, P, idx, out, NBT, mask);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best * 1e3f;
}

int main()
{
    const int R = 8192;
    float *z, *P, *out; int *idx;
    hipMalloc(&z, (size_t)R * H * 4); hipMalloc(&P, (size_t)NBT * 16 * H * 4); hipMalloc(&out, 1 << 22);
    hipMalloc(&idx, (size_t)NBT * TILES * 16 * 4);
    std::vector<float> hz((size_t)R * H), hp((size_t)NBT * 16 * H);
    for (auto &v : hz) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hp) v = (float)rand() / RAND_MAX - 0.5f;
    std::vector<int> hi((size_t)NBT * TILES * 16);
    for (auto &v : hi) v = rand() % R;
    hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(P, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    int dev = 0, cus = 0;
    hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const double bytes = (double)NBT * (TILES + 1) * 16 * H * 4;
    const double flops = 2.0 * NBT * 12 * 129 * H;
    printf("CUs %d; %.2f GB through the CUs per launch; algorithmic similarity flops %.2f G (f32 MFMA peak 157.3 TF -> %.1f us)\n", cus, bytes / 1e9,
           flops / 1e9, flops / 157.3e12 * 1e6);
    struct { const char *name; unsigned mask; } tabs[] = {{"8 MiB table", 8191u}, {"2 MiB table", 2047u}, {"16 hot rows", 15u}};
    for (auto &tb : tabs) {
        for (int cfg = 0; cfg < 5; ++cfg) {
            float t_stream, t_both;
            int wpc, ch;
            if (cfg == 0) { wpc = 8; ch = 1; t_stream = run<0, 2, 1>(z, P, idx, out, cus * 8, tb.mask); t_both = run<1, 2, 1>(z, P, idx, out, cus * 8, tb.mask); }
            else if (cfg == 1) { wpc = 8; ch = 2; t_stream = run<0, 2, 2>(z, P, idx, out, cus * 8, tb.mask); t_both = run<1, 2, 2>(z, P, idx, out, cus * 8, tb.mask); }
            else if (cfg == 2) { wpc = 12; ch = 2; t_stream = run<0, 3, 2>(z, P, idx, out, cus * 12, tb.mask); t_both = run<1, 3, 2>(z, P, idx, out, cus * 12, tb.mask); }
            else if (cfg == 3) { wpc = 16; ch = 2; t_stream = run<0, 4, 2>(z, P, idx, out, cus * 16, tb.mask); t_both = run<1, 4, 2>(z, P, idx, out, cus * 16, tb.mask); }
            else { wpc = 16; ch = 4; t_stream = run<0, 4, 4>(z, P, idx, out, cus * 16, tb.mask); t_both = run<1, 4, 4>(z, P, idx, out, cus * 16, tb.mask); }
            float t_mfma = 0.f;
            if (cfg == 0) t_mfma = run<2, 2, 1>(z, P, idx, out, cus * 8, tb.mask);
            else if (cfg == 2) t_mfma = run<2, 3, 2>(z, P, idx, out, cus * 12, tb.mask);
            if (t_mfma > 0.f) printf("    MFMAs alone (operands loaded once per item): %6.1f us\n", t_mfma);
            printf("%-12s %2d waves/CU, %d chunk(s) per tile: stream alone %6.1f us (%5.1f GB/s per CU)   stream + MFMA %6.1f us  (frac of 157.3 TF: %.3f)\n",
                   tb.name, wpc, ch, t_stream, bytes / cus / t_stream / 1e3, t_both, flops / (t_both * 1e-6) / 157.3e12);
        }
    }
    return 0;
}
