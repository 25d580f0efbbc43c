// CPU-only sanitizer driver for the host-side C++ of the C ABI (round-2 review, item 8): csrc/flac.cpp parses untrusted files,
// csrc/negidx.cpp runs a worker thread.  Built by tests/test_host_sanitizers.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all  this file + the two sources
// and run on the reference's FLAC fixtures: as they are, truncated at many lengths, and with seeded byte flips.  Any
// out-of-bounds access, use-after-free, signed overflow, misaligned access or data race on the sampler's hand-over makes the
// process exit non-zero with the sanitizer's report.  No GPU call is made (the device-upload entry point is not exercised).
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cpc2_hip.h"

namespace cpc {
static char g_err[512];
void set_error(const char *fmt, ...)       // the library's own lives in rowops.hip (device code); same contract
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cpc

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> d;
    FILE *f = fopen(path, "rb");
    if (!f) return d;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    d.resize((size_t)n);
    if (fread(d.data(), 1, (size_t)n, f) != (size_t)n) d.clear();
    fclose(f);
    return d;
}

static void spit(const std::string &path, const uint8_t *p, size_t n)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror("fopen"); exit(3); }
    fwrite(p, 1, n, f);
    fclose(f);
}

static int decode(const char *path, long *samples_out, int *md5_ok_out)
{
    int sr = 0, ch = 0, bps = 0, md5 = -1;
    long total = 0;
    int st = cpc_flac_info(path, &sr, &ch, &bps, &total);
    if (st != CPC_OK) return st;
    if (total < 0 || ch < 1 || ch > 8 || total > (1L << 28)) return -100;       // a mutated header may claim anything
    // exact capacity: a decoder that writes one float too many trips the sanitizer's red zone
    std::vector<float> out((size_t)total * ch);
    st = cpc_flac_decode_f32(path, out.data(), (long)out.size(), &md5);
    if (samples_out) *samples_out = total;
    if (md5_ok_out) *md5_ok_out = md5;
    return st;
}

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <scratch dir> <file.flac>...\n", argv[0]); return 2; }
    const std::string tmp = std::string(argv[1]) + "/mutant.flac";
    long ok = 0, rejected = 0, mutants = 0;
    for (int a = 2; a < argc; ++a) {
        long samples = 0;
        int md5 = -1;
        if (decode(argv[a], &samples, &md5) != CPC_OK || md5 != 1 || samples <= 0) {
            fprintf(stderr, "%s: the intact fixture did not decode with a matching MD5 (%s)\n", argv[a], cpc::g_err);
            return 1;
        }
        ++ok;
        const std::vector<uint8_t> data = slurp(argv[a]);
        if (data.empty()) return 3;
        // truncated: inside the metadata, inside the first frame header, inside frames, one byte short
        const size_t cuts[] = {0, 3, 4, 7, 20, 41, 42, 50, 100, 4096, data.size() / 3, data.size() / 2, data.size() - 1};
        for (size_t c : cuts) {
            if (c >= data.size()) continue;
            spit(tmp, data.data(), c);
            ++mutants;
            if (decode(tmp.c_str(), nullptr, nullptr) != CPC_OK) ++rejected;
        }
        // seeded byte flips: headers (first 64 bytes) and the frame stream
        uint32_t seed = 12345u + (uint32_t)a;
        for (int m = 0; m < 48; ++m) {
            std::vector<uint8_t> mut = data;
            const int flips = 1 + (int)(lcg(seed) % 4);
            for (int k = 0; k < flips; ++k) {
                const size_t pos = (m % 3 == 0) ? lcg(seed) % 64 : lcg(seed) % mut.size();
                mut[pos] ^= (uint8_t)(1u << (lcg(seed) % 8));
            }
            spit(tmp, mut.data(), mut.size());
            ++mutants;
            if (decode(tmp.c_str(), nullptr, nullptr) != CPC_OK) ++rejected;
        }
    }
    remove(tmp.c_str());

    // the sampler: synchronous and worker-thread draws must give one and the same stream, whatever their interleaving
    cpc_mt19937 *g1 = cpc_mt_create(1234), *g2 = cpc_mt_create(1234);
    if (!g1 || !g2) return 3;
    const int b = 3, T = 32, W = 20, nneg = 7;
    const size_t n = (size_t)b * W * nneg;
    std::vector<int32_t> e1(n), e2(n);
    std::vector<int64_t> bi(n), si(n);
    std::vector<uint32_t> raw1(2 * n), raw2(2 * n);
    for (int round = 0; round < 6; ++round) {
        if (cpc_negidx_sample_host(g1, b, T, W, nneg, round & 1, e1.data(), bi.data(), si.data()) != CPC_OK) return 1;
        if (cpc_negidx_sample_host_async(g2, b, T, W, nneg, round & 1, e2.data()) != CPC_OK) return 1;
        if (cpc_negidx_wait(g2) != CPC_OK) return 1;
        if (memcmp(e1.data(), e2.data(), n * sizeof(int32_t)) != 0) { fprintf(stderr, "async sample differs from the synchronous one\n"); return 1; }
        for (size_t i = 0; i < n; ++i)
            if (e1[i] < 0 || e1[i] >= b * T || bi[i] < 0 || bi[i] >= b || si[i] < 1 || si[i] >= T) { fprintf(stderr, "index out of range\n"); return 1; }
        if (cpc_mt_draw_host(g1, raw1.data(), 2 * n) != CPC_OK) return 1;
        if (cpc_mt_draw_host_async(g2, raw2.data(), 2 * n) != CPC_OK) return 1;
        // a second request while the first is in flight must wait for it, not race it
        if (cpc_mt_draw_host(g2, raw2.data(), 0) != CPC_OK) return 1;
        if (memcmp(raw1.data(), raw2.data(), 2 * n * sizeof(uint32_t)) != 0) { fprintf(stderr, "async draw differs\n"); return 1; }
    }
    uint32_t st[624];
    int left = 0, next = 0;
    if (cpc_mt_get_state(g1, st, &left, &next) != CPC_OK || cpc_mt_set_state(g2, st, left, next) != CPC_OK) return 1;
    if (cpc_mt_set_state(g2, st, 0, 0) == CPC_OK || cpc_mt_set_state(g2, st, 700, 3) == CPC_OK) { fprintf(stderr, "bad state accepted\n"); return 1; }
    if (cpc_negidx_sample_host(g1, 0, T, W, nneg, 0, e1.data(), nullptr, nullptr) == CPC_OK) return 1;     // bad arguments are refused
    if (cpc_negidx_sample_host(g1, b, T, T + 1, nneg, 0, e1.data(), nullptr, nullptr) == CPC_OK) return 1;
    cpc_mt_draw_host_async(g2, raw2.data(), 2 * n);          // destroyed with a draw in flight: must join, not leak or race
    cpc_mt_destroy(g1);
    cpc_mt_destroy(g2);
    printf("host sanitizer driver ok: %ld fixtures decoded, %ld mutants (%ld rejected, the rest decoded to something), sampler paths exercised\n",
           ok, mutants, rejected);
    return 0;
}
