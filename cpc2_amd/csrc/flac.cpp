// Minimal native FLAC decoder for the window feeder (the image has no torchaudio / libFLAC / sox).
// The reference loads audio with torchaudio.load (/root/reference/cpc/dataset.py:411-437, feature_loader.py:343):
// float32 [channels, samples] scaled to [-1, 1).  This decoder covers the FLAC subset format (what LibriSpeech and
// the reference's cpc/test_data ship): constant / verbatim / fixed / LPC subframes, Rice and Rice2 residuals with
// escape partitions, all stereo decorrelation modes, 4..32 bits per sample.  Every decode is verified against the
// MD5 of the unencoded audio stored in STREAMINFO.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cpc2_hip.h"

namespace cpc { void set_error(const char *fmt, ...); }

namespace {

struct Md5 {
    uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
    uint64_t len = 0;
    uint8_t buf[64];
    size_t fill = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t *p)
    {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
            0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
            0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
            0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
            0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
            0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
            0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9,
                                  14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = p[4 * i] | (p[4 * i + 1] << 8) | (p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f;
            int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D;
            D = C; C = B;
            B = B + rol(A + f + K[i] + m[g], S[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t *p, size_t n)
    {
        len += n;
        while (n > 0) {
            const size_t take = std::min(n, sizeof(buf) - fill);
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void final(uint8_t out[16])
    {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
        update(l, 8);
        const uint32_t v[4] = {a, b, c, d};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
    }
};

struct BitReader {
    const uint8_t *p;
    size_t n, pos = 0;      // byte position
    int bit = 0;            // bits already consumed of p[pos]
    bool fail = false;
    BitReader(const uint8_t *d, size_t len) : p(d), n(len) {}
    uint32_t bits(int k)    // k <= 32
    {
        uint64_t v = 0;
        while (k > 0) {
            if (pos >= n) { fail = true; return 0; }
            const int avail = 8 - bit, take = k < avail ? k : avail;
            v = (v << take) | ((p[pos] >> (avail - take)) & ((1u << take) - 1));
            bit += take; k -= take;
            if (bit == 8) { bit = 0; ++pos; }
        }
        return (uint32_t)v;
    }
    int64_t sbits(int k)
    {
        if (k == 0) return 0;
        uint64_t v = 0;
        int rem = k;
        while (rem > 0) { const int t = rem > 32 ? 32 : rem; v = (v << t) | bits(t); rem -= t; }
        const uint64_t sign = 1ull << (k - 1);
        return (int64_t)((v ^ sign) - sign);
    }
    uint32_t unary()        // number of 0 bits before the next 1
    {
        uint32_t q = 0;
        for (;;) {
            if (pos >= n) { fail = true; return 0; }
            const uint8_t rest = (uint8_t)(p[pos] << bit);
            if (rest == 0) { q += 8 - bit; bit = 0; ++pos; continue; }
            int z = 0;
            while (!((rest << z) & 0x80)) ++z;
            q += z;
            bit += z + 1;
            if (bit >= 8) { bit -= 8; ++pos; }
            return q;
        }
    }
    void align() { if (bit) { bit = 0; ++pos; } }
};

struct Info { int rate = 0, channels = 0, bps = 0; uint64_t total = 0; uint8_t md5[16]; size_t audio_off = 0; };

bool read_file(const char *path, std::vector<uint8_t> &data)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    data.resize(sz > 0 ? (size_t)sz : 0);
    const size_t got = sz > 0 ? std::fread(data.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    return got == data.size() && !data.empty();
}

int parse_info(const std::vector<uint8_t> &d, Info &info)
{
    if (d.size() < 42 || std::memcmp(d.data(), "fLaC", 4) != 0) { cpc::set_error("flac: not a FLAC stream"); return CPC_ERR_INVALID; }
    size_t off = 4;
    bool last = false, have = false;
    while (!last) {
        if (off + 4 > d.size()) { cpc::set_error("flac: truncated metadata"); return CPC_ERR_INVALID; }
        last = d[off] & 0x80;
        const int type = d[off] & 0x7f;
        const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
        off += 4;
        if (off + len > d.size()) { cpc::set_error("flac: truncated metadata block"); return CPC_ERR_INVALID; }
        if (type == 0 && len >= 34) {
            const uint8_t *s = d.data() + off;
            info.rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            info.channels = ((s[12] >> 1) & 7) + 1;
            info.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            info.total = ((uint64_t)(s[13] & 15) << 32) | ((uint64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
            std::memcpy(info.md5, s + 18, 16);
            have = true;
        }
        off += len;
    }
    if (!have) { cpc::set_error("flac: no STREAMINFO"); return CPC_ERR_INVALID; }
    info.audio_off = off;
    return CPC_OK;
}

bool decode_residual(BitReader &br, int32_t *res, int blocksize, int order)
{
    const int method = br.bits(2);
    if (method > 1) return false;
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = br.bits(4);
    const int nparts = 1 << porder;
    int idx = 0;
    for (int pt = 0; pt < nparts; ++pt) {
        int count = (blocksize >> porder) - (pt == 0 ? order : 0);
        if (porder == 0) count = blocksize - order;
        if (count < 0) return false;
        const int k = br.bits(pbits);
        if (k == esc) {
            const int nb = br.bits(5);
            for (int i = 0; i < count; ++i) res[idx++] = (int32_t)br.sbits(nb);
        } else {
            for (int i = 0; i < count; ++i) {
                const uint32_t q = br.unary();
                const uint32_t u = (q << k) | (k ? br.bits(k) : 0);
                res[idx++] = (int32_t)((u >> 1) ^ (0u - (u & 1)));
            }
        }
        if (br.fail) return false;
    }
    return idx == blocksize - order;
}

// A valid stream keeps every sample of a subframe inside its bps-bit range (bps <= 33 with the side channel's extra bit);
// a corrupt one is rejected at the first sample that leaves it, so the 64-bit predictor arithmetic below cannot overflow:
// |coef| < 2^14, |sample| <= 2^32, order <= 32.
inline bool fits(int64_t v, int bps) { return v >= -(int64_t(1) << (bps - 1)) && v < (int64_t(1) << (bps - 1)); }

bool decode_subframe(BitReader &br, int64_t *out, int blocksize, int bps, std::vector<int32_t> &res)
{
    if (br.bits(1)) return false;
    const int type = br.bits(6);
    int wasted = 0;
    if (br.bits(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps <= 0) return false;
    if (type == 0) {
        const int64_t v = br.sbits(bps);
        for (int i = 0; i < blocksize; ++i) out[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; ++i) out[i] = br.sbits(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) out[i] = br.sbits(bps);
        res.resize(blocksize);
        if (!decode_residual(br, res.data(), blocksize, order)) return false;
        for (int i = order; i < blocksize; ++i) {
            const int64_t r = res[i - order];
            switch (order) {
            case 0: out[i] = r; break;
            case 1: out[i] = r + out[i - 1]; break;
            case 2: out[i] = r + 2 * out[i - 1] - out[i - 2]; break;
            case 3: out[i] = r + 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
            default: out[i] = r + 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
            }
            if (!fits(out[i], bps)) return false;
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) out[i] = br.sbits(bps);
        const int prec = br.bits(4) + 1;
        if (prec == 16) return false;
        const int shift = (int)br.sbits(5);
        if (shift < 0) return false;
        int64_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br.sbits(prec);
        res.resize(blocksize);
        if (!decode_residual(br, res.data(), blocksize, order)) return false;
        for (int i = order; i < blocksize; ++i) {
            int64_t acc = 0;
            for (int j = 0; j < order; ++j) acc += coef[j] * out[i - 1 - j];
            out[i] = res[i - order] + (acc >> shift);
            if (!fits(out[i], bps)) return false;
        }
    } else {
        return false;
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) out[i] *= (int64_t(1) << wasted);      // (wasted < 33: bps stayed positive)
    return !br.fail;
}

// decodes the whole stream into chan-major int32 (out[c*total + i]); returns number of samples per channel
int decode_stream(const std::vector<uint8_t> &d, const Info &info, std::vector<int32_t> &pcm, uint64_t &decoded)
{
    BitReader br(d.data(), d.size());
    br.pos = info.audio_off;
    const uint64_t cap = info.total ? info.total : 0;
    if (cap) pcm.assign((size_t)cap * info.channels, 0);
    decoded = 0;
    std::vector<int64_t> ch[8];
    std::vector<int32_t> res;
    static const int bs_table[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    while (br.pos + 2 <= d.size()) {
        if (d[br.pos] != 0xFF || (d[br.pos + 1] & 0xFE) != 0xF8) { cpc::set_error("flac: lost frame sync at byte %zu", br.pos); return CPC_ERR_INVALID; }
        br.bits(16);
        const int bs_code = br.bits(4), sr_code = br.bits(4), ch_code = br.bits(4), ss_code = br.bits(3);
        br.bits(1);
        int first = br.bits(8);                  // UTF-8 style frame / sample number: skip continuation bytes
        int extra = 0;
        while (first & 0x80) { first <<= 1; ++extra; }
        for (int i = 1; i < extra; ++i) br.bits(8);
        int blocksize = bs_table[bs_code];
        if (bs_code == 6) blocksize = br.bits(8) + 1;
        else if (bs_code == 7) blocksize = br.bits(16) + 1;
        if (sr_code == 12) br.bits(8);
        else if (sr_code == 13 || sr_code == 14) br.bits(16);
        br.bits(8);                              // CRC-8
        if (blocksize <= 0 || br.fail) { cpc::set_error("flac: bad frame header"); return CPC_ERR_INVALID; }
        static const int ss_table[8] = {0, 8, 12, 0, 16, 20, 24, 32};
        const int bps = ss_table[ss_code] ? ss_table[ss_code] : info.bps;
        const int nch = ch_code < 8 ? ch_code + 1 : 2;
        if (nch != info.channels) { cpc::set_error("flac: channel count changes mid-stream"); return CPC_ERR_INVALID; }
        for (int c = 0; c < nch; ++c) {
            ch[c].resize(blocksize);
            int b = bps;
            if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) b += 1;   // side channel
            if (!decode_subframe(br, ch[c].data(), blocksize, b, res)) { cpc::set_error("flac: corrupt subframe at byte %zu", br.pos); return CPC_ERR_INVALID; }
        }
        br.align();
        br.bits(16);                             // CRC-16
        if (ch_code == 8) for (int i = 0; i < blocksize; ++i) ch[1][i] = ch[0][i] - ch[1][i];
        else if (ch_code == 9) for (int i = 0; i < blocksize; ++i) ch[0][i] = ch[0][i] + ch[1][i];
        else if (ch_code == 10)
            for (int i = 0; i < blocksize; ++i) {
                const int64_t side = ch[1][i];
                const int64_t mid = ch[0][i] * 2 + (side & 1);
                ch[0][i] = (mid + side) >> 1;
                ch[1][i] = (mid - side) >> 1;
            }
        if (!cap) pcm.resize((size_t)(decoded + blocksize) * nch);      // unknown length: interleave-free growth not needed
        for (int c = 0; c < nch; ++c)
            for (int i = 0; i < blocksize; ++i) {
                const uint64_t s = decoded + i;
                if (cap && s >= cap) break;
                if (cap) pcm[(size_t)c * cap + s] = (int32_t)ch[c][i];
            }
        decoded += blocksize;
        if (cap && decoded >= cap) { decoded = cap; break; }
    }
    if (!cap) { cpc::set_error("flac: streams without a sample count in STREAMINFO are not supported"); return CPC_ERR_INVALID; }
    return CPC_OK;
}

}  // namespace

extern "C" int cpc_flac_info(const char *path, int *sample_rate, int *channels, int *bits_per_sample, long *total_samples)
{
    std::vector<uint8_t> d;
    if (path == nullptr || !read_file(path, d)) { cpc::set_error("flac: cannot read '%s'", path ? path : "(null)"); return CPC_ERR_INVALID; }
    Info info;
    const int st = parse_info(d, info);
    if (st != CPC_OK) return st;
    if (sample_rate) *sample_rate = info.rate;
    if (channels) *channels = info.channels;
    if (bits_per_sample) *bits_per_sample = info.bps;
    if (total_samples) *total_samples = (long)info.total;
    return CPC_OK;
}

// out_host: float32 [channels][total_samples] scaled by 2^-(bps-1) (what torchaudio.load returns)
extern "C" int cpc_flac_decode_f32(const char *path, float *out_host, long capacity_floats, int *md5_ok)
{
    std::vector<uint8_t> d;
    if (path == nullptr || out_host == nullptr || !read_file(path, d)) { cpc::set_error("flac: cannot read '%s'", path ? path : "(null)"); return CPC_ERR_INVALID; }
    Info info;
    int st = parse_info(d, info);
    if (st != CPC_OK) return st;
    if ((long)(info.total * info.channels) > capacity_floats) { cpc::set_error("flac: output buffer too small"); return CPC_ERR_WORKSPACE; }
    std::vector<int32_t> pcm;
    uint64_t decoded = 0;
    st = decode_stream(d, info, pcm, decoded);
    if (st != CPC_OK) return st;
    if (decoded != info.total) { cpc::set_error("flac: decoded %llu of %llu samples", (unsigned long long)decoded, (unsigned long long)info.total); return CPC_ERR_INVALID; }
    // MD5 of the interleaved little-endian PCM, as stored by the encoder
    const int bytes = (info.bps + 7) / 8;
    Md5 md5;
    std::vector<uint8_t> row((size_t)bytes * info.channels * 4096);
    for (uint64_t s0 = 0; s0 < info.total; s0 += 4096) {
        const uint64_t cnt = std::min<uint64_t>(4096, info.total - s0);
        size_t o = 0;
        for (uint64_t s = s0; s < s0 + cnt; ++s)
            for (int c = 0; c < info.channels; ++c) {
                const int32_t v = pcm[(size_t)c * info.total + s];
                for (int b = 0; b < bytes; ++b) row[o++] = (uint8_t)((uint32_t)v >> (8 * b));
            }
        md5.update(row.data(), o);
    }
    uint8_t digest[16];
    md5.final(digest);
    bool any = false;
    for (int i = 0; i < 16; ++i) any |= info.md5[i] != 0;
    const bool ok = !any || std::memcmp(digest, info.md5, 16) == 0;
    if (md5_ok) *md5_ok = ok ? 1 : 0;
    if (!ok) { cpc::set_error("flac: MD5 of the decoded audio does not match STREAMINFO ('%s')", path); return CPC_ERR_INVALID; }
    const float scale = 1.0f / (float)(1u << (info.bps - 1));
    for (size_t i = 0; i < (size_t)info.total * info.channels; ++i) out_host[i] = (float)pcm[i] * scale;
    return CPC_OK;
}
