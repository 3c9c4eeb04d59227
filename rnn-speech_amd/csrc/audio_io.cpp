// Host-side audio file decoding for the data pipeline: RIFF/WAVE, FLAC and NIST SPHERE -> mono float32.
//
// Replaces the decode half of `librosa.load(file)` in the reference (util/audioprocessor.py:49; librosa hands
// the file to libsndfile / audioread, neither of which is in this image): LibriSpeech ships FLAC, TED-LIUM
// SPHERE, Vystadial / Shtooka WAV or FLAC.  Like librosa.load(mono=True) the channels are averaged and integer
// PCM is scaled by 2^-(bits-1).  Resampling to the reference's 22,050 Hz is a separate step (frontend).
//
// The FLAC decoder follows the format specification (RFC 9639): all subframe types (constant, verbatim,
// fixed 0-4, LPC 1-32), both Rice codings with escape partitions, wasted bits, the three stereo
// decorrelations, variable block sizes; frame CRC-16 is verified on every frame and the STREAMINFO MD5 of the
// decoded samples on request -- the stream carries its own end-to-end check, which matters because no
// third-party FLAC codec exists in this image to cross-check against ("parity unpinned" otherwise).
#include <stdint.h>
#include <algorithm>
#include <exception>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/amdspeech.h"

namespace amdspeech {
void set_error(const char* fmt, ...);
}
using amdspeech::set_error;

namespace {

struct Pcm {                       // decoded file: interleaved integer or float samples as double-free int64/float
    int rate = 0, channels = 0, bits = 0;
    bool is_float = false;
    std::vector<int32_t> ints;     // interleaved, when !is_float
    std::vector<float> floats;     // interleaved, when is_float
    long frames = 0;
};

bool read_file(const char* path, std::vector<uint8_t>& buf) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == buf.size();
}

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// ------------------------------------------------------------------------------------------ RIFF/WAVE
int decode_wav(const std::vector<uint8_t>& b, Pcm& out, bool header_only) {
    if (b.size() < 12 || memcmp(b.data() + 8, "WAVE", 4) != 0) { set_error("wav: not a RIFF/WAVE file"); return AMDSPEECH_EINVAL; }
    size_t pos = 12;
    int fmt_tag = 0, block_align = 0;
    bool have_fmt = false;
    while (pos + 8 <= b.size()) {
        const uint8_t* ck = b.data() + pos;
        size_t len = le32(ck + 4);
        const uint8_t* body = ck + 8;
        if (pos + 8 + len > b.size()) len = b.size() - pos - 8;          // truncated / streamed files
        if (!memcmp(ck, "fmt ", 4) && len >= 16) {
            fmt_tag = le16(body);
            out.channels = le16(body + 2);
            out.rate = (int)le32(body + 4);
            block_align = le16(body + 12);
            out.bits = le16(body + 14);
            if (fmt_tag == 0xFFFE && len >= 26) fmt_tag = le16(body + 24);   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID
            have_fmt = true;
        } else if (!memcmp(ck, "data", 4)) {
            if (!have_fmt || out.channels <= 0 || out.bits <= 0) { set_error("wav: data chunk before a valid fmt chunk"); return AMDSPEECH_EINVAL; }
            const int bytes = (out.bits + 7) / 8;
            if (block_align < bytes * out.channels) block_align = bytes * out.channels;
            out.frames = (long)(len / block_align);
            out.is_float = fmt_tag == 3;
            if (fmt_tag != 1 && fmt_tag != 3) { set_error("wav: unsupported format tag %d (PCM and IEEE float only)", fmt_tag); return AMDSPEECH_EINVAL; }
            if (header_only) return AMDSPEECH_OK;
            const size_t n = (size_t)out.frames * out.channels;
            if (out.is_float) {
                out.floats.resize(n);
                for (long fr = 0; fr < out.frames; ++fr)
                    for (int c = 0; c < out.channels; ++c) {
                        const uint8_t* p = body + (size_t)fr * block_align + (size_t)c * bytes;
                        if (bytes == 4) { float v; memcpy(&v, p, 4); out.floats[fr * out.channels + c] = v; }
                        else if (bytes == 8) { double v; memcpy(&v, p, 8); out.floats[fr * out.channels + c] = (float)v; }
                        else { set_error("wav: %d-bit float samples", out.bits); return AMDSPEECH_EINVAL; }
                    }
            } else {
                if (bytes < 1 || bytes > 4) { set_error("wav: %d-bit PCM samples", out.bits); return AMDSPEECH_EINVAL; }
                out.ints.resize(n);
                for (long fr = 0; fr < out.frames; ++fr)
                    for (int c = 0; c < out.channels; ++c) {
                        const uint8_t* p = body + (size_t)fr * block_align + (size_t)c * bytes;
                        int32_t v;
                        if (bytes == 1) v = (int32_t)p[0] - 128;                       // 8-bit WAV is unsigned
                        else if (bytes == 2) v = (int16_t)le16(p);
                        else if (bytes == 3) v = (int32_t)((p[0] << 8) | (p[1] << 16) | ((uint32_t)p[2] << 24)) >> 8;
                        else v = (int32_t)le32(p);
                        out.ints[fr * out.channels + c] = v;
                    }
                out.bits = bytes * 8;                                                // container width is the scale
            }
            return AMDSPEECH_OK;
        }
        pos += 8 + len + (len & 1);
    }
    set_error("wav: no data chunk");
    return AMDSPEECH_EINVAL;
}

// ------------------------------------------------------------------------------------------ NIST SPHERE
int decode_sphere(const std::vector<uint8_t>& b, Pcm& out, bool header_only) {
    if (b.size() < 16) { set_error("sphere: file too short"); return AMDSPEECH_EINVAL; }
    const long hsize = atol(std::string((const char*)b.data() + 8, 8).c_str());
    if (hsize < 16 || (size_t)hsize > b.size()) { set_error("sphere: bad header size"); return AMDSPEECH_EINVAL; }
    std::string head((const char*)b.data(), (size_t)hsize);
    long count = -1;
    int nbytes = 2;
    std::string coding = "pcm", order = "01";
    size_t p = head.find('\n', head.find('\n') + 1) + 1;
    while (p < head.size()) {
        size_t e = head.find('\n', p);
        if (e == std::string::npos) e = head.size();
        std::string line = head.substr(p, e - p);
        p = e + 1;
        if (line.compare(0, 8, "end_head") == 0) break;
        char key[64], type[16], val[128];
        if (sscanf(line.c_str(), "%63s %15s %127s", key, type, val) != 3) continue;
        if (!strcmp(key, "sample_count")) count = atol(val);
        else if (!strcmp(key, "sample_rate")) out.rate = atoi(val);
        else if (!strcmp(key, "channel_count")) out.channels = atoi(val);
        else if (!strcmp(key, "sample_n_bytes")) nbytes = atoi(val);
        else if (!strcmp(key, "sample_byte_format")) order = val;
        else if (!strcmp(key, "sample_coding")) coding = val;
    }
    if (out.channels <= 0) out.channels = 1;
    if (out.rate <= 0) { set_error("sphere: no sample_rate"); return AMDSPEECH_EINVAL; }
    if (coding.compare(0, 3, "pcm") != 0 || coding.find("shorten") != std::string::npos || nbytes != 2) {
        set_error("sphere: sample_coding '%s' / %d bytes not supported (16-bit PCM only)", coding.c_str(), nbytes);
        return AMDSPEECH_EINVAL;
    }
    const long avail = (long)((b.size() - (size_t)hsize) / (2 * out.channels));
    out.frames = (count >= 0 && count < avail) ? count : avail;
    out.bits = 16;
    if (header_only) return AMDSPEECH_OK;
    const bool big = order == "10";
    out.ints.resize((size_t)out.frames * out.channels);
    const uint8_t* d = b.data() + hsize;
    for (size_t i = 0; i < out.ints.size(); ++i)
        out.ints[i] = (int16_t)(big ? ((d[2 * i] << 8) | d[2 * i + 1]) : (d[2 * i] | (d[2 * i + 1] << 8)));
    return AMDSPEECH_OK;
}

// ------------------------------------------------------------------------------------------ FLAC
// MSB-first bit reader over a 64-bit window: `acc` holds the next `cnt` stream bits left-aligned (bits below them may
// already hold a prefix of the bytes at p[pos..] -- refill ORs whole big-endian words in, which is idempotent for them).
struct BitReader {
    const uint8_t* p; size_t n; size_t pos = 0; uint64_t acc = 0; int cnt = 0; bool overrun = false;
    BitReader(const uint8_t* data, size_t size) : p(data), n(size) {}
    inline void refill() {
        if (pos + 8 <= n) {
            uint64_t w; memcpy(&w, p + pos, 8);
            acc |= __builtin_bswap64(w) >> cnt;
            const int adv = (63 - cnt) >> 3;
            pos += (size_t)adv; cnt += adv * 8;
        } else {
            while (cnt <= 56 && pos < n) { acc |= (uint64_t)p[pos++] << (56 - cnt); cnt += 8; }
        }
    }
    inline uint64_t get(int k) {                     // k <= 57
        if (k == 0) return 0;
        if (cnt < k) { refill(); if (cnt < k) { overrun = true; acc = 0; cnt = 0; return 0; } }
        const uint64_t v = acc >> (64 - k);
        acc <<= k; cnt -= k;
        return v;
    }
    inline uint32_t get1() { return (uint32_t)get(1); }
    inline int64_t get_signed(int k) {
        if (k == 0) return 0;
        const uint64_t v = get(k);
        return (int64_t)(v << (64 - k)) >> (64 - k);
    }
    inline uint32_t unary() {                        // zeros before the next 1
        uint32_t q = 0;
        for (;;) {
            if (cnt == 0) { refill(); if (cnt == 0) { overrun = true; return q; } }
            const int z = acc ? __builtin_clzll(acc) : 64;
            if (z < cnt) { q += (uint32_t)z; acc <<= z; acc <<= 1; cnt -= z + 1; return q; }
            q += (uint32_t)cnt; acc = 0; cnt = 0;    // the window was all zeros: the bytes at p[pos..] continue the run
        }
    }
    inline void align() { const int k = cnt & 7; acc <<= k; cnt -= k; }
    inline size_t byte_pos() const { return pos - (size_t)(cnt >> 3); }   // bytes consumed (after align, or on a byte boundary)
};

uint8_t crc8(const uint8_t* d, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) { c ^= d[i]; for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1); }
    return c;
}
struct Crc16Table {
    uint16_t t[256];
    Crc16Table() {
        for (int i = 0; i < 256; ++i) { uint16_t c = (uint16_t)(i << 8); for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1); t[i] = c; }
    }
};
uint16_t crc16(const uint8_t* d, size_t n) {
    static const Crc16Table table;         // C++11 function-local static: initialised once, thread-safe (decode runs on a thread pool)
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ table.t[(c >> 8) ^ d[i]]);
    return c;
}

// MD5 (RFC 1321) of the interleaved little-endian samples -- the STREAMINFO signature
struct Md5 {
    uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476; uint64_t len = 0; uint8_t buf[64]; int fill = 0;
    static inline uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
            0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
            0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
            0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391};
        static const int S[64] = {7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22,5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
                                  4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23,6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21};
        uint32_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = le32(p + 4 * i);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B; B = B + rol(A + f + K[i] + m[g], S[i]); A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            const size_t take = (size_t)(64 - fill) < n ? (size_t)(64 - fill) : n;
            memcpy(buf + fill, p, take); fill += (int)take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
        update(l, 8);
        const uint32_t w[4] = {a, b, c, d};
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint8_t)(w[i] >> (8 * k));
    }
};

bool flac_residual(BitReader& br, int order, int blocksize, int64_t* s) {
    const int method = (int)br.get(2);
    if (method > 1) return false;
    const int pbits = method == 0 ? 4 : 5, escape = method == 0 ? 15 : 31;
    const int porder = (int)br.get(4);
    const int parts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
    int i = order;
    for (int part = 0; part < parts; ++part) {
        int count = blocksize >> porder;
        if (part == 0) count -= order;
        if (count < 0) return false;
        const int k = (int)br.get(pbits);
        if (k == escape) {
            const int raw = (int)br.get(5);
            for (int j = 0; j < count; ++j) s[i++] = br.get_signed(raw);
        } else {
            for (int j = 0; j < count; ++j) {
                const uint64_t q = br.unary();
                const uint64_t u = (q << k) | (k ? br.get(k) : 0);
                s[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (br.overrun) return false;
    }
    return i == blocksize;
}

// s[i] += (sum_j coef[j] * s[i-1-j]) >> shift, the order a compile-time constant so the inner product unrolls
template <int ORDER>
void lpc_restore_n(int64_t* s, int blocksize, const int64_t* coef, int shift) {
    for (int i = ORDER; i < blocksize; ++i) {
        int64_t acc = 0;
        for (int j = 0; j < ORDER; ++j) acc += coef[j] * s[i - 1 - j];
        s[i] += acc >> shift;
    }
}
template <int ORDER>
void lpc_restore_pick(int64_t* s, int blocksize, int order, const int64_t* coef, int shift) {
    if (order == ORDER) lpc_restore_n<ORDER>(s, blocksize, coef, shift);
    else if constexpr (ORDER < 32) lpc_restore_pick<ORDER + 1>(s, blocksize, order, coef, shift);
}
inline void lpc_restore(int64_t* s, int blocksize, int order, const int64_t* coef, int shift) {
    lpc_restore_pick<1>(s, blocksize, order, coef, shift);
}

bool flac_subframe(BitReader& br, int bps, int blocksize, std::vector<int64_t>& s) {
    if (br.get1()) return false;                                     // padding bit
    const int type = (int)br.get(6);
    int wasted = 0;
    if (br.get1()) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps <= 0) return false;
    s.assign((size_t)blocksize, 0);
    if (type == 0) {                                                 // CONSTANT
        const int64_t v = br.get_signed(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {                                          // VERBATIM
        for (int i = 0; i < blocksize; ++i) s[i] = br.get_signed(bps);
    } else if (type >= 8 && type <= 12) {                            // FIXED, order = type - 8
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.get_signed(bps);
        if (!flac_residual(br, order, blocksize, s.data())) return false;
        for (int i = order; i < blocksize; ++i) {
            int64_t pred = 0;
            switch (order) {
                case 1: pred = s[i - 1]; break;
                case 2: pred = 2 * s[i - 1] - s[i - 2]; break;
                case 3: pred = 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
                case 4: pred = 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
                default: break;
            }
            s[i] += pred;
        }
    } else if (type >= 32) {                                         // LPC, order = type - 31
        const int order = type - 31;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.get_signed(bps);
        const int prec = (int)br.get(4) + 1;
        if (prec == 16) return false;                                // 0b1111 is invalid
        const int shift = (int)br.get_signed(5);
        if (shift < 0) return false;
        int64_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.get_signed(prec);
        if (!flac_residual(br, order, blocksize, s.data())) return false;
        lpc_restore(s.data(), blocksize, order, coef, shift);
    } else {
        return false;                                                // reserved subframe types
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    return !br.overrun;
}

int decode_flac(const std::vector<uint8_t>& b, Pcm& out, bool header_only, bool check_md5) {
    size_t pos = 4;
    uint64_t total = 0;
    uint8_t md5[16] = {0};
    bool have_info = false, last = false;
    while (!last) {
        if (pos + 4 > b.size()) { set_error("flac: truncated metadata"); return AMDSPEECH_EINVAL; }
        last = (b[pos] & 0x80) != 0;
        const int type = b[pos] & 0x7F;
        const size_t len = ((size_t)b[pos + 1] << 16) | ((size_t)b[pos + 2] << 8) | b[pos + 3];
        pos += 4;
        if (pos + len > b.size()) { set_error("flac: truncated metadata block"); return AMDSPEECH_EINVAL; }
        if (type == 0 && len >= 34) {
            const uint8_t* m = b.data() + pos;
            uint64_t packed = 0;
            for (int i = 10; i < 18; ++i) packed = (packed << 8) | m[i];
            out.rate = (int)(packed >> 44);
            out.channels = (int)((packed >> 41) & 7) + 1;
            out.bits = (int)((packed >> 36) & 31) + 1;
            total = packed & ((1ull << 36) - 1);
            memcpy(md5, m + 18, 16);
            have_info = true;
        }
        pos += len;
    }
    if (!have_info || out.rate <= 0) { set_error("flac: no STREAMINFO block"); return AMDSPEECH_EINVAL; }
    // STREAMINFO's 36-bit total_samples is untrusted (callers size their buffers by it): a frame of <= 65,535
    // samples takes at least ~10 bytes, so the remaining bytes bound what the stream can hold
    const uint64_t plausible = (uint64_t)(b.size() - pos) / 10 * 65535 + 65535;
    if (total > plausible) {
        set_error("flac: STREAMINFO claims %llu samples, %zu bytes of frames cannot hold them", (unsigned long long)total, b.size() - pos);
        return AMDSPEECH_EINVAL;
    }
    out.frames = (long)total;
    if (header_only && total > 0) return AMDSPEECH_OK;

    std::vector<int64_t> ch[8];
    out.ints.clear();
    if (total) out.ints.reserve((size_t)total * out.channels);      // (bounded by `plausible` above)
    long decoded = 0;
    while (pos + 6 <= b.size()) {
        const uint8_t* f = b.data() + pos;
        if (f[0] != 0xFF || (f[1] & 0xFE) != 0xF8) { set_error("flac: lost frame sync at byte %zu", pos); return AMDSPEECH_EINVAL; }
        BitReader br(f, b.size() - pos);
        br.get(16);
        const int bs_code = (int)br.get(4), sr_code = (int)br.get(4), ch_code = (int)br.get(4), ss_code = (int)br.get(3);
        if (br.get1()) { set_error("flac: reserved bit set in frame header"); return AMDSPEECH_EINVAL; }
        int lead = (int)br.get(8);                                   // UTF-8-style coded frame / sample number
        int extra = 0;
        if (lead & 0x80) { while (extra < 6 && (lead & (0x80 >> (extra + 1)))) ++extra; }   // 110xxxxx -> 1 more byte, ...
        for (int i = 0; i < extra; ++i) br.get(8);
        int blocksize;
        if (bs_code == 0) { set_error("flac: reserved block size code"); return AMDSPEECH_EINVAL; }
        else if (bs_code == 1) blocksize = 192;
        else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
        else if (bs_code == 6) blocksize = (int)br.get(8) + 1;
        else if (bs_code == 7) blocksize = (int)br.get(16) + 1;
        else blocksize = 256 << (bs_code - 8);
        if (sr_code == 12) br.get(8); else if (sr_code == 13 || sr_code == 14) br.get(16);
        else if (sr_code == 15) { set_error("flac: invalid sample rate code"); return AMDSPEECH_EINVAL; }
        const size_t hdr_len = br.byte_pos();
        const uint8_t want8 = (uint8_t)br.get(8);
        if (br.overrun || crc8(f, hdr_len) != want8) { set_error("flac: frame header CRC mismatch at byte %zu", pos); return AMDSPEECH_EINVAL; }
        static const int ss_bits[8] = {0, 8, 12, 0, 16, 20, 24, 32};
        const int bps = ss_code == 0 ? out.bits : ss_bits[ss_code];
        if (bps == 0 || ss_code == 3) { set_error("flac: reserved sample size code"); return AMDSPEECH_EINVAL; }
        const int nch = ch_code < 8 ? ch_code + 1 : 2;
        if (ch_code > 10 || nch != out.channels) { set_error("flac: channel assignment %d does not match STREAMINFO", ch_code); return AMDSPEECH_EINVAL; }
        for (int c = 0; c < nch; ++c) {
            const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            if (!flac_subframe(br, bps + (side ? 1 : 0), blocksize, ch[c])) {
                set_error("flac: corrupt subframe in the frame at byte %zu", pos);
                return AMDSPEECH_EINVAL;
            }
        }
        br.align();
        const size_t body_len = br.byte_pos();
        const uint16_t want16 = (uint16_t)br.get(16);
        if (br.overrun || crc16(f, body_len) != want16) { set_error("flac: frame CRC-16 mismatch at byte %zu", pos); return AMDSPEECH_EINVAL; }
        pos += br.byte_pos();
        if (ch_code == 8) for (int i = 0; i < blocksize; ++i) ch[1][i] = ch[0][i] - ch[1][i];
        else if (ch_code == 9) for (int i = 0; i < blocksize; ++i) ch[0][i] = ch[0][i] + ch[1][i];
        else if (ch_code == 10)
            for (int i = 0; i < blocksize; ++i) {
                const int64_t side = ch[1][i], mid = (int64_t)((uint64_t)ch[0][i] << 1) | (side & 1);
                ch[0][i] = (mid + side) >> 1;
                ch[1][i] = (mid - side) >> 1;
            }
        const size_t base = out.ints.size();
        out.ints.resize(base + (size_t)blocksize * nch);
        if (nch == 1) for (int i = 0; i < blocksize; ++i) out.ints[base + i] = (int32_t)ch[0][i];
        else
            for (int i = 0; i < blocksize; ++i)
                for (int c = 0; c < nch; ++c) out.ints[base + (size_t)i * nch + c] = (int32_t)ch[c][i];
        decoded += blocksize;
    }
    if (total && decoded < (long)total) { set_error("flac: stream ends after %ld of %llu samples", decoded, (unsigned long long)total); return AMDSPEECH_EINVAL; }
    if (total && decoded > (long)total) { out.ints.resize((size_t)total * out.channels); decoded = (long)total; }
    out.frames = decoded;
    if (check_md5) {
        bool any = false;
        for (int i = 0; i < 16; ++i) any = any || md5[i];
        if (any) {
            Md5 h;
            const int bytes = (out.bits + 7) / 8;
            std::vector<uint8_t> raw(out.ints.size() * bytes);
            for (size_t i = 0; i < out.ints.size(); ++i)
                for (int k = 0; k < bytes; ++k) raw[i * bytes + k] = (uint8_t)((uint32_t)out.ints[i] >> (8 * k));
            h.update(raw.data(), raw.size());
            uint8_t got[16];
            h.finish(got);
            if (memcmp(got, md5, 16) != 0) { set_error("flac: MD5 of the decoded audio does not match STREAMINFO"); return AMDSPEECH_EINVAL; }
        }
    }
    return AMDSPEECH_OK;
}

int decode_any(const char* path, Pcm& pcm, bool header_only, bool check_md5) {
    if (!path) { set_error("audio: null path"); return AMDSPEECH_EINVAL; }
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) { set_error("audio: cannot read %s", path); return AMDSPEECH_EINVAL; }
    if (buf.size() >= 4 && !memcmp(buf.data(), "fLaC", 4)) return decode_flac(buf, pcm, header_only, check_md5);
    if (buf.size() >= 4 && !memcmp(buf.data(), "RIFF", 4)) return decode_wav(buf, pcm, header_only);
    if (buf.size() >= 7 && !memcmp(buf.data(), "NIST_1A", 7)) return decode_sphere(buf, pcm, header_only);
    set_error("audio: %s is not a WAVE, FLAC or NIST SPHERE file", path);
    return AMDSPEECH_EINVAL;
}

}  // namespace

// No C++ exception may cross the C ABI (a corrupt corpus file must not abort a training run): every entry point
// turns bad_alloc / length_error / anything else into an error code + message.
static int decode_any_noexcept(const char* path, Pcm& pcm, bool header_only, bool check_md5) {
    try {
        return decode_any(path, pcm, header_only, check_md5);
    } catch (const std::exception& e) {
        set_error("audio: %s: %s", path ? path : "(null)", e.what());
    } catch (...) {
        set_error("audio: %s: unknown C++ exception", path ? path : "(null)");
    }
    return AMDSPEECH_EINVAL;
}

extern "C" int amdspeech_audio_probe(const char* path, int* sample_rate, int* channels, long* frames) {
    Pcm pcm;
    if (int rc = decode_any_noexcept(path, pcm, true, false)) return rc;
    if (sample_rate) *sample_rate = pcm.rate;
    if (channels) *channels = pcm.channels;
    if (frames) *frames = pcm.frames;
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_audio_decode(const char* path, float* out, long capacity, long* frames, int* sample_rate,
                                      int verify) {
    Pcm pcm;
    if (int rc = decode_any_noexcept(path, pcm, false, verify != 0)) return rc;
    if (frames) *frames = pcm.frames;
    if (sample_rate) *sample_rate = pcm.rate;
    if (!out) return AMDSPEECH_OK;
    if (capacity < pcm.frames) { set_error("audio_decode: buffer of %ld frames, file has %ld", capacity, pcm.frames); return AMDSPEECH_EINVAL; }
    const int nch = pcm.channels;
    if (pcm.is_float) {
        for (long i = 0; i < pcm.frames; ++i) {
            float acc = 0.f;
            for (int c = 0; c < nch; ++c) acc += pcm.floats[(size_t)i * nch + c];
            out[i] = acc / (float)nch;
        }
    } else {
        const float scale = 1.0f / (float)(1ull << (pcm.bits - 1));
        for (long i = 0; i < pcm.frames; ++i) {
            // each channel is scaled to float32 first, then averaged in float32 (what librosa.to_mono sees)
            float acc = 0.f;
            for (int c = 0; c < nch; ++c) acc += (float)pcm.ints[(size_t)i * nch + c] * scale;
            out[i] = nch == 1 ? acc : acc / (float)nch;
        }
    }
    return AMDSPEECH_OK;
}
