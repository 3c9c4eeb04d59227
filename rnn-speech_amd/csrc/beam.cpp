// CTC prefix beam search on the HOST (next-tier row of SURVEY.md 8f: stands where
// tf.nn.ctc_beam_search_decoder(logits, seq_len) -- beam_width 100, top_paths 1, merge_repeated
// True -- sits at /root/reference/models/AcousticModel.py:312).  The hot path decodes greedily on
// the GPU; this is the evaluation-time decoder.  Log domain throughout.
//
// TensorFlow's decoder (tensorflow/core/util/ctc/ctc_beam_search.h, not available offline; restated)
// keeps, per prefix, the log-probability of ending in blank and in a non-blank label, extends every
// live prefix by every label, keeps the `beam_width` most probable prefixes per frame, and -- with
// merge_repeated=True -- finally collapses consecutive duplicate labels of the top path (so
// "a b b" is returned as "a b": the reference's char map carries double-letter tokens for this).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <exception>
#include <thread>
#include <limits>
#include <map>
#include <vector>

#include "common.h"

namespace {

const float NEG = -std::numeric_limits<float>::infinity();

inline float lse2(float a, float b) {
    if (a == NEG) return b;
    if (b == NEG) return a;
    const float m = a > b ? a : b;
    return m + std::log(std::exp(a - m) + std::exp(b - m));
}

struct Score { float pb = NEG, pnb = NEG; float total() const { return lse2(pb, pnb); } };

}  // namespace

extern "C" int amdspeech_ctc_beam_search_host(const float* logits, const int* lengths, int T, int B, int C,
                                              int beam_width, int merge_repeated, int* ids, int* out_len,
                                              float* log_prob) {
    using amdspeech::set_error;
    if (!logits || !lengths || !ids || !out_len || T <= 0 || B <= 0 || C <= 1 || beam_width <= 0) {
        set_error("ctc_beam_search_host: bad arguments");
        return AMDSPEECH_EINVAL;
    }
    const int blank = C - 1;
    typedef std::vector<int> Prefix;
    // utterances are independent: one host thread each (bounded by the core count); evaluation decodes whole
    // mini-batches, and at width 100 a 10 s utterance is ~10^7 prefix extensions
    auto decode_one = [&](int b) {
        std::vector<float> lp(C);
        const int Tb = std::min(std::max(lengths[b], 0), T);
        std::map<Prefix, Score> beams;
        beams[Prefix()].pb = 0.0f;
        for (int t = 0; t < Tb; ++t) {
            const float* x = logits + ((size_t)t * B + b) * C;
            float mx = x[0];
            for (int c = 1; c < C; ++c) mx = std::max(mx, x[c]);
            double sum = 0.0;
            for (int c = 0; c < C; ++c) sum += std::exp((double)x[c] - mx);
            const float lz = mx + (float)std::log(sum);
            for (int c = 0; c < C; ++c) lp[c] = x[c] - lz;

            std::map<Prefix, Score> next;
            for (const auto& kv : beams) {
                const Prefix& pre = kv.first;
                const Score& sc = kv.second;
                const float tot = sc.total();
                Score& same = next[pre];
                same.pb = lse2(same.pb, tot + lp[blank]);                       // emit blank
                if (!pre.empty()) same.pnb = lse2(same.pnb, sc.pnb + lp[pre.back()]);   // repeat last label
                Prefix ext(pre);
                ext.push_back(0);
                for (int c = 0; c < C; ++c) {
                    if (c == blank) continue;
                    // a repeated label only starts a NEW character after a blank
                    const float from = (!pre.empty() && pre.back() == c) ? sc.pb : tot;
                    if (from == NEG) continue;
                    ext.back() = c;
                    Score& e = next[ext];
                    e.pnb = lse2(e.pnb, from + lp[c]);
                }
            }
            if ((int)next.size() > beam_width) {
                std::vector<std::pair<float, const Prefix*>> order;
                order.reserve(next.size());
                for (const auto& kv : next) order.emplace_back(kv.second.total(), &kv.first);
                std::nth_element(order.begin(), order.begin() + beam_width, order.end(),
                                 [](const std::pair<float, const Prefix*>& a, const std::pair<float, const Prefix*>& b2) {
                                     return a.first > b2.first || (a.first == b2.first && *a.second < *b2.second);
                                 });
                std::map<Prefix, Score> kept;
                for (int i = 0; i < beam_width; ++i) kept[*order[i].second] = next[*order[i].second];
                beams.swap(kept);
            } else {
                beams.swap(next);
            }
        }
        const Prefix* best = nullptr;
        float best_score = NEG;
        for (const auto& kv : beams) {
            const float s = kv.second.total();
            if (best == nullptr || s > best_score) { best = &kv.first; best_score = s; }
        }
        int n = 0;
        int* row = ids + (size_t)b * T;
        if (best) {
            for (size_t i = 0; i < best->size(); ++i) {
                if (merge_repeated && i > 0 && (*best)[i] == (*best)[i - 1]) continue;
                row[n++] = (*best)[i];
            }
        }
        for (int i = n; i < T; ++i) row[i] = C;       // reference pads dense predictions with num_labels (:718)
        out_len[b] = n;
        if (log_prob) log_prob[b] = best_score;
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int nthreads = (int)std::min<unsigned>((unsigned)B, std::min(hw, 64u));
    try {
        if (nthreads <= 1) {
            for (int b = 0; b < B; ++b) decode_one(b);
        } else {
            std::atomic<int> next_row(0);
            std::vector<std::thread> pool;
            std::atomic<bool> failed(false);
            for (int w = 0; w < nthreads; ++w)
                pool.emplace_back([&] {
                    try {
                        for (int b = next_row++; b < B; b = next_row++) decode_one(b);
                    } catch (...) { failed = true; }
                });
            for (auto& th : pool) th.join();
            if (failed) { set_error("ctc_beam_search_host: out of memory in a decode thread"); return AMDSPEECH_EINVAL; }
        }
    } catch (const std::exception& e) {       // (no C++ exception crosses the C ABI)
        set_error("ctc_beam_search_host: %s", e.what());
        return AMDSPEECH_EINVAL;
    }
    return AMDSPEECH_OK;
}

// ---- CRC32C (Castagnoli), table driven: checksums of TensorFlow-bundle checkpoints (tf_bundle.py) ----
namespace {
struct Crc32cTable {
    uint32_t t[8][256];
    Crc32cTable() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFF];
    }
};
}  // namespace

extern "C" uint32_t amdspeech_crc32c(const void* data, size_t n, uint32_t crc) {
    static const Crc32cTable tab;          // function-local static: initialised once, thread-safe
    const uint32_t (&table)[8][256] = tab.t;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    crc = ~crc;
    while (n >= 8) {                                   // slice-by-8
        const uint32_t lo = crc ^ (p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24));
        crc = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
              table[3][p[4]] ^ table[2][p[5]] ^ table[1][p[6]] ^ table[0][p[7]];
        p += 8; n -= 8;
    }
    while (n--) crc = table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
