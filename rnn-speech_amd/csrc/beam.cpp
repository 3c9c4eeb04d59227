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
    // utterances are independent: one host thread each (bounded by the core count); evaluation decodes whole mini-batches.
    //
    // Data structure (round 2; the first version kept a std::map keyed by whole prefix vectors and took 45 s for a batch of
    // 32 x 1001 frames at width 100): a prefix is a node (parent, label) in an arena that only ever holds prefixes that have been
    // IN the beam (<= width per frame).  Two candidates of a frame can only denote the same prefix when one is a beam entry j
    // and the other is the extension of its parent i -- also in the beam -- by j's label, so merging needs no search: every
    // beam entry lists its children that are in the beam.  All other extensions are new, distinct prefixes and stay plain
    // (entry, label) pairs until they are selected.  Per frame: width * C additions, one nth_element.
    struct Node { int parent, label; };
    auto decode_one = [&](int b) {
        std::vector<float> lp(C);
        const int Tb = std::min(std::max(lengths[b], 0), T);
        std::vector<Node> arena;
        arena.push_back({-1, -1});                        // the empty prefix
        std::vector<int> slot_of;                         // node -> its slot in the current beam, or -1
        slot_of.push_back(0);
        std::vector<std::vector<std::pair<int, int>>> child_of(1);      // node -> (label, node) of every child ever created
        std::vector<int> node(1, 0);                      // beam slot -> node
        std::vector<Score> sc(1);
        sc[0].pb = 0.0f;
        std::vector<Score> stay;
        std::vector<float> ext;                           // [slot][label]: score of the new prefix (ends in a non-blank), NEG = none
        std::vector<std::vector<std::pair<int, int>>> kids;        // slot -> (label, slot) of its children that are in the beam
        struct Cand { float score; int slot, label; };    // label < 0: the beam entry itself
        std::vector<Cand> cands;
        std::vector<int> scratch_a, scratch_b;
        auto prefix_of = [&](int n, int extra, std::vector<int>& out) {
            out.clear();
            if (extra >= 0) out.push_back(extra);
            for (; n > 0; n = arena[n].parent) out.push_back(arena[n].label);
            std::reverse(out.begin(), out.end());
        };
        for (int t = 0; t < Tb; ++t) {
            const float* x = logits + ((size_t)t * B + b) * C;
            float mx = x[0];
            for (int c = 1; c < C; ++c) mx = std::max(mx, x[c]);
            double sum = 0.0;
            for (int c = 0; c < C; ++c) sum += std::exp((double)x[c] - mx);
            const float lz = mx + (float)std::log(sum);
            for (int c = 0; c < C; ++c) lp[c] = x[c] - lz;

            const int nb = (int)node.size();
            stay.assign(nb, Score());
            ext.assign((size_t)nb * C, NEG);
            kids.resize(nb);
            for (int i = 0; i < nb; ++i) kids[i].clear();
            for (int j = 0; j < nb; ++j) {
                const int par = arena[node[j]].parent;
                if (par >= 0 && slot_of[par] >= 0) kids[slot_of[par]].emplace_back(arena[node[j]].label, j);
            }
            for (int i = 0; i < nb; ++i) {
                const float tot = sc[i].total();
                const int last = arena[node[i]].label;                                  // -1 for the empty prefix
                stay[i].pb = lse2(stay[i].pb, tot + lp[blank]);                         // emit blank
                if (last >= 0) stay[i].pnb = lse2(stay[i].pnb, sc[i].pnb + lp[last]);   // repeat the last label
                float* e = ext.data() + (size_t)i * C;
                for (int c = 0; c < C; ++c) {
                    if (c == blank) continue;
                    const float from = (c == last) ? sc[i].pb : tot;                    // a repeat starts a NEW character only after a blank
                    if (from != NEG) e[c] = from + lp[c];
                }
                for (const auto& kid : kids[i]) {                                       // the same prefix as a beam entry: merge there
                    if (e[kid.first] != NEG) {
                        stay[kid.second].pnb = lse2(stay[kid.second].pnb, e[kid.first]);
                        e[kid.first] = NEG;
                    }
                }
            }
            cands.clear();
            for (int i = 0; i < nb; ++i) {
                const float s2 = stay[i].total();
                if (s2 != NEG) cands.push_back({s2, i, -1});
                const float* e = ext.data() + (size_t)i * C;
                for (int c = 0; c < C; ++c)
                    if (e[c] != NEG) cands.push_back({e[c], i, c});
            }
            if ((int)cands.size() > beam_width) {
                std::nth_element(cands.begin(), cands.begin() + beam_width, cands.end(), [&](const Cand& u, const Cand& v) {
                    if (u.score != v.score) return u.score > v.score;
                    prefix_of(node[u.slot], u.label, scratch_a);                        // (exact ties: lexicographic prefix order)
                    prefix_of(node[v.slot], v.label, scratch_b);
                    return scratch_a < scratch_b;
                });
                cands.resize(beam_width);
            }
            for (int i = 0; i < nb; ++i) slot_of[node[i]] = -1;
            std::vector<int> new_node(cands.size());
            std::vector<Score> new_sc(cands.size());
            for (size_t k = 0; k < cands.size(); ++k) {
                const Cand& cd = cands[k];
                if (cd.label < 0) {
                    new_node[k] = node[cd.slot];
                    new_sc[k] = stay[cd.slot];
                } else {
                    // ONE node per prefix: a prefix that fell out of the beam and comes back is found again under its parent
                    // (its children may still be in the beam and must keep meeting it)
                    const int par = node[cd.slot];
                    int id = -1;
                    for (const auto& ch : child_of[par])
                        if (ch.first == cd.label) { id = ch.second; break; }
                    if (id < 0) {
                        arena.push_back({par, cd.label});
                        slot_of.push_back(-1);
                        child_of.emplace_back();
                        id = (int)arena.size() - 1;
                        child_of[par].emplace_back(cd.label, id);
                    }
                    new_node[k] = id;
                    new_sc[k].pnb = cd.score;
                }
            }
            node.swap(new_node);
            sc.swap(new_sc);
            for (size_t k = 0; k < node.size(); ++k) slot_of[node[k]] = (int)k;
        }
        int best = -1;
        float best_score = NEG;
        std::vector<int> best_prefix, other;
        for (size_t k = 0; k < node.size(); ++k) {
            const float s2 = sc[k].total();
            bool better = best < 0 || s2 > best_score;
            if (!better && s2 == best_score) {            // (the first of equals in lexicographic order, as a sorted container gives)
                prefix_of(node[k], -1, other);
                better = other < best_prefix;
            }
            if (better) { best = (int)k; best_score = s2; prefix_of(node[k], -1, best_prefix); }
        }
        int n = 0;
        int* row = ids + (size_t)b * T;
        for (size_t i = 0; i < best_prefix.size(); ++i) {
            if (merge_repeated && i > 0 && best_prefix[i] == best_prefix[i - 1]) continue;
            row[n++] = best_prefix[i];
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
