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
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>      // (the AVX2 rank sort below; other hosts keep std::sort)
#endif
#include <cstring>
#include <cstdint>
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
    // m + log(exp(a - m) + exp(b - m)) with m = max(a, b): one of the two exponentials is exp(0) = 1 exactly and the other's
    // argument is -(|a - b|) exactly, so this is the same float result with one libm call less
    const float m = a > b ? a : b, d = a > b ? b - a : a - b;
    return m + std::log(1.0f + std::exp(d));
}

struct Score { float pb = NEG, pnb = NEG; float total() const { return lse2(pb, pnb); } };

// Sorting ~100 unique 64-bit keys per frame: std::sort spends its time in mispredicted branches (a third of a frame on an EPYC
// 9575F).  Rank sort instead -- the place of a key is the number of keys below it, counted four at a time with AVX2 compares,
// no data-dependent branch: n^2 / 4 vector operations (n <= 128 here).  Hosts without AVX2 keep std::sort.
#if defined(__x86_64__)
__attribute__((target("avx2"))) void rank_sort_avx2(const uint64_t* keys, int n, uint64_t* out, int64_t* tmp) {
    const int np = (n + 3) & ~3;
    for (int i = 0; i < n; ++i) tmp[i] = (int64_t)(keys[i] ^ 0x8000000000000000ull);      // unsigned order as signed order
    for (int i = n; i < np; ++i) tmp[i] = INT64_MAX;                                       // padding: never below anything
    for (int i = 0; i < n; ++i) {
        const __m256i ki = _mm256_set1_epi64x(tmp[i]);
        __m256i acc = _mm256_setzero_si256();
        for (int j = 0; j < np; j += 4)
            acc = _mm256_sub_epi64(acc, _mm256_cmpgt_epi64(ki, _mm256_loadu_si256(reinterpret_cast<const __m256i*>(tmp + j))));
        const __m128i h = _mm_add_epi64(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
        const int r = (int)(_mm_cvtsi128_si64(h) + _mm_extract_epi64(h, 1));
        out[r] = keys[i];
    }
}
#endif
bool has_avx2() {      // (function-local: initialised on first use, whatever the order of static constructors at load time)
#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__)      // (this file also passes through hipcc's device compilation, where the builtin does not exist)
    return false;
#else
    static const bool v = (__builtin_cpu_init(), __builtin_cpu_supports("avx2") != 0);
    return v;
#endif
}

}  // namespace

namespace {

// One decoder, two ways of finding the `beam_width` best candidates of a frame:
//   exhaustive (round 2; AMDSPEECH_BEAM_EXHAUSTIVE=1, the reference point of tests/ and tools/beam_equiv.py): score all
//       width x C extensions, one nth_element;
//   bounded (round 3, default; select_frontier): only the pairs that can be in the beam at all are scored -- see there.
//       100 ms -> ~15 ms per 1001-frame utterance at width 100.  Same scores (same float additions), same total order (score,
//       then the lexicographic order of the prefixes), hence the same beams and bit-identical results.
struct Node { int parent, label, first_child, next_sibling; };      // (children: an intrusive list -- no allocation per node)
struct Cand { float score; int slot, label; };    // label < 0: the beam entry itself

struct Decoder {
    int C, blank, width;
    bool exhaustive;
    std::vector<Node> arena;
    std::vector<int> depth;                           // node -> length of its prefix
    std::vector<int> slot_of;                         // node -> its slot in the current beam, or -1
    std::vector<int> node;                            // beam slot -> node
    std::vector<Score> sc;
    std::vector<float> tot;                           // sc[i].total(), kept from the frame that selected the entry
    std::vector<Score> stay;
    std::vector<float> ext;                           // exhaustive: [slot][label]
    std::vector<int> kid_head, kid_next;              // slot -> its children that are in the beam (a list through kid_next; the label is the child's own)
    std::vector<Cand> cands, tie_group;
    std::vector<int> new_node;
    std::vector<Score> new_sc;
    std::vector<float> new_tot;
    std::vector<float> lp;
    std::vector<int> order;                           // non-blank labels by (lp descending, label ascending)
    std::vector<int> rank;                            // beam slots by (tot descending, slot ascending)
    std::vector<unsigned char> merged;                // frontier: [slot][label] = this extension went into a beam child (zero between frames)
    std::vector<float> lpo, tr;                       // lp in `order`; tot in `rank` order
    std::vector<uint64_t> keys;
    std::vector<uint64_t> keys2;
    std::vector<int64_t> keys_tmp;
    void sort_keys() {                                // keys ascending (unique: the index is in the low half)
        const int n = (int)keys.size();
#if defined(__x86_64__)
        if (has_avx2() && n > 8 && n <= 512) {
            keys2.resize(n);
            keys_tmp.resize((n + 3) & ~3);
            rank_sort_avx2(keys.data(), n, keys2.data(), keys_tmp.data());
            keys.swap(keys2);
            return;
        }
#endif
        std::sort(keys.begin(), keys.end());
    }
    static uint64_t desc_key(float v, int idx) {      // ascending key order = (v descending, idx ascending); -0 counts as +0
        uint32_t u;
        v += 0.0f;
        memcpy(&u, &v, 4);
        u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // monotone in v
        return ((uint64_t)(~u) << 32) | (uint32_t)idx;
    }

    void prefix_of(int n, int extra, std::vector<int>& out) const {
        out.clear();
        if (extra >= 0) out.push_back(extra);
        for (; n > 0; n = arena[n].parent) out.push_back(arena[n].label);
        std::reverse(out.begin(), out.end());
    }
    // prefix(nu) [+ eu]  <  prefix(nv) [+ ev] in lexicographic order (a proper prefix is smaller), without writing the prefixes
    // out: the arena is a trie (one node per prefix), so the two paths are walked up to their lowest common ancestor and the
    // labels of the two branches under it decide.  Exact score ties are COMMON (dozens per frame: permuted label pairs reach
    // the same float sums) and are mostly between prefixes that split one or two labels ago -- with the prefixes materialised
    // (round 2's comparator) the neighbour-comparing heap spent 90 % of its time here.
    bool lex_less(int nu, int eu, int nv, int ev) const {
        int a = nu, b = nv, la = eu >= 0 ? eu : -1, lb = ev >= 0 ? ev : -1;
        while (depth[a] > depth[b]) { la = arena[a].label; a = arena[a].parent; }
        while (depth[b] > depth[a]) { lb = arena[b].label; b = arena[b].parent; }
        while (a != b) { la = arena[a].label; a = arena[a].parent; lb = arena[b].label; b = arena[b].parent; }
        if (la < 0 || lb < 0) return la < 0 && lb >= 0;
        return la < lb;
    }
    // strict total order: better candidates first (exact ties: lexicographic prefix order)
    bool better(const Cand& u, const Cand& v) const {
        if (u.score != v.score) return u.score > v.score;
        return lex_less(node[u.slot], u.label, node[v.slot], v.label);
    }

    void reset(int C_, int width_, bool exhaustive_) {
        C = C_; blank = C_ - 1; width = width_; exhaustive = exhaustive_;
        arena.assign(1, Node{-1, -1, -1, -1});
        depth.assign(1, 0);
        slot_of.assign(1, 0);
        node.assign(1, 0);
        sc.assign(1, Score());
        sc[0].pb = 0.0f;
        tot.assign(1, 0.0f);
        lp.resize(C);
    }
    void reserve_for(int frames) {                    // at most `width` new prefixes per frame
        const size_t n = (size_t)frames * width + 1;
        arena.reserve(n); depth.reserve(n); slot_of.reserve(n);
    }

    float from_of(int i, int c) const { return c == arena[node[i]].label ? sc[i].pb : tot[i]; }

    void select_exhaustive(int nb) {
        ext.assign((size_t)nb * C, NEG);
        for (int i = 0; i < nb; ++i) {
            float* e = ext.data() + (size_t)i * C;
            for (int c = 0; c < C; ++c) {
                if (c == blank) continue;
                const float from = from_of(i, c);                                   // a repeat starts a NEW character only after a blank
                if (from != NEG) e[c] = from + lp[c];
            }
            for (int j = kid_head[i]; j >= 0; j = kid_next[j]) e[arena[node[j]].label] = NEG;      // the same prefix as a beam entry: merged there
        }
        cands.clear();
        for (int i = 0; i < nb; ++i) {
            const float s2 = stay[i].total();
            if (s2 != NEG) cands.push_back({s2, i, -1});
            const float* e = ext.data() + (size_t)i * C;
            for (int c = 0; c < C; ++c)
                if (e[c] != NEG) cands.push_back({e[c], i, c});
        }
        if ((int)cands.size() > width) {
            std::nth_element(cands.begin(), cands.begin() + width, cands.end(), [&](const Cand& u, const Cand& v) { return better(u, v); });
            cands.resize(width);
        }
    }

    // The extensions of entry i by label c score tot_i + lp[c] (all but i's repeat label): rank one.  With the entries sorted by
    // tot and the labels by lp, the pair at sorted position (r, k) is dominated -- score >= its own, float addition is monotone --
    // by the (r+1)(k+1)-1 pairs of the rectangle above it, of which at most r+1 are repeat pairs (those score LESS, from pb) and
    // the merged ones stand for the stay candidate of a beam child that scores at least as much.  So a pair with (r+1) k > width
    // has `width` candidates at least as good and can only belong to the beam through an exact score tie: the candidates are the
    // pairs under that hyperbola (~width (ln width + 1.6) of them: 620 instead of 7,900 at width 100), the stay candidates and
    // the repeat pairs; one nth_element with the exact comparator; and if the first pair left out of some entry's list ties with
    // the cut, the frame falls back to the exhaustive scan (exact ties at the cut are rare; ties elsewhere cost nothing).
    void select_frontier(int nb) {
        // both orders are (value descending, index ascending): one 64-bit key per element -- the float's bits made monotone and
        // inverted, above the index -- and a plain integer sort (the comparator with two indirections was a third of a frame)
        keys.clear();
        for (int c = 0; c < C; ++c)
            if (c != blank) keys.push_back(desc_key(lp[c], c));
        sort_keys();
        const int nl = (int)keys.size();
        order.resize(nl);
        lpo.resize(nl);
        for (int k = 0; k < nl; ++k) { order[k] = (int)(keys[k] & 0xFFFFFFFFu); lpo[k] = lp[order[k]]; }
        keys.resize(nb);
        for (int i = 0; i < nb; ++i) keys[i] = desc_key(tot[i], i);
        int nsorted = nb;                               // entries in exact rank order at the front of `rank`
        if (has_avx2() && nb <= 512) {
            sort_keys();
        } else {      // (entries behind position width/2 all get the same two labels, whatever their order: only the front is sorted)
            nsorted = std::min(nb, width / 2 + 1);
            if (nsorted < nb) std::nth_element(keys.begin(), keys.begin() + nsorted, keys.end());
            std::sort(keys.begin(), keys.begin() + nsorted);
        }
        rank.resize(nb);
        for (int i = 0; i < nb; ++i) rank[i] = (int)(keys[i] & 0xFFFFFFFFu);
        // extensions that went into a beam child: marked here, unmarked behind the scan (the table stays zero between frames)
        if (merged.size() < (size_t)nb * C) merged.resize((size_t)nb * C, 0);
        for (int i = 0; i < nb; ++i)
            for (int j = kid_head[i]; j >= 0; j = kid_next[j]) merged[(size_t)i * C + arena[node[j]].label] = 1;
        // A floor under the frame's cut that costs next to nothing.  Candidates that are KNOWN to score at least v:
        //   * the corner block of the first R ranked entries x the first K labels with v = tot_(R-1) + lp_(K-1): R K pairs (both
        //     factors sorted, float addition monotone), of which at most R are repeat pairs and kids(R) went into beam children;
        //   * the stay candidate of every entry with tot_r + lp_blank >= v (its score lse(pb, pnb) >= pb = tot_r + lp_blank) --
        //     the entries are in rank order, so that is a prefix.
        // For a block height R the smallest K with R K - R - kids(R) + stays(v) >= width gives a lower bound v of the width-th
        // best candidate; the best over a few heights is the floor: tall thin blocks when the posteriors are flat, square ones
        // when they are peaked, hardly any block when the blank dominates (a young CTC model: the beam is its own stays).
        // Pairs and stays below the floor are not generated (615 -> 120-250 pairs per frame).
        float floor_ = NEG;
        {
            tr.resize(nsorted);
            int nv = 0;                                 // ranked entries with a finite score (the others sort behind them)
            for (; nv < nsorted; ++nv) {
                tr[nv] = tot[rank[nv]];
                if (tr[nv] == NEG) break;
            }
            const float lpb = lp[blank];
            if (nv >= width) floor_ = tr[width - 1] + lpb;                          // (no block at all: `width` stays)
            static const int heights[] = {1, 2, 3, 4, 6, 8, 12, 16};
            int kids_r = 0, rdone = 0;
            for (int R : heights) {
                if (R > nv || R > width / 2 + 1) break;
                for (; rdone < R; ++rdone)
                    for (int j = kid_head[rank[rdone]]; j >= 0; j = kid_next[j]) ++kids_r;
                const float base = tr[R - 1];
                int stays_v = 0;
                for (int K = 1; K <= nl; ++K) {
                    const float v = base + lpo[K - 1];
                    while (stays_v < nv && tr[stays_v] + lpb >= v) ++stays_v;
                    if (R * K - R - kids_r + stays_v >= width) { floor_ = std::max(floor_, v); break; }
                }
            }
        }
        cands.clear();
        float left_out = NEG;                                                       // the best pair that the bound left out
        for (int r = 0; r < nb; ++r) {
            const int i = rank[r];
            const int last = arena[node[i]].label;
            const unsigned char* mi = merged.data() + (size_t)i * C;
            if (last >= 0 && sc[i].pb != NEG && !mi[last]) {
                const float s2 = sc[i].pb + lp[last];                               // the repeat pair: scored from pb
                if (s2 >= floor_) cands.push_back({s2, i, last});
            }
            const float ti = tot[i];
            if (ti == NEG) continue;
            const int kmax = std::min(nl - 1, width / (r + 1));                     // pairs (r, k) with (r+1) k <= width
            int k = 0;
            for (; k <= kmax; ++k) {
                const float s2 = ti + lpo[k];
                if (s2 < floor_) break;                                             // (and so is everything behind it in this row)
                const int c = order[k];
                if (c == last || mi[c]) continue;
                cands.push_back({s2, i, c});
            }
            if (k <= kmax) continue;                                                // (stopped by the floor: nothing behind can tie with the cut)
            for (k = kmax + 1; k < nl; ++k) {                                       // the first real pair behind the bound
                const int c = order[k];
                if (c == last || mi[c]) continue;
                left_out = std::max(left_out, ti + lpo[k]);
                break;
            }
        }
        for (int i = 0; i < nb; ++i)
            for (int j = kid_head[i]; j >= 0; j = kid_next[j]) merged[(size_t)i * C + arena[node[j]].label] = 0;
        // The extensions alone give a first bound: with `width` of them at or above `bound`, the frame's cut cannot be lower, so
        // the extensions below it are dropped now (the bulk of the ~550) and a stay candidate -- whose score lse(pb, pnb) costs
        // two libm calls -- is only evaluated when max(pb, pnb) + ln 2 reaches the bound (float addition and libm's log are
        // monotone, so that IS an upper bound of the float result).
        float bound = NEG;
        if ((int)cands.size() >= width) {
            std::nth_element(cands.begin(), cands.begin() + (width - 1), cands.end(), [](const Cand& u, const Cand& v) { return u.score > v.score; });
            bound = cands[width - 1].score;
            size_t n = (size_t)width;
            for (size_t q = (size_t)width; q < cands.size(); ++q)
                if (cands[q].score == bound) cands[n++] = cands[q];
            cands.resize(n);
        }
        const float lim = std::max(bound, floor_);
        for (int i = 0; i < nb; ++i) {
            const float m = std::max(stay[i].pb, stay[i].pnb);
            if (m == NEG || m + 0.6931472f < lim) continue;
            const float s2 = stay[i].total();
            if (s2 >= lim) cands.push_back({s2, i, -1});
        }
        if ((int)cands.size() > width) {
            // by score alone first (a plain float comparison: exact ties are dozens per frame, and every comparison of a tied
            // pair walks the trie), then the exact order inside the group that ties with the cut
            std::nth_element(cands.begin(), cands.begin() + (width - 1), cands.end(), [](const Cand& u, const Cand& v) { return u.score > v.score; });
            const float cut = cands[width - 1].score;
            if (left_out >= cut) { select_exhaustive(nb); return; }                 // an exact tie across the bound: decide it exhaustively
            size_t keep = 0;
            tie_group.clear();
            for (const Cand& cd : cands) {
                if (cd.score > cut) cands[keep++] = cd;
                else if (cd.score == cut) tie_group.push_back(cd);
            }
            const size_t room = (size_t)width - keep;
            if (tie_group.size() > room) {
                std::nth_element(tie_group.begin(), tie_group.begin() + room, tie_group.end(), [&](const Cand& u, const Cand& v) { return better(u, v); });
                tie_group.resize(room);
            }
            cands.resize(keep);
            cands.insert(cands.end(), tie_group.begin(), tie_group.end());
        } else if (lim != NEG) {
            // exactly `width` candidates reach the bound (fewer cannot happen: `width` of them are known to): they are the beam,
            // unless a pair that the hyperbola left out ties with the weakest of them
            float cut = cands.empty() ? NEG : cands[0].score;
            for (const Cand& cd : cands) cut = std::min(cut, cd.score);
            if ((int)cands.size() < width || left_out >= cut) select_exhaustive(nb);
        } else if (left_out != NEG) {
            select_exhaustive(nb);                                                  // (fewer candidates than the beam is wide: take everything)
        }
    }

    void frame(const float* x) {
        float mx = x[0];
        for (int c = 1; c < C; ++c) mx = std::max(mx, x[c]);
        double sum = 0.0;
        for (int c = 0; c < C; ++c) sum += std::exp((double)x[c] - mx);
        const float lz = mx + (float)std::log(sum);
        for (int c = 0; c < C; ++c) lp[c] = x[c] - lz;

        const int nb = (int)node.size();
        stay.assign(nb, Score());
        kid_head.assign(nb, -1);
        kid_next.resize(nb);
        for (int j = nb - 1; j >= 0; --j) {                                         // (backwards: the lists come out in slot order)
            const int par = arena[node[j]].parent;
            if (par >= 0 && slot_of[par] >= 0) { kid_next[j] = kid_head[slot_of[par]]; kid_head[slot_of[par]] = j; }
        }
        for (int i = 0; i < nb; ++i) {
            const int last = arena[node[i]].label;                                  // -1 for the empty prefix
            stay[i].pb = lse2(stay[i].pb, tot[i] + lp[blank]);                      // emit blank
            if (last >= 0) stay[i].pnb = lse2(stay[i].pnb, sc[i].pnb + lp[last]);   // repeat the last label
            for (int j = kid_head[i]; j >= 0; j = kid_next[j]) {                    // the same prefix as a beam entry: merge there
                const int c = arena[node[j]].label;
                const float from = from_of(i, c);
                if (from != NEG) stay[j].pnb = lse2(stay[j].pnb, from + lp[c]);
            }
        }
        if (exhaustive) select_exhaustive(nb); else select_frontier(nb);

        for (int i = 0; i < nb; ++i) slot_of[node[i]] = -1;
        new_node.resize(cands.size());
        new_sc.assign(cands.size(), Score());
        new_tot.resize(cands.size());
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
                for (int ch = arena[par].first_child; ch >= 0; ch = arena[ch].next_sibling)
                    if (arena[ch].label == cd.label) { id = ch; break; }
                if (id < 0) {
                    id = (int)arena.size();
                    arena.push_back({par, cd.label, -1, arena[par].first_child});
                    arena[par].first_child = id;
                    depth.push_back(depth[par] + 1);
                    slot_of.push_back(-1);
                }
                new_node[k] = id;
                new_sc[k].pnb = cd.score;
            }
            new_tot[k] = cd.score;                     // (= new_sc[k].total(): the candidate's score IS its total)
        }
        node.swap(new_node);
        sc.swap(new_sc);
        tot.swap(new_tot);
        for (size_t k = 0; k < node.size(); ++k) slot_of[node[k]] = (int)k;
    }

    // -> the best prefix (first of equals in lexicographic order) and its log probability
    float best(std::vector<int>& best_prefix) {
        int bi = -1;
        float best_score = NEG;
        for (size_t k = 0; k < node.size(); ++k) {
            const float s2 = tot[k];
            if (bi < 0 || s2 > best_score || (s2 == best_score && lex_less(node[k], -1, node[bi], -1))) { bi = (int)k; best_score = s2; }
        }
        prefix_of(bi >= 0 ? node[bi] : 0, -1, best_prefix);
        return best_score;
    }
};

}  // namespace

extern "C" int amdspeech_ctc_beam_search_host_mt(const float* logits, const int* lengths, int T, int B, int C,
                                                 int beam_width, int merge_repeated, int* ids, int* out_len,
                                                 float* log_prob, int max_threads);

extern "C" int amdspeech_ctc_beam_search_host(const float* logits, const int* lengths, int T, int B, int C,
                                              int beam_width, int merge_repeated, int* ids, int* out_len,
                                              float* log_prob) {
    return amdspeech_ctc_beam_search_host_mt(logits, lengths, T, B, C, beam_width, merge_repeated, ids, out_len, log_prob, 0);
}

// max_threads > 0 caps the decode threads of this call (0: one per utterance, bounded by the core count): the training-time
// decoder runs BESIDE the training thread and wants a steady load on a few cores, not a burst on all of them.
extern "C" int amdspeech_ctc_beam_search_host_mt(const float* logits, const int* lengths, int T, int B, int C,
                                                 int beam_width, int merge_repeated, int* ids, int* out_len,
                                                 float* log_prob, int max_threads) {
    using amdspeech::set_error;
    if (!logits || !lengths || !ids || !out_len || T <= 0 || B <= 0 || C <= 1 || beam_width <= 0) {
        set_error("ctc_beam_search_host: bad arguments");
        return AMDSPEECH_EINVAL;
    }
    const bool exhaustive = amdspeech::runtime_switch("AMDSPEECH_BEAM_EXHAUSTIVE", 0) != 0;      // (read per call: tests switch it)
    // utterances are independent: one host thread each (bounded by the core count); evaluation decodes whole mini-batches.
    //
    // Data structure (round 2; the first version kept a std::map keyed by whole prefix vectors and took 45 s for a batch of
    // 32 x 1001 frames at width 100): a prefix is a node (parent, label) in an arena that only ever holds prefixes that have been
    // IN the beam (<= width per frame).  Two candidates of a frame can only denote the same prefix when one is a beam entry j
    // and the other is the extension of its parent i -- also in the beam -- by j's label, so merging needs no search: every
    // beam entry lists its children that are in the beam.  All other extensions are new, distinct prefixes and stay plain
    // (entry, label) pairs until they are selected.
    // (one Decoder per worker thread, re-used for its utterances: its arrays keep their capacity, and a prefix costs no heap
    //  allocation of its own -- with one small vector per trie node the decode threads of a mini-batch spent their time in the
    //  allocator and the page-fault path of one shared address space: 2 threads took 2.4x as long as one)
    auto decode_one = [&](Decoder& d, int b) {
        d.reset(C, beam_width, exhaustive);
        const int Tb = std::min(std::max(lengths[b], 0), T);
        d.reserve_for(Tb);
        for (int t = 0; t < Tb; ++t) d.frame(logits + ((size_t)t * B + b) * C);
        std::vector<int> best_prefix;
        const float best_score = d.best(best_prefix);
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
    int nthreads = (int)std::min<unsigned>((unsigned)B, std::min(hw, 64u));
    if (max_threads > 0) nthreads = std::min(nthreads, max_threads);
    try {
        if (nthreads <= 1) {
            Decoder d;
            for (int b = 0; b < B; ++b) decode_one(d, b);
        } else {
            std::atomic<int> next_row(0);
            std::vector<std::thread> pool;
            std::atomic<bool> failed(false);
            for (int w = 0; w < nthreads; ++w)
                pool.emplace_back([&] {
                    try {
                        Decoder d;      // (fresh per call on purpose: a process-wide pool of decoders was tried -- no page faults, but the
                        // arrays are then warm in ANOTHER core's caches / on the other socket's memory: slower on a two-socket host)
                        for (int b = next_row++; b < B; b = next_row++) decode_one(d, b);
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

// Levenshtein distance of n_pairs sequence pairs on the HOST (the training-time error rate of the asynchronous beam decoder:
// tf.edit_distance at models/AcousticModel.py:370, un-normalised): a [n_pairs, lda], b [n_pairs, ldb], out int32 [n_pairs].
extern "C" int amdspeech_edit_distance_host(const int* a, const int* a_len, int lda, const int* b, const int* b_len, int ldb,
                                            int n_pairs, int* out) {
    if (!a || !a_len || !b || !b_len || !out || n_pairs < 0 || lda < 0 || ldb < 0) {
        amdspeech::set_error("edit_distance_host: bad arguments");
        return AMDSPEECH_EINVAL;
    }
    try {
        std::vector<int> prev, cur;
        for (int p = 0; p < n_pairs; ++p) {
            const int na = std::min(std::max(a_len[p], 0), lda), nb = std::min(std::max(b_len[p], 0), ldb);
            const int* x = a + (size_t)p * lda;
            const int* y = b + (size_t)p * ldb;
            prev.resize(nb + 1); cur.resize(nb + 1);
            for (int j = 0; j <= nb; ++j) prev[j] = j;
            for (int i = 1; i <= na; ++i) {
                cur[0] = i;
                for (int j = 1; j <= nb; ++j)
                    cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + (x[i - 1] != y[j - 1] ? 1 : 0));
                prev.swap(cur);
            }
            out[p] = prev[nb];
        }
    } catch (const std::exception& e) {
        amdspeech::set_error("edit_distance_host: %s", e.what());
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
