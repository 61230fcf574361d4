// Sequential per-class leaderboard (host, exact) -- product implementation behind
// utils.clip_pseudolabels.compute_pseudo_labels (reference utils/clip_pseudolabels.py:49-112) and the
// nine assign_pseudo_labels (e.g. methods/transductive_zsl/multimodal_fpl.py:194-285).
//
// Semantics kept from the reference: while a board has never overflowed it is an
// UNSORTED append list and the admission test looks at its last element; the first overflow sorts
// it (score descending, ties by path descending, stable) and from then on it stays sorted and the
// last element is the minimum; an image rejected by its arg-max class is offered to EVERY other
// class.  The reference walks those classes in descending-probability order, but each board's
// update depends only on that board and (p[j], path), so the walk order cannot change the result:
// this implementation walks j = 0..c-1 and replaces the per-overflow full sort by one stable
// binary insertion once a board is sorted.  O(n*c) compares, no allocation inside the scan.
// The path strings are represented by their rank among all paths (dense, equal strings share a
// rank), computed by the host layer with Python's own string order.
#include <algorithm>
#include <vector>

#include "common.h"

namespace {
struct Entry {
    float score;
    int64_t rank;
    int32_t img;
};
inline bool greater_than(const Entry& a, const Entry& b) {
    if (a.score != b.score) return a.score > b.score;
    return a.rank > b.rank;
}
struct Board {
    std::vector<Entry> e;
    bool sorted = false;
};
inline void offer(Board& b, int64_t k, const Entry& x) {
    if ((int64_t)b.e.size() < k) {
        b.e.push_back(x);
        return;
    }
    if (!(b.e.back().score < x.score)) return;
    if (!b.sorted) {
        b.e.push_back(x);
        std::stable_sort(b.e.begin(), b.e.end(), greater_than);
        b.e.pop_back();
        b.sorted = true;
        return;
    }
    // sorted descending; x is newer than every equal element, so it goes after them
    auto pos = std::upper_bound(b.e.begin(), b.e.end(), x, greater_than);
    b.e.insert(pos, x);
    b.e.pop_back();
}
}  // namespace

extern "C" int grip_leaderboard_scan(const float* probs, const int32_t* pred, const int64_t* path_rank,
                                     int64_t n, int c, int64_t k, int32_t* out_img, int32_t* out_class, int64_t* out_count) {
    GRIP_REQUIRE(probs && pred && path_rank && out_img && out_class && out_count, "leaderboard: null pointer");
    GRIP_REQUIRE(n >= 0 && c > 0 && k > 0, "leaderboard: bad sizes n=%lld c=%d k=%lld", (long long)n, c, (long long)k);
    try {
    const int64_t kk = std::min<int64_t>(k, std::max<int64_t>(n, 1));
    std::vector<Board> boards((size_t)c);
    for (auto& b : boards) b.e.reserve((size_t)kk + 1);
    for (int64_t i = 0; i < n; ++i) {
        const float* p = probs + i * c;
        const int js = pred[i];
        GRIP_REQUIRE(js >= 0 && js < c, "leaderboard: pred[%lld] = %d out of range", (long long)i, js);
        Board& own = boards[(size_t)js];
        const Entry x{p[js], path_rank[i], (int32_t)i};
        if ((int64_t)own.e.size() < kk || own.e.back().score < x.score) {
            offer(own, kk, x);
        } else {
            for (int j = 0; j < c; ++j) {
                if (j == js) continue;
                offer(boards[(size_t)j], kk, Entry{p[j], path_rank[i], (int32_t)i});
            }
        }
    }
    int64_t m = 0;
    for (int j = 0; j < c; ++j)
        for (const Entry& e : boards[(size_t)j].e) {
            out_img[m] = e.img;
            out_class[m] = j;
            ++m;
        }
    *out_count = m;
    return GRIP_OK;
    } catch (...) { grip_set_error("leaderboard: out of memory"); return GRIP_ERR_ARG; }
}
