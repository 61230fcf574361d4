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
#include <math.h>
#include <sched.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "host_common.h"

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

// ---------------------------------------------------------------------------------------------------------------------------
// Error-bounded scan (screen and refine).  Same algorithm, but row i's probabilities are only known to a relative accuracy
// rel_eps[i] (f16 towers; 0 = the row was re-encoded by the exact tower and is final): the true value of a score s lies in
// [s (1 - eps), s (1 + eps)].  The scan marks the un-refined rows it would need exactly in order to PROVE that its lists are
// the lists of the scan over the true values; the caller re-encodes them (eps = 0) and scans again.  Every round with a
// non-empty marked set refines at least one new row, so the loop ends.
//
// What has to be certain (reference lines: utils/clip_pseudolabels.py):
//  (A) the arg-max of a row (:39).  Candidates = classes whose interval reaches the largest value's.  If EVERY candidate's
//      board is full and certainly rejects the row it does not matter which of them is the true arg-max: the image spills to
//      every other class either way, the candidates reject it either way, the remaining classes see the same offer.
//      Otherwise the row is marked.
//  (B) while a board has never overflowed (unsorted append list, admission test against the LAST APPENDED element, :73-76): every
//      `board[-1].score < score` -- a rejected offer is lost for good and the first success changes the board's regime.
//  (C) the own-class decision of every image (:73-76 -> spill or not, :83): from the first overflow on a board is the sorted top-k
//      of everything it was offered since, so `board[-1].score` is the k-th largest TRUE value among those offers, which lies
//      between the k-th largest lower bound T_lo and the k-th largest upper bound T_hi of the offers.  Accept is certain when
//      T_hi < lo(x), reject when T_lo >= hi(x); otherwise x and the un-refined offers whose intervals meet x's are marked -- unless
//      every offer the spill would make is certainly irrelevant (below a sorted board's T_lo): then both outcomes leave the same
//      state (no other board sees the image, its own board was offered it either way) and nothing is marked; and when only some of
//      them could matter (and every board is sorted), they become CONDITIONAL offers that are checked against the final boards.
//  (D) the FINAL content and order of every board (:103-109).  Offers from other classes' spills (:83-101) need no certain
//      decision at the time they are made: an offer whose upper bound is below T_lo can never be among the k largest (k
//      offers are certainly above it) and is dropped; the others are recorded, and at the end the nominal board B must be
//      (a) certainly ordered pair by pair and (b) certainly above every recorded offer outside it.  Then B holds the k largest
//      true values, each of them was admitted when it arrived (fewer than k offers are above it) and never evicted, and the true
//      list order is B's.  Intermediate boards may differ from the true scan's; nothing reads them except (C), which uses the
//      bounds.  An exact tie between two FINAL (eps = 0) values across the boundary of a board is the one case this argument does not cover
//      (the reference's outcome then depends on arrival order); the scan then falls back to the strict form, in which every single
//      comparison of the literal algorithm is certified (mode 1 below).
//
// Two forms of the per-row bound (ABI 8, `bound_form`):
//  0  RELATIVE: |true - s| <= eps s + abs_eps, the interval [s (1 - eps) - abs_eps, s (1 + eps) + abs_eps].
//  1  LOG-ODDS: the ODDS s / (1 - s) of every entry of the row are within a factor [e^-delta, e^delta] of the true ones, delta = eps[row]:
//     true in [s / (s + (1 - s) e^delta), s / (s + (1 - s) e^-delta)].  This is the form an error of the cheaper tower's embedding direction
//     produces: it moves the LOGITS by scale x <de, t_c>, whatever the probabilities are, and softmax(l)_c = 1 / (1 + sum_c' exp(l_c' - l_c)) turns a
//     spread d of the logit errors over the classes into a factor within e^{+-d} on every entry's odds.  For s << 1 it is the relative bound
//     e^{+-delta}; for the s ~ 0.9+ entries that sit on the board thresholds of a peaked pool it is (1 - s) times tighter, which is what keeps such
//     pools from being re-encoded wholesale.  The f32 evaluation of the softmax (s and 1 - s as floats) is covered by a relative slack kR on s
//     and an absolute slack kU on 1 - s.  Everything below works on the intervals only and is the same for both forms.
namespace {
constexpr double kR = 1.0 / (1 << 20), kU = 1.0 / (1 << 20);
struct BEntry {
    double lo, hi;      // interval of the true score
    float score;        // nominal
    float eps;
    int64_t rank;
    int32_t img;
};
// How the values of one row bound the true ones.  abs_eps: absolute slack of an un-refined value -- below ~1e-30 a softmax output has no
// relative accuracy left (denormals, underflow to 0); final rows (eps == 0) carry none and their interval is the value itself.
struct RowBound {
    float eps = 0.f;            // relative bound (form 0) or delta (form 1); 0 = final
    int form = 0;
    double slack = 0.0;         // abs_eps of a non-final row
    double up = 1.0;            // hi(s) <= s * up + slack for every s >= 0 (the quick / float screens)
    double E = 1.0, invE = 1.0; // e^delta, e^-delta (form 1)
    RowBound() {}
    RowBound(float eps_, int form_, double abs_eps) : eps(eps_), form(form_) {
        if (eps == 0.f) return;
        slack = abs_eps;
        if (form == 0) {
            up = 1.0 + (double)eps;
        } else {
            E = std::exp((double)eps);
            invE = 1.0 / E;
            up = E * (1.0 + 8.0 * kR);           // >= E (1 + kR)^2 / (1 - kU)
        }
    }
    inline void interval(double s, double& lo, double& hi) const {
        if (eps == 0.f) { lo = hi = s; return; }
        if (form == 0) {
            lo = s * (1.0 - (double)eps) - slack;
            hi = s * (1.0 + (double)eps) + slack;
            return;
        }
        const double q = 1.0 - s;
        const double a_hi = s * (1.0 + kR), a_lo = s * (1.0 - kR);
        const double q_lo = std::max(q - kU, 0.0), q_hi = std::max(q, 0.0) + kU;
        hi = a_hi / (a_hi + q_lo * invE) * (1.0 + kR) + slack;
        lo = a_lo / (a_lo + q_hi * E) * (1.0 - kR) - slack;
    }
    inline double hi_of(double s) const { double lo, hi; interval(s, lo, hi); return hi; }
    inline double lo_of(double s) const { double lo, hi; interval(s, lo, hi); return lo; }
};
inline BEntry make_entry(const RowBound& rb, float s, int64_t rank, int32_t img) {
    BEntry e{0.0, 0.0, s, rb.eps, rank, img};
    rb.interval((double)s, e.lo, e.hi);
    return e;
}
// outcome of `a.score < x.score` over the true values: 1 certainly true, 0 certainly false, -1 undecidable
inline int certainly_less(const BEntry& a, const BEntry& x) {
    if (a.eps == 0.f && x.eps == 0.f) return a.score < x.score ? 1 : 0;
    if (a.hi < x.lo) return 1;
    if (a.lo >= x.hi) return 0;
    return -1;
}
inline bool nominal_greater(const BEntry& a, const BEntry& b) {
    if (a.score != b.score) return a.score > b.score;
    return a.rank > b.rank;
}
// is "a sorts before b" (score descending, ties by path descending) decided?  a is nominally before b.
inline bool order_certain(const BEntry& a, const BEntry& b) {
    if (a.eps == 0.f && b.eps == 0.f) return true;       // both final: the nominal comparison IS the true one
    return a.lo > b.hi;
}
// the k largest values seen so far (min-heap); kth() = -inf until k values were pushed
struct TopK {
    std::vector<double> h;
    size_t k = 0;
    double kth() const { return h.size() < k ? -1e300 : h.front(); }
    void push(double v) {
        if (h.size() < k) {
            h.push_back(v);
            std::push_heap(h.begin(), h.end(), std::greater<double>());
        } else if (v > h.front()) {
            std::pop_heap(h.begin(), h.end(), std::greater<double>());
            h.back() = v;
            std::push_heap(h.begin(), h.end(), std::greater<double>());
        }
    }
};
struct BBoard {
    std::vector<BEntry> e;      // the nominal board
    bool sorted = false;
    TopK lo, hi;                // bounds of the true k-th largest offer since the first overflow
    std::vector<BEntry> rec;    // offers since the first overflow that were not certainly irrelevant when they arrived
    std::vector<BEntry> cond;   // offers that are made only if an undecidable own-class decision of their image went "reject" (see (C))
    size_t prune_at = 0;
};
struct Marks {
    uint8_t* flag;
    int64_t count = 0;
    int cat = 0;                // what the scan is deciding right now (GRIP_SCAN_DEBUG prints the first-mark tally per category)
    int64_t by_cat[6] = {0, 0, 0, 0, 0, 0};
    inline void mark(const BEntry& x) {
        if (x.eps != 0.f && !flag[x.img]) { flag[x.img] = 1; ++count; ++by_cat[cat]; }
    }
};

// Parallel pre-filter of the bounded scan.  The scan itself is sequential (order-dependent), but 99.9 % of its O(n c) work is deciding, per
// (row, class), that an offer is certainly irrelevant (upper bound below the board's certain threshold T_lo) or that a class is no arg-max
// candidate.  T_lo only ever GROWS, so a test against an OLDER copy of it is a conservative filter: worker threads run it for the next block
// of rows (against the thresholds as they were when the block was requested) while the scan walks the current block, and the scan re-tests
// just the survivors against the current thresholds -- the same decisions as the full loops, in O(survivors) per row.  Under weak scaling
// every rank scans ALL N_total rows: 0.10 s -> 0.03 s per scan at N = 400 000 x C = 102 on 8 CPUs (tools/scan_scale.py).
struct Prefilter {
    static constexpr int64_t BLOCK = 8192;
    const float* probs; const int32_t* pred; const float* rel_eps; int64_t n; int c; double abs_eps; int form;
    struct Buf {
        int64_t lo = 0, hi = 0;                 // rows [lo, hi)
        // (plain arrays: a std::vector would zero-fill tens of megabytes per scan)
        std::unique_ptr<uint16_t[]> cand, spill;   // row r: entries [ (r - lo) * c, + n_cand / n_spill )
        std::unique_ptr<float[]> cand_p, spill_p;  // ... and their probabilities, gathered (the scan thread never touches the [n, c] matrix: one
        std::unique_ptr<float[]> own;              //     cache miss per row otherwise); own[r] = the row's own-class probability
        std::unique_ptr<int32_t[]> n_cand, n_spill;
        std::vector<double> t_snap;             // T_lo per class when the block was requested
        std::vector<float> t_snap_f;            // ... and floats at or below them (the workers' screen)
    } buf[2];
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    int gen = 0, pending = 0, which = 0;
    bool quit = false;
    int nthreads = 0;

    // Floats that bound a double from below / above with room for the float arithmetic of the screen (two roundings of 2^-24 against a margin of 1e-6)
    static float f_below(double t) {
        if (t < -3e38) return -INFINITY;
        if (!(t == t)) return t > 0 ? 0.f : -INFINITY;    // (NaN: never produced by the scan; screen everything in)
        float f = (float)(t - std::fabs(t) * 1e-6 - 1e-37);
        if ((double)f > t) f = std::nextafterf(f, -INFINITY);
        return f;
    }
    static float f_above(double t) {
        float f = (float)(t + std::fabs(t) * 1e-6 + 1e-37);
        if ((double)f < t) f = std::nextafterf(f, INFINITY);
        return f;
    }
    // Two passes per row: a branch-free float SCREEN over all c classes (auto-vectorised; a superset of both tests by construction of f_below /
    // f_above), then the exact double tests of the sequential scan on the few survivors -- the lists that come out are the ones the one-pass double
    // loop produced (r04: that loop, 102 convert-multiply-compare-branch steps per row, was what the scan thread waited for at N = 400 000).
    void work(const Buf& b_, int part, int parts) {
        Buf& b = const_cast<Buf&>(b_);
        const int64_t rows = b.hi - b.lo;
        const int64_t r0 = b.lo + rows * part / parts, r1 = b.lo + rows * (part + 1) / parts;
        const int c8 = (c + 7) & ~7;
        std::vector<uint8_t> flags((size_t)c8 + 8, 0);
        uint8_t* fl = flags.data();
        const float* tf = b.t_snap_f.data();
        for (int64_t i = r0; i < r1; ++i) {
            const float* p = probs + i * c;
            const int js = pred[i];
            const float eps = rel_eps[i];
            const RowBound rb(eps, form, abs_eps);
            const double slack = rb.slack;
            const double xlo = (js >= 0 && js < c) ? rb.lo_of((double)p[js]) : 0.0;
            uint16_t* cd = b.cand.get() + (i - b.lo) * c;
            uint16_t* sp = b.spill.get() + (i - b.lo) * c;
            float* cdp = b.cand_p.get() + (i - b.lo) * c;
            float* spp = b.spill_p.get() + (i - b.lo) * c;
            b.own[(size_t)(i - b.lo)] = (js >= 0 && js < c) ? p[js] : 0.f;
            const float uf = f_above(rb.up), sf = slack != 0.0 ? f_above(slack) : 0.f;
            const float xl = eps != 0.f ? f_below(xlo) : INFINITY;     // eps == 0: no arg-max candidates at all
            for (int j = 0; j < c; ++j) {
                const float h = p[j] * uf + sf;
                fl[j] = (uint8_t)((!(h < xl) ? 1 : 0) | (!(h < tf[j]) ? 2 : 0) | (p[j] < 0.f ? 3 : 0));     // (a negative input: the float bounds assume p >= 0)
            }
            int nc = 0, ns = 0;
            for (int j8 = 0; j8 < c8; j8 += 8) {
                uint64_t w;
                memcpy(&w, fl + j8, 8);
                if (!w) continue;
                for (int j = j8; j < j8 + 8 && j < c; ++j) {
                    if (!fl[j] || j == js) continue;
                    const double hi = rb.hi_of((double)p[j]);
                    if (eps != 0.f && hi >= xlo) { cd[nc] = (uint16_t)j; cdp[nc++] = p[j]; }
                    if (!(hi < b.t_snap[(size_t)j])) { sp[ns] = (uint16_t)j; spp[ns++] = p[j]; }
                }
            }
            b.n_cand[(size_t)(i - b.lo)] = nc;
            b.n_spill[(size_t)(i - b.lo)] = ns;
        }
    }
    void start(int threads) {
        nthreads = threads;
        for (auto& b : buf) {
            const size_t rows = (size_t)std::min<int64_t>(BLOCK, n), m = rows * (size_t)c;
            b.cand.reset(new uint16_t[m]); b.spill.reset(new uint16_t[m]);
            b.cand_p.reset(new float[m]); b.spill_p.reset(new float[m]);
            b.n_cand.reset(new int32_t[rows]); b.n_spill.reset(new int32_t[rows]); b.own.reset(new float[rows]);
        }
        for (int t = 1; t < threads; ++t)
            workers.emplace_back([this, t] {
                int seen = 0;
                for (;;) {
                    int w;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return quit || gen != seen; });
                        if (quit) return;
                        seen = gen;
                        w = which;
                    }
                    work(buf[w], t - 1, nthreads - 1);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    // ask for the block starting at row lo into buffer w, filtered against the thresholds t_lo as they are NOW
    void request(int w, int64_t lo, const std::vector<double>& t_lo) {
        Buf& b = buf[w];
        b.lo = lo; b.hi = std::min(n, lo + BLOCK);
        b.t_snap = t_lo;
        b.t_snap_f.resize(t_lo.size());
        for (size_t j = 0; j < t_lo.size(); ++j) b.t_snap_f[j] = f_below(t_lo[j]);
        {
            std::lock_guard<std::mutex> lk(mu);
            which = w; pending = nthreads - 1; ++gen;
        }
        cv_go.notify_all();               // (the calling thread goes on scanning the previous block meanwhile)
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~Prefilter() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (auto& t : workers) t.join();
    }
};

struct BoundedScan {
    const float* probs; const int32_t* pred; const int64_t* path_rank; const float* rel_eps;
    int64_t n; int c; int64_t kk; bool strict; double abs_eps;
    int form = 0;                   // 0 relative, 1 log-odds (RowBound)
    int threads = 1;                // > 1: the parallel pre-filter above
    std::vector<BBoard> boards;
    std::vector<double> t_lo;       // per class: T_lo once the board is sorted, -inf before (nothing is dropped then)
    Marks mk;
    bool need_strict = false;

    void record(BBoard& b, int j, const BEntry& x) {
        b.rec.push_back(x);
        b.lo.push(x.lo);
        b.hi.push(x.hi);
        t_lo[(size_t)j] = b.lo.kth();
        if (b.rec.size() >= b.prune_at) {       // forget what has become certainly irrelevant since
            const double t = t_lo[(size_t)j];
            size_t w = 0;
            for (size_t r = 0; r < b.rec.size(); ++r)
                if (!(b.rec[r].hi < t)) b.rec[w++] = b.rec[r];
            b.rec.resize(w);
            b.prune_at = std::max<size_t>(4 * (size_t)kk + 64, 2 * w);
        }
    }
    void nominal_insert(BBoard& b, const BEntry& x) {
        if (!(b.e.back().score < x.score)) return;
        auto pos = std::upper_bound(b.e.begin(), b.e.end(), x, nominal_greater);
        b.e.insert(pos, x);
        b.e.pop_back();
    }
    // an offer to board j (its own class after an accept, or a spill)
    void offer(int j, const BEntry& x) {
        BBoard& b = boards[(size_t)j];
        if ((int64_t)b.e.size() < kk) { b.e.push_back(x); return; }
        if (b.sorted && !strict) {                                   // (D)
            if (x.hi < t_lo[(size_t)j]) return;
            record(b, j, x);
            nominal_insert(b, x);
            return;
        }
        const int lt = certainly_less(b.e.back(), x);                // (B), and every comparison in strict mode
        if (lt < 0) { mk.mark(b.e.back()); mk.mark(x); }
        const bool accept = lt >= 0 ? lt == 1 : b.e.back().score < x.score;
        if (!accept) return;
        if (!b.sorted) {
            b.e.push_back(x);
            std::stable_sort(b.e.begin(), b.e.end(), nominal_greater);
            if (strict) {
                for (size_t i = 0; i + 1 < b.e.size(); ++i)
                    if (!order_certain(b.e[i], b.e[i + 1])) { mk.mark(b.e[i]); mk.mark(b.e[i + 1]); }
            } else {                                                 // the k + 1 elements are the first offers of the sorted regime
                b.lo.k = b.hi.k = (size_t)kk;
                b.prune_at = 4 * (size_t)kk + 64;
                for (const BEntry& y : b.e) record(b, j, y);
            }
            b.e.pop_back();
            b.sorted = true;
            return;
        }
        auto pos = std::upper_bound(b.e.begin(), b.e.end(), x, nominal_greater);     // strict mode, sorted board
        if (pos != b.e.begin() && !order_certain(*(pos - 1), x)) { mk.mark(*(pos - 1)); mk.mark(x); }
        if (pos != b.e.end() && !order_certain(x, *pos)) { mk.mark(*pos); mk.mark(x); }
        b.e.insert(pos, x);
        b.e.pop_back();
    }
    // does board j certainly reject x?  (full board required)
    bool certainly_rejects(int j, const BEntry& x) const {
        const BBoard& b = boards[(size_t)j];
        if ((int64_t)b.e.size() < kk) return false;
        if (b.sorted && !strict) return t_lo[(size_t)j] >= x.hi;
        return certainly_less(b.e.back(), x) == 0;
    }

    int run(int32_t* out_img, int32_t* out_class, int64_t* out_count, uint8_t* ambiguous, int64_t* n_ambiguous, int64_t k) {
        const bool label_all = k == 10000000;      // utils/clip_pseudolabels.py:27-44: every image under its arg-max, no boards
        boards.assign((size_t)c, BBoard());
        t_lo.assign((size_t)c, -1e300);
        if (!label_all) for (auto& b : boards) b.e.reserve((size_t)kk + 2);
        std::vector<int> cand;
        std::vector<float> cand_p;
        cand.reserve((size_t)c);
        cand_p.reserve((size_t)c);
        memset(ambiguous, 0, (size_t)n);
        mk = Marks{ambiguous};
        need_strict = false;
        // the parallel pre-filter (threads > 1): block b + 1 is filtered by the workers while this thread scans block b
        std::unique_ptr<Prefilter> pf;
        if (threads > 1 && n >= 2 * Prefilter::BLOCK && c <= 65535) {
            pf.reset(new Prefilter{probs, pred, rel_eps, n, c, abs_eps, form});
            pf->start(threads);
            pf->request(0, 0, t_lo);
        }
        int n_sorted = 0;           // boards in the sorted regime (for the "every other board is sorted" tests)
        const bool dbg_t = getenv("GRIP_SCAN_DEBUG") != nullptr;
        double t_wait = 0.0;
        const auto t_begin = std::chrono::steady_clock::now();
        std::vector<uint16_t> all_j;
        for (int64_t i = 0; i < n; ++i) {
            const float* p = probs + i * c;
            const int js = pred[i];
            const float eps = rel_eps[i];
            GRIP_REQUIRE(js >= 0 && js < c, "bounded leaderboard: pred[%lld] = %d out of range", (long long)i, js);
            GRIP_REQUIRE(eps >= 0.f && eps < (form == 0 ? 1e6f : 700.f), "bounded leaderboard: rel_eps[%lld] = %g out of range", (long long)i, (double)eps);    // (relative eps >= 1: the lower bound is <= 0, i.e. "could be anything below"; log-odds: e^delta must fit a double)
            // this row's survivors of the pre-filter: arg-max candidates, and classes whose offer was not certainly irrelevant when the block was requested
            const uint16_t* cd = nullptr; const uint16_t* sp = nullptr;
            const float* cdp = nullptr; const float* spp = nullptr;
            float p_own;
            int ncd = -1, nsp = -1;
            if (pf) {
                const int w = (int)((i / Prefilter::BLOCK) & 1);
                if (i % Prefilter::BLOCK == 0) {
                    const auto tw = std::chrono::steady_clock::now();
                    pf->wait();                                                                     // block i / BLOCK is ready
                    if (dbg_t) t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
                    if (i + Prefilter::BLOCK < n) pf->request(w ^ 1, i + Prefilter::BLOCK, t_lo);   // ... and the next one starts, against the thresholds of now
                }
                const Prefilter::Buf& b = pf->buf[w];
                const int64_t r = i - b.lo;
                cd = b.cand.get() + r * c; ncd = b.n_cand[(size_t)r]; cdp = b.cand_p.get() + r * c;
                sp = b.spill.get() + r * c; nsp = b.n_spill[(size_t)r]; spp = b.spill_p.get() + r * c;
                p_own = b.own[(size_t)r];
            } else {
                p_own = p[js];
            }
            const RowBound rb(eps, form, abs_eps);
            const BEntry x = make_entry(rb, p_own, path_rank[i], (int32_t)i);
            const double slack = rb.slack;
            cand.clear();                                                   // (A)
            cand_p.clear();
            if (eps != 0.f) {
                if (cd) { cand.assign(cd, cd + ncd); cand_p.assign(cdp, cdp + ncd); }
                else
                    for (int j = 0; j < c; ++j)
                        if (j != js && rb.hi_of((double)p[j]) >= x.lo) { cand.push_back(j); cand_p.push_back(p[j]); }
            }
            if (label_all) {
                if (!cand.empty()) mk.mark(x);
                continue;
            }
            // classes j != js whose offer p[j] is not certainly irrelevant NOW (the pre-filter's survivors re-tested against the current thresholds;
            // without a pre-filter: every class): visit(j) for each, in ascending j
            auto for_live = [&](auto&& visit) {
                if (sp) {
                    for (int q = 0; q < nsp; ++q) {
                        const int j = sp[q];
                        if (!(rb.hi_of((double)spp[q]) < t_lo[(size_t)j])) visit(j, spp[q]);
                    }
                } else {
                    for (int j = 0; j < c; ++j) {
                        if (j == js) continue;
                        const double t = t_lo[(size_t)j];
                        if (p[j] >= 0.f && (double)p[j] * rb.up + slack < t) continue;      // (quick: hi(p) <= p up + slack)
                        if (!(rb.hi_of((double)p[j]) < t)) visit(j, p[j]);
                    }
                }
            };
            BBoard& own = boards[(size_t)js];
            const bool own_was_sorted = own.sorted;
            if (!cand.empty()) {
                bool all_reject = certainly_rejects(js, x);
                for (size_t q = 0; all_reject && q < cand.size(); ++q)
                    all_reject = certainly_rejects(cand[q], make_entry(rb, cand_p[q], x.rank, x.img));
                mk.cat = 1;
                if (!all_reject) mk.mark(x);
            }
            mk.cat = 2;
            bool spill;
            if ((int64_t)own.e.size() < kk) {
                spill = false;
            } else if (own.sorted && !strict) {                             // (C)
                spill = !(own.e.back().score < x.score);
                const bool sure_accept = own.hi.kth() < x.lo, sure_reject = t_lo[(size_t)js] >= x.hi;
                if (!sure_accept && !sure_reject) {
                    // Undecidable -- but if every offer the spill would make is certainly irrelevant (below the certain threshold of a
                    // sorted board), the two outcomes leave the same state: no other board sees the image either way, and its own board
                    // was OFFERED it either way (the k-th largest offer, hence every later threshold, counts it in both cases; whether it
                    // is in the final board is certified at the end like any recorded offer).
                    bool spill_irrelevant = true;         // (an unsorted board has T_lo = -inf: its offer is live)
                    for_live([&](int, float) { spill_irrelevant = false; });
                    // an unsorted board loses a rejected offer for good: no deferral unless every OTHER board is sorted
                    const bool can_defer = eps != 0.f && (n_sorted - (own.sorted ? 1 : 0) == c - 1);
                    if (spill_irrelevant) {
                        spill = false;          // record it with its own board (nominal_insert keeps or rejects it nominally)
                    } else if (can_defer) {
                        // The spill's offers become CONDITIONAL: they are made only in the world where the own class rejected the image.
                        // They stay out of every board and out of T_lo (a k-th largest lower bound over fewer offers is still a lower
                        // bound), count towards T_hi (an upper bound over more offers is still an upper bound), and at the end each must
                        // be certainly below its board's last element -- then the final boards are the same in both worlds; otherwise the
                        // image is marked (it is un-refined: eps != 0).
                        spill = false;
                        for_live([&](int j, float pj) {
                            BBoard& b = boards[(size_t)j];
                            const BEntry y = make_entry(rb, pj, x.rank, x.img);
                            b.cond.push_back(y);
                            b.hi.push(y.hi);
                        });
                    } else {
                        mk.mark(x);
                        for (const BEntry& y : own.rec)
                            if (y.eps != 0.f && y.lo <= x.hi && y.hi >= x.lo) mk.mark(y);
                    }
                }
            } else {
                const int lt = certainly_less(own.e.back(), x);
                if (lt < 0) { mk.mark(own.e.back()); mk.mark(x); }
                spill = !(lt >= 0 ? lt == 1 : own.e.back().score < x.score);
            }
            mk.cat = 3;
            if (!spill) {
                offer(js, x);
                if (!own_was_sorted && own.sorted) ++n_sorted;
            } else {
                // (an offer can move its own board's threshold only: the live test of class j does not depend on the offers made to the others)
                for_live([&](int j, float pj) {
                    BBoard& b = boards[(size_t)j];
                    const bool was = b.sorted;
                    offer(j, make_entry(rb, pj, x.rank, x.img));
                    if (!was && b.sorted) ++n_sorted;
                });
            }
        }
        pf.reset();
        if (dbg_t) fprintf(stderr, "bounded scan: row loop %.1f ms (of which waiting for the pre-filter %.1f ms), %d threads\n",
                           std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3, t_wait * 1e3, threads);
        if (!label_all && !strict) {                                        // (D): certify the final boards
            mk.cat = 4;
            std::vector<uint8_t> in_board((size_t)std::max<int64_t>(n, 1), 0);
            for (int j = 0; j < c; ++j) {
                BBoard& b = boards[(size_t)j];
                if (!b.sorted) continue;
                for (size_t i = 0; i + 1 < b.e.size(); ++i)
                    if (!order_certain(b.e[i], b.e[i + 1])) { mk.mark(b.e[i]); mk.mark(b.e[i + 1]); }
                for (const BEntry& y : b.e) in_board[(size_t)y.img] = 1;
                const BEntry& last = b.e.back();
                const double t = t_lo[(size_t)j];
                for (const BEntry& y : b.rec) {
                    if (y.hi < t || in_board[(size_t)y.img]) continue;
                    if (y.eps == 0.f && last.eps == 0.f) {
                        if (!(y.score < last.score)) need_strict = true;    // an exact tie of two final values across the boundary
                    } else if (!(y.hi < last.lo)) {
                        mk.mark(last);
                        if (last.eps == 0.f || y.hi >= (double)last.score) mk.mark(y);   // the rest waits until `last` is final
                    }
                }
                for (const BEntry& y : b.cond)
                    if (!(y.hi < last.lo)) { mk.mark(y); if (y.hi < (double)last.score) mk.mark(last); }
                for (const BEntry& y : b.e) in_board[(size_t)y.img] = 0;
            }
        }
        int64_t m = 0;
        if (label_all) {
            for (int64_t i = 0; i < n; ++i) { out_img[m] = (int32_t)i; out_class[m] = pred[i]; ++m; }
        } else {
            for (int j = 0; j < c; ++j)
                for (const BEntry& e : boards[(size_t)j].e) {
                    out_img[m] = e.img;
                    out_class[m] = j;
                    ++m;
                }
        }
        *out_count = m;
        *n_ambiguous = mk.count;
        if (getenv("GRIP_SCAN_DEBUG"))
            fprintf(stderr, "bounded scan marks: arg-max %lld, own-class decision %lld, unsorted-regime offers %lld, final boards %lld (strict %d)\n",
                    (long long)mk.by_cat[1], (long long)mk.by_cat[2], (long long)mk.by_cat[3], (long long)mk.by_cat[4], (int)strict);
        return GRIP_OK;
    }
};
}  // namespace

extern "C" int grip_leaderboard_scan_bounded(const float* probs, const int32_t* pred, const int64_t* path_rank, const float* rel_eps, float abs_eps,
                                             int bound_form, int threads, int64_t n, int c, int64_t k, int32_t* out_img, int32_t* out_class,
                                             int64_t* out_count, uint8_t* ambiguous, int64_t* n_ambiguous) {
    GRIP_REQUIRE(probs && pred && path_rank && rel_eps && out_img && out_class && out_count && ambiguous && n_ambiguous, "bounded leaderboard: null pointer");
    GRIP_REQUIRE(n >= 0 && c > 0 && k > 0, "bounded leaderboard: bad sizes n=%lld c=%d k=%lld", (long long)n, c, (long long)k);
    GRIP_REQUIRE(abs_eps >= 0.f && abs_eps < 1.f, "bounded leaderboard: abs_eps = %g out of range", (double)abs_eps);
    GRIP_REQUIRE(bound_form == 0 || bound_form == 1, "bounded leaderboard: bound_form = %d (0 relative, 1 log-odds)", bound_form);
    GRIP_REQUIRE(threads >= 0, "bounded leaderboard: threads = %d", threads);
    try {
        BoundedScan s{probs, pred, path_rank, rel_eps, n, c, std::min<int64_t>(k, std::max<int64_t>(n, 1)), false, (double)abs_eps};
        const char* env = getenv("GRIP_SCAN_STRICT");       // developer A/B: certify every comparison of the literal algorithm
        s.strict = env && env[0] == '1';
        s.form = bound_form;
        if (threads > 0) {          // the caller's choice (a root-placed scan of a multi-rank pass uses the whole node's quota)
            s.threads = std::min(threads, 16);
            if ((int64_t)n * c < (int64_t)4 << 20) s.threads = 1;
        } else {   // threads of the parallel pre-filter: $GRIP_SCAN_THREADS, else the CPUs this process may use -- affinity mask, capped by the cgroup CPU quota
            // (the MI355X boxes show 256 logical CPUs under a 16-CPU quota), divided among the ranks of the node ($LOCAL_WORLD_SIZE: every rank runs the
            // replicated scan at the same moment) -- at most 16; small problems run on one
            const char* te = getenv("GRIP_SCAN_THREADS");
            int t = te ? atoi(te) : (int)std::thread::hardware_concurrency();
            if (!te) {
                cpu_set_t set;
                if (sched_getaffinity(0, sizeof(set), &set) == 0) t = std::min(t, CPU_COUNT(&set));
                if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
                    char quota[32] = {0};
                    long period = 0;
                    if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
                        t = std::min<long>(t, std::max<long>(1, atol(quota) / period));
                    fclose(f);
                }
                const char* lw = getenv("LOCAL_WORLD_SIZE");
                if (lw && atoi(lw) > 1) t = std::max(1, t / atoi(lw));
            }
            s.threads = std::max(1, std::min(t, 16));
            if ((int64_t)n * c < (int64_t)4 << 20) s.threads = 1;
        }
        int rc = s.run(out_img, out_class, out_count, ambiguous, n_ambiguous, k);
        if (rc == GRIP_OK && s.need_strict && *n_ambiguous == 0) {
            if (getenv("GRIP_SCAN_DEBUG")) fprintf(stderr, "strict-fallback\n");
            s.strict = true;
            rc = s.run(out_img, out_class, out_count, ambiguous, n_ambiguous, k);
        }
        return rc;
    } catch (...) { grip_set_error("bounded leaderboard: out of memory"); return GRIP_ERR_ARG; }
}
