/* ORACLE (test infrastructure) -- plain-C restatement of the sequential per-class
 * leaderboard, utils/clip_pseudolabels.py:49-112 of the reference (same scan in the nine
 * assign_pseudo_labels, e.g. methods/transductive_zsl/multimodal_fpl.py:194-285).
 * Literal on purpose: unsorted append while a board fills (:73-74), admission test against the
 * LAST list element with strict '<' (:75), sorted(board + [new], reverse=True)[:k] with ties
 * broken by the path string, descending (:79-82), and the walk over every other class in
 * descending-probability order without a break (:83-101).  Pinned by tests/golden/leaderboard.json
 * (outputs of the reference function itself).  Never linked into the product library.
 *
 * Build: gcc -O2 -shared -fPIC -o _build/libleaderboard_ref.so leaderboard_ref.c
 */
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int img; } entry_t;

/* Python tuple order on (score, path): a > b ? */
static int gt(const entry_t *a, const entry_t *b, const char *const *paths) {
    if (a->score != b->score) return a->score > b->score;
    return strcmp(paths[a->img], paths[b->img]) > 0;
}

/* stable descending sort (equal tuples keep list order, as Python's sorted(reverse=True)) */
static void sort_desc(entry_t *e, int n, const char *const *paths) {
    for (int i = 1; i < n; ++i) {
        entry_t x = e[i];
        int j = i - 1;
        while (j >= 0 && gt(&x, &e[j], paths)) { e[j + 1] = e[j]; --j; }
        e[j + 1] = x;
    }
}

static void offer(entry_t *board, int *len, int k, float score, int img, const char *const *paths) {
    if (*len < k) {                                    /* :73-74 / :91-92 */
        board[*len].score = score; board[*len].img = img; ++*len;
    } else if (board[*len - 1].score < score) {        /* :75 / :93 */
        board[*len].score = score; board[*len].img = img;
        sort_desc(board, *len + 1, paths);             /* :79-82 / :95-99, [:k] drops the tail */
    }
}

typedef struct { float p; int j; } pc_t;
static int pc_desc(const void *a, const void *b) {    /* sorted([(p, j)], reverse=True): p desc, j desc */
    const pc_t *x = (const pc_t *)a, *y = (const pc_t *)b;
    if (x->p != y->p) return x->p > y->p ? -1 : 1;
    return y->j - x->j;
}

/* probs [N*C] row-major fp32, pred [N] arg-max chosen by the caller, paths [N] C strings.
 * out_img/out_class: capacity C*k; returns the number of emitted (image, class) pairs, boards
 * concatenated in class order (:103-109).  k must be < 10000000 (that branch is trivial). */
int leaderboard_ref(const float *probs, const int *pred, const char *const *paths,
                    int N, int C, int k, int *out_img, int *out_class) {
    entry_t *boards = (entry_t *)malloc(sizeof(entry_t) * (size_t)C * (size_t)(k + 1));
    int *len = (int *)calloc((size_t)C, sizeof(int));
    pc_t *order = (pc_t *)malloc(sizeof(pc_t) * (size_t)C);
    for (int i = 0; i < N; ++i) {
        const float *p = probs + (size_t)i * C;
        int js = pred[i];
        entry_t *b = boards + (size_t)js * (k + 1);
        if (len[js] < k || b[len[js] - 1].score < p[js]) {
            offer(b, &len[js], k, p[js], i, paths);
        } else {
            int n = 0;
            for (int j = 0; j < C; ++j) if (j != js) { order[n].p = p[j]; order[n].j = j; ++n; }
            qsort(order, (size_t)n, sizeof(pc_t), pc_desc);
            for (int t = 0; t < n; ++t) {
                int j = order[t].j;
                offer(boards + (size_t)j * (k + 1), &len[j], k, p[j], i, paths);
            }
        }
    }
    int m = 0;
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < len[c]; ++t) { out_img[m] = boards[(size_t)c * (k + 1) + t].img; out_class[m] = c; ++m; }
    free(boards); free(len); free(order);
    return m;
}
