/*
 * oracle/ivtree.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's interval treap,
 * lib/bx/intervals/intersection.pyx (bx-python 0.14.0).  It is the checker the
 * HIP path is compared against, and the "port" CPU baseline timed by bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under bx-python_amd/ may.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * vectors produced by importing the real Cython reference in the build
 * container (oracle/gen_golden.py) and against the known answers of the
 * reference's own tests (intersection_tests.py, the doctests in
 * intersection.pyx:335-376).
 *
 * Payloads are insertion indices (0,1,2,...): the reference stores Python
 * objects, which the host wrapper keeps in a list addressed by that index.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct Node {
    float priority;
    int32_t start, end;
    int32_t minend, maxend, minstart;
    int32_t payload;
    struct Node *left, *right;
} Node;

typedef struct {
    Node *root;
    /* slab allocator so 10M-node trees do not pay malloc per node */
    Node **slabs;
    int64_t nslabs, used_in_slab, count;
} IvTree;

#define SLAB 65536

/* intersection.pyx:55  nlog = -1.0 / log(0.5) */
static double nlog_const(void) { return -1.0 / log(0.5); }

IvTree *ivt_new(void)
{
    IvTree *t = (IvTree *)calloc(1, sizeof(IvTree));
    return t;
}

void ivt_free(IvTree *t)
{
    if (!t) return;
    for (int64_t i = 0; i < t->nslabs; i++) free(t->slabs[i]);
    free(t->slabs);
    free(t);
}

int64_t ivt_size(const IvTree *t) { return t->count; }

/* intersection.pyx:87-100  IntervalNode.__cinit__ */
static Node *node_new(IvTree *t, int32_t start, int32_t end)
{
    if (t->nslabs == 0 || t->used_in_slab == SLAB) {
        t->slabs = (Node **)realloc(t->slabs, (size_t)(t->nslabs + 1) * sizeof(Node *));
        t->slabs[t->nslabs++] = (Node *)malloc(SLAB * sizeof(Node));
        t->used_in_slab = 0;
    }
    Node *n = &t->slabs[t->nslabs - 1][t->used_in_slab++];
    /* :92  priority = ceil(nlog * log(-1.0/(1.0 * rand()/RAND_MAX - 1))) */
    n->priority = (float)ceil(nlog_const() * log(-1.0 / (1.0 * rand() / RAND_MAX - 1)));
    n->start = start;
    n->end = end;
    n->maxend = end;
    n->minstart = start;
    n->minend = end;
    n->payload = (int32_t)t->count++;
    n->left = n->right = NULL;
    return n;
}

static int32_t max2(int32_t a, int32_t b) { return b > a ? b : a; }
static int32_t min2(int32_t a, int32_t b) { return b < a ? b : a; }

/* intersection.pyx:154-166  set_ends: note a leaf keeps whatever it had */
static void set_ends(Node *n)
{
    if (n->right && n->left) {
        n->maxend = max2(n->end, max2(n->right->maxend, n->left->maxend));
        n->minend = min2(n->end, min2(n->right->minend, n->left->minend));
        n->minstart = min2(n->start, min2(n->right->minstart, n->left->minstart));
    } else if (n->right) {
        n->maxend = max2(n->end, n->right->maxend);
        n->minend = min2(n->end, n->right->minend);
        n->minstart = min2(n->start, n->right->minstart);
    } else if (n->left) {
        n->maxend = max2(n->end, n->left->maxend);
        n->minend = min2(n->end, n->left->minend);
        n->minstart = min2(n->start, n->left->minstart);
    }
}

/* intersection.pyx:140-145 */
static Node *rotate_right(Node *self)
{
    Node *root = self->left;
    self->left = root->right;
    root->right = self;
    set_ends(self);
    return root;
}

/* intersection.pyx:147-152 */
static Node *rotate_left(Node *self)
{
    Node *root = self->right;
    self->right = root->left;
    root->left = self;
    set_ends(self);
    return root;
}

/* intersection.pyx:103-138  IntervalNode.insert -- returns the new subtree root */
static Node *node_insert(IvTree *t, Node *self, int32_t start, int32_t end)
{
    Node *root = self;
    /* :111-114: on equal starts the *end* is compared against self.start */
    int32_t decision = start;
    if (start == self->start) decision = end;

    if (decision > self->start) {
        if (self->right)
            self->right = node_insert(t, self->right, start, end);
        else
            self->right = node_new(t, start, end);
        if (self->priority < self->right->priority) root = rotate_left(self);
    } else {
        if (self->left)
            self->left = node_insert(t, self->left, start, end);
        else
            self->left = node_new(t, start, end);
        if (self->priority < self->left->priority) root = rotate_right(self);
    }
    set_ends(root);
    return root;
}

/* intersection.pyx:388-395  IntervalTree.insert */
void ivt_insert(IvTree *t, int32_t start, int32_t end)
{
    if (!t->root)
        t->root = node_new(t, start, end);
    else
        t->root = node_insert(t, t->root, start, end);
}

void ivt_insert_many(IvTree *t, const int32_t *start, const int32_t *end, int64_t n)
{
    for (int64_t i = 0; i < n; i++) ivt_insert(t, start[i], end[i]);
}

typedef struct {
    int32_t *out;
    int64_t cap, n;
} Sink;

static void sink_push(Sink *s, int32_t v)
{
    if (s->out && s->n < s->cap) s->out[s->n] = v;
    s->n++;
}

/* intersection.pyx:180-189  _intersect (in-order DFS with the two prunes) */
static void node_intersect(const Node *n, int32_t start, int32_t end, Sink *s)
{
    if (n->left && n->left->maxend > start) node_intersect(n->left, start, end, s);
    if (n->end > start && n->start < end) sink_push(s, n->payload);
    if (n->right && n->start < end) node_intersect(n->right, start, end, s);
}

/* intersection.pyx:400-406  IntervalTree.find.  Writes up to cap payload
 * indices into out (may be NULL) and returns the total number of hits. */
int64_t ivt_find(const IvTree *t, int32_t start, int32_t end, int32_t *out, int64_t cap)
{
    Sink s = {out, cap, 0};
    if (t->root) node_intersect(t->root, start, end, &s);
    return s.n;
}

/* Batched find()+len(), the shape the reference is benchmarked in. */
void ivt_count_batch(const IvTree *t, const int32_t *qs, const int32_t *qe, int64_t n,
                     int32_t *out, int64_t *total)
{
    int64_t tot = 0;
    for (int64_t i = 0; i < n; i++) {
        Sink s = {NULL, 0, 0};
        if (t->root) node_intersect(t->root, qs[i], qe[i], &s);
        if (out) out[i] = (int32_t)s.n;
        tot += s.n;
    }
    if (total) *total = tot;
}

/* Batched find() into CSR: offsets[n+1] (int64), hits[cap]. Returns total hits. */
int64_t ivt_find_batch(const IvTree *t, const int32_t *qs, const int32_t *qe, int64_t n,
                       int64_t *offsets, int32_t *hits, int64_t cap)
{
    int64_t tot = 0;
    for (int64_t i = 0; i < n; i++) {
        offsets[i] = tot;
        Sink s = {hits ? hits + (tot < cap ? tot : cap) : NULL, cap > tot ? cap - tot : 0, 0};
        if (t->root) node_intersect(t->root, qs[i], qe[i], &s);
        tot += s.n;
    }
    offsets[n] = tot;
    return tot;
}

/* intersection.pyx:192-209  _seek_left */
static void seek_left(const Node *n, int32_t position, Sink *s, int32_t max_dist)
{
    if ((int64_t)n->maxend + max_dist < position) return;
    if (n->minstart > position) return;
    if (n->right) seek_left(n->right, position, s, max_dist);
    int64_t d = (int64_t)position - n->end;
    if (-1 < d && d < max_dist) sink_push(s, n->payload);
    if (n->left) seek_left(n->left, position, s, max_dist);
}

/* intersection.pyx:213-229  _seek_right */
static void seek_right(const Node *n, int32_t position, Sink *s, int32_t max_dist)
{
    if (n->maxend < position) return;
    if ((int64_t)n->minstart - max_dist > position) return;
    if (n->left) seek_right(n->left, position, s, max_dist);
    int64_t d = (int64_t)n->start - position;
    if (-1 < d && d < max_dist) sink_push(s, n->payload);
    if (n->right) seek_right(n->right, position, s, max_dist);
}

/* Raw candidate lists of left()/right() (intersection.pyx:232-260) before the
 * "if len(results) == n ... else sort and truncate" step, which the Python side
 * of the oracle applies (it needs the start/end of each payload). */
int64_t ivt_seek_left(const IvTree *t, int32_t position, int32_t max_dist, int32_t *out, int64_t cap)
{
    Sink s = {out, cap, 0};
    if (t->root) seek_left(t->root, position - 1, &s, max_dist); /* :240 */
    return s.n;
}

int64_t ivt_seek_right(const IvTree *t, int32_t position, int32_t max_dist, int32_t *out, int64_t cap)
{
    Sink s = {out, cap, 0};
    if (t->root) seek_right(t->root, position + 1, &s, max_dist); /* :255 */
    return s.n;
}

/* intersection.pyx:262-268  traverse: in-order payload sequence */
static void node_traverse(const Node *n, Sink *s)
{
    if (n->left) node_traverse(n->left, s);
    sink_push(s, n->payload);
    if (n->right) node_traverse(n->right, s);
}

int64_t ivt_traverse(const IvTree *t, int32_t *out, int64_t cap)
{
    Sink s = {out, cap, 0};
    if (t->root) node_traverse(t->root, &s);
    return s.n;
}
