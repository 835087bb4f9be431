/*
 * oracle/cluster.c -- CPU restatement of the reference's ClusterTree (TEST INFRASTRUCTURE ONLY:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/).
 *
 * Reference: src/cluster.c.  An interval joins the cluster it lands in when
 *     !(start - max_dist > node->end) && !(end + max_dist < node->start)          (cluster.c:229-236)
 * and cluster_fixup (:112-147) then absorbs every neighbouring cluster the widened range reaches
 * (maxstart - minend <= max_dist, :120).  For max_dist >= 0 (and for max_dist = -1 on intervals of positive
 * length) the final partition is the set of connected components of "gap <= max_dist", independent of the
 * insertion order; walking the intervals by start it is: open a new cluster when start - max_dist > largest end so far.
 * getregions() (lib/bx/intervals/cluster.pyx:74-98) lists the clusters by ascending start with their ids
 * sorted; the min_intervals filter (cluster.c:190) is applied by the caller.
 *
 * Pinned by tests/test_oracle_golden.py against vectors produced by the reference's own extension
 * (oracle/gen_golden_ops.py) and against oracle/_ref/libcluster_ref.so (the reference's cluster.c
 * compiled in place).
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    int32_t start, end, id;
    int64_t pos;
} item_t;

static int by_start(const void *a, const void *b)
{
    const item_t *x = (const item_t *)a, *y = (const item_t *)b;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

static int by_id(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : (x > y);
}

/* Returns the number of clusters; c_start/c_end need n entries, c_off n + 1, members n.  -1 on bad input. */
int64_t oracle_clusters(const int32_t *start, const int32_t *end, const int32_t *ids, int64_t n, int32_t max_dist,
                        int32_t *c_start, int32_t *c_end, int64_t *c_off, int32_t *members)
{
    /* max_dist = -1 ("overlap by one base or more") has an answer while every interval has a positive length: the same
     * sweep (oracle/cluster_negative_distance.py); below -1, or -1 with a zero-length interval, the reference's result
     * depends on the insertion order and on rand(): -1 = bad input. */
    if (n < 0 || max_dist < -1) return -1;
    if (max_dist < 0)
        for (int64_t i = 0; i < n; i++)
            if (end[i] <= start[i]) return -1;
    if (n == 0) {
        c_off[0] = 0;
        return 0;
    }
    item_t *it = (item_t *)malloc((size_t)n * sizeof(item_t));
    if (!it) return -1;
    for (int64_t i = 0; i < n; i++) {
        it[i].start = start[i];
        it[i].end = end[i];
        it[i].id = ids ? ids[i] : (int32_t)i;
        it[i].pos = i;
    }
    qsort(it, (size_t)n, sizeof(item_t), by_start);
    int64_t nc = 0;
    int64_t reach = 0; /* largest end of the open cluster */
    for (int64_t i = 0; i < n; i++) {
        if (i == 0 || (int64_t)it[i].start - (int64_t)max_dist > reach) { /* cluster.c:229 "to the right of this cluster" */
            c_start[nc] = it[i].start;
            c_off[nc] = i;
            reach = it[i].end;
            nc++;
        } else if (it[i].end > reach) {
            reach = it[i].end; /* cluster.c:242 node->end = max(end, node->end) */
        }
        c_end[nc - 1] = (int32_t)reach;
        members[i] = it[i].id;
    }
    c_off[nc] = n;
    for (int64_t c = 0; c < nc; c++) /* cluster.pyx:95 sorted(ids) */
        qsort(members + c_off[c], (size_t)(c_off[c + 1] - c_off[c]), sizeof(int32_t), by_id);
    free(it);
    return nc;
}
