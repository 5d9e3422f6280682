/*
 * suchtree_oracle.c -- CPU restatement of the reference's bulk patristic
 * distance / MRCA path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (suchtree_amd/ -> libsuchtree_hip.so) never links, imports or
 * calls it and fails loudly when the HIP library is missing.
 *
 * What it restates (all citations relative to /root/reference):
 *   Node layout ............ SuchTree/MuchTree.pyx:55-60   (20-byte AoS)
 *   depth .................. SuchTree/MuchTree.pyx:218-225 (nodes on the
 *                            longest leaf->root path, sizes `visited`)
 *   _mrca .................. SuchTree/MuchTree.pyx:999-1030 (visited list)
 *   _distances ............. SuchTree/MuchTree.pyx:911-943 (fp32 ordered
 *                            sum a->mrca then b->mrca, stored as double)
 *   _distance (scalar) ..... SuchTree/MuchTree.pyx:981-997
 *   common_ancestor loop ... SuchTree/MuchTree.pyx:1128-1149 (the reference
 *                            has no bulk MRCA; the oracle for bulk MRCA ids
 *                            is a loop over the scalar call)
 *   linked_distances pairs . SuchTree/MuchTree.pyx:2918-2925
 *   _quartet_topologies .... SuchTree/MuchTree.pyx:1331-1376
 *
 * The C that Cython generated from those lines ships in the snapshot and settles the
 * arithmetic: `float __pyx_v_d;` (SuchTree/MuchTree.c:32341), two plain
 * `__pyx_v_d = (__pyx_v_d + data[n].distance)` statements in path order (:32435 a-side,
 * :32475 b-side) and a widening store into the double result (:32496).
 *
 * Pinning: the reference's Python module cannot be imported in the build
 * container (hard `import dendropy` at MuchTree.pyx:3; dendropy is not
 * installed and no stand-in is written for it) -- but the C that Cython
 * generated from _distances / _mrca ships in the reference's repository and
 * contains no Python API call: oracle/ref_harness.c compiles it where it lies
 * into oracle/_ref/libref_hotpath.so, and tests/test_oracle_ref.py checks this
 * restatement against the reference's own compiled code BIT FOR BIT on
 * arbitrary inputs (round 6).  The callers around the path and the ingest stay
 * pinned by the reference's known answers: SuchTree/tests/test.matrix (225
 * distances), the values printed in docs/examples/SuchTree_examples.md and the
 * dendropy-printed leaf ids there.  See tests/test_oracle_golden.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; x86-64 SSE float
 * adds, i.e. the same arithmetic gcc emits for the Cython-generated C).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* MuchTree.pyx:55-60 */
typedef struct {
    int   parent;
    int   left_child;
    int   right_child;
    float support;
    float distance;
} oracle_node;

/* Fill the 20-byte AoS from flat arrays (test convenience). */
void oracle_fill_nodes(oracle_node *data, int64_t n,
                       const int32_t *parent, const int32_t *left,
                       const int32_t *right, const float *support,
                       const float *distance)
{
    for (int64_t i = 0; i < n; i++) {
        data[i].parent      = parent[i];
        data[i].left_child  = left ? left[i] : -1;
        data[i].right_child = right ? right[i] : -1;
        data[i].support     = support ? support[i] : -1.0f;
        data[i].distance    = distance[i];
    }
}

/* MuchTree.pyx:218-225 -- for every leaf count the nodes up to the root. */
unsigned int oracle_depth(const oracle_node *data, int64_t n)
{
    unsigned int depth = 0;
    for (int64_t leaf = 0; leaf < n; leaf++) {
        if (data[leaf].left_child != -1) continue;   /* leaves only */
        unsigned int c = 1;
        int id = (int)leaf;
        for (;;) {
            if (data[id].parent == -1) break;
            id = data[id].parent;
            c++;
        }
        if (c > depth) depth = c;
    }
    return depth;
}

/* MuchTree.pyx:999-1030 -- `visited` holds a and all its ancestors; b climbs
 * and scans the list linearly at every step. */
int oracle_mrca(const oracle_node *data, int64_t *visited, int a, int b)
{
    int n, i, mrca = -1, a_depth;

    n = a;
    i = 0;
    for (;;) {
        visited[i] = n;
        n = data[n].parent;
        i++;
        if (n == -1) break;
    }
    a_depth = i;

    n = b;
    for (;;) {
        i = 0;
        for (;;) {
            if (i >= a_depth) break;
            if (visited[i] == n) {
                mrca = (int)visited[i];
                break;
            }
            i++;
        }
        if (mrca != -1) break;
        n = data[n].parent;
        if (n == -1) {
            mrca = n;
            break;
        }
    }
    return mrca;
}

/* MuchTree.pyx:911-943.  ids is an (length,2) int64 view with byte-free
 * element strides s0,s1 (the reference takes any memoryview strides). */
void oracle_distances(const oracle_node *data, unsigned int length,
                      int64_t *visited, const int64_t *ids,
                      int64_t s0, int64_t s1, double *result)
{
    unsigned int mrca, n, a, b, i;
    float d;

    for (i = 0; i < length; i++) {
        a = (unsigned int)ids[(int64_t)i * s0];
        b = (unsigned int)ids[(int64_t)i * s0 + s1];
        mrca = (unsigned int)oracle_mrca(data, visited, (int)a, (int)b);
        n = a;
        d = 0;
        while (n != mrca) {
            d += data[n].distance;
            n = (unsigned int)data[n].parent;
        }
        n = b;
        while (n != mrca) {
            d += data[n].distance;
            n = (unsigned int)data[n].parent;
        }
        result[i] = d;
    }
}

/* 64-bit length form used by the tests and the timed baseline. */
void oracle_distances_n(const oracle_node *data, int64_t length,
                        int64_t *visited, const int64_t *ids,
                        int64_t s0, int64_t s1, double *result)
{
    const int64_t step = 1 << 30;
    for (int64_t off = 0; off < length; off += step) {
        int64_t m = length - off < step ? length - off : step;
        oracle_distances(data, (unsigned int)m, visited, ids + off * s0, s0, s1,
                         result + off);
    }
}

/* MuchTree.pyx:981-997 (scalar form; MRCA via the same walk). */
float oracle_distance(const oracle_node *data, int64_t *visited, int a, int b)
{
    int mrca = oracle_mrca(data, visited, a, b);
    float d = 0;
    int n = a;
    while (n != mrca) {
        d += data[n].distance;
        n = data[n].parent;
    }
    n = b;
    while (n != mrca) {
        d += data[n].distance;
        n = data[n].parent;
    }
    return d;
}

/* Loop over common_ancestor (MuchTree.pyx:1128-1149): the bulk-MRCA oracle. */
void oracle_mrca_bulk(const oracle_node *data, int64_t length, int64_t *visited,
                      const int64_t *ids, int64_t s0, int64_t s1, int32_t *out)
{
    for (int64_t i = 0; i < length; i++)
        out[i] = oracle_mrca(data, visited, (int)ids[i * s0], (int)ids[i * s0 + s1]);
}

/* MuchTree.pyx:2918-2925 -- link-pair enumeration of linked_distances():
 * k = i(i-1)/2 + j ; ids_a[k] = (ll[j,1], ll[i,1]) ; ids_b[k] = (ll[j,0], ll[i,0]) */
void oracle_linked_pairs(const int64_t *linklist, int64_t n_links,
                         int64_t *ids_a, int64_t *ids_b)
{
    int64_t k = 0;
    for (int64_t i = 0; i < n_links; i++)
        for (int64_t j = 0; j < i; j++) {
            ids_a[2 * k + 1] = linklist[2 * i + 1];
            ids_a[2 * k + 0] = linklist[2 * j + 1];
            ids_b[2 * k + 1] = linklist[2 * i + 0];
            ids_b[2 * k + 0] = linklist[2 * j + 0];
            k++;
        }
}

/* MuchTree.pyx:1331-1376 -- quartet topologies with the table I of :1319-1320. */
void oracle_quartets(const oracle_node *data, int64_t n, int64_t *visited,
                     const int64_t *quartets, int64_t *topologies)
{
    static const int I[6][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {0, 3, 1, 2},
                                {1, 2, 0, 3}, {1, 3, 0, 2}, {2, 3, 0, 1}};
    int64_t M[6], C[6];
    for (int64_t i = 0; i < n; i++) {
        int a = (int)quartets[4 * i], b = (int)quartets[4 * i + 1];
        int c = (int)quartets[4 * i + 2], d = (int)quartets[4 * i + 3];
        int j, k;
        M[0] = oracle_mrca(data, visited, a, b);
        M[1] = oracle_mrca(data, visited, a, c);
        M[2] = oracle_mrca(data, visited, a, d);
        M[3] = oracle_mrca(data, visited, b, c);
        M[4] = oracle_mrca(data, visited, b, d);
        M[5] = oracle_mrca(data, visited, c, d);
        for (j = 0; j < 6; j++) C[j] = 0;
        for (j = 0; j < 6; j++)
            for (k = 0; k < 6; k++)
                if (M[j] == M[k]) C[j] = C[j] + 1;
        for (j = 0; j < 6; j++)
            if (C[j] == 1) break;
        if (j == 6) j = 5;   /* Cython's range loop leaves the last value in j */
        for (k = 0; k < 4; k++) topologies[4 * i + k] = quartets[4 * i + I[j][k]];
    }
}

/* ---- "host cores" baseline ------------------------------------------------
 * The reference holds the GIL in _distances (gen-C MuchTree.c:32277); its
 * documented way to use more cores is a fork pool over contiguous chunks
 * (docs/examples/SuchTree_examples.md:462-497).  This driver is that: the
 * same serial kernel on contiguous chunks, one `visited` scratch per worker. */
typedef struct {
    const oracle_node *data;
    int64_t begin, end, depth;
    const int64_t *ids;
    int64_t s0, s1;
    double *result;
} oracle_job;

static void *oracle_worker(void *p)
{
    oracle_job *j = (oracle_job *)p;
    int64_t *visited = (int64_t *)calloc((size_t)j->depth + 1, sizeof(int64_t));
    if (!visited) return NULL;
    oracle_distances_n(j->data, j->end - j->begin, visited,
                       j->ids + j->begin * j->s0, j->s0, j->s1,
                       j->result + j->begin);
    free(visited);
    return NULL;
}

int oracle_distances_mt(const oracle_node *data, int64_t length, int64_t depth,
                        const int64_t *ids, int64_t s0, int64_t s1,
                        double *result, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    oracle_job *job = (oracle_job *)malloc(sizeof(oracle_job) * (size_t)n_threads);
    if (!tid || !job) { free(tid); free(job); return -1; }
    for (int t = 0; t < n_threads; t++) {
        job[t].data = data;
        job[t].begin = length * t / n_threads;
        job[t].end = length * (t + 1) / n_threads;
        job[t].depth = depth;
        job[t].ids = ids; job[t].s0 = s0; job[t].s1 = s1;
        job[t].result = result;
        if (pthread_create(&tid[t], NULL, oracle_worker, &job[t]) != 0) {
            for (int u = 0; u < t; u++) pthread_join(tid[u], NULL);
            free(tid); free(job);
            return -2;
        }
    }
    for (int t = 0; t < n_threads; t++) pthread_join(tid[t], NULL);
    free(tid); free(job);
    return 0;
}
