/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY: the reference's OWN compiled hot path behind the oracle.
 *
 * The reference ships the C that Cython generated from its hot path (SuchTree/MuchTree.c, in the repository).  The two
 * functions of the path -- SuchTree._distances (MuchTree.pyx:911-943 = MuchTree.c:32339-32500) and SuchTree._mrca
 * (MuchTree.pyx:999-1030 = MuchTree.c:33338-33618) -- and their caller _quartet_topologies (MuchTree.pyx:1331-1376) are
 * `cdef ... noexcept nogil` methods: plain C loops over
 * `self->data` and three memoryview-slice structs, without a single Python API call inside.  This file #includes that
 * generated C WHERE IT LIES under /root/reference (oracle/Makefile passes -I/root/reference/SuchTree; nothing of the
 * reference is copied into this repository) and exports three entry points that fill the extension type's C struct and the
 * slice structs from flat arrays and call the reference's functions.  The Python module itself is never initialised or
 * imported -- its init would `import dendropy` (MuchTree.pyx:3), which this image lacks and for which no stand-in is
 * written; nothing here stands in for anything: the headers the generated C needs (Python.h 3.10, numpy's) are in the image,
 * and the code that runs is the reference's, compiled by gcc with the flags distutils would use (-O2, no -ffast-math).
 *
 * Output: oracle/_ref/libref_hotpath.so (git-ignored; travels to the GPU box like every built .so).  Users: tests/ (the
 * oracle's restatement against it, bit for bit, on arbitrary inputs) and bench.py's cpu_baseline leg ("kind": "reference").
 * The product never loads it.  Load it inside a Python process (ctypes): the generated C refers to Python API symbols that
 * the interpreter exports; none of them is called.
 */
#include <pthread.h>

#include "MuchTree.c"

typedef struct __pyx_obj_8SuchTree_8MuchTree_SuchTree ref_tree_t;
typedef struct __pyx_t_8SuchTree_8MuchTree_Node ref_node_t;

static struct __pyx_vtabstruct_8SuchTree_8MuchTree_SuchTree ref_vtab;

/* the extension type's C struct as SuchTree.__init__ leaves it for the path: data (parent / distance per node; children and
 * support are not read by the path), depth (sizes `visited`, MuchTree.pyx:218-225, 906), the vtable with the two methods */
static int ref_fill(ref_tree_t *self, const int32_t *parent, const float *distance, int64_t n_nodes, int32_t depth)
{
    memset(self, 0, sizeof *self);
    memset(&ref_vtab, 0, sizeof ref_vtab);
    ref_vtab._distances = __pyx_f_8SuchTree_8MuchTree_8SuchTree__distances;
    ref_vtab._mrca = __pyx_f_8SuchTree_8MuchTree_8SuchTree__mrca;
    self->__pyx_vtab = &ref_vtab;
    self->data = (ref_node_t *)malloc((size_t)n_nodes * sizeof(ref_node_t));
    if (!self->data) return 1;
    for (int64_t i = 0; i < n_nodes; i++) {
        self->data[i].parent = parent[i];
        self->data[i].left_child = -1;
        self->data[i].right_child = -1;
        self->data[i].support = -1.0f;
        self->data[i].distance = distance[i];
    }
    self->length = (unsigned int)n_nodes;
    self->depth = (unsigned int)depth;
    return 0;
}

static __Pyx_memviewslice ref_slice1(void *data, Py_ssize_t n, Py_ssize_t stride_bytes)
{
    __Pyx_memviewslice s;
    memset(&s, 0, sizeof s);
    s.data = (char *)data;
    s.shape[0] = n;
    s.strides[0] = stride_bytes;
    s.suboffsets[0] = -1;
    return s;
}

static __Pyx_memviewslice ref_slice_pairs(const int64_t *pairs, Py_ssize_t n, Py_ssize_t stride0_bytes, Py_ssize_t stride1_bytes)
{
    __Pyx_memviewslice s = ref_slice1((void *)pairs, n, stride0_bytes);
    s.shape[1] = 2;
    s.strides[1] = stride1_bytes;
    s.suboffsets[1] = -1;
    return s;
}

typedef struct {
    ref_tree_t *self;
    const int64_t *pairs;
    int64_t n, stride0, stride1;
    double *out;
    int depth;
} ref_job_t;

static void *ref_worker(void *arg)
{
    ref_job_t *j = (ref_job_t *)arg;
    long *visited = (long *)calloc((size_t)j->depth + 1, sizeof(long));      /* np.zeros(depth, dtype=int), MuchTree.pyx:906 */
    if (!visited) return (void *)1;
    __pyx_f_8SuchTree_8MuchTree_8SuchTree__distances(j->self, (unsigned int)j->n, ref_slice1(visited, j->depth, sizeof(long)),
                                                    ref_slice_pairs(j->pairs, j->n, j->stride0, j->stride1), ref_slice1(j->out, j->n, sizeof(double)));
    free(visited);
    return NULL;
}

/* result[i] = SuchTree._distances on pair i, as distances_bulk calls it (MuchTree.pyx:906-908); pairs: int64, byte strides.
 * n_threads > 1: contiguous chunks, one thread each (the reference under a fork pool; every thread its own `visited`). */
int ref_hotpath_distances(const int32_t *parent, const float *distance, int64_t n_nodes, int32_t depth, const int64_t *pairs,
                          int64_t n, int64_t stride0_bytes, int64_t stride1_bytes, double *out, int n_threads)
{
    ref_tree_t self;
    if (ref_fill(&self, parent, distance, n_nodes, depth)) return 1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    int rc = 0;
    if (n_threads == 1 || n < 2 * (int64_t)n_threads) {
        ref_job_t j = {&self, pairs, n, stride0_bytes, stride1_bytes, out, depth};
        rc = ref_worker(&j) != NULL;
    } else {
        pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
        ref_job_t *jobs = (ref_job_t *)malloc((size_t)n_threads * sizeof(ref_job_t));
        int started = 0;
        for (int t = 0; th && jobs && t < n_threads; t++) {
            const int64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
            jobs[t].self = &self;
            jobs[t].pairs = (const int64_t *)((const char *)pairs + lo * stride0_bytes);
            jobs[t].n = hi - lo;
            jobs[t].stride0 = stride0_bytes;
            jobs[t].stride1 = stride1_bytes;
            jobs[t].out = out + lo;
            jobs[t].depth = depth;
            if (pthread_create(&th[t], NULL, ref_worker, &jobs[t]) != 0) {      /* (thread limit: this chunk on the calling thread) */
                rc |= ref_worker(&jobs[t]) != NULL;
                th[t] = 0;
                continue;
            }
            started |= 1;
        }
        for (int t = 0; th && jobs && t < n_threads; t++)
            if (th[t]) {
                void *r = NULL;
                pthread_join(th[t], &r);
                rc |= r != NULL;
            }
        (void)started;
        if (!th || !jobs) rc = 1;
        free(th);
        free(jobs);
    }
    free(self.data);
    return rc;
}

/* topologies[i, 0..3] = SuchTree._quartet_topologies on quartet i (MuchTree.pyx:1331-1376 = MuchTree.c:39291-39590: six _mrca calls and
 * the pick of the unique one -- C loops, no Python API call), with the scratch arrays quartet_topologies_bulk hands it
 * (MuchTree.pyx:1318-1326: visited, M, C of six zeros, the 6 x 4 matrix I -- `perm` here: complex.h owns the name).  quartets / out: C-order int64 (n, 4). */
int ref_hotpath_quartets(const int32_t *parent, const float *distance, int64_t n_nodes, int32_t depth, const int64_t *quartets, int64_t n,
                         int64_t *out)
{
    ref_tree_t self;
    if (ref_fill(&self, parent, distance, n_nodes, depth)) return 1;
    ref_vtab._quartet_topologies = __pyx_f_8SuchTree_8MuchTree_8SuchTree__quartet_topologies;
    long *visited = (long *)calloc((size_t)depth + 1, sizeof(long));
    if (!visited) { free(self.data); return 1; }
    long M[6] = {0}, C[6] = {0};
    long perm[6][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {0, 3, 1, 2}, {1, 2, 0, 3}, {1, 3, 0, 2}, {2, 3, 0, 1}};
    __Pyx_memviewslice q = ref_slice1((void *)quartets, n, 4 * sizeof(long)), t = ref_slice1(out, n, 4 * sizeof(long)),
                       i = ref_slice1(perm, 6, 4 * sizeof(long));
    q.shape[1] = t.shape[1] = i.shape[1] = 4;
    q.strides[1] = t.strides[1] = i.strides[1] = sizeof(long);
    q.suboffsets[1] = t.suboffsets[1] = i.suboffsets[1] = -1;
    __pyx_f_8SuchTree_8MuchTree_8SuchTree__quartet_topologies(&self, q, t, ref_slice1(visited, depth, sizeof(long)), ref_slice1(M, 6, sizeof(long)),
                                                              ref_slice1(C, 6, sizeof(long)), i);
    free(visited);
    free(self.data);
    return 0;
}

/* out[i] = SuchTree._mrca(visited, a_i, b_i) -- what common_ancestor runs per call (MuchTree.pyx:1128-1149) */
int ref_hotpath_mrca(const int32_t *parent, const float *distance, int64_t n_nodes, int32_t depth, const int64_t *pairs, int64_t n,
                     int64_t stride0_bytes, int64_t stride1_bytes, int32_t *out)
{
    ref_tree_t self;
    if (ref_fill(&self, parent, distance, n_nodes, depth)) return 1;
    long *visited = (long *)calloc((size_t)depth + 1, sizeof(long));
    if (!visited) { free(self.data); return 1; }
    const __Pyx_memviewslice v = ref_slice1(visited, depth, sizeof(long));
    for (int64_t i = 0; i < n; i++) {
        const int64_t a = *(const int64_t *)((const char *)pairs + i * stride0_bytes);
        const int64_t b = *(const int64_t *)((const char *)pairs + i * stride0_bytes + stride1_bytes);
        out[i] = __pyx_f_8SuchTree_8MuchTree_8SuchTree__mrca(&self, v, (int)a, (int)b);
    }
    free(visited);
    free(self.data);
    return 0;
}
