// kernels_walk.h -- part of libsuchtree_hip.so's single translation unit (included by suchtree_hip.hip,
// in this order: device_common.h, kernels_walk.h, kernels_canopy.h, kernels_misc.h).
// Walk family: k_walk, the mailbox kernel, the quartet kernels.
#pragma once

namespace st {

// --------------------------------------------------------------------------
// walk kernel
// --------------------------------------------------------------------------
struct WalkParams {
    const Node8 *nodes;
    const int32_t *depth;
    const Stride3 *stride;
    const uint64_t *rmq;     // whole-tree sparse table for the meeting node, or NULL
    long long n_nodes;
    LineageView lineage;     // a's side in one read (deep trees with lineage sums), else empty
};

template <typename Src>
__global__ __launch_bounds__(256) void k_walk(WalkParams P, Src src, long long n,
                                              DistSink out_d, int *__restrict__ out_m,
                                              Fault *fault)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long a, b;
        src.load(i, a, b);
        if ((unsigned long long)a >= (unsigned long long)P.n_nodes ||
            (unsigned long long)b >= (unsigned long long)P.n_nodes) {
            record_fault(fault, a, b, P.n_nodes);
            store_result(out_d, out_m, i, __builtin_nanf(""), -1);
            continue;
        }
        if (out_d.any()) {
            const PairResult r = pair_walk(P.nodes, P.depth, P.stride, (int32_t)a, (int32_t)b, P.rmq, P.n_nodes, P.lineage);
            store_result(out_d, out_m, i, r.dist, r.mrca);
        } else {
            out_m[i] = pair_walk_mrca(P.nodes, P.depth, P.stride, (int32_t)a, (int32_t)b, nullptr, P.rmq, P.n_nodes);
        }
    }
}

// The mailbox form of k_walk (small host batches, one lane per pair, no grid stride): pairs and
// results live in pinned host memory, and so does a completion word -- the last workgroup to
// finish publishes the call's sequence number there (system-scope release after every block's
// system-scope fence), so the host learns of completion by polling its own memory instead of
// paying a stream synchronisation (the driver's wake-up costs as much as the whole kernel).
__global__ __launch_bounds__(64) void k_walk_mailbox(WalkParams P, const long long *__restrict__ pairs, int n,
                                                     double *__restrict__ out_d, int *__restrict__ out_m,
                                                     unsigned *block_counter, unsigned *done_word, unsigned seq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const longlong2 v = reinterpret_cast<const longlong2 *>(pairs)[i];   // (ids were range-checked on the host)
        if (out_d) {
            const PairResult r = pair_walk(P.nodes, P.depth, P.stride, (int32_t)v.x, (int32_t)v.y, P.rmq, P.n_nodes, P.lineage);
            out_d[i] = (double)r.dist;
            if (out_m) out_m[i] = r.mrca;
        } else {
            out_m[i] = pair_walk_mrca(P.nodes, P.depth, P.stride, (int32_t)v.x, (int32_t)v.y, nullptr, P.rmq, P.n_nodes);
        }
    }
    __threadfence_system();            // this lane's results are visible to the host ...
    __syncthreads();                   // ... and so are the whole block's
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(block_counter, 1u);
        if (done == gridDim.x - 1) {   // last block of the launch
            *block_counter = 0;        // (launches on the mailbox stream are serial)
            __threadfence_system();
            __hip_atomic_store(done_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Quartet topologies (MuchTree.pyx:1331-1376): six MRCAs per quartet (ab ac ad bc bd cd),
// the first MRCA id that occurs exactly once names the sister pair; the row is re-ordered
// by the matching line of the table I = {0123, 0213, 0312, 1203, 1302, 2301}.  If no id is
// unique the reference's loop leaves j = 5, reproduced here.  Integer work only.
__global__ __launch_bounds__(256) void k_quartets(WalkParams P, const long long *__restrict__ q,
                                                  long long n, long long s0, long long s1,
                                                  long long *__restrict__ out, Fault *fault)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long id[4];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            id[k] = q[i * s0 + k * s1];
            ok &= (unsigned long long)id[k] < (unsigned long long)P.n_nodes;
        }
        if (!ok) {
            record_fault(fault, id[0], id[1], P.n_nodes);
            record_fault(fault, id[2], id[3], P.n_nodes);
#pragma unroll
            for (int k = 0; k < 4; k++) out[i * 4 + k] = -1;
            continue;
        }
        const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
        int M[6];
#pragma unroll
        for (int j = 0; j < 6; j++)
            M[j] = pair_walk_mrca(P.nodes, P.depth, P.stride, (int32_t)id[pa[j]], (int32_t)id[pb[j]], nullptr, P.rmq, P.n_nodes);
        int pick = 5;
#pragma unroll
        for (int j = 5; j >= 0; j--) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) c += M[j] == M[k];
            if (c == 1) pick = j;
        }
        // I[pick] = {pa, pb, the other two in increasing order}
        const int a = pa[pick], b = pb[pick];
        int rest[2], r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k != a && k != b) { if (r < 2) rest[r] = k; r++; }
        out[i * 4 + 0] = id[a];
        out[i * 4 + 1] = id[b];
        out[i * 4 + 2] = id[rest[0]];
        out[i * 4 + 3] = id[rest[1]];
    }
}

// Second half of the quartet path when the MRCA ids came from a canopy launch over
// SrcQuartet: M[6*i .. 6*i+5] are the ids of quartet i (-1 where an id was out of range).
__global__ __launch_bounds__(256) void k_quartet_pick(const long long *__restrict__ q, const int *__restrict__ M,
                                                      long long n, long long *__restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int m[6];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 6; j++) { m[j] = M[i * 6 + j]; bad |= m[j] < 0; }
        long long id[4];
#pragma unroll
        for (int k = 0; k < 4; k++) id[k] = q[i * 4 + k];
        if (bad) {
#pragma unroll
            for (int k = 0; k < 4; k++) out[i * 4 + k] = -1;
            continue;
        }
        int pick = 5;
#pragma unroll
        for (int j = 5; j >= 0; j--) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) c += m[j] == m[k];
            if (c == 1) pick = j;
        }
        const int a = (0x940 >> (2 * pick)) & 3, b = (0xFB9 >> (2 * pick)) & 3;
        int rest[2], r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k != a && k != b) { if (r < 2) rest[r] = k; r++; }
        out[i * 4 + 0] = id[a];
        out[i * 4 + 1] = id[b];
        out[i * 4 + 2] = id[rest[0]];
        out[i * 4 + 3] = id[rest[1]];
    }
}

}  // namespace st
