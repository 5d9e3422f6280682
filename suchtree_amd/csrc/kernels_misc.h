// kernels_misc.h -- included by suchtree_hip.hip (after device_common.h).
// k nearest candidates per row, dense graph matrices, (see also the copy kernels next to the host pipe).
#pragma once

namespace st {

// ---- k nearest candidates per query row (nearest_neighbors, MuchTree.pyx:1032-1082) ----------
// dist[row * n_c + c] are the float32 distances query(row) -> cands[c].  One workgroup per row
// selects the k smallest in k rounds: round r finds the smallest key greater than the one
// chosen in round r-1, key = (order-preserving image of the float) << 32 | candidate index, so
// ties resolve to the lower candidate index (the reference's argsort leaves tie order
// unspecified).  Special values order as numpy's argsort orders them: -0.0 ties with +0.0, every
// NaN (either sign) comes last.  O(k * n_c) per row, for small k; the facade sorts on the host
// beyond kKnnMaxK.
constexpr int kKnnMaxK = 256;

__device__ __forceinline__ unsigned long long knn_key(float d, unsigned idx)
{
    unsigned u = __float_as_uint(d);
    if (d != d) u = 0xFFFFFFFEu;                  // NaN: after every number (~0ull stays "nothing found")
    else {
        if (u == 0x80000000u) u = 0;               // -0.0 == +0.0
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    return ((unsigned long long)u << 32) | idx;
}

__global__ __launch_bounds__(256) void k_knn_select(const float *__restrict__ dist, long long n_c,
                                                    const long long *__restrict__ queries,
                                                    const long long *__restrict__ cands, int skip_self, int k,
                                                    long long *__restrict__ out_index, double *__restrict__ out_dist)
{
    __shared__ unsigned long long wave_min[4];
    __shared__ unsigned long long chosen;
    const long long row = blockIdx.x;
    const float *d = dist + row * n_c;
    const long long q = queries[row];
    unsigned long long prev = 0;
    bool first = true;
    for (int r = 0; r < k; r++) {
        unsigned long long best = ~0ull;
        for (long long c = threadIdx.x; c < n_c; c += blockDim.x) {
            if (skip_self && cands[c] == q) continue;
            const unsigned long long key = knn_key(d[c], (unsigned)c);
            if ((first || key > prev) && key < best) best = key;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long other = __shfl_xor(best, off);
            best = other < best ? other : best;
        }
        if ((threadIdx.x & 63) == 0) wave_min[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long m = wave_min[0];
            for (int w = 1; w < 4; w++) m = wave_min[w] < m ? wave_min[w] : m;
            chosen = m;
            const long long c = (long long)(m & 0xFFFFFFFFull);
            const bool found = m != ~0ull;
            out_index[row * k + r] = found ? c : -1;
            out_dist[row * k + r] = found ? (double)d[c] : __builtin_nan("");
        }
        __syncthreads();
        prev = chosen;
        first = false;
        if (prev == ~0ull) {     // fewer than k candidates: the remaining slots stay -1 / NaN
            for (int rr = r + 1 + (int)threadIdx.x; rr < k; rr += blockDim.x) {
                out_index[row * k + rr] = -1;
                out_dist[row * k + rr] = __builtin_nan("");
            }
            break;
        }
        __syncthreads();
    }
}

// ---- dense graph matrices of SuchLinkedTrees (adjacency / Laplacian, MuchTree.pyx:3081-3145)
// A[u][v] = A[v][u] = w for every edge; L = diag(column sums of A) - A.  The column sums run
// over the rows in increasing order, like numpy's sum(axis=0), so L is bit-identical to the
// host formula.
__global__ void k_graph_scatter(double *__restrict__ A, long long n, long long n_edges,
                                const int *__restrict__ u, const int *__restrict__ v,
                                const double *__restrict__ w)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (long long)gridDim.x * blockDim.x) {
        A[(long long)u[e] * n + v[e]] = w[e];
        A[(long long)v[e] * n + u[e]] = w[e];
    }
}

__global__ void k_graph_degree(const double *__restrict__ A, long long n, double *__restrict__ deg)
{
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (long long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (long long i = 0; i < n; i++) s += A[i * n + j];   // lanes read consecutive columns: coalesced
        deg[j] = s;
    }
}

__global__ void k_graph_laplacian(const double *__restrict__ A, const double *__restrict__ deg, long long n,
                                  double *__restrict__ L)
{
    const long long total = n * n;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total;
         k += (long long)gridDim.x * blockDim.x) {
        const long long i = k / n, j = k - i * n;
        L[k] = (i == j ? deg[j] : 0.0) - A[k];
    }
}

// Uniform random leaf pairs for the creation-time timing of a deep tree's candidate kernels (host_tune.h): pair i =
// (leaves[h(2i)], leaves[h(2i + 1)]), h a 32-bit mixer scaled to the leaf count by a multiply-high.
// 24-bit MRCA ids (device_common.h::MrcaSink) -> int32: the receiving side of a result slice that travelled packed.
// Three byte loads per lane (the packed stream may start at any byte), coalesced 4-byte stores.
__global__ __launch_bounds__(256) void k_unpack24(const unsigned char *__restrict__ src, long long n, int *__restrict__ dst)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned char *p = src + 3 * i;
        const unsigned v = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16);
        dst[i] = v == 0xFFFFFFu ? -1 : (int)v;
    }
}

__global__ __launch_bounds__(256) void k_sample_leaf_pairs(const int *__restrict__ leaves, unsigned n_leaves,
                                                           long long *__restrict__ pairs, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long v[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            unsigned x = (unsigned)(2 * i + k) * 0x9E3779B9u + 0x7F4A7C15u;
            x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
            v[k] = leaves[__umulhi(x, n_leaves)];
        }
        reinterpret_cast<longlong2 *>(pairs)[i] = make_longlong2(v[0], v[1]);
    }
}

}  // namespace st
