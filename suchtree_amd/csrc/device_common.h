// device_common.h -- included first by every translation unit of libsuchtree_hip.so.
// Fault word, pair sources (where pair i of a launch comes from), result sinks.
#pragma once

namespace st {

// --------------------------------------------------------------------------
// device-side helpers
// --------------------------------------------------------------------------
struct Fault {
    long long max_bad;   // largest offending id seen  (init INT64_MIN)
    long long min_bad;   // smallest offending id seen (init INT64_MAX)
};

__device__ __forceinline__ void record_fault(Fault *f, long long a, long long b, long long n_nodes)
{
    if (a < 0 || a >= n_nodes) { atomicMax(&f->max_bad, a); atomicMin(&f->min_bad, a); }
    if (b < 0 || b >= n_nodes) { atomicMax(&f->max_bad, b); atomicMin(&f->min_bad, b); }
}

// ---- pair sources: where pair number i of a launch comes from ----------------
// C-order int64 (n,2): one 16-byte load per lane, fully coalesced.
struct SrcContig {
    const long long *pairs;
    __device__ __forceinline__ void load(long long i, long long &a, long long &b) const
    {
        const longlong2 v = reinterpret_cast<const longlong2 *>(pairs)[i];
        a = v.x;
        b = v.y;
    }
};

// C-order int32 (n,2): what the host path ships over PCIe (node ids always fit in 31 bits;
// the packing step clamps anything wider so that it still fails the range check).
// packed48: the narrower wire format of trees with fewer than 2^24 nodes -- 24 bits per id, 6 bytes per pair, three
// 2-byte loads per lane (consecutive lanes read consecutive bytes; 0xFFFFFF stands for an id out of range, which the
// host has already judged: host_copy.h::pack_pairs48).  One type for both formats (a wave-uniform branch), so that
// no kernel is compiled twice for it.
struct SrcContig32 {
    const int *pairs;
    int packed48;
    __device__ __forceinline__ void load(long long i, long long &a, long long &b) const
    {
        if (packed48) {
            const unsigned short *q = reinterpret_cast<const unsigned short *>(pairs) + 3 * i;
            const unsigned w0 = q[0], w1 = q[1], w2 = q[2];
            a = (long long)(w0 | ((w1 & 0xFFu) << 16));
            b = (long long)((w1 >> 8) | (w2 << 8));
        } else {
            const int2 v = reinterpret_cast<const int2 *>(pairs)[i];
            a = v.x;
            b = v.y;
        }
    }
};

// Any other (n,2) view: element strides s0 (rows) and s1 (columns).
struct SrcStrided {
    const long long *pairs;
    long long s0, s1;
    __device__ __forceinline__ void load(long long i, long long &a, long long &b) const
    {
        a = pairs[i * s0];
        b = pairs[i * s0 + s1];
    }
};

// All-pairs generator: pair k = (ids[j], ids[i]) with k = i(i-1)/2 + j, 0 <= j < i,
// the enumeration of SuchLinkedTrees.linked_distances (MuchTree.pyx:2918-2925) and, up to
// order, of pairwise_distances (:1111-1114).  Nothing is read but the id list: within a
// wave b = ids[i] is (nearly) uniform and a = ids[j] walks the list, so the record reads
// of consecutive lanes fall on consecutive sectors.
struct SrcTriangle {
    const long long *ids;
    long long stride;   // element stride of ids
    long long k0;       // first pair index of this launch
    __device__ __forceinline__ void load(long long i, long long &a, long long &b) const
    {
        const long long k = k0 + i;
        long long row = (long long)((1.0 + sqrt(1.0 + 8.0 * (double)k)) * 0.5);
        if (row * (row - 1) / 2 > k) row--;
        if ((row + 1) * row / 2 <= k) row++;
        const long long col = k - row * (row - 1) / 2;
        a = ids[col * stride];
        b = ids[row * stride];
    }
};

// Grid generator: element e = e0 + i of an n_rows x n_cols grid (C order) is the pair
// (rows[r], cols[c]), r = e / n_cols, c = e % n_cols.  `symmetric` (rows and cols are the same
// id list): below the diagonal the pair is taken in the order of its mirror image above it,
// (ids[c], ids[r]) for c < r, so the square is exactly the mirrored upper triangle that
// pairwise_distances builds (MuchTree.pyx:1106-1124: d(ids[i], ids[j]) for i < j, stored at
// [i,j] and [j,i]); the diagonal comes out as d(x,x) = 0.  Consecutive lanes share one endpoint
// and walk the id list with the other, so record reads coalesce as in the triangle.
struct SrcGrid {
    const long long *rows, *cols;
    long long n_cols, e0;
    int symmetric;
    __device__ __forceinline__ void load(long long i, long long &a, long long &b) const
    {
        const long long e = e0 + i;
        const long long r = e / n_cols, c = e - r * n_cols;
        a = rows[r];
        b = cols[c];
        if (symmetric && c < r) { const long long t = a; a = b; b = t; }
    }
};

// Where distances go.  The C ABI's contract is float64 (what the reference returns); the
// values are float32 sums, so the host path ships them over PCIe as float32 and widens them
// on the host (half the D2H bytes, bit-identical result).
struct DistSink {
    double *d64;
    float *f32;
    __host__ __device__ bool any() const { return d64 != nullptr || f32 != nullptr; }
};

// The six pairs of a quartet (a,b,c,d), in the reference's order ab ac ad bc bd cd
// (MuchTree.pyx:1353-1358): pair i is combination i % 6 of quartet i / 6.  Lets the canopy
// kernels produce the six MRCA ids of every quartet without a pair array.
struct SrcQuartet {
    const long long *q;   // C-order int64 (n,4)
    __device__ __forceinline__ void load(long long i, long long &a, long long &b) const
    {
        const long long quartet = i / 6;
        const int combo = (int)(i - quartet * 6);
        const int ia = (0x940 >> (2 * combo)) & 3;    // 0 0 0 1 1 2
        const int ib = (0xFB9 >> (2 * combo)) & 3;    // 1 2 3 2 3 3
        a = q[quartet * 4 + ia];
        b = q[quartet * 4 + ib];
    }
};

__device__ __forceinline__ void store_result(const DistSink &out_d, int *__restrict__ out_m,
                                             long long i, float d, int m)
{
    if (out_d.d64) out_d.d64[i] = (double)d;
    else if (out_d.f32) out_d.f32[i] = d;
    if (out_m) out_m[i] = m;
}

}  // namespace st
