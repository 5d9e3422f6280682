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

// Where MRCA ids go: int32 per pair, or -- the host path's way back on trees of fewer than 2^24 nodes -- 24 bits
// per pair: pair i at bytes [3 i, 3 i + 3), little endian, -1 (an id out of range) as 0xFFFFFF; the buffer is padded
// to a multiple of four bytes.  7 instead of 8 bytes per pair then cross the link with the float32 distance
// (scripts/micro/link_pack_bench.hip: +13 % on the link side; host_copy.h::unpack_ids24 widens them again).
struct MrcaSink {
    int *m32;
    unsigned char *m24;
    __host__ __device__ bool any() const { return m32 != nullptr || m24 != nullptr; }
};

// One id, from any lane in any control flow (scattered form): three byte stores in the packed format.
__device__ __forceinline__ void store_mrca(const MrcaSink &out, long long i, int m)
{
    if (out.m32) {
        out.m32[i] = m;
    } else if (out.m24) {
        unsigned char *p = out.m24 + 3 * i;
        p[0] = (unsigned char)m;
        p[1] = (unsigned char)(m >> 8);
        p[2] = (unsigned char)(m >> 16);
    }
}

// The ids of 64 consecutive pairs, one per lane: EVERY lane of the wave calls this in converged control flow with
// i = (a multiple of 4) + lane; lanes without a pair pass live = false.  Packed format: the four ids of a quad are
// twelve bytes = three dwords, assembled with one quad permute (lane p takes the id of lane p + 1) and stored by the
// quad's first three lanes -- a wave writes 192 consecutive bytes in aligned dwords, no byte stores (byte-masked
// partial writes are what a link to host memory handles worst).  A dword's upper bytes belong to the next pair: if
// that one is not live (the tail of a batch) or its id is not known yet (it follows by store_mrca) they are zero.
__device__ __forceinline__ void store_mrca_wave(const MrcaSink &out, long long i, int m, bool live)
{
    if (out.m32) {
        if (live) out.m32[i] = m;
    } else if (out.m24) {
        const unsigned v = live ? ((unsigned)m & 0xFFFFFFu) : 0u;
        const unsigned next = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xF9 /* quad_perm:[1,2,3,3] */, 0xF, 0xF, true);
        const unsigned p = (unsigned)i & 3u;
        if (live && p != 3u) {
            const unsigned w = (v >> (8u * p)) | (next << (24u - 8u * p));
            reinterpret_cast<unsigned *>(out.m24)[(i >> 2) * 3 + p] = w;
        }
    }
}

__device__ __forceinline__ void store_dist(const DistSink &out_d, long long i, float d)
{
    if (out_d.d64) out_d.d64[i] = (double)d;
    else if (out_d.f32) out_d.f32[i] = d;
}

// scattered form (any lane, any control flow)
__device__ __forceinline__ void store_result(const DistSink &out_d, const MrcaSink &out_m, long long i, float d, int m)
{
    store_dist(out_d, i, d);
    store_mrca(out_m, i, m);
}

// converged form (see store_mrca_wave)
__device__ __forceinline__ void store_result_wave(const DistSink &out_d, const MrcaSink &out_m, long long i, float d, int m, bool live)
{
    if (live) store_dist(out_d, i, d);
    store_mrca_wave(out_m, i, m, live);
}

}  // namespace st
