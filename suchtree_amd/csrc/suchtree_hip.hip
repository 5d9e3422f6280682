// suchtree_hip.hip -- gfx950 kernels and the C ABI of libsuchtree_hip.so.
//
// Hot path replaced: SuchTree._distances + SuchTree._mrca
// (/root/reference/SuchTree/MuchTree.pyx:911-943, 999-1030).  One wavefront
// lane per node pair.  Two kernel families, bit-identical results:
//
//   walk    pointer chase over the 8-byte {parent,dist} table with a depth
//           cut; works for any rooted tree.
//   canopy  the top of the tree lives in LDS (BFS-numbered, 8 B per node);
//           everything below it is folded into one fixed-stride understory
//           record per node, so a pair costs two record reads from HBM plus an
//           LDS climb instead of ~h dependent global gathers.
//           k_canopy      scalar, branchy (default for deep canopies)
//           k_canopy_ilp  1-2 pairs per lane, predicated (default otherwise)
//           k_canopy_flow per-lane state machine (selectable)
//
// Every kernel is templated on a pair source (SrcContig / SrcContig32 / SrcStrided /
// SrcTriangle / SrcQuartet): an explicit (n,2) array in HBM, or pairs derived from their
// index (all-pairs triangle, the six pairs of a quartet).
//
// Host side of the C ABI: tree upload (tree_prep.cpp builds the tables), the staged
// host path (host_pipe.h), the small-batch mailbox, fault read-back.
//
// Built for gfx950 only: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/suchtree_hip.h"
#include "host_pipe.h"
#include "pair_math.h"
#include "tree_prep.h"

namespace st {

// --------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define ST_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail(ST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Selects a device for the scope of one C-ABI call and puts the caller's current device
// back afterwards (callers such as PyTorch keep their own notion of "current device").
class DeviceScope {
public:
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        err_ = (prev_ == device) ? hipSuccess : hipSetDevice(device);
        changed_ = err_ == hipSuccess && prev_ != device;
    }
    ~DeviceScope()
    {
        if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
    }
    hipError_t error() const { return err_; }

private:
    int prev_ = -1;
    hipError_t err_ = hipSuccess;
    bool changed_ = false;
};

#define ST_DEVICE(dev)                                                                     \
    DeviceScope device_scope_(dev);                                                        \
    if (device_scope_.error() != hipSuccess)                                               \
        return fail(ST_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_scope_.error()))

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
struct SrcContig32 {
    const int *pairs;
    __device__ __forceinline__ void load(long long i, long long &a, long long &b) const
    {
        const int2 v = reinterpret_cast<const int2 *>(pairs)[i];
        a = v.x;
        b = v.y;
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

// --------------------------------------------------------------------------
// walk kernel
// --------------------------------------------------------------------------
struct WalkParams {
    const Node8 *nodes;
    const int32_t *depth;
    long long n_nodes;
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
            const PairResult r = pair_walk(P.nodes, P.depth, (int32_t)a, (int32_t)b);
            store_result(out_d, out_m, i, r.dist, r.mrca);
        } else {
            out_m[i] = pair_walk_mrca(P.nodes, P.depth, (int32_t)a, (int32_t)b);
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
            M[j] = pair_walk_mrca(P.nodes, P.depth, (int32_t)id[pa[j]], (int32_t)id[pb[j]]);
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

// --------------------------------------------------------------------------
// canopy kernels
// --------------------------------------------------------------------------
struct CanopyParams {
    const CanopyEntry *canopy;     // [canopy_nodes] global copy, staged to LDS
    const int32_t *canopy_id;      // [canopy_nodes]
    const uint8_t *rec_a;          // [n_nodes * 8]            {word0, pbot}
    const uint8_t *rec_b;          // [n_nodes * rec_bytes/2]  {word0, chain lengths}
    const uint8_t *rec_i;          // [n_nodes * rec_bytes/2]  {pbot, chain node ids}
    long long n_nodes;
    long long n_leaves;
    int32_t canopy_nodes;
    int32_t rec_bytes;
    int32_t parity;                // 1: leaf records first (leaves are the even ids)
};

constexpr int kCanopyBlock = 1024;
constexpr int kFlowRounds = 8;   // climb rounds between two control points of the flow kernel

// stage the canopy image into LDS: 16 bytes (two entries) per lane per step, coalesced
__device__ __forceinline__ void stage_canopy(const CanopyParams &P, unsigned char *lds_raw)
{
    const int n16 = (P.canopy_nodes + 1) / 2;
    const uint4 *src = reinterpret_cast<const uint4 *>(P.canopy);
    uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
    for (int k = threadIdx.x; k < n16; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
}

// Scalar form (one pair per lane).  CAP = chain slots per record (rec_bytes = 8*(CAP+1));
// CAP == 0 is the generic form for records longer than 128 bytes, which reads b's chain
// through a pointer.
template <int CAP, typename Src>
__global__ __launch_bounds__(kCanopyBlock) void k_canopy(CanopyParams P, Src src, long long n,
                                                         DistSink out_d,
                                                         int *__restrict__ out_m, Fault *fault)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    CanopyEntry *can = reinterpret_cast<CanopyEntry *>(lds_raw);
    stage_canopy(P, lds_raw);

    const int rec_bytes = CAP > 0 ? 8 * (CAP + 1) : P.rec_bytes;
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
        const long long sa = record_slot(a, P.parity != 0, P.n_leaves);
        const long long sb = record_slot(b, P.parity != 0, P.n_leaves);
        const uint8_t *rb = P.rec_b + sb * (rec_bytes / 2);

        // a: word0 and pbot (8 bytes of rec_a).  b: word0 + chain lengths (rec_b).
        const uint2 va = reinterpret_cast<const uint2 *>(P.rec_a)[sa];
        const uint32_t wa = va.x;
        const float pbot_a = __uint_as_float(va.y);
        uint32_t wb;
        float Db[CAP > 0 ? CAP : 1];
        if (CAP == 1) {
            const uint2 v = *reinterpret_cast<const uint2 *>(rb);
            wb = v.x;
            Db[0] = __uint_as_float(v.y);
        } else if (CAP > 1) {
            uint32_t w[CAP + 1];
#pragma unroll
            for (int q = 0; q < (CAP + 1) / 4; q++) {
                const uint4 v = reinterpret_cast<const uint4 *>(rb)[q];
                w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
            wb = w[0];
#pragma unroll
            for (int q = 0; q < CAP; q++) Db[q] = __uint_as_float(w[q + 1]);
        } else {
            wb = *reinterpret_cast<const uint32_t *>(rb);
            Db[0] = 0.0f;
        }
        const uint32_t pa = wa & 0xFFFFu, pb = wb & 0xFFFFu;
        PairResult r;
        if (pa != pb) {
            const float *dptr = CAP > 0 ? Db : reinterpret_cast<const float *>(rb + 4);
            r = pair_canopy_split<CAP>(can, P.canopy_id, pa, pbot_a, pb, dptr, wb >> 16);
        } else {
            const RecTables R{P.rec_a, P.rec_b, P.rec_i, rec_bytes / 2};
            r = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa), rec_view(R, sb));
        }
        store_result(out_d, out_m, i, r.dist, r.mrca);
    }
}

// Same computation with PPL pairs in flight per lane.  Every lane carries PPL independent
// pairs: their pair and record loads are issued together and their canopy climbs advance in
// the same loop iteration as independent ds_read_b64 (low word = dist bits, high word =
// parent index).  All updates are predicated selects (a finished climb keeps re-reading its
// meeting node), so the PPL chains never serialise behind a branch.
template <int CAP, int PPL, bool LOCKSTEP, typename Src>
__global__ __launch_bounds__(kCanopyBlock) void k_canopy_ilp(CanopyParams P, Src src, long long n,
                                                             DistSink out_d,
                                                             int *__restrict__ out_m, Fault *fault)
{
    static_assert(CAP == 1 || CAP == 3 || CAP == 7 || CAP == 15, "register-resident chains only");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const unsigned long long *can = reinterpret_cast<const unsigned long long *>(lds_raw);
    stage_canopy(P, lds_raw);

    constexpr int rec_bytes = 8 * (CAP + 1);
    const bool parity = P.parity != 0;
    const long long tile = (long long)blockDim.x * PPL;
    for (long long base = (long long)blockIdx.x * tile; base < n; base += (long long)gridDim.x * tile) {
        long long idx[PPL], sa[PPL], sb[PPL], ida[PPL], idb[PPL];
        bool live[PPL], valid[PPL];
        // all PPL pair loads are issued before anything looks at them
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const long long i = base + (long long)j * blockDim.x + threadIdx.x;
            live[j] = i < n;
            idx[j] = live[j] ? i : n - 1;
            src.load(idx[j], ida[j], idb[j]);
        }
        bool any_bad = false;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            valid[j] = (unsigned long long)ida[j] < (unsigned long long)P.n_nodes &&
                       (unsigned long long)idb[j] < (unsigned long long)P.n_nodes;
            any_bad |= !valid[j] && live[j];
            const long long a = valid[j] ? ida[j] : 0, b = valid[j] ? idb[j] : 0;
            sa[j] = record_slot(a, parity, P.n_leaves);
            sb[j] = record_slot(b, parity, P.n_leaves);
        }
        if (any_bad) {
#pragma unroll
            for (int j = 0; j < PPL; j++)
                if (!valid[j] && live[j]) record_fault(fault, ida[j], idb[j], P.n_nodes);
        }
        uint32_t u[PPL], v[PPL], pa[PPL], pb[PPL], nb[PPL];
        float s[PPL], Db[PPL][CAP];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint8_t *rb = P.rec_b + sb[j] * (rec_bytes / 2);
            const uint2 va = reinterpret_cast<const uint2 *>(P.rec_a)[sa[j]];
            const uint32_t wa = va.x;
            s[j] = __uint_as_float(va.y);
            uint32_t wb;
            if (CAP == 1) {
                const uint2 q = *reinterpret_cast<const uint2 *>(rb);
                wb = q.x;
                Db[j][0] = __uint_as_float(q.y);
            } else {
                uint32_t w[CAP + 1];
#pragma unroll
                for (int q = 0; q < (CAP + 1) / 4; q++) {
                    const uint4 x = reinterpret_cast<const uint4 *>(rb)[q];
                    w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
                }
                wb = w[0];
#pragma unroll
                for (int q = 0; q < CAP; q++) Db[j][q] = __uint_as_float(w[q + 1]);
            }
            u[j] = wa & 0xFFFFu;
            v[j] = wb & 0xFFFFu;
            pa[j] = u[j];
            pb[j] = v[j];
            nb[j] = wb >> 16;
        }

        // climb 1: find the meeting node; the a-side sum rides along
        bool go = false;
#pragma unroll
        for (int j = 0; j < PPL; j++) go |= u[j] != v[j];
        while (go) {
            go = false;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                if (LOCKSTEP) {
                    // depth cut inside the canopy: both entries are read every round, the deeper
                    // side moves (both on a tie), so the climb takes max(ka,kb) rounds, not ka+kb
                    const unsigned long long eu = can[u[j]], ev = can[v[j]];
                    const uint32_t lu = (uint32_t)(eu >> 32), lv = (uint32_t)(ev >> 32);
                    const bool act = u[j] != v[j];
                    const bool mu = act && (lu >> 16) >= (lv >> 16);
                    const bool mv = act && (lv >> 16) >= (lu >> 16);
                    const float s_next = s[j] + __uint_as_float((uint32_t)eu);
                    s[j] = mu ? s_next : s[j];
                    u[j] = mu ? (lu & kCanopyParentMask) : u[j];
                    v[j] = mv ? (lv & kCanopyParentMask) : v[j];
                } else {
                    const bool act = u[j] != v[j];
                    const bool up_a = u[j] > v[j];
                    const unsigned long long e = can[up_a ? u[j] : v[j]];
                    const uint32_t e_parent = (uint32_t)(e >> 32) & kCanopyParentMask;
                    const float s_next = s[j] + __uint_as_float((uint32_t)e);
                    s[j] = up_a ? s_next : s[j];
                    u[j] = up_a ? e_parent : u[j];
                    v[j] = (act && !up_a) ? e_parent : v[j];
                }
                go |= u[j] != v[j];
            }
        }
        // b's understory, then climb 2 over b's canopy lineage
#pragma unroll
        for (int j = 0; j < PPL; j++) {
#pragma unroll
            for (int q = 0; q < CAP; q++) {
                const float s_next = s[j] + Db[j][q];
                s[j] = (uint32_t)q < nb[j] ? s_next : s[j];
            }
            v[j] = pb[j];
        }
        go = false;
#pragma unroll
        for (int j = 0; j < PPL; j++) go |= v[j] != u[j];
        while (go) {
            go = false;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const bool act = v[j] != u[j];
                const unsigned long long e = can[v[j]];
                const float s_next = s[j] + __uint_as_float((uint32_t)e);
                s[j] = act ? s_next : s[j];
                v[j] = act ? ((uint32_t)(e >> 32) & kCanopyParentMask) : v[j];
                go |= v[j] != u[j];
            }
        }
        int m[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) m[j] = P.canopy_id[u[j]];
        // shared portal (rare for random pairs): the MRCA is the portal or below it
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (pa[j] == pb[j]) {
                const RecTables R{P.rec_a, P.rec_b, P.rec_i, rec_bytes / 2};
                const PairResult r = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa[j]), rec_view(R, sb[j]));
                s[j] = r.dist;
                m[j] = r.mrca;
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (live[j]) {
                if (valid[j]) store_result(out_d, out_m, idx[j], s[j], m[j]);
                else store_result(out_d, out_m, idx[j], __builtin_nanf(""), -1);
            }
        }
    }
}

// "Flow" form for deep canopies (selectable with the "flow" option, off by default).  On
// trees like data/bigtrees/ml.tree the climb is hundreds of LDS rounds per pair and its
// length varies 10x between pairs, so in the kernels above a wave idles ~70 % of its lanes
// waiting for its longest lineage.  Measured on ml.tree (rocprofv3 SQ counters, 5e7 pairs):
// this form issues 45 % fewer LDS instructions but 22 % more VALU and ends 7 % slower than
// k_canopy_ilp (5.6e9 vs 6.0e9 pairs/s): both sit at ~45 % VALU issue and ~55 % LDS-pipe
// occupancy with half of the LDS cycles lost to bank conflicts of the random climbs.  Here
// every lane runs its own sequence of pairs (pair index = wave tile + 64*t + lane) as a
// small state machine -- NEED -> CLIMB1 -> CLIMB2 -> DONE -- and moves on without
// waiting for its neighbours; one unified, branch-free round serves both climbs.  The
// memory-touching stages (store a finished pair, fetch the next pair and its records) run
// for a batch of waiting lanes at a time so that their latency is paid once per batch and
// hidden by the other waves of the SIMD.
template <int CAP, typename Src>
__global__ __launch_bounds__(kCanopyBlock) void k_canopy_flow(CanopyParams P, Src src, long long n,
                                                              DistSink out_d,
                                                              int *__restrict__ out_m, Fault *fault,
                                                              int batch)
{
    static_assert(CAP == 1 || CAP == 3 || CAP == 7 || CAP == 15, "register-resident chains only");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const unsigned long long *can = reinterpret_cast<const unsigned long long *>(lds_raw);
    stage_canopy(P, lds_raw);

    constexpr int rec_bytes = 8 * (CAP + 1);
    const bool parity = P.parity != 0;
    const int lane = threadIdx.x & 63;
    const long long waves_per_block = blockDim.x >> 6;
    const long long n_waves = (long long)gridDim.x * waves_per_block;
    const long long wave_id = (long long)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const long long per_wave = ((n + n_waves - 1) / n_waves + 63) / 64 * 64;   // contiguous tile
    const long long w_begin = wave_id * per_wave;
    const long long w_end = w_begin + per_wave < n ? w_begin + per_wave : n;

    enum { NEED = 0, CLIMB1 = 1, CLIMB2 = 2, DONE = 3 };
    long long my_idx = w_begin + lane;   // my next pair, stride 64
    long long cur_idx = 0;
    int ph = NEED;
    uint32_t u = 0, v = 0, pbv = 0, nbv = 0;
    float s = 0.0f;
    float Db[CAP];
#pragma unroll
    for (int q = 0; q < CAP; q++) Db[q] = 0.0f;

    for (;;) {
        const unsigned long long act_mask = __ballot(ph == CLIMB1 || ph == CLIMB2);
        const unsigned long long wait_mask = __ballot(ph == DONE || (ph == NEED && my_idx < w_end));
        if (act_mask == 0 && wait_mask == 0) break;
        if ((int)__popcll(wait_mask) >= batch || act_mask == 0) {
            if (ph == DONE) {
                store_result(out_d, out_m, cur_idx, s, P.canopy_id[u]);
                ph = NEED;
            }
            if (ph == NEED && my_idx < w_end) {
                cur_idx = my_idx;
                my_idx += 64;
                long long a, b;
                src.load(cur_idx, a, b);
                if ((unsigned long long)a >= (unsigned long long)P.n_nodes ||
                    (unsigned long long)b >= (unsigned long long)P.n_nodes) {
                    record_fault(fault, a, b, P.n_nodes);
                    store_result(out_d, out_m, cur_idx, __builtin_nanf(""), -1);
                } else {
                    const long long sa = record_slot(a, parity, P.n_leaves);
                    const long long sb = record_slot(b, parity, P.n_leaves);
                    const uint8_t *rb = P.rec_b + sb * (rec_bytes / 2);
                    const uint2 va = reinterpret_cast<const uint2 *>(P.rec_a)[sa];
                    const uint32_t wa = va.x;
                    s = __uint_as_float(va.y);
                    uint32_t wb;
                    if (CAP == 1) {
                        const uint2 q = *reinterpret_cast<const uint2 *>(rb);
                        wb = q.x;
                        Db[0] = __uint_as_float(q.y);
                    } else {
                        uint32_t w[CAP + 1];
#pragma unroll
                        for (int q = 0; q < (CAP + 1) / 4; q++) {
                            const uint4 x = reinterpret_cast<const uint4 *>(rb)[q];
                            w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
                        }
                        wb = w[0];
#pragma unroll
                        for (int q = 0; q < CAP; q++) Db[q] = __uint_as_float(w[q + 1]);
                    }
                    u = wa & 0xFFFFu;
                    v = wb & 0xFFFFu;
                    pbv = v;
                    nbv = wb >> 16;
                    if (u == v) {   // shared portal: the MRCA is the portal or below it
                        const RecTables R{P.rec_a, P.rec_b, P.rec_i, rec_bytes / 2};
                        const PairResult r = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa), rec_view(R, sb));
                        store_result(out_d, out_m, cur_idx, r.dist, r.mrca);
                    } else {
                        ph = CLIMB1;
                    }
                }
            }
        }
        // kFlowRounds rounds of whichever climb the lane is in (both entries are always read;
        // in CLIMB2 u is the meeting node and stays put).  The phase only changes at the
        // control points around this block, so a lane that meets early idles < kFlowRounds.
        {
            const bool c1 = ph == CLIMB1;
            const bool climbing = c1 || ph == CLIMB2;
#pragma unroll
            for (int r = 0; r < kFlowRounds; r++) {
                const unsigned long long eu = can[u], ev = can[v];
                const uint32_t lu = (uint32_t)(eu >> 32), lv = (uint32_t)(ev >> 32);
                const bool act = climbing && u != v;
                const bool mu = act && c1 && (lu >> 16) >= (lv >> 16);
                const bool mv = act && (!c1 || (lv >> 16) >= (lu >> 16));
                const float addend = __uint_as_float(c1 ? (uint32_t)eu : (uint32_t)ev);
                const float s_next = s + addend;
                s = (c1 ? mu : act) ? s_next : s;
                u = mu ? (lu & kCanopyParentMask) : u;
                v = mv ? (lv & kCanopyParentMask) : v;
            }
        }
        // CLIMB1 met: add b's understory, restart v at b's portal for CLIMB2
        const bool met1 = ph == CLIMB1 && u == v;
        if (__ballot(met1)) {
            if (met1) {
#pragma unroll
                for (int q = 0; q < CAP; q++) {
                    const float s_next = s + Db[q];
                    s = (uint32_t)q < nbv ? s_next : s;
                }
                v = pbv;
                ph = v == u ? DONE : CLIMB2;
            }
        } else if (ph == CLIMB2 && v == u) {
            ph = DONE;
        }
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

}  // namespace st

// --------------------------------------------------------------------------
// host side: the tree handle
// --------------------------------------------------------------------------
using namespace st;

struct st_tree {
    int device = 0;
    int strategy = ST_STRATEGY_WALK;       // family in use
    bool has_canopy = false;
    int n_cu = 256;
    st_tree_info info{};
    // device tables
    Node8 *d_nodes = nullptr;
    int32_t *d_depth = nullptr;
    CanopyEntry *d_canopy = nullptr;
    int32_t *d_canopy_id = nullptr;
    uint8_t *d_rec_a = nullptr, *d_rec_b = nullptr, *d_rec_i = nullptr;
    Fault *d_fault = nullptr;
    // canopy geometry
    int32_t canopy_nodes = 0, rec_bytes = 0, rec_cap = 0, parity = 0;
    int64_t n_nodes = 0, n_leaves = 0;
    int pairs_per_lane = 1;   // tuning: 0 = scalar (branchy) kernel, 1/2 = predicated ILP kernel with that many pairs per lane
    int lockstep = 1;         // tuning: canopy climb 1 by depth cut (1) or by "larger index moves" (0)
    int flow = 0;             // tuning: per-lane flow kernel 1 / 0 (measured: no faster than the ILP form, kept selectable)
    int flow_batch = 16;      // lanes that must be waiting before the flow kernel refills
    int canopy_depth = 0;     // deepest canopy node (edges)
    int small_batch_path = 1; // tuning: batches <= kMailboxPairs go through the pinned mailbox
    // staging of the host entry points (one caller at a time per handle)
    std::mutex ws_mutex;
    HostPipe pipe;
    void *q_tmp = nullptr;        // MRCA ids of the quartet path (6 int32 per quartet)
    int64_t q_tmp_cap = 0;
    // mailbox of the small-batch path: pinned host memory the kernel reads and writes directly
    void *mb_host = nullptr;      // [pairs int64 x2 | dist double | mrca int32] x kMailboxPairs
    void *mb_dev = nullptr;       // device alias of mb_host
    Fault *d_fault_mb = nullptr;  // scratch fault word of that path (ids are checked on the host there)
    hipStream_t mb_stream = nullptr;
};

static const Fault kFaultInit = {std::numeric_limits<long long>::min(),
                                 std::numeric_limits<long long>::max()};

static size_t canopy_lds_bytes(const st_tree *t)
{
    return (size_t)((t->canopy_nodes + 1) / 2) * 16;
}

template <typename Kern, typename Src>
static hipError_t launch_canopy_k(Kern kern, int ppl, const st_tree *t, const CanopyParams &P,
                                  const Src &src, int64_t n, DistSink out_d, int32_t *out_m,
                                  hipStream_t stream)
{
    const size_t lds = canopy_lds_bytes(t);
    if (lds > 64 * 1024) {
        // dynamic LDS above 64 KiB has to be granted per kernel (cheap host-side call)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // one or two 1024-lane workgroups per CU, whatever the LDS image allows
    const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int64_t tile = (int64_t)kCanopyBlock * ppl;
    int64_t blocks = (n + tile - 1) / tile;
    blocks = std::min<int64_t>(blocks, (int64_t)t->n_cu * wg_per_cu);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kCanopyBlock), lds, stream, P, src,
                       (long long)n, out_d, out_m, t->d_fault);
    return hipGetLastError();
}

template <int CAP, typename Src>
static hipError_t launch_canopy_flow(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n,
                                     DistSink out_d, int32_t *out_m, hipStream_t stream)
{
    const size_t lds = canopy_lds_bytes(t);
    auto kern = k_canopy_flow<CAP, Src>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
    // every wave should own at least a few pairs per lane
    int64_t blocks = (n + (int64_t)kCanopyBlock * 8 - 1) / ((int64_t)kCanopyBlock * 8);
    blocks = std::min<int64_t>(blocks, (int64_t)t->n_cu * wg_per_cu);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kCanopyBlock), lds, stream, P, src,
                       (long long)n, out_d, out_m, t->d_fault, t->flow_batch);
    return hipGetLastError();
}

template <int CAP, typename Src>
static hipError_t launch_canopy_t(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n,
                                  DistSink out_d, int32_t *out_m, hipStream_t stream)
{
    if constexpr (CAP != 0) {
        if (t->flow == 1) return launch_canopy_flow<CAP>(t, P, src, n, out_d, out_m, stream);
    }
    if constexpr (CAP == 0) {
        return launch_canopy_k(k_canopy<0, Src>, 1, t, P, src, n, out_d, out_m, stream);
    } else {
        const int ppl = std::min(t->pairs_per_lane, 2);
        if (ppl == 0) return launch_canopy_k(k_canopy<CAP, Src>, 1, t, P, src, n, out_d, out_m, stream);
        if (t->lockstep) {
            if (ppl == 1) return launch_canopy_k(k_canopy_ilp<CAP, 1, true, Src>, 1, t, P, src, n, out_d, out_m, stream);
            return launch_canopy_k(k_canopy_ilp<CAP, 2, true, Src>, 2, t, P, src, n, out_d, out_m, stream);
        } else {
            if (ppl == 1) return launch_canopy_k(k_canopy_ilp<CAP, 1, false, Src>, 1, t, P, src, n, out_d, out_m, stream);
            return launch_canopy_k(k_canopy_ilp<CAP, 2, false, Src>, 2, t, P, src, n, out_d, out_m, stream);
        }
        return hipErrorInvalidValue;
    }
}

template <typename Src>
static hipError_t launch_canopy(const st_tree *t, const Src &src, int64_t n, DistSink out_d,
                                int32_t *out_m, hipStream_t stream)
{
    CanopyParams P;
    P.canopy = t->d_canopy;
    P.canopy_id = t->d_canopy_id;
    P.rec_a = t->d_rec_a;
    P.rec_b = t->d_rec_b;
    P.rec_i = t->d_rec_i;
    P.n_nodes = t->n_nodes;
    P.n_leaves = t->n_leaves;
    P.canopy_nodes = t->canopy_nodes;
    P.rec_bytes = t->rec_bytes;
    P.parity = t->parity;
    switch (t->rec_cap) {
        case 1: return launch_canopy_t<1>(t, P, src, n, out_d, out_m, stream);
        case 3: return launch_canopy_t<3>(t, P, src, n, out_d, out_m, stream);
        case 7: return launch_canopy_t<7>(t, P, src, n, out_d, out_m, stream);
        case 15: return launch_canopy_t<15>(t, P, src, n, out_d, out_m, stream);
        default: return launch_canopy_t<0>(t, P, src, n, out_d, out_m, stream);
    }
}

template <typename Src>
static hipError_t launch_walk(const st_tree *t, const Src &src, int64_t n, DistSink out_d,
                              int32_t *out_m, hipStream_t stream)
{
    WalkParams P;
    P.nodes = t->d_nodes;
    P.depth = t->d_depth;
    P.n_nodes = t->n_nodes;
    int64_t blocks = (n + 255) / 256;
    blocks = std::min<int64_t>(blocks, (int64_t)t->n_cu * 16);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(k_walk<Src>, dim3((unsigned)blocks), dim3(256), 0, stream, P, src,
                       (long long)n, out_d, out_m, t->d_fault);
    return hipGetLastError();
}

// Small batches are not worth staging 128 KiB of canopy per workgroup.
constexpr int64_t kCanopyMinPairs = 4096;

template <typename Src>
static int enqueue_src(st_tree *t, const Src &src, int64_t n, DistSink d_out, int32_t *d_mrca,
                       hipStream_t stream)
{
    if (n == 0) return ST_OK;
    // MRCA-only requests (d_out == NULL) also go through the canopy kernels: the id comes out
    // of the same climb, and that is ~7x faster than walking the global table
    const bool canopy = t->strategy == ST_STRATEGY_CANOPY && n >= kCanopyMinPairs;
    const hipError_t e = canopy ? launch_canopy(t, src, n, d_out, d_mrca, stream)
                                : launch_walk(t, src, n, d_out, d_mrca, stream);
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return ST_OK;
}

static int enqueue(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t s0, int64_t s1,
                   DistSink d_out, int32_t *d_mrca, hipStream_t stream)
{
    const long long *p = reinterpret_cast<const long long *>(d_pairs);
    if (s0 == 2 && s1 == 1 && (reinterpret_cast<uintptr_t>(d_pairs) & 15) == 0)
        return enqueue_src(t, SrcContig{p}, n, d_out, d_mrca, stream);
    return enqueue_src(t, SrcStrided{p, (long long)s0, (long long)s1}, n, d_out, d_mrca, stream);
}

// host_max / host_min: exact extremes of ids the host path had to clamp to 32 bits.
static int read_fault(st_tree *t, hipStream_t stream, int64_t *bad_id,
                      long long host_max = std::numeric_limits<long long>::min(),
                      long long host_min = std::numeric_limits<long long>::max())
{
    Fault f;
    ST_HIP(hipMemcpyAsync(&f, t->d_fault, sizeof(Fault), hipMemcpyDeviceToHost, stream));
    ST_HIP(hipStreamSynchronize(stream));
    if (f.max_bad == kFaultInit.max_bad && f.min_bad == kFaultInit.min_bad) return ST_OK;
    f.max_bad = std::max(f.max_bad, host_max);
    f.min_bad = std::min(f.min_bad, host_min);
    ST_HIP(hipMemcpyAsync(t->d_fault, &kFaultInit, sizeof(Fault), hipMemcpyHostToDevice, stream));
    ST_HIP(hipStreamSynchronize(stream));
    // the reference reports max_id when it is too large, else min_id (MuchTree.pyx:897-903)
    const long long bad = f.max_bad >= t->n_nodes ? f.max_bad : f.min_bad;
    if (bad_id) *bad_id = bad;
    return fail(ST_ERR_BOUNDS, "Node ID " + std::to_string(bad) + " out of bounds (tree size: " +
                                   std::to_string(t->n_nodes) + ")");
}

template <typename T>
static int upload(T **dst, const std::vector<T> &src, int64_t *bytes)
{
    const size_t sz = std::max<size_t>(src.size() * sizeof(T), 16);
    ST_HIP(hipMalloc(reinterpret_cast<void **>(dst), sz));
    if (!src.empty()) ST_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes += (int64_t)sz;
    return ST_OK;
}

constexpr int64_t kHostChunk = (int64_t)1 << 22;   // pairs per pipeline chunk
constexpr int kDeepCanopyDepth = 100;     // canopies deeper than this (edges) are "deep"
constexpr int kDeepCanopyNodes = 10240;   // 80 KiB LDS image: two 1024-lane workgroups per CU
constexpr int64_t kMailboxPairs = 2048;   // largest batch served through the mailbox

// Small batches (a scalar distance(a,b) call is a batch of one) are all latency: instead of
// H2D copy + kernel + D2H copy + fault read-back, the walk kernel reads the pairs from and
// writes the results to pinned host memory mapped into the device, so a call is one launch
// and one stream synchronisation.  Ids are range-checked here on the host (the batch is
// tiny), with the reference's choice of the id to report (MuchTree.pyx:897-903).
template <typename Id>
static int small_batch(st_tree *t, const Id *pairs, int64_t n, int64_t stride0, int64_t stride1,
                       double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    if (!t->mb_host) {
        const size_t bytes = (size_t)kMailboxPairs * (16 + 8 + 4);
        ST_HIP(hipHostMalloc(&t->mb_host, bytes, hipHostMallocMapped));
        ST_HIP(hipHostGetDevicePointer(&t->mb_dev, t->mb_host, 0));
        ST_HIP(hipMalloc(reinterpret_cast<void **>(&t->d_fault_mb), sizeof(Fault)));
        ST_HIP(hipStreamCreateWithFlags(&t->mb_stream, hipStreamNonBlocking));
    }
    int64_t *h_pairs = static_cast<int64_t *>(t->mb_host);
    double *h_dist = reinterpret_cast<double *>(h_pairs + 2 * kMailboxPairs);
    int32_t *h_mrca = reinterpret_cast<int32_t *>(h_dist + kMailboxPairs);
    long long max_id = std::numeric_limits<long long>::min(), min_id = std::numeric_limits<long long>::max();
    for (int64_t k = 0; k < n; k++) {
        const long long a = pairs[k * stride0], b = pairs[k * stride0 + stride1];
        h_pairs[2 * k] = a;
        h_pairs[2 * k + 1] = b;
        max_id = std::max(max_id, std::max(a, b));
        min_id = std::min(min_id, std::min(a, b));
    }
    if (min_id < 0 || max_id >= t->n_nodes) {
        const long long bad = max_id >= t->n_nodes ? max_id : min_id;
        if (bad_id) *bad_id = bad;
        return fail(ST_ERR_BOUNDS, "Node ID " + std::to_string(bad) + " out of bounds (tree size: " +
                                       std::to_string(t->n_nodes) + ")");
    }
    char *d_base = static_cast<char *>(t->mb_dev);
    WalkParams P;
    P.nodes = t->d_nodes;
    P.depth = t->d_depth;
    P.n_nodes = t->n_nodes;
    const SrcContig src{reinterpret_cast<const long long *>(d_base)};
    double *d_dist = reinterpret_cast<double *>(d_base + (size_t)kMailboxPairs * 16);
    int32_t *d_mrca = reinterpret_cast<int32_t *>(d_base + (size_t)kMailboxPairs * 24);
    hipLaunchKernelGGL(k_walk<SrcContig>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, t->mb_stream, P, src,
                       (long long)n, DistSink{out_dist ? d_dist : nullptr, nullptr},
                       out_mrca ? d_mrca : nullptr, t->d_fault_mb);
    ST_HIP(hipGetLastError());
    ST_HIP(hipStreamSynchronize(t->mb_stream));
    if (out_dist) std::memcpy(out_dist, h_dist, (size_t)n * 8);
    if (out_mrca) std::memcpy(out_mrca, h_mrca, (size_t)n * 4);
    return ST_OK;
}

// Push n pairs through the two-slot pipe (host_pipe.h).  pack(slot, off, m) fills
// slot.h_in for chunk [off, off+m) with in_bytes_per_pair bytes per pair (0: generated pairs, no input);
// launch(slot, off, m) enqueues the kernel on slot.stream reading slot.d_in and writing
// slot.d_d / slot.d_m.  Caller holds ws_mutex.
template <typename Pack, typename Launch>
static int run_pipe(st_tree *t, int64_t n, int in_bytes_per_pair, Pack pack, Launch launch,
                    double *out_dist, int32_t *out_mrca)
{
    HostPipe &P = t->pipe;
    const int64_t chunk = std::min<int64_t>(n, kHostChunk);
    {
        const hipError_t e = P.ensure(std::max<int64_t>(chunk, 1024));
        if (e != hipSuccess) {
            P.release_buffers();
            return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
        }
    }
    auto drain = [&](PipeSlot &s) -> hipError_t {
        if (!s.busy) return hipSuccess;
        s.busy = false;
        const hipError_t e = hipEventSynchronize(s.done);
        if (e != hipSuccess) return e;
        if (out_dist) {
            // distances crossed PCIe as float32; widen into the caller's float64 array
            const float *src = static_cast<const float *>(s.h_d);
            double *dst = out_dist + s.off;
            P.pool.parallel_for(s.m, [=](int64_t b, int64_t e) {
                for (int64_t k = b; k < e; k++) dst[k] = (double)src[k];
            });
        }
        if (out_mrca) P.pool.copy(out_mrca + s.off, s.h_m, s.m * 4);
        return hipSuccess;
    };
    auto bail = [&](int code, const std::string &msg) {
        for (auto &s : P.slot) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            s.busy = false;
        }
        return fail(code, msg);
    };
    int64_t c = 0;
    for (int64_t off = 0; off < n; off += chunk, c++) {
        const int64_t m = std::min(chunk, n - off);
        PipeSlot &s = P.slot[c & 1];
        hipError_t e = drain(s);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
        if (in_bytes_per_pair > 0) {
            pack(s, off, m);
            e = hipMemcpyAsync(s.d_in, s.h_in, (size_t)m * (size_t)in_bytes_per_pair, hipMemcpyHostToDevice, s.stream);
            if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("H2D: ") + hipGetErrorString(e));
        }
        const int rc = launch(s, off, m);
        if (rc != ST_OK) return bail(rc, g_last_error);
        if (out_dist && e == hipSuccess)
            e = hipMemcpyAsync(s.h_d, s.d_d, (size_t)m * 4, hipMemcpyDeviceToHost, s.stream);
        if (out_mrca && e == hipSuccess)
            e = hipMemcpyAsync(s.h_m, s.d_m, (size_t)m * 4, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess) e = hipEventRecord(s.done, s.stream);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("D2H: ") + hipGetErrorString(e));
        s.busy = true;
        s.off = off;
        s.m = m;
        e = drain(P.slot[(c + 1) & 1]);   // unpack the previous chunk while this one is in flight
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
    }
    for (auto &s : P.slot) {
        const hipError_t e = drain(s);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
    }
    return ST_OK;
}

extern "C" {

const char *st_last_error(void) { return g_last_error.c_str(); }

int st_device_count(int *count)
{
    if (!count) return fail(ST_ERR_ARG, "count is NULL");
    *count = 0;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ST_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return ST_OK;
}

int st_host_depths(const int32_t *parent, int64_t n_nodes, int32_t *out_depths, int32_t *out_tree_depth)
{
    if (!parent || n_nodes <= 0) return fail(ST_ERR_ARG, "parent is NULL or n_nodes <= 0");
    std::vector<float> zeros((size_t)n_nodes, 0.0f);
    TreeTables T;
    std::string err;
    if (!prepare_basic(parent, zeros.data(), n_nodes, T, err)) return fail(ST_ERR_TREE, err);
    if (out_depths) std::memcpy(out_depths, T.depth.data(), (size_t)n_nodes * 4);
    if (out_tree_depth) *out_tree_depth = T.tree_depth;
    return ST_OK;
}

int st_tree_create(const int32_t *parent, const float *distance, int64_t n_nodes, int device,
                   int strategy, st_tree **out)
{
    if (!out) return fail(ST_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!parent || !distance || n_nodes <= 0) return fail(ST_ERR_ARG, "parent/distance NULL or n_nodes <= 0");
    if (strategy != ST_STRATEGY_AUTO && strategy != ST_STRATEGY_WALK && strategy != ST_STRATEGY_CANOPY)
        return fail(ST_ERR_ARG, "unknown strategy " + std::to_string(strategy));

    TreeTables T;
    std::string err;
    if (!prepare_basic(parent, distance, n_nodes, T, err)) return fail(ST_ERR_TREE, err);
    bool canopy_ok = false;
    int max_canopy = 0;
    bool deep = false;
    if (const char *env = std::getenv("SUCHTREE_AMD_CANOPY_NODES")) max_canopy = std::atoi(env);   // tuning experiments
    if (strategy != ST_STRATEGY_WALK) {
        canopy_ok = prepare_canopy(parent, distance, T, max_canopy);
        // Deep canopies (real, unbalanced phylogenies: hundreds of levels) spend their time in
        // the LDS climb, not in memory.  There a canopy image small enough for two workgroups
        // per CU, a longer understory (more of each lineage pre-summed in its record) and the
        // branchy scalar kernel (finished lanes stop issuing LDS reads) measured 13-30 % faster.
        if (canopy_ok && max_canopy == 0) {
            int cdepth = 0;
            for (const CanopyEntry &e : T.canopy) cdepth = std::max<int>(cdepth, (int)(e.link >> 16));
            if (cdepth > kDeepCanopyDepth) {
                deep = true;
                if (T.canopy_nodes > kDeepCanopyNodes) {
                    TreeTables T2 = T;
                    if (prepare_canopy(parent, distance, T2, kDeepCanopyNodes)) T = std::move(T2);
                }
            }
        }
    }
    if (strategy == ST_STRATEGY_CANOPY && !canopy_ok)
        return fail(ST_ERR_TREE, "tree does not admit the canopy family (understory deeper than a record)");

    int n_dev = 0;
    ST_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev)
        return fail(ST_ERR_HIP, "device " + std::to_string(device) + " not available (" +
                                    std::to_string(n_dev) + " visible)");
    ST_DEVICE(device);
    hipDeviceProp_t prop;
    ST_HIP(hipGetDeviceProperties(&prop, device));

    st_tree *t = new (std::nothrow) st_tree();
    if (!t) return fail(ST_ERR_NOMEM, "out of host memory");
    t->device = device;
    t->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    t->n_nodes = T.n;
    t->n_leaves = T.n_leaves;
    if (deep) t->pairs_per_lane = 0;
    int64_t bytes = 0;
    int rc = upload(&t->d_nodes, T.nodes, &bytes);
    if (rc == ST_OK) rc = upload(&t->d_depth, T.depth, &bytes);
    if (rc == ST_OK && canopy_ok) {
        t->has_canopy = true;
        t->canopy_nodes = T.canopy_nodes;
        t->rec_bytes = T.record_bytes;
        t->rec_cap = T.record_cap;
        t->parity = T.parity_layout ? 1 : 0;
        for (const CanopyEntry &e : T.canopy) t->canopy_depth = std::max<int>(t->canopy_depth, (int)(e.link >> 16));
        if (T.canopy.size() & 1) T.canopy.push_back(CanopyEntry{0.0f, 0u});   // 16-byte staging granule
        rc = upload(&t->d_canopy, T.canopy, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_canopy_id, T.canopy_id, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_rec_a, T.rec_a, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_rec_b, T.rec_b, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_rec_i, T.rec_i, &bytes);
    }
    if (rc == ST_OK) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_fault), sizeof(Fault));
        if (e == hipSuccess) e = hipMemcpy(t->d_fault, &kFaultInit, sizeof(Fault), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(ST_ERR_HIP, std::string("tree setup: ") + hipGetErrorString(e));
    }
    if (rc != ST_OK) {
        std::string keep = g_last_error;
        st_tree_destroy(t);
        g_last_error = keep;
        return rc;
    }
    t->strategy = canopy_ok ? ST_STRATEGY_CANOPY : ST_STRATEGY_WALK;
    t->info.n_nodes = T.n;
    t->info.n_leaves = T.n_leaves;
    t->info.root = T.root;
    t->info.depth = T.tree_depth;
    t->info.device = device;
    t->info.canopy_nodes = canopy_ok ? T.canopy_nodes : 0;
    t->info.understory_max = canopy_ok ? T.understory_max : 0;
    t->info.record_bytes = canopy_ok ? T.record_bytes : 0;
    t->info.device_bytes = bytes;
    *out = t;
    return ST_OK;
}

void st_tree_destroy(st_tree *t)
{
    if (!t) return;
    DeviceScope scope(t->device);
    (void)hipFree(t->d_nodes);
    (void)hipFree(t->d_depth);
    (void)hipFree(t->d_canopy);
    (void)hipFree(t->d_canopy_id);
    (void)hipFree(t->d_rec_a);
    (void)hipFree(t->d_rec_b);
    (void)hipFree(t->d_rec_i);
    (void)hipFree(t->d_fault);
    t->pipe.destroy();
    (void)hipFree(t->q_tmp);
    if (t->mb_host) (void)hipHostFree(t->mb_host);
    (void)hipFree(t->d_fault_mb);
    if (t->mb_stream) (void)hipStreamDestroy(t->mb_stream);
    delete t;
}

int st_tree_info_get(const st_tree *t, st_tree_info *info)
{
    if (!t || !info) return fail(ST_ERR_ARG, "tree or info is NULL");
    *info = t->info;
    info->strategy = t->strategy;
    return ST_OK;
}

int st_tree_set_strategy(st_tree *t, int strategy)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (strategy == ST_STRATEGY_AUTO) strategy = t->has_canopy ? ST_STRATEGY_CANOPY : ST_STRATEGY_WALK;
    if (strategy == ST_STRATEGY_CANOPY && !t->has_canopy)
        return fail(ST_ERR_ARG, "tree was built without canopy tables");
    if (strategy != ST_STRATEGY_CANOPY && strategy != ST_STRATEGY_WALK)
        return fail(ST_ERR_ARG, "unknown strategy " + std::to_string(strategy));
    t->strategy = strategy;
    return ST_OK;
}

int st_tree_set_option(st_tree *t, const char *name, int64_t value)
{
    if (!t || !name) return fail(ST_ERR_ARG, "tree or name is NULL");
    if (std::strcmp(name, "pairs_per_lane") == 0) {
        if (value != 0 && value != 1 && value != 2)
            return fail(ST_ERR_ARG, "pairs_per_lane must be 0, 1 or 2");
        t->pairs_per_lane = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "small_batch_path") == 0) {
        t->small_batch_path = value != 0;
        return ST_OK;
    }
    if (std::strcmp(name, "flow") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "flow must be 0 or 1");
        t->flow = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "flow_batch") == 0) {
        if (value < 1 || value > 64) return fail(ST_ERR_ARG, "flow_batch must be in [1, 64]");
        t->flow_batch = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "lockstep") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "lockstep must be 0 or 1");
        t->lockstep = (int)value;
        return ST_OK;
    }
    return fail(ST_ERR_ARG, std::string("unknown option ") + name);
}

int st_distances_device(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t stride0,
                        int64_t stride1, double *d_out_dist, int32_t *d_out_mrca, void *stream)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !d_pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!d_out_dist && !d_out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    ST_DEVICE(t->device);
    return enqueue(t, d_pairs, n, stride0, stride1, DistSink{d_out_dist, nullptr}, d_out_mrca,
                   reinterpret_cast<hipStream_t>(stream));
}

int st_distances_device_f32(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t stride0,
                            int64_t stride1, float *d_out_dist, int32_t *d_out_mrca, void *stream)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !d_pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!d_out_dist && !d_out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    ST_DEVICE(t->device);
    return enqueue(t, d_pairs, n, stride0, stride1, DistSink{nullptr, d_out_dist}, d_out_mrca,
                   reinterpret_cast<hipStream_t>(stream));
}

int st_fault_check(st_tree *t, void *stream, int64_t *bad_id)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    ST_DEVICE(t->device);
    return read_fault(t, reinterpret_cast<hipStream_t>(stream), bad_id);
}

}  // extern "C"

// Host-buffer entry point for int64 ids (the reference's dtype) and int32 ids.
template <typename Id>
static int distances_host_impl(st_tree *t, const Id *pairs, int64_t n, int64_t stride0, int64_t stride1,
                               double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!out_dist && !out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    if (n == 0) return ST_OK;
    ST_DEVICE(t->device);
    std::lock_guard<std::mutex> lock(t->ws_mutex);
    if (n <= kMailboxPairs && t->small_batch_path)
        return small_batch(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);

    // Ids cross PCIe as int32 (half the H2D bytes).  Values that do not fit are clamped to
    // INT32_MAX / INT32_MIN -- still out of range for the kernel -- and their exact extremes
    // are kept here so that the reported id is the one the reference would report.
    CopyPool &pool = t->pipe.pool;
    std::mutex wide_mutex;
    long long wide_max = std::numeric_limits<long long>::min();
    long long wide_min = std::numeric_limits<long long>::max();
    auto pack = [&](PipeSlot &s, int64_t off, int64_t m) {
        const Id *src = pairs + off * stride0;
        int32_t *dst = static_cast<int32_t *>(s.h_in);
        if (sizeof(Id) == 4 && stride0 == 2 && stride1 == 1) {   // int32 C-order: already the wire format
            pool.copy(dst, src, m * 8);
            return;
        }
        pool.parallel_for(m, [&, src, dst](int64_t b, int64_t e) {
            long long hi = std::numeric_limits<long long>::min(), lo = std::numeric_limits<long long>::max();
            for (int64_t k = b; k < e; k++) {
                for (int c = 0; c < 2; c++) {
                    const long long v = src[k * stride0 + c * stride1];
                    int32_t w = (int32_t)v;
                    if (v > INT32_MAX) { w = INT32_MAX; hi = std::max(hi, v); }
                    else if (v < INT32_MIN) { w = INT32_MIN; lo = std::min(lo, v); }
                    dst[2 * k + c] = w;
                }
            }
            if (hi != std::numeric_limits<long long>::min() || lo != std::numeric_limits<long long>::max()) {
                std::lock_guard<std::mutex> g(wide_mutex);
                wide_max = std::max(wide_max, hi);
                wide_min = std::min(wide_min, lo);
            }
        });
    };
    auto launch = [&](PipeSlot &s, int64_t, int64_t m) {
        return enqueue_src(t, SrcContig32{static_cast<const int *>(s.d_in)}, m,
                           DistSink{nullptr, out_dist ? static_cast<float *>(s.d_d) : nullptr},
                           out_mrca ? static_cast<int32_t *>(s.d_m) : nullptr, s.stream);
    };
    const int rc = run_pipe(t, n, 8, pack, launch, out_dist, out_mrca);
    if (rc != ST_OK) return rc;
    return read_fault(t, t->pipe.slot[0].stream, bad_id, wide_max, wide_min);
}

extern "C" {

int st_distances_host(st_tree *t, const int64_t *pairs, int64_t n, int64_t stride0, int64_t stride1,
                      double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    return distances_host_impl(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);
}

int st_distances_host_i32(st_tree *t, const int32_t *pairs, int64_t n, int64_t stride0, int64_t stride1,
                          double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    return distances_host_impl(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);
}

static int triangle_args(st_tree *t, const int64_t *ids, int64_t m, int64_t k_begin, int64_t k_count,
                         const void *out_d, const void *out_m)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (m < 0 || k_begin < 0 || k_count < 0) return fail(ST_ERR_ARG, "negative size");
    if (m > 3000000000LL) return fail(ST_ERR_ARG, "m too large");
    const int64_t total = m * (m - 1) / 2;
    if (k_begin + k_count > total) return fail(ST_ERR_ARG, "pair range exceeds m(m-1)/2");
    if (k_count > 0 && !ids) return fail(ST_ERR_ARG, "ids is NULL");
    if (!out_d && !out_m) return fail(ST_ERR_ARG, "both outputs are NULL");
    return ST_OK;
}

int st_triangle_device(st_tree *t, const int64_t *d_ids, int64_t m, int64_t id_stride,
                       int64_t k_begin, int64_t k_count, double *d_out_dist, int32_t *d_out_mrca,
                       void *stream)
{
    int rc = triangle_args(t, d_ids, m, k_begin, k_count, d_out_dist, d_out_mrca);
    if (rc != ST_OK) return rc;
    ST_DEVICE(t->device);
    const SrcTriangle src{reinterpret_cast<const long long *>(d_ids), (long long)id_stride, (long long)k_begin};
    return enqueue_src(t, src, k_count, DistSink{d_out_dist, nullptr}, d_out_mrca,
                       reinterpret_cast<hipStream_t>(stream));
}

int st_triangle_host(st_tree *t, const int64_t *ids, int64_t m, int64_t id_stride, int64_t k_begin,
                     int64_t k_count, double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    int rc = triangle_args(t, ids, m, k_begin, k_count, out_dist, out_mrca);
    if (rc != ST_OK) return rc;
    if (k_count == 0) return ST_OK;
    ST_DEVICE(t->device);
    std::lock_guard<std::mutex> lock(t->ws_mutex);
    {
        hipError_t e = t->pipe.ensure(std::max<int64_t>(std::min<int64_t>(k_count, kHostChunk), 1024));
        if (e == hipSuccess) e = t->pipe.ensure_ids(m);
        if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
    }
    // the id list goes up once (packed); results stream back through the pipe
    std::vector<int64_t> packed;
    const int64_t *src_ids = ids;
    if (id_stride != 1) {
        packed.resize((size_t)m);
        for (int64_t i = 0; i < m; i++) packed[(size_t)i] = ids[i * id_stride];
        src_ids = packed.data();
    }
    ST_HIP(hipMemcpy(t->pipe.d_ids, src_ids, (size_t)m * 8, hipMemcpyHostToDevice));
    auto pack = [](PipeSlot &, int64_t, int64_t) {};
    auto launch = [&](PipeSlot &s, int64_t off, int64_t c) {
        const SrcTriangle src{static_cast<const long long *>(t->pipe.d_ids), 1, (long long)(k_begin + off)};
        return enqueue_src(t, src, c, DistSink{nullptr, out_dist ? static_cast<float *>(s.d_d) : nullptr},
                           out_mrca ? static_cast<int32_t *>(s.d_m) : nullptr, s.stream);
    };
    rc = run_pipe(t, k_count, 0, pack, launch, out_dist, out_mrca);
    if (rc != ST_OK) return rc;
    return read_fault(t, t->pipe.slot[0].stream, bad_id);
}

int st_quartets_host(st_tree *t, const int64_t *quartets, int64_t n, int64_t stride0, int64_t stride1,
                     int64_t *out_topologies, int64_t *bad_id)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && (!quartets || !out_topologies)) return fail(ST_ERR_ARG, "quartets or output is NULL");
    if (n == 0) return ST_OK;
    ST_DEVICE(t->device);
    std::lock_guard<std::mutex> lock(t->ws_mutex);
    // (n,4) int64 in and out are each the size of two pair rows: reuse the pipe's slots,
    // input in d_in/h_in of slot 0 and 1 back to back is not possible, so stage by halves:
    const int64_t chunk = std::min<int64_t>(n, kHostChunk / 2);
    {
        const hipError_t e = t->pipe.ensure(std::max<int64_t>(2 * chunk, 1024));
        if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
    }
    PipeSlot &in = t->pipe.slot[0], &out = t->pipe.slot[1];
    WalkParams P;
    P.nodes = t->d_nodes;
    P.depth = t->d_depth;
    P.n_nodes = t->n_nodes;
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = std::min(chunk, n - off);
        int64_t *h = static_cast<int64_t *>(in.h_in);
        const int64_t *src = quartets + off * stride0;
        t->pipe.pool.parallel_for(m, [=](int64_t b, int64_t e) {
            for (int64_t k = b; k < e; k++)
                for (int c = 0; c < 4; c++) h[4 * k + c] = src[k * stride0 + c * stride1];
        });
        ST_HIP(hipMemcpyAsync(in.d_in, in.h_in, (size_t)m * 32, hipMemcpyHostToDevice, in.stream));
        int64_t blocks = std::min<int64_t>((m + 255) / 256, (int64_t)t->n_cu * 16);
        if (t->strategy == ST_STRATEGY_CANOPY && 6 * m >= kCanopyMinPairs) {
            // six MRCA ids per quartet out of the canopy kernels, then the pick
            if (t->q_tmp_cap < m) {
                (void)hipFree(t->q_tmp);
                t->q_tmp = nullptr;
                t->q_tmp_cap = 0;
                ST_HIP(hipMalloc(&t->q_tmp, (size_t)chunk * 24));
                t->q_tmp_cap = chunk;
            }
            const int rc = enqueue_src(t, SrcQuartet{static_cast<const long long *>(in.d_in)}, 6 * m,
                                       DistSink{nullptr, nullptr}, static_cast<int32_t *>(t->q_tmp), in.stream);
            if (rc != ST_OK) return rc;
            hipLaunchKernelGGL(k_quartet_pick, dim3((unsigned)blocks), dim3(256), 0, in.stream,
                               static_cast<const long long *>(in.d_in), static_cast<const int *>(t->q_tmp),
                               (long long)m, static_cast<long long *>(out.d_in));
        } else {
            hipLaunchKernelGGL(k_quartets, dim3((unsigned)blocks), dim3(256), 0, in.stream, P,
                               static_cast<const long long *>(in.d_in), (long long)m, 4LL, 1LL,
                               static_cast<long long *>(out.d_in), t->d_fault);
        }
        ST_HIP(hipGetLastError());
        ST_HIP(hipMemcpyAsync(out.h_in, out.d_in, (size_t)m * 32, hipMemcpyDeviceToHost, in.stream));
        ST_HIP(hipStreamSynchronize(in.stream));
        t->pipe.pool.copy(out_topologies + off * 4, out.h_in, m * 32);
    }
    return read_fault(t, in.stream, bad_id);
}

int st_graph_matrices_host(int device, int64_t n, int64_t n_edges, const int32_t *u, const int32_t *v,
                           const double *w, double *out_adjacency, double *out_laplacian)
{
    if (n <= 0 || n_edges < 0) return fail(ST_ERR_ARG, "bad sizes");
    if (n_edges > 0 && (!u || !v || !w)) return fail(ST_ERR_ARG, "edge arrays are NULL");
    if (!out_adjacency && !out_laplacian) return fail(ST_ERR_ARG, "both outputs are NULL");
    if (n > 100000) return fail(ST_ERR_ARG, "dense n x n matrix too large");
    for (int64_t e = 0; e < n_edges; e++)
        if (u[e] < 0 || u[e] >= n || v[e] < 0 || v[e] >= n) return fail(ST_ERR_ARG, "edge endpoint out of range");
    ST_DEVICE(device);
    const size_t mat = (size_t)n * (size_t)n * 8;
    double *d_A = nullptr, *d_L = nullptr, *d_deg = nullptr, *d_w = nullptr;
    int *d_u = nullptr, *d_v = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_A), mat);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_L), mat);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_deg), (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_u), (size_t)std::max<int64_t>(n_edges, 1) * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_v), (size_t)std::max<int64_t>(n_edges, 1) * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_w), (size_t)std::max<int64_t>(n_edges, 1) * 8);
    if (e == hipSuccess) e = hipMemset(d_A, 0, mat);
    if (e == hipSuccess && n_edges) e = hipMemcpy(d_u, u, (size_t)n_edges * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_edges) e = hipMemcpy(d_v, v, (size_t)n_edges * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_edges) e = hipMemcpy(d_w, w, (size_t)n_edges * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (n_edges)
            hipLaunchKernelGGL(k_graph_scatter, dim3((unsigned)std::min<int64_t>((n_edges + 255) / 256, 4096)),
                               dim3(256), 0, nullptr, d_A, (long long)n, (long long)n_edges, d_u, d_v, d_w);
        hipLaunchKernelGGL(k_graph_degree, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0,
                           nullptr, d_A, (long long)n, d_deg);
        hipLaunchKernelGGL(k_graph_laplacian, dim3((unsigned)std::min<int64_t>((n * n + 255) / 256, 65536)),
                           dim3(256), 0, nullptr, d_A, d_deg, (long long)n, d_L);
        e = hipGetLastError();
    }
    if (e == hipSuccess && out_adjacency) e = hipMemcpy(out_adjacency, d_A, mat, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_laplacian) e = hipMemcpy(out_laplacian, d_L, mat, hipMemcpyDeviceToHost);
    (void)hipFree(d_A); (void)hipFree(d_L); (void)hipFree(d_deg);
    (void)hipFree(d_u); (void)hipFree(d_v); (void)hipFree(d_w);
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("graph matrices: ") + hipGetErrorString(e));
    return ST_OK;
}

int st_device_malloc(int device, int64_t bytes, void **out)
{
    if (!out || bytes < 0) return fail(ST_ERR_ARG, "bad arguments");
    ST_DEVICE(device);
    ST_HIP(hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16)));
    return ST_OK;
}

int st_device_free(int device, void *ptr)
{
    ST_DEVICE(device);
    ST_HIP(hipFree(ptr));
    return ST_OK;
}

int st_memcpy_h2d(int device, void *dst, const void *src, int64_t bytes)
{
    ST_DEVICE(device);
    ST_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return ST_OK;
}

int st_memcpy_d2h(int device, void *dst, const void *src, int64_t bytes)
{
    ST_DEVICE(device);
    ST_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return ST_OK;
}

int st_device_synchronize(int device)
{
    ST_DEVICE(device);
    ST_HIP(hipDeviceSynchronize());
    return ST_OK;
}

}  // extern "C"
