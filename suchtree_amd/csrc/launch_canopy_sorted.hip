// launch_canopy_sorted.hip -- translation unit of k_canopy_sorted (the tile-sorted ladder kernel of deep canopies with chains
// of at most seven slots; the 15- / 31-slot and pointer forms won no cell of profiles/kernel_win_matrix_r06.json and are gone):
// every record size x tile size x mode x pair source it is launched with is compiled here, in parallel
// with the other units of libsuchtree_hip.so.  Built for gfx950 only: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#define ST_LAUNCH_UNIT 1
#include "device_common.h"
#include "pair_math.h"
#include "tree_prep.h"
#include "st_tree.h"
#include "launch_decl.h"
#include "launch_policy.h"
#include "kernels_canopy.h"
#include "kernels_canopy_sorted.h"
#define ST_SORTED_UNIT 1
#include "launch_canopy_sorted.h"

namespace st {

template <int CAP, typename Src>
hipError_t launch_canopy_sorted(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n,
                                       DistSink out_d, MrcaSink out_m, Fault *fault, hipStream_t stream)
{
    const SortedShape shape = sorted_shape(t);
    const int q = batch_tile_q(shape.q, shape.sums ? 1 : 2, t->sort_tile, n, t->n_cu);      // (instantiated: 1, 2, 4 with lineage sums, else 2, 4)
    const size_t lds = ladder_image_bytes(t->canopy_nodes) + sort_scratch_bytes(q, shape.rmq, shape.sums);
    CanopyParams Pk = P;
    if (!shape.rmq) { Pk.cpos = nullptr; Pk.rmq = nullptr; }
    if (!shape.sums) { Pk.rec_p = nullptr; Pk.lineage = nullptr; }
    const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int64_t tile = (int64_t)q * kCanopyBlock;
    int64_t blocks = (n + tile - 1) / tile;
    blocks = std::max<int64_t>(std::min<int64_t>(blocks, (int64_t)t->n_cu * wg_per_cu), 1);
    int key_shift = 0;     // keys are edge counts: of both canopy lineages, or (lineage sums) of b's whole lineage
    const int key_max = shape.sums ? t->canopy_depth + t->rec_cap : 2 * t->canopy_depth;
    while ((key_max >> key_shift) >= kSortBuckets) key_shift++;
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kCanopyBlock), lds, stream, Pk, src,
                           (long long)n, out_d, out_m, fault, key_shift);
        return hipGetLastError();
    };
    if (shape.sums)
        return q == 1 ? go(k_canopy_sorted<CAP, 1, true, Src>) : q == 2 ? go(k_canopy_sorted<CAP, 2, true, Src>)
                                                                         : go(k_canopy_sorted<CAP, 4, true, Src>);
    return q == 2 ? go(k_canopy_sorted<CAP, 2, false, Src>) : go(k_canopy_sorted<CAP, 4, false, Src>);
}


#define ST_INSTANTIATE_SORTED(S)                                                                                                       \
    template hipError_t launch_canopy_sorted<1, S>(const st_tree *, const CanopyParams &, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t);  \
    template hipError_t launch_canopy_sorted<3, S>(const st_tree *, const CanopyParams &, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t);  \
    template hipError_t launch_canopy_sorted<7, S>(const st_tree *, const CanopyParams &, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t);
ST_FOR_EACH_SRC(ST_INSTANTIATE_SORTED)

}  // namespace st
