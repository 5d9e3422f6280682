// launch_canopy_sorted.h -- launch_canopy_sorted<CAP, Src>: defined and instantiated in launch_canopy_sorted.hip,
// called by launch_canopy.hip.
#pragma once

namespace st {

template <int CAP, typename Src>
hipError_t launch_canopy_sorted(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n, DistSink out_d,
                                MrcaSink out_m, Fault *fault, hipStream_t stream);

#ifndef ST_SORTED_UNIT
#define ST_EXTERN_SORTED(S)                                                                                                                   \
    extern template hipError_t launch_canopy_sorted<1, S>(const st_tree *, const CanopyParams &, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t);  \
    extern template hipError_t launch_canopy_sorted<3, S>(const st_tree *, const CanopyParams &, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t);  \
    extern template hipError_t launch_canopy_sorted<7, S>(const st_tree *, const CanopyParams &, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t);
ST_FOR_EACH_SRC(ST_EXTERN_SORTED)
#undef ST_EXTERN_SORTED
#endif

}  // namespace st
