// Explicit instantiation of the spatial kernels for one (dtype, per-head) combination.
#include "quadtree_spatial.inc"

namespace sttm {
template hipError_t launch_spatial_t<f16_t, false>(const SpatialArgs&, const BatchPtrs&, int, int, int, hipStream_t, void*);
template hipError_t launch_apply_t<f16_t>(const SpatialArgs&, int, int, hipStream_t);
}  // namespace sttm
