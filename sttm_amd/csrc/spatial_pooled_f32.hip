// Explicit instantiation of the pooled-input spatial kernel (spatial_pooled.inc) for one dtype.
#include "spatial_pooled.inc"

namespace sttm {
template hipError_t launch_spatial_pooled_t<float>(const SpatialArgs&, const BatchPtrs&, int, int, hipStream_t);
}  // namespace sttm
