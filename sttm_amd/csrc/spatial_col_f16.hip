// Explicit instantiation of the column-walk spatial kernel (spatial_col.inc) for one dtype.
#include "spatial_col.inc"

namespace sttm {
template hipError_t launch_spatial_col_t<f16_t>(const SpatialArgs&, const BatchPtrs&, const ColWalkArgs&, int, int, hipStream_t);
}  // namespace sttm
