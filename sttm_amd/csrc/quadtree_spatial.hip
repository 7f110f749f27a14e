// Dispatcher of the fused quadtree spatial kernel (K1).  The kernel templates live in quadtree_spatial.inc and are
// instantiated by one translation unit per (dtype, per-head) combination (spatial_*.hip) so they compile in parallel.
#include "sttm_kernels.h"

namespace sttm {

template <typename T, bool PERHEAD>
hipError_t launch_spatial_t(const SpatialArgs& a, const BatchPtrs& bp, int n_videos, int vec, int nt, hipStream_t stream, void* tops);

hipError_t launch_spatial(const SpatialArgs& a, const BatchPtrs& bp, int n_videos, int dtype, int vec, int nt, hipStream_t stream, void* tops) {
    const bool ph = a.n_head > 0;
    if (dtype == STTM_F32) return ph ? launch_spatial_t<float, true>(a, bp, n_videos, vec, nt, stream, tops) : launch_spatial_t<float, false>(a, bp, n_videos, vec, nt, stream, tops);
    if (dtype == STTM_BF16) return ph ? launch_spatial_t<bf16_t, true>(a, bp, n_videos, vec, nt, stream, tops) : launch_spatial_t<bf16_t, false>(a, bp, n_videos, vec, nt, stream, tops);
    if (dtype == STTM_F16) return ph ? launch_spatial_t<f16_t, true>(a, bp, n_videos, vec, nt, stream, tops) : launch_spatial_t<f16_t, false>(a, bp, n_videos, vec, nt, stream, tops);
    return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_apply_t(const SpatialArgs& a, int vec, int nt, hipStream_t stream);

template <typename T>
hipError_t launch_spatial_pooled_t(const SpatialArgs& a, const BatchPtrs& bp, int n_videos, int nt, hipStream_t stream);

hipError_t launch_spatial_pooled(const SpatialArgs& a, const BatchPtrs& bp, int n_videos, int dtype, int nt, hipStream_t stream) {
    if (dtype == STTM_F32) return launch_spatial_pooled_t<float>(a, bp, n_videos, nt, stream);
    if (dtype == STTM_BF16) return launch_spatial_pooled_t<bf16_t>(a, bp, n_videos, nt, stream);
    if (dtype == STTM_F16) return launch_spatial_pooled_t<f16_t>(a, bp, n_videos, nt, stream);
    return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_spatial_col_t(const SpatialArgs& a, const BatchPtrs& bp, const ColWalkArgs& cw, int n_videos, int nt, hipStream_t stream);

hipError_t launch_spatial_col(const SpatialArgs& a, const BatchPtrs& bp, const ColWalkArgs& cw, int n_videos, int dtype, int nt, hipStream_t stream) {
    if (dtype == STTM_F32) return launch_spatial_col_t<float>(a, bp, cw, n_videos, nt, stream);
    if (dtype == STTM_BF16) return launch_spatial_col_t<bf16_t>(a, bp, cw, n_videos, nt, stream);
    if (dtype == STTM_F16) return launch_spatial_col_t<f16_t>(a, bp, cw, n_videos, nt, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_node_apply(const SpatialArgs& a, int dtype, int vec, int nt, hipStream_t stream) {
    if (dtype == STTM_F32) return launch_apply_t<float>(a, vec, nt, stream);
    if (dtype == STTM_BF16) return launch_apply_t<bf16_t>(a, vec, nt, stream);
    if (dtype == STTM_F16) return launch_apply_t<f16_t>(a, vec, nt, stream);
    return hipErrorInvalidValue;
}

}  // namespace sttm
