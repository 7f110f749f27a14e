"""sttm_amd -- MI355X-native STTM token merging (quadtree spatial + temporal merge, ToMe baseline).

Public surface mirrors the reference's `token_merging_utils` boundary:
    get_quadtree_features, get_tome_features, replace_qwen2_by_sparse_attn
"""
from .quadtree_interface import (get_quadtree_features, get_quadtree_features_batch, get_quadtree_features_from_pooled_input,  # noqa: F401
                                 get_quadtree_features_into)
from .tome_interface import get_tome_features, get_tome_features_batch  # noqa: F401

__all__ = ["get_quadtree_features", "get_quadtree_features_batch", "get_quadtree_features_from_pooled_input", "get_quadtree_features_into",
           "get_tome_features", "get_tome_features_batch"]
