"""`mmdet3d.ops` names used by the point-cloud ReID hot path (reference: mmdet3d/ops/__init__.py).
The reference's detection-only ops (spconv, voxelization, bev_pool, iou3d, roiaware_pool3d, paconv,
sync-BN) and its re-exports of mmcv.ops are intentionally absent."""
from .point_ops import (FurthestPointSampling, FurthestPointSamplingWithDist, BallQuery, KNN, GatherPoints,
                        GroupingOperation, ThreeNN, ThreeInterpolate, furthest_point_sample,
                        furthest_point_sample_with_dist, ball_query, ball_query_cnt, knn, gather_points, grouping_operation,
                        three_nn, three_interpolate)

from .pointnet_modules import (SA_MODULES, GroupAll, PointFPModule, PointSAModule, PointSAModuleMSG, Points_Sampler,
                               QueryAndGroup, build_sa_module, calc_square_dist)

__all__ = ["SA_MODULES", "GroupAll", "PointFPModule", "PointSAModule", "PointSAModuleMSG", "Points_Sampler",
           "QueryAndGroup", "build_sa_module", "furthest_point_sample", "furthest_point_sample_with_dist", "ball_query", "knn", "gather_points",
           "grouping_operation", "three_nn", "three_interpolate"]
