"""heterofusionrcnn_amd -- the HeteroFusionRCNN point-cloud hot path on MI355X (gfx950).

Hand-written HIP kernels behind a C ABI (include/hfops.h, csrc/libhfops.so), exposed under the
reference's Python op names:

    sampling     farthest_point_sample, gather_point                (sampling/tf_sampling.py)
    grouping     query_ball_point, group_point, select_top_k, knn_point (grouping/tf_grouping.py)
    interpolate  three_nn, three_interpolate                        (interpolate/tf_interpolate.py)
    bev_iou      compute_bev_iou (= bev_iou), oriented_nms          (bev_iou/bev_iou.py)
    cropping     pc_crop_and_sample (= crop_and_resize)             (cropping/tf_cropping.py)

and, either side of the ops: modules (SA / FP modules, box utilities), mlp (fp32-MFMA shared MLP),
fusion (LiDAR -> image projection + feature gather), box_codec (bin-based box encode / decode), two_stage
(RPN -> crop -> RCNN inference flow), pipeline (geometry prefetch), dp (data parallel), kitti_io (host-side formats).

Importing this package loads libhfops.so; a missing library is an ImportError, never a fallback.
"""
from . import _lib

_lib.lib()  # fail loudly at import if the native library is absent

from .sampling import farthest_point_sample, gather_point, prob_sample  # noqa: E402
from .grouping import (group_point, knn_point, query_ball_group, query_ball_point,  # noqa: E402
                       select_top_k)
from .interpolate import three_interpolate, three_nn  # noqa: E402
from .fusion import project_gather, rect_to_image  # noqa: E402
from .bev_iou import bev_iou, compute_bev_iou, nms_mask, oriented_nms, oriented_nms_batched  # noqa: E402
from .cropping import crop_and_resize, pc_crop_and_sample  # noqa: E402

__all__ = [
    "farthest_point_sample", "gather_point", "prob_sample",
    "query_ball_point", "group_point", "query_ball_group", "select_top_k", "knn_point",
    "three_nn", "three_interpolate", "project_gather", "rect_to_image",
    "compute_bev_iou", "bev_iou", "oriented_nms", "oriented_nms_batched", "nms_mask",
    "pc_crop_and_sample", "crop_and_resize",
]
