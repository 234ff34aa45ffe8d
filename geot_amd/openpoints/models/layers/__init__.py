from .subsample import furthest_point_sample, gather_operation as gather_points, fps, random_sample  # noqa: F401
from .group import (grouping_operation, gather_operation, torch_grouping_operation, ball_query, QueryAndGroup,  # noqa: F401
                    GroupAll, KNNGroup, create_grouper, get_aggregation_feautres)
from .upsampling import three_nn, three_interpolate, three_interpolation  # noqa: F401
from .knn import knn_point, KNN, DenseDilated, DilatedKNN  # noqa: F401
