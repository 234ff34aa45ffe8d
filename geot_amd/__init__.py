"""geot_amd -- MI355X (gfx950) implementation of GeoT's point-cloud sampling / grouping /
interpolation hot path behind the reference's operator API.

    from geot_amd.pointnet2 import pointnet2_utils          # furthest_point_sample, ball_query, ...
    from geot_amd.pointops.functions import pointops        # fps, knn, ...
    from geot_amd.knn_cuda import KNN
    import geot_amd.aliases; geot_amd.aliases.install()     # reference import names -> this package

The compute path is libgeot_hip.so (C ABI in include/geot_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
