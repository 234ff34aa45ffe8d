"""Mirror of the hot-path parts of the reference's ``openpoints`` package: the op wrappers in
``openpoints.models.layers`` and ``openpoints.cpp`` plus the sampling/grouping callers of the
configured backbone.  Everything else in openpoints (registries, datasets, optimisers, ...) is
out of scope."""
