"""Mirror of the reference's top-level ``pointops`` package."""
