from . import pointops  # noqa: F401
