from .graph import Graph  # noqa: F401
