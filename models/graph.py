from egopack_amd.models.graph import Graph  # noqa: F401
