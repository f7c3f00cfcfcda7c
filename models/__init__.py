"""Top-level ``models`` package: the import paths the reference's Hydra configs and entry points use
(``_target_: models.graph.Graph`` ...) resolved to the MI355X implementation in egopack_amd.models."""
from egopack_amd.models.graph import Graph  # noqa: F401
