from egopack_amd.models.graphONE.graphONE import GraphONE, cos_dissimilarity  # noqa: F401
