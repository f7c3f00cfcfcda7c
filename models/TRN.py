from egopack_amd.models.trn_multiscale import RelationModuleMultiScale  # noqa: F401
