from egopack_amd.models.temporal_pooling.pooling import TemporalPooling  # noqa: F401
