from egopack_amd.models.temporal_pooling.trn_pooling import TRNPooling  # noqa: F401
