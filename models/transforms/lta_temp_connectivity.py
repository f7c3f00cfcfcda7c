from egopack_amd.data import LTATemporalConnectivity  # noqa: F401
