from egopack_amd.graphone import build_graphone  # noqa: F401
