from egopack_amd.data import build_dataloader, multiloader  # noqa: F401
