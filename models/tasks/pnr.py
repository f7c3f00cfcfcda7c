from egopack_amd.models.tasks.pnr import PNRTask  # noqa: F401
