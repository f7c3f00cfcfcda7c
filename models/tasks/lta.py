from egopack_amd.models.tasks.lta import LTATask  # noqa: F401
