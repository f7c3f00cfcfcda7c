from egopack_amd.models.tasks.oscc import OSCCTask  # noqa: F401
