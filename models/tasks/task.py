from egopack_amd.models.tasks.task import ProjectionTask, TaskLiteral  # noqa: F401
