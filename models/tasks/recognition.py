from egopack_amd.models.tasks.recognition import RecognitionTask  # noqa: F401
