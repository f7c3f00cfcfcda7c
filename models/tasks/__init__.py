from egopack_amd.models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask  # noqa: F401
