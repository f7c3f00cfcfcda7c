from .recognition import RecognitionTask  # noqa: F401
from .oscc import OSCCTask  # noqa: F401
from .lta import LTATask  # noqa: F401
from .pnr import PNRTask  # noqa: F401
