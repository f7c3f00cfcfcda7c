"""Reference import path ``utils.meters`` -> egopack_amd.meters."""
from egopack_amd.meters import (AnticipationMeter, BaseMeter, LTAMeter, OSCCMeter, PNRMeter, RecognitionMeter,  # noqa: F401
                                build_meter_for_dataset)

Ego4dRecognitionMeter, Ego4dAnticipationMeter, Ego4dOSCCMeter = RecognitionMeter, AnticipationMeter, OSCCMeter
Ego4dLTAMeter, Ego4dPNRMeter = LTAMeter, PNRMeter
