from .abstractions import IInferenceMachine
from .common import InferenceMachineSettings
from .evidence import Evidence, EvidenceLoader

__all__ = ["IInferenceMachine", "InferenceMachineSettings", "Evidence", "EvidenceLoader"]
