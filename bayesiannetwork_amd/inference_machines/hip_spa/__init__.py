from .hip_spa_inference_machine import HipSpaInferenceMachine, SpaInferenceMachineSettings

__all__ = ["HipSpaInferenceMachine", "SpaInferenceMachineSettings"]
