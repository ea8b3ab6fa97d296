from .hip_spa_inference_machine import (
    HipSpaInferenceMachine,
    SpaInferenceMachineSettings,
    enable_profiling,
    profile_report,
)

__all__ = ["HipSpaInferenceMachine", "SpaInferenceMachineSettings", "enable_profiling", "profile_report"]
