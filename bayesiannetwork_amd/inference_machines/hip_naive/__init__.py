from .hip_naive_inference_machine import HipNaiveInferenceMachine

__all__ = ["HipNaiveInferenceMachine"]
