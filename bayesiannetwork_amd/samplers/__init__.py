from .abstractions import IBayesianNetworkSampler
from .batched_sampler import BatchedBayesianNetworkSampler
from .hip_sampler import HipBayesianNetworkSampler

__all__ = ["IBayesianNetworkSampler", "BatchedBayesianNetworkSampler", "HipBayesianNetworkSampler"]
