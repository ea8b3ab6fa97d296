from .abstractions import IBayesianNetworkSampler
from .batched_sampler import BatchedBayesianNetworkSampler

__all__ = ["IBayesianNetworkSampler", "BatchedBayesianNetworkSampler"]
