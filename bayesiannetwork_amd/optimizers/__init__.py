from .common import BatchEvaluator, Evaluator, OptimizerLogger
from .em_batch_optimizer import EmBatchOptimizer, EmBatchOptimizerSettings
from .em_optimizer import EmOptimizer, EmOptimizerSettings

__all__ = [
    "BatchEvaluator", "Evaluator", "OptimizerLogger",
    "EmBatchOptimizer", "EmBatchOptimizerSettings", "EmOptimizer", "EmOptimizerSettings",
]
