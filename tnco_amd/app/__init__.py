from .app import BaseContractionResults, BaseOptimizer, Optimizer, dump_results
from .tn import Tensor, TensorNetwork, load_tn

__all__ = ["Optimizer", "BaseOptimizer", "BaseContractionResults", "dump_results", "Tensor",
           "TensorNetwork", "load_tn"]
