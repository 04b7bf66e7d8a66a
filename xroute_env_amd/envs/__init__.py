from .core import XRouteEnv
from .ordering_training_env import OrderingTrainingEnv
from .ordering_evaluation_env import OrderingEvaluationEnv
from .static_region_env import StaticRegionEnv
from .vector_env import XRouteVectorEnv

__all__ = ["XRouteEnv", "OrderingTrainingEnv", "OrderingEvaluationEnv", "StaticRegionEnv", "XRouteVectorEnv"]
