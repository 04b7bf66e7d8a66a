"""Env front-ends over the batched MI355X hot path.

`facade`      single-env gym-style classes carrying the names the reference registers / reserves
              (reference xroute_env/__init__.py:3-6, xroute_env/envs/*.py are empty stubs)
`vector_env`  the batched form: B regions per call, everything stays on the device
"""
from .facade import OrderingEvaluationEnv, OrderingTrainingEnv, StaticRegionEnv, XRouteEnv
from .vector_env import XRouteVectorEnv

__all__ = ["XRouteEnv", "OrderingTrainingEnv", "OrderingEvaluationEnv", "StaticRegionEnv", "XRouteVectorEnv"]
