from .core import XRouteEnv


class OrderingTrainingEnv(XRouteEnv):
    """`xroute_env/ordering-training-v0` (reference xroute_env/__init__.py:3-6): net-ordering training
    on rotating regions — Game semantics, 10 replays per region then the next."""
