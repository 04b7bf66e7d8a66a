from .core import XRouteEnv


class StaticRegionEnv(XRouteEnv):
    """One fixed region replayed forever (reference xroute_env/__init__.py:13-33 sketches
    `static-{benchmark}-v0` registrations with a `region` kwarg; the class itself is empty)."""

    def __init__(self, region, **kw):
        super().__init__([region], **kw)
