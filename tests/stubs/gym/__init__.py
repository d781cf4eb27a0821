from . import spaces          # noqa: F401
