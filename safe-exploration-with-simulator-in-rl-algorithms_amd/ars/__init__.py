from .parameters import EnvParam, ARSParam, Threshold  # noqa: F401
from .environment import Environment  # noqa: F401
from .ars_agent import ARSAgent  # noqa: F401
