"""MI355X-native implementation of the reference's swimmer-physics + ARS-rollout hot path.

Public surface (mirrors the reference's module layout for this path):
    envs.SwimmerEnv / envs.VecSwimmerEnv      envs/gym_swimmer/swimmer/remy_swimmer_env.py
    ars.Environment                           ars/environment.py
    ars.ARSAgent                              ars/ars_agent.py
    ars.EnvParam / ars.ARSParam               ars/parameters.py
    safe_ars.Basic_ARS / safe_ars.Safe_ARS    safe_ars/ars.py  (batched one-step consumers of the step kernel)
    kernels.*                                 thin wrappers of the C ABI (include/swimmer_hip.h)
"""
from . import _build, _lib, kernels  # noqa: F401
from ._lib import SwParams, SwimmerHipError  # noqa: F401
from .envs import SwimmerEnv, VecSwimmerEnv  # noqa: F401
from .ars import ARSAgent, ARSParam, EnvParam, Environment  # noqa: F401
from . import safe_ars  # noqa: F401

__all__ = ["SwParams", "SwimmerHipError", "SwimmerEnv", "VecSwimmerEnv", "ARSAgent", "ARSParam",
           "EnvParam", "Environment", "kernels"]
