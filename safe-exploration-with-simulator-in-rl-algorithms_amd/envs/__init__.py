from .swimmer import SwimmerEnv, VecSwimmerEnv, Box, register_kwargs  # noqa: F401
