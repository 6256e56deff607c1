"""Mirror of the reference's `safe_ars` package for the hot path's batched one-step consumers:
`Basic_ARS` and `Safe_ARS` of safe_ars/ars.py on top of the HIP step kernel."""
from .ars import AbsObs, Basic_ARS, MaxAbsThetaDot, NativeCost, Safe_ARS  # noqa: F401
