"""Configuration records of the swimmer / ARS path.

Drop-in for the reference's `ars/parameters.py:10-45`: the three records keep the reference's
names, field names and field ORDER (its scripts construct them positionally and by keyword,
e.g. `ars/plot_graph.py:14-20`), so experiment scripts run unchanged.  Unlike the reference's
bare dataclasses they validate what the GPU path relies on.

    EnvParam(name, n, H, l_i, m_i, h, k, epsilon)
        name     label only
        n        number of swimmer segments (kernels exist for 2..8)
        H        rollout length in physics steps
        l_i,m_i  length and mass of one segment
        h        explicit-Euler time step
        k        viscous friction coefficient
        epsilon  simulator approximation error (safe-exploration threshold only)

    ARSParam(name, V1, n_iter, H, N, b, alpha, nu, safe, threshold, initial_w)
        V1        True: plain linear policy; False: V2 state whitening
        n_iter    training iterations after the warm-up one
        H         rollout length
        N         exploration directions per iteration
        b         divisor of the update step (the reference never truncates to the top b)
        alpha,nu  step size, exploration noise scale
        safe, threshold   safe-exploration gate (sequential by construction: not on this path)
        initial_w 'Zero' or the path of a .npy policy
"""
from dataclasses import dataclass


def _require(cond, message):
    if not cond:
        raise ValueError(message)


@dataclass
class EnvParam:
    name: str
    n: int
    H: int
    l_i: float
    m_i: float
    h: float
    k: float
    epsilon: float

    def __post_init__(self):
        _require(int(self.n) == self.n and self.n >= 1, f"EnvParam.n must be a positive integer, got {self.n!r}")
        _require(int(self.H) == self.H and self.H >= 0, f"EnvParam.H must be a non-negative integer, got {self.H!r}")


@dataclass
class ARSParam:
    name: str
    V1: bool
    n_iter: int
    H: int
    N: int
    b: int
    alpha: float
    nu: float
    safe: bool
    threshold: float
    initial_w: str

    def __post_init__(self):
        _require(int(self.N) == self.N and self.N >= 1, f"ARSParam.N must be a positive integer, got {self.N!r}")
        _require(self.b != 0, "ARSParam.b divides the update step and cannot be 0")


@dataclass
class Threshold:
    """Lipschitz constants of the safe-exploration bound: reward (K), transition w.r.t.
    parameters (A) and w.r.t. states (B)."""
    K: float
    A: float
    B: float

    def compute_alpha(self, H):
        """K A / (1 - B) * (H - B (1 - B^H) / (1 - B))  (ars/parameters.py:44-45)."""
        geometric = self.B * (1.0 - self.B ** H) / (1.0 - self.B)
        return self.K * self.A / (1.0 - self.B) * (H - geometric)
