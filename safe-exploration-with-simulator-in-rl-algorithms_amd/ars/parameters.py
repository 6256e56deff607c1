"""Configuration surface of the path: the reference's dataclasses, field for field
(ars/parameters.py:10-45), so scripts that build EnvParam / ARSParam run unchanged."""
from dataclasses import dataclass


@dataclass
class EnvParam:
    name: str
    n: int          # number of segments
    H: int          # length of rollout
    l_i: float      # length of a segment
    m_i: float      # mass of a segment
    h: float        # time interval for integration
    k: float        # viscosity coefficient
    epsilon: float  # approximation error (used by the safe-exploration threshold only)


@dataclass
class ARSParam:
    name: str
    V1: bool         # True: ARS V1 (no state whitening)
    n_iter: int      # training iterations
    H: int           # rollout length
    N: int           # directions sampled per iteration
    b: int           # divisor of the update step (the reference never truncates to top-b)
    alpha: float     # step size
    nu: float        # exploration noise scale
    safe: bool       # safe exploration gate (sequential by construction: not on this path)
    threshold: float
    initial_w: str   # 'Zero' or a path to a .npy policy


@dataclass
class Threshold:
    K: float  # Lipschitz constant of the reward function
    A: float  # Lipschitz constant of the transition function w.r.t. parameters
    B: float  # Lipschitz constant of the transition function w.r.t. states

    def compute_alpha(self, H):
        return self.K * self.A / (1 - self.B) * (H - self.B * (1 - self.B ** H) / (1 - self.B))
