"""The O(n) formulation of the swimmer's accelerations that BASELINE.json's north_star names ("tridiagonal
mass-matrix accelerations", "longer tridiagonal solve") -- derived, checked against the reference's own outputs
(tests/golden/steps.npz: `compute_accelerations` of remy_swimmer_env.py for n = 2 ... 8, three parameter sets),
and COUNTED, before any kernel is written.  Design aid; nothing in the product imports it.

Derivation (notation of remy_swimmer_env.py:116-207; t_i = (cos th_i, sin th_i), n_i = (-sin th_i, cos th_i)):

  joint points      A_i = A_{i-1} + l t_i                                  (:128-152)
  centre of seg. i  a_i = (Add_{i-1} + Add_i) / 2,   Add_i - Add_{i-1} = l (thdd_i n_i - thd_i^2 t_i)
  force balance     m a_i = f_i - f_{i-1} + F_i n_i,   F_i = -k l (Gdot_i . n_i),   f_0 = f_n = 0   (:173-184, rows 0-1)
  torque balance    (m l^2 / 12) thdd_i = (l / 2) n_i . (f_i + f_{i-1}) + tau_i,
                    tau_i = k thd_i l^3 / 12 + u_{i-1} - u_i               (:191-205; c fY - s fX = n . f)

Subtracting the force balances of neighbours i, i + 1 and eliminating a_{i+1} - a_i (kinematics) and thdd (torque
balance), with N_i = n_i n_i^T:

  (I - 3 N_i) f_{i-1}  -  (2 I + 3 N_i + 3 N_{i+1}) f_i  +  (I - 3 N_{i+1}) f_{i+1}
        =  q_i - q_{i+1} + p_i + p_{i+1},      q_i = F_i n_i,   p_i = (6 tau_i / l) n_i - (m l / 2) thd_i^2 t_i

for the n - 1 internal joints i = 1 ... n - 1: a symmetric BLOCK-TRIDIAGONAL system with 2 x 2 blocks (negative
definite), solved by block Thomas in O(n); then thdd_i from the torque balance and Gdd = sum_i q_i / (n m) (the
force balances summed, as in csrc/swimmer_device.h).  Centre velocities come from the same O(n) recurrence along the chain.

  python scripts/chain_formulation.py        # parity table + operation counts
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def accelerations(n, l, m, k, state, u):
    """(Gdd[2], thdd[n]) of one swimmer by the block-tridiagonal formulation.  state = [Gdx, Gdy, th_1, thd_1, ...]."""
    gd = np.asarray(state[:2], dtype=np.float64)
    th = np.asarray(state[2::2], dtype=np.float64)
    thd = np.asarray(state[3::2], dtype=np.float64)
    c, s = np.cos(th), np.sin(th)
    t = np.stack([c, s], axis=1)
    nn = np.stack([-s, c], axis=1)
    # point velocities in the head frame, then the shift that makes the barycentre move with Gdot (:128-163)
    A = np.zeros((n + 1, 2))
    for i in range(n):
        A[i + 1] = A[i] + l * thd[i] * nn[i]
    centre = 0.5 * (A[:-1] + A[1:])
    centre += gd - centre.mean(axis=0)
    F = -k * l * np.einsum("ij,ij->i", centre, nn)                       # :180
    q = F[:, None] * nn
    uu = np.concatenate([[0.0], np.asarray(u, dtype=np.float64), [0.0]])  # u_0 = u_n = 0: free ends
    tau = k * thd * l ** 3 / 12.0 + uu[:-1] - uu[1:]                      # :201-205 (signs as coded)
    p = (6.0 * tau / l)[:, None] * nn - (0.5 * m * l) * (thd ** 2)[:, None] * t
    N = nn[:, :, None] * nn[:, None, :]
    B = np.eye(2) - 3.0 * N                                               # B_i couples f_{i-1} and f_i
    J = n - 1                                                             # internal joints 1 .. n-1
    f = np.zeros((n + 1, 2))
    if J > 0:
        # negated system:  -B_i f_{i-1} + D_i f_i - B_{i+1} f_{i+1} = -(rhs_i),  D_i = 2 I + 3 N_i + 3 N_{i+1}
        D = [2.0 * np.eye(2) + 3.0 * (N[i - 1] + N[i]) for i in range(1, n)]          # joint i sits between segs i, i+1
        r = [-(q[i - 1] - q[i] + p[i - 1] + p[i]) for i in range(1, n)]
        S = [None] * J
        y = [None] * J
        S[0], y[0] = D[0], r[0]
        for j in range(1, J):                                             # forward sweep (block Thomas)
            Bj = B[j]                                                     # segment j + 1 (0-based j) joins joints j, j+1
            P = Bj @ np.linalg.inv(S[j - 1])
            S[j] = D[j] - P @ Bj
            y[j] = r[j] + P @ y[j - 1]
        f[J] = np.linalg.solve(S[J - 1], y[J - 1])
        for j in range(J - 2, -1, -1):                                    # back substitution
            f[j + 1] = np.linalg.solve(S[j], y[j] + B[j + 1] @ f[j + 2])
    thdd = (12.0 / (m * l * l)) * (0.5 * l * np.einsum("ij,ij->i", nn, f[1:] + f[:-1]) + tau)
    gdd = q.sum(axis=0) / (n * m)
    return gdd, thdd


def operation_counts(n):
    """fp64 instructions per env-step of a one-rollout-per-lane kernel built on this formulation (FMA = 1;
    a reciprocal with one Newton step = 6), counted from the statements above; sin / cos and the records as in
    today's lane kernel (rollout_kernel<N>)."""
    J = n - 1
    c = {
        "sin/cos (Cody-Waite + two minimax polynomials per angle, as today)": 32 * n,
        "policy (m x d multiply-adds + bias)": (n - 1) * (2 * n + 2) + (n - 1),
        "point / centre velocities, barycentre shift": 2 * n + 2 * n + 2 * n + 4,
        "friction forces F_i, q_i = F_i n_i": 3 * n + 2 * n,
        "tau_i, p_i": 3 * n + 5 * n,
        "blocks N_i (3 distinct), B_i, D_j": 3 * n + 3 * n + 6 * J,
        "right-hand sides": 6 * J,
        "forward sweep (2x2 inverse 12, P = B S^-1 8, P B 6, S 3, y 4)": 33 * max(J - 1, 0) + 12,
        "back substitution (B f 4, S^-1 . 4)": 8 * J,
        "thdd_i from the torque balance": 5 * n,
        "Gdd": 2 * n + 2,
        "Euler update, reward, range tracking": 2 * n + 2 + 2 + n,
        "trajectory stores + V2 moment sums": (2 * n + 2) + 2 * (2 * n + 2),
    }
    return c


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "steps.npz"))
    psets = {"default": (1.0, 1.0, 10.0), "realworld": (0.8, 1.2, 10.2), "odd": (1.3, 0.7, 4.5)}
    print("block-tridiagonal formulation vs the reference's compute_accelerations (tests/golden/steps.npz)")
    for n in (2, 3, 4, 5, 6, 8):
        worst = 0.0
        for ps, (l, m, k) in psets.items():
            key = f"n{n}_{ps}"
            for st, u, gdd, tdd in zip(g[key + "_state"], g[key + "_action"], g[key + "_gdd"], g[key + "_tdd"]):
                a, b = accelerations(n, l, m, k, st, u)
                scale = max(1.0, np.abs(tdd).max())
                worst = max(worst, np.abs(a - gdd).max(), np.abs(b - tdd).max() / scale)
        print(f"  n = {n}: max relative difference {worst:.2e} over 144 states")
    measured = {3: 314, 6: 1054}        # today's lane kernel, instructions per step
    for n in (3, 6, 8):
        c = operation_counts(n)
        total = sum(c.values())
        print(f"\nn = {n}: {total} fp64 instructions per env-step (+ ~8 % moves / compares / loop)"
              + (f"   -- today's dense n x n lane kernel: {measured[n]} measured" if n in measured else ""))
        for what, v in c.items():
            print(f"    {v:5d}  {what}")


if __name__ == "__main__":
    sys.exit(main())
