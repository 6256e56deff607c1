// swimmer_quad3.h -- latency-optimised rollout kernel for the 3-segment swimmer:
// ONE SEGMENT PER LANE, four lanes (a DPP quad) per rollout, 16 rollouts per wave.
//
// Why: the BASELINE configs run 1024-4096 rollouts per GPU, far fewer than the chip's
// 1024 SIMDs x 64 lanes.  With one rollout per lane (rollout_kernel) a batch of 1024
// rollouts occupies 16 SIMDs and its speed is the length of one lane's instruction stream
// (~310 instructions per step, one instruction per ~4.4 cycles for a lone wave,
// scripts/ubench).  Spreading a rollout over the lanes of a quad cuts that stream to 125
// instructions per step (round 1: 145) at 4x the (idle anyway) SIMD count.  Since round 3 this
// kernel serves batches of 8193 .. 16384 rollouts; smaller ones run on the mirror-quad kernel
// (swimmer_oct3.h: two quads per rollout with sine / cosine roles, 113 instructions per step):
//
//   lane q = 0,1,2 of a quad owns segment q: its angle, angular velocity, sin/cos, its row
//   of the 3x3 joint-acceleration system and its V2 moment sums.  The joint torques are never
//   formed: lane i only needs u_{i-1} - u_i, which is linear in the observation, so it holds
//   the pre-combined policy row V_i = 12/(m l^2) (W_{i-1} - W_i) (columns in its rotated
//   order) and evaluates one dot product on the observation (the mean enters as one constant
//   per rollout; the angle part is carried from step to step instead of re-evaluated, so the
//   neighbours' angles are never exchanged); lane 3 mirrors lane 0 bit for bit (same inputs,
//   same permutation sources), so whatever it stores duplicates lane 0's stores.
//   Angles are carried in reduced form theta = r + K pi/2 (swimmer_device.h, Angle): sin / cos
//   need no per-step range reduction and no quadrant logic.
//   Neighbour data moves with DPP quad_perm moves (no LDS, no memory, two 32-bit moves per
//   double): next1 = segment (q+1)%3, next2 = segment (q+2)%3.
//   Every lane solves the SAME symmetric 3x3 system in its own rotated order
//   (own, next1, next2) and keeps only its own component (closed-form cofactor row, one
//   reciprocal), so no lane-dependent code is needed: all lane dependence sits in a few
//   per-lane constants (chain weights for its rotation) loaded in the prologue.
//   Gdot is replicated per lane (each lane integrates its own copy, equal up to rounding);
//   the state of a rollout is: Gdot_x from lane 0, Gdot_y from lane 1, (theta_i, thetadot_i)
//   from lane i.
//
//   The angle-only part of step t+1 (sin/cos, their exchange, pairwise cos/sin of differences)
//   is evaluated beside step t's solve: theta_{t+1} needs thetadot_t only (Quad3Geo).
//
// Same equations as swimmer_device.h (see the derivation there); the per-step arithmetic
// differs from rollout_kernel in summation order and in the reciprocal's last Newton step
// (2e-15 relative on thetaddot, see rcp_f64_1n).
#pragma once

#include "swimmer_device.h"

namespace sw {

// quad_perm control words: lane j of a quad reads lane perm[j]
constexpr int kDppNext1 = 1 | (2 << 2) | (0 << 4) | (1 << 6);  // [1,2,0,1]
constexpr int kDppNext2 = 2 | (0 << 2) | (1 << 4) | (2 << 6);  // [2,0,1,2]

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane is active and every source lane exists, so the "old" value is never used:
    // mov_dpp (undefined old) saves the copy update_dpp's tied operand would need
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Per-lane constants for segment i of the 3-segment chain in the rotated order
// (i, i1 = i+1 mod 3, i2 = i+2 mod 3).
struct Quad3Lane {
    double vw0, vw1, vw2;   // vel_w(i,i), vel_w(i,i1), vel_w(i,i2)
    double a0, a1, a2;      // Aw(i,i), Aw(i,i1), Aw(i,i2)
    double t1, t2, t12;     // -6 T(i,i1), -6 T(i,i2), -6 T(i1,i2)
    double d0, d12, d1, d2; // diagonal of Q in rotated order, d12 = d1*d2
};

__device__ __forceinline__ Quad3Lane quad3_lane(int seg)
{
    Quad3Lane L;
    auto pick = [&](double v0, double v1, double v2) { return seg == 0 ? v0 : (seg == 1 ? v1 : v2); };
    // 1-based indices in the weight functions
    L.vw0 = pick(vel_w<3>(1, 1), vel_w<3>(2, 2), vel_w<3>(3, 3));
    L.vw1 = pick(vel_w<3>(1, 2), vel_w<3>(2, 3), vel_w<3>(3, 1));
    L.vw2 = pick(vel_w<3>(1, 3), vel_w<3>(2, 1), vel_w<3>(3, 2));
    L.a0 = pick(Aw<3>(1, 1), Aw<3>(2, 2), Aw<3>(3, 3));
    L.a1 = pick(Aw<3>(1, 2), Aw<3>(2, 3), Aw<3>(3, 1));
    L.a2 = pick(Aw<3>(1, 3), Aw<3>(2, 1), Aw<3>(3, 2));
    L.t1 = pick(-6.0 * Tw<3>(1, 2), -6.0 * Tw<3>(2, 3), -6.0 * Tw<3>(3, 1));
    L.t2 = pick(-6.0 * Tw<3>(1, 3), -6.0 * Tw<3>(2, 1), -6.0 * Tw<3>(3, 2));
    L.t12 = pick(-6.0 * Tw<3>(2, 3), -6.0 * Tw<3>(3, 1), -6.0 * Tw<3>(1, 2));
    constexpr double D1 = -6.0 * Tw<3>(1, 1) + 1.0, D2 = -6.0 * Tw<3>(2, 2) + 1.0,
                     D3 = -6.0 * Tw<3>(3, 3) + 1.0;
    L.d0 = pick(D1, D2, D3);
    L.d1 = pick(D2, D3, D1);
    L.d2 = pick(D3, D1, D2);
    L.d12 = L.d1 * L.d2;
    return L;
}

// Everything of a step that depends on the angles only: own and neighbours' sin / cos and the
// pairwise cos(th_i - th_k), sin(th_k - th_i).  theta_{t+1} = theta_t + h thetadot_t is known
// at the START of step t, so the caller evaluates the geometry of step t+1 while step t's
// solve is still in flight: the sincos chain (the longest dependent chain of the step) leaves
// the critical path thetadot_t -> thetadot_{t+1}.
struct Quad3Geo {
    double s, c, s1, c1, s2, c2;   // own, next1, next2
    double cc1, cc2, cc12;         // cos(th_i - th_i1), cos(th_i - th_i2), cos(th_i1 - th_i2)
    double ss1, ss2;               // sin(th_i1 - th_i), sin(th_i2 - th_i)
};

__device__ __forceinline__ Quad3Geo quad3_geometry(const Angle &A, const TrigK &K)
{
    Quad3Geo G;
    sincos_angle(A, G.s, G.c, K);
    G.s1 = dpp_f64<kDppNext1>(G.s);
    G.c1 = dpp_f64<kDppNext1>(G.c);
    G.s2 = dpp_f64<kDppNext2>(G.s);
    G.c2 = dpp_f64<kDppNext2>(G.c);
    G.cc1 = __builtin_fma(G.c, G.c1, G.s * G.s1);
    G.cc2 = __builtin_fma(G.c, G.c2, G.s * G.s2);
    G.cc12 = __builtin_fma(G.c1, G.c2, G.s1 * G.s2);
    G.ss1 = __builtin_fma(G.c, G.s1, -G.s * G.c1);
    G.ss2 = __builtin_fma(G.c, G.s2, -G.s * G.c2);
    return G;
}

// The velocity-dependent part of one explicit-Euler step of one rollout spread over a quad:
// updates gdx, gdy (replicated) and thd (own segment); theta is advanced by the caller.
// w1, w2: the neighbours' angular velocities (the caller exchanged them for the policy
// already); tq_scaled = 12/(m l^2) (u_{i-1} - u_i), this segment's joint-torque balance.
// Returns det of the (rotated) system for the singularity check.
__device__ __forceinline__ double quad3_dynamics(const Consts &C, const Quad3Lane &L,
                                                 const Quad3Geo &G, double &gdx, double &gdy,
                                                 double &thd, double w1, double w2,
                                                 double tq_scaled)
{
    // normal velocity of this segment's centre
    double g = __builtin_fma(gdy, G.c, -gdx * G.s);
    g = __builtin_fma(L.vw0 * C.l, thd, g);
    g = __builtin_fma((L.vw1 * C.l) * G.cc1, w1, g);
    g = __builtin_fma((L.vw2 * C.l) * G.cc2, w2, g);
    const double g1 = dpp_f64<kDppNext1>(g), g2 = dpp_f64<kDppNext2>(g);
    // barycentre acceleration (rotated summation order; re-synchronised below)
    const double sx = __builtin_fma(g2, G.s2, __builtin_fma(g1, G.s1, g * G.s));
    const double sy = __builtin_fma(g2, G.c2, __builtin_fma(g1, G.c1, g * G.c));
    // this segment's row of Q thdd = r (the accumulation starts from the torque balance)
    double cent = __builtin_fma(L.t1 * (w1 * w1), G.ss1, tq_scaled);
    cent = __builtin_fma(L.t2 * (w2 * w2), G.ss2, cent);
    double fric = L.a0 * g;
    fric = __builtin_fma(L.a1 * G.cc1, g1, fric);
    fric = __builtin_fma(L.a2 * G.cc2, g2, fric);
    double r0 = __builtin_fma(-C.six_k_m, fric, cent);
    r0 = __builtin_fma(C.kl_m, thd, r0);
    const double r1 = dpp_f64<kDppNext1>(r0), r2 = dpp_f64<kDppNext2>(r0);
    // first row of the adjugate of [[d0,a,b],[a,d1,e],[b,e,d2]]
    const double a = L.t1 * G.cc1, b = L.t2 * G.cc2, e = L.t12 * G.cc12;
    const double c00 = __builtin_fma(-e, e, L.d12);
    const double c01 = __builtin_fma(b, e, -a * L.d2);
    const double c02 = __builtin_fma(a, e, -b * L.d1);
    const double det = __builtin_fma(L.d0, c00, __builtin_fma(a, c01, b * c02));
    const double num = __builtin_fma(c00, r0, __builtin_fma(c01, r1, c02 * r2));
    const double tdd = num * rcp_f64_1n(det);
    // explicit Euler (remy_swimmer_env.py:87-91); Gddot = (k l / (n m)) (sx, -sy), folded with
    // h into one FMA per component
    gdx = __builtin_fma(C.h_kl_nm, sx, gdx);
    gdy = __builtin_fma(-C.h_kl_nm, sy, gdy);
    thd = __builtin_fma(C.h, tdd, thd);
    // Gdot is NOT re-synchronised across the quad: the lanes sum the same three terms in their
    // own rotated order, so their copies differ by rounding (~1e-17 per step) and each lane
    // integrates its own -- three roundings of one contracting ODE (friction), which stay
    // within ~1e-16 of each other.  The rollout's Gdot_x is lane 0's copy, Gdot_y lane 1's.
    return det;
}

}  // namespace sw
