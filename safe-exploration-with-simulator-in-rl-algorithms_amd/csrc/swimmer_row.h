// swimmer_row.h -- latency-optimised rollout step for longer chains (n = 4..8):
// ONE SEGMENT PER LANE, ONE ROLLOUT PER 16-LANE DPP ROW (4 rollouts per wave).
//
// Same idea as swimmer_quad3.h (shorten the one instruction stream a lone wave has to issue
// by spreading a rollout over lanes), with the exchange primitive that scales past a quad:
// gfx950's 64-bit DPP move with row_newbcast (`v_mov_b64_dpp ... row_newbcast:k`), ONE
// instruction that hands lane k's double to all 16 lanes of the row.  Every lane therefore
// sees neighbours in canonical segment order (no rotated frames), replicated quantities
// (Gdot, barycentre sums) are bit-identical on all lanes, and lane dependence sits in
// per-lane constant vectors only.
//
//   lane i < N of a row owns segment i: theta_i, thetadot_i, sin/cos, row i of Q thdd = r.
//   The n x n SPD system is solved COOPERATIVELY: unpivoted Gaussian elimination where step j
//   broadcasts pivot row j (lane j's registers) and every lane below updates its own row,
//   then a broadcast back-substitution; lane i ends with thdd_i.  (~120 instructions for
//   n = 6 instead of ~170 for a redundant per-lane LDL^T plus the full matrix build.)
//   The joint torques never exist as such: lane i needs only u_{i-1} - u_i, which is linear in
//   the observation, so it holds the pre-combined policy row V_i = c12 (W_{i-1} - W_i) and
//   evaluates one 2n+2-term dot product on (state - mean).
//   Lanes N..15 of a row mirror lane 0 and are never read; their stores are dropped by the
//   buffer range check.
//
// Per step for n = 6: ~300 instructions per lane instead of ~1050 in rollout_kernel<6>.
// Equations and notation: swimmer_device.h.
#pragma once

#include "swimmer_device.h"

namespace sw {

// lane K of this lane's 16-lane row -> every lane of the row
template <int K>
__device__ __forceinline__ double row_bcast(double v)
{
    const long long r = __builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v),
                                                    0x150 + K, 0xf, 0xf, true);
    return __longlong_as_double(r);
}

template <int N>
struct RowLane {
    double vwl[N];    // l * vel_w(i,k)
    double af[N];     // -(6k/m) * Aw(i,k)
    double t6[N];     // -6 T(i,k), 0 for k = i
    double qd;        // -6 T(i,i) + 1
    double one[N];    // 1 at k = i
    double below[N];  // 1 where i > k   (elimination step k updates this lane)
    double above[N];  // 1 where i < k   (back-substitution step k updates this lane)
};

template <int N>
__device__ __forceinline__ RowLane<N> row_lane(const Consts &C, int seg)
{
    RowLane<N> L;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double vw = 0.0, a = 0.0, t = 0.0;
#pragma unroll
        for (int ii = 0; ii < N; ++ii)
            if (seg == ii) {
                vw = vel_w<N>(ii + 1, k + 1);
                a = Aw<N>(ii + 1, k + 1);
                t = (ii == k) ? 0.0 : -6.0 * Tw<N>(ii + 1, k + 1);
            }
        L.vwl[k] = vw * C.l;
        L.af[k] = -C.six_k_m * a;
        L.t6[k] = t;
        L.one[k] = (seg == k) ? 1.0 : 0.0;
        L.below[k] = (seg > k) ? 1.0 : 0.0;
        L.above[k] = (seg < k) ? 1.0 : 0.0;
    }
    double qd = 0.0;
#pragma unroll
    for (int ii = 0; ii < N; ++ii)
        if (seg == ii) qd = -6.0 * Tw<N>(ii + 1, ii + 1) + 1.0;
    L.qd = qd;
    return L;
}

// Gather lane k's value for every k (compile-time unrolled).
template <int N, int K = 0>
struct RowGather {
    static __device__ __forceinline__ void run(double v, double (&out)[N])
    {
        out[K] = row_bcast<K>(v);
        RowGather<N, K + 1>::run(v, out);
    }
};
template <int N>
struct RowGather<N, N> {
    static __device__ __forceinline__ void run(double, double (&)[N]) {}
};

// Cooperative solve of Q x = b: lane i holds row i (a[0..N-1]) and b_i; ends with x_i on lane i.
// rq: lane i's 1 / (pivot i) -- its sign doubles as the positive-definiteness check.
template <int N, int J = 0>
struct RowEliminate {
    static __device__ __forceinline__ void run(const RowLane<N> &L, double (&a)[N], double &b, double &rq)
    {
        const double rp = rcp_f64_1n(row_bcast<J>(a[J]));
        rq = __builtin_fma(L.one[J], rp, rq);          // lane J keeps 1 / Q_JJ
        if (J < N - 1) {
            const double f = (L.below[J] * a[J]) * rp;  // Q_iJ / Q_JJ on lanes i > J, else 0
#pragma unroll
            for (int k = J + 1; k < N; ++k) a[k] = __builtin_fma(-f, row_bcast<J>(a[k]), a[k]);
            b = __builtin_fma(-f, row_bcast<J>(b), b);
        }
        RowEliminate<N, J + 1>::run(L, a, b, rq);
    }
};
template <int N>
struct RowEliminate<N, N> {
    static __device__ __forceinline__ void run(const RowLane<N> &, double (&)[N], double &, double &) {}
};

// Back-substitution: step J broadcasts x_J = b_J / Q_JJ (lane J's b is final by then) and the
// lanes above subtract their U entry times it.  Lane i's b only changes at steps J > i, so
// after the last step b * rq IS lane i's solution -- no per-step capture needed.
template <int N, int J = N - 1>
struct RowBackSub {
    static __device__ __forceinline__ void run(const RowLane<N> &L, const double (&a)[N], double &b, double rq)
    {
        if (J > 0) {
            const double xj = row_bcast<J>(b * rq);
            b = __builtin_fma(-(L.above[J] * a[J]), xj, b);
        }
        RowBackSub<N, J - 1>::run(L, a, b, rq);
    }
};
template <int N>
struct RowBackSub<N, -1> {
    static __device__ __forceinline__ void run(const RowLane<N> &, const double (&)[N], double &, double) {}
};

// Every segment's sin / cos, gathered in canonical order (the angle-only part of a step).
template <int N>
struct RowGeo {
    double s, c;            // own segment
    double sk[N], ck[N];    // every segment, canonical order
};

template <int N>
__device__ __forceinline__ RowGeo<N> row_geometry(double th)
{
    RowGeo<N> G;
    sincos_fast(th, G.s, G.c);
    RowGather<N>::run(G.s, G.sk);
    RowGather<N>::run(G.c, G.ck);
    return G;
}

// The velocity-dependent part of one explicit-Euler step: updates gdx, gdy (replicated,
// bit-identical on all lanes) and thd (own segment); theta is advanced by the caller.
// wk: every segment's thetadot (already gathered by the caller for the policy);
// tq_scaled = c12 (u_{i-1} - u_i) for this lane's segment.  Returns this lane's reciprocal
// pivot (positive for a positive definite system).
template <int N>
__device__ __forceinline__ double row_dynamics(const Consts &C, const RowLane<N> &L,
                                               const RowGeo<N> &G, double &gdx, double &gdy,
                                               double &thd, const double (&wk)[N],
                                               double tq_scaled)
{
    // own row of cos(th_i - th_k), sin(th_k - th_i); the k = i entries come out as
    // c^2 + s^2 (= 1 to an ulp) and 0 and carry weight 1 resp. 0 below
    double cc[N], ss[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        cc[k] = __builtin_fma(G.c, G.ck[k], G.s * G.sk[k]);
        ss[k] = __builtin_fma(G.c, G.sk[k], -G.s * G.ck[k]);
    }
    // normal velocity of this segment's centre
    double g = __builtin_fma(gdy, G.c, -gdx * G.s);
#pragma unroll
    for (int k = 0; k < N; ++k) g = __builtin_fma(L.vwl[k] * cc[k], wk[k], g);
    double gk[N];
    RowGather<N>::run(g, gk);
    // barycentre acceleration, canonical order -> identical on every lane
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        sx = __builtin_fma(gk[k], G.sk[k], sx);
        sy = __builtin_fma(gk[k], G.ck[k], sy);
    }
    // right-hand side of this segment's row
    double r = __builtin_fma(C.kl_m, thd, tq_scaled);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        r = __builtin_fma(L.af[k] * cc[k], gk[k], r);
        r = __builtin_fma(L.t6[k] * ss[k], wk[k] * wk[k], r);
    }
    // own row of Q
    double a[N];
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = __builtin_fma(L.t6[k], cc[k], L.one[k] * L.qd);
    double rq = 0.0;
    RowEliminate<N>::run(L, a, r, rq);
    RowBackSub<N>::run(L, a, r, rq);
    const double tdd = r * rq;
    // Gddot = (k l / (n m)) (sx, -sy), folded with h into one FMA per component
    gdx = __builtin_fma(C.h_kl_nm, sx, gdx);
    gdy = __builtin_fma(-C.h_kl_nm, sy, gdy);
    thd = __builtin_fma(C.h, tdd, thd);
    return rq;
}

}  // namespace sw
