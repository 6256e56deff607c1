// swimmer_row.h -- latency-optimised rollout step for longer chains (n = 4..8):
// ONE SEGMENT PER LANE, ONE ROLLOUT PER 16-LANE DPP ROW (4 rollouts per wave).
//
// Same idea as swimmer_quad3.h (shorten the one instruction stream a lone wave has to issue
// by spreading a rollout over lanes), with the exchange primitive that scales past a quad:
// gfx950's 64-bit DPP with row_newbcast, which hands lane k's double to all 16 lanes of the
// row -- as a move (`v_mov_b64_dpp`) or fused into a multiply-add (`v_fmac_f64_dpp`: acc +=
// [lane k].x * y, swimmer_row_fused.h).  Every lane therefore sees the other segments in
// canonical order (no rotated frames), replicated quantities (Gdot, barycentre sums) are
// bit-identical on all lanes, and lane dependence sits in per-lane constant vectors only.
//
//   lane i < N of a row owns segment i: theta_i, thetadot_i, sin/cos, row i of Q thdd = r.
//   Sums over segments (policy dot product, normal velocity, barycentre acceleration, right-hand
//   side) are chains of fused broadcast-FMAs reading the other lanes' registers directly.
//   The n x n SPD system is solved COOPERATIVELY: unpivoted, division-free Gaussian elimination
//   where step j broadcasts pivot row j out of lane j's registers and every lane below updates
//   its own row (a multiply and a fused broadcast-FMA per entry, no reciprocal on the pivot
//   chain), then one reciprocal per lane and a broadcast back-substitution; lane i ends with
//   thdd_i.
//   The joint torques never exist as such: lane i needs only u_{i-1} - u_i, which is linear in
//   the observation, so it holds the pre-combined policy row V_i = c12 (W_{i-1} - W_i) and
//   evaluates one 2n+2-term dot product on the state (the mean enters as one constant).
//   Lane i + 8 of a row mirrors lane i (it evaluates the cosine of that segment's angle, see
//   row_step); lanes N..7 and N + 8..15 mirror lane 0 / 8.  Mirrors are never read by a broadcast;
//   their stores are dropped by the buffer range check.
//
// Per step for n = 6: ~250 instructions per lane instead of ~1050 in rollout_kernel<6>.
// Equations and notation: swimmer_device.h.
#pragma once

#include "swimmer_device.h"
#include "swimmer_oct3.h"
#include "swimmer_row_fused.h"

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
    double vwl[N];     // l * vel_w(i,k)
    double af[N];      // -(6k/m) * Aw(i,k)
    double t6[N];      // row i of Q as a factor of cos(th_i - th_k): -6 T(i,k), and the diagonal -6 T(i,i) + 1 at
                       // k = i (there cos(0) comes out as c^2 + s^2 = 1 to an ulp and sin(0) as a product's rounding
                       // residue ~1e-17: the centripetal weight t6[i] * ss stays at rounding level)
    double one[N];     // 1 at k = i
    double nbelow[N];  // -1 where i > k   (elimination step k updates this lane)
    double nabove[N];  // -1 where i < k   (back-substitution level k updates this lane)
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
                t = (ii == k) ? -6.0 * Tw<N>(ii + 1, ii + 1) + 1.0 : -6.0 * Tw<N>(ii + 1, k + 1);
            }
        L.vwl[k] = vw * C.l;
        L.af[k] = -C.six_k_m * a;
        L.t6[k] = t;
        L.one[k] = (seg == k) ? 1.0 : 0.0;
        L.nbelow[k] = (seg > k) ? -1.0 : 0.0;
        L.nabove[k] = (seg < k) ? -1.0 : 0.0;
    }
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
// Division-free Gaussian elimination: step J broadcasts the pivot Q_JJ out of lane J (the
// compiler's own DPP move, it manages that hazard itself) and every lane below replaces its row by
// Q_JJ * row_i - Q_iJ * row_J in ONE asm block of multiplies and fused broadcast-FMAs
// (RowFused<N>::eliminate<J>, which also documents the hazard distances).  Rows are scaled by
// the pivots above them (products of O(1) numbers for these chains), which the back-substitution
// undoes with one reciprocal per lane.
template <int N, int J = 0>
struct RowEliminate {
    static __device__ __forceinline__ void run(const RowLane<N> &L, double (&a)[N], double &b)
    {
        if constexpr (J < N - 1) {
            const double piv = row_bcast<J>(a[J]);
            const double sel = __builtin_fma(L.nbelow[J], 1.0 - piv, 1.0);   // pivot on lanes i > J, else 1
            const double nm = L.nbelow[J] * a[J];                            // -Q_iJ on lanes i > J, else 0
            RowFused<N>::template eliminate<J>(a, b, sel, nm);
            RowEliminate<N, J + 1>::run(L, a, b);
        }
    }
};

// One explicit-Euler step.  gdx, gdy: replicated (bit-identical on all lanes); th, thd: own
// segment; V: this lane's pre-combined policy row, nbias = -V . mean.  Returns this lane's
// reciprocal pivot (positive for a positive definite system).
// sin / cos with LANE ROLES (round 3, as in swimmer_oct3.h): lane i < 8 of the row evaluates the
// SINE of segment i's reduced angle, lane i + 8 -- a full mirror of lane i, same segment, same
// state bit for bit -- the COSINE, with one Horner chain whose coefficients are per-lane state
// (OctTrig: the quarter turns are folded into which polynomial a lane evaluates and its sign);
// `v_mov_b32_dpp row_ror:8` hands each lane its partner's value and a per-lane select puts both in
// canonical (s, c) order, so both halves go on with identical numbers: 10 + 2 + 4 instructions
// instead of 20 (two polynomials + rotation).  T is authoritative for the angle, th = fl(K pi/2 +
// r) is what the policy, the trajectory and the statistics see; the caller checks the range once
// per trip of four steps (oct3_renorm), or -- SLOW -- this function does after the angle update.
// SLOW: re-normalise inside the step, between the angle update and the next step's sin / cos
// (trips during which an angle moves more than kTripSlack, see the kernel).
template <int N, bool SLOW>
__device__ __forceinline__ double row_step(const Consts &C, const RowLane<N> &L,
                                           const double (&V)[2 * N + 2], double nbias,
                                           bool cosine, int designation, double &gdx, double &gdy,
                                           OctTrig &T, double &th, double &thd, double &thmax)
{
    double s, c;
    {
        const double r = T.r, z = r * r;
        double p = fma3(T.k[0], z, T.k[1]);
        p = fma3(p, z, T.k[2]);
        p = fma3(p, z, T.k[3]);
        p = fma3(p, z, T.k[4]);
        p = fma3(p, z, T.k[5]);
        p = fma3(p, z, T.k[6]);
        const double X = __builtin_fma(T.selS, r, T.selC);
        const double own = __builtin_fma(X * z, p, X);        // sin theta on lanes 0..7, cos theta on 8..15
        const double other = dpp_row_f64<kDppRowRor8>(own);
        s = cosine ? other : own;
        c = cosine ? own : other;
    }
    double sk[N], ck[N];
    RowGather<N>::run(s, sk);
    RowGather<N>::run(c, ck);
    // this segment's torque balance c12 (u_{i-1} - u_i) = V_i . obs - V_i . mean
    double tq0 = __builtin_fma(V[0], gdx, nbias), tq1 = V[1] * gdy;
    RowFused<N>::policy(tq0, tq1, th, thd, V, s);
    // own row of cos(th_i - th_k), sin(th_k - th_i) and everything that is linear in them; the
    // k = i entries come out as c^2 + s^2 (= 1 to an ulp) and 0 and carry weight 1 resp. 0
    double vc[N], ac[N], tc[N], a[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double cc = __builtin_fma(c, ck[k], s * sk[k]);
        const double ss = __builtin_fma(c, sk[k], -s * ck[k]);
        vc[k] = L.vwl[k] * cc;                        // normal-velocity weights
        ac[k] = L.af[k] * cc;                         // friction weights of the right-hand side
        tc[k] = L.t6[k] * ss;                         // centripetal weights
        a[k] = L.t6[k] * cc;                          // own row of Q
    }
    // g: normal velocity of this segment's centre;  r: right-hand side of this segment's row
    double g = __builtin_fma(gdy, c, -gdx * s);
    double r = __builtin_fma(C.kl_m, thd, tq0 + tq1);
    const double sq = thd * thd;
    RowFused<N>::velocity(g, r, thd, sq, vc, tc);
    double sx = 0.0, sy = 0.0;
    RowFused<N>::sums(sx, sy, r, g, sk, ck, ac);
    T.r = __builtin_fma(C.h, thd, T.r);               // explicit Euler: the OLD thetadot
    if constexpr (SLOW) {
        if (__any(fabs(T.r) > kPio4)) oct3_renorm(T, designation, thmax);
    }
    th = __builtin_fma(T.kd, kPio2Hi, T.r);
    // everything the elimination reads by DPP is written before the first pivot is broadcast
    RowFused<N>::fence(a);
    RowEliminate<N>::run(L, a, r);
    // this lane's own (scaled) pivot and its reciprocal -- the sign doubles as the positive-
    // definiteness check (the scale factors are pivots of the rows above, positive themselves)
    double dg = L.one[0] * a[0];
#pragma unroll
    for (int k = 1; k < N; ++k) dg = __builtin_fma(L.one[k], a[k], dg);
    const double rq = rcp_f64_1n(dg);
    // back-substitution on the scaled right-hand side (RowFused<N>::backsub)
    double tdd = r * rq;
    const double nu_top = (L.nabove[N - 1] * a[N - 1]) * rq;
    const double na_next = L.nabove[N - 2] * a[N - 2];
    RowFused<N>::backsub(tdd, rq, nu_top, na_next, a, L.nabove);
    // Gddot = (k l / (n m)) (sx, -sy), folded with h into one FMA per component
    gdx = __builtin_fma(C.h_kl_nm, sx, gdx);
    gdy = __builtin_fma(-C.h_kl_nm, sy, gdy);
    thd = __builtin_fma(C.h, tdd, thd);
    return rq;
}

}  // namespace sw
