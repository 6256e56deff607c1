// swimmer_twin.h -- the reference's NATIVE swimmer model
// (rlglue/environment/SwimmerEnvironment.cpp) on the device, selected by SW_FLAG_MODEL_TWIN.
//
// It is a different numerical model from the Gym env and is reproduced as
// written, quirks included (they are what its recorded outputs pin):
//   * the torque equation of segment i couples f_i and f_{i+1} instead of f_{i-1} and f_i,
//     and for i = n the "f_{n+1}" columns are Gdd_1's (SwimmerEnvironment.cpp:151-157);
//   * friction torque enters as -k thd l^3/12 (:268);
//   * the first segment's centre velocity G1_dot weights thd_1 n_1 by n/2 * (l/n), not by
//     (n - 1/2) * (l/n) (:250-257);
//   * semi-implicit Euler: thd first, then th with the NEW thd (:231-235);
//   * row 0's read of torque[-1] (:160, undefined behaviour) is taken as 0.
//
// The reference assembles a dense (5n+2)^2 system in (thdd_i, f_0..f_n, Gdd_1..Gdd_n) and
// solves it with Eigen's ColPivHouseholderQR.  Eliminating f and Gdd analytically (the force
// balance rows telescope exactly as in swimmer_device.h: Gdd_j = Gdd + C_j - Cbar with
// n m Gdd = sum_j F_j) leaves an n x n system in thdd with the same cos/sin(th_i - th_k)
// entries but shifted weight tables; it is NOT symmetric, so it is solved by Gaussian
// elimination with partial pivoting (branch-free row exchanges).  Checked against the dense
// restatement (oracle/twin_oracle.c, which reproduces the reference's two recorded outputs):
// <= 8e-14 for n = 2..8.
#pragma once

#include "swimmer_device.h"

namespace sw {

// chain weight of thd_t n_t (t 1-based) in the head-frame centre of segment j
template <int N> __host__ __device__ constexpr double chain_w(int j1, int t1)
{
    return t1 < j1 ? 1.0 : (t1 == j1 ? 0.5 : 0.0);
}
// twin velocity weights: Gdot_j = Gdot + l sum_t vwt(j,t) thd_t n_t
template <int N> __host__ __device__ constexpr double vel_w_twin(int j1, int t1)
{
    const double e = (t1 == 1) ? N / 2.0 : (N - t1 + 0.5);   // (:250-257)
    return chain_w<N>(j1, t1) - chain_w<N>(1, t1) - e / N;
}
// W(i,k) = sum_{j<=i} (w(j,k) - wbar_k): coefficient of l thdd_k n_k in f_i / m
template <int N> __host__ __device__ constexpr double Wsum(int i1, int k1)
{
    double a = 0.0;
    for (int j = 1; j <= i1; ++j) a += chain_w<N>(j, k1) - wbar<N>(k1);
    return a;
}

// A x = b, general N x N, partial pivoting, fully unrolled; returns false on a zero pivot.
template <int N>
__device__ __forceinline__ bool pivoted_solve(double (&A)[N][N], double (&b)[N])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // bring the largest |A[i][j]|, i >= j, to row j by conditional row exchanges
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            const bool sw_ = __builtin_fabs(A[i][j]) > __builtin_fabs(A[j][j]);
#pragma unroll
            for (int c = j; c < N; ++c) {
                const double u = A[j][c], v = A[i][c];
                A[j][c] = sw_ ? v : u;
                A[i][c] = sw_ ? u : v;
            }
            const double u = b[j], v = b[i];
            b[j] = sw_ ? v : u;
            b[i] = sw_ ? u : v;
        }
        ok = ok && (A[j][j] != 0.0) && (__builtin_fabs(A[j][j]) < 1.0e300);
        const double rp = 1.0 / A[j][j];
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            const double f = A[i][j] * rp;
#pragma unroll
            for (int c = j + 1; c < N; ++c) A[i][c] = __builtin_fma(-f, A[j][c], A[i][c]);
            b[i] = __builtin_fma(-f, b[j], b[i]);
        }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int c = i + 1; c < N; ++c) s = __builtin_fma(-A[i][c], b[c], s);
        b[i] = s / A[i][i];
    }
    return ok;
}

struct TwinConsts {
    double l, m, k, h, dirx, diry;
};

template <int N>
__device__ __forceinline__ bool accelerations_twin(const TwinConsts &C, double gdx, double gdy,
                                                   const double (&th)[N], const double (&thd)[N],
                                                   const double (&u)[N > 1 ? N - 1 : 1],
                                                   double &gddx, double &gddy, double (&tdd)[N])
{
    const double l = C.l, m = C.m, k = C.k;
    double s[N], c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) sincos_fast(th[i], s[i], c[i]);
    double cc[N][N], ss[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cc[i][i] = 1.0;
        ss[i][i] = 0.0;
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
            cc[i][j] = cc[j][i] = __builtin_fma(c[i], c[j], s[i] * s[j]);
            ss[i][j] = __builtin_fma(c[i], s[j], -s[i] * c[j]);
            ss[j][i] = -ss[i][j];
        }
    }
    // friction (compute_friction, :238-271): F_j = -k l (Gdot_j . n_j)
    double F[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double g = __builtin_fma(gdy, c[j], -gdx * s[j]);
#pragma unroll
        for (int t = 0; t < N; ++t) g = __builtin_fma((l * vel_w_twin<N>(j + 1, t + 1)) * cc[j][t], thd[t], g);
        F[j] = -k * l * g;
    }
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        sx = __builtin_fma(F[j], s[j], sx);
        sy = __builtin_fma(F[j], c[j], sy);
    }
    gddx = -sx / (N * m);
    gddy = sy / (N * m);

    double A[N][N], r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double nG = __builtin_fma(c[i], gddy, -s[i] * gddx);
        double cst = 0.0;
#pragma unroll
        for (int kk = 0; kk < N; ++kk) {
            // weight of (l thdd_k n_k - l thd_k^2 p_k) in  n_i . (f_i + f_{i+1}) / m   (i < n)
            // resp. in  n_n . Gdd_1  (i = n)
            const double wgt = (i < N - 1) ? (Wsum<N>(i + 1, kk + 1) + Wsum<N>(i + 2, kk + 1))
                                           : (chain_w<N>(1, kk + 1) - wbar<N>(kk + 1));
            const double scale = (i < N - 1) ? m * l : l;
            A[i][kk] = (-(l / 2) * scale * wgt) * cc[i][kk];
            cst = __builtin_fma((scale * wgt) * (-thd[kk] * thd[kk]), ss[i][kk], cst);
        }
        if (i < N - 1) {
            cst = __builtin_fma(m * (2 * (i + 1) + 1), nG, cst);
#pragma unroll
            for (int j = 0; j <= i + 1; ++j) cst = __builtin_fma(-((j <= i) ? 2.0 : 1.0) * F[j], cc[i][j], cst);
        } else {
            cst += nG;
        }
        A[i][i] += m * l * l / 12.0;
        double tq = 0.0;
        if (i >= 1) tq += u[i - 1];
        if (i < N - 1) tq -= u[i];
        r[i] = __builtin_fma(l / 2, cst, tq - k * thd[i] * (l * l * l) / 12.0);
    }
    const bool ok = pivoted_solve<N>(A, r);
#pragma unroll
    for (int i = 0; i < N; ++i) tdd[i] = r[i];
    return ok;
}

// updateState with semi_implicit_euler (:228-236), in place; reward = Gdot_new . direction
template <int N>
__device__ __forceinline__ bool twin_step(const TwinConsts &C, double &gdx, double &gdy,
                                          double (&th)[N], double (&thd)[N],
                                          const double (&u)[N > 1 ? N - 1 : 1], double &reward)
{
    double gddx, gddy, tdd[N];
    const bool ok = accelerations_twin<N>(C, gdx, gdy, th, thd, u, gddx, gddy, tdd);
    gdx = __builtin_fma(C.h, gddx, gdx);
    gdy = __builtin_fma(C.h, gddy, gdy);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        thd[i] = __builtin_fma(C.h, tdd[i], thd[i]);
        th[i] = __builtin_fma(C.h, thd[i], th[i]);     // new theta_dot
    }
    reward = __builtin_fma(gdx, C.dirx, gdy * C.diry);
    return ok;
}

}  // namespace sw
