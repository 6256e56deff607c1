/*
 * swimmer_oracle.c -- CPU restatement (plain C, fp64) of the reference's Gym swimmer step and
 * of the rollout loop that drives it.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg do, and only
 * as the checker / reported CPU baseline.  The shipped path is the HIP library in
 * safe-exploration-with-simulator-in-rl-algorithms_amd/csrc/ and fails loudly without it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * the .npz files in tests/golden/, which tests/golden/make_golden.py produced by running the reference
 * itself (envs/gym_swimmer/swimmer/remy_swimmer_env.py, ars/environment.py) in the build
 * container, and against the reference-authored known answer
 * rlglue/test/acceleration-compare.txt:5-6 (Coulom's barycentre acceleration).
 *
 * The arithmetic follows the reference statement by statement, in the same evaluation
 * order (Python's left-to-right operator order is kept, e.g. `1 / n * (a + b) / 2` is
 * ((1/n)*(a+b))/2), so the only sources of difference are libm sin/cos/pow vs numpy's and
 * the pivoted LU's summation order vs the LAPACK build numpy links.
 *
 * Third-party arithmetic on the path: numpy.linalg.solve -> LAPACK dgesv (partial-pivot
 * LU); numpy is unpinned in the reference (README.md:9-15).  swo_lu_solve() restates dgesv's
 * published algorithm (dgetf2: column pivot search by max |a|, row swap, scale by the
 * reciprocal pivot, rank-1 update; then forward/back substitution).
 *
 * Layout: states are the reference's AoS observation [Gdx, Gdy, th1, thd1, ..., thn, thdn]
 * (remy_swimmer_env.py:216-224).
 */
#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "swimmer_oracle.h"

#define NMAX SWO_NMAX
#define ROW (NMAX + 3) /* affine row: [Gddx, Gddy, thdd_1..thdd_n, const] */

/* numpy.linalg.solve(A, B) for one right-hand side; A is nn x nn row-major (destroyed).
 * Returns 0, or 1 when a pivot is exactly zero (numpy raises LinAlgError: Singular matrix). */
static int swo_lu_solve(int nn, double A[][NMAX + 2], double *b)
{
    for (int j = 0; j < nn; ++j) {
        int p = j;
        double best = fabs(A[j][j]);
        for (int i = j + 1; i < nn; ++i) {
            double v = fabs(A[i][j]);
            if (v > best) { best = v; p = i; }
        }
        if (A[p][j] == 0.0) return 1;
        if (p != j) {
            for (int c = 0; c < nn; ++c) { double t = A[j][c]; A[j][c] = A[p][c]; A[p][c] = t; }
            double t = b[j]; b[j] = b[p]; b[p] = t;
        }
        double rp = 1.0 / A[j][j];
        for (int i = j + 1; i < nn; ++i) A[i][j] *= rp;
        for (int i = j + 1; i < nn; ++i) {
            double lij = A[i][j];
            for (int c = j + 1; c < nn; ++c) A[i][c] -= lij * A[j][c];
        }
    }
    /* L y = P b (unit lower), then U x = y */
    for (int i = 1; i < nn; ++i) {
        double s = b[i];
        for (int c = 0; c < i; ++c) s -= A[i][c] * b[c];
        b[i] = s;
    }
    for (int i = nn - 1; i >= 0; --i) {
        double s = b[i];
        for (int c = i + 1; c < nn; ++c) s -= A[i][c] * b[c];
        b[i] = s / A[i][i];
    }
    return 0;
}

/* SwimmerEnv.compute_accelerations (remy_swimmer_env.py:95-114) */
int swo_accelerations(const swo_params *p, const double *state, const double *u,
                      double *gdd, double *tdd)
{
    const int n = p->n;
    if (n < 1 || n > NMAX) return -1;
    const int K = n + 2; /* index of the constant term in an affine row */
    const double l = p->l_i, m = p->m_i, k = p->k;
    double Gd[2] = { state[0], state[1] };
    double th[NMAX], thd[NMAX], s[NMAX], c[NMAX];
    for (int i = 0; i < n; ++i) {
        th[i] = state[2 + 2 * i];
        thd[i] = state[3 + 2 * i];
        s[i] = sin(th[i]);
        c[i] = cos(th[i]);
    }

    /* compute_points_speed_acc (remy_swimmer_env.py:116-166) */
    double AdX[NMAX + 1], AdY[NMAX + 1];
    double AddX[NMAX + 1][ROW], AddY[NMAX + 1][ROW];
    double GdhX = 0.0, GdhY = 0.0, GddhX[ROW], GddhY[ROW];
    memset(AddX, 0, sizeof AddX);
    memset(AddY, 0, sizeof AddY);
    memset(GddhX, 0, sizeof GddhX);
    memset(GddhY, 0, sizeof GddhY);
    AdX[0] = AdY[0] = 0.0;
    const double inv_n = 1.0 / (double)n;
    for (int i = 1; i <= n; ++i) {
        AdX[i] = AdX[i - 1] - l * thd[i - 1] * s[i - 1];                       /* :130 */
        AdY[i] = AdY[i - 1] + l * thd[i - 1] * c[i - 1];                       /* :132 */
        memcpy(AddX[i], AddX[i - 1], sizeof AddX[i]);                          /* :137 */
        memcpy(AddY[i], AddY[i - 1], sizeof AddY[i]);
        AddX[i][2 + (i - 1)] -= l * s[i - 1];                                  /* :139 */
        AddY[i][2 + (i - 1)] += l * c[i - 1];                                  /* :140 */
        AddX[i][K] -= l * pow(thd[i - 1], 2.0) * c[i - 1];                     /* :141 */
        AddY[i][K] -= l * pow(thd[i - 1], 2.0) * s[i - 1];                     /* :143 */
        GdhX += inv_n * (AdX[i - 1] + AdX[i]) / 2.0;                           /* :147 */
        GdhY += inv_n * (AdY[i - 1] + AdY[i]) / 2.0;
        for (int q = 0; q <= K; ++q) {                                         /* :151-152 */
            GddhX[q] += inv_n * (AddX[i - 1][q] + AddX[i][q]) / 2.0;
            GddhY[q] += inv_n * (AddY[i - 1][q] + AddY[i][q]) / 2.0;
        }
    }
    /* change of frame (:157-163) */
    {
        const double dx = Gd[0] - GdhX, dy = Gd[1] - GdhY;
        for (int i = 0; i <= n; ++i) { AdX[i] += dx; AdY[i] += dy; }
        for (int i = 0; i <= n; ++i) {
            AddX[i][0] += 1.0;
            AddY[i][1] += 1.0;
            for (int q = 0; q <= K; ++q) { AddX[i][q] -= GddhX[q]; AddY[i][q] -= GddhY[q]; }
        }
    }

    /* compute_joint_force (:168-187) */
    double fX[NMAX + 1][ROW], fY[NMAX + 1][ROW];
    memset(fX, 0, sizeof fX);
    memset(fY, 0, sizeof fY);
    for (int i = 1; i <= n; ++i) {
        for (int q = 0; q <= K; ++q) {
            fX[i][q] = fX[i - 1][q] + m * (AddX[i - 1][q] + AddX[i][q]) / 2.0; /* :174 */
            fY[i][q] = fY[i - 1][q] + m * (AddY[i - 1][q] + AddY[i][q]) / 2.0;
        }
        const double nx = -s[i - 1], ny = c[i - 1];                            /* :179 */
        const double gx = (AdX[i - 1] + AdX[i]) / 2.0, gy = (AdY[i - 1] + AdY[i]) / 2.0;
        const double F = -k * l * (gx * nx + gy * ny);                         /* :182 */
        fX[i][K] -= F * nx;
        fY[i][K] -= F * ny;
    }

    /* compute_dynamic_matrix (:189-207) */
    double S[NMAX + 2][ROW];
    memset(S, 0, sizeof S);
    for (int q = 0; q <= K; ++q) { S[0][q] = fX[n][q]; S[1][q] = fY[n][q]; }
    for (int i = 1; i <= n; ++i) {
        double *r = S[2 + (i - 1)];
        for (int q = 0; q <= K; ++q)
            r[q] += l / 2.0 * (c[i - 1] * (fY[i][q] + fY[i - 1][q])
                               - s[i - 1] * (fX[i][q] + fX[i - 1][q]));         /* :196-198 */
        r[2 + (i - 1)] -= m * pow(l, 2.0) / 12.0;                              /* :199 */
        r[K] += k * thd[i - 1] * pow(l, 3.0) / 12.0;                           /* :200 */
        if (i - 2 >= 0) r[K] += u[i - 2];                                      /* :202 */
        if (i - 1 < n - 1) r[K] -= u[i - 1];                                   /* :204 */
    }

    /* solve (:209-214) */
    double A[NMAX + 2][NMAX + 2], B[NMAX + 2];
    for (int i = 0; i < K; ++i) {
        for (int q = 0; q < K; ++q) A[i][q] = S[i][q];
        B[i] = -S[i][K];
    }
    if (swo_lu_solve(K, A, B)) return 1;
    gdd[0] = B[0];
    gdd[1] = B[1];
    for (int i = 0; i < n; ++i) tdd[i] = B[2 + i];
    return 0;
}

/* SwimmerEnv.step -> next_observation (explicit Euler, :87-91) -> get_state / get_reward */
int swo_step(const swo_params *p, const double *state, const double *u,
             double *next, double *reward)
{
    const int n = p->n;
    double gdd[2], tdd[NMAX];
    int rc = swo_accelerations(p, state, u, gdd, tdd);
    if (rc) return rc;
    const double h = p->h;
    double out[2 * NMAX + 2];
    out[0] = state[0] + h * gdd[0];
    out[1] = state[1] + h * gdd[1];
    for (int i = 0; i < n; ++i) {
        const double th = state[2 + 2 * i], thd = state[3 + 2 * i];
        out[3 + 2 * i] = thd + h * tdd[i];
        out[2 + 2 * i] = th + h * thd; /* old theta_dot: explicit Euler */
    }
    memcpy(next, out, sizeof(double) * (size_t)(2 * n + 2));
    if (reward) *reward = out[0] * p->dir_x + out[1] * p->dir_y;              /* :243 */
    return 0;
}

/* SwimmerEnv.reset (:58-67) */
void swo_reset(const swo_params *p, double *state)
{
    state[0] = 0.0;
    state[1] = 0.0;
    for (int i = 0; i < p->n; ++i) {
        state[2 + 2 * i] = M_PI / 2;
        state[3 + 2 * i] = 0.0;
    }
}

/* Environment.select_action + Environment.rollout (ars/environment.py:19-57).
 * mean == NULL or cov_diag == NULL selects the V1 action a = P s; otherwise V2:
 * a = (P diag(cov_diag ** -0.5)) (s - mean).  state0 == NULL starts from reset(). */
int swo_rollout(const swo_params *p, int H, const double *policy, const double *mean,
                const double *cov_diag, const double *state0, double *ret, double *traj)
{
    const int n = p->n;
    if (n < 1 || n > NMAX) return -1;
    const int d = 2 * n + 2, m = n - 1;
    double s[2 * NMAX + 2], a[NMAX], W[NMAX * (2 * NMAX + 2)];
    const int v2 = (mean != NULL && cov_diag != NULL);
    if (state0) memcpy(s, state0, sizeof(double) * (size_t)d);
    else swo_reset(p, s);
    if (v2) {
        for (int j = 0; j < d; ++j) {
            double sc = pow(cov_diag[j], -0.5);                /* environment.py:32 */
            for (int i = 0; i < m; ++i) W[i * d + j] = policy[i * d + j] * sc;
        }
    } else {
        memcpy(W, policy, sizeof(double) * (size_t)(m * d));
    }
    double total = 0.0;
    for (int t = 0; t < H; ++t) {
        for (int i = 0; i < m; ++i) {
            double acc = 0.0;
            for (int j = 0; j < d; ++j)
                acc += W[i * d + j] * (v2 ? (s[j] - mean[j]) : s[j]);
            a[i] = acc;
        }
        double r;
        int rc = swo_step(p, s, a, s, &r);
        if (rc) return rc;
        if (traj) memcpy(traj + (size_t)t * d, s, sizeof(double) * (size_t)d);
        total += r;
    }
    *ret = total;
    return 0;
}

/* Threads the batched variants run on (1 without OpenMP). */
int swo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void swo_set_num_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* Batched variants.  Rollouts / envs are independent, so they are spread over the host
 * cores with OpenMP; this is what bench.py times as `cpu_baseline`. */
int swo_step_batch(const swo_params *p, long n_env, const double *states, const double *actions,
                   double *next, double *rewards)
{
    const int d = 2 * p->n + 2, m = p->n - 1;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (long e = 0; e < n_env; ++e)
        bad |= swo_step(p, states + e * d, actions + e * m, next + e * d, rewards + e);
    return bad;
}

int swo_rollout_batch(const swo_params *p, long n_roll, int H, const double *policies,
                      const double *mean, const double *cov_diag, double *returns,
                      double *traj)
{
    const int d = 2 * p->n + 2, m = p->n - 1;
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad)
    for (long r = 0; r < n_roll; ++r)
        bad |= swo_rollout(p, H, policies + r * m * d, mean, cov_diag, NULL, returns + r,
                           traj ? traj + (size_t)r * H * d : NULL);
    return bad;
}
