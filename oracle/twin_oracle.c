/*
 * twin_oracle.c -- CPU restatement (plain C, fp64) of the reference's NATIVE swimmer
 * environment, rlglue/environment/SwimmerEnvironment.cpp: compute_friction (:238-271),
 * compute_accelerations (:139-226: the (5n+2) x (5n+2) system in the unknowns
 * thdd_i, f_0..f_n, Gdd_1..Gdd_n), semi_implicit_euler (:228-236), calculate_reward (:273-277).
 *
 * TEST INFRASTRUCTURE ONLY (see swimmer_oracle.c's header for the rule).
 *
 * This is a DIFFERENT numerical model from the Gym env and is restated
 * as written, quirks included:
 *   - row i-1 of the torque equations couples f_i and f_{i+1} (columns n+2i .. n+2i+3); for
 *     i = n those columns are Gdd_1's (:153-157);
 *   - `if (i-2>=0)` on a size_t is always true, so row 0 reads torque[(size_t)-1]
 *     (:160, undefined behaviour; the recorded outputs are reproduced with that read = 0,
 *     which is what this restatement uses);
 *   - compute_friction's G1_dot sum gives the first segment weight 1/2 even for i = 1 (:251-257).
 * The reference solves with Eigen's ColPivHouseholderQR (:213; Eigen 3.3.7 vendored under
 * rlglue/Eigen, not shipped); the matrix is square and nonsingular, so the solution is
 * unique and is computed here by LU with partial pivoting (agreement to rounding).
 *
 * Parity status: pinned ONLY by the two reference-authored known answers
 * rlglue/test/acceleration-compare.txt:102-103 and rlglue/test/swimmer-compare.txt:100
 * (6 significant digits each; tests/test_twin.py).  The C++ source cannot be built here without
 * RL-Glue's headers, which the image lacks, so there is no oracle/_ref.
 */
#include <math.h>
#include <string.h>

#include "swimmer_oracle.h"

#define NMAX SWO_NMAX
#define DIM (5 * NMAX + 2)

static int lu_solve(int nn, double A[][DIM], double *b)
{
    for (int j = 0; j < nn; ++j) {
        int p = j;
        double best = fabs(A[j][j]);
        for (int i = j + 1; i < nn; ++i)
            if (fabs(A[i][j]) > best) { best = fabs(A[i][j]); p = i; }
        if (best == 0.0) return 1;
        if (p != j) {
            for (int c = 0; c < nn; ++c) { double t = A[j][c]; A[j][c] = A[p][c]; A[p][c] = t; }
            double t = b[j]; b[j] = b[p]; b[p] = t;
        }
        for (int i = j + 1; i < nn; ++i) {
            const double f = A[i][j] / A[j][j];
            if (f == 0.0) continue;
            for (int c = j; c < nn; ++c) A[i][c] -= f * A[j][c];
            b[i] -= f * b[j];
        }
    }
    for (int i = nn - 1; i >= 0; --i) {
        double s = b[i];
        for (int c = i + 1; c < nn; ++c) s -= A[i][c] * b[c];
        b[i] = s / A[i][i];
    }
    return 0;
}

/* compute_friction (SwimmerEnvironment.cpp:238-271) */
static void twin_friction(const swo_params *p, const double *Gd, const double *th, const double *thd,
                          double F[][2], double *M)
{
    const int n = p->n;
    const double l = p->l_i, k = p->k;
    double nx[NMAX], ny[NMAX];
    for (int i = 0; i < n; ++i) { nx[i] = -sin(th[i]); ny[i] = cos(th[i]); }
    double g1x = Gd[0], g1y = Gd[1];
    for (int i = 1; i <= n; ++i) {                               /* :250-257 */
        double sx = 0.0, sy = 0.0;
        for (int j = 0; j < i; ++j) {
            const double e = (j == 0 || j == i - 1) ? 0.5 : 1.0;
            sx += e * thd[j] * nx[j];
            sy += e * thd[j] * ny[j];
        }
        g1x -= l / n * sx;
        g1y -= l / n * sy;
    }
    double gx[NMAX], gy[NMAX];
    gx[0] = g1x; gy[0] = g1y;
    for (int i = 1; i < n; ++i) {                                /* :261-263 */
        gx[i] = gx[i - 1] + l / 2 * thd[i - 1] * nx[i - 1] + l / 2 * thd[i] * nx[i];
        gy[i] = gy[i - 1] + l / 2 * thd[i - 1] * ny[i - 1] + l / 2 * thd[i] * ny[i];
    }
    for (int i = 0; i < n; ++i) {                                /* :266-269 */
        const double dot = gx[i] * nx[i] + gy[i] * ny[i];
        F[i][0] = -k * l * dot * nx[i];
        F[i][1] = -k * l * dot * ny[i];
        M[i] = -k * thd[i] * pow(l, 3.0) / 12.0;
    }
}

/* the dense system A X = B of compute_accelerations (:139-211), unknowns
 * X = (thdd_1..n | f_0x f_0y .. f_nx f_ny | Gdd_1x Gdd_1y .. Gdd_nx Gdd_ny) */
static void twin_assemble(const swo_params *p, const double *state, const double *u,
                          double A[][DIM], double *B)
{
    const int n = p->n;
    const int nn = 5 * n + 2;
    const double l = p->l_i, m = p->m_i;
    double th[NMAX], thd[NMAX], F[NMAX][2], Mf[NMAX];
    for (int i = 0; i < n; ++i) { th[i] = state[2 + 2 * i]; thd[i] = state[3 + 2 * i]; }
    twin_friction(p, state, th, thd, F, Mf);
    for (int i = 0; i < nn; ++i) { memset(A[i], 0, sizeof(double) * (size_t)nn); B[i] = 0.0; }
    for (int i = 1; i <= n; ++i) {                               /* :151-164 */
        A[i - 1][i - 1] = m * pow(l, 2.0) / 12.0;
        A[i - 1][n + 2 * i + 0] = +l / 2 * sin(th[i - 1]);
        A[i - 1][n + 2 * i + 2] = +l / 2 * sin(th[i - 1]);
        A[i - 1][n + 2 * i + 1] = -l / 2 * cos(th[i - 1]);
        A[i - 1][n + 2 * i + 3] = -l / 2 * cos(th[i - 1]);
        B[i - 1] = Mf[i - 1];
        if (i - 2 >= 0) B[i - 1] += u[i - 2];                    /* :160, torque[-1] taken as 0 */
        if (i - 1 < n - 1) B[i - 1] -= u[i - 1];
    }
    A[n][n] = 1.0;                                               /* :169-170 */
    A[n + 1][n + 1] = 1.0;
    for (int i = 1; i <= n; ++i)                                 /* :172-182 */
        for (int d = 0; d < 2; ++d) {
            const int row = n + 2 + 2 * (i - 1) + d;
            A[row][d + n + 2 * (i - 1)] = 1.0;
            A[row][d + n + 2 * i] = -1.0;
            A[row][d + 3 * n + 2 * i] = m;
            B[row] = F[i - 1][d];
        }
    A[3 * n + 2][3 * n] = 1.0;                                   /* :184-185 */
    A[3 * n + 3][3 * n + 1] = 1.0;
    for (int i = 1; i < n; ++i)                                  /* :188-209 */
        for (int d = 0; d < 2; ++d) {
            const int row = 3 * n + 4 + 2 * (i - 1) + d;
            A[row][d + 3 * n + 2 * i] = 1.0;
            A[row][d + 3 * n + 2 * (i + 1)] = -1.0;
            if (d == 0) {
                A[row][i - 1] = -l / 2 * sin(th[i - 1]);
                A[row][i] = -l / 2 * sin(th[i]);
                B[row] = l / 2 * (cos(th[i - 1]) * pow(thd[i - 1], 2.0) + cos(th[i]) * pow(thd[i], 2.0));
            } else {
                A[row][i - 1] = +l / 2 * cos(th[i - 1]);
                A[row][i] = +l / 2 * cos(th[i]);
                B[row] = l / 2 * (sin(th[i - 1]) * pow(thd[i - 1], 2.0) + sin(th[i]) * pow(thd[i], 2.0));
            }
        }
}

/* compute_accelerations (:139-226) */
int swt_accelerations(const swo_params *p, const double *state, const double *u,
                      double *gdd, double *tdd)
{
    const int n = p->n;
    if (n < 1 || n > NMAX) return -1;
    const int nn = 5 * n + 2;
    static _Thread_local double A[DIM][DIM];
    double B[DIM];
    twin_assemble(p, state, u, A, B);
    if (lu_solve(nn, A, B)) return 1;
    gdd[0] = gdd[1] = 0.0;
    for (int i = 1; i <= n; ++i) {                               /* :222-225 */
        tdd[i - 1] = B[i - 1];
        gdd[0] += 1.0 / n * B[3 * n + 2 * i];
        gdd[1] += 1.0 / n * B[3 * n + 2 * i + 1];
    }
    return 0;
}

/* The assembled system and its solution, row-major A [nn][nn], B [nn], X [nn] (nn = 5n+2):
 * what the reference's test driver prints (rlglue/test/acceleration-compare.txt:27-100). */
int swt_system(const swo_params *p, const double *state, const double *u, double *A_out,
               double *B_out, double *X_out)
{
    const int n = p->n;
    if (n < 1 || n > NMAX) return -1;
    const int nn = 5 * n + 2;
    static _Thread_local double A[DIM][DIM];
    double B[DIM];
    twin_assemble(p, state, u, A, B);
    for (int i = 0; i < nn; ++i) {
        memcpy(A_out + (size_t)i * nn, A[i], sizeof(double) * (size_t)nn);
        B_out[i] = B[i];
    }
    if (lu_solve(nn, A, B)) return 1;
    memcpy(X_out, B, sizeof(double) * (size_t)nn);
    return 0;
}

/* updateState (:102-137) with semi_implicit_euler (:228-236); reward = Gdot_new . direction */
int swt_step(const swo_params *p, const double *state, const double *u, double *next, double *reward)
{
    const int n = p->n;
    double gdd[2], tdd[NMAX], out[2 * NMAX + 2];
    int rc = swt_accelerations(p, state, u, gdd, tdd);
    if (rc) return rc;
    const double h = p->h;
    out[0] = state[0] + h * gdd[0];
    out[1] = state[1] + h * gdd[1];
    for (int i = 0; i < n; ++i) {
        const double thd_new = state[3 + 2 * i] + h * tdd[i];
        out[3 + 2 * i] = thd_new;
        out[2 + 2 * i] = state[2 + 2 * i] + h * thd_new;
    }
    memcpy(next, out, sizeof(double) * (size_t)(2 * n + 2));
    if (reward) *reward = out[0] * p->dir_x + out[1] * p->dir_y;
    return 0;
}

int swt_step_batch(const swo_params *p, long n_env, const double *states, const double *actions,
                   double *next, double *rewards)
{
    const int d = 2 * p->n + 2, m = p->n - 1;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (long e = 0; e < n_env; ++e)
        bad |= swt_step(p, states + e * d, actions + e * m, next + e * d, rewards + e);
    return bad;
}
