// swimmer_device.h -- per-swimmer fp64 device math for gfx950 (one swimmer per lane).
//
// What it computes is the reference's SwimmerEnv.compute_accelerations + explicit Euler
// (envs/gym_swimmer/swimmer/remy_swimmer_env.py:69-214).  HOW it computes it is not the
// reference's (n+2)x(n+3) affine-row build + dense LAPACK solve: the same equations are
// reduced analytically so that a lane needs ~120 fp64 operations for n = 3 instead of ~500,
// no dynamic indexing and no pivoting.
//
// Derivation (i, j, k are 1-based segment indices; s_i = sin th_i, c_i = cos th_i,
// n_i = (-s_i, c_i), cc_ij = cos(th_i - th_j), ss_ik = sin(th_k - th_i)):
//
//  * Rows 0-1 of the reference system (joint force at the free tail = 0, :193-194) sum the
//    segment force balances; the segment-centre accelerations relative to the barycentre
//    cancel, leaving  n m Gdd = sum_j F_j n_j  with the friction magnitudes
//    F_j = -k l (Gdot_j . n_j) (:182).  Gdd therefore decouples from thdd and the torques.
//  * Segment-centre velocity/acceleration in the head frame are prefix sums along the chain
//    (:128-152); subtracting their mean (:157-163) gives fixed weights
//        vw(j,k) = [k<j] + 1/2 [k=j] - wbar_k,   wbar_k = (n - k + 1/2) / n
//    so  Gdot_j . n_j = Gdot . n_j + l sum_k vw(j,k) thd_k cc_jk  =: g_j.
//  * Row 1+i (:196-205) is  (l/2) n_i . (f_i + f_{i-1}) - (m l^2/12) thdd_i
//    + k thd_i l^3/12 + u_{i-1} - u_i = 0.  With f_i the running sum of the segment
//    balances (:174-184) the thdd coefficients are  (m l^2 / 2) T(i,k) cc_ik  with
//        T(i,k) = W(i,k) + W(i-1,k),  W(i,k) = sum_{j<=i} vw(j,k)
//    (symmetric in i,k), so after scaling by -12/(m l^2) the system is
//        Q thdd = r,   Q_ik = -6 T(i,k) cc_ik + [i=k]      (symmetric positive definite)
//        r_i = -6 sum_{k!=i} T(i,k) thd_k^2 ss_ik
//              - (6k/m) sum_j A(i,j) g_j cc_ij + (k l/m) thd_i + 12/(m l^2) (u_{i-1} - u_i)
//        A(i,j) = (2i-1)/n - 2 [j<i] - [j=i]
//    and is solved by an unpivoted LDL^T (n = 3: closed-form adjugate).
//
// The reduction was checked against the reference's own outputs before any kernel was
// written (tests/golden/steps.npz, max relative difference 8e-15 over n = 2..8 and three
// parameter sets) and tests/test_hip_parity.py checks the compiled kernels the same way.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sw {

// Wave-uniform constants derived on the host from sw_params (live in SGPRs).
struct Consts {
    double l;        // l_i
    double h;        // Euler step
    double dirx, diry;
    double kl_nm;    // k l / (n m)
    double h_kl_nm;  // h k l / (n m): barycentre Euler update in one FMA (segment-per-lane kernels)
    double six_k_m;  // 6 k / m
    double kl_m;     // k l / m
    double c12;      // 12 / (m l^2)
};

// ---- compile-time chain weights ------------------------------------------------------
template <int N> __host__ __device__ constexpr double wbar(int k1) { return (N - k1 + 0.5) / N; }
template <int N> __host__ __device__ constexpr double vel_w(int j1, int k1)
{
    return (k1 < j1 ? 1.0 : (k1 == j1 ? 0.5 : 0.0)) - wbar<N>(k1);
}
template <int N> __host__ __device__ constexpr double Tw(int i1, int k1)
{
    return i1 < k1 ? -(2 * i1 - 1) * wbar<N>(k1)
                   : (i1 == k1 ? 0.5 - (2 * k1 - 1) * wbar<N>(k1)
                               : 2.0 * (i1 - k1) - (2 * i1 - 1) * wbar<N>(k1));
}
template <int N> __host__ __device__ constexpr double Aw(int i1, int j1)
{
    return (2 * i1 - 1) / (double)N - (j1 < i1 ? 2.0 : (j1 == i1 ? 1.0 : 0.0));
}

// ---- sin and cos ----------------------------------------------------------------------
// Cody-Waite reduction by pi/2 with two FMAs (pi/2 = H + M + ..., H = fl(pi/2); the first FMA
// is exact, the second rounds once relative to the *reduced* argument; the next term of the
// split is 1.5e-33 and would move the result by < 1.5e-24 |k| <= 5e-15 ulp(1) inside the valid
// range, so it is not spent) + the classic degree-13 /
// degree-14 minimax kernels on [-pi/4, pi/4] (coefficients: Sun fdlibm k_sin.c / k_cos.c,
// public domain).  Valid for |x| < kAngleLimit (quadrant index must fit 31 bits); callers
// flag larger / non-finite angles with SW_STATUS_RANGE instead of computing garbage.
//
// A lone wave issues ONE instruction (of any kind) per ~4.4 cycles (scripts/ubench), so the
// instruction count is the cost: there is no library fallback inside the hot loops (its
// constants alone cost SGPR spills there), and fma(p, z, C) with a loop-invariant VGPR
// constant C is requested in three-address form (hipcc otherwise emits
// `v_mov_b64 tmp, C; v_fmac_f64 tmp, p, z`, two slots).
constexpr double kAngleLimit = 3.0e9;  // < 2^31 * pi/2

__device__ __forceinline__ double fma3(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// a * b + coef with the coefficient read straight from a scalar register pair (one constant-bus
// operand per VALU instruction on gfx9): the polynomial coefficients never occupy VGPRs and the
// compiler has no reason to re-materialise them with v_mov inside an unrolled loop
__device__ __forceinline__ double fma3_sc(double a, double b, double coef)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(coef));
    return d;
}

// The leading coefficients of the two kernels: the first Horner step is fma(K, z, C) with two
// constants, and only one of them can come from a scalar register.  Rollout kernels create this
// pair once, before the step loop (trig_consts() makes the values opaque, so the compiler keeps
// them in VGPRs instead of re-materialising them with a v_mov per evaluation).
struct TrigK {
    double s6, c6;
};

__device__ __forceinline__ TrigK trig_consts()
{
    TrigK k{1.58969099521155010221e-10, -1.13596475577881948265e-11};
    asm volatile("" : "+v"(k.s6), "+v"(k.c6));
    return k;
}

__device__ __forceinline__ void sincos_fast(double x, double &s_out, double &c_out, const TrigK &K)
{
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52: integer part lands in the low bits
    const double kd_m = __builtin_fma(x, 0.63661977236758134308, MAGIC);
    const double kd = kd_m - MAGIC;
    const uint32_t q = (uint32_t)__double_as_longlong(kd_m);
    double r = __builtin_fma(-kd, 1.5707963267948966, x);       // exact
    r = __builtin_fma(-kd, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = fma3_sc(K.s6, z, -2.50507602534068634195e-08);
    ps = fma3_sc(ps, z, 2.75573137070700676789e-06);
    ps = fma3_sc(ps, z, -1.98412698298579493134e-04);
    ps = fma3_sc(ps, z, 8.33333333332248946124e-03);
    ps = fma3_sc(ps, z, -1.66666666666666324348e-01);
    double pc = fma3_sc(K.c6, z, 2.08757232129817482790e-09);
    pc = fma3_sc(pc, z, -2.75573143513906633035e-07);
    pc = fma3_sc(pc, z, 2.48015872894767294178e-05);
    pc = fma3_sc(pc, z, -1.38888888888741095749e-03);
    pc = fma3_sc(pc, z, 4.16666666666666019037e-02);
    const double sr = __builtin_fma(r * z, ps, r);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    // quadrant: bit0 of q swaps sin/cos, bit1 negates sin, bit1 ^ bit0 (= (q+1)&2) negates cos
    const uint32_t w = q << 30, t31 = q << 31;
    const bool swap = t31 != 0u;
    const double sv = swap ? cr : sr;
    const double cv = swap ? sr : cr;
    const uint32_t sfl = w & 0x80000000u, cfl = (w ^ t31) & 0x80000000u;
    s_out = __hiloint2double((int)((uint32_t)__double2hiint(sv) ^ sfl), __double2loint(sv));
    c_out = __hiloint2double((int)((uint32_t)__double2hiint(cv) ^ cfl), __double2loint(cv));
}

// ---- angles kept in reduced form (segment-per-lane rollout kernels) -----------------------
// theta = r + K (pi/2) with |r| <= pi/4: the rollout integrates r (r += h thetadot, the same
// FMA the Euler update of theta is) and re-normalises -- rarely, on a wave-uniform branch -- when
// |r| leaves [-pi/4, pi/4].  sin / cos then need NO per-step range reduction and NO quadrant
// logic: the two minimax kernels on r and a rotation by the quarter turns (sa, sb) =
// (cos, sin)(K pi/2), exact numbers in {0, +-1}: 4 instructions instead of 16.  The reduced angle
// is the authoritative one; theta = fl(K pi/2 + r) is formed from it for the policy, the
// trajectory and the statistics (one FMA, within an ulp of the angle the sin / cos are taken of).
// Where the reference rounds theta to ulp(theta) every step, r is rounded to ulp(r) <= ulp(theta):
// same equations, rounding differences of the order the reference's own step has.
struct Angle {
    double r;       // reduced angle
    double kd;      // K, an exact integer
    double sa, sb;  // cos(K pi/2), sin(K pi/2)
};

constexpr double kPio2Hi = 1.5707963267948966;      // fl(pi/2)
constexpr double kPio2Lo = 6.123233995736766e-17;   // pi/2 - fl(pi/2)
constexpr double kPio4 = 0.78539816339744830962;

// Move whole quarter turns from r into (K, sa, sb); a no-op on lanes with |r| <= pi/4.  Valid for
// |r| < kAngleLimit (callers flag larger start angles with SW_STATUS_RANGE).
__device__ __forceinline__ void angle_renorm(Angle &A)
{
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double km = __builtin_fma(A.r, 0.63661977236758134308, MAGIC);
    const double k = km - MAGIC;
    const uint32_t q = (uint32_t)__double_as_longlong(km);
    double r = __builtin_fma(-k, kPio2Hi, A.r);      // exact
    A.r = __builtin_fma(-k, kPio2Lo, r);
    A.kd += k;
    const double ck = (q & 1u) ? 0.0 : ((q & 2u) ? -1.0 : 1.0);
    const double sk = (q & 1u) ? ((q & 2u) ? -1.0 : 1.0) : 0.0;
    const double sa = A.sa * ck - A.sb * sk, sb = A.sa * sk + A.sb * ck;
    A.sa = sa;
    A.sb = sb;
}

// The per-step form of angle_renorm for the quad kernel's hot loop: a vector compare, a scalar compare
// and ONE (normally not taken) scalar branch; the re-normalisation behind it works IN PLACE on A and thmax.  Written as an asm
// block because the compiler lays the same C++ out with the rare path as the fall-through and
// five register copies on the common one.  No lane predication: lanes inside [-pi/4, pi/4] move
// k = 0 quarter turns.  thmax <- max(thmax, |theta|) on that path: every way to a huge angle
// leads through it (|theta| moves less than pi/2 between two re-normalisations).
// magic = 1.5 * 2^52 in a VGPR pair (VOP3 takes one scalar operand on gfx9).
// Compare and branch are TWO statements (angle_range_test / angle_keep_reduced) with the lane mask in a
// scalar register pair between them: the caller puts independent work between the vector compare and
// the scalar branch that waits for its result (back to back the pair costs ~3 % of a rollout launch
// beyond its issue slots, profiles/r03_g_ab_range_check_variants.log).
__device__ __forceinline__ unsigned long long angle_range_test(double r)
{
    unsigned long long mask;
    asm volatile("v_cmp_gt_f64_e64 %[m], |%[r]|, %[lim]" : [m] "=s"(mask) : [r] "v"(r), [lim] "s"(kPio4));
    return mask;
}

__device__ __forceinline__ void angle_keep_reduced(Angle &A, double &thmax, double magic, unsigned long long outside)
{
    double t0, t1, p, w;
    int q, q1;
    asm volatile(
        "s_cmp_lg_u64 %[m], 0\n\t"
        // the rare path lives out of line (end of this function's section): the common path
        // falls through an UNTAKEN branch
        "s_cbranch_scc1 .Lsw_renorm_%=\n"
        ".Lsw_reduced_%=:\n\t"
        ".subsection 1\n"
        ".Lsw_renorm_%=:\n\t"
        "v_fma_f64 %[t1], %[r], %[c2opi], %[magic]\n\t"     // k + magic
        "v_add_f64 %[t0], %[t1], -%[magic]\n\t"             // k = rint(r * 2/pi)
        "v_fma_f64 %[r], -%[t0], %[hi], %[r]\n\t"           // exact
        "v_fma_f64 %[r], -%[t0], %[lo], %[r]\n\t"
        "v_add_f64 %[kd], %[kd], %[t0]\n\t"
        "v_cvt_i32_f64_e32 %[q], %[t0]\n\t"
        "v_and_b32_e32 %[q1], 1, %[q]\n\t"
        "v_and_b32_e32 %[q], 2, %[q]\n\t"
        "v_cvt_f64_i32_e32 %[t0], %[q]\n\t"                 // 0 or 2
        "v_add_f64 %[t0], 1.0, -%[t0]\n\t"                  // u = cos / sin of the even part
        "v_cvt_f64_i32_e32 %[t1], %[q1]\n\t"                // odd: 0 or 1
        "v_mul_f64 %[p], %[sa], %[t0]\n\t"
        "v_mul_f64 %[w], %[sb], %[t0]\n\t"
        "v_add_f64 %[t0], 1.0, -%[t1]\n\t"                  // even: 1 or 0
        "v_mul_f64 %[sa], %[t0], %[p]\n\t"
        "v_mul_f64 %[sb], %[t0], %[w]\n\t"
        "v_fma_f64 %[sa], -%[t1], %[w], %[sa]\n\t"          // odd: (sa, sb) <- (-sb u, sa u)
        "v_fma_f64 %[sb], %[t1], %[p], %[sb]\n\t"
        "v_fma_f64 %[t0], %[kd], %[hi], %[r]\n\t"
        "v_max_f64 %[thmax], %[thmax], |%[t0]|\n\t"
        "s_branch .Lsw_reduced_%=\n\t"
        ".subsection 0"
        : [r] "+v"(A.r), [kd] "+v"(A.kd), [sa] "+v"(A.sa), [sb] "+v"(A.sb), [thmax] "+v"(thmax),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [p] "=&v"(p), [w] "=&v"(w), [q] "=&v"(q), [q1] "=&v"(q1)
        : [c2opi] "s"(0.63661977236758134308), [hi] "s"(kPio2Hi),
          [lo] "s"(kPio2Lo), [magic] "v"(magic), [m] "s"(outside)
        : "scc");
}

__device__ __forceinline__ Angle angle_make(double theta)
{
    Angle A{theta, 0.0, 1.0, 0.0};
    angle_renorm(A);
    return A;
}

__device__ __forceinline__ double angle_theta(const Angle &A)
{
    return __builtin_fma(A.kd, kPio2Hi, A.r);
}

__device__ __forceinline__ void sincos_angle(const Angle &A, double &s_out, double &c_out, const TrigK &K)
{
    const double r = A.r, z = r * r;
    double ps = fma3_sc(K.s6, z, -2.50507602534068634195e-08);
    ps = fma3_sc(ps, z, 2.75573137070700676789e-06);
    ps = fma3_sc(ps, z, -1.98412698298579493134e-04);
    ps = fma3_sc(ps, z, 8.33333333332248946124e-03);
    ps = fma3_sc(ps, z, -1.66666666666666324348e-01);
    double pc = fma3_sc(K.c6, z, 2.08757232129817482790e-09);
    pc = fma3_sc(pc, z, -2.75573143513906633035e-07);
    pc = fma3_sc(pc, z, 2.48015872894767294178e-05);
    pc = fma3_sc(pc, z, -1.38888888888741095749e-03);
    pc = fma3_sc(pc, z, 4.16666666666666019037e-02);
    const double sr = __builtin_fma(r * z, ps, r);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    s_out = __builtin_fma(A.sa, sr, A.sb * cr);      // sin(r + K pi/2)
    c_out = __builtin_fma(A.sa, cr, -(A.sb * sr));   // cos(r + K pi/2)
}

__device__ __forceinline__ void sincos_fast(double x, double &s_out, double &c_out)
{
    sincos_fast(x, s_out, c_out, TrigK{1.58969099521155010221e-10, -1.13596475577881948265e-11});
}

// max(acc, |theta_i|): one v_max_f64 per angle; NaN angles are caught by the finiteness
// check of the final state instead (v_max drops a NaN operand).
template <int N>
__device__ __forceinline__ double track_angle_range(double acc, const double (&th)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i)   // plain v_max_f64 (the builtin adds a canonicalising max)
        asm("v_max_f64 %0, %1, |%2|" : "=v"(acc) : "v"(acc), "v"(th[i]));
    return acc;
}

// 1/x for a well-scaled x: hardware estimate + two Newton steps (full fp64).  The second step
// reuses the squared residual (r1 = r0 (1 + e) has residual e^2) instead of recomputing it from
// x: same five instructions, one level less on the dependency chain that ends every step.
__device__ __forceinline__ double rcp_f64(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r0, 1.0);
    const double r1 = __builtin_fma(r0, e, r0);
    const double e2 = e * e;
    return __builtin_fma(r1, e2, r1);
}

// 1/x from the hardware estimate + ONE Newton step: relative error <= 2.2e-15 (measured over
// 2^20 arguments, scripts/ubench/store_rcp.hip; the raw estimate is good to 4.6e-8).  Used where
// the reciprocal sits on the serial pivot chain of a segment-per-lane solve and 20 ulp in
// thetaddot (h * 2e-15 * |thetaddot| ~ 1e-17 in thetadot per step) is far inside the rounding
// noise of the step.
__device__ __forceinline__ double rcp_f64_1n(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r0, 1.0);
    return __builtin_fma(r0, e, r0);
}

// ---- symmetric positive definite solve ---------------------------------------------
// Q x = b, Q symmetric (only i >= j read), unpivoted LDL^T fully unrolled.
// Returns false when a pivot is not positive / not finite.
template <int N>
__device__ __forceinline__ bool spd_solve(double (&Q)[N][N], double (&b)[N])
{
    double D[N], rD[N];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double v[N];
        double dj = Q[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) {
            v[k] = Q[j][k] * D[k];
            dj = __builtin_fma(-Q[j][k], v[k], dj);
        }
        D[j] = dj;
        ok = ok && (dj > 0.0) && (dj < 1.0e300);
        rD[j] = rcp_f64(dj);
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double a = Q[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) a = __builtin_fma(-Q[i][k], v[k], a);
            Q[i][j] = a * rD[j];
        }
    }
#pragma unroll
    for (int i = 1; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) b[i] = __builtin_fma(-Q[i][k], b[k], b[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] *= rD[i];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) {
#pragma unroll
        for (int k = i + 1; k < N; ++k) b[i] = __builtin_fma(-Q[k][i], b[k], b[i]);
    }
    return ok;
}

// n = 3: closed-form adjugate (one reciprocal, short dependency chain).
template <>
__device__ __forceinline__ bool spd_solve<3>(double (&Q)[3][3], double (&b)[3])
{
    const double a = Q[0][0], bb = Q[1][0], c = Q[2][0], d = Q[1][1], e = Q[2][1], f = Q[2][2];
    const double c00 = __builtin_fma(d, f, -e * e);
    const double c01 = __builtin_fma(c, e, -bb * f);
    const double c02 = __builtin_fma(bb, e, -c * d);
    const double c11 = __builtin_fma(a, f, -c * c);
    const double c12 = __builtin_fma(bb, c, -a * e);
    const double c22 = __builtin_fma(a, d, -bb * bb);
    const double det = __builtin_fma(a, c00, __builtin_fma(bb, c01, c * c02));
    const bool ok = (det > 0.0) && (det < 1.0e300) && (a > 0.0) && (c22 > 0.0);
    const double rdet = rcp_f64(det);
    const double x0 = __builtin_fma(c00, b[0], __builtin_fma(c01, b[1], c02 * b[2]));
    const double x1 = __builtin_fma(c01, b[0], __builtin_fma(c11, b[1], c12 * b[2]));
    const double x2 = __builtin_fma(c02, b[0], __builtin_fma(c12, b[1], c22 * b[2]));
    b[0] = x0 * rdet;
    b[1] = x1 * rdet;
    b[2] = x2 * rdet;
    return ok;
}

template <>
__device__ __forceinline__ bool spd_solve<2>(double (&Q)[2][2], double (&b)[2])
{
    const double a = Q[0][0], bb = Q[1][0], d = Q[1][1];
    const double det = __builtin_fma(a, d, -bb * bb);
    const bool ok = (det > 0.0) && (det < 1.0e300) && (a > 0.0);
    const double rdet = rcp_f64(det);
    const double x0 = __builtin_fma(d, b[0], -bb * b[1]);
    const double x1 = __builtin_fma(a, b[1], -bb * b[0]);
    b[0] = x0 * rdet;
    b[1] = x1 * rdet;
    return ok;
}

// ---- accelerations -------------------------------------------------------------------
// u[] holds the N-1 joint torques.  Outputs Gdd and thdd; returns false on a bad pivot.
template <int N>
__device__ __forceinline__ bool accelerations(const Consts &C, double gdx, double gdy,
                                              const double (&th)[N], const double (&thd)[N],
                                              const double (&u)[N > 1 ? N - 1 : 1],
                                              double &gddx, double &gddy, double (&tdd)[N])
{
    double s[N], c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) sincos_fast(th[i], s[i], c[i]);

    double cc[N][N], ss[N][N];  // cc[i][j] = cos(th_i - th_j), ss[i][k] = sin(th_k - th_i)
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

    double lthd[N], thd2[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        lthd[k] = C.l * thd[k];
        thd2[k] = thd[k] * thd[k];
    }

    // g_j = Gdot_j . n_j  (normal velocity of segment j's centre, :180-182)
    double g[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double a = __builtin_fma(gdy, c[j], -gdx * s[j]);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double w = vel_w<N>(j + 1, k + 1);
            a = __builtin_fma(w * cc[j][k], lthd[k], a);
        }
        g[j] = a;
    }

    // barycentre acceleration: n m Gdd = sum_j F_j n_j, F_j = -k l g_j
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        sx = __builtin_fma(g[j], s[j], sx);
        sy = __builtin_fma(g[j], c[j], sy);
    }
    gddx = C.kl_nm * sx;
    gddy = -C.kl_nm * sy;

    // right-hand side and matrix of Q thdd = r
    double Q[N][N], r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double cent = 0.0, fric = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k != i) cent = __builtin_fma((-6.0 * Tw<N>(i + 1, k + 1)) * thd2[k], ss[i][k], cent);
            fric = __builtin_fma(Aw<N>(i + 1, k + 1) * cc[i][k], g[k], fric);
        }
        double tq = 0.0;
        if (i >= 1) tq += u[i - 1];
        if (i < N - 1) tq -= u[i];
        double ri = __builtin_fma(-C.six_k_m, fric, cent);
        ri = __builtin_fma(C.kl_m, thd[i], ri);
        ri = __builtin_fma(C.c12, tq, ri);
        r[i] = ri;
#pragma unroll
        for (int k = 0; k <= i; ++k)
            Q[i][k] = (k == i) ? (-6.0 * Tw<N>(i + 1, i + 1) + 1.0)
                               : (-6.0 * Tw<N>(i + 1, k + 1)) * cc[i][k];
    }
    const bool ok = spd_solve<N>(Q, r);
#pragma unroll
    for (int i = 0; i < N; ++i) tdd[i] = r[i];
    return ok;
}

// ---- one explicit-Euler step (remy_swimmer_env.py:87-91), in place -------------------
// reward = Gdot_new . direction (:243).
template <int N>
__device__ __forceinline__ bool euler_step(const Consts &C, double &gdx, double &gdy,
                                           double (&th)[N], double (&thd)[N],
                                           const double (&u)[N > 1 ? N - 1 : 1], double &reward)
{
    double gddx, gddy, tdd[N];
    const bool ok = accelerations<N>(C, gdx, gdy, th, thd, u, gddx, gddy, tdd);
    gdx = __builtin_fma(C.h, gddx, gdx);
    gdy = __builtin_fma(C.h, gddy, gdy);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        th[i] = __builtin_fma(C.h, thd[i], th[i]);    // old theta_dot
        thd[i] = __builtin_fma(C.h, tdd[i], thd[i]);
    }
    reward = __builtin_fma(gdx, C.dirx, gdy * C.diry);
    return ok;
}

}  // namespace sw
