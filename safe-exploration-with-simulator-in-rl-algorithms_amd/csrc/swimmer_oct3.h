// swimmer_oct3.h -- the 3-segment rollout step with LANE ROLES: two mirror quads per rollout.
//
// swimmer_quad3.h spreads a rollout over one DPP quad (lane q = segment q) and every lane evaluates
// BOTH minimax polynomials of its angle (sin and cos: 20 of the step's 124 instructions) and BOTH
// barycentre sums.  Here a rollout owns two quads of a 16-lane DPP row, eight lanes apart:
//
//   quad A (row lanes 0-3 / 4-7)    evaluates the SINE of its segment's reduced angle and carries Gdot_x
//   quad B (row lanes 8-11 / 12-15) evaluates the COSINE                              and carries Gdot_y
//
// with the SAME instruction stream: one Horner chain whose seven coefficients sit in per-lane
// registers (sine set on A, cosine set on B), and `v_mov_b32_dpp row_ror:8` hands each lane its
// partner's value (two moves per double).  Everything else of the step is computed by both quads,
// each on its own copy of the state (they differ by rounding only, like the quad kernel's per-lane
// Gdot copies), written in terms of
//        u = the value this lane evaluated      (A: sin theta_i,  B: cos theta_i)
//        v = its partner's                      (A: cos theta_i,  B: sin theta_i)
//        Pu = the Gdot component this quad integrates (A: x, B: y),  Pv = the partner's
// so that no lane-dependent code is needed: cos(th_i - th_k) = u u_k + v v_k is symmetric in the roles;
// sin(th_k - th_i), the segment's normal velocity and everything linear in it change SIGN on B, and
// that sign lives in the per-lane constants (OctLane).  The barycentre update comes out the same
// on both quads:  Pu += (h k l / (n m)) sum_j g~_j u_j.
//
// Quarter turns cost nothing per step either.  The angle is carried reduced, theta = r + K pi/2, and
//        sin(r + K pi/2) = +sin r, +cos r, -sin r, -cos r   for K mod 4 = 0, 1, 2, 3
//        cos(theta) = sin(theta + pi/2): the same table read at K + 1,
// so a lane's DESIGNATED output (A: sin theta, B: cos theta) is one of the two polynomials of r with
// a sign: the lane's coefficient set and the sign (in selS / selC) are state, re-chosen only when K
// changes -- in the rare re-normalisation block behind one vector compare, one scalar compare and one
// untaken branch per step (oct3_range_test / oct3_keep_reduced), taken right after the angle update
// and before the polynomial is evaluated: exact for any angular velocity.  (The row kernel, which
// cannot afford the asm block's registers at n >= 6, checks once per trip of four steps instead and
// runs trips whose angles travel more than kTripSlack = 0.04 rad in a loop that checks inside every
// step; up to pi/4 + 0.04 the polynomials are accurate to 2.5e-16, 1.2e-16 inside, checked against
// long-double libm.)
//
// Per step: 10 instructions of sin/cos + 2 moves instead of 20 + 4 of rotation, one barycentre sum
// instead of two, no per-lane select for the recorded Gdot component: 113 instead of 124
// (scripts/isa_loop_stats.py; SQ counters 112.7).  The rollout's trajectory is quad A's copy of (theta, thetadot,
// Gdot_x) and quad B's Gdot_y.
//
// Same equations as swimmer_device.h (derivation there).  sin(r) = r + r z p(z) is the fdlibm form;
// cos(r) = 1 + z q(z) folds fdlibm's  1 - z/2 + z^2 c(z)  into one Horner chain with a single
// final rounding at magnitude 1.
#pragma once

#include "swimmer_quad3.h"

namespace sw {

// Row kernel (swimmer_row.h): what an angle may move during an unchecked trip of four steps -- the
// polynomials are evaluated at most this far (plus what thetadot gains within the trip) outside
// [-pi/4, pi/4], where they are accurate to 2.5e-16.  Trips with a faster lane check inside every step.
constexpr double kTripSlack = 0.04;

constexpr int kDppRowRor8 = 0x128;   // dpp_ctrl row_ror:8 -- lane L of a 16-lane row reads lane (L + 8) % 16

template <int CTRL>
__device__ __forceinline__ double dpp_row_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Per-lane constants: Quad3Lane's with the role sign sigma (+1 on quad A, -1 on quad B) folded in
// where the quantity is odd in the roles.
struct OctLane {
    double kvw0, kvw1, kvw2;   // sigma l vel_w(i, .)
    double ka0, ka1, ka2;      // sigma Aw(i, .)
    double kt1, kt2;           // sigma (-6 T(i, i1)), sigma (-6 T(i, i2))
    double t1, t2, t12;        // unsigned: the matrix entries
    double d0, d12, d1, d2;
};

// What a lane evaluates of its reduced angle: X (1 + z p(z)), z = r^2, X = selS r + selC.
// Sine type: p = S1 + z (S2 + ... z S6), (selS, selC) = (sign, 0); cosine type: p = -1/2 + z (C1 +
// ... z C6), (selS, selC) = (0, sign).  fdlibm k_sin.c / k_cos.c coefficients (public domain);
// sin r = r + r z p(z) is the fdlibm form, cos r = 1 + z q(z) folds fdlibm's 1 - z/2 + z^2 c(z)
// into one Horner chain with a single final rounding at magnitude 1.
struct OctTrig {
    double r, kd;      // theta = r + kd pi/2, kd an exact integer
    double k[7];
    double selS, selC;
};

// (Re)choose type and sign from K = kd and the lane's designation (0: sin theta, 1: cos theta).
__device__ __forceinline__ void oct3_retype(OctTrig &T, int designation)
{
    const double S[7] = {0.0, 1.58969099521155010221e-10, -2.50507602534068634195e-08,
                         2.75573137070700676789e-06, -1.98412698298579493134e-04,
                         8.33333333332248946124e-03, -1.66666666666666324348e-01};
    const double Cc[7] = {-1.13596475577881948265e-11, 2.08757232129817482790e-09,
                          -2.75573143513906633035e-07, 2.48015872894767294178e-05,
                          -1.38888888888741095749e-03, 4.16666666666666019037e-02, -0.5};
    const int q = (int)T.kd + designation;      // |kd| < 2^31 inside the valid angle range
    const bool cos_type = (q & 1) != 0;
    const double sign = (q & 2) ? -1.0 : 1.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) T.k[j] = cos_type ? Cc[j] : S[j];
    T.selS = cos_type ? 0.0 : sign;
    T.selC = cos_type ? sign : 0.0;
}

// Move whole quarter turns from r into kd and re-choose the polynomial (all lanes; a no-op where
// |r| <= pi/4).  thmax <- max(thmax, |theta|): every way to a huge angle leads through here.
__device__ __forceinline__ void oct3_renorm(OctTrig &T, int designation, double &thmax)
{
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double km = __builtin_fma(T.r, 0.63661977236758134308, MAGIC);
    const double k = km - MAGIC;
    const double r = __builtin_fma(-k, kPio2Hi, T.r);      // exact
    T.r = __builtin_fma(-k, kPio2Lo, r);
    T.kd += k;
    oct3_retype(T, designation);
    thmax = fmax(thmax, fabs(__builtin_fma(T.kd, kPio2Hi, T.r)));
}

// The per-step form of oct3_renorm for the hot loop: ONE compare and ONE (normally not taken) scalar
// branch; the re-normalisation AND the re-choice of the polynomial behind it work in place on T and
// thmax, out of line (end of the function's section: the common path falls through an untaken branch;
// written as asm because the compiler lays the same C++ out with the rare path inline and register
// copies on the common one -- measured +12 % on the launch).  No lane predication: lanes inside
// [-pi/4, pi/4] move k = 0 quarter turns and re-choose what they had.  The coefficient sets come in
// as scalar registers; a lane's set is type * C_j + (1 - type) * S_j with type in {0, 1}: exact.
// Checked after every angle update, before the polynomial is evaluated: exact for ANY angular velocity.
// The compare and the branch are two statements (oct3_range_test / oct3_keep_reduced) with the lane
// mask in a scalar register pair between them, so that the caller can put independent work between
// the vector compare and the scalar branch that waits for its result (back to back the pair cost
// ~2.7 % of the launch beyond its two issue slots).
__device__ __forceinline__ unsigned long long oct3_range_test(double r)
{
    unsigned long long mask;
    asm volatile("v_cmp_gt_f64_e64 %[m], |%[r]|, %[lim]" : [m] "=s"(mask) : [r] "v"(r), [lim] "s"(kPio4));
    return mask;
}

__device__ __forceinline__ void oct3_keep_reduced(OctTrig &T, double &thmax, double magic, int designation,
                                                  unsigned long long outside)
{
    double t0, t1, t2;
    int q, q1;
    asm volatile(
        "s_cmp_lg_u64 %[m], 0\n\t"
        "s_cbranch_scc1 .Lsw_oct_renorm_%=\n"
        ".Lsw_oct_reduced_%=:\n\t"
        ".subsection 1\n"
        ".Lsw_oct_renorm_%=:\n\t"
        "v_fma_f64 %[t1], %[r], %[c2opi], %[magic]\n\t"     // k + magic
        "v_add_f64 %[t0], %[t1], -%[magic]\n\t"             // k = rint(r * 2/pi)
        "v_fma_f64 %[r], -%[t0], %[hi], %[r]\n\t"           // exact
        "v_fma_f64 %[r], -%[t0], %[lo], %[r]\n\t"
        "v_add_f64 %[kd], %[kd], %[t0]\n\t"
        "v_cvt_i32_f64_e32 %[q], %[kd]\n\t"                 // K (|K| < 2^31 inside the valid range)
        "v_add_u32_e32 %[q], %[q], %[des]\n\t"              // cos(theta) = sin(theta + pi/2): K + 1
        "v_and_b32_e32 %[q1], 1, %[q]\n\t"                  // type: 1 = cosine polynomial
        "v_and_b32_e32 %[q], 2, %[q]\n\t"
        "v_cvt_f64_i32_e32 %[t0], %[q]\n\t"                 // 0 or 2
        "v_add_f64 %[t0], 1.0, -%[t0]\n\t"                  // sign
        "v_cvt_f64_i32_e32 %[t1], %[q1]\n\t"                // type as 0.0 / 1.0
        "v_mul_f64 %[selC], %[t1], %[t0]\n\t"               // type * sign
        "v_add_f64 %[selS], %[t0], -%[selC]\n\t"            // (1 - type) * sign
        "v_add_f64 %[t0], 1.0, -%[t1]\n\t"                  // 1 - type
        "v_mul_f64 %[t2], %[t0], %[s0]\n\t"
        "v_fma_f64 %[k0], %[t1], %[c0], %[t2]\n\t"
        "v_mul_f64 %[t2], %[t0], %[s1]\n\t"
        "v_fma_f64 %[k1], %[t1], %[c1], %[t2]\n\t"
        "v_mul_f64 %[t2], %[t0], %[s2]\n\t"
        "v_fma_f64 %[k2], %[t1], %[c2], %[t2]\n\t"
        "v_mul_f64 %[t2], %[t0], %[s3]\n\t"
        "v_fma_f64 %[k3], %[t1], %[c3], %[t2]\n\t"
        "v_mul_f64 %[t2], %[t0], %[s4]\n\t"
        "v_fma_f64 %[k4], %[t1], %[c4], %[t2]\n\t"
        "v_mul_f64 %[t2], %[t0], %[s5]\n\t"
        "v_fma_f64 %[k5], %[t1], %[c5], %[t2]\n\t"
        "v_mul_f64 %[t2], %[t0], %[s6]\n\t"
        "v_fma_f64 %[k6], %[t1], %[c6], %[t2]\n\t"
        "v_fma_f64 %[t0], %[kd], %[hi], %[r]\n\t"
        "v_max_f64 %[thmax], %[thmax], |%[t0]|\n\t"
        "s_branch .Lsw_oct_reduced_%=\n\t"
        ".subsection 0"
        : [r] "+v"(T.r), [kd] "+v"(T.kd), [selS] "+v"(T.selS), [selC] "+v"(T.selC), [thmax] "+v"(thmax),
          [k0] "+v"(T.k[0]), [k1] "+v"(T.k[1]), [k2] "+v"(T.k[2]), [k3] "+v"(T.k[3]), [k4] "+v"(T.k[4]),
          [k5] "+v"(T.k[5]), [k6] "+v"(T.k[6]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [q] "=&v"(q), [q1] "=&v"(q1)
        : [lim] "s"(kPio4), [c2opi] "s"(0.63661977236758134308), [hi] "s"(kPio2Hi), [lo] "s"(kPio2Lo),
          [magic] "v"(magic), [des] "v"(designation), [m] "s"(outside),
          [s0] "s"(0.0), [s1] "s"(1.58969099521155010221e-10), [s2] "s"(-2.50507602534068634195e-08),
          [s3] "s"(2.75573137070700676789e-06), [s4] "s"(-1.98412698298579493134e-04),
          [s5] "s"(8.33333333332248946124e-03), [s6] "s"(-1.66666666666666324348e-01),
          [c0] "s"(-1.13596475577881948265e-11), [c1] "s"(2.08757232129817482790e-09),
          [c2] "s"(-2.75573143513906633035e-07), [c3] "s"(2.48015872894767294178e-05),
          [c4] "s"(-1.38888888888741095749e-03), [c5] "s"(4.16666666666666019037e-02), [c6] "s"(-0.5)
        : "scc");
}

__device__ __forceinline__ OctLane oct3_lane(const Consts &C, int seg, bool cosine)
{
    const Quad3Lane L = quad3_lane(seg);
    const double sg = cosine ? -1.0 : 1.0;
    OctLane O;
    O.kvw0 = sg * (L.vw0 * C.l);
    O.kvw1 = sg * (L.vw1 * C.l);
    O.kvw2 = sg * (L.vw2 * C.l);
    O.ka0 = sg * L.a0;
    O.ka1 = sg * L.a1;
    O.ka2 = sg * L.a2;
    O.kt1 = sg * L.t1;
    O.kt2 = sg * L.t2;
    O.t1 = L.t1;
    O.t2 = L.t2;
    O.t12 = L.t12;
    O.d0 = L.d0;
    O.d12 = L.d12;
    O.d1 = L.d1;
    O.d2 = L.d2;
    return O;
}

struct OctGeo {
    double u, v, u1, v1, u2, v2;
    double cc1, cc2, cc12;   // cos(th_i - th_i1), cos(th_i - th_i2), cos(th_i1 - th_i2)
    double f1, f2;           // sigma sin(th_i1 - th_i), sigma sin(th_i2 - th_i)
};

// Everything of a step that depends on the angles only.
__device__ __forceinline__ OctGeo oct3_geometry(const OctTrig &T)
{
    OctGeo G;
    const double r = T.r, z = r * r;
    double p = fma3(T.k[0], z, T.k[1]);
    p = fma3(p, z, T.k[2]);
    p = fma3(p, z, T.k[3]);
    p = fma3(p, z, T.k[4]);
    p = fma3(p, z, T.k[5]);
    p = fma3(p, z, T.k[6]);
    const double X = __builtin_fma(T.selS, r, T.selC);
    G.u = __builtin_fma(X * z, p, X);                  // sin theta on A, cos theta on B
    G.v = dpp_row_f64<kDppRowRor8>(G.u);               // cos theta on A, sin theta on B
    G.u1 = dpp_f64<kDppNext1>(G.u);
    G.v1 = dpp_f64<kDppNext1>(G.v);
    G.u2 = dpp_f64<kDppNext2>(G.u);
    G.v2 = dpp_f64<kDppNext2>(G.v);
    G.cc1 = __builtin_fma(G.u, G.u1, G.v * G.v1);
    G.cc2 = __builtin_fma(G.u, G.u2, G.v * G.v2);
    G.cc12 = __builtin_fma(G.u1, G.u2, G.v1 * G.v2);
    G.f1 = __builtin_fma(G.v, G.u1, -G.u * G.v1);
    G.f2 = __builtin_fma(G.v, G.u2, -G.u * G.v2);
    return G;
}

// The velocity-dependent part of one explicit-Euler step: updates Pu (this quad's Gdot component) and
// thd; the caller exchanges Pv afterwards.  Returns det for the singularity check.
__device__ __forceinline__ double oct3_dynamics(const Consts &C, const OctLane &O, const OctGeo &G,
                                                double &Pu, double Pv, double &thd, double w1, double w2,
                                                double tq_scaled)
{
    // g~ = sigma (normal velocity of this segment's centre)
    double g = __builtin_fma(Pv, G.v, -Pu * G.u);
    g = __builtin_fma(O.kvw0, thd, g);
    g = __builtin_fma(O.kvw1 * G.cc1, w1, g);
    g = __builtin_fma(O.kvw2 * G.cc2, w2, g);
    const double g1 = dpp_f64<kDppNext1>(g), g2 = dpp_f64<kDppNext2>(g);
    // this quad's barycentre sum: A: sum g_j sin th_j, B: -sum g_j cos th_j
    const double su = __builtin_fma(g2, G.u2, __builtin_fma(g1, G.u1, g * G.u));
    double cent = __builtin_fma(O.kt1 * (w1 * w1), G.f1, tq_scaled);
    cent = __builtin_fma(O.kt2 * (w2 * w2), G.f2, cent);
    double fric = O.ka0 * g;
    fric = __builtin_fma(O.ka1 * G.cc1, g1, fric);
    fric = __builtin_fma(O.ka2 * G.cc2, g2, fric);
    double r0 = __builtin_fma(-C.six_k_m, fric, cent);
    r0 = __builtin_fma(C.kl_m, thd, r0);
    const double r1 = dpp_f64<kDppNext1>(r0), r2 = dpp_f64<kDppNext2>(r0);
    const double a = O.t1 * G.cc1, b = O.t2 * G.cc2, e = O.t12 * G.cc12;
    const double c00 = __builtin_fma(-e, e, O.d12);
    const double c01 = __builtin_fma(b, e, -a * O.d2);
    const double c02 = __builtin_fma(a, e, -b * O.d1);
    const double det = __builtin_fma(O.d0, c00, __builtin_fma(a, c01, b * c02));
    const double num = __builtin_fma(c00, r0, __builtin_fma(c01, r1, c02 * r2));
    const double tdd = num * rcp_f64_1n(det);
    // A: Gdot_x += (h k l / (n m)) sum g_j sin th_j;  B: Gdot_y -= ... sum g_j cos th_j: the same FMA
    Pu = __builtin_fma(C.h_kl_nm, su, Pu);
    thd = __builtin_fma(C.h, tdd, thd);
    return det;
}

}  // namespace sw
