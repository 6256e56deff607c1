"""How the device's sin / cos polynomials (fdlibm's minimax kernels for |r| <= pi/4, csrc/swimmer_oct3.h OctTrig /
swimmer_device.h) behave OUTSIDE pi/4: the segment-per-lane kernels carry reduced angles and the row kernel
(n = 4 ... 8) re-normalises once per trip of four steps, so r may overshoot pi/4 by what an angle travels in a trip:
kTripSlack = 0.04 rad by the check at the trip's start, plus what thetadot gains inside the trip.  The
polynomials are restated here in NumPy (same coefficients, same Horner order) and compared with long-double libm:
the error grows smoothly, so an overshoot of a few hundredths of a radian costs rounding-level accuracy only."""
import numpy as np

S = [0.0, 1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,
     -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01]
C = [-1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,
     2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02, -0.5]


def _poly(k, r, X):
    z = r * r
    p = k[0] * z + k[1]
    for c in k[2:]:
        p = p * z + c
    return (X * z) * p + X            # sin: X = r;  cos: X = 1


def _worst(lo, hi):
    r = np.linspace(lo, hi, 200001)
    rl = r.astype(np.longdouble)
    es = np.abs(_poly(S, r, r).astype(np.longdouble) - np.sin(rl)).max()
    ec = np.abs(_poly(C, r, np.ones_like(r)).astype(np.longdouble) - np.cos(rl)).max()
    return float(max(es, ec))


def test_polynomials_inside_and_beyond_their_range():
    q = np.pi / 4
    inside = _worst(-q, q)
    slack = _worst(-(q + 0.04), q + 0.04)          # what the trip-start check allows
    double = _worst(-(q + 0.10), q + 0.10)         # + an in-trip gain of 0.06 rad (thetadotdot = 10 000 rad/s^2)
    far = _worst(-(q + 0.25), q + 0.25)
    print(f"max |error|: inside {inside:.2e}, +0.04 {slack:.2e}, +0.10 {double:.2e}, +0.25 {far:.2e}")
    assert inside <= 2.3e-16 and slack <= 4e-16
    assert double <= 3e-15            # still rounding level: five orders below any state error that matters
    assert far <= 1e-12
