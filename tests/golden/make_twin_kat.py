"""Extract the numbers the reference RECORDED for its native environment
(rlglue/test/acceleration-compare.txt, the stdout of rlglue/test/TestComputeAcceleration.cpp
kept in the reference repository) into tests/golden/twin_kat.json:

  :5-7    Coulom's program, state 1: state, barycentre acceleration, angle accelerations
  :10-12  Coulom's program, state 2
  :27-43  the assembled 17 x 17 matrix A of the native env for state 1, torque = max_u / 2
  :46-62  the right-hand side B
  :84-100 the solution X (thdd_1..3 | f_0..f_3 | Gdd_1..3)
  :102-103 G_dotdot and the angle accelerations

Data only (recorded outputs, 6 significant digits), no source text.  Run in the build container:
    python tests/golden/make_twin_kat.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "twin_kat.json")

NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def numbers(text):
    return [float(x) for x in re.findall(NUM, text)]


def main():
    lines = open(os.path.join(REF, "rlglue", "test", "acceleration-compare.txt")).read().split("\n")
    ln = lambda a, b: lines[a - 1:b]          # 1-based inclusive, as cited above
    coulom = []
    for first in (5, 10):
        st, g, t = ln(first, first + 2)
        assert st.startswith("State:") and g.startswith("Barycenter") and t.startswith("Angles")
        coulom.append({"state": numbers(st), "barycenter_acceleration": numbers(g),
                       "angle_accelerations": numbers(t)})
    assert lines[25].startswith("-----------------Matrix A")
    A = [numbers(r) for r in ln(27, 43)]
    assert lines[44].startswith("-----------------Vector B")
    B = [numbers(r)[0] for r in ln(46, 62)]
    assert lines[82].startswith("-----------------Vector X")
    X = [numbers(r)[0] for r in ln(84, 100)]
    assert len(A) == 17 and all(len(r) == 17 for r in A) and len(B) == 17 and len(X) == 17
    gdd = numbers(lines[101].split("=")[1])
    tdd = numbers(lines[102].split(":")[1])
    rec = {"source": "rlglue/test/acceleration-compare.txt (recorded stdout, 6 significant digits)",
           "n": 3, "l_i": 1.0, "m_i": 1.0, "k": 10.0, "torque": [2.5, 2.5],
           "state": numbers(lines[23].split("=")[1]),
           "A": A, "B": B, "X": X, "G_dotdot": gdd, "angle_accelerations": tdd,
           "coulom": coulom, "coulom_delta_t": 0.0025}
    with open(OUT, "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
