"""Time the REFERENCE itself (Python/NumPy) in the build container -- it cannot travel to the
GPU box.  Single process and 8 processes (the reference's own parallel mode is one Ray actor
per seed, ars/experiment.py:63-72; emulated with multiprocessing).  Container-only script."""
import importlib.util, multiprocessing as mp, os, sys, time
import numpy as np

def load():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m

def one(seed):
    m = load()
    ep = m.EnvParam("t", n=3, H=1000, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    env = m.Environment(ep)
    P = 0.1 * (2 * np.random.RandomState(seed).rand(2, 8) - 1)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 5.0:
        env.rollout(P, covariance=np.identity(8), mean=np.zeros(8)); n += 1
    return n * 1000 / (time.perf_counter() - t0)

if __name__ == "__main__":
    print("single process: %.0f env-steps/s" % one(0))
    with mp.Pool(8) as pool:
        r = pool.map(one, range(8))
    print("8 processes: %.0f env-steps/s total (%s)" % (sum(r), ", ".join("%.0f" % x for x in r)))
