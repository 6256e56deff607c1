"""Same-box A/B of the throughput-form lane kernel (rollout_kernel<3,false,false>, every state captured)
between library builds, and what the chip's power state does to the same launch.

  python scripts/lane_ab.py [libA.so libB.so ...]      (default: the in-tree build only)

An older build to compare with (built in the container, travels to the GPU box with the snapshot; *.so is git-ignored):
  mkdir -p /tmp/r2 scripts/ab && git archive f756c65 safe-exploration-with-simulator-in-rl-algorithms_amd/csrc include | tar -x -C /tmp/r2
  (cd /tmp/r2/*/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC swimmer_kernels.hip host_rng.cpp -ldl -o $OLDPWD/scripts/ab/lib_r2.so)

Raw ctypes on sw_rollout_f64 (same signature since ABI 2), so that a round-2 build can be loaded beside
the current one in ONE process; the builds are timed interleaved (A B A B ...), every launch with its own
HIP events, median reported.  Then, with the first library only: the same launch (i) after two seconds of
idle, (ii) right after two seconds of full-chip f64 load (the n = 6 / 4096-rollout row kernel), (iii) after
another two seconds of idle -- the bench's `rollout_saturated` legs used to time THREE launches after ONE
warm-up, wherever the legs before them had left the chip.
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import swimmer_amd as sw

HBM = 8000.0


class P(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("flags", ctypes.c_int32), ("l_i", ctypes.c_double),
                ("m_i", ctypes.c_double), ("k", ctypes.c_double), ("h", ctypes.c_double),
                ("dir_x", ctypes.c_double), ("dir_y", ctypes.c_double)]


def bind(path):
    lib = ctypes.CDLL(path)
    lib.sw_rollout_f64.restype = ctypes.c_int
    lib.sw_rollout_f64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 10
    return lib


def launcher(lib, n, n_roll, H, pol, traj, rets, flags=1):
    p = P(n, flags, 1.0, 1.0, 10.0, 1e-3, 1.0, 0.0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def go():
        rc = lib.sw_rollout_f64(ctypes.byref(p), n_roll, H, pol.data_ptr(), None, None, None, rets.data_ptr(),
                                traj.data_ptr() if traj is not None else None, None, None, None, st)
        assert rc == 0, rc
    return go


def timed(go, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        go()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def main():
    libs = sys.argv[1:] or [sw._lib.library_path()]
    bound = [(os.path.basename(p), bind(p)) for p in libs]
    dev = "cuda:0"
    rng = np.random.RandomState(1)
    for n_roll in (65536, 262144):
        n, H, d, m = 3, 1000, 8, 2
        pol = torch.as_tensor(0.01 * (2 * rng.rand(4096, m, d) - 1), device=dev).repeat(n_roll // 4096, 1, 1)
        traj = torch.empty((H, d, n_roll), dtype=torch.float64, device=dev)
        rets = torch.empty(n_roll, dtype=torch.float64, device=dev)
        byts = n_roll * H * 8 * d + n_roll * (8 * m * d + 8)
        gos = [(name, launcher(lib, n, n_roll, H, pol, traj, rets)) for name, lib in bound]
        for _, go in gos:
            timed(go, 3)
        samples = {name: [] for name, _ in gos}
        for rnd in range(6):
            for name, go in gos:
                samples[name] += timed(go, 4)
        for name, _ in gos:
            s = sorted(samples[name])
            med = s[len(s) // 2]
            print(json.dumps({"rollouts": n_roll, "lib": name, "launches": len(s), "ms_median": round(med, 4),
                              "ms_min": round(s[0], 4), "ms_max": round(s[-1], 4),
                              "hbm_frac_median": round(byts / (med * 1e-3) / 1e9 / HBM, 4)}), flush=True)
        if n_roll == 65536:
            # ---- power state: the first library only
            name, go = gos[0]
            n6, r6 = 6, 4096
            pol6 = torch.as_tensor(0.01 * (2 * rng.rand(r6, 5, 14) - 1), device=dev)
            ret6 = torch.empty(r6, dtype=torch.float64, device=dev)
            load = launcher(bound[0][1], n6, r6, H, pol6, None, ret6, flags=2)

            def first3(tag):
                t = timed(go, 1)      # ONE warm-up, then three launches: the bench leg's old recipe
                t3 = timed(go, 3)
                ms = sum(t3) / 3
                print(json.dumps({"rollouts": n_roll, "state": tag, "warm_ms": round(t[0], 4),
                                  "ms_mean_of_3": round(ms, 4), "hbm_frac": round(byts / (ms * 1e-3) / 1e9 / HBM, 4)}),
                      flush=True)
            torch.cuda.synchronize()
            time.sleep(2.0)
            first3("after 2 s idle")
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 2.0:
                for _ in range(50):
                    load()
                torch.cuda.synchronize()
            first3("right after 2 s of full-chip f64 load")
            time.sleep(2.0)
            first3("after 2 more s idle")
            s = sorted(timed(go, 40))
            print(json.dumps({"rollouts": n_roll, "state": "40 launches back to back", "ms_median": round(s[20], 4),
                              "ms_first": None, "ms_min": round(s[0], 4), "ms_max": round(s[-1], 4),
                              "hbm_frac_median": round(byts / (s[20] * 1e-3) / 1e9 / HBM, 4)}), flush=True)
        del traj, pol, rets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
