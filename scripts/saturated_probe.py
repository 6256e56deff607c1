"""Saturated rollout regime (lane kernel, rollouts filling the chip, every state captured):
ms per launch and achieved store bandwidth at three batch shapes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
import bench
sw._lib.load()
for n_roll, H in ((262144, 1000), (65536, 1000), (1 << 20, 250)):
    r = bench.aux_rollout_saturated(sw, torch, "cuda:0", n_roll=n_roll, H=H)
    print(json.dumps({k: r[k] for k in ("rollouts", "horizon", "ms", "env_steps_per_s", "achieved_GBps", "hbm_frac")}))
