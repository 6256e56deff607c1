"""Is bench.py's timed region (W warm-up iterations, then K = 20) in steady state?  Fresh agents, W warm-ups, then
consecutive blocks of 20 iterations timed one after the other (design aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
import bench

torch.cuda.set_stream(torch.cuda.Stream())
for W in (3, 5, 10, 50):
    rows = []
    for rep in range(3):
        leg = bench.ArsLeg(sw, torch, 3, 1000, 512, "cuda:0")
        ag = leg.agent
        for _ in range(W):
            ag.run_iteration_async(want_returns=False)
        torch.cuda.synchronize()
        blocks = []
        for b in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                ag.run_iteration_async(want_returns=False)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 20 * 1e3)
        rows.append(blocks)
        del leg, ag
    print(f"W = {W:3d}: ms/iteration of consecutive 20-iteration blocks, three fresh agents: "
          + " | ".join(" ".join(f"{x:.4f}" for x in r) for r in rows), flush=True)
