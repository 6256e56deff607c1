"""Per-iteration cost of the all-gather machinery (PyTorch ProcessGroupNCCL + RCCL launch) with
a single rank: what an iteration pays on top when world > 1, minus the wire time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import swimmer_amd as sw
ep = sw.EnvParam("x", n=3, H=1000, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
ap = sw.ARSParam("x", V1=False, n_iter=1, H=1000, N=512, b=512, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
agent = sw.ARSAgent(ep, ap, seed=0)
tag = "forced collective" if os.environ.get("SWIMMER_FORCE_COLLECTIVE") else "no collective"
for name, stream in (("null stream", None), ("side stream", torch.cuda.Stream())):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5): agent.run_iteration_async(want_returns=False)
        torch.cuda.synchronize()
        K = 200
        t0 = time.perf_counter()
        for _ in range(K): agent.run_iteration_async(want_returns=False)
        torch.cuda.synchronize()
        print(tag, name, f": {(time.perf_counter()-t0)/K*1e3:.4f} ms/iter")
dist.destroy_process_group()
