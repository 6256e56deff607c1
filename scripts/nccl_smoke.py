"""RCCL sanity on the one-GPU box: world_size 1, the exact collective the ARS path uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
send = torch.arange(100, dtype=torch.float64, device="cuda:0")
out = torch.empty(100, dtype=torch.float64, device="cuda:0")
dist.all_gather_into_tensor(out, send)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("nccl ok", bool(torch.equal(out, send)), float(t))
import swimmer_amd as sw
ep = sw.EnvParam("x", n=3, H=50, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
ap = sw.ARSParam("x", V1=False, n_iter=1, H=50, N=8, b=8, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
a = sw.ARSAgent(ep, ap, seed=0)
print("agent under an initialised nccl group: world", a.world, "returns", len(a.runOneIteration()))
dist.destroy_process_group()
