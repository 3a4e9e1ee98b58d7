import sys, os.path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from nefes_amd import dist as D
D.ONE_RANK_COLLECTIVES = True          # the group of one still issues the pose-gradient all-reduce
c = torch.arange(12, dtype=torch.float32, device="cuda").reshape(3, 4).requires_grad_()
y = (D.replicate_pose(c) ** 2).sum(); y.backward()
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("nccl 1-rank ok", float(t.sum()), float(c.grad.sum()))
dist.destroy_process_group()
