import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for graph in (False, True):
    sec, rays, err = bench.refinement_loop(dev, graph=graph, mode="2")
    print("mode 2", "graph" if graph else "eager", round(sec * 1e3, 2), "ms per image", err)
