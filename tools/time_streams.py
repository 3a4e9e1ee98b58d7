import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for s in (1, 2, 3, 4):
    sec, rays, err = bench.refinement_loop(dev, graph=True, mode="2", streams=s)
    print("mode 2, images on streams:", s, round(sec * 1e3, 2), "ms per image", err["hip"], flush=True)
for s in (1, 2, 3, 4):
    sec, rays, err = bench.refinement_loop(dev, graph=True, mode="3", streams=s)
    print("mode 3, images on streams:", s, round(sec * 1e3, 2), "ms per image", flush=True)
