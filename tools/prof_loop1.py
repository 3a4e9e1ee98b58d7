#!/usr/bin/env python3
"""One query image, 50 refinement iterations as replayed HIP graphs (bench.py --workload loop50's single-image leg only): the
program to put under `rocprofv3 --kernel-trace --stats` for the per-kernel times of one iteration.  --eager: no graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

graph = "--eager" not in sys.argv
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sec, rays, err = bench.refinement_loop(dev, iters=50, graph=graph)
print(f"{sec * 1e3:.2f} ms per image (50 iterations, {'graph' if graph else 'eager'}): {sec * 1e3 / 50:.3f} ms per iteration; pose error after 50 iterations {err}")
