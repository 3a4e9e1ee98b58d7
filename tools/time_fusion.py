import sys, time, torch
sys.path.insert(0, '.')
from nefes_amd.field import NeRFH_NFF
net = NeRFH_NFF('coarse', W=128, f_dim=128).requires_grad_(False).cuda()
for B in (1, 3, 8, 1, 8):
    rgb = torch.rand(B * 4800, 3, device='cuda', requires_grad=True); feat = torch.randn(B * 4800, 128, device='cuda', requires_grad=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, _, f = net.run_fusion_net(rgb, feat, 60, 80, B, per_image_norm=B > 1)
    f.sum().backward()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5):
        _, _, f = net.run_fusion_net(rgb, feat, 60, 80, B, per_image_norm=B > 1); f.sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: first call {t1 - t0:.2f} s, then {(t2 - t1) / 5 * 1e3:.2f} ms per fwd+bwd")
