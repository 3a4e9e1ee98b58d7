"""Train-mode step at the headline network shape (8x256, C=16, fine net with transient head): forward with saved
pre-activations + weight gradients, 4096 rays x 192 samples."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from nefes_amd import lib as L, ops, train as TR
from nefes_amd.field import NeRFH_NFF
dev = 'cuda'
N, S = 4096, 192
fine = NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).to(dev)
g = torch.Generator().manual_seed(0)
o = (torch.randn(N, 3, generator=g) * 0.3).to(dev)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].to(dev)
G = torch.randn(N, 25, S, generator=g).to(dev)
for it in range(3):
    for p in fine.parameters(): p.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    raw = TR.field_train(fine, L.FIELD_FULL, o, d, d, z)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (raw * G).sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
mac = 665088
print(f"train 8x256 FULL, {N*S} samples: forward {1e3*(t1-t0):.1f} ms ({2*mac*N*S/(t1-t0)/1e12:.0f} TFLOP/s), weight-gradient pass {1e3*(t2-t1):.1f} ms ({2*2*mac*N*S/(t2-t1)/1e12:.0f} TFLOP/s for dX+dW)")
