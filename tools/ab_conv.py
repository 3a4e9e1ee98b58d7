"""csrc/conv.hip of the shipped library against a side build (tools/ab_conv.py path/to/old.so): bit equality at FusionNet's launch
shapes, odd sizes and batch 8, distance from float64, microseconds per call."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefes_amd import lib as L, ops
old = C.CDLL(os.path.abspath(sys.argv[1]))
res, args = L.SIGNATURES["nefes_conv2d_same"]
old.nefes_conv2d_same.restype, old.nefes_conv2d_same.argtypes = res, args
new = L.load()
dev = "cuda"
def run(lib, x, wp, cout, k, b, relu, m):
    B, cin, H, W = x.shape
    y = torch.empty(B, cout, H, W, device=dev)
    rc = lib.nefes_conv2d_same(B, cin, cout, H, W, k, x.data_ptr(), m.data_ptr() if m is not None else None, wp.data_ptr(),
                               b.data_ptr() if b is not None else None, int(relu), y.data_ptr(), None)
    assert rc == 0, rc
    return y
torch.manual_seed(0)
for name, B, cin, cout, k, mask, H, W in (("L1 fwd", 1, 131, 64, 3, False, 60, 80), ("L2 fwd", 1, 64, 64, 3, False, 60, 80), ("L4 fwd", 1, 64, 128, 5, False, 60, 80),
                                   ("L4 dgrad", 1, 128, 64, 5, False, 60, 80), ("L2 dgrad", 1, 64, 64, 3, True, 60, 80), ("L1 dgrad", 1, 64, 131, 3, True, 60, 80),
                                   ("odd", 2, 7, 5, 3, True, 13, 9), ("odd5", 3, 9, 33, 5, True, 7, 11), ("one", 1, 1, 1, 5, False, 3, 3), ("b8", 8, 64, 64, 3, True, 60, 80)):
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    m = torch.randn(B, cin, H, W, device=dev) if mask else None
    wp = ops._pack_conv(w, False)
    yn, yo = run(new, x, wp, cout, k, b, True, m), run(old, x, wp, cout, k, b, True, m)
    torch.cuda.synchronize()
    ref = torch.relu(torch.nn.functional.conv2d((x * (m > 0) if mask else x).double(), w.double(), b.double(), padding=k // 2))
    t = {}
    for tag, lib in (("new", new), ("old", old)):
        for _ in range(20): run(lib, x, wp, cout, k, b, True, m)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run(lib, x, wp, cout, k, b, True, m)
        e1.record(); torch.cuda.synchronize()
        t[tag] = e0.elapsed_time(e1) / 200 * 1e3
    print(f"{name:9s} B{B} {cin:3d}->{cout:3d} {k}x{k} {H}x{W} bit-identical {bool((yn == yo).all())}  max|new-f64| {float((yn.double() - ref).abs().max()):.2e}  new {t['new']:6.1f} us  old {t['old']:6.1f} us")
