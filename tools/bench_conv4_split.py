import os, sys, torch
sys.path.insert(0, "/root/repo")
from robustmvd_amd import ops, _lib as L
dev = torch.device("cuda:0")
CIN = int(sys.argv[1]) if len(sys.argv) > 1 else 32   # 32: conv4 at 64x48x72, 16: conv2 at 128x96x144
shape = (1, 64, 48, 72, 32) if CIN == 32 else (1, 128, 96, 144, 16)
x = torch.rand(*shape, device=dev)
wt = torch.randn(CIN, CIN, 3, 3, 3, device=dev) * 0.05
sc, sh = torch.rand(CIN, device=dev) + 0.5, torch.randn(CIN, device=dev) * 0.1
w32, _, _ = ops.pack_conv3d_weights(wt, L.CONV3D_STRIDE1)
wsp = ops.pack_conv3d_weights_split(wt)
def timeit(fn, n=20):
    for _ in range(5): y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y
t0, y0 = timeit(lambda: ops.conv3d_bn_relu(x, w32, CIN, CIN, sc, sh, L.CONV3D_STRIDE1, relu=True))
am = ops.absmax(x)
t1, y1 = timeit(lambda: ops.conv3d_bn_relu_split(x, wsp, sc, sh, relu=True, x_absmax=am))
t2, _ = timeit(lambda: ops.conv3d_bn_relu_split(x, wsp, sc, sh, relu=True))
t3, _ = timeit(lambda: ops.absmax(x))
print(f"conv {CIN}->{CIN} {shape[1]}x{shape[2]}x{shape[3]}: fp32 {t0*1e3:.1f} us, split {t1*1e3:.1f} us, split+absmax {t2*1e3:.1f} us, absmax {t3*1e3:.1f} us, maxdiff {float((y1-y0).abs().max()):.2e}")
