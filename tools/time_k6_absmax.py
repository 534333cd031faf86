import os, sys, torch
sys.path.insert(0, "/root/repo")
from robustmvd_amd import ops, _lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.rand(5, 768, 1152, 8, generator=g).to(dev)
wt = (torch.randn(8, 8, 3, 3, generator=g) * 0.1).to(dev)
sc, sh = (torch.rand(8, generator=g) + 0.5).to(dev), (torch.randn(8, generator=g) * 0.1).to(dev)
pk = ops.pack_conv2d_weights(wt)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
am = torch.zeros(1, device=dev)
print("plain  %.1f us" % timeit(lambda: ops.conv2d_bn_relu(x, pk[0], 8, 8, 3, 1, sc, sh)))
print("absmax %.1f us" % timeit(lambda: ops.conv2d_bn_relu(x, pk[0], 8, 8, 3, 1, sc, sh, out_absmax=am)))
y = ops.conv2d_bn_relu(x, pk[0], 8, 8, 3, 1, sc, sh)
print("pass   %.1f us" % timeit(lambda: ops.absmax(y)))
