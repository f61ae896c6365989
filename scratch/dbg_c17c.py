import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
from e2ehip import _lib as L, conv
lib = L.load()
DEV = "cuda:0"
B, Cx, H, W, Cout, k = 2, 256, 30, 40, 256, 3
lib.e2e_conv_gemm_force(64, 64, 3)
g = torch.Generator().manual_seed(1)
rnd = lambda *shape: torch.randn(*shape, generator=g)
x = rnd(B, Cx, H, W).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = (rnd(Cout, Cx, k, k) / (Cx * k * k) ** 0.5).to(DEV).requires_grad_(True)
scale = (rnd(Cout).abs() + 0.5).to(DEV); shift = rnd(Cout).to(DEV)
res = rnd(B, Cout, H, W).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
y = conv.conv2d(x, w, None, 1, 1, "zeros", "relu", (scale, shift), res, None, 1, None)
y0 = y.detach().clone()
gy = rnd(*y.shape).to(DEV)
torch.cuda.synchronize()
gs = torch.autograd.grad(y, [x, w, res], gy)
torch.cuda.synchronize()
print("y changed by backward:", int((y.detach() != y0).sum()))
dres_ref = gy * (y0 > 0)
bad = (gs[2] != dres_ref)
print("dres bad elements", int(bad.sum()))
if int(bad.sum()):
    idx = bad.permute(0, 2, 3, 1).reshape(-1, Cout).any(1).nonzero().flatten()     # bad pixels (NHWC rows)
    print("bad pixel rows:", idx.tolist()[:40], "count", idx.numel())
    chans = bad.permute(0, 2, 3, 1).reshape(-1, Cout)[idx[0]].nonzero().flatten()
    print("bad channels at first bad pixel:", chans.tolist()[:16], "count", chans.numel())
    v = gs[2].permute(0, 2, 3, 1).reshape(-1, Cout)[idx[0], chans[:6]]
    print("values there:", v.tolist(), "expected", dres_ref.permute(0, 2, 3, 1).reshape(-1, Cout)[idx[0], chans[:6]].tolist())
lib.e2e_conv_gemm_force(0, 0, 0)
