import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
import test_gpu_conv as T
from e2ehip import _lib as L, conv
lib = L.load()
DEV = "cuda:0"
case = (2, 256, 0, 1, 30, 40, 256, 3, 1, 1, "zeros", "relu", True, False, True)
B, Cx, Cs, up, H, W, Cout, k, s, p, pad_mode, act, bn, use_bias, use_res = case
for force in ((0, 0, 0), (64, 64, 1), (64, 64, 3), (64, 64, 2), (32, 128, 3)):
    lib.e2e_conv_gemm_force(*force)
    g = torch.Generator().manual_seed(sum(case[:7]))
    rnd = lambda *shape: torch.randn(*shape, generator=g)
    x = rnd(B, Cx, H, W).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (rnd(Cout, Cx, k, k) / (Cx * k * k) ** 0.5).to(DEV).requires_grad_(True)
    scale = (rnd(Cout).abs() + 0.5).to(DEV); shift = rnd(Cout).to(DEV)
    res = rnd(B, Cout, H, W).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv.conv2d(x, w, None, s, p, pad_mode, act, (scale, shift), res, None, up, None)
    yr = T._ref(x, None, w, None, scale, shift, res, up, s, p, pad_mode, act, None)
    gy = rnd(*y.shape).to(DEV)
    gs = torch.autograd.grad(y, [x, w, res], gy)
    gr = torch.autograd.grad(yr, [x, w, res], gy.double())
    errs = [float(((a.double() - b).abs().max() / b.abs().max())) for a, b in zip(gs, gr)]
    print(force, "fwd", float((y.double() - yr).abs().max() / yr.abs().max()), "dx dw dres", errs, flush=True)
    if errs[0] > 1e-3:
        d = (gs[0].double() - gr[0]).abs()
        idx = (d > 1e-3 * gr[0].abs().max()).nonzero()
        print("  bad elements:", idx.shape[0], "of", d.numel(), "first", idx[:5].tolist(), "last", idx[-5:].tolist())
lib.e2e_conv_gemm_force(0, 0, 0)
