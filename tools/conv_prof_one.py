import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import nn_ops
DEV = "cuda:0"
cases = {"up01": (16, 0, 2, 480, 640, 16, 3, 1, 1, "reflect", "elu"), "layer1": (64, 0, 1, 120, 160, 64, 3, 1, 1, "zeros", "relu"),
         "up11": (32, 64, 2, 240, 320, 32, 3, 1, 1, "reflect", "elu"), "l2s2": (64, 0, 1, 120, 160, 128, 3, 2, 1, "zeros", "relu"),
         "conv1": (3, 0, 1, 480, 640, 64, 7, 2, 3, "zeros", "relu"), "layer2": (128, 0, 1, 60, 80, 128, 3, 1, 1, "zeros", "relu"),
         "layer3": (256, 0, 1, 30, 40, 256, 3, 1, 1, "zeros", "relu"), "layer4": (512, 0, 1, 15, 20, 512, 3, 1, 1, "zeros", "relu"),
         "up41": (256, 256, 2, 30, 40, 256, 3, 1, 1, "reflect", "elu"), "up00": (32, 0, 1, 240, 320, 16, 3, 1, 1, "reflect", "elu"),
         "up21": (64, 64, 2, 120, 160, 64, 3, 1, 1, "reflect", "elu")}
FWD_ONLY = sys.argv[1].endswith("fwd")
Cx, Cs, up, H, W, Cout, k, s, p, pm, act = cases[sys.argv[1][:-3] if FWD_ONLY else sys.argv[1]]
B = 2
x = torch.randn(B, Cx, H // up, W // up, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
skip = torch.randn(B, Cs, H, W, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) if Cs else None
w = (torch.randn(Cout, Cx + Cs, k, k, device=DEV) * 0.05).requires_grad_(True)
bias = torch.randn(Cout, device=DEV).requires_grad_(True) if pm == "reflect" else None
leaves = [t for t in (x, skip, w, bias) if t is not None]
for _ in range(12):
    if FWD_ONLY:
        with torch.no_grad():
            y = nn_ops.conv2d(x, w, bias, s, p, pm, act, None, None, skip, up, None)
        continue
    y = nn_ops.conv2d(x, w, bias, s, p, pm, act, None, None, skip, up, None)
    torch.autograd.grad(y, leaves, torch.ones_like(y))
torch.cuda.synchronize()
