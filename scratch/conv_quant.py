import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import nn_ops
DEV = "cuda:0"
def t(B, C, H, W, Co):
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, 3, 3, device=DEV) * 0.05
    with torch.no_grad():
        for _ in range(5): nn_ops.conv2d(x, w, None, 1, 1, "zeros", "relu", None, None, None, 1, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): nn_ops.conv2d(x, w, None, 1, 1, "zeros", "relu", None, None, None, 1, None)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    M = B * H * W; gf = 2 * M * Co * C * 9 / 1e9
    print(f"B{B} C{C} {H}x{W} Co{Co}: M={M} tiles64={(M+63)//64 * ((Co+63)//64)}  {us:7.1f} us  {gf/us*1e-3*1e3:6.1f} TF/s")
for (H, W) in ((128, 64), (128, 128), (120, 160), (128, 192), (128, 256), (256, 256), (512, 256)):
    t(2, 64, H, W, 64)
t(2, 128, 60, 80, 128); t(2, 128, 64, 64, 128); t(2, 128, 64, 96, 128); t(2, 128, 64, 128, 128)
