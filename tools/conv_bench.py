#!/usr/bin/env python3
"""Per-layer timing of the depth network's convolutions (B=2, 480x640 input): native kernels vs the MIOpen scaffold.
    python tools/conv_bench.py            (on an MI355X)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
import torch.nn.functional as F
from e2ehip import nn_ops


def torch_conv(x, weight, bias, stride, padding, pad_mode, act, skip, upsample, in_norm):
    """The torch / MIOpen composition of the same layer: the A/B reference of this tool only (never imported by the package)."""
    if in_norm is not None:
        x = (x - in_norm[0]) * in_norm[1]
    if upsample != 1:
        x = F.interpolate(x, scale_factor=upsample, mode="nearest")
    if skip is not None:
        x = torch.cat([x, skip], 1)
    if pad_mode == "reflect" and padding:
        x = F.pad(x, (padding,) * 4, mode="reflect")
        padding = 0
    y = F.conv2d(x, weight, bias, stride, padding)
    return F.relu(y) if act == "relu" else (F.elu(y) if act == "elu" else y)

DEV = "cuda:0"
# name, Cx, Cskip, up, H, W (of the conv's full-res input), Cout, k, s, p, pad_mode, act
LAYERS = [
    ("conv1 7x7/2", 3, 0, 1, 480, 640, 64, 7, 2, 3, "zeros", "relu"),
    ("layer1 x4", 64, 0, 1, 120, 160, 64, 3, 1, 1, "zeros", "relu"),
    ("layer2.0.c1 /2", 64, 0, 1, 120, 160, 128, 3, 2, 1, "zeros", "relu"),
    ("layer2 x3", 128, 0, 1, 60, 80, 128, 3, 1, 1, "zeros", "relu"),
    ("layer3.0.c1 /2", 128, 0, 1, 60, 80, 256, 3, 2, 1, "zeros", "relu"),
    ("layer3 x3", 256, 0, 1, 30, 40, 256, 3, 1, 1, "zeros", "relu"),
    ("layer4.0.c1 /2", 256, 0, 1, 30, 40, 512, 3, 2, 1, "zeros", "relu"),
    ("layer4 x3", 512, 0, 1, 15, 20, 512, 3, 1, 1, "zeros", "relu"),
    ("up(4,0)", 512, 0, 1, 15, 20, 256, 3, 1, 1, "reflect", "elu"),
    ("up(4,1)", 256, 256, 2, 30, 40, 256, 3, 1, 1, "reflect", "elu"),
    ("up(3,0)", 256, 0, 1, 30, 40, 128, 3, 1, 1, "reflect", "elu"),
    ("up(3,1)", 128, 128, 2, 60, 80, 128, 3, 1, 1, "reflect", "elu"),
    ("up(2,0)", 128, 0, 1, 60, 80, 64, 3, 1, 1, "reflect", "elu"),
    ("up(2,1)", 64, 64, 2, 120, 160, 64, 3, 1, 1, "reflect", "elu"),
    ("up(1,0)", 64, 0, 1, 120, 160, 32, 3, 1, 1, "reflect", "elu"),
    ("up(1,1)", 32, 64, 2, 240, 320, 32, 3, 1, 1, "reflect", "elu"),
    ("up(0,0)", 32, 0, 1, 240, 320, 16, 3, 1, 1, "reflect", "elu"),
    ("up(0,1)", 16, 0, 2, 480, 640, 16, 3, 1, 1, "reflect", "elu"),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


def main():
    B = 2
    tot = {"hip_f": 0, "hip_b": 0, "mi_f": 0, "mi_b": 0}
    print(f"{'layer':16s} {'GFLOP':>7s} | {'hip fwd':>9s} {'TF/s':>6s} {'hip bwd':>9s} {'TF/s':>6s} | {'miopen fwd':>10s} {'bwd':>9s}")
    for name, Cx, Cs, up, H, W, Cout, k, s, p, pm, act in LAYERS:
        x = torch.randn(B, Cx, H // up, W // up, device=DEV).contiguous(memory_format=torch.channels_last)
        skip = torch.randn(B, Cs, H, W, device=DEV).contiguous(memory_format=torch.channels_last) if Cs else None
        Cin = Cx + Cs
        w = torch.randn(Cout, Cin, k, k, device=DEV) * 0.05
        bias = torch.randn(Cout, device=DEV) if pm == "reflect" else None
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        gf = 2.0 * B * Ho * Wo * Cout * Cin * k * k / 1e9
        stem = Cx == 3
        for t in (x, skip, w, bias):
            if t is not None:
                t.requires_grad_(not (stem and t is x))
        res = {}
        for backend in ("hip", "miopen"):
            if backend == "hip":
                f = lambda: nn_ops.conv2d(x, w, bias, s, p, pm, act, None, None, skip, up, (0.45, 4.44) if stem else None)
            else:
                f = lambda: torch_conv(x, w, bias, s, p, pm, act, skip, up, (0.45, 4.44) if stem else None)
            tf = timeit(lambda: f())
            y = f()
            gy = torch.randn_like(y)
            leaves = [t for t in (x, skip, w, bias) if t is not None and t.requires_grad]
            def fb():
                yy = f()
                torch.autograd.grad(yy, leaves, gy)
            tfb = timeit(fb)
            res[backend] = (tf, tfb - tf)
        tot["hip_f"] += res["hip"][0]; tot["hip_b"] += res["hip"][1]; tot["mi_f"] += res["miopen"][0]; tot["mi_b"] += res["miopen"][1]
        print(f"{name:16s} {gf:7.2f} | {res['hip'][0]:9.1f} {gf / res['hip'][0] * 1e3:6.1f} {res['hip'][1]:9.1f} {2 * gf / res['hip'][1] * 1e3:6.1f} | {res['miopen'][0]:10.1f} {res['miopen'][1]:9.1f}")
    print("totals (us, one instance per row):", {k: round(v) for k, v in tot.items()})


if __name__ == "__main__":
    main()
