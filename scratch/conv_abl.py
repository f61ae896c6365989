import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import nn_ops
DEV = "cuda:0"
Cx, H, W, Cout = 64, 120, 160, 64
x = torch.randn(2, Cx, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, Cx, 3, 3, device=DEV) * 0.05
for dbg in (0, 4, 32, 96, 64):
    os.environ["E2E_CONV_DBG"] = str(dbg)
    with torch.no_grad():
        for _ in range(5): nn_ops.conv2d(x, w, None, 1, 1, "zeros", "relu", None, None, None, 1, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): nn_ops.conv2d(x, w, None, 1, 1, "zeros", "relu", None, None, None, 1, None)
        e1.record(); torch.cuda.synchronize()
    print(f"dbg={dbg:2d} (4 noMFMA 32 noloop 64 nostore): {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
