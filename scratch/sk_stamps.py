#!/usr/bin/env python3
"""Phase stamps of the stream-K kernel (diagnostic build, see scratch/conv_stamps.py): layer1 forward on G persistent workgroups."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import _lib as L  # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "scratch", "_stamped", "libe2eslam_hip_stamped.so")
lib = L.load()
lib.e2e_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
DEV = "cuda:0"
B, H, W, Cin, Cout = 2, 120, 160, 64, 64
x = torch.randn(B, H, W, Cin, device=DEV)
wf = torch.randn(9 * Cin, Cout, device=DEV) * 0.05
bias = torch.randn(Cout, device=DEV)
out = torch.empty(B, H, W, Cout, device=DEV)
ws = torch.zeros(lib.e2e_conv_tuned_workspace_floats(B * H * W, Cout), device=DEV)
for G in (512, 768):
    for rep in range(3):
        L.call("e2e_conv2d_fwd_tuned", L.ptr(x), None, Cin, 1, L.ptr(wf), Cout, None, L.ptr(bias), None, L.ptr(out), B, H, W, Cin, Cout, 3, 3, 1, 1, 0, 1, 0.0, 1.0,
               L.ptr(ws), 64, 64, -G, L.stream())
        torch.cuda.synchronize()
    st = np.zeros(G * 8, dtype=np.uint64)
    lib.e2e_debug_read_stamps(st.ctypes.data, G * 8)
    st = st.reshape(G, 8)
    t0 = st[:, 0].min()
    rel = (st[:, :7].astype(np.int64) - int(t0)) * 0.01
    names = ["start", "first piece: prologue done", "first piece: K loop done", "first piece: finished (slab / epilogue)", "last piece: wait begins", "last piece: flags seen", "end"]
    print(f"G = {G}: launch span {rel[:, 6].max():.2f} us; pieces per workgroup: {np.bincount(st[:, 7].astype(int))}")
    for i, nm in enumerate(names):
        v = rel[:, i][st[:, i] >= t0] if i in (4, 5) else rel[:, i]
        print(f"  {nm:42s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}   (n = {len(v)})")
    w = (st[:, 5].astype(np.int64) - st[:, 4].astype(np.int64)) * 0.01
    w = w[st[:, 4] >= t0]
    print(f"  wait for partial tiles: median {np.median(w):.2f} max {w.max():.2f} us")
