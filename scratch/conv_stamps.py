#!/usr/bin/env python3
"""Phase stamps of k_conv_gemm workgroups (diagnostic build scratch/_stamped/libe2eslam_hip_stamped.so, -DE2E_CONV_STAMPS): when does every
workgroup of ONE layer1-forward launch start, finish its prologue, its first chunk, its K loop and its epilogue, and on which XCD / CU.

    (cd end-to-end-self-supervised-slam_amd/csrc && for f in *.hip; do hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 \
        -DE2E_CONV_STAMPS -x hip -c $f -o ../../scratch/_stamped/$f.o; done; hipcc --offload-arch=gfx950 -shared -fPIC \
        -o ../../scratch/_stamped/libe2eslam_hip_stamped.so ../../scratch/_stamped/*.o)
    python scratch/conv_stamps.py        (on an MI355X; s_memrealtime ticks at 100 MHz = 10 ns)"""
import ctypes
import os
import sys

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
for tile in ((64, 64, 1), (32, 64, 1)):
    for rep in range(3):
        L.call("e2e_conv2d_fwd_tuned", L.ptr(x), None, Cin, 1, L.ptr(wf), Cout, None, L.ptr(bias), None, L.ptr(out), B, H, W, Cin, Cout, 3, 3, 1, 1, 0, 1, 0.0, 1.0,
               L.ptr(ws), tile[0], tile[1], tile[2], L.stream())
        torch.cuda.synchronize()
    n = (B * H * W + tile[0] - 1) // tile[0]
    st = np.zeros(n * 8, dtype=np.uint64)
    lib.e2e_debug_read_stamps(st.ctypes.data, n * 8)
    st = st.reshape(n, 8)
    st2 = np.zeros(n * 4, dtype=np.uint64)
    lib.e2e_debug_read_stamps(st2.ctypes.data, -n * 4)
    st2 = st2.reshape(n, 4)
    t0 = st[:, 0].min()
    rel2 = (st2[:, :3].astype(np.int64) - st[:, 0:1].astype(np.int64)) * 0.01
    print("  prologue, us after the workgroup's own start: " + "  ".join(f"{nm} median {np.median(rel2[:, i]):.2f} max {rel2[:, i].max():.2f}" for i, nm in enumerate(["kernel arguments read", "rows decoded", "tap tables built"])))
    rel = (st[:, :6].astype(np.int64) - int(t0)) * 0.01          # us
    print(f"tile {tile}: {n} workgroups; launch span {rel[:, 5].max():.2f} us")
    names = ["start", "first loads issued", "prologue done (1st barrier)", "K loop done", "first chunk done", "epilogue stores landed"]
    for i, nm in enumerate(names):
        print(f"  {nm:32s} min {rel[:, i].min():7.2f}  median {np.median(rel[:, i]):7.2f}  max {rel[:, i].max():7.2f}")
    d = rel[:, 3] - rel[:, 2]
    print(f"  K loop duration  min {d.min():.2f} median {np.median(d):.2f} max {d.max():.2f} us; prologue median {np.median(rel[:, 2] - rel[:, 0]):.2f}; "
          f"first chunk median {np.median(rel[:, 4] - rel[:, 2]):.2f}; epilogue median {np.median(rel[:, 5] - rel[:, 3]):.2f}")
    xcc = st[:, 7] & 0xF
    hw = st[:, 6]
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    key = xcc * 10000 + se * 100 + cu
    u, cnt = np.unique(key, return_counts=True)
    print(f"  distinct (xcc, se, cu) = {len(u)}; workgroups per CU: " + ", ".join(f"{c}:{(cnt == c).sum()}" for c in sorted(set(cnt))))
    for c in sorted(set(cnt)):
        sel = np.isin(key, u[cnt == c])
        print(f"    CUs with {c} workgroups: K loop median {np.median(d[sel]):.2f} us, end of epilogue median {np.median(rel[sel, 5]):.2f} max {rel[sel, 5].max():.2f}")
    ph = np.zeros(n * 8, dtype=np.uint64)
    lib.e2e_debug_read_stamps(ph.ctypes.data, -(1 << 20) - n * 8)
    ph = ph.reshape(n, 8)[:, :6].astype(np.float64)
    nch = 9 * Cin // 32
    names_ph = ["issue loads c+2 / next tap", "fragment reads + MFMA issue", "wait chunk c+1 + LDS store", "barrier", "-", "loop control"]
    if ph.sum() == 0:
        print("    (K-loop phase clocks: recorded by the plain two-step loop only -- the software-pipelined loop every product launch takes has no "
              "phase boundaries to pin; profiles/r03_conv_phase_clocks_plain_loop.txt holds the last run of the plain loop)")
    else:
      print(f"    K-loop phases, shader-clock cycles per chunk (wave 0; {nch} chunks; pinned schedule): " +
          "  ".join(f"{nm}: {np.median(ph[:, i]) / nch:.0f}" for i, nm in enumerate(names_ph) if nm != "-") + f"  | sum {np.median(ph.sum(1)) / nch:.0f}")
      for c in sorted(set(cnt)):
        sel = np.isin(key, u[cnt == c])
        print(f"      CUs with {c} workgroups: " + "  ".join(f"{np.median(ph[sel, i]) / nch:.0f}" for i in (0, 1, 2, 3, 5)))
    print("    per XCD: workgroups, CUs, K loop median, end of epilogue median / max")
    for xc in sorted(set(xcc)):
        sel = xcc == xc
        print(f"      xcc {int(xc)}: {int(sel.sum()):4d} wgs {len(set(key[sel])):3d} CUs  K loop {np.median(d[sel]):6.2f}  end {np.median(rel[sel, 5]):6.2f} / {rel[sel, 5].max():6.2f}")
    order = np.argsort(-rel[:, 5])[:12]
    print("    slowest workgroups: id xcc se cu wgs_on_cu | start prologue_done kloop_done end")
    for i in order:
        print(f"      {int(i):5d} {int(xcc[i])} {int(se[i])} {int(cu[i]):2d} {int(cnt[np.searchsorted(u, key[i])])} | {rel[i, 0]:6.2f} {rel[i, 2]:6.2f} {rel[i, 3]:6.2f} {rel[i, 5]:6.2f}")
    slow_cu = key[order[0]]
    print("    every workgroup of the slowest workgroup's CU:")
    for i in np.nonzero(key == slow_cu)[0]:
        print(f"      {int(i):5d} | {rel[i, 0]:6.2f} {rel[i, 2]:6.2f} {rel[i, 3]:6.2f} {rel[i, 5]:6.2f}")
