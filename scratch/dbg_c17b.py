import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import _lib as L
lib = L.load()
DEV = "cuda:0"
B, Cin, Cout, H, W, k = 2, 256, 256, 30, 40, 3
ld = lambda n: (n + 3) // 4 * 4
torch.manual_seed(0)
dz = torch.randn(B, H, W, Cout, device=DEV)
wb = torch.randn(k * k * Cout, ld(Cin), device=DEV) * 0.05
rows = B * H * W
for S in (1, 2, 3, 4, 6):
    lib.e2e_conv_gemm_force(64, 64, S)
    nws = max(lib.e2e_conv2d_splitk_workspace_floats(rows, Cin, k * k * Cout), 1)
    PADF = 1 << 20
    big = torch.full((PADF + nws + PADF,), 7.0, device=DEV)
    ws = big[PADF:PADF + nws]
    bigo = torch.full((PADF + rows * Cin + PADF,), 7.0, device=DEV)
    dxp = bigo[PADF:PADF + rows * Cin]
    L.call("e2e_conv2d_bwd_data", L.ptr(dz), L.ptr(wb), ld(Cin), L.ptr(dxp), B, H, W, Cin, Cout, H, W, k, k, 1, 1, 0, L.ptr(ws) if S > 1 else None, L.stream())
    torch.cuda.synchronize()
    pre, post = big[:PADF], big[PADF + nws:]
    preo, posto = bigo[:PADF], bigo[PADF + rows * Cin:]
    ref = None
    if S == 1:
        ref0 = dxp.clone()
    err = float((dxp - ref0).abs().max() / ref0.abs().max())
    print("S", S, "nws", nws, "ws guard pre/post corrupted:", int((pre != 7).sum()), int((post != 7).sum()), "out guard:", int((preo != 7).sum()), int((posto != 7).sum()),
          "err vs S=1", err, flush=True)
    if int((post != 7).sum()):
        idx = (post != 7).nonzero().flatten()
        print("   post idx range", int(idx.min()), int(idx.max()))
    if int((pre != 7).sum()):
        idx = (pre != 7).nonzero().flatten()
        print("   pre idx range", int(idx.min()), int(idx.max()))
lib.e2e_conv_gemm_force(0, 0, 0)
