import ctypes, sys, os, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
so = "/tmp/lg_stamped.so"
subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-x", "hip",
                os.path.join(ROOT, "scratch/lossgrad_stamped.hip"), os.path.join(ROOT, "end-to-end-self-supervised-slam_amd/csrc/abi.cpp"), "-o", so], check=True)
lib = ctypes.CDLL(so)
from e2ehip import _lib as L
from synth import make_pair
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W = 480, 640
dev = "cuda:0"
s = make_pair(H, W, B=B)
t = {k: v.to(dev).contiguous() for k, v in s.items()}
src, tgt = t["src"].permute(0, 3, 1, 2), t["tgt"].permute(0, 3, 1, 2)
dsrc = (s["depth"] + 0.1).to(dev); it = (s["depth"] + 0.05).to(dev); is_ = (s["depth"] + 0.15).to(dev)
loss = torch.zeros(2, device=dev); g1 = torch.empty_like(t["depth"]); g2 = torch.empty_like(t["depth"])
ws = torch.zeros(4 * 20 * 60 * B + 16, device=dev)
nblk = 20 * 30 * B
dbg = torch.zeros(nblk * 8, device=dev, dtype=torch.int64)
vp = ctypes.c_void_p
f = lib.dbg_lossgrad
f.argtypes = [vp, vp, L.Strides, vp, L.Strides, vp, vp, vp] + [ctypes.c_int] * 3 + [vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp] + [ctypes.c_int] * 3 + [vp, vp]
p = lambda x: vp(x.data_ptr())
for _ in range(5):
    rc = f(p(t["depth"]), p(src), L.strides4(src), p(tgt), L.strides4(tgt), p(t["K"]), p(t["invK"]), p(t["T"]), 1, 1, 2, p(it), p(is_), p(dsrc), 1.0, 0.01,
           p(loss), p(g1), p(g2), p(ws), B, H, W, None, p(dbg))
    assert rc == 0
torch.cuda.synchronize()
d = dbg.view(nblk, 8).cpu().double()
t0 = d[:, 0].min()
names = ["start", "loads issued+geo", "gathers landed", "phase1 done (LDS)", "phase2 done", "phase3 done"]
print(f"B={B}: per-workgroup stamps (cycles @100MHz ticks? s_memtime = shader clock), median over {nblk} workgroups")
for i in range(1, 6):
    seg = (d[:, i] - d[:, i - 1])
    print(f"  {names[i]:24s} median {seg.median():9.0f}  p10 {seg.quantile(0.1):9.0f}  p90 {seg.quantile(0.9):9.0f}")
print("  start spread (last - first workgroup start):", (d[:, 0].max() - t0).item(), " total span:", (d[:, 5].max() - t0).item(), " median WG life:", (d[:, 5] - d[:, 0]).median().item())
