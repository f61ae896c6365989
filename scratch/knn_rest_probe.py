"""How many queries of a refinement step reach the wave-per-query pass, over a pass of the benchmark sequence: GridInfo of the resident index
(read back after a keyframe: the counters of its LAST query set).  usage: python3 scratch/knn_rest_probe.py"""
import contextlib, os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import torch
from e2ehip.synthetic import make_sequence
from online_adaption import SLAM, default_config
H, W, L = 480, 640, 60
cfg = default_config(H, W, L)
cfg.DEBUG.print_metrics = False
cfg.MODEL.odom = "gt"
cfg.DATA.name = "ICL"
cfg.DEMO.frame_threshold = 0.05
seq = make_sequence(L, H, W, seed=1234, step=0.06, K=None, holes=0.0, scene="plane")
torch.manual_seed(bench.NET_SEED)
with contextlib.redirect_stdout(sys.stderr):
    slam = SLAM(cfg, sequence=seq)
slam.set_refinement_mode()
from e2ehip import fusionmap
LAST = {}
_build = fusionmap.ResidentKnnIndex.build
def build(self):
    torch.cuda.synchronize()
    raw = bytes(self.ws[:72].cpu().numpy())
    LAST["unresolved"] = struct.unpack("I", raw[56:60])[0]
    LAST["h"] = struct.unpack("f", raw[36:40])[0]
    nc = 256 ** 3; nb = (nc + 1 + 1023) // 1024
    off = 256 + 8 * ((nc + 4) & ~3) + 4 * (nb + 1) + 4 * 6 * 2048
    n = LAST["unresolved"]
    plan = slam.step_plan
    if n and plan is not None:
        un = self.ws[off:off + 4 * n].view(torch.int32).long()
        d = plan.nn_d[un].sqrt() / LAST["h"]
        q = torch.quantile(d, torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], device=d.device)).tolist()
        LAST["cells"] = " ".join(f"{v:.1f}" for v in q)
        LAST["sum_r2"] = float((d * d).sum())
    return _build(self)
fusionmap.ResidentKnnIndex.build = build
slam.first_iter = True
sched = slam.keyframe_schedule()
for i, pair in enumerate(sched):
    nxt = sched[i + 1] if i + 1 < len(sched) else None
    slam.refinement(*pair, max_steps=3, next_pair=nxt)
    slam.first_iter = False
    if i in (2, 5, 10, 20, 30, 40, 50, 57):
        torch.cuda.synchronize()
        raw = bytes(slam.map._knn.ws[:72].cpu().numpy())
        bb = struct.unpack("6I", raw[:24]); org = struct.unpack("3f", raw[24:36]); h, ih = struct.unpack("2f", raw[36:44]); dims = struct.unpack("3i", raw[44:56])
        nun, npts, ncell = struct.unpack("3I", raw[56:68])
        print(f"keyframe {i + 1:3d}  map {int(slam.map.count[0]):9d}  index over {npts:9d} pts  h {h:.4f}  dims {dims}  cells {ncell}  pts/cell {npts / ncell:.2f}  unresolved in the last query set BEFORE this rebuild {LAST.get('unresolved')} (h {LAST.get('h', 0):.4f})  NN distance of those in cells p10/50/90/99/max: {LAST.get('cells')}  sum r^2 {LAST.get('sum_r2', 0):.0f}")
