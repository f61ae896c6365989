"""diagnostic: where does NetPlan.forward first differ from the module path (tests/test_gpu_netplan.py::test_plan_matches_module_path)?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
from test_gpu_netplan import _model, DEV
from e2ehip.netplan import NetPlan
for overlap in (False, True, True):
    B, H, W = 2, 64, 96
    m = _model()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, H, W, 3, generator=g).to(DEV)
    gd = torch.randn(B, 1, H, W, generator=g).to(DEV)
    disp = m(x, 0)[("disp", 0, 0)]
    disp2 = m(x, 0)[("disp", 0, 0)]
    print("module path deterministic:", torch.equal(disp, disp2))
    disp.backward(gd)
    feats_ref = [f.detach().clone() for f in m.encoder.features]
    plan = NetPlan(m, B, H, W, DEV, overlap=overlap)
    plan.refresh_layouts()
    d2 = plan.forward(x).clone()
    d3 = plan.forward(x).clone()
    print("overlap", overlap, "plan == module:", torch.equal(d2, disp.detach()), "plan deterministic:", torch.equal(d2, d3), "max diff", float((d2 - disp.detach()).abs().max()))
    for i, (a, b) in enumerate(zip(plan.features, feats_ref)):
        print("  feature", i, tuple(b.shape), "equal", torch.equal(a.nchw(), b), float((a.nchw() - b).abs().max()))
    for op in plan.ops:
        pass
