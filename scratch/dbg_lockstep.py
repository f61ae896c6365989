import sys, os, io, contextlib, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import depthnet
from e2ehip.synthetic import make_sequence
from online_adaption import SLAM, default_config
H, W, Ln = 64, 96, 3
seq = make_sequence(Ln, H, W, seed=7)
sd = depthnet.random_state_dict(0)
sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0
def mk():
    cfg = default_config(H, W, Ln); cfg.DEMO.frame_threshold = 0.0; cfg.DEBUG.print_metrics = False
    s = SLAM(cfg, sequence=seq, state_dict=sd); s.use_graphs = False
    s.set_refinement_mode(); s.first_iter = True
    return s
a, b = mk(), mk()
for nsteps in (1, 2, 3):
    a2, b2 = mk(), mk()
    a2.refinement(0, 1, max_steps=nsteps)
    b2.refinement_autograd(0, 1, max_steps=nsteps)
    torch.cuda.synchronize()
    pa, pb = dict(a2.models["depth"].named_parameters()), dict(b2.models["depth"].named_parameters())
    worst = sorted(((float((pa[n].detach() - pb[n].detach()).abs().max()), n) for n in pa), reverse=True)[:5]
    print("after", nsteps, "steps: max |w_plan - w_autograd| :", [(f"{e:.2e}", n) for e, n in worst], flush=True)
    ma = a2.optimizer.m; mb = b2.optimizer.m
    print("    Adam m diff", float((ma - mb).abs().max()), "v diff", float((a2.optimizer.v - b2.optimizer.v).abs().max()),
          "counters", a2.optimizer.steps_done(), b2.optimizer.steps_done())
