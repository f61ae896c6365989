import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import depthnet
from e2ehip import _lib as L, ops, conv as e2e_conv
from e2ehip.synthetic import make_sequence
from online_adaption import SLAM, default_config
from utils.training_utils import torch_poses_to_transforms
lib = L.load()
H, W, Ln = 64, 96, 3
seq = make_sequence(Ln, H, W, seed=7)
sd = depthnet.random_state_dict(0)
sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0
cfg = default_config(H, W, Ln); cfg.DEMO.frame_threshold = 0.0; cfg.DEBUG.print_metrics = False
slam = SLAM(cfg, sequence=seq, state_dict=sd)
slam.use_graphs = False
slam.set_refinement_mode(); slam.first_iter = True
sp = slam._step_plan()
T = torch_poses_to_transforms(slam._poses_h[:, [0, 1]])[0, 1]
sp.set_pair(slam.colors[0, 0], slam.colors[0, 1], slam.gt_depths[0, 0], slam.gt_depths[0, 1], slam.intrinsics[0, 0], T, slam.poses[0, 1])
sp.inv_K[0].copy_(slam._inv_K)
m = slam.models["depth"]
names = {id(p): n for n, p in m.named_parameters()}
for it in range(2):
    sp._forward()
    if it == 0: sp.init.copy_(sp.delta)
    st = L.stream()
    sp.loss.step()
    L.call("e2e_depth_scale_bwd", L.ptr(sp.g_depth), L.ptr(sp.delta), L.ptr(sp.median_gt), L.ptr(sp.md), L.ptr(sp.net.disp.g), L.ptr(sp.ws_scale), sp.g_depth.numel(), st)
    gdisp = sp.net.disp.g.clone().view(2, 1, H, W)
    sp.net.backward()
    torch.cuda.synchronize()
    plan_g = {id(p): sp.net.sink(p).clone() for p in sp.net.parameters()}
    # autograd path on the same weights, same upstream gradient
    for p in m.parameters():
        pass
    e2e_conv.WEIGHT_EPOCH[0] += 1
    disp = m(sp.colors, 0)[("disp", 0, 0)]
    print("iter", it, "disp equal:", bool(torch.equal(disp.detach(), sp.net.disp.t.view(2, 1, H, W))))
    params = [p for p in sp.net.parameters()]
    grads = torch.autograd.grad(disp, params, gdisp)
    worst = []
    for p, g in zip(params, grads):
        a = plan_g[id(p)]
        err = float((a - g).abs().max()) / (float(g.abs().max()) + 1e-30)
        worst.append((err, names[id(p)]))
    worst.sort(reverse=True)
    print("   worst grads:", [(f"{e:.2e}", n) for e, n in worst[:6]])
    # take an optimiser step so that iteration 1 sees changed weights
    sp._adam()
    e2e_conv.WEIGHT_EPOCH[0] += 1
    sp.net._epoch = e2e_conv.WEIGHT_EPOCH[0]
