"""Diagnostic (not a pytest module): the free-running two-keyframe comparison GPU vs oracle on the seed-7 case of
tests/test_gpu_driver.py, plus a per-tensor comparison of the first step's gradients.

    python tests/diag_free_run.py [scale_of_head_weights=40]        (on an MI355X)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import depthnet, refine  # noqa: E402


def main():
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM, default_config
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    H, W, L = 64, 96, 3
    seq = make_sequence(L, H, W, seed=7)
    colors, gt, K, poses = seq
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * scale
    # ---- oracle: free run, gradients of every step captured before the optimiser step -------------------------------------------
    ora = refine.Refiner(sd, refine.Config())
    ograds, orig = [], ora.opt.step

    def step(*a, **k):
        ograds.append({kk: ora.sd[kk].grad.detach().clone() for kk in ora.train_keys})
        return orig(*a, **k)
    ora.opt.step = step
    recs = []
    for a, b in ((0, 1), (1, 2)):
        recs += ora.refine_pair(colors[:, [a, b]], gt[:, [a, b]], poses[:, [a, b]], K)
    # ---- GPU: first-step gradients from the same weights ---------------------------------------------------------------------------
    cfg = default_config(H, W, L)
    cfg.DEMO.frame_threshold = 0.0
    cfg.DEBUG.print_metrics = False
    slam = SLAM(cfg, sequence=seq, state_dict=sd)
    slam.set_refinement_mode()
    sp = slam._step_plan()
    slam._load_pair(sp, 0, 1)
    flat = slam.optimizer.flat
    sp.use_graphs = False
    sp._forward()
    sp.init.copy_(sp.delta)
    flat.grad.zero_()
    sp._backward(False, False)
    torch.cuda.synchronize()
    params = dict(slam.models["depth"].named_parameters())
    offs = {id(p): o for p, o in zip(flat.params, flat.offsets)}
    print(f"{'tensor':48s} {'|g|max':>10s} {'maxerr/|g|max':>14s} {'rel L2':>10s} {'sign flips':>11s} {'flip weight':>12s}")
    tot_w, tot_flip = 0.0, 0.0
    for k in ora.train_keys:
        p = params[k]
        g = flat.grad[offs[id(p)]: offs[id(p)] + p.numel()].view_as(p).cpu().double()
        r = ograds[0][k].double()
        err = (g - r).abs().max() / (r.abs().max() + 1e-300)
        rel = (g - r).norm() / (r.norm() + 1e-300)
        flip = (torch.sign(g) != torch.sign(r))
        fw = r.abs()[flip].sum() / (r.abs().sum() + 1e-300)
        tot_w += float(r.abs().sum()); tot_flip += float(r.abs()[flip].sum())
        print(f"{k:48s} {float(r.abs().max()):10.3e} {float(err):14.3e} {float(rel):10.3e} {int(flip.sum()):7d}/{r.numel():<8d} {float(fw):12.3e}")
    print(f"sum |g| over flipped signs / sum |g| = {tot_flip / tot_w:.3e}")
    lp, lr = float(sp.loss.loss[0]), float(sp.loss.loss[1])
    print("step 0: GPU photometric", lp, "oracle", recs[0]["photometric"], "reg", lr, recs[0]["reg"], "ratio", float(sp.ratio), recs[0]["ratio"])
    # ---- GPU free run ---------------------------------------------------------------------------------------------------------------
    cfg.DEBUG.print_metrics = True
    slam2 = SLAM(cfg, sequence=seq, state_dict=sd)
    slam2.main()
    log = torch.stack(slam2.log).numpy()
    print("step   GPU photometric   oracle photometric   rel diff     GPU ratio  oracle ratio")
    for i, r in enumerate(recs):
        print(f"{i}   {log[i, 1]:.6f}   {r['photometric']:.6f}   {abs(log[i, 1] - r['photometric']) / r['photometric']:.2e}   {log[i, 3]:.6f}  {r['ratio']:.6f}")


if __name__ == "__main__":
    main()
