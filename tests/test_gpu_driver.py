"""The build's online_adaption driver (fused launches) against the oracle's refinement loop and the golden
trajectory captured from the reference's own modules (tests/golden/g8)."""
import numpy as np
import pytest
import torch

from oracle import depthnet, refine

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-4          # per tensor, relative to its largest element (tests/test_gpu_network.py uses the same bound)
# the teacher-forced test compares the GPU's fp32 gradients with the fp32 ORACLE's through the whole loss chain: both sides round, and the
# oracle's CPU convolutions differently for every thread count (64x96, worst tensor of the six steps: 0.97e-4 with the GPU box's default
# count, 1.3e-4 ... 2.4e-3 with 16 threads -- fewer, longer partial sums on the oracle's side), so the bound there is 2 x GRAD_TOL and no
# test changes the thread count for the tests after it (tests/conftest.py)
TEACHER_FORCED_TOL = 2e-4


def _cfg(H, W, L):
    from online_adaption import default_config
    cfg = default_config(H, W, L)
    cfg.DEMO.frame_threshold = 0.0
    return cfg


def test_first_pair_matches_reference_trajectory(golden):
    """3 refinement steps on the first keyframe pair (no map yet): loss trajectory, median ratios, depths."""
    from online_adaption import SLAM
    g = golden("g8_refine")
    H, W = g["colors"].shape[2:4]
    slam = SLAM(_cfg(H, W, 2), sequence=(g["colors"], g["gt_depths"], g["K"], g["poses"]), state_dict=depthnet.random_state_dict(0))
    slam.main()
    log = torch.stack(slam.log)
    np.testing.assert_allclose(log[:, 0].numpy(), g["losses"].numpy(), rtol=1e-4)
    np.testing.assert_allclose(log[:, 3].numpy(), g["ratios"].numpy(), rtol=1e-4)
    with torch.no_grad():
        disp = slam.models["depth"](slam.colors[0], 0)[("disp", 0, 0)].cpu()
    torch.testing.assert_close(disp[0:1], g["final_disp0"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(disp[1:2], g["final_disp1"], rtol=1e-4, atol=1e-5)
    assert slam.map.M >= H * W


@pytest.mark.parametrize("H,W", [(64, 96), (480, 640)], ids=["64x96", "480x640"])
def test_every_step_of_two_keyframes_vs_oracle_teacher_forced(H, W):
    """Both keyframe pairs of a 3-frame sequence (the second one with the 3-D nearest-neighbour loss against the fused map), every
    refinement step against the oracle's same step: loss terms, median ratio, metrics at 1e-4, d loss / d depth, ALL 48 parameter
    gradients in every step, and the parameters after the step.

    TEACHER FORCED: before each GPU step the network weights, Adam's moments / step count and (before the second pair) the map are
    set to the oracle's state at that point.  Free-running trajectories of two correct fp32 implementations decouple after 2-3
    steps for a reason that has nothing to do with either implementation: the reference keeps the median ratio inside the graph
    (online_adaption.py:295-298), whose backward puts -(rho / median) * sum(g * delta) -- a sum over ALL pixels -- on the ONE
    pixel that is the median; Adam's first update is lr * sign(g), so parameters whose gradient is at rounding level move by
    +-1e-5 differently, a few of those flips change WHICH pixel is the median, and from then on the two runs follow different
    (equally valid) paths -- measured here: plan path vs autograd path, same kernels, |w| differences of 1.5e-5 after one step
    turn into gradient differences of 0.27 in the next.  (The free-running 3-step trajectory of the first pair IS pinned, against
    the reference's own modules: test_first_pair_matches_reference_trajectory / golden g8; free-running GPU vs oracle over two
    keyframes: test_two_keyframes_free_running_vs_oracle.)

    480x640 = the benchmark's frame size (BASELINE configs[2]): the same six steps through the full-size launch plan -- every GEMM
    decomposition the bench uses, the 307 200-query KNN against the fused map (oracle: C brute force), the full-size PointFusion step."""
    from e2ehip import ops
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM
    from utils.training_utils import torch_poses_to_transforms
    L = 3
    seq = make_sequence(L, H, W, seed=7)
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0      # unique median element (see above)
    colors, gt, K, poses = seq
    # ---- oracle run with a snapshot of (weights, Adam state, map) BEFORE every step ------------------------------------------------
    ora = refine.Refiner(sd, refine.Config())
    snaps, orig = [], ora.opt.step

    def snapshot():
        st = ora.opt.state
        snaps.append({"w": {k: ora.sd[k].detach().clone() for k in ora.train_keys},
                      "m": {k: (st[ora.sd[k]]["exp_avg"].clone() if ora.sd[k] in st else torch.zeros_like(ora.sd[k])) for k in ora.train_keys},
                      "v": {k: (st[ora.sd[k]]["exp_avg_sq"].clone() if ora.sd[k] in st else torch.zeros_like(ora.sd[k])) for k in ora.train_keys},
                      "t": len(snaps), "map": {k: v.clone() for k, v in ora.map.items()}})

    def step_and_snapshot(*a, **k):
        out = orig(*a, **k)
        snapshot()
        return out
    ora.opt.step = step_and_snapshot
    snapshot()
    recs = []
    for a, b in ((0, 1), (1, 2)):
        recs += ora.refine_pair(colors[:, [a, b]], gt[:, [a, b]], poses[:, [a, b]], K)
    assert len(recs) == 6 and len(snaps) == 7
    map_pair2 = snaps[4]["map"]            # the first keyframe's map update runs after step 3: the map the second pair's 3-D loss sees
    assert snaps[3]["map"]["points"].shape[0] == 0 and map_pair2["points"].shape[0] >= H * W
    # ---- GPU, one step at a time from the oracle's state ----------------------------------------------------------------------------
    cfg = _cfg(H, W, L)
    cfg.DEBUG.print_metrics = False
    slam = SLAM(cfg, sequence=seq, state_dict=sd)
    slam.set_refinement_mode()
    sp = slam._step_plan()
    params = dict(slam.models["depth"].named_parameters())
    opt = slam.optimizer
    opt._resident_state()
    offs = {id(p): o for p, o in zip(opt.flat.params, opt.flat.offsets)}

    def load(snap):
        with torch.no_grad():
            for k in ora.train_keys:
                p, o = params[k], offs[id(params[k])]
                p.data.copy_(snap["w"][k])
                opt.m[o:o + p.numel()].copy_(snap["m"][k].reshape(-1))
                opt.v[o:o + p.numel()].copy_(snap["v"][k].reshape(-1))
            opt._counter.fill_(snap["t"] + 1)
        sp.net.refresh_layouts()

    step, agreed, gstats, failures = 0, [], [], []
    for pair, (a, b) in enumerate(((0, 1), (1, 2))):
        T = torch_poses_to_transforms(poses[:, [a, b]])[0, 1]
        sp.set_pair(slam.colors[0, a], slam.colors[0, b], slam.gt_depths[0, a], slam.gt_depths[0, b], slam.intrinsics[0, 0], T, slam.poses[0, b])
        sp.inv_K[0].copy_(torch.pinverse(slam.intrinsics[0, 0]))
        index = None
        if pair == 1:
            slam.map.load_state(map_pair2["points"].cuda(), map_pair2["normals"].cuda(), map_pair2["colors"].cuda(), map_pair2["ccounts"].cuda())
            index = slam.map.knn_index(H * W)
        for k in range(3):
            load(snaps[step])
            r = recs[step]
            sp.median_elements_override = torch.tensor(r["median_indices"], dtype=torch.int32, device="cuda")
            sp.step(k == 0, index)
            lp, lr, l3 = (float(v) for v in sp.losses())
            np.testing.assert_allclose(lp, r["photometric"], rtol=1e-4)
            np.testing.assert_allclose(lr, r["reg"], rtol=1e-4, atol=1e-9)
            np.testing.assert_allclose(float(sp.ratio), r["ratio"], rtol=1e-4)
            if pair == 1:
                np.testing.assert_allclose(l3, r["knn"], rtol=1e-4)
            total = lp + 1e-2 * lr + (l3 if pair == 1 else 0.0)
            np.testing.assert_allclose(total, r["loss"], rtol=1e-4)
            met = ops.depth_metrics(sp.gt[1], sp.depth[1], False).cpu().numpy()
            np.testing.assert_allclose(met, np.array(r["metrics"]), rtol=1e-4, atol=1e-6)
            # d loss / d depth of both frames (fused warp + SSIM + regulariser kernel, + the 3-D loss adjoint for the second pair) against
            # autograd's: within 1e-4 of the largest gradient at all but a handful of pixels (<= max(4, 1e-3 N)), and 5e-4 in the L2 norm
            # over the others (measured 1.1 - 1.6e-4 at 480x640, <= 1.2e-4 at 64x96).  The handful: a projection that lands within rounding of the image border (validity mask 0 / 1), of an
            # integer coordinate (bilinear tap set) or of an SSIM clamp -- the loss is discontinuous in its gradient there, two fp32
            # evaluations take different sides, and the difference at such a pixel is of the size of the gradient itself (measured at
            # 480x640: 38 - 194 of 307 200 pixels of the warped frame, none of the other frame, whose only term is the regulariser)
            for f in range(2):
                ga, gb = sp.g_depth[f, 0].cpu(), r["g_depth"][f][0, 0]
                err = (ga - gb).abs()
                out = err > 1e-4 * float(gb.abs().max())
                n_out = int(out.sum())
                rel2 = float(((ga - gb) * ~out).norm()) / (float(gb.norm()) + 1e-30)
                gstats.append((step, f, n_out, rel2))
                if n_out > max(4, 1e-3 * H * W) or rel2 > 5e-4:
                    failures.append(("g_depth", step, f, n_out, rel2))
            # parameter gradients, recovered from Adam's first moment (m' = b1 m + (1 - b1) g with m loaded from the oracle): every tensor
            # within GRAD_TOL of its largest element, the bound of the network-gradient tests -- here through the whole loss chain, in
            # EVERY step.  The reference keeps the median ratio inside the graph (online_adaption.py:295-298): its backward puts a sum over
            # all pixels on the ONE element torch.median names, and that element carries most of the parameter gradient.  Among 614 400
            # values the neighbours of the median are ~1e-6 apart, closer than two fp32 evaluations of a depth agree, so WHICH element
            # it is differs between two correct evaluations (round 3 compared gradients only where it happened to agree: 1 step of 6).
            # The choice is taken out of the comparison instead: the plan's median_elements_override names the oracle's elements (every
            # element equal to its median value: torch.median(x) shares the gradient among them, evenly_distribute_backward) as the
            # ones the gradient lands on (e2e_depth_scale_bwd_at), after checking that on the GPU too their values are the median to
            # within fp32 rounding of a depth; the median VALUE, the ratio and every loss term stay the GPU's own.
            # At 480x640 the bound is 2e-3: the kink pixels above feed the parameter gradients (measured 0.8 - 1.2e-3).
            md_at = (sp.delta.reshape(-1) == sp.md).nonzero().reshape(-1).tolist()
            same_median = sorted(md_at) == sorted(r["median_indices"])
            assert same_median or (H, W) != (64, 96), (step, r["median_indices"], sorted(md_at)[:4])
            named = sp.delta.reshape(-1)[torch.tensor(r["median_indices"], device="cuda")]
            assert float((named - sp.md).abs().max()) <= 1e-5 * abs(float(sp.md)), (step, named.tolist(), float(sp.md))
            # parameters after the step: Adam's update is lr * m^ / (sqrt(v^) + eps), sign-like in the first steps; where |g| is not far
            # above eps = 1e-8, or far below the tensor's largest gradient (relative error of g up to GRAD_TOL * max / |g|), the update
            # amplifies rounding-level differences of g to a sizeable part of lr.  So: NO element further than one update apart
            # (2 lr = 2e-5), and the well-conditioned elements (|g| >= 1 % of the tensor's maximum) equal to 3e-7 but for 0.2 % of them
            nxt, cur, bad, tot, worst, gworst = snaps[step + 1], snaps[step], 0, 0, 0.0, (0.0, "")
            for kname in ora.train_keys:
                p, o = params[kname], offs[id(params[kname])]
                g_gpu = (opt.m[o:o + p.numel()].view_as(p).cpu() - 0.9 * cur["m"][kname]) / 0.1
                g_ora = (nxt["m"][kname] - 0.9 * cur["m"][kname]) / 0.1
                gmax = float(g_ora.abs().max())
                gerr = float((g_gpu - g_ora).abs().max()) / max(gmax, 1e-30)
                if gerr > gworst[0]:
                    gworst = (gerr, kname)
                d = (p.detach().cpu() - nxt["w"][kname]).abs()
                well = g_ora.abs() >= 1e-2 * gmax
                bad += int((d[well] > 3e-7).sum())
                tot += int(well.sum())
                worst = max(worst, float(d.max()))
            agreed.append((same_median, gworst, worst, bad / tot))
            if worst > 2.2e-5 or gworst[0] > (TEACHER_FORCED_TOL if (H, W) == (64, 96) else 2e-3) or bad / tot >= 2e-3:
                failures.append(("parameters", step, same_median, gworst, worst, bad / tot))
            step += 1
        if pair == 0:                                   # the first keyframe's map update from the same weights (index tables: test_gpu_pointfusion_knn)
            load(snaps[3])
            sp.median_elements_override = None
            depth = sp.predict_depths()
            slam.first_iter = True
            slam._update_map(slam.colors[0, 0], slam.colors[0, 1], depth, slam.poses[0, 0], slam.poses[0, 1])
            slam.first_iter = False
            assert abs(slam.map.M - map_pair2["points"].shape[0]) <= 0.002 * slam.map.M
            # first frame's points: the depths behind them agree to ~1e-6, which flips an association / fusion decision for an isolated
            # point at 480x640 (with IDENTICAL inputs the tables are exact: tests/test_gpu_pointfusion_knn.py) -- allow 1e-5 of the points
            pa, pb = slam.map.live()[0][: H * W].cpu(), map_pair2["points"][: H * W]
            off = ((pa - pb).abs() > 1e-5 + 1e-4 * pb.abs()).any(1)
            assert int(off.sum()) <= 1e-5 * H * W, int(off.sum())
    print(f"[teacher forced {H}x{W}] per step (GPU's own median element == the oracle's, worst gradient error / tensor, worst parameter difference, fraction of "
          f"well-conditioned elements beyond 3e-7): {agreed}")
    print(f"[teacher forced {H}x{W}] d loss / d depth (step, frame, pixels beyond 1e-4 of max, relative L2 error over the others): {gstats}")
    assert not failures, failures


@pytest.mark.parametrize("H,W", [(64, 96), (480, 640)], ids=["64x96", "480x640"])
def test_default_head_first_step_vs_oracle(H, W):
    """Every other driver-level comparison scales the disparity head by 40 (a unique median, a well-conditioned loop).  This is the
    network as the reference initialises it (head x 1: a nearly flat disparity, near-ties around the median): ONE teacher-forced step
    from the initial state -- loss terms, ratio, d loss / d depth of both frames and all 48 parameter gradients (Adam's first moment
    after the first step is 0.1 g), the median elements named by the oracle (see the test above).

    The oracle runs in float64 here: with a flat disparity the fp32 CPU evaluation is itself 2e-3 off its own fp64 value on the worst
    tensor (the head's bias, ONE number that is a cancelling sum over all pixels -- measured on the CPU alone, 64 x 96), and the GPU is
    to be held to the better reference (tests/test_oracle_conditioning.py).  One-element tensors get that measured
    conditioning as their bound (1e-2: relative error of a single cancelling sum); every other tensor the usual 1e-4 / 1e-3 of its maximum."""
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM
    seq = make_sequence(2, H, W, seed=9)
    sd = depthnet.random_state_dict(0)
    colors, gt, K, poses = (t.double() for t in seq)
    ocfg = refine.Config()
    ocfg.refinement_steps = 1
    ora = refine.Refiner({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}, ocfg)
    grads = {}
    orig = ora.opt.step

    def keep_grads(*a, **k):
        grads.update({kk: ora.sd[kk].grad.detach().clone() for kk in ora.train_keys})
        return orig(*a, **k)
    ora.opt.step = keep_grads
    r = ora.refine_pair(colors, gt, poses, K, update_map=False)[0]
    cfg = _cfg(H, W, 2)
    cfg.DEBUG.print_metrics = False
    slam = SLAM(cfg, sequence=seq, state_dict=sd)
    slam.set_refinement_mode()
    sp = slam._step_plan()
    slam._load_pair(sp, 0, 1)
    assert len(r["median_indices"]) <= 64, len(r["median_indices"])
    sp.median_elements_override = torch.tensor(r["median_indices"], dtype=torch.int32, device="cuda")
    sp.step(True, None)
    lp, lr, _ = (float(v) for v in sp.losses())
    np.testing.assert_allclose(lp, r["photometric"], rtol=1e-4)
    np.testing.assert_allclose(lr, r["reg"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(float(sp.ratio), r["ratio"], rtol=1e-4)
    named = sp.delta.reshape(-1)[torch.tensor(r["median_indices"], device="cuda")]
    assert float((named - sp.md).abs().max()) <= 1e-5 * abs(float(sp.md))
    gstats = []
    for f in range(2):
        ga, gb = sp.g_depth[f, 0].cpu().double(), r["g_depth"][f][0, 0]
        out = (ga - gb).abs() > 1e-4 * float(gb.abs().max())
        gstats.append((f, int(out.sum()), float(((ga - gb) * ~out).norm()) / (float(gb.norm()) + 1e-30)))
    print(f"[default head {H}x{W}] d loss / d depth (frame, pixels beyond 1e-4 of max, relative L2 error over the others): {gstats}")
    # flat disparity: more projections sit within rounding of an integer coordinate / an SSIM clamp than with the x 40 head (3e-3 of the pixels)
    assert all(n <= max(4, 3e-3 * H * W) and e <= 5e-4 for _, n, e in gstats), gstats
    params, opt = dict(slam.models["depth"].named_parameters()), slam.optimizer
    offs = {id(p): o for p, o in zip(opt.flat.params, opt.flat.offsets)}
    errs = []
    for k in ora.train_keys:
        p = params[k]
        g_gpu = opt.m[offs[id(p)]:offs[id(p)] + p.numel()].view_as(p).cpu().double() / 0.1
        errs.append((float((g_gpu - grads[k]).abs().max()) / max(float(grads[k].abs().max()), 1e-30), k, p.numel()))
    errs.sort(reverse=True)
    print(f"[default head {H}x{W}] worst parameter-gradient errors / tensor max (vs the fp64 oracle): {errs[:4]}")
    tol = GRAD_TOL if (H, W) == (64, 96) else 1e-3
    assert all(e <= (1e-2 if n == 1 else tol) for e, _, n in errs), errs[:6]
    slam.close()


def _run_two_keyframes(mode, median_elements=None):
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM
    H, W, L = 64, 96, 3
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0
    slam = SLAM(_cfg(H, W, L), sequence=make_sequence(L, H, W, seed=11), state_dict=sd)
    slam.use_graphs = mode == "graphs"
    slam.median_elements = median_elements
    slam.median_elements_log = [] if median_elements is None else None
    slam.set_refinement_mode()
    slam.first_iter = True
    for prev, cur in slam.keyframe_schedule():
        (slam.refinement_autograd if mode == "autograd" else slam.refinement)(prev, cur)
        slam.first_iter = False
    torch.cuda.synchronize()
    return (torch.stack(slam.log), slam.map.live()[0].clone(), {k: v.detach().clone() for k, v in slam.models["depth"].state_dict().items()},
            slam.median_elements_log)


def test_launch_plan_equals_autograd_path_and_graph_replay_is_exact():
    """The static launch plan (e2ehip.stepplan) against torch.autograd over the per-layer Functions (same kernels), and its
    eager form against the captured-hipGraph form: the second and third step of every keyframe are graph replays."""
    # the graphs run names, step by step, WHICH element of its predictions was the median; the other two runs put the ratio's gradient on
    # the same elements (SLAM.median_elements).  For the eager run that changes nothing (it is the same arithmetic, checked bit for bit);
    # the autograd run differs from the plan in rounding -- enough, now and then, to make a neighbour 1e-7 away the median, after which
    # the two trajectories are different (equally valid) experiments and the comparison below would measure that, not the kernels
    log_g, map_g, sd_g, elems = _run_two_keyframes("graphs")
    assert len(elems) == 6 and all(e.numel() >= 1 for e in elems)
    log_e, map_e, sd_e, _ = _run_two_keyframes("eager", elems)
    log_a, map_a, sd_a, _ = _run_two_keyframes("autograd", elems)
    assert torch.equal(log_g, log_e) and torch.equal(map_g, map_e)          # replaying == launching, bit for bit
    for k in sd_g:
        assert torch.equal(sd_g[k], sd_e[k]), k
    # plan vs autograd: identical kernels; multi-consumer gradients are accumulated in a different order and the 3-D loss is a
    # masked mean instead of a mean over gathered rows -> agreement to a few fp32 ulps of each quantity
    # (columns 8-10 are a1 / a2 / a3: FRACTIONS OF PIXELS under a threshold -- a depth that differs in its last bit moves a pixel across
    # it, one pixel of 64 x 96 is 1.6e-4)
    cont = [c for c in range(log_g.shape[1]) if c not in (8, 9, 10)]
    np.testing.assert_allclose(log_g[:, cont].numpy(), log_a[:, cont].numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(log_g[:, 8:11].numpy(), log_a[:, 8:11].numpy(), rtol=0, atol=2.5 / (64 * 96))
    assert map_g.shape == map_a.shape
    torch.testing.assert_close(map_g, map_a, rtol=5e-5, atol=5e-6)     # the networks are up to 2 Adam steps of 1e-5 apart (below)
    for k in sd_g:
        if sd_g[k].dtype.is_floating_point:
            torch.testing.assert_close(sd_g[k], sd_a[k], rtol=0, atol=2.5e-5, msg=k)     # <= 2 Adam steps of lr 1e-5 apart where a gradient sign is at noise level


@pytest.mark.parametrize("H,W", [(64, 96), (480, 640)], ids=["64x96", "480x640"])
def test_tum_shaped_sequence_vs_oracle_first_keyframe(H, W):
    """BASELINE configs[3], in small and at its own 480x640: TUM intrinsics (fx = fy = 525, positive fy), 10 % zero-depth holes in the ground truth,
    DATA.name TUM (the holes are masked in depth_metrics only, losses.py:167-169; the median of the ground truth includes them,
    online_adaption.py:295), keyframe threshold 0.12: 3 refinement steps of the first keyframe against the oracle, then the map."""
    from e2ehip.synthetic import make_sequence, tum_intrinsics
    from online_adaption import SLAM
    L = 3
    seq = make_sequence(L, H, W, seed=13, step=0.13, K=tum_intrinsics(H, W), holes=0.1)
    assert float((seq[1] == 0).float().mean()) > 0.05
    sd = depthnet.random_state_dict(0)
    cfg = _cfg(H, W, L)
    cfg.DATA.name = "TUM"
    cfg.DEMO.frame_threshold = 0.12
    colors, gt, K, poses = seq
    ocfg = refine.Config()
    ocfg.dataset = "TUM"
    ora = refine.Refiner(sd, ocfg)
    recs = ora.refine_pair(colors[:, [0, 1]], gt[:, [0, 1]], poses[:, [0, 1]], K)
    slam = SLAM(cfg, sequence=seq, state_dict=sd)
    assert len(slam.keyframe_schedule()) == L - 1                    # 0.13 m steps: every frame is a keyframe at threshold 0.12
    slam.set_refinement_mode()
    slam.first_iter = True
    # the one discrete choice of a step -- which near-tied prediction is the median element -- is the oracle's (see the teacher-forced test)
    slam.median_elements = [torch.tensor(r["median_indices"][:64], dtype=torch.int32, device="cuda") for r in recs]
    slam.refinement(0, 1)
    log = torch.stack(slam.log)
    np.testing.assert_allclose(log[:, 1].numpy(), [r["photometric"] for r in recs], rtol=1e-4)
    np.testing.assert_allclose(log[:, 2].numpy(), [r["reg"] for r in recs], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(log[:, 3].numpy(), [r["ratio"] for r in recs], rtol=1e-4)
    np.testing.assert_allclose(log[:, 4:11].numpy(), np.array([r["metrics"] for r in recs]), rtol=2e-4, atol=1e-6)   # masked: holes excluded
    assert abs(slam.map.M - ora.map["points"].shape[0]) <= 0.002 * slam.map.M


@pytest.mark.parametrize("flags", [("geometric", "smoothness"), ("min_reprojection", "auto_masking"), ("auto_masking",),
                                   ("geometric", "smoothness", "min_reprojection", "auto_masking")])
def test_off_by_default_loss_flags_vs_oracle(flags):
    """online_adaption.py:486-523 with the flags the recommended configuration leaves off: geometric consistency (align_corners=True
    sampling of the source frame, :431-434), smoothness on the mean-normalised disparity of frame 0, minimum reprojection and
    auto-masking -- 3 refinement steps of the first keyframe, total loss / photometric part / regulariser / ratio vs the oracle.
    The reference's random tie-break noise (randn * 1e-5, :498) is switched off on both sides."""
    import online_adaption as oa
    from e2ehip.synthetic import make_sequence
    H, W, L = 64, 96, 2
    seq = make_sequence(L, H, W, seed=21)
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0       # unique median (see the teacher-forced test)
    cfg = _cfg(H, W, L)
    ocfg = refine.Config()
    for f in flags:
        setattr(cfg.LOSS, f, True)
        setattr(ocfg, f, True)
    cfg.LOSS.geometric_weight, cfg.LOSS.smoothness_weight = ocfg.geometric_weight, ocfg.smoothness_weight
    real = torch.randn
    try:
        torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: v for kk, v in k.items() if kk == "device"})
        slam = oa.SLAM(cfg, sequence=seq, state_dict=sd)
        slam.main()
    finally:
        torch.randn = real
    log = torch.stack(slam.log)
    colors, gt, K, poses = seq
    recs = refine.Refiner(sd, ocfg).refine_pair(colors, gt, poses, K)
    np.testing.assert_allclose(log[:, 0].numpy(), [r["loss"] for r in recs], rtol=2e-4)
    np.testing.assert_allclose(log[:, 1].numpy(), [r["photometric"] for r in recs], rtol=2e-4)
    np.testing.assert_allclose(log[:, 2].numpy(), [r["reg"] for r in recs], rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(log[:, 3].numpy(), [r["ratio"] for r in recs], rtol=1e-4)
    assert slam.map.M >= H * W


def test_map_update_forward_is_reused_as_the_next_keyframes_first_forward():
    """online_adaption.py:329-345 runs the refined network on the pair (p, c) for the map update; :281 of the NEXT keyframe (c, n) runs it on
    frame c again -- same weights (no optimiser step in between), same input.  The plan does not compute it twice:
      "source"   (no next pair known at map-update time): c's activations move from batch slot 1 to slot 0 and only n is forwarded
                 (RefineStepPlan._forward_new_target);
      "prefetch" (next pair known): n is forwarded NEXT TO the map step, on a second stream inside the captured map graph
                 (update_map(prefetch=True)); the next keyframe's first forward is the median scaling alone (_scale_only).
    Checked: (a) every activation, the scaled depths and the ratio of both shortcuts against the plain two-frame forward -- equal to fp32
    rounding (a one-image launch may split K differently); (b) three keyframes in each mode against recomputation: first keyframe
    bit-identical, later losses to 1e-5, map size to 0.2 %."""
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM
    H, W, L = 64, 96, 4
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0
    seq = make_sequence(L, H, W, seed=17)
    logs, maps = {}, {}
    for mode in ("prefetch", "source", "off"):
        slam = SLAM(_cfg(H, W, L), sequence=seq, state_dict=sd)
        slam.reuse_forward = mode != "off"
        slam.prefetch_forward = mode == "prefetch"
        slam.set_refinement_mode()
        slam.first_iter = True
        sched = slam.keyframe_schedule()
        assert sched == [(0, 1), (1, 2), (2, 3)]
        nxt = (lambda i: sched[i + 1] if (mode != "source" and i + 1 < len(sched)) else None)
        slam.refinement(*sched[0], next_pair=nxt(0))
        slam.first_iter = False
        sp = slam.step_plan
        if mode != "off":                                   # (a) at the hand-over point of keyframe 1 -> 2
            assert slam._forward_holds == ((1, 2) if mode == "prefetch" else (None, 1))
            if mode == "source":
                slam._load_pair(sp, 1, 2)
                sp._forward_new_target()
            else:
                sp._scale_only()
            torch.cuda.synchronize()
            short = {"acts": [op.out.t.clone() for op in sp.net.ops], "depth": sp.depth.clone(), "ratio": sp.ratio.clone()}
            sp._forward()
            torch.cuda.synchronize()
            for i, (act, op) in enumerate(zip(short["acts"], sp.net.ops)):
                for slot in (0, 1):
                    assert float((act[slot] - op.out.t[slot]).abs().max()) <= 2e-5 * float(op.out.t[slot].abs().max()) + 1e-12, (mode, i, slot)
            torch.testing.assert_close(short["depth"], sp.depth, rtol=2e-5, atol=1e-7)
            torch.testing.assert_close(short["ratio"], sp.ratio, rtol=2e-6, atol=0)
            slam._forward_holds = None                      # the slots now hold the plain forward of (1, 2): keyframe 2 takes the plain path,
            slam._preloaded = (1, 2)                        # keyframe 3 the shortcut again
        for i in (1, 2):
            slam.refinement(*sched[i], next_pair=nxt(i))
        logs[mode], maps[mode] = torch.stack(slam.log), slam.map.M
        if mode == "prefetch":
            assert any(isinstance(k, tuple) and k[0] == "map" and k[-1] is True for k in sp._graphs) and "scale_only" in sp._graphs
        slam.close()
    for mode in ("prefetch", "source"):
        assert torch.equal(logs[mode][:3], logs["off"][:3]), mode
        np.testing.assert_allclose(logs[mode][3:].numpy(), logs["off"][3:].numpy(), rtol=1e-5, atol=2.5 / (H * W))
        assert abs(maps[mode] - maps["off"]) <= 0.002 * maps["off"], (mode, maps)


def test_sparse_depth_supervision_runs():
    """LOSS.supervise_depth (random sparse sampling of the ground truth, so no value parity): the term is positive and enters the loss."""
    import online_adaption as oa
    from e2ehip.synthetic import make_sequence
    H, W, L = 64, 96, 2
    seq = make_sequence(L, H, W, seed=21)
    sd = depthnet.random_state_dict(0)
    logs = []
    for on in (False, True):
        cfg = _cfg(H, W, L)
        cfg.LOSS.smoothness = True                       # any flag that selects the operator-by-operator path
        cfg.LOSS.supervise_depth, cfg.LOSS.sampling_prob = on, 0.05
        slam = oa.SLAM(cfg, sequence=seq, state_dict=sd)
        slam.main()
        logs.append(torch.stack(slam.log)[:, 0])
    assert torch.isfinite(logs[1]).all() and logs[1][0] > logs[0][0]


def test_split_backward_graphs_reproduce_the_single_pass_gradient():
    """The data-parallel form of a step runs the backward pass as TWO captured graphs (head / decoder / layer4, then layer3 ... stem,
    e2ehip.stepplan.step) so that the first segment's all-reduce travels under the second.  Needs no torch.distributed: both halves go
    through RefineStepPlan._run (eager execution + capture, then replays) for three steps with frozen weights, and the flat gradient
    bucket must equal the one-pass backward bit for bit every time -- the accumulate flags of the early half are launch arguments
    frozen into the graph and must not depend on how often either half ran before (round-2 defect: the capture pass saw the flags
    the eager pass had left and recorded `accumulate = 1` for the first contribution to 12 gradient buffers)."""
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM
    H, W, L = 64, 96, 2
    cfg = _cfg(H, W, L)
    cfg.DEBUG.print_metrics = False
    slam = SLAM(cfg, sequence=make_sequence(L, H, W, seed=5), state_dict=depthnet.random_state_dict(0))
    slam.set_refinement_mode()
    sp = slam._step_plan()
    slam._load_pair(sp, 0, 1)
    flat = slam.optimizer.flat
    sp._run("fwd", sp._forward)
    sp.init.copy_(sp.delta)
    sp.use_graphs = False
    flat.grad.zero_()
    sp._backward(False, False)                          # reference: late + early layers in one eager pass, no Adam
    torch.cuda.synchronize()
    ref = flat.grad.clone()
    assert float(ref.abs().sum()) > 0
    sp.use_graphs = True
    for it in range(3):                                 # it == 0: eager + capture; 1, 2: replays
        flat.grad.zero_()
        sp._run(("bwd_late", False), lambda: sp._backward(False, False, late_only=True))
        sp._run("bwd_early", sp.net.backward_early_layers)
        torch.cuda.synchronize()
        assert torch.equal(flat.grad, ref), (it, float((flat.grad - ref).abs().max()))
    assert ("bwd_late", False) in sp._graphs and "bwd_early" in sp._graphs


@pytest.mark.parametrize("head_scale,dtype,tol", [(40.0, torch.float32, 1e-4), (1.0, torch.float64, 1e-4)], ids=["well_conditioned_vs_fp32_oracle", "flat_disparity_vs_fp64_oracle"])
def test_two_keyframes_free_running_vs_oracle(head_scale, dtype, tol):
    """FREE-RUNNING: both keyframe pairs of a 3-frame sequence through SLAM.main() -- six refinement steps, the second pair with the 3-D
    nearest-neighbour loss against the map the first pair fused, no state reset anywhere -- against the oracle's own free run: photometric
    loss, regulariser, median ratio, 3-D loss and total loss of EVERY step, and the map size.

    This is the comparison round 2 replaced by the teacher-forced test above after it had failed in a mid-round build.  It holds (measured:
    1e-5 on every quantity), and tests/test_oracle_conditioning.py shows when it has to: with the head weights x 40 the loop is well
    conditioned (the oracle agrees with its own fp64 evaluation to 5e-6), with the default initialisation it is not (fp32 vs fp64 oracle:
    1e-3 after five steps) -- there the GPU path is compared with the fp64 oracle, which it follows to ~3e-5 (closer than the fp32 CPU
    oracle does)."""
    from test_oracle_conditioning import run_two_keyframes
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM
    H, W, L = 64, 96, 3
    recs = run_two_keyframes(dtype, head_scale)
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * head_scale
    slam = SLAM(_cfg(H, W, L), sequence=make_sequence(L, H, W, seed=7), state_dict=sd)
    # free-running in everything but ONE discrete choice: which of the (near-)tied predictions is the median element, i.e. where the
    # ratio's gradient lands, is the oracle's at every step (its own trajectory's torch.median; see the teacher-forced test)
    slam.median_elements = [torch.tensor(r["median_indices"][:64], dtype=torch.int32, device="cuda") for r in recs]
    slam.main()
    log = torch.stack(slam.log).double().numpy()
    assert log.shape[0] == 6
    np.testing.assert_allclose(log[:, 1], [r["photometric"] for r in recs], rtol=tol)
    np.testing.assert_allclose(log[:, 2], [r["reg"] for r in recs], rtol=tol, atol=1e-9)
    np.testing.assert_allclose(log[:, 3], [r["ratio"] for r in recs], rtol=tol)
    np.testing.assert_allclose(log[3:, 11], [r["knn"] for r in recs[3:]], rtol=3 * tol)      # mean of ~6000 squared distances of 1e-3 .. 1e-2
    np.testing.assert_allclose(log[:, 0], [r["loss"] for r in recs], rtol=tol)
