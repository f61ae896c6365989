import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import depthnet, refine
from e2ehip import _lib as L
from e2ehip.synthetic import make_sequence
from online_adaption import SLAM, default_config
lib = L.load()
H, W, Ln = 64, 96, 3
seq = make_sequence(Ln, H, W, seed=7)
sd = depthnet.random_state_dict(0)
sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0
colors, gt, K, poses = seq
ora = refine.Refiner(sd, refine.Config())
recs = []
for a, b in ((0, 1), (1, 2)):
    recs += ora.refine_pair(colors[:, [a, b]], gt[:, [a, b]], poses[:, [a, b]], K)
want = np.array([r["photometric"] for r in recs])
for force, mode, graphs, overlap in (((0, 0, 0), "plan", True, True), ((0, 0, 0), "plan", False, True), ((0, 0, 0), "plan", True, False), ((0, 0, 0), "plan", False, False),
                                     ((64, 64, 1), "plan", True, True), ((64, 64, 2), "plan", True, True), ((0, 0, 0), "plan", True, True)):
    lib.e2e_conv_gemm_force(*force)
    cfg = default_config(H, W, Ln); cfg.DEMO.frame_threshold = 0.0
    slam = SLAM(cfg, sequence=seq, state_dict=sd)
    slam.use_graphs, slam.overlap_wgrad = graphs, overlap
    slam.set_refinement_mode(); slam.first_iter = True
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        for prev, cur in slam.keyframe_schedule():
            (slam.refinement_autograd if mode == "autograd" else slam.refinement)(prev, cur)
            slam.first_iter = False
    log = torch.stack(slam.log)
    print(force, mode, "graphs", graphs, "overlap", overlap, "photometric rel err per step:", np.abs(log[:, 1].numpy() - want) / want, flush=True)
lib.e2e_conv_gemm_force(0, 0, 0)
