"""The build's online_adaption driver (fused launches) against the oracle's refinement loop and the golden
trajectory captured from the reference's own modules (tests/golden/g8)."""
import numpy as np
import pytest
import torch

from oracle import depthnet, refine

pytestmark = pytest.mark.gpu


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


def test_two_pairs_with_map_and_3d_loss_vs_oracle():
    """second keyframe pair: 3-D nearest-neighbour loss against the fused map, all loss terms vs the oracle."""
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM
    H, W, L = 64, 96, 3
    seq = make_sequence(L, H, W, seed=7)
    sd = depthnet.random_state_dict(0)
    # A random-init network predicts an almost constant disparity, so 1/disp collides on a few thousand fp32 values
    # and the MEDIAN ELEMENT IS TIED; torch.median then routes its gradient to an implementation-defined one of the
    # tied elements (CPU nth_element vs any GPU select differ), which no implementation can be "bit-compatible"
    # with.  Spread the head's output so the median is unique, as it is for a trained network.
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0
    slam = SLAM(_cfg(H, W, L), sequence=seq, state_dict=sd)
    slam.main()
    log = torch.stack(slam.log)                      # rows: total, photometric, reg, ratio, 7 metrics, knn
    colors, gt, K, poses = seq
    ora = refine.Refiner(sd, refine.Config())
    recs = []
    for a, b in ((0, 1), (1, 2)):
        recs += ora.refine_pair(colors[:, [a, b]], gt[:, [a, b]], poses[:, [a, b]], K)
    assert len(recs) == log.shape[0] == 6
    np.testing.assert_allclose(log[:, 1].numpy(), [r["photometric"] for r in recs], rtol=2e-4)
    np.testing.assert_allclose(log[:, 2].numpy(), [r["reg"] for r in recs], rtol=2e-3, atol=1e-9)
    np.testing.assert_allclose(log[:, 3].numpy(), [r["ratio"] for r in recs], rtol=1e-4)
    np.testing.assert_allclose(log[3:, 11].numpy(), [r["knn"] for r in recs[3:]], rtol=2e-3)
    np.testing.assert_allclose(log[:, 0].numpy(), [r["loss"] for r in recs], rtol=5e-4)
    np.testing.assert_allclose(log[:, 4:11].numpy(), np.array([r["metrics"] for r in recs]), rtol=2e-3, atol=1e-6)
    assert abs(slam.map.M - ora.map["points"].shape[0]) <= 0.002 * slam.map.M
