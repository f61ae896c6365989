"""How well-conditioned is a free-running refinement trajectory?  (CPU only: the oracle against ITSELF in another precision.)

Round 2 replaced a free-running two-keyframe GPU-vs-oracle comparison by a teacher-forced one, arguing that two correct fp32 evaluations
of this loop decouple after two or three steps (the median ratio's backward puts a whole-image sum on ONE pixel, Adam's first steps are
sign-like).  This file measures that claim on the oracle alone -- the same two keyframes (seed-7 sequence, 64 x 96) run in fp32 and in fp64
(weights, activations, losses, Adam; the PointFusion map stays fp32 in both):

  * head weights x 40 (disparities spread over their whole range, a unique median element -- the case the GPU tests use): the fp32 and
    the fp64 trajectory agree to ~5e-6 over all six steps, 3-D loss included.  The loop is WELL-conditioned there, so a free-running
    comparison is a meaningful test and tests/test_gpu_driver.py::test_two_keyframes_free_running_vs_oracle holds the GPU to 1e-4;
  * default initialisation (disparity nearly constant over the frame, near-ties around the median): fp32 and fp64 drift apart by 2e-4
    after ONE update and 1e-3 after five.  There a free-running fp32 trajectory is not pinned by the arithmetic to better than that --
    which is the regime the round-2 argument describes -- and the GPU test compares against the fp64 oracle instead (the GPU path
    follows the fp64 trajectory to ~3e-5 there, closer than the fp32 CPU oracle does)."""
import os
import sys

import numpy as np
import torch

from oracle import depthnet, refine


def run_two_keyframes(dtype, head_scale, seed=7, H=64, W=96):
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "end-to-end-self-supervised-slam_amd")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    from e2ehip.synthetic import make_sequence
    colors, gt, K, poses = (t.to(dtype) for t in make_sequence(3, H, W, seed=seed))
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * head_scale
    sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    ora = refine.Refiner(sd, refine.Config())
    recs = []
    for a, b in ((0, 1), (1, 2)):
        recs += ora.refine_pair(colors[:, [a, b]], gt[:, [a, b]], poses[:, [a, b]], K)
    return recs


def _spread(r32, r64, key):
    return [abs(a[key] - b[key]) / abs(b[key]) for a, b in zip(r32, r64) if key in a]


def test_well_conditioned_case_fp32_equals_fp64():
    r32, r64 = run_two_keyframes(torch.float32, 40.0), run_two_keyframes(torch.float64, 40.0)
    assert len(r32) == 6 and "knn" in r32[3]
    for key in ("photometric", "reg", "ratio", "knn", "loss"):
        assert max(_spread(r32, r64, key)) < 5e-5, (key, _spread(r32, r64, key))


def test_flat_disparity_case_is_ill_conditioned():
    r32, r64 = run_two_keyframes(torch.float32, 1.0), run_two_keyframes(torch.float64, 1.0)
    s = _spread(r32, r64, "photometric")
    assert s[0] < 5e-5                                   # the first evaluation (no update yet) agrees
    assert s[1] > 5e-5 and max(s) > 3e-4, s              # ... and one sign-like Adam update later the two precisions have parted
    np.testing.assert_allclose([r["photometric"] for r in r32], [r["photometric"] for r in r64], rtol=5e-3)     # (still the same experiment)
