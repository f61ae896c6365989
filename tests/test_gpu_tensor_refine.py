"""N4: tensor optimisation (OFT), scale learning and map export."""
import numpy as np
import pytest
import torch

from oracle import warp_loss as O
from synth import make_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(H=96, W=128):
    s = make_pair(H, W, seed=3)
    t = {k: v.to(DEV).contiguous() for k, v in s.items()}
    return s, t, t["src"].permute(0, 3, 1, 2), t["tgt"].permute(0, 3, 1, 2)


def test_oft_matches_cpu_adam_on_the_oracle_loss():
    """3 Adam steps on the depth tensor itself: GPU (fused lossgrad + fused Adam) vs torch autograd + torch.optim.Adam."""
    from e2ehip.tensor_refine import refine_depth_tensor
    s, t, src, tgt = _pair()
    d0 = s["depth"] * 1.05
    refined, trace = refine_depth_tensor(d0.to(DEV), src, tgt, t["K"], t["T"], steps=3, lr=2e-3)
    p = torch.nn.Parameter(d0.clone())
    opt = torch.optim.Adam([p], lr=2e-3)
    ref_trace = []
    for _ in range(3):
        opt.zero_grad()
        synth, valid = O.inverse_warp(p, s["src"].permute(0, 3, 1, 2), s["K"], s["invK"], s["T"], "border")[:2]
        loss = O.masked_photometric_mean(synth, s["tgt"].permute(0, 3, 1, 2), valid, True)[0]
        loss.backward()
        opt.step()
        ref_trace.append(float(loss.detach()))
    np.testing.assert_allclose(trace, ref_trace, rtol=1e-4)
    # Adam normalises the step, so a gradient that is off by rounding where it is ~0 still moves that pixel by lr:
    # compare where the oracle's gradient is not negligible
    moved = (p.detach() - d0).abs() > 1e-4
    agree = ((refined.cpu() - p.detach()).abs() < 2e-3) | ~moved
    assert float(agree.float().mean()) > 0.995


def test_oft_reduces_the_loss_and_scale_learning_recovers_the_scale():
    """A photo-consistent pair (the target is the source warped with the true depth): the loss is ~0 at the true depth,
    OFT pulls a perturbed depth back towards it and the scale layer finds the factor the prediction is off by."""
    from e2ehip.tensor_refine import learn_depth_scale, refine_depth_tensor
    s, t, src, _ = _pair()
    tgt_cpu = O.inverse_warp(s["depth"], s["src"].permute(0, 3, 1, 2), s["K"], s["invK"], s["T"], "border")[0]
    tgt = tgt_cpu.permute(0, 2, 3, 1).contiguous().to(DEV).permute(0, 3, 1, 2)          # NHWC memory, NCHW view
    _, at_truth = refine_depth_tensor(s["depth"].to(DEV), src, tgt, t["K"], t["T"], steps=1, lr=1e-9)
    _, trace = refine_depth_tensor((s["depth"] * 1.08).to(DEV), src, tgt, t["K"], t["T"], steps=60, lr=5e-3)
    assert at_truth[0] < 0.05 * trace[0] and trace[-1] < 0.6 * trace[0]
    w, b, tr = learn_depth_scale((s["depth"] / 0.7).to(DEV), src, tgt, t["K"], t["T"], steps=200, lr=1e-2, init_value=0.5)
    assert abs(w - 0.7) < 0.03 and tr[-1] < 0.3 * tr[0]
    w2, b2, tr2 = learn_depth_scale((s["depth"] / 0.7).to(DEV), src, tgt, t["K"], t["T"], steps=200, lr=1e-2, init_value=0.5, affine=True)
    assert tr2[-1] < 0.5 * tr2[0] and np.isfinite([w2, b2]).all()


def test_map_export_roundtrip(tmp_path):
    from utils.export import load_ply, save_ply
    g = torch.Generator().manual_seed(1)
    pts = torch.randn(1000, 3, generator=g).to(DEV)
    col = (torch.rand(1000, 3, generator=g) * 255).to(DEV)
    nrm = torch.nn.functional.normalize(torch.randn(1000, 3, generator=g), dim=1).to(DEV)
    rec = load_ply(save_ply(str(tmp_path / "map.ply"), pts, col, nrm))
    assert rec.shape[0] == 1000
    np.testing.assert_array_equal(np.stack([rec["x"], rec["y"], rec["z"]], 1), pts.cpu().numpy())
    np.testing.assert_array_equal(np.stack([rec["red"], rec["green"], rec["blue"]], 1), np.rint(col.cpu().numpy()).astype(np.uint8))
    np.testing.assert_array_equal(rec["nz"], nrm.cpu().numpy()[:, 2])
