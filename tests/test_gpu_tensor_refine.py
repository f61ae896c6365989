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


def test_scale_learning_and_grid_search_match_cpu_adam_on_the_oracle_loss():
    """absolute_scale.py:207-240, :268: for every initial scale of the grid a fresh Conv1x1(1, 1, bias=True) (w, b) on top of the frozen
    depth prediction, trained by Adam through the photometric loss -- GPU (fused lossgrad + <g, d> / sum(g) + fused Adam) against torch
    autograd + torch.optim.Adam on the oracle's loss: loss trajectory and the learnt (w, b) of every grid point."""
    from e2ehip.tensor_refine import scale_grid_search
    s, t, src, tgt = _pair()
    base = s["depth"] / 0.8
    grid, steps, lr, b0 = (0.5, 0.8, 1.1), 6, 1e-2, 0.03
    got = scale_grid_search(base.to(DEV), src, tgt, t["K"], t["T"], grid, steps=steps, lr=lr, affine=True, init_bias=b0)
    assert [g["init"] for g in got] == list(grid)
    for rec in got:
        w = torch.nn.Parameter(torch.tensor([rec["init"]]))
        b = torch.nn.Parameter(torch.tensor([b0]))
        opt = torch.optim.Adam([w, b], lr=lr)
        ref = []
        for _ in range(steps):
            opt.zero_grad()
            d = base * w + b
            synth, valid = O.inverse_warp(d, s["src"].permute(0, 3, 1, 2), s["K"], s["invK"], s["T"], "border")[:2]
            loss = O.masked_photometric_mean(synth, s["tgt"].permute(0, 3, 1, 2), valid, True)[0]
            loss.backward()
            opt.step()
            ref.append(float(loss.detach()))
        np.testing.assert_allclose(rec["losses"], ref, rtol=1e-4)
        # the parameters travelled steps x lr = 0.06: agreement to 3e-4 of the distance covered (b crosses zero, so no relative test on b)
        np.testing.assert_allclose([rec["w"], rec["b"]], [float(w.detach()), float(b.detach())], rtol=0, atol=3e-4 * steps * lr)
    assert len({round(g["losses"][0], 6) for g in got}) == 3 and all(np.isfinite(g["losses"]).all() for g in got)     # three different experiments


def test_fused_adam_state_round_trip_and_load_optimizer(tmp_path):
    """train_depth.py:849-863: the optimiser resumes from `<load_depth_path>/Adam.pth`.  FusedAdam writes and reads torch.optim.Adam's
    state-dict format: (a) save after 3 steps, load into a fresh FusedAdam over equal parameters, one more step on both -> bitwise equal
    parameters; (b) the same file loads into torch.optim.Adam on the CPU and its next step agrees; (c) Depth_Estimation.load_optimizer."""
    from e2ehip.optim import FusedAdam
    g = torch.Generator().manual_seed(4)
    shapes = [(7, 5), (33,), (4, 3, 3, 3)]
    grads = [[torch.randn(*sh, generator=g) for sh in shapes] for _ in range(4)]

    def make():
        gg = torch.Generator().manual_seed(8)
        ps = [torch.nn.Parameter(torch.randn(*sh, generator=gg).to(DEV)) for sh in shapes]
        ps.insert(1, torch.nn.Parameter(torch.randn(3, generator=gg).to(DEV), requires_grad=False))        # a frozen parameter in the list
        return ps

    def step(opt, ps, k):
        opt.zero_grad()
        for p, gr in zip([p for p in ps if p.requires_grad], grads[k]):
            if p.grad is None:
                p.grad = gr.to(p.device).clone()
            else:
                p.grad.copy_(gr.to(p.device))
        opt.step()

    pa = make()
    oa = FusedAdam(pa, lr=3e-3)
    for k in range(3):
        step(oa, pa, k)
    path = str(tmp_path / "Adam.pth")
    torch.save(oa.state_dict(), path)
    sd = torch.load(path, map_location=DEV)
    assert sorted(sd["state"]) == [0, 2, 3] and int(sd["state"][0]["step"]) == 3 and sd["param_groups"][0]["params"] == [0, 1, 2, 3]
    pb = make()
    with torch.no_grad():
        for a, b in zip(pa, pb):
            b.copy_(a)
    ob = FusedAdam(pb, lr=1.0)                           # the loaded group overrides lr
    ob.load_state_dict(sd)
    # (b) torch.optim.Adam reads the same file
    pc = [torch.nn.Parameter(p.detach().cpu().clone(), requires_grad=p.requires_grad) for p in pa]
    oc = torch.optim.Adam(pc, lr=1.0)
    oc.load_state_dict(torch.load(path, map_location="cpu"))
    step(oa, pa, 3)
    step(ob, pb, 3)
    step(oc, pc, 3)
    torch.cuda.synchronize()
    for a, b, c in zip(pa, pb, pc):
        assert torch.equal(a, b)
        torch.testing.assert_close(a.detach().cpu(), c.detach(), rtol=1e-6, atol=1e-7)
    assert ob.steps_done() == 4 and ob.param_groups[0]["lr"] == 3e-3


def test_fused_adam_loads_a_partial_torch_adam_state():
    """a file written by the reference's torch.optim.Adam holds state only for the parameters that had received gradients.
    Round 3 laid the flat bucket out over exactly those and then never updated any other parameter.  Now the bucket is decided by the
    gradients / prebuild() as always and the file is applied to it: same gradient pattern -> the next step equals torch's; a trained
    parameter the file does not know, next to others at step 2 -> ValueError (one step counter cannot run it), never a silent skip."""
    from e2ehip.optim import FusedAdam
    g = torch.Generator().manual_seed(12)
    shapes = [(6, 4), (9,), (2, 3, 3, 3), (5,)]
    init = [torch.randn(*sh, generator=g) for sh in shapes]
    grads = [[torch.randn(*sh, generator=g) for sh in shapes] for _ in range(3)]
    with_grad = (0, 2)                                              # parameters 1 and 3 never see a gradient in the saved run

    def cpu_run(nsteps):
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        opt = torch.optim.Adam(ps, lr=2e-3)
        for k in range(nsteps):
            opt.zero_grad()
            for i in with_grad:
                ps[i].grad = grads[k][i].clone()
            opt.step()
        return ps, opt

    ps2, o2 = cpu_run(2)
    sd = o2.state_dict()
    assert sorted(sd["state"]) == list(with_grad)
    ps3, _ = cpu_run(3)

    def gpu(params_with_grad):
        ps = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ps2]
        opt = FusedAdam(ps, lr=1.0)
        opt.load_state_dict(sd)
        assert sorted(opt.state_dict()["state"]) == list(with_grad)          # loaded, not applied yet: round-trips unchanged
        for i in params_with_grad:
            ps[i].grad = grads[2][i].to(DEV).clone()
        return ps, opt

    ps, opt = gpu(with_grad)
    opt.step()
    torch.cuda.synchronize()
    for a, b in zip(ps, ps3):
        torch.testing.assert_close(a.detach().cpu(), b.detach(), rtol=1e-6, atol=1e-7)
    assert opt.steps_done() == 3 and sorted(opt.state_dict()["state"]) == list(with_grad)
    ps, opt = gpu((0, 1, 2))                                        # parameter 1 trains now but the file has no state for it
    with pytest.raises(ValueError, match="carry no state"):
        opt.step()
    # prebuilt but never stepped: no state entries, like torch; several parameter groups: refused, not half-honoured
    ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    opt = FusedAdam(ps, lr=1e-3)
    opt.prebuild(ps)
    assert opt.state_dict()["state"] == {}
    with pytest.raises(NotImplementedError):
        FusedAdam([{"params": ps[:2]}, {"params": ps[2:], "lr": 1e-4}], lr=1e-3)


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
