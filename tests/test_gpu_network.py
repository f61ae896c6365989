"""Depth network on the GPU against the golden vectors captured from the reference's DispResNet_Indoor."""
import pytest
import torch

from oracle import depthnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model():
    from depth_estimation.networks import DispResNet_Indoor
    m = DispResNet_Indoor(18, False)
    m.load_state_dict(depthnet.random_state_dict(0))
    m.to(DEV).eval()
    for name, p in m.named_parameters():
        if name.find("bn") != -1:
            p.requires_grad = False
    return m


def test_forward_backward_vs_golden(golden):
    g = golden("g7_net")
    m = _model()
    out = m(g["img"].to(DEV), 0)
    disp = out[("disp", 0, 0)]
    assert list(out.keys()) == [("disp", 0, 0)]
    for i, f in enumerate(m.encoder.features):
        torch.testing.assert_close(f.cpu(), g[f"f{i}"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(disp.cpu(), g["disp"], rtol=1e-4, atol=1e-5)
    (disp * g["wgt"].to(DEV)).sum().backward()
    names = list(g["grad_names"])
    params = dict(m.named_parameters())
    for n, ref in zip(names, g["grad_norms"]):
        if ref < 0:
            assert params[n].grad is None
        else:
            torch.testing.assert_close(params[n].grad.norm().cpu(), ref, rtol=1e-3, atol=1e-7)
    for k in [k for k in g if k.startswith("gs_")]:
        name = [n for n in names if "gs_" + n.replace(".", "_") == k][0]
        ref = g[k]
        got = params[name].grad.flatten()[:16].cpu()
        assert (got - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-7, name


def test_batch_of_two_equals_two_calls(golden):
    """BN runs in eval mode, so the keyframe pair can go through the network as ONE batch of 2."""
    g = golden("g8_refine")
    m = _model()
    colors = g["colors"].to(DEV)
    with torch.no_grad():
        a = m(colors[:, 0], 0)[("disp", 0, 0)]
        b = m(colors[:, 1], 1)[("disp", 1, 0)]
        both = m(colors[0], 0)[("disp", 0, 0)]
    torch.testing.assert_close(both, torch.cat([a, b], 0), rtol=1e-5, atol=1e-6)
