"""CPU-side checks of the depth network's structure: the 150 state-dict keys / shapes of the reference
(captured in tests/golden/g7 from the reference's own DispResNet_Indoor), the refinement-mode freezing rule,
and that there is no CPU execution path."""
import pytest
import torch


def test_state_dict_layout_matches_reference(golden):
    from depth_estimation.networks import DispResNet_Indoor
    g = golden("g7_net")
    m = DispResNet_Indoor(num_layers=18, pretrained=False)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert sum(p.numel() for p in m.parameters()) == 14842236
    # online_adaption.py:175-184: freeze by substring "bn"
    for name, p in m.named_parameters():
        if name.find("bn") != -1:
            p.requires_grad = False
    assert sum(p.numel() for p in m.parameters() if not p.requires_grad) == 7808
    assert list(m.encoder.num_ch_enc) == [64, 64, 128, 256, 512]
    # train_depth.py:161-169 introspects decoder.decoder[i].conv
    assert all(hasattr(b, "conv") for b in m.decoder.decoder)
    with pytest.raises(RuntimeError):
        DispResNet_Indoor(num_layers=18, pretrained=True)
    with pytest.raises(ValueError):
        from depth_estimation.networks import ResnetEncoder
        ResnetEncoder(19, False)


def test_network_refuses_cpu():
    from depth_estimation.networks import DispResNet_Indoor
    from e2ehip import E2EError
    m = DispResNet_Indoor(18, False).eval()
    with pytest.raises(E2EError):
        m(torch.rand(1, 32, 32, 3), 0)
