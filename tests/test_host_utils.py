"""Host-side pieces that need no GPU: PLY export round trip, and the N3/N4 entry points refusing CPU tensors (there is no
CPU fallback anywhere in the product path)."""
import numpy as np
import pytest
import torch


def test_ply_roundtrip_cpu(tmp_path):
    from utils.export import load_ply, save_ply
    g = torch.Generator().manual_seed(2)
    pts = torch.randn(257, 3, generator=g)
    col = torch.rand(257, 3, generator=g)                               # 0..1 colours are scaled to 0..255
    rec = load_ply(save_ply(str(tmp_path / "m.ply"), pts, col))
    assert rec.dtype.names == ("x", "y", "z", "red", "green", "blue") and rec.shape[0] == 257
    np.testing.assert_array_equal(np.stack([rec["x"], rec["y"], rec["z"]], 1), pts.numpy())
    np.testing.assert_array_equal(rec["green"], np.clip(np.rint(col.numpy()[:, 1] * 255.0), 0, 255).astype(np.uint8))
    with pytest.raises(ValueError):
        save_ply(str(tmp_path / "bad.ply"), torch.zeros(4, 2))


def test_new_entry_points_have_no_cpu_fallback():
    from e2ehip import ops
    from e2ehip._lib import E2EError
    from e2ehip.tensor_refine import learn_depth_scale, refine_depth_tensor
    d = torch.rand(1, 1, 8, 8) + 0.5
    img = torch.rand(1, 3, 8, 8)
    K = torch.eye(4).reshape(1, 4, 4)
    for call in (lambda: ops.smoothness(d, img), lambda: ops.geometric_consistency(d, d, torch.ones_like(d)),
                 lambda: ops.masked_l1(d, d[0, 0], torch.ones(8, 8)), lambda: ops.min_reprojection(torch.rand(1, 2, 8, 8)),
                 lambda: ops.process_disparity(torch.rand(2, 1, 8, 8)), lambda: ops.KnnIndex(torch.rand(10, 3), 4),
                 lambda: refine_depth_tensor(d, img, img, K, K), lambda: learn_depth_scale(d, img, img, K, K, steps=1)):
        with pytest.raises(E2EError):
            call()
