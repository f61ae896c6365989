"""Native fp32-MFMA convolution kernels against a float64 torch composition of the same operator
(conv + folded eval-BN / bias + residual + activation over cat(upsample(x), skip) with zero / reflection padding)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, skip, w, bias, scale, shift, res, up, stride, pad, pad_mode, act, in_norm):
    x = x.double()
    if in_norm is not None:
        x = (x - in_norm[0]) * in_norm[1]
    if up != 1:
        x = F.interpolate(x, scale_factor=up, mode="nearest")
    if skip is not None:
        x = torch.cat([x, skip.double()], 1)
    if pad_mode == "reflect" and pad:
        x = F.pad(x, (pad,) * 4, mode="reflect")
        y = F.conv2d(x, w.double(), None, stride, 0)
    else:
        y = F.conv2d(x, w.double(), None, stride, pad)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    if bias is not None:
        y = y + bias.double().view(1, -1, 1, 1)
    if res is not None:
        y = y + res.double()
    if act == "relu":
        y = F.relu(y)
    elif act == "elu":
        y = F.elu(y)
    return y


def _mask_kinks(gy, yr, act):
    """ReLU / ELU are only piecewise differentiable: an output whose pre-activation is within rounding of 0 may sit on different
    sides of the kink in the fp32 kernel and in the fp64 reference (a different summation order -- e.g. another split-K factor --
    moves which ones do), and ONE such element changes 9 x Cin input-gradient values by O(1).  The upstream gradient is zeroed
    at those outputs, so the comparison tests the arithmetic and not the coin flips (DESIGN.md section 5 documents the same
    exclusion for the warp kernel's bilinear kinks)."""
    if act not in ("relu", "elu"):
        return gy
    near = yr.detach().abs() < 1e-4 * yr.detach().abs().max()            # relu: output ~ 0+; elu: output ~ 0 on either side
    return torch.where(near.to(gy.device), torch.zeros_like(gy), gy)


CASES = [
    # B, Cin(x), Cskip, up, H,  W,  Cout, k, s, p, pad_mode, act,   bn,    bias,  res
    (2, 64, 0, 1, 12, 20, 64, 3, 1, 1, "zeros", "relu", True, False, True),      # BasicBlock conv2 + residual
    (2, 64, 0, 1, 12, 20, 128, 3, 2, 1, "zeros", "relu", True, False, False),     # stride-2 stage entry
    (1, 64, 0, 1, 13, 9, 128, 1, 2, 0, "zeros", None, False, False, False),       # 1x1/2 downsample, odd sizes
    (2, 3, 0, 1, 32, 48, 64, 7, 2, 3, "zeros", "relu", True, False, False),       # RGB stem (scalar gather + input normalisation)
    (2, 512, 0, 1, 2, 3, 256, 3, 1, 1, "reflect", "elu", False, True, False),     # upconv(4,0) at the smallest size
    (2, 256, 256, 2, 4, 6, 256, 3, 1, 1, "reflect", "elu", False, True, False),   # upconv(4,1): upsample + skip concat
    (1, 32, 64, 2, 16, 24, 32, 3, 1, 1, "reflect", "elu", False, True, False),    # upconv(1,1): 96 input channels
    (1, 32, 0, 1, 20, 36, 16, 3, 1, 1, "reflect", "elu", False, True, False),     # Cout = 16 (half MFMA tile)
    (1, 16, 0, 2, 24, 40, 16, 3, 1, 1, "reflect", "elu", False, True, False),     # upconv(0,1): upsample without skip
    (1, 128, 0, 1, 30, 40, 128, 3, 1, 1, "zeros", None, False, False, False),     # medium tile config
    (2, 64, 0, 1, 30, 44, 128, 3, 2, 1, "zeros", "relu", True, False, False),     # stride 2, several tiles per parity class (backward-data classes)
    (1, 128, 0, 1, 17, 23, 256, 3, 2, 1, "zeros", "relu", True, False, False),    # stride 2, odd height and width (unequal classes)
    (2, 64, 0, 1, 16, 24, 128, 1, 2, 0, "zeros", None, False, False, False),      # 1x1 / 2 downsample: three of the four classes are zeros
    (1, 16, 0, 1, 18, 26, 32, 3, 2, 1, "zeros", None, False, False, False),       # stride 2 with Cin = 16 (thin 128x32 tiles, chunk depth 16)
    # the shapes the 480x640 network runs at (BASELINE configs[2]): large grids, every tile family the cost model picks there
    (2, 64, 0, 1, 120, 160, 64, 3, 1, 1, "zeros", "relu", True, False, True),     # layer1: 38 400 rows x 64 columns
    (2, 32, 64, 2, 240, 320, 32, 3, 1, 1, "reflect", "elu", False, True, False),  # upconv(1,1): 96 -> 32 with upsample + skip, 153 600 rows
    (2, 16, 0, 2, 480, 640, 16, 3, 1, 1, "reflect", "elu", False, True, False),   # upconv(0,1): 16 -> 16 at full resolution, 614 400 rows
    (2, 256, 0, 1, 30, 40, 256, 3, 1, 1, "zeros", "relu", True, False, True),     # layer3: split-K territory
    # 32-channel-tile 3x3 layers: backward-weight runs the tap-reuse patch kernel (k_wgrad3x3_taps) -- odd sizes (patches that overhang the
    # image), reflection padding with a bias column, upsample + concat with both sources multiples of 64 wide, several ci / co tiles
    (2, 128, 0, 1, 15, 20, 64, 3, 1, 1, "reflect", "elu", False, True, False),    # upconv(k,0)-like, 15 x 20 (layer4 resolution)
    (1, 64, 64, 2, 16, 24, 64, 3, 1, 1, "reflect", "elu", False, True, False),    # upconv(2,1): 64 upsampled + 64 skip channels
    (2, 192, 0, 1, 9, 11, 128, 3, 1, 1, "zeros", None, False, False, False),      # 6 x 4 tiles, patches wider than the image remainder
    (2, 64, 0, 1, 24, 40, 32, 3, 1, 1, "reflect", "elu", False, True, False),     # upconv(1,0): 64 -> 32, one co tile with the bias column
    (1, 32, 32, 2, 16, 16, 32, 3, 1, 1, "reflect", "elu", False, True, False),    # concat with 32-channel sources (one ci tile per source)
]


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_conv_forward_backward(case):
    from e2ehip import conv
    B, Cx, Cs, up, H, W, Cout, k, s, p, pad_mode, act, bn, use_bias, use_res = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    rnd = lambda *shape: torch.randn(*shape, generator=g)
    x = rnd(B, Cx, H // up if up > 1 else H, W // up if up > 1 else W).to(DEV).contiguous(memory_format=torch.channels_last)
    if up > 1:
        H, W = x.shape[2] * up, x.shape[3] * up
    skip = rnd(B, Cs, H, W).to(DEV).contiguous(memory_format=torch.channels_last) if Cs else None
    Cin = Cx + Cs
    w = (rnd(Cout, Cin, k, k) / (Cin * k * k) ** 0.5).to(DEV)
    bias = rnd(Cout).to(DEV) if use_bias else None
    scale = (rnd(Cout).abs() + 0.5).to(DEV) if bn else None
    shift = rnd(Cout).to(DEV) if bn else None
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rnd(B, Cout, Ho, Wo).to(DEV).contiguous(memory_format=torch.channels_last) if use_res else None
    in_norm = (0.45, 1 / 0.225) if Cx == 3 else None
    leaves = [t for t in (x, skip, w, bias, res) if t is not None]
    stem = Cx == 3
    for t in leaves:
        t.requires_grad_(not (stem and t is x))
    y = conv.conv2d(x, w, bias, s, p, pad_mode, act, (scale, shift) if bn else None, res, skip, up, in_norm)
    yr = _ref(x, skip, w, bias, scale, shift, res, up, s, p, pad_mode, act, in_norm)
    assert tuple(y.shape) == tuple(yr.shape)
    err = ((y.double() - yr).abs().max() / yr.abs().max()).item()
    assert err < 2e-5, f"forward: {err:.2e}"
    gy = _mask_kinks(rnd(*y.shape).to(DEV), yr, act)
    diff = [t for t in leaves if t.requires_grad]
    grads = torch.autograd.grad(y, diff, gy, retain_graph=False)
    grads_r = torch.autograd.grad(yr, diff, gy.double())
    for t, a, b in zip(diff, grads, grads_r):
        e = ((a.double() - b).abs().max() / (b.abs().max() + 1e-30)).item()
        assert e < 5e-5, f"grad of tensor {tuple(t.shape)}: {e:.2e}"


@pytest.mark.parametrize("case", [c for c in CASES if c[7] == 3 and c[8] == 1 and (c[1] + c[2]) % 32 == 0 and c[6] % 32 == 0 and c[1] % 32 == 0],
                         ids=lambda c: f"{c[1]}+{c[2]}to{c[6]}_{c[4]}x{c[5]}")
def test_tap_reuse_backward_weight_kernel(case, monkeypatch):
    """k_wgrad3x3_taps (opt-in, E2E_WGRAD_TAPS=1: profiles/r04_wgrad_taps.txt) on every eligible case of the table above -- same comparison
    against float64, so the experimental kernel stays correct while it is not the product's path."""
    monkeypatch.setenv("E2E_WGRAD_TAPS", "1")
    test_conv_forward_backward(case)


@pytest.mark.parametrize("H,W,act", [(20, 28, "disp"), (5, 7, None), (64, 96, "disp")])
def test_disparity_head(H, W, act):
    """Conv3x3(reflect) 16 -> 1 (+ 10*sigmoid+0.01): dedicated kernels, forward and all three gradients."""
    from e2ehip import conv
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(2, 16, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(1, 16, 3, 3, generator=g) * 0.1).to(DEV).requires_grad_(True)
    b = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    y = conv.conv2d(x, w, b, 1, 1, "reflect", act)
    yr = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), b.double())
    if act == "disp":
        yr = 10 * torch.sigmoid(yr) + 0.01
    assert ((y.double() - yr).abs().max() / yr.abs().max()).item() < 2e-5
    gy = torch.randn(y.shape, generator=g).to(DEV)
    for a, r in zip(torch.autograd.grad(y, [x, w, b], gy), torch.autograd.grad(yr, [x, w, b], gy.double())):
        assert ((a.double() - r).abs().max() / (r.abs().max() + 1e-30)).item() < 5e-5


@pytest.mark.parametrize("tile", [(64, 64), (128, 64), (128, 128), (128, 32), (32, 128), (32, 64), (64, 32), (32, 32)], ids=lambda t: f"{t[0]}x{t[1]}")
@pytest.mark.parametrize("ksplit", [1, 3])
def test_every_gemm_decomposition(tile, ksplit):
    """Each workgroup-tile family and split-K forced in turn (per call: conv2d(..., tuning=) -> e2e_conv2d_*_tuned) on one layer that all
    of them fit: forward, backward-data and -- through the same gather -- a stride-2 parity-class backward."""
    from e2ehip import conv
    g = torch.Generator().manual_seed(tile[0] * 7 + tile[1] + ksplit)
    rnd = lambda *shape: torch.randn(*shape, generator=g)
    for (Cin, Cout, H, W, s) in ((128, 128, 22, 36, 1), (48, 160, 21, 27, 2)):
        x = rnd(2, Cin, H, W).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (rnd(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).to(DEV).requires_grad_(True)
        scale, shift = (rnd(Cout).abs() + 0.5).to(DEV), rnd(Cout).to(DEV)
        y = conv.conv2d(x, w, None, s, 1, "zeros", "relu", (scale, shift), tuning=(tile[0], tile[1], ksplit))
        yr = _ref(x, None, w, None, scale, shift, None, 1, s, 1, "zeros", "relu", None)
        gy = _mask_kinks(rnd(*y.shape).to(DEV), yr, "relu")
        gx, gw = torch.autograd.grad(y, [x, w], gy)
        gxr, gwr = torch.autograd.grad(yr, [x, w], gy.double())
        for a, b, name in ((y, yr, "y"), (gx, gxr, "dx"), (gw, gwr, "dw")):
            e = ((a.double() - b).abs().max() / (b.abs().max() + 1e-30)).item()
            assert e < 5e-5, f"{name} {Cin}->{Cout} stride {s}: {e:.2e}"


@pytest.mark.parametrize("G", [1, 7, 64, 512, 768])
def test_streamk_decomposition(G):
    """Stream-K (k_conv_gemm_sk): G persistent workgroups share the (tile, K chunk) space in equal contiguous ranges; partial tiles are
    handed over through slabs + flags and summed in K order by the workgroup that holds the tile's head.  Every G from one workgroup
    (all tiles sequentially, no hand-off) to more workgroups than tiles (every tile split several ways): forward with folded BN +
    ReLU + residual, concat + upsample gather, reflection padding, and backward-data -- against fp64, and bitwise reproducible."""
    from e2ehip import conv
    g = torch.Generator().manual_seed(100 + G)
    rnd = lambda *shape: torch.randn(*shape, generator=g)
    cases = ((64, 0, 1, 64, 30, 44, "zeros", "relu"), (128, 0, 1, 160, 15, 21, "zeros", None), (32, 32, 2, 64, 24, 40, "reflect", "elu"))
    for (Cx, Cs, up, Cout, H, W, pm, act) in cases:
        Cin = Cx + Cs
        x = rnd(2, Cx, H // up, W // up).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        skip = rnd(2, Cs, H, W).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) if Cs else None
        w = (rnd(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).to(DEV).requires_grad_(True)
        bias = rnd(Cout).to(DEV) if pm == "reflect" else None
        bn = ((rnd(Cout).abs() + 0.5).to(DEV), rnd(Cout).to(DEV)) if pm == "zeros" else None
        res = rnd(2, Cout, H, W).to(DEV).contiguous(memory_format=torch.channels_last) if (pm == "zeros" and act) else None
        outs = []
        for rep in range(2):
            y = conv.conv2d(x, w, bias, 1, 1, pm, act, bn, res, skip, up, tuning=(64, 64, -G))
            ins = [x] + ([skip] if Cs else [])
            yr = _ref(x, skip, w, bias, bn[0] if bn else None, bn[1] if bn else None, res, up, 1, 1, pm, act, None)
            gy = _mask_kinks(torch.randn(*y.shape, generator=torch.Generator().manual_seed(5)).to(DEV), yr, act)
            gs = torch.autograd.grad(y, ins, gy)
            outs.append((y.detach().clone(), [t.clone() for t in gs]))
        assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
        gr = torch.autograd.grad(yr, ins, gy.double())
        for a, b, name in [(y, yr, "y")] + [(a, b, f"d_in{i}") for i, (a, b) in enumerate(zip(gs, gr))]:
            e = ((a.double() - b).abs().max() / (b.abs().max() + 1e-30)).item()
            assert e < 5e-5, f"G={G} {name} {Cin}->{Cout} {pm}: {e:.2e}"


def test_deferred_slab_reductions_in_one_launch_equal_the_per_layer_calls():
    """e2e_conv2d_bwd_weight_scaled_deferred + ONE e2e_wgrad_reduce_batched over several layers (what a NetPlan backward pass ends with) writes
    exactly the bits e2e_conv2d_bwd_weight_scaled writes layer by layer: the batched kernel keeps each layer's association of the slab sum
    (8 waves per group of quads from 8 slabs on, 2 below).  Covers the implicit-GEMM kernels with many and with few slabs, the 16-output-channel
    kernel, the thin 3x3 patch kernel, the RGB stem, a concat layer, bias columns and the folded-BatchNorm scale."""
    import ctypes
    from e2ehip import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    # B, Cin, C1, up, Hs, Ws, Cout, k, stride, pad, pad_mode(1 = reflect), bias, scale
    layers = [(2, 64, 64, 1, 60, 80, 64, 3, 1, 1, 0, False, True),        # many slabs (zl = 8)
              (2, 256, 256, 1, 8, 10, 512, 3, 2, 1, 0, False, True),      # few slabs (zl = 2)
              (2, 3, 3, 1, 64, 96, 64, 7, 2, 3, 0, False, True),          # RGB stem patch kernel
              (1, 96, 32, 2, 32, 48, 32, 3, 1, 1, 1, True, False),        # thin 3x3, upsample + skip concat, bias column
              (1, 32, 32, 1, 40, 56, 16, 3, 1, 1, 1, True, False),        # 16 output channels
              (2, 128, 128, 1, 15, 20, 256, 1, 1, 0, 0, True, False)]     # 1x1
    calls, descs, outs_ref, outs_def, keep = [], [], [], [], []
    for (B, Cin, C1, up, Hs, Ws, Cout, k, stride, pad, pm, has_bias, has_scale) in layers:
        Ho, Wo = (Hs + 2 * pad - k) // stride + 1, (Ws + 2 * pad - k) // stride + 1
        src0 = torch.randn(B, Hs // up, Ws // up, C1, generator=g).to(DEV)
        src1 = torch.randn(B, Hs, Ws, Cin - C1, generator=g).to(DEV) if C1 < Cin else None
        da = torch.randn(B, Ho, Wo, Cout, generator=g).to(DEV)
        scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV) if has_scale else None
        n_ws = lib.e2e_conv2d_wgrad_workspace_floats(B, Ho, Wo, Cin, Cout, k, k, 1 if has_bias else 0)
        res = []
        for deferred in (False, True):
            dw = torch.full((Cout, Cin, k, k), float("nan"), device=DEV)
            db = torch.full((Cout,), float("nan"), device=DEV) if has_bias else None
            ws = torch.empty(n_ws, device=DEV)
            args = [L.ptr(da), L.ptr(scale), L.ptr(src0), L.ptr(src1), C1, up, L.ptr(dw), L.ptr(db), L.ptr(ws), B, Hs, Ws, Cin, Cout, Ho, Wo, k, k, stride, pad, pm, 0,
                    0.45 if Cin == 3 else 0.0, 1 / 0.225 if Cin == 3 else 1.0]
            if deferred:
                d = L.WgradReduceDesc()
                L.call("e2e_conv2d_bwd_weight_scaled_deferred", *args, ctypes.byref(d), L.stream())
                descs.append(d)
            else:
                L.call("e2e_conv2d_bwd_weight_scaled", *args, L.stream())
            res.append((dw, db))
            keep += [ws, dw, db]
        outs_ref.append(res[0])
        outs_def.append(res[1])
        keep += [src0, src1, da, scale]
    assert {d.zl for d in descs} == {2, 8}
    arr = (L.WgradReduceDesc * len(descs))(*descs)
    total = lib.e2e_wgrad_reduce_batch_prepare(arr, len(descs))
    assert total > 0 and arr[0].first_item == 0 and all(arr[i].first_item < arr[i + 1].first_item for i in range(len(descs) - 1))
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    assert torch.isnan(outs_def[0][0]).all()                       # nothing was reduced yet
    L.call("e2e_wgrad_reduce_batched", L.ptr(table), len(descs), total, L.stream())
    torch.cuda.synchronize()
    for (dw_r, db_r), (dw_d, db_d) in zip(outs_ref, outs_def):
        assert torch.isfinite(dw_r).all() and torch.equal(dw_r, dw_d)
        if db_r is not None:
            assert torch.equal(db_r, db_d)
    bad = (L.WgradReduceDesc * 1)(L.WgradReduceDesc())
    assert lib.e2e_wgrad_reduce_batch_prepare(bad, 1) == -1
    with pytest.raises(L.E2EError):
        L.call("e2e_wgrad_reduce_batched", None, 1, 1, L.stream())
