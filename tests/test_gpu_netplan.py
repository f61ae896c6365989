"""The static launch plan of the depth network (e2ehip.netplan) against the nn.Module path (per-layer autograd Functions over
the same kernels, itself pinned by golden g7): same disparity bit for bit, same parameter gradients up to the order in which
multi-consumer gradients are accumulated; and the whole forward + backward replayed from a captured hipGraph."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(seed=0):
    from depth_estimation.networks import DispResNet_Indoor
    torch.manual_seed(seed)
    m = DispResNet_Indoor(18, False)
    with torch.no_grad():                               # non-trivial BatchNorm statistics / affine
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.8, 1.2)
                mod.bias.normal_(0, 0.1)
    m.to(DEV).eval()
    for name, p in m.named_parameters():                # set_refinement_mode (online_adaption.py:175-184)
        if name.find("bn") != -1:
            p.requires_grad = False
    return m


@pytest.mark.parametrize("B,H,W,overlap", [(2, 64, 96, False), (2, 64, 96, True), (1, 96, 128, True)])
def test_plan_matches_module_path(B, H, W, overlap):
    from e2ehip.netplan import NetPlan
    m = _model()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, H, W, 3, generator=g).to(DEV)
    gd = torch.randn(B, 1, H, W, generator=g).to(DEV)
    disp = m(x, 0)[("disp", 0, 0)]
    disp.backward(gd)
    used = m.used_parameters()
    ref = {id(p): p.grad.clone() for p in used if p.requires_grad}
    feats_ref = [f.detach().clone() for f in m.encoder.features]
    for p in m.parameters():
        p.grad = None
    plan = NetPlan(m, B, H, W, DEV, overlap=overlap)
    plan.refresh_layouts()
    d2 = plan.forward(x)
    assert torch.equal(d2, disp.detach())
    for a, b in zip(plan.features, feats_ref):
        assert torch.equal(a.nchw(), b)
    plan.backward(gd)
    torch.cuda.synchronize()
    assert {id(p) for p in plan.parameters()} == set(ref)
    for p in plan.parameters():
        a, b = plan.sink(p), ref[id(p)]
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        assert err < 2e-5, (tuple(p.shape), err)
    # a second pass over the same buffers gives the same result (no state leaks between passes)
    keep = [plan.sink(p).clone() for p in plan.parameters()]
    plan.forward(x)
    plan.backward(gd)
    torch.cuda.synchronize()
    for p, k in zip(plan.parameters(), keep):
        assert torch.equal(plan.sink(p), k)


def test_plan_forward_backward_as_hip_graph():
    from e2ehip.netplan import NetPlan
    m = _model(3)
    B, H, W = 2, 64, 96
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, H, W, 3, generator=g).to(DEV)
    gd = torch.randn(B, 1, H, W, generator=g).to(DEV)
    plan = NetPlan(m, B, H, W, DEV, overlap=True)
    plan.refresh_layouts()
    plan.forward(x)
    plan.backward(gd)
    torch.cuda.synchronize()
    ref_disp = plan.disp.t.clone()
    ref = [plan.sink(p).clone() for p in plan.parameters()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        plan.forward()
        plan.backward()                                 # warm-up on the capture stream
        s.synchronize()
        with torch.cuda.graph(graph, stream=s):
            plan.forward()                              # input buffer and disp gradient buffer are read in place
            plan.backward()
    for p in plan.parameters():
        plan.sink(p).zero_()
    plan.disp.t.zero_()
    plan.x.t.copy_(x)
    plan.disp.g.copy_(gd.reshape(plan.disp.g.shape))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.disp.t, ref_disp)
    for p, r in zip(plan.parameters(), ref):
        assert torch.equal(plan.sink(p), r)


def test_batched_copies_equal_plain_copies():
    """e2e_copy_batched (NetPlan.move_slot: every layer's activations of one image to another batch slot in ONE launch) against tensor copies:
    sizes from one 16-byte quad to several work items plus a ragged tail; malformed descriptors are refused on the host."""
    import ctypes
    from e2ehip import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    sizes = [4, 8, 4096, 4100, 16384 // 4, 16384 // 4 + 4, 3 * 16384 // 4 - 4, 1_000_000, 2_621_440]
    srcs = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    dsts = [torch.full((n + 8,), float("nan"), device=DEV) for n in sizes]           # 8 guard floats behind every destination
    arr = (L.CopyDesc * len(sizes))(*[L.CopyDesc(s.data_ptr(), d.data_ptr(), 4 * n, 0) for s, d, n in zip(srcs, dsts, sizes)])
    total = lib.e2e_copy_batch_prepare(arr, len(sizes))
    assert total == sum((4 * n + 16383) // 16384 for n in sizes)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    L.call("e2e_copy_batched", L.ptr(table), len(sizes), total, L.stream())
    torch.cuda.synchronize()
    for s, d, n in zip(srcs, dsts, sizes):
        assert torch.equal(d[:n], s) and torch.isnan(d[n:]).all()
    for bad in (L.CopyDesc(srcs[0].data_ptr(), dsts[0].data_ptr(), 12, 0), L.CopyDesc(srcs[0].data_ptr() + 4, dsts[0].data_ptr(), 16, 0),
                L.CopyDesc(None, dsts[0].data_ptr(), 16, 0), L.CopyDesc(srcs[0].data_ptr(), dsts[0].data_ptr(), 0, 0)):
        assert lib.e2e_copy_batch_prepare((L.CopyDesc * 1)(bad), 1) == -1
    with pytest.raises(L.E2EError):
        L.call("e2e_copy_batched", None, 1, 1, L.stream())
