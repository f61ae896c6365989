#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own Python files.

Run in the build container only (needs /root/reference, or $REFERENCE_DIR):
    python tests/golden/make_golden.py

The reference files are loaded BY PATH (importing the package `depth_estimation` would pull in
torchvision through its __init__), with two stubs for un-installed third-party modules:
  * chamferdist.chamfer.knn_points = None   (only knn_points_loss needs it; not exercised)
  * torchvision.models -> oracle.depthnet's restatement of the torchvision ResNet (so the
    decoder / wiring / stem of networks.py are genuine reference code, the ResNet-18 body is ours).
Nothing from the reference is copied: the fixtures hold inputs and expected outputs only.
"""
import importlib.util
import math
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("REFERENCE_DIR", "/root/reference")

from oracle import depthnet as odn  # noqa: E402  (torchvision stand-in + seeded state dict)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    cd, cdc = types.ModuleType("chamferdist"), types.ModuleType("chamferdist.chamfer")
    cdc.knn_points = None
    tv, tvm, tvr = types.ModuleType("torchvision"), types.ModuleType("torchvision.models"), types.ModuleType("torchvision.models.resnet")
    tvr.BasicBlock, tvr.Bottleneck, tvr.model_urls = odn.BasicBlock, odn.Bottleneck, {}
    tvm.ResNet, tvm.resnet = odn.ResNet, tvr
    for n in (18, 34, 50, 101, 152):
        setattr(tvm, f"resnet{n}", odn.resnet18)
    tv.models = tvm
    sys.modules.update({"chamferdist": cd, "chamferdist.chamfer": cdc, "torchvision": tv,
                        "torchvision.models": tvm, "torchvision.models.resnet": tvr})
    return (_load("ref_vs", "depth_estimation/view_synthesis.py"), _load("ref_ls", "loss/losses.py"),
            _load("ref_tu", "utils/training_utils.py"), _load("ref_net", "depth_estimation/networks.py"))


def icl_K(H, W):
    """ICL intrinsics scaled to HxW (fx=481.2, fy=-480, cx=319.5, cy=239.5 at 480x640)."""
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 481.2 * W / 640, -480.0 * H / 480, 319.5 * W / 640, 239.5 * H / 480
    return K.unsqueeze(0)


def small_motion(rz_deg=1.0, ry_deg=0.5, t=(0.05, 0.01, 0.02)):
    a, b = math.radians(rz_deg), math.radians(ry_deg)
    Rz = torch.tensor([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.0]])
    Ry = torch.tensor([[math.cos(b), 0, math.sin(b)], [0, 1.0, 0], [-math.sin(b), 0, math.cos(b)]])
    T = torch.eye(4)
    T[:3, :3] = Rz @ Ry
    T[:3, 3] = torch.tensor(t)
    return T.unsqueeze(0)


def smooth_depth(H, W, g):
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d = 2.0 + 0.5 * torch.sin(2 * math.pi * xs / W) * torch.cos(2 * math.pi * ys / H) + 0.05 * torch.rand(H, W, generator=g)
    return d.view(1, 1, H, W)


def smooth_image(H, W, g):
    """Low-frequency texture + noise so bilinear gradients are informative."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    ch = [0.5 + 0.3 * torch.sin(xs * (0.11 + 0.05 * c) + c) * torch.cos(ys * (0.07 + 0.03 * c)) for c in range(3)]
    return (torch.stack(ch, -1) + 0.2 * torch.rand(H, W, 3, generator=g)).clamp(0, 1).unsqueeze(0)


def save(name, **arrs):
    out = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items()})


def warp_loss_ref(vs, ls, depth, src_nhwc, tgt_nhwc, K, T, padding, use_mask=True):
    """The reference's own modules composed as online_adaption.py:388-455,473-511,544-564 does."""
    B, _, H, W = depth.shape
    bp, pr, ssim = vs.BackprojectDepth(B, H, W), vs.Project3D(B, H, W), ls.SSIM()
    src, tgt = src_nhwc.permute(0, 3, 1, 2), tgt_nhwc.permute(0, 3, 1, 2)
    invK = torch.pinverse(K)
    cam = bp(depth, invK)
    grid, valid = pr(points=cam, K=K, T=T, geometric=False)
    synth = F.grid_sample(src, grid, padding_mode=padding, align_corners=False)
    if use_mask:
        pm = ls.photometric_loss(ssim=ssim, prediction=synth * valid, target=tgt * valid)
    else:
        pm = ls.photometric_loss(ssim=ssim, prediction=synth, target=tgt)
    loss = pm.mean(1, keepdim=True).mean()
    return dict(cam=cam, grid=grid, valid=valid, synth=synth, pmap=pm, loss=loss)


def main():
    vs, ls, tu, net = load_reference()
    torch.set_num_threads(8)

    # ---- G1: backproject / project, G2: grid_sample modes, G4: end-to-end dL/ddepth -------------
    for tag, (H, W), motion in (("a", (24, 32), small_motion(2.0, 1.0, (0.3, 0.05, 0.1))),
                                ("b", (48, 64), small_motion())):
        g = torch.Generator().manual_seed(1234)
        depth = smooth_depth(H, W, g)
        src, tgt = smooth_image(H, W, g), smooth_image(H, W, g)
        K, T = icl_K(H, W), motion
        rec = dict(depth=depth, src=src, tgt=tgt, K=K, T=T, invK=torch.pinverse(K))
        for pad in ("border", "zeros"):
            d = depth.clone().requires_grad_(True)
            o = warp_loss_ref(vs, ls, d, src, tgt, K, T, pad)
            gd, = torch.autograd.grad(o["loss"], d, retain_graph=True)
            gs, = torch.autograd.grad(o["loss"], o["synth"])
            rec.update({f"{pad}_synth": o["synth"], f"{pad}_loss": o["loss"], f"{pad}_gdepth": gd,
                        f"{pad}_gsynth": gs, f"{pad}_pmap": o["pmap"]})
            if pad == "border":
                rec.update(cam=o["cam"], grid=o["grid"], valid=o["valid"])
                o2 = warp_loss_ref(vs, ls, d, src, tgt, K, T, pad, use_mask=False)
                gd2, = torch.autograd.grad(o2["loss"], d)
                rec.update(nomask_loss=o2["loss"], nomask_gdepth=gd2)
        # geometric branch of Project3D (view_synthesis.py:73-76) + align_corners=True sampling
        pr = vs.Project3D(1, H, W)
        gridg, zg, validg = pr(points=vs.BackprojectDepth(1, H, W)(depth, rec["invK"]), K=K, T=T, geometric=True)
        rec.update(geo_depth=zg, geo_synth_ac=F.grid_sample(src.permute(0, 3, 1, 2), gridg, padding_mode="border", align_corners=True))
        save(f"g1_warp_{tag}", **rec)

    # ---- G3: SSIM / photometric on plain images (fwd + grad wrt prediction) --------------------
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, 20, 28, generator=g).requires_grad_(True)
    y = torch.rand(2, 3, 20, 28, generator=g)
    ssim = ls.SSIM()
    s = ssim(x, y)
    p = ls.photometric_loss(ssim=ssim, prediction=x, target=y)
    gx, = torch.autograd.grad(p.mean(), x)
    save("g3_ssim", x=x, y=y, ssim=s, pmap=p, gx=gx)

    # ---- G5: auxiliary losses + metrics ---------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    disp = (torch.rand(1, 1, 16, 24, generator=g) + 0.1).requires_grad_(True)
    img = torch.rand(1, 3, 16, 24, generator=g)
    sm = ls.disparity_smoothness_loss(disp, img)
    gsm, = torch.autograd.grad(sm, disp)
    d0 = torch.rand(1, 1, 16, 24, generator=g) + 1.0
    d1 = (torch.rand(1, 1, 16, 24, generator=g) + 1.0).requires_grad_(True)
    r1 = ls.depth_reguralizer(d0, d1, "l1"); r2 = ls.depth_reguralizer(d0, d1, "l2")
    gr2, = torch.autograd.grad(r2, d1)
    gt = torch.rand(16, 24, generator=g) * 3 + 0.5
    gt_holes = gt.clone(); gt_holes[torch.rand(16, 24, generator=g) < 0.1] = 0.0
    pred = gt * (1 + 0.2 * (torch.rand(16, 24, generator=g) - 0.5))
    m_icl = torch.stack(ls.depth_metrics("ICL", gt, pred))
    m_tum = torch.stack(ls.depth_metrics("TUM", gt_holes, pred))
    wd = torch.rand(1, 1, 120, 100, generator=g) + 0.5
    idp = torch.rand(1, 1, 120, 100, generator=g) + 0.5
    vm = (torch.rand(1, 1, 120, 100, generator=g) > 0.05).float()
    geo = ls.geometric_consistency_loss({("warped_depth", -1): wd, ("interpolated_depth", -1): idp, ("valid_mask", -1): vm}, -1, "cpu")
    smask = (torch.rand(1, 1, 16, 24, generator=g) < 0.3).float()
    gtl = ls.depth_gt_loss(d1.detach(), d0 * smask, smask)
    save("g5_aux", disp=disp, img=img, smooth=sm, gsmooth=gsm, d0=d0, d1=d1, reg_l1=r1, reg_l2=r2, greg_l2=gr2,
         gt=gt, gt_holes=gt_holes, pred=pred, metrics_icl=m_icl, metrics_tum=m_tum,
         wd=wd, idp=idp, vm=vm, geo=geo, smask=smask, gt_loss=gtl)

    # ---- G6: poses -> transforms, inverse, sparse sampling (seeded), disp<->depth -------------
    g = torch.Generator().manual_seed(3)
    poses = torch.eye(4).repeat(1, 4, 1, 1)
    for s in range(4):
        poses[0, s] = small_motion(3.0 * s, 2.0 * s, (0.1 * s, -0.05 * s, 0.02 * s))[0]
    tr = tu.torch_poses_to_transforms(poses)
    inv = tu.inverse_T_matrix(poses[0])
    torch.manual_seed(99)
    dep = torch.rand(1, 12, 16, 1); dep[0, :2, :3] = 0.0
    torch.manual_seed(5)
    md, mk = tu.sparse_sampling("random", 0.3, dep)
    dd = tu.convert_disp_to_depth(torch.linspace(0, 1, 9), 0.1, 80.0)
    save("g6_pose", poses=poses, transforms=tr, inverse=inv, sp_depth=dep, sp_masked=md, sp_mask=mk, d2d=dd)

    # ---- G7: depth network (reference decoder/wiring, stub ResNet body) -------------------------
    H, W = 64, 96
    sd = odn.random_state_dict(0)
    model = net.DispResNet_Indoor(num_layers=18, pretrained=False)
    keys = list(model.state_dict().keys())
    assert keys == list(sd.keys()), "state-dict key order mismatch"
    model.load_state_dict(sd)
    model.eval()
    for name, p in model.named_parameters():            # online_adaption.py:175-184
        if name.find("bn") != -1:
            p.requires_grad = False
    g = torch.Generator().manual_seed(21)
    img = smooth_image(H, W, g)
    out = model(img, 0)
    disp = out[("disp", 0, 0)]
    feats = model.encoder.features
    wgt = torch.rand(disp.shape, generator=g)
    (disp * wgt).sum().backward()
    gnorm = OrderedDict((n, (p.grad.norm() if p.grad is not None else torch.tensor(-1.0))) for n, p in model.named_parameters())
    sample = {n: model.get_parameter(n).grad.flatten()[:16].clone() for n in
              ("encoder.encoder.conv1.weight", "encoder.encoder.layer2.0.downsample.0.weight",
               "encoder.encoder.layer4.1.conv2.weight", "decoder.decoder.0.conv.conv.weight",
               "decoder.decoder.7.conv.conv.weight", "decoder.decoder.9.conv.conv.bias", "decoder.decoder.10.conv.weight")}
    sd_sum = sum(float(v.double().abs().sum()) for v in sd.values() if v.dtype.is_floating_point)
    save("g7_net", img=img, disp=disp, wgt=wgt, sd_abs_sum=np.float64(sd_sum),
         keys=np.array(keys), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
         grad_names=np.array(list(gnorm.keys())), grad_norms=torch.stack(list(gnorm.values())),
         **{"f%d" % i: f for i, f in enumerate(feats)},
         **{"gs_" + k.replace(".", "_"): v for k, v in sample.items()})

    # ---- G8: three refinement steps without a map (first keyframe) ---------------------------------
    # composition per online_adaption.py:259-327 (refinement), :369-455, :473-542 with
    # LOSS = photometric(mask) + depth_regularizer(l2, 1e-2); first_iter => no 3-D loss.
    args = types.SimpleNamespace(OPTIMIZATION=types.SimpleNamespace(optimizer="Adam", learning_rate=1e-5))
    model.zero_grad()
    model.load_state_dict(sd)
    opt = tu.define_optim(args, list(model.parameters()))
    g = torch.Generator().manual_seed(1234)
    colors = torch.cat([smooth_image(H, W, g), smooth_image(H, W, g)], 0).unsqueeze(0)       # (1,2,H,W,3)
    gt_depths = torch.cat([smooth_depth(H, W, g), smooth_depth(H, W, g)], 1).unsqueeze(-1)   # (1,2,H,W,1)
    poses = torch.stack([torch.eye(4), small_motion()[0]], 0).unsqueeze(0)
    K = icl_K(H, W).unsqueeze(0)                                                             # (1,1,4,4)
    transform = tu.torch_poses_to_transforms(poses)
    bp, pr, ssim_m = vs.BackprojectDepth(1, H, W), vs.Project3D(1, H, W), ls.SSIM()
    losses, ratios, initial = [], [], {}
    for step in range(3):
        depths = []
        for idx in range(2):
            dsp = model(colors[:, idx], idx)[("disp", idx, 0)]
            dpt = 1 / dsp
            if step == 0:
                initial[idx] = dpt.clone().detach()
            depths.append(dpt)
        dt = torch.cat([d.unsqueeze(1) for d in depths], 1).permute(0, 1, 3, 4, 2)
        ratio = torch.median(gt_depths) / torch.median(dt)
        depths[0] *= ratio
        depths[1] *= ratio
        src, tgt = colors[:, 0].permute(0, 3, 1, 2), colors[:, 1].permute(0, 3, 1, 2)
        Kc = K[:, 0]
        cam = bp(depths[1], torch.pinverse(Kc))
        grid, valid = pr(points=cam, K=Kc, T=transform[:, 1], geometric=False)
        synth = F.grid_sample(src, grid, padding_mode="border", align_corners=False)
        opt.zero_grad()
        pm = ls.photometric_loss(ssim=ssim_m, prediction=synth * valid, target=tgt * valid)
        loss = pm.mean(1, keepdim=True).mean()
        reg = sum(ls.depth_reguralizer(initial[i], depths[i], "l2") for i in range(2))
        loss = loss + reg * 1e-2
        loss.backward()
        opt.step()
        losses.append(loss.item()); ratios.append(ratio.item())
        if step == 0:
            first_depth1 = depths[1].detach().clone()
    with torch.no_grad():
        final_disp = [model(colors[:, i], i)[("disp", i, 0)] for i in range(2)]
    save("g8_refine", colors=colors, gt_depths=gt_depths, poses=poses, K=K, losses=np.array(losses), ratios=np.array(ratios),
         first_depth1=first_depth1, final_disp0=final_disp[0], final_disp1=final_disp[1])

    # ---- G9: 480x640 checksums of the warp+loss path (size-dependent indexing) -------------------
    H, W = 480, 640
    g = torch.Generator().manual_seed(1234)
    depth = smooth_depth(H, W, g)
    src, tgt = smooth_image(H, W, g), smooth_image(H, W, g)
    K, T = icl_K(H, W), small_motion()
    d = depth.clone().requires_grad_(True)
    o = warp_loss_ref(vs, ls, d, src, tgt, K, T, "border")
    gd, = torch.autograd.grad(o["loss"], d)
    pick = torch.randint(0, H * W, (64,), generator=g)
    save("g9_full_checksums", seed=1234, loss=o["loss"], valid_sum=o["valid"].sum(), synth_sum=o["synth"].double().sum(),
         synth_abs=o["synth"].double().abs().sum(), gdepth_sum=gd.double().sum(), gdepth_abs=gd.double().abs().sum(),
         pick=pick, synth_pick=o["synth"][0, :, :, :].reshape(3, -1)[:, pick], gdepth_pick=gd.view(-1)[pick],
         pmap_pick=o["pmap"].view(-1)[pick])


if __name__ == "__main__":
    main()
