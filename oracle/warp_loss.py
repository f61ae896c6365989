"""CPU restatement (torch, fp32) of the view-synthesis warp and the image-space losses.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity: PINNED by tests/golden/g1..g5
(captured from the reference's own view_synthesis.py / losses.py by make_golden.py).

Every function names the reference lines it follows.  The functions are written as plain
tensor expressions so autograd provides the backward the reference gets from autograd.
"""
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# view synthesis  (reference: depth_estimation/view_synthesis.py)
# ----------------------------------------------------------------------------------------
def pixel_grid(batch, height, width, dtype=torch.float32):
    """Homogeneous pixel coordinates (B,3,H*W): rows x(=w), y(=h), 1.
    reference: view_synthesis.py:17-32 (BackprojectDepth.__init__)."""
    ys, xs = torch.meshgrid(torch.arange(height, dtype=dtype),
                            torch.arange(width, dtype=dtype), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width, dtype=dtype)], 0)
    return pix.unsqueeze(0).repeat(batch, 1, 1)


def backproject(depth, inv_K):
    """depth (B,1,H,W), inv_K (B,4,4) -> camera points (B,4,H*W).
    reference: view_synthesis.py:34-40 (BackprojectDepth.forward)."""
    B, _, H, W = depth.shape
    pix = pixel_grid(B, H, W, depth.dtype)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(B, 1, -1) * cam
    return torch.cat([cam, torch.ones(B, 1, H * W, dtype=depth.dtype)], 1)


def project(points, K, T, height, width, eps=1e-7, geometric=False):
    """points (B,4,N) -> normalised sampling grid (B,H,W,2) + in-bounds mask (B,1,H,W).
    reference: view_synthesis.py:54-78 (Project3D.forward).  The grid is normalised with
    /(W-1), /(H-1) (view_synthesis.py:66-67) although it is later sampled with
    align_corners=False (online_adaption.py:453): that quirk is kept."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(B, 2, height, width).permute(0, 2, 3, 1)
    gx = pix[..., 0] / (width - 1)
    gy = pix[..., 1] / (height - 1)
    grid = (torch.stack([gx, gy], -1) - 0.5) * 2
    valid = (grid.abs().max(dim=-1)[0] <= 1).unsqueeze(1).float()
    if geometric:
        z = cam[:, 2].clamp(min=1e-3).reshape(B, 1, height, width)
        return grid, z, valid
    return grid, valid


def inverse_warp(depth, src_nchw, K, inv_K, T, padding_mode="border", align_corners=False):
    """The synthesis step of the refinement: target depth + source frame -> synthesized
    target view and validity mask.
    reference: online_adaption.py:412-455 (novel_view_synthesis, non-geometric branch)."""
    B, _, H, W = depth.shape
    pts = backproject(depth, inv_K)
    grid, valid = project(pts, K, T, H, W)
    synth = F.grid_sample(src_nchw, grid, padding_mode=padding_mode, align_corners=align_corners)
    return synth, valid, grid


# ----------------------------------------------------------------------------------------
# losses  (reference: loss/losses.py)
# ----------------------------------------------------------------------------------------
def ssim(x, y):
    """monodepth2-style SSIM distance map, 3x3 mean filter over a 1-px reflection pad.
    reference: losses.py:10-37 (C1=0.01**2, C2=0.03**2, clamp((1-n/d)/2,0,1))."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(x, 3, 1)
    mu_y = F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def photometric(prediction, target):
    """0.85 * mean_c SSIM + 0.15 * mean_c |target - prediction| -> (B,1,H,W).
    reference: losses.py:97-117."""
    s = ssim(prediction, target).mean(1, True)
    l1 = torch.abs(target - prediction).mean(1, True)
    return 0.85 * s + 0.15 * l1


def masked_photometric_mean(synth, target, valid, use_mask=True):
    """The scalar the refinement optimises from one source frame.
    reference: online_adaption.py:544-564 (both images multiplied by the mask),
    online_adaption.py:482-511 (.mean(1, keepdim) then .mean())."""
    if use_mask:
        p = photometric(synth * valid, target * valid)
    else:
        p = photometric(synth, target)
    return p.mean(1, keepdim=True).mean(), p


def depth_regularizer(initial_depth, refined_depth, kind="l2"):
    """reference: losses.py:134-148 (nn.L1Loss / nn.MSELoss, mean reduction)."""
    if kind == "l1":
        return torch.mean(torch.abs(initial_depth - refined_depth))
    if kind == "l2":
        return torch.mean((initial_depth - refined_depth) ** 2)
    raise ValueError("please specify a correct norm")


def smoothness(disp, img):
    """Edge-aware first-order smoothness. reference: losses.py:119-132."""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def normalised_smoothness(disp, img):
    """reference: online_adaption.py:600-610 (mean-normalised disparity)."""
    m = disp.mean(2, True).mean(3, True)
    return smoothness(disp / (m + 1e-7), img)


def geometric_consistency(warped_depth, interpolated_depth, valid):
    """reference: losses.py:84-95 (host-side `if mask.sum() > 10000`)."""
    diff = ((warped_depth - interpolated_depth).abs() / (warped_depth + interpolated_depth)).clamp(0, 1)
    mask = valid.expand_as(diff)
    if mask.sum() > 10000:
        return (diff * mask).sum() / mask.sum()
    return torch.tensor(0.0)


def depth_gt(prediction, sparse_gt, sparse_mask):
    """reference: losses.py:151-160."""
    return torch.mean(torch.abs(prediction.squeeze() * sparse_mask.squeeze() - sparse_gt.squeeze()))


def min_reprojection(error_maps):
    """reference: train_depth.py:657-661 (`optimize, indexs = torch.min(photmetric, dim=1); optimize.mean()`)."""
    return torch.min(error_maps, dim=1)[0].mean()


def process_disparity(disp_pair):
    """reference: train_depth.py:224-237.  `l_mesh, _ = torch.meshgrid(linspace(0,1,h), linspace(0,1,w))` varies along
    the ROWS (ij indexing), so the flipped mask equals the mask; the expression is kept term by term."""
    left = disp_pair[:1]
    right = torch.flip(disp_pair[1:], [3])
    middle = 0.5 * (left + right)
    h, w = left.shape[2], left.shape[3]
    l_mesh, _ = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    l_mask = (1.0 - torch.clip(20 * (l_mesh - 0.05), 0, 1)).unsqueeze(0).unsqueeze(0)
    r_mask = torch.flip(l_mask, [3])
    return r_mask * left + l_mask * right + (1.0 - l_mask - r_mask) * middle


def depth_errors(gt, pred):
    """reference: losses.py:183-201 -> abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3."""
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def depth_metrics(dataset, gt, pred):
    """reference: losses.py:162-181 (TUM masks gt==0; ICL keeps everything)."""
    pred = pred.squeeze().detach()
    gt = gt.squeeze().detach()
    if dataset == "TUM":
        keep = gt != 0.0
    elif dataset == "ICL":
        keep = torch.ones_like(gt, dtype=torch.bool)
    else:
        raise ValueError("Dataset Not Found")
    return depth_errors(gt[keep], pred[keep])


# ----------------------------------------------------------------------------------------
# median scaling (reference: online_adaption.py:282,287-298)
# ----------------------------------------------------------------------------------------
def median_scale(depths, gt_depths):
    """depths: list of (B,1,H,W) predicted depths (1/disp); gt_depths (B,L,H,W,1).
    ratio = median(gt)/median(pred) with torch.median = LOWER median over all elements; the
    ratio stays in the autograd graph (online_adaption.py:295-298)."""
    stacked = torch.cat([d.unsqueeze(1) for d in depths], 1).permute(0, 1, 3, 4, 2)
    ratio = torch.median(gt_depths) / torch.median(stacked)
    return [d * ratio for d in depths], ratio
