"""CPU restatement of the online-refinement loop (one keyframe pair) built from the oracle parts.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  It is also what bench.py times as the
`cpu_baseline` ("port") on the GPU node's host cores.

Parity: the no-map path (first keyframe) is PINNED by tests/golden/g8 (loss trajectory + depths
captured from the reference's own modules); the map / 3-D loss part inherits "unpinned" from
pointfusion.py / knn.py.

Follows online_adaption.py:259-327 (refinement), :329-366 (create_refined_pointcloud),
:369-386 (depth_refinement), :388-455, :457-471, :473-542, :612-623, :638-645.
"""
from collections import OrderedDict

import torch

from . import depthnet, knn, pointfusion, poses as oposes, warp_loss


class Config:
    """The LOSS / MODEL / OPTIMIZATION keys the path reads (configs/config.yaml), with the
    README-recommended online-adaption values (README.md:146-158)."""
    padding_mode = "border"
    photometric_mask = True
    depth_regularizer = True
    depth_regularizer_weight = 1e-2
    depth_regularizer_type = "l2"
    three3d_loss = True
    three3d_loss_weight = 1.0
    refinement_steps = 3
    learning_rate = 1e-5
    dist_th, angle_th, sigma = 0.05, 20.0, 0.6
    dataset = "ICL"
    # off in the recommended configuration (configs/config.yaml); online_adaption.py:486-532
    min_reprojection = False
    auto_masking = False
    geometric = False
    geometric_weight = 0.5
    smoothness = False
    smoothness_weight = 1e-3
    tie_break_noise = None          # auto-masking + min-reprojection adds randn * 1e-5 (:498); a fixed tensor for tests


class Refiner:
    def __init__(self, state_dict, cfg=None):
        self.cfg = cfg or Config()
        self.sd = OrderedDict((k, v.clone()) for k, v in state_dict.items())
        self.train_keys = depthnet.trainable_keys(self.sd)
        for k in self.train_keys:
            self.sd[k].requires_grad_(True)
        # the reference builds Adam over ALL params (online_adaption.py:154); params whose grad is
        # None are skipped by torch.optim.Adam, so only train_keys ever move.
        self.opt = torch.optim.Adam([self.sd[k] for k in self.train_keys], lr=self.cfg.learning_rate)
        self.first_iter = True
        self.map = pointfusion.empty_state()

    def predict_depths(self, colors):
        """colors (1,2,H,W,3) -> [depth0, depth1] each (1,1,H,W) = 1/disp (online_adaption.py:281-282)."""
        self.disps = [depthnet.disp_forward(self.sd, colors[:, i]) for i in range(2)]
        return [1 / d for d in self.disps]

    def refine_pair(self, colors, gt_depths, poses, K, update_map=True):
        """colors (1,2,H,W,3) in [0,1]; gt_depths (1,2,H,W,1); poses (1,2,4,4); K (1,1,4,4).
        Runs cfg.refinement_steps optimisation steps then the map step.  Returns per-step records."""
        cfg = self.cfg
        transform = oposes.poses_to_transforms(poses)
        Kc = K[:, 0]
        invK = torch.pinverse(Kc)
        src = colors[:, 0].permute(0, 3, 1, 2)
        tgt = colors[:, 1].permute(0, 3, 1, 2)
        T = transform[:, 1]
        initial = None
        records = []
        for step in range(cfg.refinement_steps):
            depths = self.predict_depths(colors)
            if step == 0:
                initial = [d.clone().detach() for d in depths]          # pre-scaling (:284-285)
            # (test diagnostics) which element IS the lower median of the stacked predictions: the ratio's backward puts a sum over all
            # pixels on that one element, so two evaluations that pick different (near-tied) elements have different -- equally valid --
            # parameter gradients; tests compare those only when the element agrees
            stacked = torch.cat([d.detach().reshape(-1) for d in depths])
            median_index = int(stacked.median(0).indices)
            # torch.median(x) (no dim, as online_adaption.py:295 calls it) differentiates as evenly_distribute_backward: ALL elements equal
            # to the median value share its gradient
            median_indices = (stacked == stacked.median()).nonzero().reshape(-1).tolist()
            depths, ratio = warp_loss.median_scale(depths, gt_depths)  # (:292-298)
            for d in depths:
                d.retain_grad()                                        # (test diagnostics) d loss / d scaled depth, rec["g_depth"]
            self.opt.zero_grad()
            if cfg.min_reprojection or cfg.auto_masking or cfg.geometric or cfg.smoothness:
                loss, photo = self.flagged_image_losses(depths, src, tgt, Kc, invK, T)
            else:
                synth, valid, _ = warp_loss.inverse_warp(depths[1], src, Kc, invK, T, cfg.padding_mode)
                loss, _ = warp_loss.masked_photometric_mean(synth, tgt, valid, cfg.photometric_mask)
                photo = loss
            rec = {"photometric": photo.item(), "ratio": ratio.item()}
            if cfg.depth_regularizer:
                reg = sum(warp_loss.depth_regularizer(initial[i], depths[i], cfg.depth_regularizer_type) for i in range(2))
                loss = loss + reg * cfg.depth_regularizer_weight
                rec["reg"] = reg.item()
            if cfg.three3d_loss and not self.first_iter:
                # local cloud of the target frame with its pose (:457-471), transformed AGAIN by T (:642)
                maps = pointfusion.vertex_normal_maps(depths[1][0, 0], Kc[0], poses[0, 1])
                target_pc = maps["Vg"][maps["valid"]]
                moved = pointfusion.transform_pointcloud(target_pc, T[0])
                l3d, _ = knn.knn_points_loss(self.map["points"].detach().to(moved.dtype).unsqueeze(0), moved.unsqueeze(0))
                loss = loss + l3d * cfg.three3d_loss_weight
                rec["knn"] = l3d.item()
            loss.backward()
            self.opt.step()
            rec["loss"] = loss.item()
            rec["median_index"] = median_index
            rec["median_indices"] = median_indices
            rec["g_depth"] = [d.grad.detach().clone() for d in depths]
            rec["depth1"] = depths[1].detach()
            rec["metrics"] = [m.item() for m in warp_loss.depth_metrics(cfg.dataset, gt_depths[0][1], depths[1][0])]
            records.append(rec)
        if update_map:                                     # False: bench.py's cpu_baseline times the refinement steps only
            self.update_map(colors, gt_depths, poses, Kc)
            self.first_iter = False
        return records

    def flagged_image_losses(self, depths, src, tgt, Kc, invK, T):
        """online_adaption.py:412-455 (view synthesis, geometric variant :421-439) + :483-523 with the off-by-default flags: minimum
        reprojection, auto-masking, geometric consistency, smoothness on the mean-normalised disparity of frame 0."""
        import torch.nn.functional as F
        cfg = self.cfg
        pts = warp_loss.backproject(depths[1], invK)
        H, W = tgt.shape[2:]
        if cfg.geometric:
            grid, wdepth, valid = warp_loss.project(pts, Kc, T, H, W, geometric=True)
            synth = F.grid_sample(src, grid, padding_mode=cfg.padding_mode, align_corners=True)
            idepth = F.grid_sample(depths[0], grid, padding_mode=cfg.padding_mode, align_corners=False)
        else:
            grid, valid = warp_loss.project(pts, Kc, T, H, W)
            synth = F.grid_sample(src, grid, padding_mode=cfg.padding_mode, align_corners=False)
        m = valid if cfg.photometric_mask else 1.0
        photo = warp_loss.photometric(synth * m, tgt * m)               # one source frame: (1,1,H,W)
        if not cfg.min_reprojection:
            photo = photo.mean(1, keepdim=True)
        if cfg.auto_masking:
            auto = warp_loss.photometric(src * m, tgt * m)
            if cfg.min_reprojection:
                auto = auto + (cfg.tie_break_noise if cfg.tie_break_noise is not None else torch.zeros_like(auto))
            else:
                auto = auto.mean(1, keepdim=True)
            photo = torch.cat((auto, photo), 1)
        loss = photo_value = photo.mean() if photo.shape[1] == 1 else torch.min(photo, dim=1)[0].mean()
        if cfg.geometric:
            loss = loss + torch.stack([warp_loss.geometric_consistency(wdepth, idepth, valid)], 0).mean() * cfg.geometric_weight
        if cfg.smoothness:
            loss = loss + warp_loss.normalised_smoothness(self.disps[0], tgt) * cfg.smoothness_weight     # inputs[("disp", 0, 0)] (:603)
        return loss, photo_value

    @torch.no_grad()
    def update_map(self, colors, gt_depths, poses, Kc):
        """online_adaption.py:329-366 (outputs unaffected by running under no_grad: SURVEY App. C.8)."""
        depths = self.predict_depths(colors)
        depths, _ = warp_loss.median_scale(depths, gt_depths)
        cfg = self.cfg
        self.tables = []
        # the map is an fp32 artefact whatever precision the network runs in (the fp64 variant of this restatement exists to measure how
        # well-conditioned a trajectory is: tests/test_oracle_conditioning.py)
        f = lambda t: t.float()
        if self.first_iter:
            self.map, t = pointfusion.pointfusion_step(self.map, f(colors[0, 0]), f(depths[0][0, 0]), f(Kc[0]), f(poses[0, 0]),
                                                       cfg.dist_th, cfg.angle_th, cfg.sigma)
            self.tables.append(t)
        self.map, t = pointfusion.pointfusion_step(self.map, f(colors[0, 1]), f(depths[1][0, 0]), f(Kc[0]), f(poses[0, 1]),
                                                   cfg.dist_th, cfg.angle_th, cfg.sigma)
        self.tables.append(t)
