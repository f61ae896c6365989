"""CPU restatement of the reference's development harness train_depth.py for ONE loader batch (the keyframe pair / triple):
refinement steps with the fixed depth scale, the dual-disparity blend and the loss flag matrix.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED for the driver glue: train_depth.py cannot be imported here (it imports gradslam / chamferdist / tensorboardX at
module level), so this file restates its text -- train_depth.py:224-237 (process_disparity), :239-428 (train),
:442-543 (process_inputs), :545-613 (novel_view_synthesis), :615-705 (compute_losses), :706-799 (loss helpers) -- on top of the
oracle parts that ARE pinned by golden vectors (warp_loss.py: g1-g5, depthnet.py decoder: g7).  The SLAM reconstruction of the
predicted depths (train_depth.py:378) is only consumed by the knn / chamfer terms (off by default) and is not restated here."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import depthnet, poses as oposes, warp_loss


class Config:
    frames = (0, -1)
    refinement_steps = 3
    learning_rate = 1e-5
    padding_mode = "border"
    dual_disparity = True
    scaled_depth = True
    scaling_depth = 6.9
    photometric_mask = True
    min_reprojection = False
    auto_masking = False
    geometric = False
    geometric_weight = 0.5
    smoothness = False
    smoothness_weight = 1e-3
    depth_regularizer = False
    depth_regularizer_weight = 1e-2
    depth_regularizer_type = "l2"
    supervise_depth = False
    gt_depth_weight = 1
    tie_break_noise = None          # auto-masking + min-reprojection adds randn * 1e-5 (train_depth.py:650); a fixed tensor for tests


class Trainer:
    def __init__(self, state_dict, cfg=None):
        self.cfg = cfg or Config()
        self.sd = OrderedDict((k, v.clone()) for k, v in state_dict.items())
        self.train_keys = depthnet.trainable_keys(self.sd)
        for k in self.train_keys:
            self.sd[k].requires_grad_(True)
        self.opt = torch.optim.Adam([self.sd[k] for k in self.train_keys], lr=self.cfg.learning_rate)

    def predict(self, frame):
        """frame (1,H,W,3) -> disparity (1,1,H,W) (train_depth.py:322-329)."""
        if self.cfg.dual_disparity:
            pair = depthnet.disp_forward(self.sd, torch.cat([frame, torch.flip(frame, [2])], 0))
            return warp_loss.process_disparity(pair)
        return depthnet.disp_forward(self.sd, frame)

    def train_batch(self, colors, gt_depths, poses, K, sparse=None):
        """colors (1,L,H,W,3) in [0,1], gt (1,L,H,W,1), poses (1,L,4,4), K (1,1,4,4) -> list of per-step records."""
        c = self.cfg
        L = colors.shape[1]
        transform = oposes.poses_to_transforms(poses)
        Kc = K[:, 0]
        invK = torch.pinverse(Kc)
        nchw = lambda i: colors[:, i].permute(0, 3, 1, 2)
        recs, initial = [], None
        for step in range(c.refinement_steps):
            disps = [self.predict(colors[:, i]) for i in range(L)]
            depths = [(1 / d) * (c.scaling_depth if c.scaled_depth else 1.0) for d in disps]
            if step == 0 and c.depth_regularizer:
                initial = [d.clone().detach() for d in depths]
            # process_inputs: target / sources and relative poses
            if L == 3:
                tgt_i, srcs = 1, {-1: (0, transform[:, 1]), 1: (2, oposes.inverse_T(transform[:, 2]))}
            elif c.frames[1] < 0:
                tgt_i, srcs = 1, {-1: (0, transform[:, 1])}
            else:
                tgt_i, srcs = 0, {1: (1, oposes.inverse_T(transform[:, 1]))}
            tgt = nchw(tgt_i)
            self.opt.zero_grad()
            maps, autos, geos = [], [], []
            for f in c.frames[1:]:
                si, T = srcs[f]
                pts = warp_loss.backproject(depths[tgt_i], invK)
                if c.geometric:
                    grid, wdepth, valid = warp_loss.project(pts, Kc, T, *tgt.shape[2:], geometric=True)
                    synth = F.grid_sample(nchw(si), grid, padding_mode=c.padding_mode, align_corners=True)
                    idepth = F.grid_sample(depths[si], grid, padding_mode=c.padding_mode, align_corners=False)
                    geos.append(warp_loss.geometric_consistency(wdepth, idepth, valid))
                else:
                    grid, valid = warp_loss.project(pts, Kc, T, *tgt.shape[2:])
                    synth = F.grid_sample(nchw(si), grid, padding_mode=c.padding_mode, align_corners=False)
                m = valid if c.photometric_mask else 1.0
                maps.append(warp_loss.photometric(synth * m, tgt * m))
                if c.auto_masking:
                    autos.append(warp_loss.photometric(nchw(si) * m, tgt * m))
            photo = torch.cat(maps, 1)
            if not c.min_reprojection:
                photo = photo.mean(1, keepdim=True)
            if c.auto_masking:
                auto = torch.cat(autos, 1)
                if c.min_reprojection:
                    auto = auto + (c.tie_break_noise if c.tie_break_noise is not None else torch.zeros_like(auto))
                else:
                    auto = auto.mean(1, keepdim=True)
                photo = torch.cat((auto, photo), 1)
            loss = photo.mean() if photo.shape[1] == 1 else torch.min(photo, dim=1)[0].mean()
            rec = {"photometric": loss.item()}
            if c.geometric:
                g = torch.stack(geos, 0).mean()
                loss = loss + g * c.geometric_weight
            if c.smoothness:
                loss = loss + warp_loss.normalised_smoothness(disps[0], tgt) * c.smoothness_weight
            if c.depth_regularizer:
                loss = loss + sum(warp_loss.depth_regularizer(initial[i], depths[i], c.depth_regularizer_type) for i in range(L)) * c.depth_regularizer_weight
            if c.supervise_depth:
                loss = loss + sum(warp_loss.depth_gt(depths[i], sparse[i][0], sparse[i][1]) for i in range(L)) * c.gt_depth_weight
            loss.backward()
            self.opt.step()
            rec["loss"] = loss.item()
            rec["depth1"] = depths[1].detach()
            recs.append(rec)
        return recs
