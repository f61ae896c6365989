"""MI355X counterpart of the reference's development harness, train_depth.py (BASELINE.json configs[0]): self-supervised depth
refinement on ONE short keyframe sequence (`len(DATA.frames)` = 2 or 3 frames) with a fixed depth scale
(`ABLATION.scaling_depth`), the dual-disparity flip trick, a full-sequence SLAM reconstruction of the predicted depths and of
the ground truth every refinement step, and the whole loss flag matrix of the reference:

    photometric [mask] [min-reprojection] [auto-masking]      train_depth.py:706-750, :621-661
    geometric consistency, smoothness, depth regulariser       :752-788
    sparse ground-truth supervision                            :790-799 (utils.sparse_sampling)
    knn_points / chamfer_distance against the GT reconstruction  :682-695

Same class / method names and the same order of operations as the reference (train_depth.py:40-57 __init__, :198-207
set_refinement_mode, :224-237 process_disparity, :239-428 train, :430-440 depth_refinement, :442-543 process_inputs, :545-613
novel_view_synthesis, :615-705 compute_losses); every tensor operation on the path is one of this package's HIP kernels behind
the reference's own operator names (BackprojectDepth / Project3D / grid_sample, SSIM / photometric_loss, the losses of
loss/losses.py, PointFusion / RGBDImages, knn_points / ChamferDistance, the depth network, fused Adam).  With the reference's
default flags (configs/config.yaml:37-58: masked photometric loss only) the image-space part of a step is the fused
e2e_warp_photo_* pair instead of the operator-by-operator composition -- `fused_losses=False` keeps the latter.

    python train_depth.py --config_path configs/config_train_depth.yaml
"""
import os
import sys
from collections import OrderedDict

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from chamferdist import ChamferDistance  # noqa: E402
from depth_estimation.networks import DispResNet_Indoor  # noqa: E402
from depth_estimation.view_synthesis import BackprojectDepth, Project3D  # noqa: E402
from e2ehip import ops  # noqa: E402
from e2ehip.synthetic import make_sequence  # noqa: E402
from gradslam import RGBDImages  # noqa: E402
from gradslam.slam import ICPSLAM, PointFusion  # noqa: E402
from loss.losses import (SSIM, depth_gt_loss, depth_metrics, depth_reguralizer, disparity_smoothness_loss,  # noqa: E402
                         geometric_consistency_loss, knn_points_loss, min_reprojection_loss, photometric_loss, process_disparity)
from tensorboardX import SummaryWriter  # noqa: E402
from utils.training_utils import (define_optim, define_schedular, inverse_T_matrix, sparse_sampling,  # noqa: E402
                                  torch_poses_to_transforms)
from utils.yaml_configs import load_yaml  # noqa: E402


class Depth_Estimation:
    def __init__(self, arguments, sequence=None, state_dict=None, fused_losses=True):
        """arguments: the reference's config tree.  sequence: optional pre-loaded batch (colors 0-1 (1,L,H,W,3), depths (1,L,H,W,1),
        intrinsics (1,1,4,4), poses (1,L,4,4)) instead of the ICL / TUM loader; default when no dataset is on the machine: a
        synthetic sequence with the same tensor contract."""
        self.args = arguments
        if self.args.SETTINGS.device != "cuda":
            raise RuntimeError("this implementation runs on the MI355X only (SETTINGS.device: cuda); there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.sequence_length = len(self.args.DATA.frames)
        if self.sequence_length not in (2, 3):
            raise ValueError("Sequence Length of 2 and 3 is only supported")
        self.writer = SummaryWriter("tensorboard_outputs")             # constructed unconditionally by the reference (train_depth.py:48)
        if getattr(self.args.VIZ, "tensorboard", False):
            raise NotImplementedError("out of scope: SURVEY.md section 5.5 -- TensorBoard gradient hooks (VIZ.tensorboard)")
        self._sequence, self._state_dict, self.fused_losses = sequence, state_dict, fused_losses
        self.dataset_init()
        self.model_init()
        self.view_reconstruction_init()
        self.losses_init()
        if self.args.ABLATION.scale_intrinsics:
            print("Scaling Intrinsics")
        if self.args.ABLATION.scaled_depth:
            print("Scaling Depth Maps")
        self.log = []

    # ---------------------------------------------------------------------------------------------------------------------------
    def dataset_init(self):
        a = self.args
        print("Loading Images of Size {} x {}".format(a.DATA.width, a.DATA.height))
        self.batches = None
        if self._sequence is None:
            root = os.path.join(str(getattr(a.DATA, "data_path", "") or ""), a.DATA.name)
            if a.DATA.name in ("ICL", "TUM") and os.path.isdir(root):
                from gradslam.datasets import ICL, TUM
                from torch.utils.data import DataLoader
                ds = (ICL if a.DATA.name == "ICL" else TUM)(basedir=root, seqlen=self.sequence_length, height=a.DATA.height, width=a.DATA.width,
                                                            dilation=a.DATA.dilation, stride=a.DATA.stride, start=a.DATA.start)
                self.batches = DataLoader(dataset=ds, batch_size=a.OPTIMIZATION.batch_size, shuffle=False, num_workers=0, drop_last=True)
            elif a.DATA.name in ("ICL", "TUM", "synthetic"):
                self._sequence = make_sequence(self.sequence_length, a.DATA.height, a.DATA.width, seed=int(getattr(a.DATA, "seed", 1234)),
                                               holes=0.1 if a.DATA.name == "TUM" else 0.0)
            else:
                raise ValueError("Dataset Not Found")
        print("{} Dataset Loaded".format(a.DATA.name))

    def _iter_batches(self):
        """(colors in [0,1], gt_depths, intrinsics, poses, transform) per loader item (train_depth.py:254-263)."""
        if self.batches is None:
            c, d, k, p = self._sequence
            yield c, d, k, p, torch_poses_to_transforms(p)
            return
        for batch in self.batches:
            colors, depths, K, poses, transform = batch[0], batch[1], batch[2], batch[3], batch[4]
            yield colors / 255.0, depths, K, poses, transform

    def model_init(self):
        a = self.args
        self.models = {}
        print("Initializing Models")
        if a.MODEL.slam == "ICPSLAM":
            self.models["SLAM"] = ICPSLAM(odom=a.MODEL.odom, numiters=a.MODEL.numiters, device=self.device)
            self.models["GT_SLAM"] = ICPSLAM(odom="gt", device=self.device)
        elif a.MODEL.slam == "PointFusion":
            self.models["SLAM"] = PointFusion(odom=a.MODEL.odom, dist_th=a.MODEL.dist_th, angle_th=a.MODEL.angle_th, sigma=a.MODEL.sigma,
                                              numiters=a.MODEL.numiters, device=self.device)
            self.models["GT_SLAM"] = PointFusion(odom="gt", device=self.device)
        else:
            raise ValueError("MODEL.slam must be ICPSLAM or PointFusion")
        print("Using the {} for SLAM".format(a.MODEL.slam))
        if a.MODEL.depth_network != "indoor":
            # the reference keeps a monodepth2 branch "in case we decide to do outdoor" (train_depth.py:299); its refinement code indexes
            # self.models["depth"], i.e. the indoor network -- that is the network on the path
            raise NotImplementedError("MODEL.depth_network: only the indoor network is on the refinement path")
        self.models["depth"] = DispResNet_Indoor(num_layers=a.MODEL.num_layers, pretrained=a.MODEL.weights_init_encoder == "imagenet")
        if self._state_dict is not None:
            self.models["depth"].load_state_dict(self._state_dict)
        elif a.MODEL.use_pretrained_models:
            self.load_model_indoor()
        self.models["depth"].to(self.device)
        self.train_params = list(self.models["depth"].parameters())
        self.optimizer = define_optim(a, self.train_params)
        self.schedular = define_schedular(a, self.optimizer)

    def load_model_indoor(self):
        path = os.path.join(os.path.expanduser(self.args.MODEL.load_depth_path), "depth.pth.tar")
        assert os.path.isfile(path), "Cannot find {}".format(path)
        self.models["depth"].load_state_dict(torch.load(path, map_location="cpu")["state_dict"])

    def load_optimizer(self):
        """Resume the optimiser from `<MODEL.load_depth_path>/<OPTIMIZATION.optimizer>.pth` when that file exists (train_depth.py:849-863;
        torch.optim.Adam's state-dict format, which e2ehip.optim.FusedAdam reads and writes)."""
        path = os.path.join(os.path.expanduser(str(self.args.MODEL.load_depth_path)), "{}.pth".format(self.args.OPTIMIZATION.optimizer))
        if os.path.isfile(path):
            print("Loading Optimizer Weights")
            self.optimizer.load_state_dict(torch.load(path, map_location=self.device))
        else:
            print("Optimizer Not Found. Randomly Initialized")

    def save_optimizer(self):
        """Writes the file load_optimizer() reads (the reference leaves saving as a TODO, train_depth.py:847)."""
        path = os.path.join(os.path.expanduser(str(self.args.MODEL.load_depth_path)), "{}.pth".format(self.args.OPTIMIZATION.optimizer))
        torch.save(self.optimizer.state_dict(), path)
        return path

    def view_reconstruction_init(self):
        a = self.args
        self.backproject_depth = BackprojectDepth(a.OPTIMIZATION.batch_size, a.DATA.height, a.DATA.width)
        self.project_3d = Project3D(a.OPTIMIZATION.batch_size, a.DATA.height, a.DATA.width)

    def losses_init(self):
        self.ssim = SSIM()
        self.chamfer = ChamferDistance()

    def set_refinement_mode(self):
        """eval mode everywhere + freeze every parameter whose NAME contains "bn" (train_depth.py:198-207)."""
        for m in self.models.values():
            m.eval()
            for name, p in m.named_parameters():
                if name.find("bn") != -1:
                    p.requires_grad = False

    def process_disparity(self, inputs, index):
        """train_depth.py:224-237: blend of the disparity of the frame and of its flipped copy (kernel: e2e_disp_blend_*)."""
        inputs[("disp", index, 0)] = process_disparity(inputs[("disp", index, 0)])
        return inputs

    # ---------------------------------------------------------------------------------------------------------------------------
    def train(self):
        a = self.args
        self.epoch = self.step = 0
        if a.MODEL.refinement_mode:
            self.set_refinement_mode()
        print("SLAM Reconstruction Started")
        for it, (colors, gt_depths, intrinsics, poses, transform) in enumerate(self._iter_batches()):
            colors, gt_depths, intrinsics, poses, transform = (t.to(self.device).contiguous() for t in (colors, gt_depths, intrinsics, poses, transform))
            rgbd = RGBDImages(colors, gt_depths, intrinsics, poses)
            with torch.no_grad():
                self.gt_reconstruction, _ = self.models["GT_SLAM"](rgbd)             # GT reconstruction (train_depth.py:265-267)
            self.gt_reconstruction = self.gt_reconstruction.detach()
            scale = 0
            self.initial_depths = {}
            for refine_step in range(a.OPTIMIZATION.refinement_steps):
                inputs = OrderedDict()
                depth_tensor = []
                for index in range(self.sequence_length):
                    frame = colors[:, index]
                    if a.ABLATION.dual_disparity:
                        inputs.update(self.models["depth"](torch.cat([frame, torch.flip(frame, [2])], 0), index))
                        inputs.update(self.process_disparity(inputs, index))
                    else:
                        inputs.update(self.models["depth"](frame, index))
                    k = 1.0
                    if a.ABLATION.scale_intrinsics:
                        k *= float(intrinsics[0, 0, 0, 0]) / a.ABLATION.focal_pretrain                  # scale_by_f (training_utils.py:142-152)
                    if a.ABLATION.scaled_depth:
                        k *= a.ABLATION.scaling_depth                                                    # fixed "median" scale (:343-345)
                    inputs[("depth", index, scale)] = ops.depth_from_disp_fixed_scale(inputs[("disp", index, scale)], k)
                    if a.OPTIMIZATION.refinement == "PFT" and a.LOSS.depth_regularizer and refine_step == 0:
                        self.initial_depths[("initial_depth", index, scale)] = inputs[("depth", index, scale)].clone().detach()
                    depth_tensor.append(inputs[("depth", index, scale)].unsqueeze(1))
                    inputs[("gt_depth", index, scale)] = gt_depths[:, index]
                depth_tensor = torch.cat(depth_tensor, dim=1).permute(0, 1, 3, 4, 2)
                if a.DATA.use_gt_pose:
                    new_poses = poses
                elif self.step == 0:
                    new_poses = torch.eye(4, device=self.device).view(1, 1, 4, 4).repeat(a.OPTIMIZATION.batch_size, self.sequence_length, 1, 1)
                noisy_rgbd = RGBDImages(rgb_image=colors, depth_image=depth_tensor, intrinsics=intrinsics, poses=new_poses)
                want_cloud = a.LOSS.knn_points or a.LOSS.chamfer_distance
                if a.DATA.use_gt_pose and not want_cloud:
                    # the reference reconstructs here in every step (:378) and uses the result only under knn_points / chamfer_distance
                    # (:682-695); with ground-truth poses the reconstruction has no other consumer, so it is skipped
                    noisy_reconstruction, new_transform = None, transform
                elif a.DATA.use_gt_pose:
                    noisy_reconstruction, _ = self.models["SLAM"](noisy_rgbd)
                    new_transform = transform
                else:
                    noisy_reconstruction, new_poses = self.models["SLAM"](noisy_rgbd)
                    new_transform = torch_poses_to_transforms(new_poses)
                if noisy_reconstruction is not None:
                    inputs["noisy_pointcloud"] = noisy_reconstruction.points_list[0].unsqueeze(0).contiguous()
                total_loss = self.depth_refinement(colors, inputs, intrinsics, new_transform)
                new_poses = new_poses.detach()
                self.step += 1
                if a.DEBUG.print_metrics:
                    m = depth_metrics(dataset="TUM" if a.DATA.name == "TUM" else "ICL", gt=gt_depths[0][1], pred=inputs[("depth", 1, 0)][0])
                    print("Iter:", it, "Refine_Step:", refine_step, "Total_Loss:", round(total_loss, 5), "abs_rel: ", round(m[0].item(), 5),
                          "rmse: ", round(m[2].item(), 5), "a1: ", round(m[4].item(), 5))
                else:
                    print("Iter:", it, "Refine_Step:", refine_step, "Total_Loss:", round(total_loss, 5))
                self.log.append(total_loss)
            self.schedular.step()
            if a.DEBUG.early_stop and it == a.DEBUG.iter_stop:
                break
        return self.log

    def depth_refinement(self, colors, inputs, intrinsics, poses):
        outputs = {}
        inputs.update(self.process_inputs(colors, inputs, intrinsics, poses))
        if self._fused_ok():
            return self.compute_losses_fused(inputs)
        outputs.update(self.novel_view_synthesis(inputs))
        return self.compute_losses(inputs, outputs)

    # ---------------------------------------------------------------------------------------------------------------------------
    def process_inputs(self, colors, inputs, intrinsics, poses):
        """train_depth.py:442-543: which frame is the target, which are the sources, and the relative poses."""
        a = self.args
        nhwc = lambda i: colors[:, i].permute(0, 3, 1, 2)                 # NCHW views of NHWC memory: read in place by the kernels
        inputs["K"] = intrinsics[:, 0]
        if getattr(a.DATA, "normalize_intrinsics", False):
            raise NotImplementedError("DATA.normalize_intrinsics applies to the monodepth2 branch only (train_depth.py:454)")
        inputs["Inverse_K"] = torch.pinverse(inputs["K"])
        if self.sequence_length == 3:
            inputs["source_frame", -1], inputs["target_frame"], inputs["source_frame", 1] = nhwc(0), nhwc(1), nhwc(2)
            inputs["source_depth", -1], inputs["target_depth"], inputs["source_depth", 1] = inputs["depth", 0, 0], inputs["depth", 1, 0], inputs["depth", 2, 0]
            inputs["source_disp", -1], inputs["target_disp"], inputs["source_disp", 1] = inputs["disp", 0, 0], inputs["disp", 1, 0], inputs["disp", 2, 0]
            inputs["T", -1] = poses[:, 1].contiguous()
            inputs["T", 1] = inverse_T_matrix(poses[:, 2]).contiguous()
            self._tgt_index = 1
        elif a.DATA.frames[1] < 0:
            inputs["source_frame", -1], inputs["target_frame"] = nhwc(0), nhwc(1)
            inputs["source_depth", -1], inputs["target_depth"] = inputs["depth", 0, 0], inputs["depth", 1, 0]
            inputs["source_disp", -1], inputs["target_disp"] = inputs["disp", 0, 0], inputs["disp", 1, 0]
            inputs["T", -1] = poses[:, 1].contiguous()
            self._tgt_index = 1
        else:
            inputs["target_frame"], inputs["source_frame", 1] = nhwc(0), nhwc(1)
            inputs["target_depth"], inputs["source_depth", 1] = inputs["depth", 0, 0], inputs["depth", 1, 0]
            inputs["target_disp"], inputs["source_disp", 1] = inputs["disp", 0, 0], inputs["disp", 1, 0]
            inputs["T", 1] = inverse_T_matrix(poses[:, 1]).contiguous()
            self._tgt_index = 0
        if a.LOSS.supervise_depth:
            for index in range(self.sequence_length):
                inputs[("sparse_gt_depth", index, 0)], inputs[("sparse_mask", index, 0)] = sparse_sampling(
                    a.LOSS.sampling_type, a.LOSS.sampling_prob, inputs[("gt_depth", index, 0)])
        return inputs

    def novel_view_synthesis(self, inputs):
        """train_depth.py:545-613 operator by operator (each a HIP kernel behind the reference's module names)."""
        a = self.args
        outputs = {}
        for frame in a.DATA.frames[1:]:
            camera_points = self.backproject_depth(inputs["target_depth"], inputs["Inverse_K"])
            if a.LOSS.geometric:
                grid, warped_depth, valid = self.project_3d(points=camera_points, K=inputs["K"], T=inputs["T", frame], geometric=True)
                outputs[("warped_depth", frame)] = warped_depth
                outputs[("valid_mask", frame)] = valid
                outputs[("synthesized_frame", frame)] = ops.grid_sample(inputs["source_frame", frame], grid, padding_mode=a.MODEL.padding_mode,
                                                                        align_corners=True)        # the reference's geometric branch (:570-573)
                outputs[("interpolated_depth", frame)] = ops.grid_sample(inputs["source_depth", frame], grid, padding_mode=a.MODEL.padding_mode,
                                                                         align_corners=False)
            else:
                grid, valid = self.project_3d(points=camera_points, K=inputs["K"], T=inputs["T", frame], geometric=False)
                outputs[("valid_mask", frame)] = valid
                outputs[("synthesized_frame", frame)] = ops.grid_sample(inputs["source_frame", frame], grid, padding_mode=a.MODEL.padding_mode,
                                                                        align_corners=False)
        return outputs

    # ---------------------------------------------------------------------------------------------------------------------------
    def _fused_ok(self):
        lo = self.args.LOSS
        return self.fused_losses and self.sequence_length == 2 and lo.photometric_mask is not None and not (
            lo.min_reprojection or lo.auto_masking or lo.geometric)

    def compute_losses_fused(self, inputs):
        """compute_losses for the reference's default flag set with the image-space part as ONE fused forward (warp + mask + SSIM/L1 +
        mean) and ONE fused backward (e2e_warp_photo_fwd / _bwd): photometric [+ regulariser], then the remaining terms."""
        a = self.args
        frame = a.DATA.frames[1]
        self.optimizer.zero_grad()
        out = ops.warp_photometric(inputs["target_depth"], inputs["source_frame", frame], inputs["target_frame"], inputs["K"], inputs["Inverse_K"],
                                   inputs["T", frame], padding_mode=a.MODEL.padding_mode, use_mask=bool(a.LOSS.photometric_mask))
        loss = out["photometric"]
        loss = self._add_common_terms(loss, inputs, None)
        loss.backward()
        self.optimizer.step()
        return loss.item()

    def compute_losses(self, inputs, outputs):
        """train_depth.py:615-705."""
        a = self.args
        self.optimizer.zero_grad()
        photometric = self.compute_photometric_loss(inputs, outputs)                 # (B, #sources, H, W)
        if a.LOSS.auto_masking:
            auto = self.compute_automasking_loss(inputs, outputs)
            if a.LOSS.min_reprojection:
                auto = auto + torch.randn(auto.shape, device=auto.device) * 0.00001     # "Break tie's" (:650)
            else:
                auto = ops.channel_mean(auto)
                photometric = ops.channel_mean(photometric)
            photometric = torch.cat((auto, photometric), dim=1)
        elif not a.LOSS.min_reprojection:
            photometric = ops.channel_mean(photometric)
        # one map: its mean; several: mean of the per-pixel minimum, gradient to the first minimal map (e2e_min_reprojection_lossgrad)
        loss = min_reprojection_loss(photometric)
        loss = self._add_common_terms(loss, inputs, outputs)
        loss.backward()
        self.optimizer.step()
        return loss.item()

    def _add_common_terms(self, loss, inputs, outputs):
        a = self.args
        if a.LOSS.geometric:
            geometric = self.compute_geometric_loss(outputs).mean()
            loss = loss + geometric * a.LOSS.geometric_weight
        if a.LOSS.smoothness:
            loss = loss + self.compute_smoothness_loss(inputs) * a.LOSS.smoothness_weight
        if a.LOSS.depth_regularizer:
            loss = loss + self.compute_depth_regularizer(inputs) * a.LOSS.depth_regularizer_weight
        if a.LOSS.knn_points:
            knn_loss, _ = knn_points_loss(gt_pointcloud=self.gt_reconstruction.points_list[0].unsqueeze(0).contiguous(),
                                          noisy_pointcloud=inputs["noisy_pointcloud"])
            loss = loss + knn_loss * a.LOSS.knn_points_weight
            print("knn_loss", knn_loss.item())
        if a.LOSS.chamfer_distance:
            chamfer_dist = 0.5 * self.chamfer(inputs["noisy_pointcloud"], self.gt_reconstruction.points_list[0].unsqueeze(0).contiguous(),
                                              bidirectional=True)
            loss = loss + chamfer_dist * a.LOSS.chamfer_weight
            print("chamfer_loss", chamfer_dist.item())
        if a.LOSS.supervise_depth:
            loss = loss + self.compute_gt_depth_loss(inputs) * a.LOSS.gt_depth_weight
        return loss

    def compute_photometric_loss(self, inputs, outputs):
        maps = []
        for frame in self.args.DATA.frames[1:]:
            pred, tgt = outputs[("synthesized_frame", frame)], inputs["target_frame"]
            if self.args.LOSS.photometric_mask:
                pred, tgt = ops.mask_mul(pred, outputs["valid_mask", frame]), ops.mask_mul(tgt, outputs["valid_mask", frame])
            maps.append(photometric_loss(ssim=self.ssim, prediction=pred, target=tgt))
        return torch.cat(maps, 1) if len(maps) > 1 else maps[0]

    def compute_automasking_loss(self, inputs, outputs):
        maps = []
        for frame in self.args.DATA.frames[1:]:
            pred, tgt = inputs[("source_frame", frame)], inputs["target_frame"]
            if self.args.LOSS.photometric_mask:
                pred, tgt = ops.mask_mul(pred, outputs["valid_mask", frame]), ops.mask_mul(tgt, outputs["valid_mask", frame])
            maps.append(photometric_loss(ssim=self.ssim, prediction=pred, target=tgt))
        return torch.cat(maps, 1) if len(maps) > 1 else maps[0]

    def compute_geometric_loss(self, outputs):
        return torch.stack([geometric_consistency_loss(outputs, frame, self.device) for frame in self.args.DATA.frames[1:]], dim=0)

    def compute_smoothness_loss(self, inputs):
        """mean-normalised disparity of frame 0 against the target image (train_depth.py:763-773)."""
        return disparity_smoothness_loss(disp=ops.mean_normalize(inputs[("disp", 0, 0)]), img=inputs["target_frame"])

    def compute_depth_regularizer(self, inputs):
        reg = 0
        for frame in range(self.sequence_length):
            reg = reg + depth_reguralizer(initial_depth=self.initial_depths[("initial_depth", frame, 0)], refined_depth=inputs[("depth", frame, 0)],
                                          loss_func=self.args.LOSS.depth_regularizer_type)
        return reg

    def compute_gt_depth_loss(self, inputs):
        gt_loss = 0
        for frame in range(self.sequence_length):
            gt_loss = gt_loss + depth_gt_loss(prediction=inputs[("depth", frame, 0)], sparse_groundtruth=inputs[("sparse_gt_depth", frame, 0)],
                                              sparse_mask=inputs[("sparse_mask", frame, 0)])
        return gt_loss


def default_config(height=256, width=320, frames=(0, -1), refinement_steps=3):
    """The reference's configs/config.yaml restricted to the keys train_depth.py reads, with its default values
    (dual disparity, fixed scale 6.9, masked photometric loss only, PointFusion + ground-truth poses)."""
    from utils.yaml_configs import _wrap
    return _wrap(OrderedDict(
        SETTINGS=dict(name="Training_1", num_workers=0, device="cuda"),
        DATA=dict(name="synthetic", height=height, width=width, frames=list(frames), scales=[0], seed=1234, dilation=2, stride=2, start=418,
                  use_gt_pose=True, normalize_intrinsics=False, min_depth=0.1, max_depth=80.0),
        MODEL=dict(depth_network="indoor", num_layers=18, weights_init_encoder=False, use_pretrained_models=False, load_depth_path="",
                   slam="PointFusion", odom="gt", dist_th=0.05, angle_th=20, sigma=0.6, numiters=20, padding_mode="border", refinement_mode=True),
        LOSS=dict(chamfer_distance=False, chamfer_weight=0.25, knn_points=False, knn_points_weight=0.25, auto_masking=False, min_reprojection=False,
                  photometric_mask=True, geometric=False, geometric_weight=0.5, smoothness=False, smoothness_weight=1e-3, depth_regularizer=False,
                  depth_regularizer_weight=1e-2, depth_regularizer_type="l2", supervise_depth=False, sampling_type="random", sampling_prob=0.012,
                  gt_depth_weight=1, three3d_loss=True, three3d_loss_weight=1.0),
        OPTIMIZATION=dict(batch_size=1, refinement="PFT", refinement_steps=refinement_steps, learning_rate=1e-5, optimizer="Adam", schedular="StepLR",
                          schedular_step_size=100, schedular_gamma=0.5),
        ABLATION=dict(scale_intrinsics=False, focal_pretrain=285.8, scaled_depth=True, scaling_depth=6.9, dual_disparity=True),
        VIZ=dict(plot_first_step=False, plot_gt=False, plot_final_step=False, tensorboard=False),
        DEBUG=dict(early_stop=True, iter_stop=0, plot=False, print_metrics=True),
    ))


if __name__ == "__main__":
    from utils.arguments import arguments
    cli = arguments()
    cfg = load_yaml(cli["config_path"])
    cfg.SETTINGS.name = cli["name"]
    Depth_Estimation(cfg).train()
