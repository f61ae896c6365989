"""MI355X counterpart of the reference's final system, online_adaption.py: online depth refinement on keyframe
pairs + PointFusion mapping, same class / method names and the same order of operations, but launched the
MI355X way:

  * the keyframe pair goes through the depth network as ONE batch of two (BN is in eval mode);
  * 1/disp, the median ratio and its backward chain are the e2e_depth_scale_* kernels;
  * warp + mask + SSIM/L1 + depth regulariser, forward AND backward, are ONE launch (e2e_warp_photo_lossgrad)
    whose d(loss)/d(depth) is injected into the network's backward -- synth / valid / loss map never reach HBM;
  * the 3-D loss is unproject -> rigid transform -> exact nearest neighbour (HIP kernels) on the resident map;
  * Adam is one fused launch over one flat parameter bucket, which is also the single RCCL all-reduce bucket when
    several GPUs refine several sequences (one sequence per rank);
  * a whole refinement step is a static launch plan over resident buffers, replayed from captured hipGraphs
    (e2ehip.stepplan / e2ehip.netplan): no autograd graph, no allocator traffic, no host synchronisation in a step;
    `refinement_autograd` keeps the torch.autograd form of the same step (same kernels) for the off-by-default loss
    terms and as the cross-check of the plan;
  * the global map is a resident e2ehip.FusionMap updated in place.

reference: online_adaption.py:39-57 (SLAM.__init__), :98-155 (model_init), :175-205, :207-257 (main),
:259-327 (refinement), :329-366 (create_refined_pointcloud), :369-645 (losses).

    python online_adaption.py --config_path configs/config_synthetic.yaml
"""
import os
import sys
from collections import OrderedDict

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from depth_estimation.networks import DispResNet_Indoor  # noqa: E402
from depth_estimation.view_synthesis import BackprojectDepth, Project3D  # noqa: E402
from e2ehip import conv as e2e_conv  # noqa: E402
from e2ehip import dist as edist  # noqa: E402
from e2ehip import ops  # noqa: E402
from e2ehip.fused import LossGradPlan  # noqa: E402
from e2ehip.fusionmap import FusionMap  # noqa: E402
from e2ehip.stepplan import RefineStepPlan  # noqa: E402
from e2ehip.synthetic import make_sequence  # noqa: E402
from loss.losses import (SSIM, depth_gt_loss, depth_reguralizer, disparity_smoothness_loss, geometric_consistency_loss,  # noqa: E402
                         min_reprojection_loss, photometric_loss)
from utils.training_utils import define_optim, define_schedular, sparse_sampling, torch_poses_to_transforms  # noqa: E402
from utils.yaml_configs import load_yaml  # noqa: E402


class SLAM:
    def __init__(self, arguments, sequence=None, state_dict=None):
        """arguments: the reference's config tree (configs/config.yaml).  sequence: optional pre-loaded
        (colors 0-1 (1,L,H,W,3), depths (1,L,H,W,1), intrinsics (1,1,4,4), poses (1,L,4,4)); default: synthetic."""
        self.args = arguments
        if self.args.SETTINGS.device != "cuda":
            raise RuntimeError("this implementation runs on the MI355X only (SETTINGS.device: cuda); there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.sequence_length = self.args.DEMO.sequence_length
        self._sequence, self._state_dict = sequence, state_dict
        self.dataset_init()
        self.model_init()
        self.mean_abs = []
        self.log = []
        self.refinement_steps_done = 0
        # backward-weight chains on a second stream next to backward-data.  Launch plan (captured graphs): OFF by default -- a graph
        # replay pays 10-16 us for every dependency that crosses queues (profiles/r02_*timeline*), and one convolution GEMM already
        # fills the GPU: 6.35-6.42 ms/step on one stream against 6.67-6.77 with the side stream.  torch.autograd form (one launch at a
        # time from the host): ON by default, the second stream hides launch latency there.  E2E_WGRAD_OVERLAP=0/1 forces both.
        env = os.environ.get("E2E_WGRAD_OVERLAP")
        self.overlap_wgrad = env == "1"
        self.overlap_wgrad_autograd = env != "0"
        self.use_graphs = os.environ.get("E2E_STEP_GRAPHS", "1") == "1"
        self.step_plan = None
        # frame whose forward pass (current weights) sits in batch slot 1 of the step plan: the map-update forward of keyframe (p, c) is the
        # first forward of frame c in keyframe (c, n) -- same weights, same input -- and is not computed twice.  E2E_REUSE_FORWARD=0: off
        self.reuse_forward = os.environ.get("E2E_REUSE_FORWARD", "1") == "1"
        self._forward_holds = None
        self.prefetch_forward = os.environ.get("E2E_PREFETCH_FORWARD", "0") == "1"
        # test hooks (the median ELEMENT of the predictions is where two correct fp32 evaluations of this loop can part: among 614 400
        # depths the median's neighbours lie ~1e-6 away; tests name one run's elements to the other and compare everything else):
        # median_elements[k]: device int32 indices for refinement step k of this object; median_elements_log: filled when it is a list
        self.median_elements = None
        self.median_elements_log = None
        self._preloaded = None             # keyframe pair whose inputs already sit in the plan's buffers (refinement(next_pair=...))

    # ------------------------------------------------------------------------------------------------
    def dataset_init(self):
        a = self.args
        if self._sequence is None:
            if a.DATA.name not in ("ICL", "TUM", "synthetic"):
                raise ValueError("Dataset Not Found")
            root = os.path.join(str(getattr(a.DATA, "data_path", "") or ""), a.DATA.name)
            if a.DATA.name in ("ICL", "TUM") and os.path.isdir(root):
                # the reference's loader call (online_adaption.py:69-84); the first item is the whole sequence (:212)
                from gradslam.datasets import ICL, TUM
                ds = (ICL if a.DATA.name == "ICL" else TUM)(basedir=root, seqlen=self.sequence_length, height=a.DATA.height, width=a.DATA.width,
                                                            dilation=a.DATA.dilation, stride=a.DATA.stride, start=a.DATA.start)
                colors, depths, K, poses = ds[0][:4]
                self._sequence = (colors[None] / 255.0, depths[None], K[None], poses[None])            # `colors /= 255.0` (:215)
            else:
                # no dataset on this machine: a synthetic sequence with the same tensor contract (colours already in [0,1])
                self._sequence = make_sequence(self.sequence_length, a.DATA.height, a.DATA.width, seed=int(getattr(a.DATA, "seed", 1234)),
                                               holes=0.1 if a.DATA.name == "TUM" else 0.0)
        self._K_host = self._sequence[2][0, 0].detach().cpu().clone()
        self.colors, self.gt_depths, self.intrinsics, self.poses = (t.to(self.device).contiguous() for t in self._sequence)
        _, L, self.H, self.W, _ = self.colors.shape
        self.sequence_length = min(self.sequence_length, L)

    def model_init(self):
        a = self.args
        self.models = {}
        if a.MODEL.slam != "PointFusion":
            raise NotImplementedError("MODEL.slam: only PointFusion is on the online-adaption path")
        if a.MODEL.odom not in ("gt", "icp", "gradicp"):
            raise ValueError("MODEL.odom must be gt, icp or gradicp")
        self.estimated_poses = []          # (frame index, estimated pose, ground-truth pose) when odometry runs
        if a.MODEL.depth_network != "indoor":
            raise ValueError("Given {} is not a valid depth network option".format(a.MODEL.depth_network))
        self.models["depth"] = DispResNet_Indoor(num_layers=a.MODEL.num_layers, pretrained=a.MODEL.weights_init_encoder == "imagenet")
        if self._state_dict is not None:
            self.models["depth"].load_state_dict(self._state_dict)
        elif a.MODEL.use_pretrained_models:
            self.load_model_indoor()
        self.models["depth"].to(self.device)
        self.train_params = list(self.models["depth"].parameters())
        self.optimizer = define_optim(a, self.train_params)
        self.schedular = define_schedular(a, self.optimizer)
        cap = int(getattr(a.MODEL, "map_capacity", 0)) or (self.sequence_length + 1) * self.H * self.W
        self.map = FusionMap(cap, self.H, self.W, self.device, a.MODEL.dist_th, a.MODEL.angle_th, a.MODEL.sigma)
        reg = a.LOSS.depth_regularizer_type if a.LOSS.depth_regularizer else None
        self.plan = LossGradPlan(1, self.H, self.W, self.device, a.MODEL.padding_mode, a.LOSS.photometric_mask, reg,
                                 1.0, a.LOSS.depth_regularizer_weight if reg else 0.0)
        # the off-by-default loss terms go operator by operator (refinement_autograd): the reference's modules of the same names
        self.backproject_depth = BackprojectDepth(a.OPTIMIZATION.batch_size, self.H, self.W)
        self.project_3d = Project3D(a.OPTIMIZATION.batch_size, self.H, self.W)
        self.ssim = SSIM()

    def load_model_indoor(self):
        path = os.path.join(os.path.expanduser(self.args.MODEL.load_depth_path), "depth.pth.tar")
        assert os.path.isfile(path), "Cannot find {}".format(path)
        self.models["depth"].load_state_dict(torch.load(path, map_location="cpu")["state_dict"])

    def set_refinement_mode(self):
        """eval mode everywhere + freeze every parameter whose NAME contains "bn" (online_adaption.py:175-184)."""
        for m in self.models.values():
            m.eval()
            for name, p in m.named_parameters():
                if name.find("bn") != -1:
                    p.requires_grad = False

    @staticmethod
    def compute_frame_distance(prev, cur):
        """distance between the camera centres -R^T t of two (1,4,4) extrinsics (online_adaption.py:186-205)."""
        pc = -1 * torch.matmul(prev[0, :3, :3].transpose(0, 1), prev[0, :3, -1])
        cc = -1 * torch.matmul(cur[0, :3, :3].transpose(0, 1), cur[0, :3, -1])
        return torch.linalg.norm(pc - cc)

    # ------------------------------------------------------------------------------------------------
    def keyframe_schedule(self):
        """[(previous keyframe, new keyframe)] decided once on the host -- poses are dataset inputs, not results, so the
        reference's per-frame device sync (`if dist > threshold`, online_adaption.py:231-234) is not needed."""
        poses_h = self.poses.cpu()
        prev, out = 0, []
        for frame in range(1, self.sequence_length):
            if self.compute_frame_distance(poses_h[:, prev], poses_h[:, frame]) > self.args.DEMO.frame_threshold:
                out.append((prev, frame))
                prev = frame
        return out

    def main(self):
        if self.args.MODEL.refinement_mode:
            self.set_refinement_mode()
        self.first_iter = True
        schedule = self.keyframe_schedule()
        # one sequence per rank: every rank joins the same number of gradient exchanges; a rank whose
        # sequence has fewer keyframes idles through the surplus rounds as a non-participant
        rounds = edist.common_rounds(len(schedule), self.device)
        if self._plan_eligible():
            self._step_plan()
        elif edist.world() > 1:
            self.optimizer.prebuild(self.models["depth"].used_parameters())
            edist.broadcast_parameters_(self.optimizer.flat)
        for i in range(rounds):
            if i < len(schedule):
                self.refinement(*schedule[i], next_pair=schedule[i + 1] if i + 1 < len(schedule) else None)
                self.first_iter = False
            else:
                self.idle_round()
        if self.args.DEBUG.print_metrics and self.mean_abs:
            print(torch.tensor(self.mean_abs).mean().item())
        self.map.check_capacity()               # the one host read of the map size (and of its overflow flag) of the whole run
        if getattr(self, "_odo", None) is not None:
            self._odo.check()                   # ... and of the resident odometry's error flags
        if edist.world() > 1:                   # end of run: variable-length gather of the per-rank maps
            self.gathered_map = edist.gather_maps(*self.map.live(), dst=0)     # rank 0 receives the maps, every rank their sizes
        return self.map

    def close(self):
        """End of this SLAM object's GPU work: the step plan's streams joined and its graphs destroyed (RefineStepPlan.close), the
        device idle.  main() may not be called again afterwards; map and network stay readable."""
        if self.step_plan is not None:
            self.step_plan.close()
            self.step_plan = None
        torch.cuda.synchronize(self.device)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def idle_round(self):
        """A keyframe round of another rank: contribute a zero bucket to each of its gradient exchanges and apply the
        same averaged update, so that the shared depth network stays identical on every rank."""
        self._forward_holds = None                  # the averaged update changes the weights
        for _ in range(self.args.OPTIMIZATION.refinement_steps):
            if self.step_plan is not None:
                self.step_plan.idle_step()
                continue
            self.optimizer.zero_grad()
            self._exchange_gradients(participating=False)
            self.optimizer.step()

    def _pair(self, prev, cur):
        idx = torch.tensor([prev, cur], device=self.device)
        colors = self.colors[:, idx]            # (1,2,H,W,3)
        gt = self.gt_depths[:, idx]
        poses = self.poses[:, idx]
        return colors, gt, poses

    def reset_map(self):
        """Start a new sequence pass with the current network: empty global map, first-keyframe rules apply again."""
        self.map.M = 0
        self.first_iter = True
        self.estimated_poses = []

    def _plan_eligible(self):
        """The launch plan covers the reference's recommended loss set (README.md:146-158: photometric [+ mask] + depth regulariser
        + 3-D loss); the off-by-default terms go through refinement_autograd."""
        lo = self.args.LOSS
        return not any(getattr(lo, f, False) for f in ("geometric", "smoothness", "supervise_depth", "auto_masking", "min_reprojection"))

    def _step_plan(self):
        if self.step_plan is None:
            a = self.args
            use_reg = a.LOSS.depth_regularizer and a.OPTIMIZATION.refinement == "PFT"
            self.step_plan = RefineStepPlan(self.models["depth"], self.optimizer, self.H, self.W, self.device, a.MODEL.padding_mode,
                                            a.LOSS.photometric_mask, a.LOSS.depth_regularizer_type if use_reg else None,
                                            a.LOSS.depth_regularizer_weight if use_reg else 0.0, a.LOSS.three3d_loss_weight, self.map.sigma,
                                            overlap=self.overlap_wgrad, use_graphs=self.use_graphs)
            self._inv_K = torch.pinverse(self.intrinsics[0, 0])
            self._poses_h = self.poses.detach().cpu()
            # (prev, cur) -> relative transform pinv(P_prev) P_cur (training_utils.py:191-216) on the device: the whole keyframe schedule
            # in ONE upload (poses are dataset inputs; a host-to-device copy per keyframe is a synchronisation point per keyframe)
            sched = self.keyframe_schedule()
            self._T_dev = {}
            if sched:
                Td = torch.stack([torch_poses_to_transforms(self._poses_h[:, [a, b]])[0, 1] for a, b in sched]).to(self.device)
                self._T_dev = {pair: Td[i] for i, pair in enumerate(sched)}
            if edist.world() > 1:                   # one shared depth network: start from rank 0's parameters
                edist.broadcast_parameters_(self.optimizer.flat)
                self.step_plan.net.refresh_layouts()
        self.step_plan.net.overlap = self.overlap_wgrad
        return self.step_plan

    def _load_pair(self, sp, prev, cur):
        """The keyframe pair's inputs into the plan's resident buffers (device-to-device copies + the ground-truth median)."""
        # T = pinv(P_prev) P_cur (training_utils.py:191-216) on the host: poses are dataset inputs, 4x4 algebra; uploaded once per pair
        # (a pageable host-to-device copy inside the keyframe loop is a stream synchronisation point)
        T = self._T_dev.get((prev, cur))
        if T is None:
            T = self._T_dev[(prev, cur)] = torch_poses_to_transforms(self._poses_h[:, [prev, cur]])[0, 1].to(self.device)
        sp.set_pair(self.colors[0, prev], self.colors[0, cur], self.gt_depths[0, prev], self.gt_depths[0, cur], self.intrinsics[0, 0], T, self.poses[0, cur],
                    inv_K=self._inv_K, pose_src=self.poses[0, prev])

    def refinement(self, prev, cur, max_steps=None, next_pair=None):
        """One keyframe: `OPTIMIZATION.refinement_steps` optimisation steps on the pair (prev, cur), then the map update
        (online_adaption.py:259-327).  max_steps (bench.py) truncates the optimisation loop to time an exact step count.
        next_pair: the keyframe pair that follows in the schedule -- its input copies are queued BEFORE this keyframe's map update,
        whose last action is a host read-back (the number of appended points): the GPU then finds them behind the map kernels
        instead of idling while the host issues them after the synchronisation (0.3-0.9 ms per keyframe in the kernel trace)."""
        if not self._plan_eligible():
            return self.refinement_autograd(prev, cur, max_steps)
        a = self.args
        sp = self._step_plan()
        if self._preloaded != (prev, cur):
            self._load_pair(sp, prev, cur)
        self._preloaded = None
        use_3d = a.LOSS.three3d_loss and not self.first_iter
        index = self.map.knn_index(self.H * self.W) if use_3d else None     # one grid build per keyframe, three queries
        nsteps = a.OPTIMIZATION.refinement_steps if max_steps is None else min(int(max_steps), a.OPTIMIZATION.refinement_steps)
        for refine_step in range(nsteps):
            held = self._forward_holds if (refine_step == 0 and self.reuse_forward and self._forward_holds is not None) else (None, None)
            if self.median_elements is not None:
                sp.median_elements_override = self.median_elements[self.refinement_steps_done]
            sp.step(refine_step == 0, index, source_forward_is_current=held == (None, prev), both_forwards_are_current=held == (prev, cur))
            if self.median_elements_log is not None:
                self.median_elements_log.append((sp.delta.reshape(-1) == sp.md).nonzero().reshape(-1).to(torch.int32))
            self._forward_holds = None              # Adam stepped
            self.refinement_steps_done += 1
            if a.DEBUG.print_metrics:
                lp, lr, l3 = sp.losses()
                m = ops.depth_metrics(sp.gt[1], sp.depth[1], a.DATA.name == "TUM")
                total = lp + (a.LOSS.depth_regularizer_weight * lr if sp.reg else 0.0)
                l3v = l3 if use_3d else torch.zeros((), device=self.device)
                if use_3d:
                    total = total + a.LOSS.three3d_loss_weight * l3
                rec = torch.cat([total.reshape(1), lp.reshape(1), lr.reshape(1), sp.ratio.reshape(1), m, l3v.reshape(1)]).cpu()
                self._log_step(rec, refine_step, nsteps)
        # map update (online_adaption.py:329-366): one more forward with the refined network, then PointFusion.  Everything the map step
        # reads sits in resident buffers (stashed frames and poses, scaled depths, intrinsics): one captured graph incl. the rebuild of the
        # nearest-neighbour index.  The NEXT pair's inputs are loaded before it, and -- its source frame being this pair's target, already
        # forwarded with the weights the next keyframe starts from -- its target frame goes through the network next to the map step
        sp.predict_depths()
        sp.stash_map_inputs()
        prefetch = False
        if next_pair is not None:
            self._load_pair(sp, *next_pair)
            self._preloaded = tuple(next_pair)
            # (measured, profiles/r04_prefetch_forward_ab.txt: the second stream inside the map graph makes the step SLOWER -- 194.5 against
            #  201.1 steps/s -- the forward's GEMMs and the map step's memory-bound kernels take more from each other than the overlap gives
            #  back, as round 3 found for backward-weight next to backward-data; the sequential single-frame forward stays the default)
            prefetch = self.prefetch_forward and self.reuse_forward and next_pair[0] == cur
        if a.MODEL.odom == "gt":
            sp.update_map(self.map, self.first_iter, prefetch=prefetch)
        else:
            # MODEL.odom icp / gradicp (the reference's default, configs/config.yaml:30): frame-to-model odometry from the previous keyframe's
            # pose inside the same captured map step (e2ehip.icp.ResidentOdometry: no host round trip per iteration); the map is fused with
            # the ESTIMATED pose and the pose itself, which the reference drops (online_adaption.py:362-363), is kept for the trajectory error
            odo = self._odometry()
            sp.update_map_odom(self.map, self.first_iter, odo, prefetch=prefetch)
            self.estimated_poses.append((odo.pose.clone(), self.poses[0, cur]))
        # slots of the plan's batch that hold a complete forward pass made with the current weights: (frame in slot 0, frame in slot 1)
        self._forward_holds = (cur, next_pair[1]) if prefetch else (None, cur)

    def _odometry(self):
        if getattr(self, "_odo", None) is None:
            from e2ehip.icp import ResidentOdometry
            self._odo = ResidentOdometry(self.map, dsratio=4, numiters=self.args.MODEL.numiters, mode=self.args.MODEL.odom,
                                         grid_cells=int(os.environ.get("E2E_ICP_CELLS", "256")))
        return self._odo

    def _log_step(self, rec, refine_step, nsteps):
        self.log.append(rec)
        print("Refine_Step:", refine_step, "Total_Loss:", round(rec[0].item(), 5), "abs_rel: ", round(rec[4].item(), 5),
              "rmse: ", round(rec[6].item(), 5), "a1: ", round(rec[8].item(), 5))
        if refine_step == nsteps - 1:
            self.mean_abs.append(rec[4].item())

    def refinement_autograd(self, prev, cur, max_steps=None):
        """The same keyframe through torch.autograd over the per-layer Functions (same kernels, one launch at a time): the form
        that carries the off-by-default loss terms, and the cross-check of the launch plan (tests/test_gpu_driver.py)."""
        a = self.args
        self._forward_holds = None                  # this form keeps no activations between keyframes
        colors, gt, poses = self._pair(prev, cur)
        transform = torch_poses_to_transforms(poses)
        K = self.intrinsics[:, 0]
        inv_K = torch.pinverse(K)
        T = transform[:, 1].contiguous()
        median_gt = ops.median_lower(gt)
        src, tgt = colors[:, 0].permute(0, 3, 1, 2), colors[:, 1].permute(0, 3, 1, 2)     # NHWC memory, NCHW views
        initial = None
        use_reg = a.LOSS.depth_regularizer and a.OPTIMIZATION.refinement == "PFT"
        use_3d = a.LOSS.three3d_loss and not self.first_iter
        nsteps = a.OPTIMIZATION.refinement_steps if max_steps is None else min(int(max_steps), a.OPTIMIZATION.refinement_steps)
        flagged = not self._plan_eligible()
        for refine_step in range(nsteps):
            self.optimizer.zero_grad()
            disp = self.models["depth"](colors[0], 0)[("disp", 0, 0)]                     # (2,1,H,W): pair as one batch
            depth, delta, ratio = ops.depth_from_disp_median_scaled(
                disp, median_gt, None if self.median_elements is None else self.median_elements[self.refinement_steps_done])
            if refine_step == 0 and use_reg:
                initial = delta.clone()                                                  # 1/disp BEFORE scaling (:284-285)
            d_src, d_tgt = depth[0:1], depth[1:2]
            if flagged:
                image_loss, loss2 = self.compute_flagged_losses(disp, depth, initial if use_reg else None, src, tgt, gt, K, inv_K, T)
                roots, grads = [image_loss], [None]
            else:
                self.plan.bind(d_tgt.detach(), d_src.detach(), initial[1:2] if use_reg else None, initial[0:1] if use_reg else None,
                               src, tgt, K, inv_K, T)
                if refine_step == 0:                     # the pair's geometry is host data: pass it as kernel arguments
                    self.plan.set_host_geometry(self._K_host, torch.pinverse(self._K_host), transform[0, 1].cpu())
                loss2, g_tgt, g_src = self.plan.step()
                g_depth = torch.cat([g_src if use_reg else torch.zeros_like(g_tgt), g_tgt], 0)
                roots, grads = [depth], [g_depth]
            l3 = None
            if use_3d:
                l3 = self.compute_3d_loss(d_tgt, K, poses[:, 1], T)
                roots.append(l3 * a.LOSS.three3d_loss_weight)
                grads.append(None)
            with e2e_conv.direct_weight_grads(self.optimizer, overlap=self.overlap_wgrad_autograd):     # weight gradients accumulate straight into FusedAdam's flat bucket
                torch.autograd.backward(roots, grads)
            self._exchange_gradients()
            self.optimizer.step()
            self.refinement_steps_done += 1
            if a.DEBUG.print_metrics:
                m = ops.depth_metrics(gt[0, 1], d_tgt.detach(), a.DATA.name == "TUM")
                total = image_loss.detach() if flagged else loss2[0] + (a.LOSS.depth_regularizer_weight * loss2[1] if use_reg else 0.0)
                if l3 is not None:
                    total = total + a.LOSS.three3d_loss_weight * l3.detach()
                rec = torch.cat([total.reshape(1), loss2, ratio.reshape(1), m, (l3.detach() if l3 is not None else torch.zeros((), device=self.device)).reshape(1)]).cpu()
                self._log_step(rec, refine_step, nsteps)
        self.create_refined_pointcloud(colors, gt, poses, median_gt)

    def compute_flagged_losses(self, disp, depth, initial, src, tgt, gt, K, inv_K, T):
        """novel_view_synthesis + compute_losses of the reference (online_adaption.py:412-455, :473-532) operator by operator -- each
        operator a HIP kernel behind the reference's module name -- for the flags its recommended configuration leaves off:
        LOSS.min_reprojection, auto_masking, geometric, smoothness, supervise_depth.  disp / depth (2,1,H,W): [source, target].
        Returns (everything but the 3-D term, the two-value (photometric, regulariser) log row).

        LOSS.supervise_depth: the reference reads inputs["sparse_gt_depth", ...] here (:633) without ever filling it (its
        process_inputs :388-410 has no sparse_sampling call, unlike train_depth.py:535-541) and dies with a KeyError; here the keys
        are filled the way train_depth.py does, so the flag works."""
        a = self.args
        lo = a.LOSS
        d_src, d_tgt = depth[0:1], depth[1:2]
        outputs = {}
        camera_points = self.backproject_depth(d_tgt, inv_K)
        if lo.geometric:
            grid, warped_depth, valid = self.project_3d(points=camera_points, K=K, T=T, geometric=True)
            outputs["warped_depth", -1], outputs["valid_mask", -1] = warped_depth, valid
            synth = ops.grid_sample(src, grid, padding_mode=a.MODEL.padding_mode, align_corners=True)          # sic (:431-434)
            outputs["interpolated_depth", -1] = ops.grid_sample(d_src, grid, padding_mode=a.MODEL.padding_mode, align_corners=False)
        else:
            grid, valid = self.project_3d(points=camera_points, K=K, T=T, geometric=False)
            synth = ops.grid_sample(src, grid, padding_mode=a.MODEL.padding_mode, align_corners=False)
        masked = (lambda x: ops.mask_mul(x, valid)) if lo.photometric_mask else (lambda x: x)
        masked_tgt = masked(tgt)
        photometric = photometric_loss(ssim=self.ssim, prediction=masked(synth), target=masked_tgt)        # (1,1,H,W): one source frame
        if lo.auto_masking:
            auto = photometric_loss(ssim=self.ssim, prediction=masked(src), target=masked_tgt)
            if lo.min_reprojection:
                auto = auto + torch.randn(auto.shape, device=auto.device) * 0.00001                          # "Break tie's" (:498)
            photometric = torch.cat((auto, photometric), dim=1)
        # one map: its mean; two: the mean of the per-pixel minimum, gradient to the first minimal map (e2e_min_reprojection_lossgrad)
        loss = min_reprojection_loss(photometric)
        photo_value = loss.detach()
        if lo.geometric:
            loss = loss + torch.stack([geometric_consistency_loss(outputs, -1, self.device)], dim=0).mean() * lo.geometric_weight
        if lo.smoothness:
            loss = loss + disparity_smoothness_loss(disp=ops.mean_normalize(disp[0:1]), img=tgt) * lo.smoothness_weight
        reg_value = torch.zeros((), device=self.device)
        if initial is not None:
            reg = depth_reguralizer(initial_depth=initial[0:1], refined_depth=d_src, loss_func=lo.depth_regularizer_type) \
                + depth_reguralizer(initial_depth=initial[1:2], refined_depth=d_tgt, loss_func=lo.depth_regularizer_type)
            loss = loss + reg * lo.depth_regularizer_weight
            reg_value = reg.detach()
        if lo.supervise_depth:
            gt_loss = 0
            for index, d in enumerate((d_src, d_tgt)):
                sparse, mask = sparse_sampling(lo.sampling_type, lo.sampling_prob, gt[:, index].permute(0, 3, 1, 2))
                gt_loss = gt_loss + depth_gt_loss(prediction=d, sparse_groundtruth=sparse, sparse_mask=mask)
            loss = loss + gt_loss * lo.gt_depth_weight
        return loss, torch.stack([photo_value, reg_value])

    def compute_3d_loss(self, d_tgt, K, pose_tgt, T):
        """End-2-end point supervision (online_adaption.py:457-471 + :638-645): the target frame's local cloud (in world
        coordinates through its pose) is transformed AGAIN by T (reference quirk) and pulled
        towards its nearest neighbours in the detached global map."""
        maps = ops.vertex_normal_maps(d_tgt.reshape(1, self.H, self.W), K, pose_tgt, self.map.sigma)
        cloud = ops.select_rows(maps["Vg"][0].reshape(-1, 3), maps["valid"][0].reshape(-1))
        moved = ops.transform_points(cloud, T[0])
        d, _ = ops.knn1(moved, self.map.knn_index(self.H * self.W))    # one grid build per keyframe, three queries
        return d.mean()

    def _exchange_gradients(self, participating=True):
        """between loss.backward() and optimizer.step() (online_adaption.py:539-540): ONE all-reduce of the flat bucket."""
        if edist.world() > 1:
            if self.optimizer.flat is None:
                self.optimizer._build()
            edist.exchange_gradients_(self.optimizer.flat, participating)

    @torch.no_grad()
    def create_refined_pointcloud(self, colors, gt, poses, median_gt):
        """online_adaption.py:329-366 (autograd form: one more network forward, median scaling, then the map update)."""
        disp = self.models["depth"](colors[0], 0)[("disp", 0, 0)]
        depth, _, _ = ops.depth_from_disp_median_scaled(disp, median_gt)
        return self._update_map(colors[0, 0], colors[0, 1], depth, poses[0, 0], poses[0, 1])

    @torch.no_grad()
    def _update_map(self, rgb_prev, rgb_cur, depth, pose_prev, pose_cur):
        """PointFusion map step(s) with the refined, median-scaled depths (2,1,H,W) of the pair (online_adaption.py:347-363)."""
        K = self.intrinsics[0, 0]
        if self.args.MODEL.odom == "gt":
            # resident form: map size, association tables and the appended rows never leave the device (no host read per keyframe)
            if self.first_iter:
                self.map.step_resident(rgb_prev, depth[0, 0], K, pose_prev)
            self.map.step_resident(rgb_cur, depth[1, 0], K, pose_cur)
            return self.map
        if self.first_iter:
            self.map.step(rgb_prev, depth[0, 0], K, pose_prev)
        live_pose = pose_cur
        if self.args.MODEL.odom != "gt":
            # the reference passes prev_frame here (online_adaption.py:362-363): frame-to-model odometry from the
            # previous keyframe's pose; the map is fused with the ESTIMATED pose and the pose itself is dropped there --
            # we keep it to report the trajectory error
            from e2ehip import icp
            live_pose, _ = icp.frame_to_model(self.map, depth[1, 0], K, pose_prev, numiters=self.args.MODEL.numiters, mode=self.args.MODEL.odom)
            self.estimated_poses.append((live_pose, pose_cur))
        self.map.step(rgb_cur, depth[1, 0], K, live_pose)
        return self.map

    def absolute_trajectory_error(self):
        """RMSE of the camera-centre error of the odometry estimates against the dataset poses (metres)."""
        if not self.estimated_poses:
            return 0.0
        e = torch.stack([(p[:3, 3] - g[:3, 3]).norm() for p, g in self.estimated_poses])
        return float(torch.sqrt((e ** 2).mean()))


def default_config(height=480, width=640, sequence_length=60):
    """The reference's config tree (configs/config.yaml) with the README's online-adaption settings
    (README.md:146-158: 3 refinement steps, photometric + 3-D + depth regulariser) and odom: gt."""
    from utils.yaml_configs import _wrap
    return _wrap(OrderedDict(
        SETTINGS=dict(name="run", num_workers=0, device="cuda"),
        DATA=dict(name="synthetic", height=height, width=width, frames=[0, -1], scales=[0], seed=1234),
        MODEL=dict(depth_network="indoor", num_layers=18, weights_init_encoder=False, use_pretrained_models=False, load_depth_path="",
                   slam="PointFusion", odom="gt", dist_th=0.05, angle_th=20, sigma=0.6, numiters=20, padding_mode="border",
                   refinement_mode=True),
        LOSS=dict(chamfer_distance=False, knn_points=False, auto_masking=False, min_reprojection=False, photometric_mask=True,
                  geometric=False, geometric_weight=0.5, smoothness=False, smoothness_weight=1e-3, depth_regularizer=True,
                  depth_regularizer_weight=1e-2, depth_regularizer_type="l2", supervise_depth=False, gt_depth_weight=1, sampling_type="random",
                  sampling_prob=0.05, three3d_loss=True, three3d_loss_weight=1.0),
        OPTIMIZATION=dict(batch_size=1, refinement="PFT", refinement_steps=3, learning_rate=1e-5, optimizer="Adam", schedular="StepLR",
                          schedular_step_size=100, schedular_gamma=0.5),
        DEBUG=dict(print_metrics=True),
        DEMO=dict(sequence_length=sequence_length, frame_threshold=0.05),
    ))


if __name__ == "__main__":
    from utils.arguments import arguments
    cli = arguments()
    cfg = load_yaml(cli["config_path"])
    cfg.SETTINGS.name = cli["name"]
    SLAM(cfg).main()
