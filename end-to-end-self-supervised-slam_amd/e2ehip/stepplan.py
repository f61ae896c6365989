"""One refinement step on a keyframe pair as TWO captured hipGraphs over resident buffers (online_adaption.py:274-318 of the
reference: two depth-net forwards, median scaling, [3-D point loss], warp + photometric + regulariser, backward, Adam):

  graph A   depth network forward on the pair (e2ehip.netplan) + 1/disp + median ratio                 (e2e_depth_scale_fwd)
  graph B   3-D point loss against the global map, forward AND backward wrt the target depth: unproject -> rigid transform ->
            exact nearest neighbour -> masked mean -> adjoints (7 launches; the map size is DEVICE data and the index one
            resident buffer, so the launch arguments never change: e2ehip.fusionmap), fused warp + photometric + regulariser
            loss-and-gradient (e2e_warp_photo_lossgrad, device geometry), + the 3-D
            gradient, median-chain backward, depth network backward (weight gradients straight into the flat bucket),
            [Adam + refresh of the GEMM weight layouts]
  (N > 1 GPUs: the all-reduce of the bucket runs eagerly between graph B and a third graph with Adam + the refresh)

Replaying a graph costs the host ~15 us; the ~190 launches of a step cost it several milliseconds when issued one by one
through torch.autograd, which is what bounded round 1's step.  The first call of every graph variant runs eagerly (it IS a
real step) and is captured right after for the following calls; inside a KernelTimer pass everything runs eagerly so that
bench.py can time the individual launches.  No host synchronisation anywhere in a step."""
import torch

from . import _lib as L
from . import dist as edist
from .fused import LossGradPlan
from .netplan import NetPlan
from .ops import fusion_alpha_den

_f32 = torch.float32


class RefineStepPlan:
    def __init__(self, model, optimizer, H, W, device, padding_mode="border", use_mask=True, reg_kind="l2", w_reg=1e-2, w_3d=1.0, sigma=0.6,
                 overlap=True, use_graphs=True):
        self.dev, self.H, self.W, self.N = torch.device(device), H, W, H * W
        self.opt, self.use_graphs, self.w_3d = optimizer, use_graphs, float(w_3d)
        self.net = NetPlan(model, 2, H, W, self.dev, overlap=overlap)
        optimizer.prebuild(model.used_parameters())                 # parameters move into the flat bucket; gradient sinks exist from here on
        if {id(p) for p in self.net.parameters()} != {id(p) for p in model.used_parameters() if p.requires_grad}:
            raise RuntimeError("launch plan and model disagree about the trainable parameters")
        f = dict(device=self.dev, dtype=_f32)
        lib = L.load()
        # ---- resident tensors of a step (pair order: index 0 = previous keyframe / source, 1 = new keyframe / target) ----
        self.colors = self.net.x.t                                  # (2,H,W,3) NHWC, the network's input buffer
        self.gt = torch.zeros(2, H, W, 1, **f)
        self.K, self.inv_K, self.T, self.pose_tgt, self.pose_src = (torch.eye(4, **f).reshape(1, 4, 4).clone() for _ in range(5))
        self.median_gt = torch.zeros(1, **f)
        self.ws_med = torch.empty(lib.e2e_median_workspace_bytes(), device=self.dev, dtype=torch.uint8)
        self.delta, self.depth, self.init = (torch.empty(2, 1, H, W, **f) for _ in range(3))
        self.md, self.ratio = torch.empty(1, **f), torch.empty(1, **f)
        self.ws_scale = torch.empty(lib.e2e_depth_scale_workspace_bytes(), device=self.dev, dtype=torch.uint8)
        self.g_depth = torch.zeros(2, 1, H, W, **f)                 # [d loss / d depth_src, d loss / d depth_tgt]
        self.reg = reg_kind
        self._split = None                                          # bucket offset of the late layers' parameters (data-parallel exchange)
        self.loss = LossGradPlan(1, H, W, self.dev, padding_mode, use_mask, reg_kind, 1.0, float(w_reg) if reg_kind else 0.0)
        self.loss.g_depth_src, self.loss.g_depth_tgt = self.g_depth[0:1], self.g_depth[1:2]
        src, tgt = self.colors[0:1].permute(0, 3, 1, 2), self.colors[1:2].permute(0, 3, 1, 2)       # NHWC memory, NCHW views
        self.loss.bind(self.depth[1:2], self.depth[0:1], self.init[1:2] if reg_kind else None, self.init[0:1] if reg_kind else None,
                       src, tgt, self.K, self.inv_K, self.T)
        # ---- 3-D point loss -----------------------------------------------------------------------------------------------
        self.alpha_den = fusion_alpha_den(sigma)
        self.V, self.Nm, self.Vg, self.Ng = (torch.empty(1, H, W, 3, **f) for _ in range(4))
        self.alpha = torch.empty(1, H, W, **f)
        self.moved, self.g_moved, self.g_cloud = (torch.empty(self.N, 3, **f) for _ in range(3))
        self.nn_d = torch.empty(self.N, **f)
        self.nn_idx = torch.full((self.N,), -1, device=self.dev, dtype=torch.int64)     # -1 = no candidate (the first query of a run)
        self.l3 = torch.zeros(3, **f)                               # {mean nearest-neighbour distance, #valid rows, weight / #valid}
        self.g_nn = torch.empty(self.N, **f)
        self.g3 = torch.zeros(1, 1, H, W, **f)
        self.ws_aux = torch.empty(lib.e2e_aux_workspace_floats(), **f)
        self._graphs, self._gstream = {}, None
        # what the map step reads of the pair it belongs to (frames, both poses), as private copies (stash_map_inputs): the NEXT pair can then
        # be loaded into the network's input buffers before the map step runs, and its target frame forwarded NEXT TO the map step
        self.map_rgb = torch.empty(2, H, W, 3, **f)
        self.map_pose_src, self.map_pose_tgt = (torch.eye(4, **f).reshape(1, 4, 4).clone() for _ in range(2))
        self._pstream = None
        # tests: device int32 indices (into the stacked (2,1,H,W) predictions) of the elements the CPU evaluation's torch.median holds
        # as the median -- the ratio's gradient then lands on exactly those (e2e_depth_scale_bwd_at) instead of on this evaluation's own
        self.median_elements_override = None
        self.net.refresh_layouts()

    # ---- per keyframe -----------------------------------------------------------------------------------------------------
    def set_pair(self, colors_prev, colors_cur, gt_prev, gt_cur, K, T_host, pose_tgt, inv_K=None, pose_src=None):
        """Load a keyframe pair into the resident buffers: frames (H,W,3) in [0,1], ground-truth depths (H,W,1), intrinsics K
        (4,4), relative transform T (4,4) = pinv(P_prev) P_cur and the target pose (4,4).  All copies, no kernels of ours but the
        median of the ground-truth depths (online_adaption.py:295: torch.median(gt_depths))."""
        self.colors[0].copy_(colors_prev)
        self.colors[1].copy_(colors_cur)
        self.gt[0].copy_(gt_prev)
        self.gt[1].copy_(gt_cur)
        self.K[0].copy_(K)
        self.inv_K[0].copy_(torch.pinverse(K) if inv_K is None else inv_K)      # callers with constant intrinsics pass the inverse
        self.T[0].copy_(T_host, non_blocking=True)
        self.pose_tgt[0].copy_(pose_tgt)
        if pose_src is not None:
            self.pose_src[0].copy_(pose_src)
        L.call("e2e_median_lower", L.ptr(self.gt), self.gt.numel(), L.ptr(self.median_gt), L.ptr(self.ws_med), L.stream())

    # ---- graph plumbing ---------------------------------------------------------------------------------------------------
    def _run(self, key, fn):
        if getattr(self, "_closed", False):
            raise RuntimeError("RefineStepPlan used after close()")
        if not self.use_graphs or L.PROFILE_HOOK[0] is not None:
            return fn()
        g = self._graphs.get(key)
        if g is not None:
            return g.replay()
        fn()                                                        # first call: a real, eager execution ...
        cur = torch.cuda.current_stream(self.dev)
        if self._gstream is None:
            self._gstream = torch.cuda.Stream(self.dev)
        self._gstream.wait_stream(cur)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self._gstream):
            # ... then the same launches recorded (not executed) for the next ones.  thread_local: in data-parallel runs a collective may
            # be in flight on the communication stream and its watchdog thread queries events -- legal outside the capturing thread
            with torch.cuda.graph(g, stream=self._gstream, capture_error_mode="thread_local"):
                fn()
        cur.wait_stream(self._gstream)
        self._graphs[key] = g

    # ---- the pieces -------------------------------------------------------------------------------------------------------
    def _forward(self):
        st = L.stream()
        disp = self.net.forward()
        L.call("e2e_depth_scale_fwd", L.ptr(disp), L.ptr(self.median_gt), L.ptr(self.delta), L.ptr(self.depth), L.ptr(self.md), L.ptr(self.ratio),
               L.ptr(self.ws_scale), disp.numel(), st)

    def stash_map_inputs(self):
        """Private copies of the loaded pair's frames and poses for the map step that follows (three device-to-device copies)."""
        self.map_rgb.copy_(self.colors)
        self.map_pose_src.copy_(self.pose_src)
        self.map_pose_tgt.copy_(self.pose_tgt)

    def _scale_only(self):
        """First 'forward' of a keyframe whose two frames already went through the network with the current weights (the source frame in
        the previous keyframe's map-update forward, the target frame next to that keyframe's map step: update_map(prefetch=True))."""
        disp = self.net.disp.t.view(2, 1, self.H, self.W)
        L.call("e2e_depth_scale_fwd", L.ptr(disp), L.ptr(self.median_gt), L.ptr(self.delta), L.ptr(self.depth), L.ptr(self.md), L.ptr(self.ratio),
               L.ptr(self.ws_scale), disp.numel(), L.stream())

    def _prefetch_fork(self):
        """Inside the map-step graph: the activations of the old target frame move to slot 0 and the NEW target frame (already in the
        input buffer) goes through the network on a second stream, concurrently with the map step -- a chain of small latency-bound
        kernels (association over the map, compaction scans, the index rebuild's atomics) that leaves most of the GPU idle."""
        cur = torch.cuda.current_stream(self.dev)
        if self._pstream is None:
            self._pstream = torch.cuda.Stream(self.dev)
        self._pstream.wait_stream(cur)
        with torch.cuda.stream(self._pstream):
            self.net.move_slot(1, 0)
            self.net.forward_one(1)

    def _prefetch_join(self):
        torch.cuda.current_stream(self.dev).wait_stream(self._pstream)

    def _forward_new_target(self):
        """First forward of a keyframe whose SOURCE frame was the previous keyframe's target: the map-update forward of that keyframe
        (predict_depths, online_adaption.py:329-345) ran this very frame through these very weights -- no optimiser step lies between it and
        the first forward of the next keyframe (:281) -- so its activations are moved from batch slot 1 to slot 0 and only the new
        target frame goes through the network (1/8 of all forward work of a keyframe; the reference computes both frames again)."""
        st = L.stream()
        self.net.move_slot(1, 0)
        self.net.forward_one(1)
        disp = self.net.disp.t.view(2, 1, self.H, self.W)
        L.call("e2e_depth_scale_fwd", L.ptr(disp), L.ptr(self.median_gt), L.ptr(self.delta), L.ptr(self.depth), L.ptr(self.md), L.ptr(self.ratio),
               L.ptr(self.ws_scale), disp.numel(), st)

    def _loss3d(self, index, warm=False):
        """online_adaption.py:457-471 + :638-645: the target frame's cloud in world coordinates (its pose), transformed AGAIN by T
        (reference quirk), pulled to its nearest neighbours in the detached global map."""
        st, N = L.stream(), self.N
        d = self.depth[1:2]
        L.call("e2e_vertex_normal_maps", L.ptr(d), L.ptr(self.K), L.ptr(self.pose_tgt), float(self.alpha_den), L.ptr(self.V), L.ptr(self.Nm), L.ptr(self.Vg),
               L.ptr(self.Ng), L.ptr(self.alpha), 1, self.H, self.W, st)
        L.call("e2e_transform_points", L.ptr(self.Vg), L.ptr(self.T), L.ptr(self.moved), N, 0, st)
        # the previous query's neighbours bound the search from the start: steps 2 and 3 of a keyframe ask about the same pixels against the
        # same map; the FIRST step of a keyframe inherits the last answers of the keyframe before (a neighbouring view: the same pixel looks at
        # a point a few centimetres away; map rows are never removed, so the index still names a real point -- all the bound needs)
        index.query(self.moved, N, self.nn_d, self.nn_idx, st, row_len=self.W, warm=self.nn_idx if warm else None)
        L.call("e2e_masked_mean_lossgrad", L.ptr(self.nn_d), L.ptr(d), N, self.w_3d, L.ptr(self.l3), L.ptr(self.g_nn), L.ptr(self.ws_aux), st)
        L.call("e2e_knn1_bwd", L.ptr(self.g_nn), L.ptr(self.moved), L.ptr(index.ref), L.ptr(self.nn_idx), N, L.ptr(self.g_moved), st)
        L.call("e2e_transform_points", L.ptr(self.g_moved), L.ptr(self.T), L.ptr(self.g_cloud), N, 1, st)
        L.call("e2e_vertex_maps_bwd", L.ptr(d), L.ptr(self.K), L.ptr(self.pose_tgt), None, L.ptr(self.g_cloud), L.ptr(self.g3), 1, self.H, self.W, st)

    def _backward(self, use_3d, with_adam, late_only=False, index=None, warm=False):
        st = L.stream()
        if index is not None:                                       # resident index: constant launch arguments, part of the captured graph
            self._loss3d(index, warm)
        self.loss.step()                                            # losses -> self.loss.loss[0..1]; d/d depth -> self.g_depth
        if use_3d:                                                  # g_depth_tgt += d(w_3d * l3)/d depth_tgt
            L.call("e2e_conv2d_act_bwd_acc", L.ptr(self.g3), L.ptr(self.g3), None, L.ptr(self.g_depth[1:2]), self.N, 1, 0, 1, st)
        ov = self.median_elements_override
        L.call("e2e_depth_scale_bwd_at", L.ptr(self.g_depth), L.ptr(self.delta), L.ptr(self.median_gt), L.ptr(self.md), L.ptr(ov),
               0 if ov is None else int(ov.numel()), L.ptr(self.net.disp.g), L.ptr(self.ws_scale), self.g_depth.numel(), st)
        if late_only:                                               # data-parallel runs: head, decoder, layer4 -- the bucket's tail
            self.net.backward_late_layers()
            return
        self.net.backward()
        if with_adam:
            self._adam()

    def _adam(self):
        self.opt.step()
        self.net.refresh_layouts()

    # ---- public -----------------------------------------------------------------------------------------------------------
    def step(self, first_step, knn_index=None, source_forward_is_current=False, both_forwards_are_current=False):
        """One refinement step on the loaded pair.  first_step: stash 1/disp BEFORE scaling as the regulariser's reference
        (online_adaption.py:284-285).  knn_index: e2ehip.ops.KnnIndex over the global map, or None on the first keyframe.
        source_forward_is_current: batch slot 1 holds a forward pass of the frame that is now the SOURCE, made with the current weights
        (the caller's promise: SLAM.refinement keeps track) -- see _forward_new_target.  both_forwards_are_current: both slots do
        (update_map(prefetch=True) ran the new target next to the map step) -- see _scale_only."""
        use_3d = knn_index is not None
        if both_forwards_are_current:
            self._run("scale_only", self._scale_only)
        elif source_forward_is_current:
            self._run("fwd_new_target", self._forward_new_target)
        else:
            self._run("fwd", self._forward)
        if first_step and self.reg:
            self.init.copy_(self.delta)
        # the 3-D loss against a RESIDENT index (e2ehip.fusionmap: one buffer per run, map size on the device) has constant launch
        # arguments and rides in the backward graph; any other index object is queried eagerly here
        cap_idx, ikey = None, None
        warm = use_3d                                               # every query starts from the previous one's answers (see _loss3d)
        if use_3d and getattr(knn_index, "resident", False):
            cap_idx, ikey = knn_index, (knn_index.ws.data_ptr(), warm)
        elif use_3d:
            self._loss3d(knn_index)
        if not edist.data_parallel():
            okey = None if self.median_elements_override is None else (self.median_elements_override.data_ptr(), self.median_elements_override.numel())
            self._run(("bwd", use_3d, True, ikey, okey), lambda: self._backward(use_3d, True, index=cap_idx, warm=warm))
        else:
            # data-parallel: the exchange of the bucket's tail (head, decoder, layer4: 80 % of the bytes, complete after the first
            # part of the backward pass) travels while the early layers' backward computes; then the remaining 20 %, then Adam
            if self._split is None:
                self._split = self.net.split_offset(self.opt.flat)
            self._run(("bwd_late", use_3d, ikey), lambda: self._backward(use_3d, False, late_only=True, index=cap_idx, warm=warm))
            handle = edist.exchange_gradients_late_(self.opt.flat, self._split, True)
            self._run("bwd_early", self.net.backward_early_layers)
            edist.exchange_gradients_early_(self.opt.flat, self._split, handle, True)
            self._run("adam", self._adam)
        self._parameters_stepped()

    def _parameters_stepped(self):
        """A (possibly replayed) Adam launch rewrote the parameters behind torch's version counters: module-path layout caches of
        these parameters are stale (FlatParams.touched), the plan's own layouts were refreshed inside the same graph."""
        self.opt.flat.touched()
        self.net.mark_layouts_current()

    def idle_step(self):
        """A step of a rank without a keyframe in this round (data-parallel runs): zero bucket in, averaged update out."""
        if self._split is None:
            self._split = self.net.split_offset(self.opt.flat)
        handle = edist.exchange_gradients_late_(self.opt.flat, self._split, False)      # the same two collectives as a participating rank
        edist.exchange_gradients_early_(self.opt.flat, self._split, handle, False)
        self._run("adam", self._adam)
        self._parameters_stepped()

    def predict_depths(self):
        """Median-scaled depths of the loaded pair with the current network (the map update's forward pass,
        online_adaption.py:329-344) -> (2,1,H,W) resident buffer."""
        self._run("fwd", self._forward)
        return self.depth

    def update_map(self, fmap, first, prefetch=False):
        """The keyframe's PointFusion map step(s) (online_adaption.py:347-363 with the ground-truth poses) and the rebuild of the
        nearest-neighbour index over the grown map, from RESIDENT buffers -- the stashed frames and poses of the pair (stash_map_inputs),
        the median-scaled depths of the last forward (predict_depths), the intrinsics -- so that every launch argument is constant and the
        ~25 launches replay as one captured graph (issued eagerly they left the GPU idle ~8 us per launch: the host cannot run ahead of a
        replaying graph).  first: the map is empty, the previous keyframe's frame is fused before the new one's.
        prefetch: the NEXT pair is already loaded (its source frame = this pair's target): forward its target frame concurrently."""
        def fn():
            if prefetch:
                self._prefetch_fork()
            if first:
                fmap.step_resident(self.map_rgb[0], self.depth[0, 0], self.K[0], self.map_pose_src[0])
            fmap.step_resident(self.map_rgb[1], self.depth[1, 0], self.K[0], self.map_pose_tgt[0])
            fmap.knn_index(self.N)                                  # rebuilt in place from the device-resident point count
            if prefetch:
                self._prefetch_join()
        self._run(("map", bool(first), id(fmap), bool(prefetch)), fn)
        fmap.mark_updated_on_device(index_current=True)

    def update_map_odom(self, fmap, first, odometry, prefetch=False):
        """update_map with the reference's default map step (configs/config.yaml:30 odom: gradicp; online_adaption.py:362 passes
        prev_frame): the new keyframe's pose comes from frame-to-model odometry against the map, started at the previous keyframe's
        pose, and the frame is fused with the ESTIMATED pose.  odometry: e2ehip.icp.ResidentOdometry over `fmap` -- source / target
        selection, index build and its numiters iterations incl. the 6x6 solves are launches over resident buffers, so the whole
        map step (odometry, fusion, rebuild of the nearest-neighbour index) replays as ONE captured graph.  The estimate is left in
        odometry.pose (device)."""
        def fn():
            if prefetch:
                self._prefetch_fork()
            if first:
                fmap.step_resident(self.map_rgb[0], self.depth[0, 0], self.K[0], self.map_pose_src[0])
            pose = odometry.run(self.depth[1, 0], self.K[0], self.map_pose_src[0])
            fmap.step_resident(self.map_rgb[1], self.depth[1, 0], self.K[0], pose)
            fmap.knn_index(self.N)
            if prefetch:
                self._prefetch_join()
        self._run(("map_odom", bool(first), id(fmap), id(odometry), bool(prefetch)), fn)
        fmap.mark_updated_on_device(index_current=True)

    def close(self):
        """Deterministic end of the plan (tests build many; a product process builds one): wait for everything it launched -- current
        stream, capture stream, backward-weight side stream --, destroy the captured graphs and with them their private memory pools,
        drop the network plan's buffers.  Using the plan afterwards raises."""
        torch.cuda.current_stream(self.dev).synchronize()
        if self._gstream is not None:
            self._gstream.synchronize()
        if self._pstream is not None:
            self._pstream.synchronize()
        self._graphs.clear()
        self.net.close()
        self._closed = True

    def losses(self):
        """(photometric mean, regulariser sum of means, 3-D loss mean) of the last step as device tensors (no sync)."""
        return self.loss.loss[0], self.loss.loss[1], self.l3[0]
