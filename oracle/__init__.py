"""oracle/ -- CPU restatement of the reference's online-refinement hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
anything from here, and only as the checker / the timed CPU baseline.  The product
path (``end-to-end-self-supervised-slam_amd/``) never imports it and has no CPU fallback.

Parity status (see DESIGN.md "Oracle"):
  * warp_loss.py, poses.py, depthnet.py (decoder + wiring)  -- PINNED by golden vectors that
    tests/golden/make_golden.py captured from the reference's own files
    (depth_estimation/view_synthesis.py, loss/losses.py, utils/training_utils.py,
    depth_estimation/networks.py) loaded by path in the build container.
  * warp_loss.py::min_reprojection / process_disparity restate train_depth.py:224-237,657-661 from
    the text of that file (its import block needs gradslam / cv2 / tensorboardX) -- PARITY UNPINNED.
  * pointfusion.py, knn.py (gradslam / chamferdist semantics) and the ResNet-18 body of
    depthnet.py (torchvision) -- PARITY UNPINNED: those packages are absent from
    /root/reference and are not installed (README.md:5-33 gives no version pins); the
    restatement follows SURVEY.md Appendix A and the reference's call sites only.
"""
