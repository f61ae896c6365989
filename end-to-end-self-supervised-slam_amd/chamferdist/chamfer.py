from collections import namedtuple

import torch
import torch.nn as nn

from e2ehip import ops

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False, return_sorted=True):
    """K=1 nearest neighbours of p1 (N,P1,3) in p2 (N,P2,3): dists (N,P1,1) SQUARED, idx (N,P1,1) int64.
    Only what loss/losses.py:57 needs: K = 1, D = 3, full-length clouds."""
    if K != 1:
        raise NotImplementedError("only K=1 is on the reference's path (loss/losses.py:57)")
    if lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("ragged batches are not on the reference's path")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension")
    if p1.shape[2] != 3:
        raise NotImplementedError("only 3-D points are on the reference's path")
    ds, ids = [], []
    for b in range(p1.shape[0]):
        d, i = ops.knn1(p1[b], p2[b])
        ds.append(d)
        ids.append(i)
    dists, idx = torch.stack(ds).unsqueeze(-1), torch.stack(ids).unsqueeze(-1)
    nn_pts = None
    if return_nn:
        nn_pts = torch.stack([p2[b][idx[b, :, 0]] for b in range(p1.shape[0])]).unsqueeze(2)
    return _KNN(dists=dists, idx=idx, knn=nn_pts)


class ChamferDistance(nn.Module):
    """forward(source, target, bidirectional=False, reverse=False, reduction="mean") as called at train_depth.py:690-692."""

    def forward(self, source_cloud, target_cloud, bidirectional=False, reverse=False, reduction="mean"):
        if reduction not in ("mean", "sum"):
            raise ValueError('reduction must be "mean" or "sum"')
        red = torch.mean if reduction == "mean" else torch.sum

        def one_way(a, b):
            return red(knn_points(a, b).dists[..., 0], dim=1)          # gradients to BOTH clouds, as chamferdist's knn_points

        fwd = one_way(source_cloud, target_cloud)
        bwd = one_way(target_cloud, source_cloud) if (bidirectional or reverse) else None
        if bidirectional:
            out = fwd + bwd
        elif reverse:
            out = bwd
        else:
            out = fwd
        return red(out) if reduction == "mean" else out.sum()
