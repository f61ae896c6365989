"""Host-side helpers shared by the ICL / TUM loaders (SURVEY.md 8f row N2, Appendix A "datasets").
Decoding uses Pillow + numpy; no OpenCV / imageio dependency."""
import numpy as np
import torch
from PIL import Image


def read_color(path, height, width):
    """RGB uint8 -> float32 (H,W,3) in 0..255, bilinear resize when needed."""
    im = Image.open(path).convert("RGB")
    if im.size != (width, height):
        im = im.resize((width, height), Image.BILINEAR)
    return np.asarray(im, dtype=np.float32)


def read_depth(path, height, width, scale):
    """16-bit depth PNG -> float32 (H,W,1) metres = raw / scale, nearest-neighbour resize."""
    im = Image.open(path)
    if im.size != (width, height):
        im = im.resize((width, height), Image.NEAREST)
    d = np.asarray(im).astype(np.float32) / float(scale)
    return d[..., None]


def scale_intrinsics(K, h_ratio, w_ratio):
    """Resize a pinhole matrix: row 0 (fx, cx) by the width ratio, row 1 (fy, cy) by the height ratio."""
    K = K.clone()
    K[0, 0] *= w_ratio
    K[0, 2] *= w_ratio
    K[1, 1] *= h_ratio
    K[1, 2] *= h_ratio
    return K


def relative_poses(poses):
    """(L,4,4) absolute camera-to-world poses -> poses relative to the first frame of the sequence."""
    inv0 = torch.linalg.inv(poses[0].double())
    return torch.stack([(inv0 @ p.double()).float() for p in poses])


def poses_to_transforms(poses):
    """T_0 = I, T_i = inv(P_{i-1}) P_i  (frame-to-frame)."""
    out = [torch.eye(4)]
    for i in range(1, poses.shape[0]):
        out.append((torch.linalg.inv(poses[i - 1].double()) @ poses[i].double()).float())
    return torch.stack(out)


def sequence_starts(num_frames, seqlen, dilation, stride, start, end=None):
    """First-frame indices of the extracted sequences: frame i of a sequence is first + i*(dilation+1); consecutive
    sequences start `stride` frames apart (default: back to back)."""
    dilation = 0 if dilation is None else int(dilation)
    span = (seqlen - 1) * (dilation + 1) + 1
    stride = seqlen * (dilation + 1) if stride is None else int(stride)
    start = 0 if start is None else int(start)
    end = num_frames if end is None else min(int(end), num_frames)
    if stride <= 0:
        raise ValueError(f"stride must be positive. Got {stride}.")
    if start < 0 or start >= num_frames:
        raise ValueError(f"start must be in [0, {num_frames}). Got {start}.")
    return [s for s in range(start, end - span + 1, stride)], dilation + 1


def quaternion_pose(tx, ty, tz, qx, qy, qz, qw):
    """TUM ground truth line -> 4x4 camera-to-world matrix."""
    q = np.array([qx, qy, qz, qw], dtype=np.float64)
    n = np.dot(q, q)
    T = np.eye(4)
    if n > 1e-12:
        q = q * np.sqrt(2.0 / n)
        o = np.outer(q, q)
        T[:3, :3] = np.array([[1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]],
                              [o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]],
                              [o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1]]])
    T[:3, 3] = (tx, ty, tz)
    return T
