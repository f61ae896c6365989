"""kornia.geometry.linalg.inverse_transformation -- imported by the reference's driver (online_adaption.py:15) and
never called on the refinement path.  Host-side 4x4 algebra (SURVEY.md 2.3 keeps such math in torch)."""
import torch


def inverse_transformation(trans_12):
    """(..., 4, 4) rigid transform -> its inverse [R^T | -R^T t]."""
    if not torch.is_tensor(trans_12):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(trans_12)}")
    if trans_12.dim() not in (2, 3) or tuple(trans_12.shape[-2:]) != (4, 4):
        raise ValueError(f"Input size must be a Nx4x4 or 4x4. Got {tuple(trans_12.shape)}")
    R, t = trans_12[..., :3, :3], trans_12[..., :3, 3:4]
    Rt = R.transpose(-1, -2)
    out = torch.zeros_like(trans_12)
    out[..., :3, :3] = Rt
    out[..., :3, 3:4] = -Rt @ t
    out[..., 3, 3] = 1.0
    return out
