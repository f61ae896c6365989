"""CPU restatement of the 4x4 pose helpers on the path.  TEST INFRASTRUCTURE ONLY.
Parity: PINNED by tests/golden/g6 (captured from the reference's utils/training_utils.py) and by the one known answer the reference holds
for the loader's frame-to-frame `transforms` (pose_checker.py:57-82: T_12 = [R1^T R2 | R1^T (t2 - t1)] "should match Transform 2"):
tests/test_host_utils.py::test_transforms_match_the_pose_checker_closed_form checks this function, the product's
utils.training_utils.torch_poses_to_transforms and the product's ICL loader against that closed form."""
import torch


def poses_to_transforms(poses):
    """(B,L,4,4) absolute poses -> frame-to-frame transforms, T_0 = I, T_s = pinv(P_{s-1}) P_s.
    reference: utils/training_utils.py:191-216."""
    out = poses.detach().clone()
    for b in range(poses.shape[0]):
        for s in range(poses.shape[1]):
            out[b, s] = torch.eye(4) if s == 0 else torch.pinverse(poses[b, s - 1]).matmul(poses[b, s])
    return out


def inverse_T(T):
    """reference: utils/training_utils.py:130-140 (torch.pinverse)."""
    return torch.pinverse(T)


def frame_distance(prev, cur):
    """Camera-centre distance ||-R1^T t1 + R2^T t2||.  reference: online_adaption.py:186-205."""
    pr, pt = prev[0, :3, :3], prev[0, :3, -1]
    cr, ct = cur[0, :3, :3], cur[0, :3, -1]
    pc = -1 * torch.matmul(pr.transpose(0, 1), pt)
    cc = -1 * torch.matmul(cr.transpose(0, 1), ct)
    return torch.linalg.norm(pc - cc)


def sparse_sampling(prob, depth):
    """reference: utils/training_utils.py:176-189 (consumes torch's global RNG)."""
    mask = torch.rand_like(depth)
    mask[mask >= prob] = 0.0
    mask[mask > 0.0] = 1.0
    mask[depth == 0.0] = 0.0
    return depth * mask, mask


def disp_to_depth(disp, min_depth, max_depth):
    """reference: utils/training_utils.py:106-118."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    return 1 / (min_disp + (max_disp - min_disp) * disp)
