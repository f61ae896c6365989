"""Drop-in for utils/training_utils.py (reference: utils/training_utils.py:11-216): optimiser / scheduler
factories and the 4x4 pose helpers.  The 4x4 algebra stays in torch on whatever device the poses live on
(host-side 4x4 math); the Adam option returns the fused multi-tensor HIP optimiser when the
parameters live on the GPU."""
import torch


def define_optim(args, parameters):
    name, lr = args.OPTIMIZATION.optimizer, args.OPTIMIZATION.learning_rate
    parameters = list(parameters)
    if name == "Adam":
        from e2ehip.optim import FusedAdam
        opt = FusedAdam(parameters, lr=lr) if parameters and parameters[0].is_cuda else torch.optim.Adam(parameters, lr=lr)
    elif name == "SparseAdam":
        opt = torch.optim.SparseAdam(parameters, lr=lr)
    elif name == "SGD":
        opt = torch.optim.SGD(parameters, lr=lr, momentum=0.9, weight_decay=1e-3)
    elif name == "RMSprop":
        opt = torch.optim.RMSprop(parameters, lr=lr)
    elif name == "Adagrad":
        opt = torch.optim.Adagrad(parameters, lr=lr)
    else:
        raise ValueError("Define an optimizer")
    print("{} Optimizer Defined with initial LR = {}".format(name, lr))
    return opt


def define_schedular(args, optimizer):
    o = args.OPTIMIZATION
    if o.schedular == "StepLR":
        sched = torch.optim.lr_scheduler.StepLR(optimizer, step_size=o.schedular_step_size, gamma=o.schedular_gamma)
        print("Learning Rate Decayed by StepLR")
    elif o.schedular == "MultiStepLR":
        sched = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=o.schedular_milestones, gamma=o.schedular_gamma)
    elif o.schedular == "ExponentialLR":
        sched = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=o.schedular_gamma)
    else:
        raise ValueError("decay_lr in config set to True but no schedular given")
    print("{} Schedular Defined with gamma = {} ".format(o.schedular, o.schedular_gamma))
    return sched


def set_train(models):
    for m in models.values():
        m.train()
    return models


def set_eval(models):
    for m in models.values():
        m.eval()
    return models


def convert_disp_to_depth(disp, min_depth, max_depth):
    lo, hi = 1 / max_depth, 1 / min_depth
    return 1 / (lo + (hi - lo) * disp)


def scale_disp(disp, min_depth, max_depth):
    lo, hi = 1 / max_depth, 1 / min_depth
    return lo + (hi - lo) * disp


def inverse_T_matrix(T):
    return torch.pinverse(T)


def scale_by_f(focal_data, focal_pretrain, depth):
    return depth * (focal_data / focal_pretrain)


def normalize_intrinsics(args, K):
    if args.DATA.name not in ("ICL", "TUM"):
        raise ValueError("normalize intrinsics not supported for this dataset")
    K[:, 0, :] /= 640.0
    K[:, 1, :] /= 480.0
    return K


def sparse_sampling(sampling_type, prob, depth):
    if sampling_type != "random":
        raise ValueError("Sampling type not implemented")
    mask = torch.rand_like(depth)
    mask[mask >= prob] = 0.0
    mask[mask > 0.0] = 1.0
    mask[depth == 0.0] = 0.0
    return depth * mask, mask


def torch_poses_to_transforms(poses):
    """(B,L,4,4) absolute poses -> T_0 = I, T_s = pinv(P_{s-1}) P_s (reference: training_utils.py:191-216)."""
    out = poses.detach().clone()
    eye = torch.eye(4, device=poses.device, dtype=poses.dtype)
    for b in range(poses.shape[0]):
        for s in range(poses.shape[1]):
            out[b, s] = eye if s == 0 else torch.pinverse(poses[b, s - 1]).matmul(poses[b, s])
    return out
