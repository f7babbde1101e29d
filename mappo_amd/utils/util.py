"""Shape / schedule helpers with the reference's names (onpolicy/utils/util.py:9-51,
onpolicy/algorithms/utils/util.py:15-17).  Spaces are duck-typed by class name, so gym is not needed."""
import math

import numpy as np
import torch


class Discrete:
    """Stand-in for gym.spaces.Discrete (the reference matches spaces by class *name*)."""

    def __init__(self, n):
        self.n = int(n)

    def __repr__(self):
        return f"Discrete({self.n})"


def check(x):
    """onpolicy/algorithms/utils/util.py:15-17."""
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def get_shape_from_obs_space(obs_space):
    name = obs_space.__class__.__name__
    if name == "Box":
        return obs_space.shape
    if name == "list":
        return obs_space
    raise NotImplementedError(name)


def get_shape_from_act_space(act_space):
    name = act_space.__class__.__name__
    if name == "Discrete":
        return 1
    if name in ("MultiDiscrete", "Box", "MultiBinary"):
        raise NotImplementedError(f"{name} action spaces are outside this build (BASELINE configs are all Discrete; "
                                  "Box / MultiBinary are broken in the reference itself, SURVEY.md §8c)")
    raise NotImplementedError(name)


def obs_dim_of(space):
    shape = get_shape_from_obs_space(space)
    if isinstance(shape[-1], list):            # shared_buffer.py:39-43 (SMAC-style [dim, [..], ...])
        shape = shape[:1]
    if len(shape) != 1:
        raise NotImplementedError("image observations (CNNBase) are outside this build (SURVEY.md §2.1 #11)")
    return int(shape[0])


def get_gard_norm(it):
    """utils/util.py:9-15 (spelling of the reference kept): L2 norm over the .grad of every parameter that has one."""
    total = 0.0
    for p in it:
        if p.grad is not None:
            total += float(p.grad.norm()) ** 2
    return math.sqrt(total)


def huber_loss(e, d):
    """utils/util.py:23-26: e^2/2 inside |e| <= d, d(|e| - d/2) outside (the fused kernels use the same formula)."""
    inside = (e.abs() <= d).to(e.dtype)
    return inside * e * e / 2 + (1 - inside) * d * (e.abs() - d / 2)


def mse_loss(e):
    """utils/util.py:28-29"""
    return e * e / 2


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """utils/util.py:17-21: lr = initial_lr * (1 - epoch/total)."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for group in optimizer.param_groups:
        group["lr"] = lr
    if hasattr(optimizer, "sync_lr"):
        optimizer.sync_lr()


def to_device_f32(x, device):
    """numpy / torch (any device, any float/bool dtype) -> contiguous float32 tensor on `device`."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=torch.float32, non_blocking=True).contiguous()
