"""Rigid-transform API of the hot path (reference: src/utils/se3_torch.py).

Computed by the HIP library:
  compute_rigid_transform(a, b, weights)   se3_torch.py:109-163 (weighted Kabsch)
The Sinkhorn variant (se3_torch.py:166-239) is fused with the affinity
construction in the library (ops.sinkhorn_correspondences) and therefore has
no stand-alone "affinity matrix in" entry point here.

3x4 pose algebra (compose / invert / apply / compare, se3_torch.py:32-106) is
twelve numbers per pair -- plain tensor expressions, device agnostic.
"""
import math
from typing import List, Sequence, Union

import torch
from torch import Tensor

from . import ops


def _split(pose: Tensor):
    return pose[..., :, :3], pose[..., :, 3:]


def se3_init(rot: Tensor = None, trans: Tensor = None) -> Tensor:
    """[R|t] from a rotation (…,3,3) and/or translation (…,3,1); the missing
    part defaults to identity / zero."""
    if rot is None and trans is None:
        raise ValueError("need a rotation or a translation")
    if rot is None:
        rot = torch.eye(3, dtype=trans.dtype, device=trans.device).expand(*trans.shape[:-2], 3, 3)
    if trans is None:
        trans = rot.new_zeros(*rot.shape[:-2], 3, 1)
    return torch.cat((rot, trans), dim=-1)


def se3_cat(a: Tensor, b: Tensor) -> Tensor:
    """a o b (apply b first)."""
    ra, ta = _split(a)
    rb, tb = _split(b)
    return torch.cat((ra @ rb, ra @ tb + ta), dim=-1)


def se3_inv(pose: Tensor) -> Tensor:
    r, t = _split(pose)
    rt = r.transpose(-1, -2)
    return torch.cat((rt, -(rt @ t)), dim=-1)


def se3_transform(pose: Tensor, xyz: Tensor) -> Tensor:
    """R x + t for xyz ([B,] N, 3)."""
    if xyz.shape[-1] != 3 or pose.shape[:-2] != xyz.shape[:-2]:
        raise ValueError(f"shape mismatch: pose {tuple(pose.shape)} xyz {tuple(xyz.shape)}")
    r, t = _split(pose)
    return xyz @ r.transpose(-1, -2) + t.transpose(-1, -2)


def se3_transform_list(pose: Union[Sequence[Tensor], Tensor], xyz: List[Tensor]) -> List[Tensor]:
    return [se3_transform(pose[i], pts) for i, pts in enumerate(xyz)]


def se3_compare(a: Tensor, b: Tensor) -> dict:
    """Rotation error (degrees, from the trace) and translation error of a o b^-1."""
    delta = se3_cat(a, se3_inv(b))
    cos = 0.5 * (torch.diagonal(delta[..., :, :3], dim1=-2, dim2=-1).sum(-1) - 1.0)
    return {
        'rot_deg': torch.rad2deg(torch.acos(cos.clamp(-1.0, 1.0))),
        'trans': delta[..., :, 3].norm(dim=-1),
        'chamfer': 0,
    }


def compute_rigid_transform(a: Tensor, b: Tensor, weights: Tensor = None,
                            check_weights: bool = True) -> Tensor:
    """T ([*,]3,4) with T a ~ b in the weighted least-squares sense.

    a, b: ([*,] N, 3); weights: ([*,] N) in [0, 1] or None.  Error behaviour
    follows the reference: shape mismatches and out-of-range weights raise
    AssertionError (the range check costs a device->host sync, se3_torch.py:132;
    the batched RegTR path passes check_weights=False).
    """
    if a.shape != b.shape or a.shape[-1] != 3:
        raise AssertionError(f"a {tuple(a.shape)} / b {tuple(b.shape)}")
    if weights is not None:
        if a.shape[:-1] != weights.shape:
            raise AssertionError(f"weights {tuple(weights.shape)} vs points {tuple(a.shape)}")
        if check_weights:
            lo, hi = torch.aminmax(weights)
            if not (float(lo) >= 0.0 and float(hi) <= 1.0):
                raise AssertionError("weights must lie in [0, 1]")
    lead, n = a.shape[:-2], a.shape[-2]
    n_sets = int(math.prod(lead)) if len(lead) else 1
    set_cu = torch.arange(n_sets + 1, dtype=torch.int32, device=a.device) * n
    flat_w = None if weights is None else weights.reshape(-1)
    pose = ops.weighted_procrustes(a.reshape(-1, 3), b.reshape(-1, 3), flat_w, set_cu)
    return pose.view(*lead, 3, 4)
