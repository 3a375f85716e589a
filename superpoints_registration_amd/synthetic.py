"""Deterministic synthetic inputs and weights.

There is no network for datasets or checkpoints, so benchmarks, smoke tests
and golden fixtures all use: (a) seeded surface-like point-cloud pairs and
(b) weights filled by a generator keyed on the parameter NAME, so that the
reference model (in the dev container) and this package get bit-identical
parameters without shipping a 31 MB state dict.
"""
import hashlib
import math

import numpy as np
import torch


def rotation_z(theta: float) -> np.ndarray:
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def box_faces(n: int, rng: np.random.Generator, extent: float = 2.0) -> np.ndarray:
    """n points, uniform on three mutually orthogonal faces of a box
    (surface-like density, SURVEY.md 8d cfg 1/2)."""
    face = rng.integers(0, 3, n)
    uv = rng.uniform(0.0, extent, (n, 2))
    p = np.zeros((n, 3))
    for a in range(3):
        m = face == a
        others = [d for d in range(3) if d != a]
        p[np.ix_(m, others)] = uv[m]
    return p


def make_pair(n: int, seed: int = 0, extent: float = 2.0, jitter: float = 0.005,
              theta: float = 0.2, trans=(0.1, -0.05, 0.02)):
    """(src [n,3], tgt [n,3], pose_gt [3,4]) float32; tgt = R src + t + noise."""
    rng = np.random.default_rng(seed)
    src = box_faces(n, rng, extent) + rng.normal(0.0, jitter, (n, 3))
    R = rotation_z(theta)
    t = np.asarray(trans, dtype=np.float64)
    tgt = src @ R.T + t + rng.normal(0.0, jitter, (n, 3))
    tgt = tgt[rng.permutation(n)]
    pose = np.concatenate([R, t[:, None]], axis=1)
    return src.astype(np.float32), tgt.astype(np.float32), pose.astype(np.float32)


def make_sphere_pair(n: int, seed: int = 0, radius: float = 0.5, keep: float = 0.7):
    """ModelNet-shaped: points on a unit-ish sphere, partial crops (cfg 5)."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    v = radius * v / np.linalg.norm(v, axis=1, keepdims=True)
    d1, d2 = rng.normal(size=3), rng.normal(size=3)
    k = int(n * keep)
    src = v[np.argsort(v @ d1)[-k:]]
    R = rotation_z(0.4)
    tgt = v[np.argsort(v @ d2)[-k:]] @ R.T + np.array([0.1, 0.05, -0.02])
    pose = np.concatenate([R, np.array([[0.1], [0.05], [-0.02]])], axis=1)
    return src.astype(np.float32), tgt.astype(np.float32), pose.astype(np.float32)


def _seed_of(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little") % (2 ** 63 - 1)


def voxel_first_point(points: np.ndarray, voxel: float) -> np.ndarray:
    """numpy statement of the KITTI loader's pre-downsampling (kitti_pred.py:12-14: kiss_icp keeps the
    first point of every voxel, voxel = trunc(p / voxel) in float64); kept points in original order."""
    vox = np.trunc(points.astype(np.float64) / voxel).astype(np.int64)
    _, first = np.unique(vox, axis=0, return_index=True)
    return points[np.sort(first)]


def make_lidar_pair(n: int = 120000, seed: int = 0, r_max: float = 26.0, voxel: float = 0.3):
    """KITTI-odometry-shaped pair (SURVEY.md 8d cfg 4): a rotating-LiDAR-like scan -- radial point
    density ~ 1/r on a ground plane out to r_max with 0.3 m height noise, plus building-like
    vertical faces and a few poles -- seen from two poses ~1.5 m apart, each pre-voxelised at
    `voxel` like the reference loader.  Returns (src, tgt, pose_gt [3,4]) float32; tuned so that
    the KITTI config ends with ~1-2 k superpoints per cloud."""
    rng = np.random.default_rng(seed)

    def scan():
        n_g = int(0.62 * n)
        r = rng.uniform(2.5, r_max, n_g)                      # area density ~ 1/r
        th = rng.uniform(0, 2 * np.pi, n_g)
        ground = np.stack([r * np.cos(th), r * np.sin(th), rng.normal(0.0, 0.3, n_g) - 1.7], 1)
        walls = []
        n_w = n - n_g
        for k in range(24):                                   # facades between 6 m and r_max - 3 m, 2..6 m tall
            m = n_w // 24
            ang = rng.uniform(0, 2 * np.pi)
            dist = rng.uniform(6, r_max - 3)
            length, height = rng.uniform(5, 14), rng.uniform(2, 6)
            u = rng.uniform(-0.5, 0.5, m) * length
            h = rng.uniform(0, 1, m) ** 1.5 * height - 1.7
            c, s_ = np.cos(ang), np.sin(ang)
            base = np.array([dist * c, dist * s_])
            tangent = np.array([-s_, c])
            xy = base[None] + u[:, None] * tangent[None] + rng.normal(0, 0.03, (m, 2))
            walls.append(np.concatenate([xy, h[:, None]], 1))
        return np.concatenate([ground] + walls)
    world = scan()
    R = rotation_z(0.035)
    t = np.array([1.45, -0.3, 0.02])
    src = world + rng.normal(0, 0.02, world.shape)
    tgt = (world[rng.permutation(len(world))] + rng.normal(0, 0.02, world.shape)) @ R.T + t
    pose = np.concatenate([R, t[:, None]], 1)
    return (voxel_first_point(src.astype(np.float32), voxel), voxel_first_point(tgt.astype(np.float32), voxel),
            pose.astype(np.float32))


def make_lidar_translated_pair(n: int = 120000, seed: int = 0, shift=(4.8, -2.4, 0.0)):
    """KITTI-shaped pair whose pose solve is WELL conditioned whatever the network weights: the target
    is the (pre-voxelised) source scan translated by a multiple of every pyramid voxel size (0.3 m *
    2^level), in a random point order.  Both clouds then get the same voxel grids, neighbourhoods and
    therefore (up to the positional embedding) the same encoder features point for point, so even a
    randomly initialised matcher finds the true correspondences with high confidence, and the weighted
    Kabsch solve sees a consistent rigid motion instead of random matches.  Returns (src, tgt, pose_gt)."""
    src, _, _ = make_lidar_pair(n, seed)
    rng = np.random.default_rng(seed + 1000)
    t = np.asarray(shift, np.float32)
    tgt = (src[rng.permutation(len(src))] + t[None]).astype(np.float32)
    pose = np.concatenate([np.eye(3, dtype=np.float32), t[:, None]], 1)
    return src, tgt, pose


@torch.no_grad()
def fill_parameters(model: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every entry of model.state_dict() with values drawn from a
    CPU torch.Generator seeded by (seed, parameter name).

    Scales keep activations O(1) through the network: matrices ~ U(-b, b) with
    b = sqrt(3 / fan_in); LayerNorm weights ~ 1 + U(-0.1, 0.1); biases
    ~ U(-0.1, 0.1); kernel points = the disposition table scaled by the block's
    radius, rotated about z by a name-derived angle."""
    from .kernel_points import K015_CENTER_3D

    modules = dict(model.named_modules())
    sd = model.state_dict()
    for name, t in sd.items():
        g = torch.Generator().manual_seed(_seed_of(name, seed))
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'kernel_points':
            radius = modules[name.rsplit('.', 1)[0]].radius
            ang = float(torch.rand((), generator=g)) * 2 * math.pi
            kp = (K015_CENTER_3D * radius) @ rotation_z(ang)
            new = torch.tensor(kp, dtype=torch.float32)
        elif t.dim() == 0:
            new = torch.tensor(1.0) + 0.2 * (torch.rand((), generator=g) - 0.5)
        elif t.dim() == 1:
            u = torch.rand(t.shape, generator=g) * 2 - 1
            is_norm_weight = leaf == 'weight' and ('norm' in name)
            new = (1.0 + 0.1 * u) if is_norm_weight else 0.1 * u
        else:
            if leaf == 'weights':           # KPConv [K, Cin, Cout]
                fan_in = t.shape[0] * t.shape[1] / 4.0
            elif name.endswith('.W'):       # loss-only bilinear forms
                fan_in = t.shape[1] * 10.0
            else:                            # Linear / in_proj [out, in]
                fan_in = t.shape[1]
            b = math.sqrt(3.0 / max(fan_in, 1.0))
            new = (torch.rand(t.shape, generator=g) * 2 - 1) * b
        t.copy_(new.to(t.dtype).to(t.device))


def rand(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """Reproducible U(lo, hi) float32 CPU tensor (fixtures store the seed, not
    the data)."""
    g = torch.Generator().manual_seed(int(seed))
    return torch.rand(tuple(shape), generator=g) * (hi - lo) + lo
