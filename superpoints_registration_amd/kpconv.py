"""KPConv pyramid preprocessing and encoder on the HIP library.

Public names follow the reference's ``src/models/backbone_kpconv/kpconv.py`` so
that RegTR-style callers keep working: KPFEncoder (:22-92), Preprocessor
(:295-418) / PreprocessorGPU (:421-549), batch_grid_subsampling_kpconv (:174),
batch_neighbors_kpconv (:247) and their *_gpu aliases (:217, :265).

The pinned contract is the reference's CPU ``Preprocessor`` (its C++
extensions): identical point order at every level, bit-exact barycentres and
neighbour indices.  The reference's ``PreprocessorGPU`` depends on
MinkowskiEngine + PyTorch3D (not vendored, no runnable oracle); here both
names resolve to the same HIP implementation.

Structure differs from the reference: the architecture string list is parsed
ONCE into a pyramid plan (`plan_pyramid`) that both the encoder and the
preprocessor consume, instead of two hand-rolled loops with running state.
"""
from dataclasses import dataclass
from typing import List

import os
import torch
import torch.nn as nn

from . import ops
from .kpconv_blocks import block_decider

_DOWN = ('pool', 'strided')
_STOP = ('global', 'upsample')


@dataclass
class BlockPlan:
    name: str
    level: int       # pyramid level the block reads from
    radius: float    # convolution radius at that level
    in_dim: int
    out_dim: int
    down: bool       # ends the level (strided / pool)


@dataclass
class LevelPlan:
    radius: float    # r_normal of the level (kpconv.py:319, doubles per level :406)
    has_conv: bool   # at least one non-strided block -> conv neighbours needed
    down: bool       # the level ends with a strided block -> subsample + pools
    limit: int       # neighborhood_limits[level]


def plan_pyramid(cfg):
    """Parse cfg.architecture into per-block and per-level plans.

    Channel / radius bookkeeping follows KPFEncoder.__init__ (kpconv.py:27-79):
    'simple' halves the advertised width, every strided block doubles radius
    and width and opens a new level.  Level bookkeeping follows
    Preprocessor.forward (kpconv.py:334-408)."""
    r = cfg.first_subsampling_dl * cfg.conv_radius
    in_dim, out_dim, level = cfg.in_feats_dim, cfg.first_feats_dim, 0
    blocks: List[BlockPlan] = []
    members: List[List[BlockPlan]] = [[]]
    for name in cfg.architecture:
        if any(s in name for s in _STOP):
            break
        down = any(s in name for s in _DOWN)
        bp = BlockPlan(name, level, r, in_dim, out_dim, down)
        blocks.append(bp)
        members[-1].append(bp)
        in_dim = out_dim // 2 if 'simple' in name else out_dim
        if down:
            level += 1
            r *= 2
            out_dim *= 2
            members.append([])
    if not members[-1]:
        members.pop()
    levels = []
    for l, mem in enumerate(members):
        levels.append(LevelPlan(radius=mem[0].radius,
                                has_conv=any(not b.down for b in mem),
                                down=mem[-1].down,
                                limit=int(cfg.neighborhood_limits[l])))
    return blocks, levels, in_dim


class KPFEncoder(nn.Module):
    """Sequential KPConv encoder.  Attributes `encoder_blocks`,
    `encoder_skips`, `encoder_skip_dims` and the (x, skip_x) return value match
    the reference (kpconv.py:22-92)."""

    def __init__(self, config, d_bottle, increase_channel_when_downsample=True):
        super().__init__()
        if not increase_channel_when_downsample:
            raise NotImplementedError("constant-width pyramids are not used by RegTR")
        blocks, _, final_dim = plan_pyramid(config)
        self._plans = blocks
        self.encoder_blocks = nn.ModuleList(
            block_decider(b.name, b.radius, b.in_dim, b.out_dim, b.level, config) for b in blocks)
        # skip taps: the input of every strided block, plus the final features
        self.encoder_skips = [i for i, b in enumerate(blocks) if b.down]
        self.encoder_skip_dims = [b.in_dim for b in blocks if b.down]
        self.encoder_skips.append(len(blocks) - 1)
        self.encoder_skip_dims.append(final_dim)

    def forward(self, x, batch):
        skip_x = []
        for i, block in enumerate(self.encoder_blocks):
            if i in self.encoder_skips:
                skip_x.append(x)
            x = block(x, batch)
        return x, skip_x

    def forward_streamed(self, x, stream):
        """forward() driven by Preprocessor.stream(): every block is launched as soon as its part
        of the pyramid exists.  Same kernels, same operands -> bitwise the result of
        forward(x, meta).  Returns (x, skip_x, meta)."""
        plans = self._plans
        skip_x, i, meta = [], 0, None
        for meta, kind, level in stream:
            while i < len(self.encoder_blocks):
                bp = plans[i]
                ready = bp.level < level or (bp.level == level and (kind == 'down' or not bp.down))
                if not ready:
                    break
                if i in self.encoder_skips:
                    skip_x.append(x)
                x = self.encoder_blocks[i](x, meta)
                i += 1
        assert i == len(self.encoder_blocks), "pyramid ended before the encoder"
        return x, skip_x, meta


# --------------------------------------------------------------------------- #
def batch_grid_subsampling_kpconv(points, batches_len, features=None, labels=None, sampleDl=0.1,
                                  max_p=0, verbose=0, random_grid_orient=True,
                                  order=ops.ORDER_REFERENCE):
    """Replaces cpp_subsampling.subsample_batch behind kpconv.py:174-214.
    points [N,3] on the device, batches_len [B] -> (sub_points, sub_lengths)."""
    if features is not None or labels is not None:
        raise NotImplementedError('subsampling with features / labels is not on the RegTR path')
    return ops.grid_subsample(points, ops.lengths_to_cu(batches_len, points.device), sampleDl,
                              max_p=max_p, order=order)


def batch_neighbors_kpconv(queries, supports, q_batches, s_batches, radius, max_neighbors):
    """Replaces cpp_neighbors.batch_query + the column slice (kpconv.py:247-262).
    Returns int32 [Nq, min(max_count, max_neighbors)], shadow index = Ns."""
    limit = int(max_neighbors) if max_neighbors > 0 else 128
    idx, _ = ops.radius_neighbors(queries, supports,
                                  ops.lengths_to_cu(q_batches, queries.device),
                                  ops.lengths_to_cu(s_batches, queries.device),
                                  radius, limit, exact_width=True)
    return idx


batch_grid_subsampling_kpconv_gpu = batch_grid_subsampling_kpconv  # kpconv.py:217
batch_neighbors_kpconv_gpu = batch_neighbors_kpconv                # kpconv.py:265


class _IndexList(list):
    """List of per-level index matrices that holds the kernels' int32 tensors and converts
    an entry to the public dtype (int64, like the reference's torch.long indices) the first
    time it is READ.  The forward pass itself only uses the int32 views, so the three
    [N, limit] int64 copies per level (hundreds of MB per step) are only made for callers
    that actually look at them."""

    def __init__(self, dtype):
        super().__init__()
        self._dtype = dtype

    def _conv(self, i):
        t = list.__getitem__(self, i)
        if t.dtype != self._dtype:
            t = t.to(self._dtype)
            list.__setitem__(self, i, t)
        return t

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._conv(j) for j in range(*i.indices(len(self)))]
        return self._conv(i if i >= 0 else len(self) + i)

    def __iter__(self):
        return (self._conv(i) for i in range(len(self)))


_POOL_ORDER = os.environ.get("SPR_NO_POOL_ORDER", "") == ""   # A/B switch: max-pool in storage order


class Preprocessor(nn.Module):
    """Builds the KPConv pyramid metadata for a list of clouds.

    Returns boundary B2 of SURVEY.md section 8b: {'points', 'neighbors',
    'pools', 'upsamples', 'stack_lengths'}, lists over levels; index tensors
    are int64 (`index_dtype`) with shadow index = number of support points;
    the last level's pools / upsamples are (0, 1) placeholders like the
    reference (kpconv.py:389-392).  Private '_' keys carry what the kernels
    want: int32 index views, cu_seqlens, host-side lengths.

    compute_upsamples=False skips the up-sampling search the encoder-only
    RegTR never reads (kpconv.py:384 computes it regardless).
    order=ops.ORDER_CANONICAL emits subsampled points in ascending voxel-key
    order instead of the reference's hash-map order (better gather locality;
    poses agree to rounding, indices are a relabelling).
    """

    def __init__(self, cfg, compute_upsamples=True, order=ops.ORDER_REFERENCE,
                 index_dtype=torch.int64):
        super().__init__()
        self.cfg = cfg
        self.compute_upsamples = compute_upsamples
        self.order = order
        self.index_dtype = index_dtype
        # untruncated max row count of every search of the previous forward, keyed (level, kind): a search whose
        # rows were dense last time (more supports in range than 1.5 x the limit: LiDAR-shaped clouds) takes the
        # wave-per-query selection, the others the thread-per-query one (ops.RadiusTable.query; identical rows)
        self._row_counts = {}

    def forward(self, pts: List[torch.Tensor]):
        meta = None
        for meta, _, _ in self.stream(pts):
            pass
        return meta

    def stream(self, pts: List[torch.Tensor]):
        """Generator form of forward(): yields (meta, 'conv' | 'down', level) as soon as what an
        encoder block of that kind needs is in `meta` -- 'conv': points / lengths / conv
        neighbours of the level; 'down': pools (and upsamples) of the level plus the points and
        lengths of the next one.  RegTR runs it on a side stream and launches the encoder blocks
        between the yields (KPFEncoder.forward_streamed): the index work -- small, latency-bound
        kernels and one device->host read per subsampling and per neighbour matrix -- then runs
        beside the convolutions instead of in front of them.  Exhausting the generator leaves
        `meta` exactly as forward() returns it."""
        cfg = self.cfg
        _, levels, _ = plan_pyramid(cfg)
        device = pts[0].device
        points = torch.cat(pts, dim=0).to(torch.float32).contiguous()
        lens_host = [int(p.shape[0]) for p in pts]

        meta = {k: [] for k in ('points', 'stack_lengths')}
        meta.update({k: _IndexList(self.index_dtype) for k in ('neighbors', 'pools', 'upsamples')})
        meta.update(_cu={}, _i32={}, _lens_host=[], _rows_sorted=True, pool_order=[])
        placeholder = torch.zeros((0, 1), dtype=self.index_dtype, device=device)

        def publish(key, level, idx):
            if idx is None:
                meta[key].append(placeholder)
                return
            meta['_i32'][(key, level)] = idx
            meta[key].append(idx)                 # converted to index_dtype on first read
            if key != 'upsamples':                # conv / pool matrices feed KPConvs: tile plan built here, off the main stream
                ops.kpconv_plan_prefetch(idx, plan_ns[0])

        plan_ns = [0]                             # supports of the matrices being published (= points of the current level)

        def open_level(l, points, lens_host, cu):
            meta['points'].append(points)
            meta['stack_lengths'].append(torch.tensor(lens_host, dtype=torch.int32, device=device))
            meta['_cu'][l] = cu
            meta['_lens_host'].append(lens_host)

        # One cell table per (supports, radius): a level's conv and pool searches and the previous level's
        # up-sampling search (radius 2 r = the next level's r) all look into the same one.
        table = [None]

        def search(key, queries, q_cu, supports, s_cu, radius, limit):
            if table[0] is None or not table[0].matches(supports, s_cu, radius):
                table[0] = ops.RadiusTable(supports, s_cu, radius)
            prev = self._row_counts.get(key)
            idx, m = table[0].query(queries, q_cu, limit, dense=None if prev is None else prev > 1.5 * limit)
            self._row_counts[key] = m
            return idx

        cu = ops.lengths_to_cu(lens_host, device)
        open_level(0, points, lens_host, cu)
        for l, lv in enumerate(levels):
            conv = pool = up = None
            plan_ns[0] = points.shape[0]
            if lv.has_conv:
                conv = search((l, 'conv'), points, cu, points, cu, lv.radius, lv.limit)
            publish('neighbors', l, conv)
            yield meta, 'conv', l
            if lv.down:
                dl = 2 * lv.radius / cfg.conv_radius          # kpconv.py:367
                sub_points, sub_lens = ops.grid_subsample(points, cu, dl, order=self.order)
                sub_lens_host = sub_lens.tolist()
                sub_cu = ops.lengths_to_cu(sub_lens_host, device)
                pool = search((l, 'pool'), sub_points, sub_cu, points, cu, lv.radius, lv.limit)
                # spatial walk order of the level's pooled queries (kpconv_blocks.max_pool): one key + sort pass here,
                # off the main stream, saves the max-pool most of its re-reads of the finer level's features
                meta['pool_order'].append(ops.cell_order(sub_points, sub_cu, dl) if _POOL_ORDER else None)
                if self.compute_upsamples:
                    up = search((l, 'up'), points, cu, sub_points, sub_cu, 2 * lv.radius, lv.limit)
            if not lv.down:
                meta['pool_order'].append(None)
            publish('pools', l, pool)
            publish('upsamples', l, up)
            if lv.down:
                points, lens_host, cu = sub_points, sub_lens_host, sub_cu
                open_level(l + 1, points, lens_host, cu)
            yield meta, 'down', l


PreprocessorGPU = Preprocessor  # kpconv.py:421
