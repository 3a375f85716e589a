"""KPConv operator and block modules -- host-side mirror of the reference's
``src/models/backbone_kpconv/kpconv_blocks.py`` (same class names, constructor
arguments, forward signatures and state-dict names), computing through the HIP
library.  Under torch.no_grad() (inference) the calls are plain C-ABI launches; with
gradients enabled they go through the explicit HIP backward of autograd.py.

Not mirrored (never selected by a shipped config):
deformable / modulated KPConv, 'closest' aggregation, 'constant'/'gaussian'
influence, GlobalAverage / NearestUpsample / MaxPool blocks.
"""
import math

import torch
import torch.nn as nn
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

import os

from . import _concurrency, ops
from .kernel_points import load_kernels

_NORM_FOLD = os.environ.get("SPR_NO_NORM_FOLD", "0") != "1"   # experiment switch (A/B timing)


def _cu_of(batch, layer_ind):
    """int32 cu_seqlens of pyramid level `layer_ind` (cached in the meta dict)."""
    cache = batch.setdefault('_cu', {})
    cu = cache.get(layer_ind)
    if cu is None:
        cu = ops.lengths_to_cu(batch['stack_lengths'][layer_ind],
                               batch['points'][layer_ind].device)
        cache[layer_ind] = cu
    return cu


def _maxlen_of(batch, layer_ind):
    """Host-side longest cloud of a level (None when the meta dict did not come
    from our Preprocessor: the op then falls back to a safe upper bound)."""
    lens = batch.get('_lens_host')
    return max(lens[layer_ind]) if lens is not None else None


def _idx_of(batch, key, layer_ind):
    """int32 view of an index matrix of the meta dict (the public entries are
    int64 like the reference's; the kernels read int32)."""
    cache = batch.setdefault('_i32', {})
    t = cache.get((key, layer_ind))
    if t is None:
        t = batch[key][layer_ind].to(torch.int32)
        cache[(key, layer_ind)] = t
    return t


def gather(x, idx, method=2):
    """kpconv_blocks.py:68-99.  x[idx] for a 1-D or 2-D index tensor."""
    if idx.dim() == 1:
        return ops.gather_rows(x, idx.to(torch.int32))
    flat = ops.gather_rows(x, idx.reshape(-1).to(torch.int32))
    return flat.view(*idx.shape, x.shape[1])


def _pool_order_of(batch, level):
    po = batch.get('pool_order') if isinstance(batch, dict) else None
    return po[level] if po is not None and level < len(po) else None


def max_pool(x, inds, order=None):
    """kpconv_blocks.py:127-143 (shadow index reads a zero row).  order: optional spatial walk order of the query
    rows (ops.cell_order; the pyramid builder supplies it as batch['pool_order']) -- same result, fewer HBM reads."""
    return ops.maxpool(x, inds, order)


def closest_pool(x, inds):
    """kpconv_blocks.py:112-124."""
    return ops.gather_rows(x, inds[:, 0].contiguous().to(torch.int32))


class KPConv(nn.Module):
    """kpconv_blocks.py:175-420 (rigid, 'linear' influence, 'sum' aggregation)."""

    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius,
                 fixed_kernel_points='center', KP_influence='linear', aggregation_mode='sum',
                 deformable=False, modulated=False):
        super().__init__()
        if deformable or modulated:
            raise NotImplementedError("deformable / modulated KPConv is outside the hot-path scope")
        if KP_influence != 'linear' or aggregation_mode != 'sum':
            raise NotImplementedError("only KP_influence='linear', aggregation_mode='sum' "
                                      "(the only mode the shipped configs use)")
        self.K = kernel_size
        self.p_dim = p_dim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.radius = radius
        self.KP_extent = KP_extent
        self.fixed_kernel_points = fixed_kernel_points
        self.KP_influence = KP_influence
        self.aggregation_mode = aggregation_mode
        self.deformable = deformable
        self.modulated = modulated
        self.rows_sorted = False  # set by blocks fed from our Preprocessor
        self.impl = 0

        self.weights = Parameter(torch.zeros((self.K, in_channels, out_channels), dtype=torch.float32),
                                 requires_grad=True)
        self.reset_parameters()
        self.kernel_points = self.init_KP()

    def reset_parameters(self):
        kaiming_uniform_(self.weights, a=math.sqrt(5))

    def init_KP(self):
        K_points_numpy = load_kernels(self.radius, self.K, dimension=self.p_dim,
                                      fixed=self.fixed_kernel_points)
        return Parameter(torch.tensor(K_points_numpy, dtype=torch.float32), requires_grad=False)

    def forward(self, q_pts, s_pts, neighb_inds, x):
        return ops.kpconv(q_pts, s_pts, neighb_inds, x, self.weights,
                          self.kernel_points.detach(), self.KP_extent,
                          rows_sorted=self.rows_sorted, impl=self.impl)

    def __repr__(self):
        return 'KPConv(radius: {:.2f}, extent: {:.2f}, in_feat: {:d}, out_feat: {:d})'.format(
            self.radius, self.KP_extent, self.in_channels, self.out_channels)


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, config):
    """kpconv_blocks.py:429-471."""
    if block_name == 'unary':
        return UnaryBlock(in_dim, out_dim, config.use_batch_norm, config.batch_norm_momentum)
    elif block_name in ['simple', 'simple_strided']:
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    elif block_name in ['resnetb', 'resnetb_strided']:
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    else:
        raise ValueError('Unknown / unsupported block name in the architecture definition : ' + block_name)


class BatchNormBlock(nn.Module):
    """kpconv_blocks.py:474-531: per-cloud InstanceNorm1d (no parameters), or a
    learnable bias when use_bn is False."""

    def __init__(self, in_dim, use_bn, bn_momentum):
        super().__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.in_dim = in_dim
        self.eps = 1e-5
        if not self.use_bn:
            self.bias = Parameter(torch.zeros(in_dim, dtype=torch.float32), requires_grad=True)

    def forward(self, x, stack_lengths, cu=None, add=None, slope=1.0, max_len=None):
        """`cu`, `add`, `slope`, `max_len` are extensions used by the fused
        blocks below: out = lrelu(norm(x) + add, slope)."""
        if cu is None:
            cu = ops.lengths_to_cu(stack_lengths, x.device)
        if self.use_bn:
            return ops.instnorm(x, cu, eps=self.eps, norm=True, add=add, slope=slope, max_len=max_len)
        y = x + self.bias
        return ops.instnorm(y, cu, norm=False, add=add, slope=slope, max_len=max_len)

    def __repr__(self):
        return 'BatchNormBlock(in_feat: {:d}, momentum: {:.3f}, only_bias: {:s})'.format(
            self.in_dim, self.bn_momentum, str(not self.use_bn))


class UnaryBlock(nn.Module):
    """kpconv_blocks.py:533-566: Linear(no bias) -> norm -> LeakyReLU(0.1)."""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.no_relu = no_relu
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self.batch_norm = BatchNormBlock(out_dim, self.use_bn, self.bn_momentum)
        if not no_relu:
            self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, stack_lengths=None, cu=None, add=None, final_slope=None, max_len=None):
        """out = act(norm(x W^T) [+ add]); `add` / `final_slope` let the
        bottleneck block fuse its residual add + LeakyReLU into this pass."""
        y = ops.linear(x, self.mlp.weight)
        slope = 1.0 if self.no_relu else 0.1
        if final_slope is not None:
            slope = final_slope
        return self.batch_norm(y, stack_lengths, cu=cu, add=add, slope=slope, max_len=max_len)

    def __repr__(self):
        return 'UnaryBlock(in_feat: {:d}, out_feat: {:d}, BN: {:s}, ReLU: {:s})'.format(
            self.in_dim, self.out_dim, str(self.use_bn), str(not self.no_relu))


class SimpleBlock(nn.Module):
    """kpconv_blocks.py:590-646."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim // 2,
                             current_extent, radius,
                             fixed_kernel_points=config.fixed_kernel_points,
                             KP_influence=config.KP_influence,
                             aggregation_mode=config.aggregation_mode,
                             deformable='deform' in block_name, modulated=config.modulated)
        self.batch_norm = BatchNormBlock(out_dim // 2, self.use_bn, self.bn_momentum)
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch):
        li = self.layer_ind
        if 'strided' in self.block_name:
            q_pts, s_pts = batch['points'][li + 1], batch['points'][li]
            neighb_inds = _idx_of(batch, 'pools', li)
            stack_lengths, cu, ml = batch['stack_lengths'][li + 1], _cu_of(batch, li + 1), _maxlen_of(batch, li + 1)
        else:
            q_pts = s_pts = batch['points'][li]
            neighb_inds = _idx_of(batch, 'neighbors', li)
            stack_lengths, cu, ml = batch['stack_lengths'][li], _cu_of(batch, li), _maxlen_of(batch, li)
        self.KPConv.rows_sorted = bool(batch.get('_rows_sorted', False))
        x = self.KPConv(q_pts, s_pts, neighb_inds, x)
        return self.batch_norm(x, stack_lengths, cu=cu, slope=0.1, max_len=ml)


class ResnetBottleneckBlock(nn.Module):
    """kpconv_blocks.py:649-741."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.block_name = block_name
        self.layer_ind = layer_ind
        self.in_dim = in_dim
        self.out_dim = out_dim
        if in_dim != out_dim // 4:
            self.unary1 = UnaryBlock(in_dim, out_dim // 4, self.use_bn, self.bn_momentum)
        else:
            self.unary1 = nn.Identity()
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, out_dim // 4,
                             out_dim // 4, current_extent, radius,
                             fixed_kernel_points=config.fixed_kernel_points,
                             KP_influence=config.KP_influence,
                             aggregation_mode=config.aggregation_mode,
                             deformable='deform' in block_name, modulated=config.modulated)
        self.batch_norm_conv = BatchNormBlock(out_dim // 4, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(out_dim // 4, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        if in_dim != out_dim:
            self.unary_shortcut = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum,
                                             no_relu=True)
        else:
            self.unary_shortcut = nn.Identity()
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, features, batch):
        li = self.layer_ind
        stack_lengths_pre, cu_pre, ml_pre = batch['stack_lengths'][li], _cu_of(batch, li), _maxlen_of(batch, li)
        if 'strided' in self.block_name:
            q_pts, s_pts = batch['points'][li + 1], batch['points'][li]
            neighb_inds = _idx_of(batch, 'pools', li)
            stack_lengths_post, cu_post, ml_post = (batch['stack_lengths'][li + 1], _cu_of(batch, li + 1),
                                                    _maxlen_of(batch, li + 1))
        else:
            q_pts = s_pts = batch['points'][li]
            neighb_inds = _idx_of(batch, 'neighbors', li)
            stack_lengths_post, cu_post, ml_post = batch['stack_lengths'][li], _cu_of(batch, li), ml_pre

        strided = 'strided' in self.block_name
        projected = isinstance(self.unary_shortcut, UnaryBlock)
        # Inference: unary2, the shortcut projection, both norms, the add and the LeakyReLU are ONE fused
        # operator that never writes the un-normalised projections (ops.block_tail; csrc/block_tail.hip).
        mid = self.unary2.in_dim
        fused = (features.is_cuda and not torch.is_grad_enabled() and self.use_bn
                 and ops.block_tail_tile_rows(mid, self.in_dim if projected else 0, self.out_dim) > 0)

        def shortcut_branch():
            sc = max_pool(features, neighb_inds, _pool_order_of(batch, li)) if strided else features
            if projected and not fused:
                sc = self.unary_shortcut(sc, stack_lengths_post, cu=cu_post, max_len=ml_post)
            return sc

        # The shortcut branch (max-pool gather, projection, norm) depends on the block input only:
        # in inference it runs on its own stream beside unary1 -> KPConv -> norm and joins at unary2.
        has_work = strided or (projected and not fused)
        fork = (has_work and features.is_cuda and not torch.is_grad_enabled()
                and _concurrency.active(features.device))
        if fork:
            main = torch.cuda.current_stream(features.device)
            branch = _concurrency.aux_stream(main, features.device, 'shortcut')
            ready = torch.cuda.Event()
            ready.record(main)                    # block input (and, transitively, the pyramid level) is ready
            branch.wait_event(ready)
            with torch.cuda.stream(branch):
                shortcut = shortcut_branch()
                joined = torch.cuda.Event()
                joined.record(branch)
            features.record_stream(branch)

        x = self.unary1(features, stack_lengths_pre, cu=cu_pre, max_len=ml_pre) \
            if isinstance(self.unary1, UnaryBlock) else features
        self.KPConv.rows_sorted = bool(batch.get('_rows_sorted', False))
        x = self.KPConv(q_pts, s_pts, neighb_inds, x)
        # Inference with the fused tail: the norm + LeakyReLU behind the KPConv runs while the tail stages its
        # tiles (ops.block_tail(xa_stats=...)): only the statistics passes remain of it, the normalised tensor is
        # never written or read back (SPR_NO_NORM_FOLD=1: the separate operator, for A/B).
        fold = fused and self.use_bn and _NORM_FOLD
        conv_stats = None
        if fold:
            conv_stats = ops.instnorm_stats(x, cu_post, eps=self.batch_norm_conv.eps, max_len=ml_post)
        else:
            x = self.batch_norm_conv(x, stack_lengths_post, cu=cu_post, slope=0.1, max_len=ml_post)

        if fork:
            main.wait_event(joined)
            shortcut.record_stream(main)
        else:
            shortcut = shortcut_branch()
        if fused:
            if projected:
                return ops.block_tail(x, self.unary2.mlp.weight, cu_post, xb=shortcut,
                                      wb=self.unary_shortcut.mlp.weight, eps=self.unary2.batch_norm.eps, slope=0.1,
                                      xa_stats=conv_stats, xa_slope=0.1, xa_max_len=ml_post)
            return ops.block_tail(x, self.unary2.mlp.weight, cu_post, add=shortcut,
                                  eps=self.unary2.batch_norm.eps, slope=0.1,
                                  xa_stats=conv_stats, xa_slope=0.1, xa_max_len=ml_post)
        # unary2 (no relu) + shortcut, then LeakyReLU: fused into unary2's norm pass
        return self.unary2(x, stack_lengths_post, cu=cu_post, add=shortcut, final_slope=0.1,
                           max_len=ml_post)
