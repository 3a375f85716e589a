"""Tensor-level wrappers over the C ABI (include/spr.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream;
every computation below happens in libspr_hip.so.  Inputs must be CUDA(HIP)
tensors -- a CPU tensor raises, there is no fallback.
"""
import ctypes
import math
import os
import threading
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
ORDER_REFERENCE, ORDER_CANONICAL = 0, 1


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dev(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on the MI355X device (got "
                           f"{t.device if isinstance(t, torch.Tensor) else type(t)}); "
                           "the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


_ws_cache = {}


# ---- operand-range hand-over (include/spr.h, "Operand-range hand-over") ------------------------
# A producer kernel that has just written a tensor can publish max |x| as a few per-workgroup
# partials; the consuming GEMM then scales its operand without a measuring pass over x.  The
# range rides on the Python tensor object as an attribute and is honoured only while the tensor
# is untouched (same storage pointer, same version counter): views, copies and in-place edits drop it.
_range_epoch = [0]


def invalidate_ranges() -> None:
    """Drops every cached / published operand range.  Needed only after mutating a tensor behind
    autograd's back (`w.data.copy_(...)`, raw pointer writes): such edits do not move the version
    counter the ranges are keyed on."""
    _range_epoch[0] += 1


def _set_range(t: torch.Tensor, parts: torch.Tensor, n: int, guard=None) -> None:
    if n > 0:     # ONE attribute store: a reader on another thread sees the range and its guard together
        t._spr_range = (parts, int(n), t._version, t.data_ptr(), _range_epoch[0], guard)


_HANDOVER = os.environ.get("SPR_NO_RANGE_HANDOVER", "0") != "1"   # experiment switch (A/B timing)


def _get_range(t):
    if not _HANDOVER:
        return None, 0
    r = getattr(t, '_spr_range', None)
    if r is None or r[2] != t._version or r[3] != t.data_ptr() or r[4] != _range_epoch[0]:
        return None, 0
    if r[5] is not None:
        r[5].acquire()
    return r[0], r[1]

_RANGE_CAP = 4096   # partials a GEMM may publish (one per workgroup)
_STREAM_SLOTS = 512  # atomic-max slots of the streaming producers (InstanceNorm apply, max-pool)
_zero_pool = {}


def _zero_slots(n: int, device) -> torch.Tensor:
    """n zero-initialised floats from a per-(device, stream) pool: one memset per 256 K slots instead
    of one per LayerNorm call (its range slots are combined with atomic max and must start at 0)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    pool = _zero_pool.get(key)
    if pool is None or pool[1] + n > pool[0].numel():
        pool = [torch.zeros(1 << 18, dtype=torch.float32, device=device), 0]
        _zero_pool[key] = pool
    out = pool[0][pool[1]:pool[1] + n]
    pool[1] += n
    return out


class _StreamGuard:
    """Ordering of a cached, weight-side measurement against consumers on OTHER streams.  The
    measuring kernel runs on the stream of the thread that first needed it; the cache lives on the
    shared Parameter, and StreamedForward threads / the side streams of one forward read it from
    their own streams.  The producing stream records an event behind the measurement; the first
    consumer on any other stream waits for it (and registers the buffer with its stream)."""

    def __init__(self, buf: torch.Tensor):
        self.stream = torch.cuda.current_stream(buf.device)
        self.event = torch.cuda.Event()
        self.event.record(self.stream)
        self.buf = buf
        self.seen = {self.stream.cuda_stream}

    def acquire(self):
        cur = torch.cuda.current_stream(self.buf.device)
        if cur.cuda_stream not in self.seen:
            cur.wait_event(self.event)
            self.buf.record_stream(cur)
            self.seen.add(cur.cuda_stream)


def _static_range(w: torch.Tensor):
    """Range partials of a tensor that rarely changes (weights): measured once per (storage,
    version) with spr_absmax and kept on the tensor like a published range -- an optimizer step
    bumps the version counter and the next call measures again."""
    if not _HANDOVER:
        return None, 0
    r, n = _get_range(w)
    if r is not None:
        return r, n
    L = _lib.lib()
    n = L.spr_range_parts()
    flat = w.reshape(-1, w.shape[-1])
    parts = torch.empty((n,), dtype=torch.float32, device=w.device)
    _lib.check(L.spr_absmax(_ptr(flat), flat.shape[0], flat.shape[1], flat.shape[1], _ptr(parts), _stream(w)),
               "spr_absmax")
    _set_range(w, parts, n, _StreamGuard(parts))
    return parts, n


_PRIME_PARTS = 16


def prime_weight_ranges(params) -> int:
    """Measures the ranges of all (float32, contiguous, device) parameters whose cached range is stale in ONE launch
    (spr_absmax_multi) and attaches them like _static_range would: after an optimizer step every weight's version
    counter has moved, and ~150 separate measuring launches (16 us each) opened every training forward.  Returns the
    number of tensors measured."""
    if not _HANDOVER:
        return 0
    todo = [p for p in params if isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == torch.float32
            and p.dim() >= 2 and p.is_contiguous() and p.numel() > 0 and _get_range(p)[0] is None]
    if not todo:
        return 0
    dev = todo[0].device
    todo = [p for p in todo if p.device == dev]
    rec = np.zeros((len(todo), 2), dtype=np.int64)
    for i, p in enumerate(todo):
        rec[i, 0] = p.data_ptr()
        rec[i, 1] = p.numel()
    jobs = torch.from_numpy(rec).to(dev)
    parts = torch.empty((len(todo), _PRIME_PARTS), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().spr_absmax_multi(_ptr(jobs), len(todo), _PRIME_PARTS, _ptr(parts), _stream(parts)),
               "spr_absmax_multi")
    guard = _StreamGuard(parts)
    for i, p in enumerate(todo):
        _set_range(p, parts[i], _PRIME_PARTS, guard)
    return len(todo)


def weight_transposed(w: torch.Tensor) -> torch.Tensor:
    """Contiguous W^T of a [n, k] weight, kept on the tensor per (storage, version) like its range (the
    backward's dX = g W runs as the forward's NT product on it); guarded for readers on other streams."""
    c = getattr(w, '_spr_wt', None)
    if c is not None and c[1] == w._version and c[2] == w.data_ptr() and c[3] == _range_epoch[0]:
        c[4].acquire()
        return c[0]
    wt = w.detach().t().contiguous()
    # max |W^T| = max |W|: the transposed copy shares the weight's measured range instead of being scanned itself
    r, n = _static_range(w)
    if r is not None:
        rw = getattr(w, '_spr_range', None)
        _set_range(wt, r, n, rw[5] if rw is not None else None)
    w._spr_wt = (wt, w._version, w.data_ptr(), _range_epoch[0], _StreamGuard(wt))
    return wt


def ensure_range(t: torch.Tensor) -> torch.Tensor:
    """Measures max |t| once (spr_absmax) and publishes it on the tensor like a producer kernel would, unless a
    valid range is already attached: a gradient that feeds two products (dX and dW of a projection) is then
    scanned once instead of once per product."""
    if not _HANDOVER or t.dim() != 2 or not t.is_contiguous() or _get_range(t)[0] is not None:
        return t
    L = _lib.lib()
    n = L.spr_range_parts()
    parts = torch.empty((n,), dtype=torch.float32, device=t.device)
    _lib.check(L.spr_absmax(_ptr(t), t.shape[0], t.shape[1], t.shape[1], _ptr(parts), _stream(t)), "spr_absmax")
    _set_range(t, parts, n)
    return t


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per device+stream (stream-ordered reuse)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def lengths_to_cu(lengths, device) -> torch.Tensor:
    """int32 [nb+1] exclusive prefix of the per-cloud lengths."""
    if isinstance(lengths, torch.Tensor):
        l = lengths.to(device=device, dtype=torch.int32)
        cu = torch.zeros(l.numel() + 1, dtype=torch.int32, device=device)
        cu[1:] = torch.cumsum(l, 0)
        return cu
    arr = np.zeros(len(lengths) + 1, dtype=np.int32)
    arr[1:] = np.cumsum(np.asarray(lengths, dtype=np.int64))
    return torch.from_numpy(arr).to(device)


# --------------------------------------------------------------------------- #
def selftest() -> None:
    st = ctypes.c_int(-1)
    _lib.check(_lib.lib().spr_selftest(ctypes.byref(st)), "spr_selftest")


def grid_subsample(points: torch.Tensor, cu: torch.Tensor, dl: float, max_p: int = 0,
                   order: int = ORDER_REFERENCE) -> Tuple[torch.Tensor, torch.Tensor]:
    """a1.  Returns (sub_points [N',3] f32, lengths [nb] i32 on device).
    One device->host read of the output size (the reference returns exact
    shapes too)."""
    points = _dev(points, "points", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    n, nb = points.shape[0], cu.numel() - 1
    L = _lib.lib()
    ws = _workspace(L.spr_grid_subsample_workspace_bytes(n, nb), points.device)
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=points.device)
    out_lens = torch.empty((nb,), dtype=torch.int32, device=points.device)
    total = torch.empty((1,), dtype=torch.int32, device=points.device)
    _lib.check(L.spr_grid_subsample(_ptr(points), _ptr(cu), n, nb, float(dl), int(max_p), int(order),
                                    _ptr(out), _ptr(out_lens), _ptr(total), _ptr(ws), ws.numel(),
                                    _stream(points)), "spr_grid_subsample")
    m = int(total.item())
    if m < 0:
        raise RuntimeError("spr_grid_subsample: voxel grid too large for 40-bit keys")
    return out[:m], out_lens


def voxel_downsample(points: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """One point per voxel (the first one, voxel = trunc(p / voxel_size)): the GPU counterpart of
    the KITTI loader's kiss_icp pre-downsampling (kitti_pred.py:12-14, :203-204).  Returns the kept
    points in ascending original index (one device->host read of the count)."""
    points = _dev(points, "points", torch.float32)
    n = points.shape[0]
    L = _lib.lib()
    ws = _workspace(L.spr_voxel_downsample_workspace_bytes(n), points.device)
    idx = torch.empty((n,), dtype=torch.int32, device=points.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=points.device)
    _lib.check(L.spr_voxel_downsample(_ptr(points), n, float(voxel_size), _ptr(idx), _ptr(cnt), _ptr(ws), ws.numel(),
                                      _stream(points)), "spr_voxel_downsample")
    m = int(cnt.item())
    if m < 0:
        raise RuntimeError("spr_voxel_downsample: coordinates exceed 2^20 voxels")
    return points[idx[:m].long()]


def radius_neighbors(queries: torch.Tensor, supports: torch.Tensor, q_cu: torch.Tensor,
                     s_cu: torch.Tensor, radius: float, limit: int,
                     exact_width: bool = True, algo: int = 0) -> Tuple[torch.Tensor, int]:
    """a2.  int32 [Nq, W] neighbour indices (shadow = Ns) and the untruncated
    max count.  exact_width=True slices to W = min(max_count, limit) like the
    reference (one device->host read); False keeps W = limit (no sync)."""
    queries = _dev(queries, "queries", torch.float32)
    supports = _dev(supports, "supports", torch.float32)
    q_cu = _dev(q_cu, "q_cu", torch.int32)
    s_cu = _dev(s_cu, "s_cu", torch.int32)
    nq, ns, nb = queries.shape[0], supports.shape[0], q_cu.numel() - 1
    L = _lib.lib()
    ws = _workspace(L.spr_radius_neighbors_workspace_bytes(nq, ns, nb), queries.device)
    out = torch.empty((nq, limit), dtype=torch.int32, device=queries.device)
    mc = torch.empty((1,), dtype=torch.int32, device=queries.device)
    _lib.check(L.spr_radius_neighbors(_ptr(queries), _ptr(q_cu), nq, _ptr(supports), _ptr(s_cu), ns,
                                      nb, float(radius), int(limit), int(algo), _ptr(out), _ptr(mc),
                                      _ptr(ws), ws.numel(), _stream(queries)), "spr_radius_neighbors")
    if not exact_width:
        return out, -1
    m = int(mc.item())
    if m == -2 and algo == 0:   # cell table too small for this geometry: exact same result, slower path
        return radius_neighbors(queries, supports, q_cu, s_cu, radius, limit, exact_width, algo=1)
    if m < 0:
        raise RuntimeError("spr_radius_neighbors: cloud extent / radius exceeds 8191 cells per axis")
    if m < 1:  # cpp_neighbors/wrapper.cpp:201-205
        raise RuntimeError("Error")
    return out[:, :min(m, limit)], m


class RadiusTable:
    """Cell table of one support set at one radius (spr_radius_table_build), queried by several neighbour
    searches.  The pyramid builds one per level: the conv search, the pool search and the previous level's
    up-sampling search share supports and radius (reference kpconv.py:352, :377, :384)."""

    def __init__(self, supports: torch.Tensor, s_cu: torch.Tensor, radius: float):
        self.supports = _dev(supports, "supports", torch.float32)
        self.s_cu = _dev(s_cu, "s_cu", torch.int32)
        self.radius = float(radius)
        self.ns, self.nb = self.supports.shape[0], self.s_cu.numel() - 1
        L = _lib.lib()
        self.blob = torch.empty((L.spr_radius_table_bytes(self.ns, self.nb),), dtype=torch.uint8, device=supports.device)
        ws = _workspace(L.spr_radius_table_build_workspace_bytes(self.ns, self.nb), supports.device)
        _lib.check(L.spr_radius_table_build(_ptr(self.supports), _ptr(self.s_cu), self.ns, self.nb, self.radius,
                                            _ptr(self.blob), self.blob.numel(), _ptr(ws), ws.numel(),
                                            _stream(self.supports)), "spr_radius_table_build")
        self._slot = 0
        self._slots = L.spr_radius_table_slots()
        self._versions = (self.supports._version, self.s_cu._version)   # the table is stale after an in-place edit

    def matches(self, supports: torch.Tensor, s_cu: torch.Tensor, radius: float) -> bool:
        """True when this table was built from exactly these tensors (same storage, shape AND content: an
        in-place update of the supports bumps their version counter) at this radius and has a result slot left."""
        return (supports.data_ptr() == self.supports.data_ptr() and supports.shape[0] == self.ns
                and s_cu.data_ptr() == self.s_cu.data_ptr() and float(radius) == self.radius
                and (self.supports._version, self.s_cu._version) == self._versions
                and supports._version == self.supports._version and s_cu._version == self.s_cu._version
                and self._slot < self._slots)

    def query(self, queries: torch.Tensor, q_cu: torch.Tensor, limit: int,
              dense: Optional[bool] = None) -> Tuple[torch.Tensor, int]:
        """int32 [Nq, min(max_count, limit)] and the untruncated max count, like radius_neighbors.
        dense: True -> one wave per query (one pass, no scratch: faster when more supports lie in range than
        `limit`), False -> one thread per query (cheaper for sparse rows), None -> the library's default.  Same rows."""
        queries = _dev(queries, "queries", torch.float32)
        q_cu = _dev(q_cu, "q_cu", torch.int32)
        nq = queries.shape[0]
        self_search = int(queries.data_ptr() == self.supports.data_ptr() and nq == self.ns
                          and q_cu.data_ptr() == self.s_cu.data_ptr())
        L = _lib.lib()
        ws = _workspace(L.spr_radius_table_query_workspace_bytes(nq), queries.device)
        out = torch.empty((nq, limit), dtype=torch.int32, device=queries.device)
        mc = torch.empty((1,), dtype=torch.int32, device=queries.device)
        slot, self._slot = self._slot, self._slot + 1
        if dense is None:
            _lib.check(L.spr_radius_table_query(_ptr(queries), _ptr(q_cu), nq, self_search, self.ns, self.nb, self.radius,
                                                int(limit), slot, _ptr(self.blob), _ptr(out), _ptr(mc), _ptr(ws),
                                                ws.numel(), _stream(queries)), "spr_radius_table_query")
        else:
            _lib.check(L.spr_radius_table_query_a(_ptr(queries), _ptr(q_cu), nq, self_search, self.ns, self.nb,
                                                  self.radius, int(limit), slot, _ptr(self.blob), _ptr(out), _ptr(mc),
                                                  1 if dense else 0, _ptr(ws), ws.numel(), _stream(queries)),
                       "spr_radius_table_query_a")
        m = int(mc.item())
        if m == -2:     # cell table too small for this geometry: exact same result, slower path
            return radius_neighbors(queries, self.supports, q_cu, self.s_cu, self.radius, limit, True, algo=1)
        if m < 0:
            raise RuntimeError("spr_radius_neighbors: cloud extent / radius exceeds 8191 cells per axis")
        if m < 1:  # cpp_neighbors/wrapper.cpp:201-205
            raise RuntimeError("Error")
        return out[:, :min(m, limit)], m


def _wants_grad(*tensors) -> bool:
    """True when the call must go through autograd.py (gradients enabled and asked for)."""
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def kpconv(q_pts, s_pts, nbr, x, weights, kernel_points, kp_extent: float, rows_sorted: bool = False,
           impl: int = 0) -> torch.Tensor:
    """a4.  nbr may be int32 or int64 [Nq, K] (possibly a column slice).  Differentiable in x and
    weights (autograd.KPConvFn)."""
    if _wants_grad(x, weights):
        from .autograd import KPConvFn
        return KPConvFn.apply(_dev(q_pts, "q_pts", torch.float32), _dev(s_pts, "s_pts", torch.float32), nbr,
                              _dev(x, "x", torch.float32), weights, kernel_points, float(kp_extent),
                              bool(rows_sorted), int(impl))
    return kpconv_raw(q_pts, s_pts, nbr, x, weights, kernel_points, kp_extent, rows_sorted, impl)


_KP_IMPL_AB = os.environ.get("SPR_KPCONV_IMPL")   # experiment switch (A/B timing): "2" = round-2 streamed kernel


def kpconv_raw(q_pts, s_pts, nbr, x, weights, kernel_points, kp_extent: float, rows_sorted: bool = False,
               impl: int = 0, order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """order: optional int32 permutation of the queries (the tile walk of the ring kernel, e.g. a
    spatial order so that the workgroups of an XCD share neighbour rows in its L2); the output is
    bitwise independent of it."""
    if _KP_IMPL_AB is not None and impl == 0:
        impl = int(_KP_IMPL_AB)
    q_pts = _dev(q_pts, "q_pts", torch.float32)
    s_pts = _dev(s_pts, "s_pts", torch.float32)
    x = _dev(x, "x", torch.float32)
    weights = _dev(weights, "weights", torch.float32)
    kernel_points = _dev(kernel_points, "kernel_points", torch.float32)
    if not nbr.is_cuda:
        raise RuntimeError("neighb_inds must be on the device")
    if nbr.dtype != torch.int32:
        nbr = nbr.to(torch.int32)
    stride = nbr.stride(0) if nbr.stride(1) == 1 and nbr.shape[1] > 0 else None
    if stride is None:
        nbr = nbr.contiguous()
        stride = nbr.shape[1]
    nq, ns, kmax = q_pts.shape[0], s_pts.shape[0], nbr.shape[1]
    n_kp, cin, cout = weights.shape
    assert x.shape == (ns, cin), (x.shape, ns, cin)
    L = _lib.lib()
    ws = _workspace(L.spr_kpconv_workspace_bytes(nq, ns, cin, cout), x.device)
    out = torch.empty((nq, cout), dtype=torch.float32, device=x.device)
    xr, xr_n = _get_range(x)
    wr, wr_n = _static_range(weights)
    plan = wplanes = None
    if impl == 0 and n_kp == 15 and cin % 32 == 0 and cout % 32 == 0 and cout <= 256:
        if cin in (32, 64) and cin * cout <= 4096 and kmax <= 128:
            plan = _kpconv_plan(nbr, nq, ns, int(stride), kmax, bool(rows_sorted), order)
        if wr is not None:
            wplanes = _kpconv_wplanes(weights, wr, wr_n)
    _lib.check(L.spr_kpconv_fwd_p(_ptr(q_pts), nq, _ptr(s_pts), ns, _ptr(nbr), int(stride), kmax,
                                  int(bool(rows_sorted)), _ptr(x), cin, _ptr(weights), cout,
                                  _ptr(kernel_points), n_kp, float(kp_extent), _ptr(out), int(impl),
                                  _ptr(xr), int(xr_n), _ptr(wr), int(wr_n), _ptr(plan), _ptr(wplanes),
                                  _ptr(ws), ws.numel(), _stream(x)), "spr_kpconv_fwd_p")
    return out


def _kpconv_plan(nbr: torch.Tensor, nq: int, ns: int, stride: int, kmax: int, rows_sorted: bool,
                 order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Tile descriptors of a neighbour matrix for the ring KPConv (spr_kpconv_plan), cached on the
    index tensor: the pyramid hands the same int32 tensor to every block of a level, and training
    re-uses it for the recomputation in the backward.  Keyed like a published range (storage pointer
    + version counter), with the arguments the plan depends on; guarded for readers on other streams."""
    if order is not None:
        order = _dev(order, "order", torch.int32)
        assert order.shape == (nq,)
    key = (nbr.data_ptr(), nbr._version, nq, ns, stride, kmax, rows_sorted, _range_epoch[0],
           None if order is None else (order.data_ptr(), order._version))
    c = getattr(nbr, '_spr_kp_plan', None)
    if c is not None and c[0] == key:
        c[2].acquire()
        return c[1]
    L = _lib.lib()
    nbytes = L.spr_kpconv_plan_bytes(nq)
    plan = torch.empty((nbytes,), dtype=torch.uint8, device=nbr.device)
    _lib.check(L.spr_kpconv_plan(_ptr(nbr), nq, ns, stride, kmax, int(rows_sorted), _ptr(order), _ptr(plan), nbytes,
                                 _stream(nbr)), "spr_kpconv_plan")
    nbr._spr_kp_plan = (key, plan, _StreamGuard(plan))
    return plan


def kpconv_plan_prefetch(nbr: torch.Tensor, ns: int, rows_sorted: bool = True) -> None:
    """Builds (and caches on the tensor) the ring KPConv's tile plan of a neighbour matrix on the CURRENT
    stream: the pyramid builder calls it right behind each radius search on its side stream, so the plan
    kernel runs beside the encoder instead of in front of the first convolution that uses the matrix."""
    if nbr is None or not nbr.is_cuda or nbr.dtype != torch.int32 or nbr.dim() != 2 or nbr.shape[1] < 1:
        return
    if nbr.stride(1) != 1 or nbr.shape[1] > 128 or nbr.shape[0] < 1:
        return
    _kpconv_plan(nbr, nbr.shape[0], int(ns), int(nbr.stride(0)), nbr.shape[1], bool(rows_sorted))


def _kpconv_wplanes(weights: torch.Tensor, wr: torch.Tensor, wr_n: int) -> torch.Tensor:
    """Split-fp16 fragment-order planes of a KPConv weight tensor (spr_kpconv_prep_weights), cached per
    weight version next to its range."""
    key = (weights.data_ptr(), weights._version, wr.data_ptr(), _range_epoch[0])
    c = getattr(weights, '_spr_kp_wplanes', None)
    if c is not None and c[0] == key:
        c[2].acquire()
        return c[1]
    L = _lib.lib()
    n_kp, cin, cout = weights.shape
    nbytes = L.spr_kpconv_wplanes_bytes(cin, cout)
    planes = torch.empty((nbytes,), dtype=torch.uint8, device=weights.device)
    _lib.check(L.spr_kpconv_prep_weights(_ptr(weights), n_kp, cin, cout, _ptr(wr), int(wr_n), _ptr(planes), nbytes,
                                         _stream(weights)), "spr_kpconv_prep_weights")
    weights._spr_kp_wplanes = (key, planes, _StreamGuard(planes))
    return planes


def instnorm(x, cu, eps: float = 1e-5, norm: bool = True, add=None, slope: float = 1.0,
             out=None, max_len: Optional[int] = None) -> torch.Tensor:
    """a5.  out = lrelu(InstanceNorm_per_cloud(x) + add, slope).  max_len: host
    upper bound of the longest cloud (defaults to n: correct, just a larger
    statistics grid).  Differentiable in x and add (autograd.InstNormFn)."""
    if _wants_grad(x, add):
        from .autograd import InstNormFn
        return InstNormFn.apply(_dev(x, "x", torch.float32), _dev(cu, "cu", torch.int32), float(eps), bool(norm),
                                None if add is None else _dev(add, "add", torch.float32), float(slope), max_len)
    return instnorm_raw(x, cu, eps, norm, add, slope, out, max_len)


def instnorm_raw(x, cu, eps: float = 1e-5, norm: bool = True, add=None, slope: float = 1.0,
                 out=None, max_len: Optional[int] = None) -> torch.Tensor:
    x = _dev(x, "x", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    n, c = x.shape
    nb = cu.numel() - 1
    if add is not None:
        add = _dev(add, "add", torch.float32)
        assert add.shape == x.shape
    if out is None:
        out = torch.empty_like(x)
    L = _lib.lib()
    max_len = n if max_len is None else max(1, min(int(max_len), n))
    ws = _workspace(L.spr_instnorm_workspace_bytes(max_len, nb, c), x.device)
    cnt = _STREAM_SLOTS
    rng = _zero_slots(cnt, x.device) if _HANDOVER else None
    _lib.check(L.spr_instnorm_r(_ptr(x), _ptr(cu), n, nb, max_len, c, float(eps), int(bool(norm)), _ptr(add),
                                float(slope), _ptr(out), _ptr(rng), cnt, _ptr(ws), ws.numel(), _stream(x)),
               "spr_instnorm_r")
    if rng is not None:
        _set_range(out, rng, cnt)
    return out


_BLOCK_TAIL = os.environ.get("SPR_NO_BLOCK_TAIL", "0") != "1"   # experiment switch (A/B timing)


def block_tail_tile_rows(ka: int, kb: int, n_out: int) -> int:
    """Rows per statistics tile of the fused block tail for this shape; 0 = no kernel (use the separate
    operators).  Also 0 outside the split-fp16 product mode and under SPR_NO_BLOCK_TAIL=1."""
    if not _BLOCK_TAIL or _modes["gemm"] != 1:
        return 0
    return int(_lib.lib().spr_block_tail_tile_rows(int(ka), int(kb), int(n_out)))


def _tail_tiles(cu: torch.Tensor, n: int, tr: int) -> torch.Tensor:
    """tile table of spr_block_tail for a cu_seqlens tensor, cached on it (one per pyramid level and tile
    height; the pyramid hands the same cu tensor to every block of a level) and guarded for readers on
    other streams like the KPConv plans."""
    key = (cu.data_ptr(), cu._version, n, tr, _range_epoch[0])
    cache = getattr(cu, '_spr_tail_tiles', None)
    if cache is None:
        cache = {}
        cu._spr_tail_tiles = cache
    c = cache.get(tr)
    if c is not None and c[0] == key:
        c[2].acquire()
        return c[1]
    L = _lib.lib()
    t = torch.empty((L.spr_block_tail_tiles_len(n, cu.numel() - 1, tr),), dtype=torch.int32, device=cu.device)
    _lib.check(L.spr_block_tail_tiles(_ptr(cu), n, cu.numel() - 1, tr, _ptr(t), _stream(cu)), "spr_block_tail_tiles")
    cache[tr] = (key, t, _StreamGuard(t))
    return t


def instnorm_stats(x, cu, eps: float = 1e-5, max_len: Optional[int] = None):
    """(mean, rstd), each [nb, c]: the statistics passes of `instnorm` alone, for a consumer that normalises on
    load (block_tail(..., xa_stats=...))."""
    x = _dev(x, "x", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    n, c = x.shape
    nb = cu.numel() - 1
    L = _lib.lib()
    max_len = n if max_len is None else max(1, min(int(max_len), n))
    ws = _workspace(L.spr_instnorm_workspace_bytes(max_len, nb, c), x.device)
    st = torch.empty((2, nb, c), dtype=torch.float32, device=x.device)
    _lib.check(L.spr_instnorm_stats(_ptr(x), _ptr(cu), n, nb, max_len, c, float(eps), _ptr(st[0]), _ptr(st[1]),
                                    _ptr(ws), ws.numel(), _stream(x)), "spr_instnorm_stats")
    return st[0], st[1]


_NORM_BOUNDS = {}


def _norm_bound(max_len: int, device) -> torch.Tensor:
    """One-slot operand range holding sqrt(max_len): |x - mean| <= sqrt(n - 1) sigma for any n values, so an
    instance-normalised (and LeakyReLU'd, slope <= 1) column of a cloud of at most max_len points is bounded by it."""
    key = (int(max_len), str(device))
    t = _NORM_BOUNDS.get(key)
    if t is None:
        t = torch.full((1,), float(max(1, int(max_len))) ** 0.5, dtype=torch.float32, device=device)
        _NORM_BOUNDS[key] = t
    return t


def block_tail(xa, wa, cu, xb=None, wb=None, add=None, eps: float = 1e-5, slope: float = 0.1,
               xa_stats=None, xa_slope: float = 0.1, xa_max_len: Optional[int] = None) -> torch.Tensor:
    """a5, inference: lrelu(IN(xa wa^T) + (IN(xb wb^T) | add), slope) without the un-normalised
    projections ever being written (spr_block_tail).  The caller checks block_tail_tile_rows first.
    xa_stats = (mean, rstd) of instnorm_stats(xa): xa is then the RAW input of a per-cloud InstanceNorm +
    LeakyReLU(xa_slope) that runs while the tiles are staged (spr_block_tail_n) -- the bottleneck block's norm
    behind its KPConv costs no pass of its own."""
    xa = _dev(xa, "xa", torch.float32)
    wa = _dev(wa, "wa", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    n, ka = xa.shape
    n_out = wa.shape[0]
    assert wa.shape[1] == ka
    kb = 0
    if xb is not None:
        xb = _dev(xb, "xb", torch.float32)
        wb = _dev(wb, "wb", torch.float32)
        kb = xb.shape[1]
        assert xb.shape[0] == n and wb.shape == (n_out, kb) and add is None
    if add is not None:
        add = _dev(add, "add", torch.float32)
        assert add.shape == (n, n_out)
    nb = cu.numel() - 1
    L = _lib.lib()
    tr = L.spr_block_tail_tile_rows(ka, kb, n_out)
    if tr <= 0:
        raise RuntimeError(f"block_tail: no kernel for ka={ka} kb={kb} n_out={n_out}")
    tiles = _tail_tiles(cu, n, tr)
    out = torch.empty((n, n_out), dtype=torch.float32, device=xa.device)
    ws = _workspace(L.spr_block_tail_workspace_bytes(n, nb, kb, n_out, tr), xa.device)
    if xa_stats is not None:
        xar, xar_n = _norm_bound(n if xa_max_len is None else xa_max_len, xa.device), 1
    else:
        xar, xar_n = _get_range(xa)
    war, war_n = _static_range(wa)
    xbr, xbr_n, wbr, wbr_n = None, 0, None, 0
    if kb > 0:
        xbr, xbr_n = _get_range(xb)
        wbr, wbr_n = _static_range(wb)
    cnt = _STREAM_SLOTS
    rng = _zero_slots(cnt, xa.device) if _HANDOVER else None
    if xa_stats is not None:
        mean, rstd = xa_stats
        assert mean.shape == (nb, ka) and rstd.shape == (nb, ka) and mean.is_contiguous() and rstd.is_contiguous()
        _lib.check(L.spr_block_tail_n(_ptr(xa), ka, _ptr(wa), _ptr(xb), kb, _ptr(wb), _ptr(add), _ptr(cu), _ptr(tiles),
                                      n, nb, n_out, float(eps), float(slope), _ptr(out), _ptr(xar), int(xar_n),
                                      _ptr(war), int(war_n), _ptr(xbr), int(xbr_n), _ptr(wbr), int(wbr_n),
                                      _ptr(rng), cnt, _ptr(mean), _ptr(rstd), float(xa_slope), _ptr(ws), ws.numel(),
                                      _stream(xa)), "spr_block_tail_n")
    else:
        _lib.check(L.spr_block_tail(_ptr(xa), ka, _ptr(wa), _ptr(xb), kb, _ptr(wb), _ptr(add), _ptr(cu), _ptr(tiles),
                                    n, nb, n_out, float(eps), float(slope), _ptr(out), _ptr(xar), int(xar_n),
                                    _ptr(war), int(war_n), _ptr(xbr), int(xbr_n), _ptr(wbr), int(wbr_n),
                                    _ptr(rng), cnt, _ptr(ws), ws.numel(), _stream(xa)), "spr_block_tail")
    if rng is not None:
        _set_range(out, rng, cnt)
    return out


def cell_order(points: torch.Tensor, cu: torch.Tensor, cell: float) -> torch.Tensor:
    """int32 [N]: the points of every cloud sorted by the Morton code of their `cell`-sized grid cell -- a spatial WALK
    order for gather operators (maxpool(order=...)); results never depend on it."""
    points = _dev(points, "points", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    n, nb = points.shape[0], cu.numel() - 1
    L = _lib.lib()
    out = torch.empty((n,), dtype=torch.int32, device=points.device)
    ws = _workspace(L.spr_cell_order_workspace_bytes(n), points.device)
    _lib.check(L.spr_cell_order(_ptr(points), _ptr(cu), n, nb, float(cell), _ptr(out), _ptr(ws), ws.numel(),
                                _stream(points)), "spr_cell_order")
    return out


def maxpool(x, idx, order=None) -> torch.Tensor:
    if _wants_grad(x):
        from .autograd import MaxPoolFn
        return MaxPoolFn.apply(_dev(x, "x", torch.float32), idx)
    return maxpool_raw(x, idx, order)


def maxpool_raw(x, idx, order=None) -> torch.Tensor:
    x = _dev(x, "x", torch.float32)
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    stride = idx.stride(0) if idx.stride(1) == 1 else None
    if stride is None:
        idx = idx.contiguous()
        stride = idx.shape[1]
    ns, c = x.shape
    nq, k = idx.shape
    out = torch.empty((nq, c), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    cnt = _STREAM_SLOTS
    rng = _zero_slots(cnt, x.device) if _HANDOVER else None
    if order is not None:
        if order.dtype != torch.int32 or order.numel() != nq or not order.is_contiguous():
            raise ValueError("maxpool: order must be a contiguous int32 permutation of the query rows")
        _lib.check(L.spr_maxpool_gather_o(_ptr(x), ns, c, _ptr(idx), nq, int(stride), k, _ptr(order), _ptr(out),
                                          _ptr(rng), cnt, _stream(x)), "spr_maxpool_gather_o")
    else:
        _lib.check(L.spr_maxpool_gather_r(_ptr(x), ns, c, _ptr(idx), nq, int(stride), k, _ptr(out), _ptr(rng), cnt,
                                          _stream(x)), "spr_maxpool_gather_r")
    if rng is not None:
        _set_range(out, rng, cnt)
    return out


def linear(x, weight, bias=None, residual=None, act: int = ACT_NONE) -> torch.Tensor:
    """act(x W^T + bias + residual).  Differentiable in all four tensors (autograd.LinearFn)."""
    if _wants_grad(x, weight, bias, residual):
        from .autograd import LinearFn
        return LinearFn.apply(_dev(x, "x", torch.float32), weight, bias, residual, int(act))
    return linear_raw(x, weight, bias, residual, act)


def linear_raw(x, weight, bias=None, residual=None, act: int = ACT_NONE) -> torch.Tensor:
    x = _dev(x, "x", torch.float32)
    weight = _dev(weight, "weight", torch.float32)
    m, k = x.shape
    n = weight.shape[0]
    assert weight.shape[1] == k
    if bias is not None:
        bias = _dev(bias, "bias", torch.float32)
    if residual is not None:
        residual = _dev(residual, "residual", torch.float32)
        assert residual.shape == (m, n)
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _workspace(L.spr_linear_workspace_bytes(), x.device)
    xr, xr_n = _get_range(x)
    wr, wr_n = _static_range(weight)
    # outputs that go on into another GEMM (no residual: FFN hidden, projections) publish their range
    orng = torch.empty((_RANGE_CAP,), dtype=torch.float32, device=x.device) if residual is None else None
    on = ctypes.c_int(0)
    _lib.check(L.spr_linear_r(_ptr(x), m, k, _ptr(weight), n, _ptr(bias), _ptr(residual), int(act), _ptr(out),
                              _ptr(xr), int(xr_n), _ptr(wr), int(wr_n), _ptr(orng),
                              _RANGE_CAP if orng is not None else 0, ctypes.byref(on),
                              _ptr(ws), ws.numel(), _stream(x)), "spr_linear_r")
    if orng is not None:
        _set_range(out, orng, on.value)
    return out


_modes = {"gemm": 1, "attn": 1}   # host mirror of the library's arithmetic switches


def set_gemm_mode(mode: int) -> None:
    """1 = split-fp16 MFMA (default), 0 = exact f32 MFMA."""
    _lib.check(_lib.lib().spr_set_gemm_mode(int(mode)), "spr_set_gemm_mode")
    _modes["gemm"] = int(mode)


def layernorm(x, gamma, beta, eps: float = 1e-5, pos=None, want_norm: bool = True):
    """Returns (LN(x) or None, LN(x)+pos or None).  Differentiable in x, gamma, beta
    (autograd.LayerNormFn)."""
    if _wants_grad(x, gamma, beta):
        from .autograd import LayerNormFn
        n, p = LayerNormFn.apply(_dev(x, "x", torch.float32), gamma, beta, float(eps), pos, bool(want_norm))
        return (n if (want_norm or pos is None) else None), (p if pos is not None else None)
    return layernorm_raw(x, gamma, beta, eps, pos, want_norm)


def layernorm_raw(x, gamma, beta, eps: float = 1e-5, pos=None, want_norm: bool = True):
    x = _dev(x, "x", torch.float32)
    m, c = x.shape
    out_norm = torch.empty_like(x) if want_norm else None
    out_pos = None
    if pos is not None:
        pos = _dev(pos, "pos", torch.float32)
        out_pos = torch.empty_like(x)
    L = _lib.lib()
    cnt = L.spr_layernorm_range_count(m)
    rn = _zero_slots(cnt, x.device) if out_norm is not None else None
    rp = _zero_slots(cnt, x.device) if out_pos is not None else None
    _lib.check(L.spr_layernorm_r(_ptr(x), m, c, _ptr(_dev(gamma, "gamma", torch.float32)),
                                 _ptr(_dev(beta, "beta", torch.float32)), float(eps), _ptr(pos),
                                 _ptr(out_norm), _ptr(out_pos), _ptr(rn), _ptr(rp), _stream(x)), "spr_layernorm_r")
    if out_norm is not None:
        _set_range(out_norm, rn, cnt)
    if out_pos is not None:
        _set_range(out_pos, rp, cnt)
    return out_norm, out_pos


def posemb_sine(xyz, d_model: int, scale: float = 1.0, temperature: float = 10000.0) -> torch.Tensor:
    xyz = _dev(xyz, "xyz", torch.float32)
    n = xyz.shape[0]
    out = torch.empty((n, d_model), dtype=torch.float32, device=xyz.device)
    _lib.check(_lib.lib().spr_posemb_sine(_ptr(xyz), n, d_model, float(scale * 2 * math.pi),
                                          float(temperature), _ptr(out), _stream(xyz)),
               "spr_posemb_sine")
    return out


def attention(q, k, v, cu, kv_seg, max_len: int, nhead: int, out=None, lens_host=None,
              kv_seg_host=None) -> torch.Tensor:
    """a9.  q,k,v: [T, nhead*32] views (row stride may exceed the width, e.g.
    slices of a fused [T, 3*d] projection).  Differentiable in q, k, v (autograd.AttentionFn;
    needs the host-side segment lengths / kv map, read back from the device if not given)."""
    if _wants_grad(q, k, v):
        from .autograd import AttentionFn
        if lens_host is None:
            c = cu.cpu().tolist()
            lens_host = [c[i + 1] - c[i] for i in range(len(c) - 1)]
        if kv_seg_host is None:
            kv_seg_host = kv_seg.cpu().tolist()
        return AttentionFn.apply(q, k, v, _dev(cu, "cu", torch.int32), _dev(kv_seg, "kv_seg", torch.int32),
                                 int(max_len), int(nhead), list(lens_host), list(kv_seg_host))
    return attention_raw(q, k, v, cu, kv_seg, max_len, nhead, out)


def attention_raw(q, k, v, cu, kv_seg, max_len: int, nhead: int, out=None, want_lse: bool = False):
    """want_lse: returns (out, lse) with lse [T, nhead] = the per-query log2-sum-exp the backward can reuse
    (spr_attn_varlen_fwd_lse), or (out, None) when the configured core does not produce it."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(1) != 1:
            raise RuntimeError(f"attention: {nm} must be a float32 device tensor with unit inner stride")
    T, d = q.shape
    hd = d // nhead
    cu = _dev(cu, "cu", torch.int32)
    kv_seg = _dev(kv_seg, "kv_seg", torch.int32)
    nseg = cu.numel() - 1
    if out is None:
        out = torch.empty((T, d), dtype=torch.float32, device=q.device)
    L = _lib.lib()
    ws = _workspace(L.spr_attn_workspace_bytes(T, nseg, nhead, hd), q.device)
    if want_lse:
        lse = torch.empty((T, nhead), dtype=torch.float32, device=q.device)
        written = ctypes.c_int(0)
        _lib.check(L.spr_attn_varlen_fwd_lse(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0),
                                             _ptr(cu), _ptr(kv_seg), T, nseg, int(max_len), nhead, hd,
                                             1.0 / math.sqrt(hd), _ptr(out), out.stride(0), _ptr(lse),
                                             ctypes.byref(written), _ptr(ws), ws.numel(), _stream(q)),
                   "spr_attn_varlen_fwd_lse")
        return out, (lse if written.value else None)
    _lib.check(L.spr_attn_varlen_fwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0),
                                     _ptr(cu), _ptr(kv_seg), T, nseg, int(max_len), nhead, hd,
                                     1.0 / math.sqrt(hd), _ptr(out), out.stride(0), _ptr(ws), ws.numel(),
                                     _stream(q)), "spr_attn_varlen_fwd")
    return out


_seg_perm_cache = {}


def _segment_perms(kv_seg_host, device):
    """(kv_seg, inverse) as int32 device tensors for a host-side permutation (cached: a model uses two --
    the identity for self attention and the src <-> tgt swap for cross attention)."""
    key = (tuple(int(x) for x in kv_seg_host), str(device))
    c = _seg_perm_cache.get(key)
    if c is None:
        kv = np.asarray(key[0], dtype=np.int32)
        inv = np.empty_like(kv)
        inv[kv] = np.arange(kv.size, dtype=np.int32)
        c = (torch.from_numpy(kv).to(device), torch.from_numpy(inv).to(device))
        _seg_perm_cache[key] = c
    return c


def attention_bwd(q, k, v, out, dout, cu, kv_seg_host, max_len: int, nhead: int, lse=None):
    """Gradients (dq, dk, dv) of attention_raw (spr_attn_varlen_bwd: flash-style; the arithmetic follows
    set_attn_mode: split-fp16 like the forward, or exact f32 MFMA in mode 0)."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout")):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(1) != 1:
            raise RuntimeError(f"attention_bwd: {nm} must be a float32 device tensor with unit inner stride")
    T, d = q.shape
    hd = d // nhead
    cu = _dev(cu, "cu", torch.int32)
    nseg = cu.numel() - 1
    kv, inv = _segment_perms(kv_seg_host, q.device)
    dq = torch.empty((T, d), dtype=torch.float32, device=q.device)
    dk = torch.empty_like(dq)
    dv = torch.empty_like(dq)
    L = _lib.lib()
    ws = _workspace(L.spr_attn_bwd_workspace_bytes2(T, nseg, nhead), q.device)      # with room for the operand planes
    _lib.check(L.spr_attn_varlen_bwd_lse(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(out),
                                         out.stride(0), _ptr(dout), dout.stride(0), _ptr(lse), _ptr(cu), _ptr(kv),
                                         _ptr(inv), T, nseg, int(max_len), nhead, hd, 1.0 / math.sqrt(hd), _ptr(dq),
                                         _ptr(dk), _ptr(dv), _ptr(ws), ws.numel(), _stream(q)), "spr_attn_varlen_bwd_lse")
    return dq, dk, dv


def inproj_prepare(w_in: torch.Tensor):
    """Weight-side inputs of the fused in-projection (max |w| partials + row L1 norms), measured
    once per weight version and cached on the tensor like _static_range."""
    if not _HANDOVER:
        return None
    r = getattr(w_in, '_spr_inproj', None)
    if r is not None and r[1] == w_in._version and r[2] == w_in.data_ptr() and r[3] == _range_epoch[0]:
        r[4].acquire()
        return r[0]
    L = _lib.lib()
    d = w_in.shape[1]
    buf = torch.empty((L.spr_range_parts() + 3 * d,), dtype=torch.float32, device=w_in.device)
    _lib.check(L.spr_attn_inproj_prepare(_ptr(w_in), d, _ptr(buf), _stream(w_in)), "spr_attn_inproj_prepare")
    w_in._spr_inproj = (buf, w_in._version, w_in.data_ptr(), _range_epoch[0], _StreamGuard(buf))
    return buf


def attention_inproj(x_qk, x_v, w_in, b_in, cu, kv_seg, max_len: int, nhead: int, w_prep=None) -> torch.Tensor:
    """In-projection (packed [3d, d] weight, q/k from x_qk, v from x_v) + attention core.
    w_prep: inproj_prepare(weight) of the SAME weight (callers that hold the parameter object
    pass it so that the measurement is cached across calls)."""
    x_qk = _dev(x_qk, "x_qk", torch.float32)
    x_v = x_qk if x_v is x_qk else _dev(x_v, "x_v", torch.float32)
    w_in = _dev(w_in, "w_in", torch.float32)
    b_in = _dev(b_in, "b_in", torch.float32)
    T, d = x_qk.shape
    assert x_v.shape == (T, d) and w_in.shape == (3 * d, d) and b_in.shape == (3 * d,)
    hd = d // nhead
    cu = _dev(cu, "cu", torch.int32)
    kv_seg = _dev(kv_seg, "kv_seg", torch.int32)
    nseg = cu.numel() - 1
    out = torch.empty((T, d), dtype=torch.float32, device=x_qk.device)
    L = _lib.lib()
    ws = _workspace(L.spr_attn_inproj_workspace_bytes(T, nseg, nhead, hd), x_qk.device)
    fused = _modes["attn"] != 0 and _modes["gemm"] == 1 and T >= 256     # the route that can take / publish ranges
    qr, qn = _get_range(x_qk) if fused else (None, 0)
    vr, vn = (qr, qn) if x_v is x_qk else (_get_range(x_v) if fused else (None, 0))
    orng = torch.empty((1,), dtype=torch.float32, device=x_qk.device) if fused else None
    _lib.check(L.spr_attn_inproj_varlen_fwd_r(_ptr(x_qk), _ptr(x_v), T, _ptr(w_in), _ptr(b_in), _ptr(cu),
                                              _ptr(kv_seg), nseg, int(max_len), nhead, hd, 1.0 / math.sqrt(hd),
                                              _ptr(out), out.stride(0), _ptr(qr), int(qn), _ptr(vr), int(vn),
                                              _ptr(orng), _ptr(w_prep if fused else None), _ptr(ws), ws.numel(),
                                              _stream(x_qk)),
               "spr_attn_inproj_varlen_fwd_r")
    if orng is not None:
        _set_range(out, orng, 1)
    return out


# ---- fused cross-encoder stack (csrc/xenc.hip) ---------------------------------------------------
_XENC_ON = os.environ.get("SPR_NO_XENC", "0") != "1"   # experiment switch (A/B against the per-operator route)
_xenc_lock = threading.Lock()
XENC_PTRS_PER_LAYER = 18                                # SPR_XENC_PTRS_PER_LAYER of include/spr.h


class XencPlan:
    """Prepared weights of a cross-encoder stack: split-fp16 fragment streams on the device + the
    host-side plan (spr_xenc_prepare).  Valid while the parameters it was built from are untouched
    (storage pointers and version counters are part of `key`)."""

    def __init__(self, key, prepared, plan, keep):
        self.key, self.prepared, self.plan, self.keep = key, prepared, plan, keep

    # The host plan holds device addresses inside `prepared`: a copy must not outlive the original.  A copied or
    # unpickled module simply has no plan and builds its own on first use (copy.deepcopy(model), torch.save(model)).
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_plan, ())


def _no_plan():
    return None


def xenc_available() -> bool:
    return _XENC_ON and _modes["gemm"] == 1 and _modes["attn"] in (1, 2, 3, 4)


def xenc_prepare(layer_params, layer_eps, final, nhead: int, d_ff: int, pos_bound: float, cached=None) -> XencPlan:
    """layer_params: per layer the 18 parameter tensors in the order of include/spr.h
    (SPR_XENC_PTRS_PER_LAYER); layer_eps: per layer (eps1, eps2, eps3); final: (weight, bias, eps) of
    the stack's last LayerNorm or None.  `cached`: a previous XencPlan, returned as is when nothing changed."""
    flat = [t for lp in layer_params for t in lp] + ([final[0], final[1]] if final is not None else [])
    key = (tuple((t.data_ptr(), t._version) for t in flat), tuple(float(e) for le in layer_eps for e in le),
           None if final is None else float(final[2]), int(nhead), int(d_ff), float(pos_bound), _range_epoch[0])
    if cached is not None and cached.key == key:
        return cached
    with _xenc_lock:
        L = _lib.lib()
        n_layers = len(layer_params)
        keep = [_dev(t.detach(), "parameter", torch.float32) for t in flat]
        per = XENC_PTRS_PER_LAYER
        assert all(len(lp) == per for lp in layer_params)
        ptrs = (ctypes.c_void_p * (n_layers * per))(*[t.data_ptr() for t in keep[:n_layers * per]])
        eps = (ctypes.c_float * (3 * n_layers))(*[float(e) for le in layer_eps for e in le])
        dev = keep[0].device
        nbytes = L.spr_xenc_prepared_bytes(n_layers, int(d_ff))
        if nbytes == 0:
            raise RuntimeError("xenc_prepare: unsupported stack shape")
        prepared = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        plan = ctypes.create_string_buffer(L.spr_xenc_plan_bytes())
        fg = keep[-2] if final is not None else None
        fb = keep[-1] if final is not None else None
        _lib.check(L.spr_xenc_prepare(ptrs, eps, n_layers, 256, int(nhead), int(d_ff), _ptr(fg), _ptr(fb),
                                      float(final[2]) if final is not None else 0.0, float(pos_bound),
                                      _ptr(prepared), nbytes, plan, len(plan), _stream(prepared)), "spr_xenc_prepare")
        return XencPlan(key, prepared, plan, keep)


def xenc_forward(plan: XencPlan, x, pos, cu, seg_self, seg_cross, max_len: int) -> torch.Tensor:
    x = _dev(x, "x", torch.float32)
    pos = _dev(pos, "pos", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    seg_self = _dev(seg_self, "seg_self", torch.int32)
    seg_cross = _dev(seg_cross, "seg_cross", torch.int32)
    T, d = x.shape
    assert d == 256 and pos.shape == (T, d)
    nseg = cu.numel() - 1
    out = torch.empty_like(x)
    L = _lib.lib()
    ws = _workspace(L.spr_xenc_workspace_bytes(T, nseg), x.device)
    plan.prepared.record_stream(torch.cuda.current_stream(x.device))
    _lib.check(L.spr_xenc_forward(plan.plan, _ptr(x), _ptr(pos), _ptr(cu), _ptr(seg_self), _ptr(seg_cross), T, nseg,
                                  int(max_len), _ptr(out), _ptr(ws), ws.numel(), _stream(x)), "spr_xenc_forward")
    return out


# the attention arithmetic a fresh process runs (csrc/attention.hip, g_attn_mode): split-fp16 scores, split-fp16
# probabilities where a 32-key block holds a weight of at least 2^-5 of the running row sum, one rounded plane elsewhere
DEFAULT_ATTN_MODE = 4


def set_attn_mode(mode: int) -> None:
    """1 = split-fp16 MFMA (default), 0 = exact f32 MFMA, 2 = single-pass fp16 MFMA, 3 = split-fp16 scores with ONE
    probability plane (weights rounded to 11 bits, row sum from the rounded plane), 4 = as 1 with the lo plane of the
    probabilities only on tiles that hold a weight of at least 2^-7 of the running row sum (csrc/attention.hip,
    k_attn_s; accuracy table in DESIGN.md section 4)."""
    _lib.check(_lib.lib().spr_set_attn_mode(int(mode)), "spr_set_attn_mode")
    _modes["attn"] = int(mode)


def _cu_host_arr(cu_host: Sequence[int]):
    arr = (ctypes.c_int * len(cu_host))(*[int(v) for v in cu_host])
    return arr


def match_dualsoftmax(feat, cu, cu_host: Sequence[int], npairs: int):
    """a11.  Returns (val [T] f32, ind [T] i32) -- see include/spr.h.  val is differentiable in
    feat (autograd.MatchDualSoftmaxFn)."""
    if _wants_grad(feat):
        from .autograd import MatchDualSoftmaxFn
        return MatchDualSoftmaxFn.apply(_dev(feat, "feat", torch.float32), _dev(cu, "cu", torch.int32),
                                        list(cu_host), int(npairs))
    return match_dualsoftmax_raw(feat, cu, cu_host, npairs)


def match_dualsoftmax_top2(feat, cu, cu_host: Sequence[int], npairs: int):
    """(val, val2, ind): the best and the runner-up dual-softmax value of every match (Lowe ratio
    test of RegTR.ratio_test, qk_regtr_full.py:370-384).  Inference only."""
    feat = _dev(feat, "feat", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    T, d = feat.shape
    arr = _cu_host_arr(cu_host)
    L = _lib.lib()
    ws = _workspace(L.spr_match_workspace_bytes(arr, npairs), feat.device)
    val = torch.zeros((T,), dtype=torch.float32, device=feat.device)
    val2 = torch.zeros((T,), dtype=torch.float32, device=feat.device)
    ind = torch.zeros((T,), dtype=torch.int32, device=feat.device)
    _lib.check(L.spr_match_dualsoftmax2(_ptr(feat), d, _ptr(cu), arr, npairs, _ptr(val), _ptr(val2), _ptr(ind),
                                        _ptr(ws), ws.numel(), _stream(feat)), "spr_match_dualsoftmax2")
    return val, val2, ind


def pose_residuals(pose, a, b, pair_cu) -> torch.Tensor:
    """res[i] = ||b_i - T_s a_i|| with one pose [3,4] per set of pair_cu (LGR re-weighting,
    qk_regtr_full.py:386-398)."""
    a, b = _dev(a, "a", torch.float32), _dev(b, "b", torch.float32)
    pose = _dev(pose, "pose", torch.float32)
    pair_cu = _dev(pair_cu, "pair_cu", torch.int32)
    n = a.shape[0]
    res = torch.empty((n,), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().spr_pose_residuals(_ptr(pose), _ptr(a), _ptr(b), _ptr(pair_cu), pair_cu.numel() - 1, n,
                                             _ptr(res), _stream(a)), "spr_pose_residuals")
    return res


def pose_scores(poses, a, b) -> torch.Tensor:
    """score[h] = mean_i ||b_i - T_h a_i|| for H hypotheses [H,3,4] over one point set (RANSAC
    scoring, qk_regtr_full.py:400-421)."""
    a, b = _dev(a, "a", torch.float32), _dev(b, "b", torch.float32)
    poses = _dev(poses, "poses", torch.float32)
    h = poses.shape[0]
    out = torch.empty((h,), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().spr_pose_scores(_ptr(poses), h, _ptr(a), _ptr(b), a.shape[0], _ptr(out), _stream(a)),
               "spr_pose_scores")
    return out


def match_dualsoftmax_raw(feat, cu, cu_host: Sequence[int], npairs: int):
    feat = _dev(feat, "feat", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    T, d = feat.shape
    arr = _cu_host_arr(cu_host)
    L = _lib.lib()
    ws = _workspace(L.spr_match_workspace_bytes(arr, npairs), feat.device)
    val = torch.zeros((T,), dtype=torch.float32, device=feat.device)
    ind = torch.zeros((T,), dtype=torch.int32, device=feat.device)
    _lib.check(L.spr_match_dualsoftmax(_ptr(feat), d, _ptr(cu), arr, npairs, _ptr(val), _ptr(ind),
                                       _ptr(ws), ws.numel(), _stream(feat)), "spr_match_dualsoftmax")
    return val, ind


def _dev_scalar(v, device) -> torch.Tensor:
    """A 0-d / 1-element float32 device tensor for a learnable scalar (device tensors are
    passed through without a host read; Python numbers are uploaded)."""
    if isinstance(v, torch.Tensor):
        return v.detach().to(device=device, dtype=torch.float32).reshape(1).contiguous()
    return torch.tensor([float(v)], dtype=torch.float32, device=device)


def sinkhorn_correspondences(feat, xyz, cu, cu_host: Sequence[int], npairs: int, alpha,
                             beta, n_iters: int, slack: bool = True):
    """a13.  Returns (w [Tsrc] f32, t_hat [Tsrc,3] f32) for the src tokens.  alpha / beta:
    device tensors (the model's parameters -- read on the device, no sync) or floats.
    Differentiable in feat, alpha, beta (autograd.SinkhornFn)."""
    if _wants_grad(feat, alpha, beta):
        from .autograd import SinkhornFn
        dev = feat.device
        a = alpha if isinstance(alpha, torch.Tensor) else torch.tensor(float(alpha), device=dev)
        b = beta if isinstance(beta, torch.Tensor) else torch.tensor(float(beta), device=dev)
        return SinkhornFn.apply(_dev(feat, "feat", torch.float32), _dev(xyz, "xyz", torch.float32),
                                _dev(cu, "cu", torch.int32), list(cu_host), int(npairs), a, b, int(n_iters),
                                bool(slack))
    return sinkhorn_correspondences_raw(feat, xyz, cu, cu_host, npairs, alpha, beta, n_iters, slack)


def sinkhorn_correspondences_raw(feat, xyz, cu, cu_host: Sequence[int], npairs: int, alpha,
                                 beta, n_iters: int, slack: bool = True):
    feat = _dev(feat, "feat", torch.float32)
    alpha_t, beta_t = _dev_scalar(alpha, feat.device), _dev_scalar(beta, feat.device)
    xyz = _dev(xyz, "xyz", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    T, d = feat.shape
    tsrc = int(cu_host[npairs])
    arr = _cu_host_arr(cu_host)
    L = _lib.lib()
    ws = _workspace(L.spr_sinkhorn_workspace_bytes(arr, npairs), feat.device)
    w = torch.empty((tsrc,), dtype=torch.float32, device=feat.device)
    that = torch.empty((tsrc, 3), dtype=torch.float32, device=feat.device)
    _lib.check(L.spr_sinkhorn_correspondences(_ptr(feat), d, _ptr(xyz), _ptr(cu), arr, npairs,
                                              _ptr(alpha_t), _ptr(beta_t), int(n_iters), int(bool(slack)),
                                              _ptr(w), _ptr(that), _ptr(ws), ws.numel(), _stream(feat)),
               "spr_sinkhorn_correspondences")
    return w, that


def match_and_sinkhorn(feat, xyz, cu, cu_host: Sequence[int], npairs: int, alpha, beta, n_iters: int,
                       top2: bool = False):
    """match_dualsoftmax(_top2) and sinkhorn_correspondences of the same features in one call
    (spr_match_sinkhorn: the correlation matrices are computed and stored once).  Inference only -- with gradients
    wanted use the two operators.  Returns (val, val2 or None, ind, w, t_hat), bit for bit the separate results."""
    if _wants_grad(feat, alpha, beta):
        raise RuntimeError("match_and_sinkhorn is an inference operator (use match_dualsoftmax + sinkhorn_correspondences)")
    feat = _dev(feat, "feat", torch.float32)
    alpha_t, beta_t = _dev_scalar(alpha, feat.device), _dev_scalar(beta, feat.device)
    xyz = _dev(xyz, "xyz", torch.float32)
    cu = _dev(cu, "cu", torch.int32)
    T, d = feat.shape
    tsrc = int(cu_host[npairs])
    arr = _cu_host_arr(cu_host)
    L = _lib.lib()
    ws = _workspace(L.spr_match_workspace_bytes(arr, npairs), feat.device)
    val = torch.zeros((T,), dtype=torch.float32, device=feat.device)
    val2 = torch.zeros((T,), dtype=torch.float32, device=feat.device) if top2 else None
    ind = torch.zeros((T,), dtype=torch.int32, device=feat.device)
    w = torch.empty((tsrc,), dtype=torch.float32, device=feat.device)
    that = torch.empty((tsrc, 3), dtype=torch.float32, device=feat.device)
    _lib.check(L.spr_match_sinkhorn(_ptr(feat), d, _ptr(xyz), _ptr(cu), arr, npairs, _ptr(alpha_t), _ptr(beta_t),
                                    int(n_iters), _ptr(val), _ptr(val2), _ptr(ind), _ptr(w), _ptr(that), _ptr(ws),
                                    ws.numel(), _stream(feat)), "spr_match_sinkhorn")
    return val, val2, ind, w, that


def weighted_procrustes(a, b, w, pair_cu) -> torch.Tensor:
    """a12.  a,b [T,3], w [T] or None, pair_cu int32 [P+1] -> [P,3,4].  Differentiable in a, b, w
    (autograd.ProcrustesFn)."""
    if _wants_grad(a, b, w):
        from .autograd import ProcrustesFn
        return ProcrustesFn.apply(_dev(a, "a", torch.float32), _dev(b, "b", torch.float32),
                                  None if w is None else _dev(w, "w", torch.float32), _dev(pair_cu, "pair_cu", torch.int32))
    return weighted_procrustes_raw(a, b, w, pair_cu)


def weighted_procrustes_raw(a, b, w, pair_cu) -> torch.Tensor:
    a = _dev(a, "a", torch.float32)
    b = _dev(b, "b", torch.float32)
    if w is not None:
        w = _dev(w, "w", torch.float32)
    pair_cu = _dev(pair_cu, "pair_cu", torch.int32)
    p = pair_cu.numel() - 1
    out = torch.empty((p, 3, 4), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().spr_weighted_procrustes(_ptr(a), _ptr(b), _ptr(w), _ptr(pair_cu), p,
                                                  _ptr(out), _stream(a)), "spr_weighted_procrustes")
    return out


# ---- losses (forward), SURVEY 8f row 1 -------------------------------------------
def overlap_pool(ov_prev, pool_idx, ns_prev: int):
    """One level of compute_overlaps (kpconv.py:552-578).  pool_idx int32 [Nq, W]."""
    ov_prev = _dev(ov_prev, "ov_prev", torch.float32)
    pool_idx = _dev(pool_idx, "pool_idx", torch.int32)
    nq, w = pool_idx.shape
    out = torch.empty((nq,), dtype=torch.float32, device=ov_prev.device)
    _lib.check(_lib.lib().spr_overlap_pool(_ptr(ov_prev), int(ns_prev), _ptr(pool_idx), pool_idx.stride(0), w,
                                           nq, _ptr(out), _stream(ov_prev)), "spr_overlap_pool")
    return out


def _loss_ws(n, m, d, device):
    return _workspace(_lib.lib().spr_loss_workspace_bytes(int(n), int(m), int(d)), device)


def bce_logits_mean(x, y):
    if _wants_grad(x):
        from .autograd import BCELogitsMeanFn
        return BCELogitsMeanFn.apply(_dev(x, "x", torch.float32), _dev(y, "y", torch.float32))
    return bce_logits_mean_raw(x, y)


def bce_logits_mean_raw(x, y):
    x = _dev(x, "x", torch.float32)
    y = _dev(y, "y", torch.float32)
    assert x.shape == y.shape and x.dim() == 1
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    ws = _loss_ws(x.numel(), 1, 32, x.device)
    _lib.check(_lib.lib().spr_bce_logits_mean(_ptr(x), _ptr(y), x.numel(), _ptr(out), _ptr(ws), ws.numel(),
                                              _stream(x)), "spr_bce_logits_mean")
    return out[0]


def infonce_pair(anchor_feat, positive_feat, anchor_xyz, pose_gt, positive_xyz, W, r_p: float, r_n: float):
    """InfoNCELossFull.compute_infonce for one pair; anchor_xyz is transformed by pose_gt [3,4] inside.
    Differentiable in the two feature sets and W (autograd.InfoNCEFn)."""
    if _wants_grad(anchor_feat, positive_feat, W):
        from .autograd import InfoNCEFn
        return InfoNCEFn.apply(_dev(anchor_feat, "anchor_feat", torch.float32),
                               _dev(positive_feat, "positive_feat", torch.float32), anchor_xyz, pose_gt, positive_xyz,
                               W, float(r_p), float(r_n))
    return infonce_pair_raw(anchor_feat, positive_feat, anchor_xyz, pose_gt, positive_xyz, W, r_p, r_n)


def infonce_pair_raw(anchor_feat, positive_feat, anchor_xyz, pose_gt, positive_xyz, W, r_p: float, r_n: float):
    a = _dev(anchor_feat, "anchor_feat", torch.float32)
    p = _dev(positive_feat, "positive_feat", torch.float32)
    n, d = a.shape
    m = p.shape[0]
    out = torch.empty((1,), dtype=torch.float32, device=a.device)
    ws = _loss_ws(n, m, d, a.device)
    _lib.check(_lib.lib().spr_infonce_pair(_ptr(a), n, _ptr(p), m, d, _ptr(_dev(anchor_xyz, "anchor_xyz", torch.float32)),
                                           _ptr(_dev(pose_gt, "pose_gt", torch.float32)),
                                           _ptr(_dev(positive_xyz, "positive_xyz", torch.float32)),
                                           _ptr(_dev(W, "W", torch.float32)), float(r_p), float(r_n), _ptr(out),
                                           _ptr(ws), ws.numel(), _stream(a)), "spr_infonce_pair")
    return out[0]


def transform_l1_pair(pose_gt, pose_pred, xyz):
    if _wants_grad(pose_pred):
        from .autograd import TransformL1Fn
        return TransformL1Fn.apply(_dev(pose_gt, "pose_gt", torch.float32), pose_pred, _dev(xyz, "xyz", torch.float32))
    return transform_l1_pair_raw(pose_gt, pose_pred, xyz)


def transform_l1_pair_raw(pose_gt, pose_pred, xyz):
    xyz = _dev(xyz, "xyz", torch.float32)
    out = torch.empty((1,), dtype=torch.float32, device=xyz.device)
    ws = _loss_ws(xyz.shape[0], 1, 32, xyz.device)
    _lib.check(_lib.lib().spr_transform_l1_pair(_ptr(_dev(pose_gt, "pose_gt", torch.float32)),
                                                _ptr(_dev(pose_pred, "pose_pred", torch.float32)), _ptr(xyz),
                                                xyz.shape[0], _ptr(out), _ptr(ws), ws.numel(), _stream(xyz)),
               "spr_transform_l1_pair")
    return out[0]


def sum_scaled(values, scale: float = 1.0):
    values = _dev(values, "values", torch.float32)
    out = torch.empty((1,), dtype=torch.float32, device=values.device)
    _lib.check(_lib.lib().spr_sum_scaled(_ptr(values), values.numel(), float(scale), _ptr(out), _stream(values)),
               "spr_sum_scaled")
    return out[0]


def gather_rows(x, idx) -> torch.Tensor:
    if _wants_grad(x):
        from .autograd import GatherRowsFn
        return GatherRowsFn.apply(_dev(x, "x", torch.float32), _dev(idx, "idx", torch.int32))
    return gather_rows_raw(x, idx)


def gather_rows_raw(x, idx) -> torch.Tensor:
    x = _dev(x, "x", torch.float32)
    idx = _dev(idx, "idx", torch.int32)
    n_src, c = x.shape
    n = idx.numel()
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().spr_gather_rows(_ptr(x), n_src, c, _ptr(idx), n, _ptr(out), _stream(x)),
               "spr_gather_rows")
    return out
