"""Autograd for the HIP operators -- SURVEY.md section 8f row 1.

The reference trains THROUGH the hot path with torch autograd
(generic_reg_model.py:82-84 training_step, trainer.py:107-124 backward / clip /
step).  The forward operators of this package are opaque C-ABI calls, so each
one gets an explicit backward here: a ``torch.autograd.Function`` whose
``backward`` launches HIP kernels through the same C ABI (include/spr.h,
"backward" section: spr_bgemm and friends).  torch is used for what it is in
this project -- device memory, the tape that orders the backward calls, and a
few elementwise glue steps on gradient tensors.

``ops.py`` routes a call here when gradients are enabled and an input requires
them; with ``torch.no_grad()`` (inference, bench) the plain forward wrappers run
and nothing below is touched.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from . import ops as _ops

_DESC = np.dtype([("a", "<i8"), ("b", "<i8"), ("c", "<i8"), ("m", "<i4"), ("n", "<i4"), ("k", "<i4"), ("pad", "<i4")])


def _desc(records, device):
    """Device array of BgemmDesc records (bgemm.hip) from (a_off, b_off, c_off, m, n, k) tuples."""
    arr = np.zeros(len(records), dtype=_DESC)
    for i, r in enumerate(records):
        arr[i] = (r[0], r[1], r[2], r[3], r[4], r[5], 0)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


_BGEMM_LOG = None   # experiment hook: a list collects (batches, m, n, k, strides) of every call


def bgemm(A, B, C, records, sa, sb, sc, alpha=1.0, beta=0.0):
    """C_b(i,j) = alpha sum_k A_b(i,k) B_b(k,j) + beta C_b(i,j); sa = (sa_i, sa_k), sb = (sb_k, sb_j),
    sc = (sc_i, sc_j) element strides; records = per-batch (a_off, b_off, c_off, m, n, k)."""
    d = _desc(records, A.device)
    if _BGEMM_LOG is not None:
        _BGEMM_LOG.append((len(records), records[0][3], records[0][4], records[0][5], tuple(sa), tuple(sb)))
    max_m = max(r[3] for r in records)
    max_n = max(r[4] for r in records)
    _lib.check(_lib.lib().spr_bgemm(_ops._ptr(A), _ops._ptr(B), _ops._ptr(C), _ops._ptr(d), len(records), max_m, max_n,
                                    int(sa[0]), int(sa[1]), int(sb[0]), int(sb[1]), int(sc[0]), int(sc[1]),
                                    float(alpha), float(beta), _ops._stream(A)), "spr_bgemm")
    return C


def _reduce_parts(parts, nparts, n, out):
    _lib.check(_lib.lib().spr_reduce_parts(_ops._ptr(parts), int(nparts), int(n), 1.0, _ops._ptr(out), 0,
                                           _ops._stream(parts)), "spr_reduce_parts")
    return out


def _tn_chunk(rows, nl, nr):
    """Rows of the long dimension per split-K batch: enough batches that tiles x batches fills the chip
    (a [256, 256] weight gradient is FOUR 128 x 128 tiles: with 2 048-row batches a 15 k-token product ran
    on 32 of 256 CUs), but at least 256 rows per batch.  A fixed function of the shapes, so the summation
    order -- and with it the result, bit for bit -- does not depend on anything else."""
    tiles = ((nl + 127) // 128) * ((nr + 127) // 128 if nr > 32 else 1)
    nchunk = max(1, min(rows // 256, max(1, 512 // tiles)))
    chunk = (rows + nchunk - 1) // nchunk
    return (chunk + 15) // 16 * 16       # whole K slabs


def _tn_product(L, Rm, rows, nl, nr, f64=False):
    """out[nl, nr] = L[rows, nl]^T @ R[rows, nr] (both row-major, contiguous): deterministic split
    over `rows` into batches (_tn_chunk) + fixed-order reduction -- weight gradients.  f64: float64
    accumulation (spr_tn_product_f64) for small outputs whose summands nearly cancel."""
    if f64 and nl * nr <= 65536:
        lib = _lib.lib()
        ws = _ops._workspace(lib.spr_tn_product_f64_workspace_bytes(rows, nl, nr), L.device)
        out = torch.empty((nl, nr), dtype=torch.float32, device=L.device)
        _lib.check(lib.spr_tn_product_f64(_ops._ptr(L), _ops._ptr(Rm), rows, nl, nr, _ops._ptr(out), _ops._ptr(ws),
                                          ws.numel(), _ops._stream(L)), "spr_tn_product_f64")
        return out
    chunk = _tn_chunk(rows, nl, nr)
    nchunk = (rows + chunk - 1) // chunk
    parts = torch.empty((nchunk, nl, nr), dtype=torch.float32, device=L.device)
    if _ops._modes["gemm"] == 1 and nl % 4 == 0 and nr % 4 == 0 and nl >= 32 and nr >= 32 and nchunk <= 65535:
        # the forward's arithmetic (range-scaled split-fp16 MFMA): spr_tn_product_split + the same fixed-order sum
        lib = _lib.lib()
        ws = _ops._workspace(lib.spr_tn_product_split_workspace_bytes(), L.device)
        lr, lr_n = _ops._get_range(L)
        rr, rr_n = _ops._get_range(Rm)
        _lib.check(lib.spr_tn_product_split(_ops._ptr(L), _ops._ptr(Rm), rows, nl, nr, chunk, _ops._ptr(lr), int(lr_n),
                                            _ops._ptr(rr), int(rr_n), _ops._ptr(parts), _ops._ptr(ws), ws.numel(),
                                            _ops._stream(L)), "spr_tn_product_split")
        if nchunk == 1:
            return parts[0]
        out = torch.empty((nl, nr), dtype=torch.float32, device=L.device)
        return _reduce_parts(parts, nchunk, nl * nr, out)
    recs = []
    for c in range(nchunk):
        r0 = c * chunk
        recs.append((r0 * nl, r0 * nr, c * nl * nr, nl, nr, min(chunk, rows - r0)))
    bgemm(L, Rm, parts, recs, (1, nl), (nr, 1), (nr, 1))
    if nchunk == 1:
        return parts[0]
    out = torch.empty((nl, nr), dtype=torch.float32, device=L.device)
    return _reduce_parts(parts, nchunk, nl * nr, out)


def _nn_product(A, B, n, k, d):
    """out[n, d] = A[n, k] @ B[k, d] (row-major, contiguous, exact f32).  A product whose output is a handful of
    tiles but whose contraction is long (the per-pair loss-head products: 38 tiles, k ~ 1 500) is split over k into
    batches of one launch + the fixed-order reduction of the weight gradients: the chip is filled and the K loop
    is a quarter as long.  The split is a function of the shapes only (bitwise reproducible)."""
    tiles = ((n + 127) // 128) * ((d + 127) // 128)
    nsplit = max(1, min(8, k // 256, 256 // max(tiles, 1)))
    if nsplit == 1:
        out = torch.empty((n, d), dtype=torch.float32, device=A.device)
        return bgemm(A, B, out, [(0, 0, 0, n, d, k)], (k, 1), (d, 1), (d, 1))
    chunk = ((k + nsplit - 1) // nsplit + 15) // 16 * 16
    nsplit = (k + chunk - 1) // chunk
    parts = torch.empty((nsplit, n, d), dtype=torch.float32, device=A.device)
    recs = [(c * chunk, c * chunk * d, c * n * d, n, d, min(chunk, k - c * chunk)) for c in range(nsplit)]
    bgemm(A, B, parts, recs, (k, 1), (d, 1), (d, 1))
    out = torch.empty((n, d), dtype=torch.float32, device=A.device)
    return _reduce_parts(parts, nsplit, n * d, out)


def _colsum(x):
    m, n = x.shape
    L = _lib.lib()
    ws = _ops._workspace(L.spr_colsum_workspace_bytes(n), x.device)
    out = torch.empty((n,), dtype=torch.float32, device=x.device)
    _lib.check(L.spr_colsum(_ops._ptr(x), m, n, _ops._ptr(out), _ops._ptr(ws), ws.numel(), _ops._stream(x)), "spr_colsum")
    return out


# --------------------------------------------------------------------------------------------- #
class LinearFn(torch.autograd.Function):
    """out = act(x W^T + b + residual)  (spr_linear).  dX = g W, dW = g^T X, db = colsum(g),
    d residual = g with g = dout * act'(out)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act):
        with torch.no_grad():
            if _ops._modes["gemm"] == 1 and weight.requires_grad:
                _ops.ensure_range(x)          # published once: the forward product and dW in the backward both scale x
            y = _ops.linear_raw(x, weight, bias, residual, act)
        ctx.act = act
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(x, weight, y if act != _ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        g = dy
        if ctx.act != _ops.ACT_NONE:
            g = torch.empty_like(dy)
            _lib.check(_lib.lib().spr_act_bwd(_ops._ptr(y), _ops._ptr(dy), int(ctx.act), dy.numel(), _ops._ptr(g),
                                              _ops._stream(dy)), "spr_act_bwd")
        m, k = x.shape
        n = w.shape[0]
        dx = dw = db = None
        if _ops._modes["gemm"] == 1 and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            _ops.ensure_range(g)              # dX and dW both scale g: one measuring pass
        if ctx.needs_input_grad[0]:
            if _ops._modes["gemm"] == 1 and n % 32 == 0 and k >= 16:
                # dX = g W = g (W^T)^T: the forward's NT product (range-scaled split-fp16 MFMA, the arithmetic the
                # forward itself ran in) on a transposed copy of the weight kept per weight version
                dx = _ops.linear_raw(g, _ops.weight_transposed(w))
            else:
                dx = torch.empty((m, k), dtype=torch.float32, device=x.device)
                bgemm(g, w.detach().contiguous(), dx, [(0, 0, 0, m, k, n)], (n, 1), (k, 1), (k, 1))
        if ctx.needs_input_grad[1]:
            xd = x.detach().contiguous()
            r = getattr(x, '_spr_range', None)
            if r is not None and xd is not x:
                xd._spr_range = r             # same storage and version counter: the forward's measurement still holds
            dw = _tn_product(g, xd, m, n, k)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(g)
        dres = g if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None


class KPConvFn(torch.autograd.Function):
    """spr_kpconv_fwd; backward (kpconv_blocks.py:388-412 differentiated):
       g = dout / count;  d wf = g W_flat^T;  dW_flat = wf^T g;  dx = scatter(influence * d wf)."""

    @staticmethod
    def forward(ctx, q_pts, s_pts, nbr, x, weights, kernel_points, kp_extent, rows_sorted, impl):
        with torch.no_grad():
            y = _ops.kpconv_raw(q_pts, s_pts, nbr, x, weights, kernel_points, kp_extent, rows_sorted, impl)
        ctx.kp_extent = float(kp_extent)
        ctx.save_for_backward(q_pts, s_pts, nbr, x, weights, kernel_points)
        return y

    @staticmethod
    def backward(ctx, dout):
        q_pts, s_pts, nbr, x, w, kp = ctx.saved_tensors
        dout = dout.contiguous()
        nq, ns = q_pts.shape[0], s_pts.shape[0]
        n_kp, cin, cout = w.shape
        nbr32 = nbr if nbr.dtype == torch.int32 else nbr.to(torch.int32)
        if not (nbr32.stride(1) == 1 and nbr32.shape[1] > 0):
            nbr32 = nbr32.contiguous()
        stride, kmax = nbr32.stride(0), nbr32.shape[1]
        L = _lib.lib()
        xd = x.detach().contiguous()
        wf = torch.empty((nq, n_kp * cin), dtype=torch.float32, device=x.device)
        cnt = torch.empty((nq,), dtype=torch.float32, device=x.device)
        _lib.check(L.spr_kpconv_weighted_features(_ops._ptr(q_pts), nq, _ops._ptr(s_pts), ns, _ops._ptr(nbr32), int(stride),
                                                  kmax, _ops._ptr(xd), cin, _ops._ptr(kp.detach().contiguous()), n_kp,
                                                  ctx.kp_extent, _ops._ptr(wf), _ops._ptr(cnt), _ops._stream(x)),
                   "spr_kpconv_weighted_features")
        g = (dout / cnt.unsqueeze(1)).contiguous()
        # |wf| <= kmax max|x| (every influence weight is <= 1): a bound within 2^3..2^5 of the true maximum, well
        # inside what the split arithmetic absorbs -- the 1 GB tensor is not scanned for its range
        xr, xr_n = _ops._get_range(x)
        if xr is not None:
            _ops._set_range(wf, xr[:xr_n] * float(kmax), int(xr_n))
        wflat = w.detach().contiguous().view(n_kp * cin, cout)
        dx = dw = None
        if ctx.needs_input_grad[3]:
            if _ops._modes["gemm"] == 1 and cout % 32 == 0:
                # d wf = g W_flat^T is the forward's NT product with W_flat [15 cin, cout] as the weight: the
                # range-scaled split-fp16 MFMA path (g's range is measured once, dW below reuses it)
                _ops.ensure_range(g)
                dwf = _ops.linear_raw(g, wflat)
            else:
                dwf = torch.empty((nq, n_kp * cin), dtype=torch.float32, device=x.device)
                bgemm(g, wflat, dwf, [(0, 0, 0, nq, n_kp * cin, cout)], (cout, 1), (1, cout), (n_kp * cin, 1))
            dx = torch.empty((ns, cin), dtype=torch.float32, device=x.device)
            ws = _ops._workspace(L.spr_scatter_workspace_bytes(ns, cin), x.device)
            dr, dr_n = _ops._get_range(dwf)          # published by the product that wrote dwf: no second scan of it
            _lib.check(L.spr_kpconv_bwd_dx_r(_ops._ptr(q_pts), nq, _ops._ptr(s_pts), ns, _ops._ptr(nbr32), int(stride), kmax,
                                             cin, _ops._ptr(kp.detach().contiguous()), n_kp, ctx.kp_extent,
                                             _ops._ptr(dwf), _ops._ptr(dr), int(dr_n), _ops._ptr(dx), _ops._ptr(ws),
                                             ws.numel(), _ops._stream(x)), "spr_kpconv_bwd_dx_r")
        if ctx.needs_input_grad[4]:
            # the first layer (cin = 1: constant input) is the ill-conditioned sum: float64 accumulation
            dw = _tn_product(wf, g, nq, n_kp * cin, cout, f64=(cin == 1)).view(n_kp, cin, cout)
        return None, None, None, dx, dw, None, None, None, None


class InstNormFn(torch.autograd.Function):
    """out = lrelu(InstanceNorm_per_cloud(x) + add, slope)  (spr_instnorm / spr_instnorm_bwd)."""

    @staticmethod
    def forward(ctx, x, cu, eps, norm, add, slope, max_len):
        with torch.no_grad():
            out = _ops.instnorm_raw(x, cu, eps, norm, add, slope, None, max_len)
        ctx.args = (float(eps), bool(norm), float(slope), max_len, add is not None)
        ctx.save_for_backward(x, cu, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, cu, out = ctx.saved_tensors
        eps, norm, slope, max_len, has_add = ctx.args
        dout = dout.contiguous()
        n, c = x.shape
        nb = cu.numel() - 1
        ml = n if max_len is None else max(1, min(int(max_len), n))
        L = _lib.lib()
        ws = _ops._workspace(L.spr_instnorm_bwd_workspace_bytes(ml, nb, c), x.device)
        dx = torch.empty_like(x)
        dadd = torch.empty_like(x) if (has_add and ctx.needs_input_grad[4]) else None
        _lib.check(L.spr_instnorm_bwd(_ops._ptr(x.detach()), _ops._ptr(out), _ops._ptr(dout), _ops._ptr(cu), n, nb, ml, c,
                                      eps, int(norm), slope, _ops._ptr(dx), _ops._ptr(dadd), _ops._ptr(ws), ws.numel(),
                                      _ops._stream(x)), "spr_instnorm_bwd")
        return dx, None, None, None, dadd, None, None


class LayerNormFn(torch.autograd.Function):
    """(LN(x), LN(x) + pos)  (spr_layernorm / spr_layernorm_bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pos, want_norm):
        with torch.no_grad():
            n, p = _ops.layernorm_raw(x, gamma, beta, eps, pos, want_norm)
        ctx.eps = float(eps)
        ctx.save_for_backward(x, gamma)
        ctx.has = (n is not None, p is not None)
        # an absent output is returned as an empty, non-differentiable placeholder
        if n is None:
            n = x.new_zeros((0,))
            ctx.mark_non_differentiable(n)
        if p is None:
            p = x.new_zeros((0,))
            ctx.mark_non_differentiable(p)
        return n, p

    @staticmethod
    def backward(ctx, dn, dp):
        x, gamma = ctx.saved_tensors
        m, c = x.shape
        dn = dn.contiguous() if ctx.has[0] else None
        dp = dp.contiguous() if ctx.has[1] else None
        L = _lib.lib()
        ws = _ops._workspace(L.spr_layernorm_bwd_workspace_bytes(c), x.device)
        dx = torch.empty_like(x)
        dg = torch.empty((c,), dtype=torch.float32, device=x.device)
        db = torch.empty((c,), dtype=torch.float32, device=x.device)
        _lib.check(L.spr_layernorm_bwd(_ops._ptr(x.detach()), m, c, _ops._ptr(gamma.detach().contiguous()), ctx.eps,
                                       _ops._ptr(dn), _ops._ptr(dp), _ops._ptr(dx), _ops._ptr(dg), _ops._ptr(db),
                                       _ops._ptr(ws), ws.numel(), _ops._stream(x)), "spr_layernorm_bwd")
        return dx, dg, db, None, None, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        with torch.no_grad():
            y = _ops.maxpool_raw(x, idx)
        ctx.save_for_backward(x, idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, idx = ctx.saved_tensors
        idx32 = idx if idx.dtype == torch.int32 else idx.to(torch.int32)
        if idx32.stride(1) != 1:
            idx32 = idx32.contiguous()
        ns, c = x.shape
        nq, k = idx32.shape
        dx = torch.empty_like(x)
        L = _lib.lib()
        ws = _ops._workspace(L.spr_scatter_workspace_bytes(ns, c), x.device)
        _lib.check(L.spr_maxpool_bwd(_ops._ptr(x.detach()), ns, c, _ops._ptr(idx32), nq, int(idx32.stride(0)), k,
                                     _ops._ptr(dy.contiguous()), _ops._ptr(dx), _ops._ptr(ws), ws.numel(), _ops._stream(x)),
                   "spr_maxpool_bwd")
        return dx, None


class GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        with torch.no_grad():
            y = _ops.gather_rows_raw(x, idx)
        ctx.save_for_backward(idx)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n_src, c = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        L = _lib.lib()
        ws = _ops._workspace(L.spr_scatter_workspace_bytes(n_src, c), dy.device)
        _lib.check(L.spr_scatter_rows_add(_ops._ptr(dy.contiguous()), _ops._ptr(idx), idx.numel(), c, n_src, _ops._ptr(dx),
                                          _ops._ptr(ws), ws.numel(), _ops._stream(dy)), "spr_scatter_rows_add")
        return dx, None


_LSE_HANDOVER = os.environ.get("SPR_NO_LSE_HANDOVER", "0") != "1"   # experiment switch (A/B timing)


class AttentionFn(torch.autograd.Function):
    """Varlen multi-head attention core (spr_attn_varlen_fwd).  Backward = spr_attn_varlen_bwd: the
    probabilities are recomputed tile by tile inside two kernels (dQ; dK and dV) -- nothing of size
    Lq x Lk is kept from the forward or written by the backward."""

    @staticmethod
    def forward(ctx, q, k, v, cu, kv_seg, max_len, nhead, lens_host, kv_seg_host):
        with torch.no_grad():
            if _LSE_HANDOVER:
                out, lse = _ops.attention_raw(q, k, v, cu, kv_seg, max_len, nhead, want_lse=True)
            else:
                out, lse = _ops.attention_raw(q, k, v, cu, kv_seg, max_len, nhead), None
        kvs = [int(x) for x in kv_seg_host]
        if sorted(kvs) != list(range(len(kvs))):
            raise NotImplementedError("attention backward needs kv_seg to be a permutation of the segments")
        ctx.meta = (int(nhead), int(max_len), kvs)
        ctx.lse = lse                       # [T, nhead] log2-sum-exp of the forward (or None): no grad, not an output
        ctx.save_for_backward(q, k, v, out, cu)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, cu = ctx.saved_tensors
        nhead, max_len, kvs = ctx.meta
        dq, dk, dv = _ops.attention_bwd(q.detach(), k.detach(), v.detach(), out.detach(), dout.contiguous(), cu, kvs,
                                        max_len, nhead, lse=ctx.lse)
        return dq, dk, dv, None, None, None, None, None, None


# ---- pose head ---------------------------------------------------------------------------------- #
class ProcrustesFn(torch.autograd.Function):
    """spr_weighted_procrustes; backward = implicit differentiation of the SVD-based rotation
    (spr_weighted_procrustes_bwd)."""

    @staticmethod
    def forward(ctx, a, b, w, pair_cu):
        with torch.no_grad():
            pose = _ops.weighted_procrustes_raw(a, b, w, pair_cu)
        ctx.has_w = w is not None
        ctx.save_for_backward(a, b, w, pair_cu)
        return pose

    @staticmethod
    def backward(ctx, dpose):
        a, b, w, pair_cu = ctx.saved_tensors
        dpose = dpose.contiguous()
        p = pair_cu.numel() - 1
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        dw = torch.empty_like(w) if (ctx.has_w and ctx.needs_input_grad[2]) else None
        _lib.check(_lib.lib().spr_weighted_procrustes_bwd(_ops._ptr(a.detach()), _ops._ptr(b.detach()),
                                                          _ops._ptr(w.detach() if w is not None else None),
                                                          _ops._ptr(pair_cu), p, _ops._ptr(dpose), _ops._ptr(da),
                                                          _ops._ptr(db), _ops._ptr(dw), _ops._stream(a)),
                   "spr_weighted_procrustes_bwd")
        return da, db, dw, None


class SinkhornFn(torch.autograd.Function):
    """(w, t_hat) = spr_sinkhorn_correspondences(feat, xyz, ..., alpha, beta); backward =
    spr_sinkhorn_bwd (unrolled slack-Sinkhorn iterations)."""

    @staticmethod
    def forward(ctx, feat, xyz, cu, cu_host, npairs, alpha, beta, n_iters, slack):
        with torch.no_grad():
            w, that = _ops.sinkhorn_correspondences_raw(feat, xyz, cu, cu_host, npairs, alpha, beta, n_iters, slack)
        ctx.meta = (list(cu_host), int(npairs), int(n_iters))
        ctx.save_for_backward(feat, xyz, cu, alpha, beta)
        return w, that

    @staticmethod
    def backward(ctx, dw, dthat):
        feat, xyz, cu, alpha, beta = ctx.saved_tensors
        cu_host, npairs, n_iters = ctx.meta
        arr = _ops._cu_host_arr(cu_host)
        L = _lib.lib()
        dev = feat.device
        ws = _ops._workspace(L.spr_sinkhorn_bwd_workspace_bytes(arr, npairs, n_iters), dev)
        T, d = feat.shape
        dfeat = torch.empty((T, d), dtype=torch.float32, device=dev)
        dal = torch.zeros((1,), dtype=torch.float32, device=dev)
        dbe = torch.zeros((1,), dtype=torch.float32, device=dev)
        a_t, b_t = _ops._dev_scalar(alpha, dev), _ops._dev_scalar(beta, dev)
        _lib.check(L.spr_sinkhorn_bwd(_ops._ptr(feat.detach().contiguous()), d, _ops._ptr(xyz), _ops._ptr(cu), arr, npairs,
                                      _ops._ptr(a_t), _ops._ptr(b_t), n_iters, _ops._ptr(dw.contiguous()),
                                      _ops._ptr(dthat.contiguous()), _ops._ptr(dfeat), _ops._ptr(dal), _ops._ptr(dbe),
                                      _ops._ptr(ws), ws.numel(), _ops._stream(feat)), "spr_sinkhorn_bwd")
        return dfeat, None, None, None, None, dal.reshape(alpha.shape), dbe.reshape(beta.shape), None, None


class MatchDualSoftmaxFn(torch.autograd.Function):
    """(val, ind) = spr_match_dualsoftmax(feat).  val_m = attn[i_m, j_m] with
    attn = softmax_rows(c) * softmax_cols(c), c = F_s F_t^T / sqrt(d)  (qk_regtr_full.py:453-468):
        d c_ij = 2 [ (i,j) matched ] G - Gr_i r_ij - Gc_j q_ij,   G_m = dval_m val_m,
    Gr / Gc = G summed over the matches of a row / column, r / q the two softmaxes.  The products
    and the softmaxes run in HIP (spr_bgemm, spr_softmax_rows); the combination above is
    elementwise glue on the (small) per-pair matrices."""

    @staticmethod
    def forward(ctx, feat, cu, cu_host, npairs):
        with torch.no_grad():
            val, ind = _ops.match_dualsoftmax_raw(feat, cu, cu_host, npairs)
        ctx.meta = (list(cu_host), int(npairs))
        ctx.save_for_backward(feat, val, ind)
        ctx.mark_non_differentiable(ind)
        return val, ind

    @staticmethod
    def backward(ctx, dval, _dind):
        feat, val, ind = ctx.saved_tensors
        cu_host, B = ctx.meta
        dev = feat.device
        T, d = feat.shape
        scale = 1.0 / math.sqrt(d)
        f = feat.detach().contiguous()
        dfeat = torch.zeros((T, d), dtype=torch.float32, device=dev)
        L = _lib.lib()
        for b in range(B):
            s0, s1, t0, t1 = cu_host[b], cu_host[b + 1], cu_host[B + b], cu_host[B + b + 1]
            n, m = s1 - s0, t1 - t0
            c = torch.empty((n, m), dtype=torch.float32, device=dev)
            ct = torch.empty((m, n), dtype=torch.float32, device=dev)
            bgemm(f, f, c, [(s0 * d, t0 * d, 0, n, m, d)], (d, 1), (1, d), (m, 1), alpha=scale)
            bgemm(f, f, ct, [(t0 * d, s0 * d, 0, m, n, d)], (d, 1), (1, d), (n, 1), alpha=scale)
            _lib.check(L.spr_softmax_rows(_ops._ptr(c), _ops._ptr(_desc([(0, 0, 0, n, m, 0)], dev)), 1, n, _ops._stream(f)),
                       "spr_softmax_rows")            # r: softmax over j
            _lib.check(L.spr_softmax_rows(_ops._ptr(ct), _ops._ptr(_desc([(0, 0, 0, m, n, 0)], dev)), 1, m, _ops._stream(f)),
                       "spr_softmax_rows")            # q^T: softmax over i
            if n > m:      # matches live on the tgt tokens: (ind[j], j)
                G = dval[t0:t1] * val[t0:t1]
                ii, jj = ind[t0:t1].long(), torch.arange(m, device=dev)
            else:          # matches live on the src tokens: (i, ind[i])
                G = dval[s0:s1] * val[s0:s1]
                ii, jj = torch.arange(n, device=dev), ind[s0:s1].long()
            Gr = torch.zeros((n,), dtype=torch.float32, device=dev).index_add_(0, ii, G)
            Gc = torch.zeros((m,), dtype=torch.float32, device=dev).index_add_(0, jj, G)
            dc = -(Gr.unsqueeze(1) * c) - (Gc.unsqueeze(1) * ct).t()
            dc.index_put_((ii, jj), 2.0 * G, accumulate=True)
            dc = dc.contiguous()
            # dFs = scale dc Ft ; dFt = scale dc^T Fs
            bgemm(dc, f, dfeat, [(0, t0 * d, s0 * d, n, d, m)], (m, 1), (d, 1), (d, 1), alpha=scale)
            bgemm(dc, f, dfeat, [(0, s0 * d, t0 * d, m, d, n)], (1, m), (d, 1), (d, 1), alpha=scale)
        return dfeat, None, None, None


# ---- losses -------------------------------------------------------------------------------------- #
class BCELogitsMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        with torch.no_grad():
            out = _ops.bce_logits_mean_raw(x, y)
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y = ctx.saved_tensors
        dx = torch.empty_like(x)
        g = gout.reshape(1).contiguous().float()
        _lib.check(_lib.lib().spr_bce_logits_mean_bwd(_ops._ptr(x.detach()), _ops._ptr(y), x.numel(), _ops._ptr(g),
                                                      _ops._ptr(dx), _ops._stream(x)), "spr_bce_logits_mean_bwd")
        return dx, None


class TransformL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose_gt, pose_pred, xyz):
        with torch.no_grad():
            out = _ops.transform_l1_pair_raw(pose_gt, pose_pred, xyz)
        ctx.save_for_backward(pose_gt, pose_pred, xyz)
        return out

    @staticmethod
    def backward(ctx, gout):
        pose_gt, pose_pred, xyz = ctx.saved_tensors
        dp = torch.empty((3, 4), dtype=torch.float32, device=xyz.device)
        g = gout.reshape(1).contiguous().float()
        _lib.check(_lib.lib().spr_transform_l1_pair_bwd(_ops._ptr(pose_gt.contiguous()), _ops._ptr(pose_pred.detach().contiguous()),
                                                        _ops._ptr(xyz), xyz.shape[0], _ops._ptr(g), _ops._ptr(dp),
                                                        _ops._stream(xyz)), "spr_transform_l1_pair_bwd")
        return None, dp, None


class InfoNCEFn(torch.autograd.Function):
    """InfoNCELossFull.compute_infonce of one pair (spr_infonce_pair); backward:
    d logits (spr_infonce_pair_dlogits) then dA = (dl B) W_sym, dB = dl^T (A W_sym),
    dW_sym = A^T (dl B) through spr_bgemm, dW = spr_wsym_bwd(dW_sym)."""

    @staticmethod
    def forward(ctx, a_feat, p_feat, a_xyz, pose_gt, p_xyz, W, r_p, r_n):
        with torch.no_grad():
            out = _ops.infonce_pair_raw(a_feat, p_feat, a_xyz, pose_gt, p_xyz, W, r_p, r_n)
        ctx.r = (float(r_p), float(r_n))
        ctx.save_for_backward(a_feat, p_feat, a_xyz, pose_gt, p_xyz, W)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, p, a_xyz, pose_gt, p_xyz, W = ctx.saved_tensors
        r_p, r_n = ctx.r
        dev = a.device
        a, p = a.detach().contiguous(), p.detach().contiguous()
        n, d = a.shape
        m = p.shape[0]
        L = _lib.lib()
        ws = _ops._loss_ws(n, m, d, dev)
        dl = torch.empty((n, m), dtype=torch.float32, device=dev)
        mask = torch.empty((n,), dtype=torch.float32, device=dev)
        wsym = torch.empty((d, d), dtype=torch.float32, device=dev)
        t = torch.empty((n, d), dtype=torch.float32, device=dev)
        _lib.check(L.spr_infonce_pair_dlogits(_ops._ptr(a), n, _ops._ptr(p), m, d, _ops._ptr(a_xyz.contiguous()),
                                              _ops._ptr(pose_gt.contiguous()), _ops._ptr(p_xyz.contiguous()),
                                              _ops._ptr(W.detach().contiguous()), r_p, r_n, _ops._ptr(dl), _ops._ptr(mask),
                                              _ops._ptr(wsym), _ops._ptr(t), _ops._ptr(ws), ws.numel(), _ops._stream(a)),
                   "spr_infonce_pair_dlogits")
        dl = (dl * (gout / mask.sum())).contiguous()
        dt = _nn_product(dl, p, n, m, d)                                    # d(A W_sym) = dl B
        da = torch.empty((n, d), dtype=torch.float32, device=dev)          # dA = dt W_sym^T = dt W_sym
        bgemm(dt, wsym, da, [(0, 0, 0, n, d, d)], (d, 1), (d, 1), (d, 1))
        # dB = dl^T t and dW_sym = A^T dt contract over the n anchors into small outputs ([m, d], [d, d] = 24 and 4
        # tiles): split-K batches like the weight gradients (as single products they ran on 24 / 4 CUs, 0.3 ms each)
        dp = _tn_product(dl, t, n, m, d)
        dws = _tn_product(a, dt, n, d, d)
        dW = torch.empty((d, d), dtype=torch.float32, device=dev)
        _lib.check(L.spr_wsym_bwd(_ops._ptr(dws), d, _ops._ptr(dW), _ops._stream(a)), "spr_wsym_bwd")
        return da, dp, None, None, None, dW, None, None
