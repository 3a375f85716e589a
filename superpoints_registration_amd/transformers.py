"""Superpoint self/cross-attention encoder -- host-side mirror of the
reference's ``src/models/transformer/transformers.py`` (TransformerCrossEncoder
:18-82, TransformerCrossEncoderLayer :84-258) and
``position_embedding.py`` (PositionEmbeddingCoordsSine :7-50) on the HIP
library.

Parameter containers are the same torch modules the reference uses
(nn.MultiheadAttention, nn.Linear, nn.LayerNorm), so state-dict names and
shapes are identical; their forward() is never called -- projections, the
attention core and the norms run in libspr_hip.so on PACKED tokens
(cu_seqlens), not on the reference's zero-padded (L, B, D) layout.

Two entry points:
  * forward(src, tgt, ..., src_key_padding_mask, ...)  the reference signature
    (padded (L,B,D) in, (1,L,B,D) out) for drop-in use;
  * forward_packed(x, cu, ...)  the native packed path RegTR uses.
"""
import copy
import math
from typing import Optional

import torch
from torch import nn, Tensor

from . import ops


class PositionEmbeddingCoordsSine(nn.Module):
    """position_embedding.py:7-50."""

    def __init__(self, n_dim: int = 1, d_model: int = 256, temperature=10000, scale=None):
        super().__init__()
        if n_dim != 3:
            raise NotImplementedError("the hot path embeds 3-D coordinates")
        self.n_dim = n_dim
        self.num_pos_feats = d_model // n_dim // 2 * 2
        self.temperature = temperature
        self.padding = d_model - self.num_pos_feats * self.n_dim
        self.d_model = d_model
        if scale is None:
            scale = 1.0
        self.scale_arg = scale
        self.scale = scale * 2 * math.pi

    def forward(self, xyz: torch.Tensor) -> torch.Tensor:
        assert xyz.shape[-1] == self.n_dim
        lead = xyz.shape[:-1]
        out = ops.posemb_sine(xyz.reshape(-1, 3), self.d_model, scale=self.scale_arg,
                              temperature=float(self.temperature))
        return out.view(*lead, self.d_model)


def _pack(padded: Tensor, key_padding_mask: Optional[Tensor]):
    """(L,B,D) + mask (B,L) [True = pad]  ->  packed [sum L_b, D], lengths."""
    L, B, D = padded.shape
    if key_padding_mask is None:
        lens = [L] * B
    else:
        lens = (~key_padding_mask).sum(dim=1).tolist()
    seqs = [padded[:lens[b], b, :] for b in range(B)]
    return torch.cat(seqs, dim=0).contiguous(), lens


def _unpack(packed: Tensor, lens, L: int):
    B, D = len(lens), packed.shape[1]
    out = packed.new_zeros((L, B, D))
    off = 0
    for b, n in enumerate(lens):
        out[:n, b, :] = packed[off:off + n]
        off += n
    return out


class TransformerCrossEncoderLayer(nn.Module):
    """transformers.py:84-258."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False, sa_val_has_pos_emb=False, ca_val_has_pos_emb=False,
                 attention_type='dot_prod', batch_first=False):
        super().__init__()
        if attention_type != 'dot_prod':
            raise NotImplementedError
        if activation != "relu":
            raise NotImplementedError("only transformer_act='relu' (all shipped configs)")
        if dropout != 0.0:
            raise NotImplementedError("dropout must be 0.0 (all shipped configs; inference path)")
        if (d_model // nhead) != 32:
            raise NotImplementedError("head_dim must be 32 (d_embed 256 / nhead 8)")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.d_model = d_model
        self.nhead = nhead
        self.normalize_before = normalize_before
        self.sa_val_has_pos_emb = sa_val_has_pos_emb
        self.ca_val_has_pos_emb = ca_val_has_pos_emb
        self.satt_weights, self.xatt_weights = None, None  # never materialised here

    # -- one MHA over packed tokens: in_proj GEMM, attention core, out_proj GEMM (+ residual)
    def _mha(self, mha: nn.MultiheadAttention, qk_in, v_in, cu, kv_seg, max_len, residual, seg_host=None):
        d = self.d_model
        W, b = mha.in_proj_weight, mha.in_proj_bias
        training = torch.is_grad_enabled() and (qk_in.requires_grad or W.requires_grad)
        if d == 256 and not training:
            # in-projection GEMM writes the attention operand planes directly (inference path)
            o = ops.attention_inproj(qk_in, v_in, W.detach(), b.detach(), cu, kv_seg, max_len, self.nhead,
                                     w_prep=ops.inproj_prepare(W))   # cached on the parameter object
        else:
            if v_in is qk_in:
                qkv = ops.linear(qk_in, W, b)                      # [T, 3d]
                q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            else:
                qk = ops.linear(qk_in, W[:2 * d], b[:2 * d])       # [T, 2d]
                q, k = qk[:, :d], qk[:, d:]
                v = ops.linear(v_in, W[2 * d:], b[2 * d:])
            lens_host, kv_host = seg_host if seg_host is not None else (None, None)
            o = ops.attention(q, k, v, cu, kv_seg, max_len, self.nhead, lens_host=lens_host, kv_seg_host=kv_host)
        return ops.linear(o, mha.out_proj.weight, mha.out_proj.bias, residual=residual)

    def _ln(self, norm: nn.LayerNorm, x, pos, need_plain):
        plain, with_pos = ops.layernorm(x, norm.weight, norm.bias, norm.eps,
                                        pos=pos, want_norm=need_plain or pos is None)
        return plain, (with_pos if pos is not None else plain)

    def forward_packed(self, x, cu, seg_self, seg_cross, max_len, pos=None, seg_host=None):
        """x: [T, d] packed tokens of all 2B clouds; returns the updated tokens.
        Pre-norm (transformers.py:184-245) or post-norm (:124-182).  seg_host: optional
        (lens, self map, cross map) host lists -- spares the attention backward a device read."""
        sh_self = (seg_host[0], seg_host[1]) if seg_host is not None else None
        sh_cross = (seg_host[0], seg_host[2]) if seg_host is not None else None
        if self.normalize_before:
            # self attention (same weights for src and tgt clouds)
            x2, x2p = self._ln(self.norm1, x, pos, need_plain=not self.sa_val_has_pos_emb)
            x = self._mha(self.self_attn, x2p, x2p if self.sa_val_has_pos_emb else x2, cu, seg_self,
                          max_len, residual=x, seg_host=sh_self)
            # cross attention (keys/values from the partner cloud)
            x2, x2p = self._ln(self.norm2, x, pos, need_plain=not self.ca_val_has_pos_emb)
            x = self._mha(self.multihead_attn, x2p, x2p if self.ca_val_has_pos_emb else x2, cu,
                          seg_cross, max_len, residual=x, seg_host=sh_cross)
            # feed forward
            x2, _ = self._ln(self.norm3, x, None, need_plain=True)
            h = ops.linear(x2, self.linear1.weight, self.linear1.bias, act=ops.ACT_RELU)
            x = ops.linear(h, self.linear2.weight, self.linear2.bias, residual=x)
            return x
        # post-norm
        xp = x + pos if pos is not None else x
        y = self._mha(self.self_attn, xp, xp if self.sa_val_has_pos_emb else x, cu, seg_self, max_len,
                      residual=x, seg_host=sh_self)
        x, _ = self._ln(self.norm1, y, None, True)
        xp = x + pos if pos is not None else x
        y = self._mha(self.multihead_attn, xp, xp if self.ca_val_has_pos_emb else x, cu, seg_cross,
                      max_len, residual=x, seg_host=sh_cross)
        x, _ = self._ln(self.norm2, y, None, True)
        h = ops.linear(x, self.linear1.weight, self.linear1.bias, act=ops.ACT_RELU)
        y = ops.linear(h, self.linear2.weight, self.linear2.bias, residual=x)
        x, _ = self._ln(self.norm3, y, None, True)
        return x


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def make_segments(src_lens, tgt_lens, device):
    """cu_seqlens + self / cross kv segment maps for clouds stacked
    [src_0..src_{B-1}, tgt_0..tgt_{B-1}]."""
    B = len(src_lens)
    lens = list(src_lens) + list(tgt_lens)
    cu = ops.lengths_to_cu(lens, device)
    seg_self = torch.arange(2 * B, dtype=torch.int32, device=device)
    seg_cross = torch.cat([torch.arange(B, 2 * B, dtype=torch.int32, device=device),
                           torch.arange(0, B, dtype=torch.int32, device=device)])
    return cu, seg_self, seg_cross, max(lens)


class TransformerCrossEncoder(nn.Module):
    """transformers.py:18-82."""

    def __init__(self, cross_encoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        if return_intermediate:
            raise NotImplementedError("return_intermediate=False in RegTR (qk_regtr_full.py:72-74)")
        self.layers = _get_clones(cross_encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    # -- fused stack (csrc/xenc.hip): two attention cores + two row-chain kernels per layer ---------
    def _xenc_eligible(self, x, pos, pos_bound, nseg: int = 0) -> bool:
        if pos is None or pos_bound is None or not ops.xenc_available() or x.shape[1] != 256:
            return False
        # the chains keep cu_seqlens and the tile prefix of every segment in LDS behind the 128 KB weight ring
        # (csrc/xenc.hip xenc_lds_bytes: 23 232 bytes for 4 d_ff + 8 (segments + 1))
        if 4 * max(l.linear1.out_features for l in self.layers) + 8 * (nseg + 1) > 23232:
            return False
        # one plan = at most 16 layers (csrc/xenc.hip kMaxLayers) of ONE feed-forward width (the plan is prepared with
        # the first layer's): anything else takes the operator route instead of failing in xenc_prepare
        if len(self.layers) > 16 or len({l.linear1.out_features for l in self.layers}) != 1:
            return False
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False                                  # training: the differentiable per-operator route
        return all(l.normalize_before and l.sa_val_has_pos_emb and l.ca_val_has_pos_emb and l.nhead == 8
                   and l.linear1.out_features % 64 == 0 for l in self.layers)

    def _xenc_plan(self, pos_bound: float):
        lp, le = [], []
        for l in self.layers:
            lp.append([l.self_attn.in_proj_weight, l.self_attn.in_proj_bias, l.self_attn.out_proj.weight,
                       l.self_attn.out_proj.bias, l.multihead_attn.in_proj_weight, l.multihead_attn.in_proj_bias,
                       l.multihead_attn.out_proj.weight, l.multihead_attn.out_proj.bias, l.linear1.weight,
                       l.linear1.bias, l.linear2.weight, l.linear2.bias, l.norm1.weight, l.norm1.bias,
                       l.norm2.weight, l.norm2.bias, l.norm3.weight, l.norm3.bias])
            le.append((l.norm1.eps, l.norm2.eps, l.norm3.eps))
        final = (self.norm.weight, self.norm.bias, self.norm.eps) if self.norm is not None else None
        plan = ops.xenc_prepare(lp, le, final, 8, self.layers[0].linear1.out_features, pos_bound,
                                cached=getattr(self, '_spr_xenc', None))
        self._spr_xenc = plan
        return plan

    def forward_packed(self, x, cu, seg_self, seg_cross, max_len, pos=None, seg_host=None, pos_bound=None):
        """pos_bound: an upper bound of max |pos| the caller guarantees (1.0 for the sine embedding).  With it
        (and the shipped layer configuration, inference) the stack runs as the fused chains of csrc/xenc.hip;
        without it operator by operator."""
        if self._xenc_eligible(x, pos, pos_bound, int(cu.numel()) - 1):
            return ops.xenc_forward(self._xenc_plan(float(pos_bound)), x, pos, cu, seg_self, seg_cross, max_len)
        for layer in self.layers:
            x = layer.forward_packed(x, cu, seg_self, seg_cross, max_len, pos=pos, seg_host=seg_host)
        if self.norm is not None:
            x, _ = ops.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x

    def forward(self, src, tgt, src_mask: Optional[Tensor] = None, tgt_mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                src_pos: Optional[Tensor] = None, tgt_pos: Optional[Tensor] = None):
        """Reference signature: padded (L,B,D) in, (1,L,B,D) x 2 out."""
        assert src_mask is None and tgt_mask is None, 'Masking not implemented'
        Ls, Lt = src.shape[0], tgt.shape[0]
        sp, slens = _pack(src, src_key_padding_mask)
        tp, tlens = _pack(tgt, tgt_key_padding_mask)
        x = torch.cat([sp, tp], dim=0)
        pos = None
        if src_pos is not None:
            spp, _ = _pack(src_pos, src_key_padding_mask)
            tpp, _ = _pack(tgt_pos, tgt_key_padding_mask)
            pos = torch.cat([spp, tpp], dim=0)
        cu, seg_self, seg_cross, max_len = make_segments(slens, tlens, x.device)
        y = self.forward_packed(x, cu, seg_self, seg_cross, max_len, pos=pos)
        ns = sum(slens)
        return _unpack(y[:ns], slens, Ls).unsqueeze(0), _unpack(y[ns:], tlens, Lt).unsqueeze(0)
