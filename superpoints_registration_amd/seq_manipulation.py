"""Varlen <-> padded layout helpers exposing the call signatures RegTR's
callers know (reference: src/utils/seq_manipulation.py -- pad_sequence :6,
unpad_sequences :36, split_src_tgt :42).

The HIP path keeps tokens packed ([sum L, D] + cu_seqlens) end to end; these
helpers exist only so code written against the padded (L_max, B, D) convention
keeps working at the boundary.
"""
from typing import List, Optional, Sequence, Tuple

import torch


def pad_sequence(sequences: Sequence[torch.Tensor], require_padding_mask: bool = False,
                 require_lens: bool = False, batch_first: bool = False
                 ) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[List[int]]]:
    """List of (L_b, D) -> zero padded (L_max, B, D) [or (B, L_max, D)], an
    optional bool mask (B, L_max) that is True on padding, optional lengths."""
    lens = [int(s.shape[0]) for s in sequences]
    n_seq, l_max = len(sequences), max(lens)
    tail = tuple(sequences[0].shape[1:])
    out = sequences[0].new_zeros((n_seq, l_max) + tail)
    for b, (s, n) in enumerate(zip(sequences, lens)):
        out[b, :n] = s
    if not batch_first:
        out = out.transpose(0, 1).contiguous()
    mask = None
    if require_padding_mask:
        steps = torch.arange(l_max, device=out.device).unsqueeze(0)
        mask = steps >= torch.tensor(lens, device=out.device).unsqueeze(1)
    return out, mask, (lens if require_lens else None)


def unpad_sequences(padded: torch.Tensor, seq_lens: Sequence[int]) -> List[torch.Tensor]:
    """(..., L_max, B, D) -> list over b of (..., L_b, D) views."""
    return [padded.select(-2, b).narrow(-2, 0, int(n)) for b, n in enumerate(seq_lens)]


def split_src_tgt(feats: torch.Tensor, stack_lengths, dim: int = 0):
    """Clouds are stacked [src_0..src_{B-1}, tgt_0..tgt_{B-1}]; returns the
    tuple of src chunks and the tuple of tgt chunks."""
    lens = stack_lengths.tolist() if isinstance(stack_lengths, torch.Tensor) else list(stack_lengths)
    half = len(lens) // 2
    chunks = torch.split(feats, lens, dim=dim)
    return chunks[:half], chunks[half:]
