"""Concurrent forwards on separate HIP streams.

Every pair is independent, so a step of B pairs can run as `n_streams`
independent forwards of B / n_streams pairs, each enqueued on its own HIP stream
by its own host thread.  Most kernels of the path leave a CU's issue slots or
memory pipeline idle part of the time (one LDS-sized workgroup per CU, host
round trips in the pyramid builder); a second forward in flight fills those
gaps -- measured on MI355X: 2 streams x 16 pairs = 1.17x the pairs/s of one
32-pair forward (4 streams is slower again: the kernels start to queue).

The reference has no counterpart (its DataLoader feeds one batch at a time to
one CUDA stream, trainer.py:172); results per pair are those of a forward of
that pair's group (the reference's outputs depend on the batch composition
through the neighbour-matrix width).
"""
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List

import torch

_LIST_KEYS = ('attn', 'src_feat', 'tgt_feat', 'src_kp', 'tgt_kp', 'src_corr', 'tgt_corr',
              'src_overlap', 'tgt_overlap', 'overlap_prob_list', 'ind_list')


def split_batch(batch: Dict, n: int) -> List[Dict]:
    """Contiguous groups of pairs (sizes differ by at most one)."""
    B = len(batch['src_xyz'])
    n = max(1, min(n, B))
    base, rem = divmod(B, n)
    out, start = [], 0
    for g in range(n):
        size = base + (1 if g < rem else 0)
        sl = slice(start, start + size)
        sub = {k: (v[sl] if isinstance(v, (list, tuple)) and len(v) == B else v) for k, v in batch.items()
               if k != 'kpconv_meta'}
        out.append(sub)
        start += size
    return out


def merge_outputs(outs: List[Dict]) -> Dict:
    merged = {'pose': torch.cat([o['pose'] for o in outs], dim=0)}
    for k in _LIST_KEYS:
        if k in outs[0]:
            merged[k] = [x for o in outs for x in o[k]]
    return merged


class StreamedForward:
    """model(batch) as `n_streams` concurrent group forwards.  The caller's current
    stream is ordered before and after the groups, so the result can be used like
    that of a plain forward.  batch['kpconv_meta'] becomes the list of the groups'
    pyramids."""

    def __init__(self, model: torch.nn.Module, n_streams: int = 2, device=None):
        self.model = model
        self.n = int(n_streams)
        self.device = device if device is not None else next(model.parameters()).device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
        self.pool = ThreadPoolExecutor(max_workers=self.n) if self.n > 1 else None

    def _one(self, i: int, sub: Dict, ready: torch.cuda.Event):
        torch.cuda.set_device(self.device)
        s = self.streams[i]
        from .regtr import no_side_stream
        with torch.cuda.stream(s), torch.no_grad(), no_side_stream():
            s.wait_event(ready)
            out = self.model(sub)
            done = torch.cuda.Event()
            done.record(s)
        return out, done, sub.get('kpconv_meta')

    def __call__(self, batch: Dict) -> Dict:
        if self.n <= 1:
            return self.model(batch)
        subs = split_batch(batch, self.n)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        futs = [self.pool.submit(self._one, i, sub, ready) for i, sub in enumerate(subs)]
        res = [f.result() for f in futs]
        cur = torch.cuda.current_stream(self.device)
        for _, done, _ in res:
            cur.wait_event(done)
        batch['kpconv_meta'] = [m for _, _, m in res]
        return merge_outputs([o for o, _, _ in res])
