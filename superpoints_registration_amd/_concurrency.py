"""Auxiliary HIP streams of one forward (regtr.py: pyramid; kpconv_blocks.py: shortcut branch).

One set per (device, caller's stream), so that streams.StreamedForward's host threads -- each on
its own stream -- do not share them.  SPR_NO_SIDE_STREAM=1 or the no_side_stream() context put
everything back on the caller's stream (A/B timing, and StreamedForward's group forwards, which
already run beside each other)."""
import os
import threading

import torch

ENABLED = os.environ.get("SPR_NO_SIDE_STREAM", "0") != "1"
_streams = {}
_tls = threading.local()


class no_side_stream:
    """Forwards issued inside (by this thread) use the caller's stream only."""

    def __enter__(self):
        self._prev = getattr(_tls, 'off', False)
        _tls.off = True

    def __exit__(self, *exc):
        _tls.off = self._prev


def active(device) -> bool:
    return ENABLED and device.type == 'cuda' and not getattr(_tls, 'off', False)


def aux_stream(main, device, role: str):
    key = (device, main.cuda_stream, role)
    st = _streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _streams[key] = st
    return st
