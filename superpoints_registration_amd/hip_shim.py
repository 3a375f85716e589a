"""Native-level drop-in for the reference's two CPython extension modules.

The reference's pyramid builder imports ``cpp_subsampling`` / ``cpp_neighbors``
(src/models/backbone_kpconv/kpconv.py:14-15) and calls
``cpp_subsampling.subsample_batch(points, batches, sampleDl=, max_p=, verbose=)`` (kpconv.py:179-183)
and ``cpp_neighbors.batch_query(queries, supports, q_batches, s_batches, radius=)`` (kpconv.py:258)
with numpy arrays.  The two classes below have those names, argument meanings, return types and
error behaviour (RuntimeError, like cpp_wrappers/*/wrapper.cpp) and run on ``libspr_hip.so``
through ctypes: a maintainer drops this file next to kpconv.py and replaces the two imports by
``from .hip_shim import cpp_subsampling, cpp_neighbors``.  torch is used for device memory only.
"""
import ctypes
import os

import numpy as np
import torch

_LIB = os.environ.get("SPR_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspr_hip.so")
_L = None


def _lib():
    global _L
    if _L is None:
        L = ctypes.CDLL(_LIB)          # OSError if the library is missing: there is no CPU fallback
        L.spr_grid_subsample_workspace_bytes.restype = ctypes.c_size_t
        L.spr_radius_neighbors_workspace_bytes.restype = ctypes.c_size_t
        L.spr_last_error.restype = ctypes.c_char_p
        _L = L
    return _L


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _cu(lens, dev):
    cu = np.zeros(len(lens) + 1, np.int32)
    cu[1:] = np.cumsum(np.asarray(lens, dtype=np.int64))
    return torch.from_numpy(cu).to(dev)


def _stream(dev):
    """The caller's current HIP stream: the tensors above are allocated and copied on it, and the results are
    read back on it (torch streams do not synchronise with the NULL stream)."""
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _raise(L):
    msg = L.spr_last_error()
    raise RuntimeError(msg.decode() if msg else "Error")


class cpp_subsampling:
    """Replaces grid_subsampling.cpython-*.so (cpp_subsampling/wrapper.cpp)."""

    @staticmethod
    def subsample_batch(points, batches, sampleDl=0.1, max_p=0, verbose=0):
        """(points [N,3] f32, batches [B] i32) -> (sub_points [N',3] f32, sub_batches [B] i32), numpy."""
        L = _lib()
        dev = torch.device("cuda")
        x = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32)).to(dev)
        n, nb = int(x.shape[0]), len(batches)
        if n < 1 or nb < 1 or int(np.sum(batches)) != n:
            raise RuntimeError("Error converting input to numpy arrays of the right type")   # wrapper.cpp:60-66
        ws = torch.empty(L.spr_grid_subsample_workspace_bytes(n, nb), dtype=torch.uint8, device=dev)
        out = torch.empty((n, 3), dtype=torch.float32, device=dev)
        lens = torch.empty(nb, dtype=torch.int32, device=dev)
        tot = torch.empty(1, dtype=torch.int32, device=dev)
        cu = _cu(batches, dev)       # (held in a variable: a temporary would be freed before the kernels run)
        rc = L.spr_grid_subsample(_p(x), _p(cu), n, nb, ctypes.c_float(sampleDl), int(max_p), 0,
                                  _p(out), _p(lens), _p(tot), _p(ws), ctypes.c_size_t(ws.numel()), _stream(dev))
        if rc:
            _raise(L)
        m = int(tot.item())
        if m < 0:
            raise RuntimeError("voxel grid too large for 40-bit keys")
        return out[:m].cpu().numpy(), lens.cpu().numpy()


class cpp_neighbors:
    """Replaces radius_neighbors.cpython-*.so (cpp_neighbors/wrapper.cpp)."""

    @staticmethod
    def batch_query(queries, supports, q_batches, s_batches, radius=0.1, limit=128):
        """-> int32 [Nq, max_count] neighbour indices into `supports`, shadow index = Ns, rows ordered
        by distance.  `limit` = the library's cap on the row width (128); the reference's callers
        slice to neighborhood_limits (<= 74) anyway (kpconv.py:259-260)."""
        L = _lib()
        dev = torch.device("cuda")
        q = torch.as_tensor(np.ascontiguousarray(queries, dtype=np.float32)).to(dev)
        s = torch.as_tensor(np.ascontiguousarray(supports, dtype=np.float32)).to(dev)
        nq, ns, nb = int(q.shape[0]), int(s.shape[0]), len(q_batches)
        if nq < 1 or ns < 1 or nb < 1 or len(s_batches) != nb:
            raise RuntimeError("Error")                                   # wrapper.cpp:201-205
        ws = torch.empty(L.spr_radius_neighbors_workspace_bytes(nq, ns, nb), dtype=torch.uint8, device=dev)
        out = torch.empty((nq, limit), dtype=torch.int32, device=dev)
        mc = torch.empty(1, dtype=torch.int32, device=dev)
        m = -2
        q_cu, s_cu = _cu(q_batches, dev), _cu(s_batches, dev)   # both alive until the results are read back
        for algo in (0, 1):                  # 0 = cell table, 1 = sorted keys (no geometry limit)
            rc = L.spr_radius_neighbors(_p(q), _p(q_cu), nq, _p(s), _p(s_cu), ns, nb,
                                        ctypes.c_float(radius), int(limit), algo, _p(out), _p(mc), _p(ws),
                                        ctypes.c_size_t(ws.numel()), _stream(dev))
            if rc:
                _raise(L)
            m = int(mc.item())
            if m != -2:
                break
        if m < 1:
            raise RuntimeError("Error")                                   # wrapper.cpp:201-205
        return out[:, :min(m, limit)].cpu().numpy()
