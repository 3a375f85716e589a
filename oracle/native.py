"""TEST INFRASTRUCTURE ONLY -- ctypes bindings for the CPU oracle libraries.

* ``libspr_oracle.so``  our plain-C restatement (oracle/spr_oracle.c)
* ``_ref/libspr_ref.so`` the reference's own C++ core compiled in place
  (oracle/ref_driver.cpp); exists only where /root/reference was present at
  build time (the dev container) -- it travels to the GPU box as a built file.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libspr_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libspr_ref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(ref: bool = True) -> None:
    """(Re)build the oracle libs with oracle/Makefile (gcc/g++ only)."""
    targets = ["oracle"]
    if ref and os.path.isdir("/root/reference/src/models/backbone_kpconv/cpp_wrappers"):
        targets.append("_ref")
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


def _load(path):
    if not os.path.exists(path):
        return None
    return ctypes.CDLL(path)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        _oracle = ctypes.CDLL(_ORACLE_SO)
        _oracle.spr_oracle_grid_subsample.restype = ctypes.c_long
        _oracle.spr_oracle_grid_subsample.argtypes = [
            _f32p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
            ctypes.c_int, _f32p, _i32p, _u64p, _i32p]
        _oracle.spr_oracle_radius_neighbors.restype = ctypes.c_long
        _oracle.spr_oracle_radius_neighbors.argtypes = [
            _f32p, ctypes.c_int, _f32p, ctypes.c_int, _i32p, _i32p, ctypes.c_int,
            ctypes.c_float, ctypes.c_int, _i32p, _i32p]
        _oracle.spr_oracle_umap_order.restype = ctypes.c_int
        _oracle.spr_oracle_umap_order.argtypes = [_u64p, ctypes.c_int, _i32p]
    return _oracle


def ref_lib():
    """The compiled reference core, or None when it was never built."""
    global _ref
    if _ref is None and os.path.exists(_REF_SO):
        _ref = ctypes.CDLL(_REF_SO)
        _ref.ref_batch_query.restype = ctypes.c_long
        _ref.ref_batch_query.argtypes = [
            _f32p, ctypes.c_int, _f32p, ctypes.c_int, _i32p, _i32p, ctypes.c_int,
            ctypes.c_float, _i32p, ctypes.c_long]
        _ref.ref_subsample_batch.restype = ctypes.c_long
        _ref.ref_subsample_batch.argtypes = [
            _f32p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
            _f32p, _i32p]
    return _ref


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------- #
# our restatement
# --------------------------------------------------------------------------- #
def grid_subsample(points, lengths, dl, max_p=0, order="reference", return_keys=False):
    """Oracle of batch_grid_subsampling (grid_subsampling.cpp:109).

    order: "reference" = libstdc++ unordered_map iteration order (what the
    reference emits), "canonical" = ascending voxel key.
    """
    points, lengths = _f32(points), _i32(lengths)
    n, nb = points.shape[0], lengths.shape[0]
    out = np.empty((max(n, 1), 3), np.float32)
    out_len = np.empty((nb,), np.int32)
    keys = np.empty((max(n, 1),), np.uint64)
    first = np.empty((max(n, 1),), np.int32)
    m = oracle_lib().spr_oracle_grid_subsample(
        _p(points, _f32p), n, _p(lengths, _i32p), nb, float(dl), int(max_p),
        0 if order == "reference" else 1, _p(out, _f32p), _p(out_len, _i32p),
        _p(keys, _u64p), _p(first, _i32p))
    if m < 0:
        raise RuntimeError("oracle grid_subsample failed")
    if return_keys:
        return out[:m].copy(), out_len, keys[:m].copy(), first[:m].copy()
    return out[:m].copy(), out_len


def radius_neighbors(queries, supports, q_lengths, s_lengths, radius, limit=0):
    """Oracle of batch_nanoflann_neighbors + the [:, :limit] slice
    (neighbors.cpp:211, kpconv.py:258-262).  Returns (int32 [Nq, W], max_count).
    """
    queries, supports = _f32(queries), _f32(supports)
    qb, sb = _i32(q_lengths), _i32(s_lengths)
    nq, ns, nb = queries.shape[0], supports.shape[0], qb.shape[0]
    mc = ctypes.c_int(0)
    lib = oracle_lib()
    w = lib.spr_oracle_radius_neighbors(
        _p(queries, _f32p), nq, _p(supports, _f32p), ns, _p(qb, _i32p), _p(sb, _i32p),
        nb, float(radius), int(limit), None, ctypes.byref(mc))
    out = np.empty((nq, max(w, 0)), np.int32)
    if w > 0:
        lib.spr_oracle_radius_neighbors(
            _p(queries, _f32p), nq, _p(supports, _f32p), ns, _p(qb, _i32p),
            _p(sb, _i32p), nb, float(radius), int(limit), _p(out, _i32p),
            ctypes.byref(mc))
    return out, int(mc.value)


def umap_order(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    order = np.empty((keys.shape[0],), np.int32)
    rc = oracle_lib().spr_oracle_umap_order(_p(keys, _u64p), keys.shape[0], _p(order, _i32p))
    if rc != 0:
        raise RuntimeError("umap_order failed")
    return order


# --------------------------------------------------------------------------- #
# the compiled reference (oracle/_ref)
# --------------------------------------------------------------------------- #
def ref_available():
    return ref_lib() is not None


def ref_grid_subsample(points, lengths, dl, max_p=0):
    points, lengths = _f32(points), _i32(lengths)
    n, nb = points.shape[0], lengths.shape[0]
    out = np.empty((max(n, 1), 3), np.float32)
    out_len = np.empty((nb,), np.int32)
    m = ref_lib().ref_subsample_batch(_p(points, _f32p), n, _p(lengths, _i32p), nb,
                                      float(dl), int(max_p), _p(out, _f32p),
                                      _p(out_len, _i32p))
    return out[:m].copy(), out_len


def ref_radius_neighbors(queries, supports, q_lengths, s_lengths, radius):
    queries, supports = _f32(queries), _f32(supports)
    qb, sb = _i32(q_lengths), _i32(s_lengths)
    nq, ns, nb = queries.shape[0], supports.shape[0], qb.shape[0]
    lib = ref_lib()
    mc = lib.ref_batch_query(_p(queries, _f32p), nq, _p(supports, _f32p), ns,
                             _p(qb, _i32p), _p(sb, _i32p), nb, float(radius), None, 0)
    out = np.empty((nq, mc), np.int32)
    if mc > 0:
        lib.ref_batch_query(_p(queries, _f32p), nq, _p(supports, _f32p), ns,
                            _p(qb, _i32p), _p(sb, _i32p), nb, float(radius),
                            _p(out, _i32p), out.size)
    return out
