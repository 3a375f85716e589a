"""CPU: the C-ABI library loads and exports every symbol include/spr.h
declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from superpoints_registration_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "spr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spr_[a-z0-9_]+)\s*\(", text)))


def test_library_exists_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    ctypes.CDLL(_lib.LIB_PATH)


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in spr.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_channel():
    L = _lib.lib()
    assert L.spr_version() == 5
    assert isinstance(L.spr_last_error(), bytes)


def test_host_side_argument_validation_needs_no_gpu():
    # bad arguments are rejected on the host before any HIP call
    L = _lib.lib()
    rc = L.spr_linear(None, 0, 32, None, 32, None, None, 0, None, None, 0, None)
    assert rc != 0 and b"linear" in L.spr_last_error()
    rc = L.spr_attn_varlen_fwd(None, 256, None, 256, None, 256, None, None, 20, 2, 10, 8, 64, 0.1, None, 256,
                               None, 0, None)
    assert rc != 0 and b"head_dim" in L.spr_last_error()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from superpoints_registration_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 32), torch.zeros(8, 32))
