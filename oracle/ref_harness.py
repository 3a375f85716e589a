"""TEST INFRASTRUCTURE ONLY -- imports the *reference's own Python* (read-only,
from /root/reference) in the dev container so that golden vectors can be
generated from it (oracle/gen_golden.py).  Nothing here runs on the GPU box:
/root/reference does not exist there.

Recipe = SURVEY.md Appendix A:
  * the reference's native core is reached through oracle/_ref (ctypes) because
    its CPython wrappers do not compile against numpy 2;
  * absent third-party imports (MinkowskiEngine, pytorch3d, tensorboard,
    nibabel, easydict) are stubbed in sys.modules -- they are not on the CPU
    path we pin (Preprocessor, not PreprocessorGPU);
  * models/__init__.py (auto-imports everything) is bypassed by registering an
    empty package object first.
"""
import importlib
import os
import sys
import types

import numpy as np

REF_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isdir(REF_SRC)


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_loaded = {}


def load():
    """Returns a namespace dict with the reference modules on the hot path."""
    if _loaded:
        return _loaded
    from . import native

    if not native.ref_available():
        native.build(ref=True)
    assert native.ref_available(), "oracle/_ref could not be built"

    sys.dont_write_bytecode = True
    os.chdir(REF_SRC)  # kernel_points.py:390 looks up kernels/dispositions relative to cwd
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)

    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF_SRC, "models")]
    sys.modules["models"] = pkg

    _stub("MinkowskiEngine")
    p3d = _stub("pytorch3d")
    p3d.ops = _stub("pytorch3d.ops", packed_to_padded=None, ball_query=None)

    class _SW:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_histogram(self, *a, **k):
            pass

        def flush(self):
            pass

    _stub("tensorboard")
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        _stub("torch.utils.tensorboard", SummaryWriter=_SW)
    nib = _stub("nibabel")
    nib.quaternions = _stub("nibabel.quaternions")
    _stub("easydict", EasyDict=_EasyDict)

    kp = importlib.import_module("models.backbone_kpconv.kpconv")

    class _Sub:
        @staticmethod
        def subsample_batch(points, batches, sampleDl=0.1, max_p=0, verbose=0, **kw):
            assert not kw, "features/classes are not used on this path"
            return native.ref_grid_subsample(np.asarray(points), np.asarray(batches),
                                             sampleDl, max_p)

    class _Nbr:
        @staticmethod
        def batch_query(queries, supports, q_batches, s_batches, radius=0.1):
            out = native.ref_radius_neighbors(np.asarray(queries), np.asarray(supports),
                                              np.asarray(q_batches), np.asarray(s_batches),
                                              radius)
            if out.size < 1:  # cpp_neighbors/wrapper.cpp:201-205
                raise RuntimeError("Error")
            return out

    kp.cpp_subsampling = _Sub
    kp.cpp_neighbors = _Nbr

    _loaded.update(
        kpconv=kp,
        blocks=importlib.import_module("models.backbone_kpconv.kpconv_blocks"),
        se3=importlib.import_module("utils.se3_torch"),
        seq=importlib.import_module("utils.seq_manipulation"),
        transformers=importlib.import_module("models.transformer.transformers"),
        posemb=importlib.import_module("models.transformer.position_embedding"),
        misc=importlib.import_module("utils.misc"),
        EasyDict=_EasyDict,
    )
    return _loaded


def _mat2quat(M):
    """nibabel.quaternions.mat2quat restated (nibabel is not installed here and the reference does not
    pin a version; algorithm of nibabel 2.x-5.x, Bar-Itzhack 2000): the unit quaternion (w, x, y, z),
    w >= 0, is the eigenvector of the largest eigenvalue of the symmetric 4x4 matrix K built from M."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, dtype=np.float64).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = q * -1
    return q


def load_benchmark():
    """The reference's 3DMatch evaluator, src/benchmark/benchmark_predator.py, imported as it stands.
    It was written against numpy < 1.24 (np.int / np.float) and imports nibabel: the two aliases are
    restored for the import and nibabel.quaternions.mat2quat is provided by the restatement above."""
    load()
    if not hasattr(np, 'int'):
        np.int = int          # noqa: NPY001 -- aliases the reference file expects (numpy < 1.24)
    if not hasattr(np, 'float'):
        np.float = float
    sys.modules['nibabel.quaternions'].mat2quat = _mat2quat
    sys.modules['nibabel'].quaternions = sys.modules['nibabel.quaternions']
    return importlib.import_module("benchmark.benchmark_predator")


def load_regtr():
    ns = load()
    if "regtr" not in ns:
        ns["regtr"] = importlib.import_module("models.qk_regtr_full")
    return ns


def make_model(cfg_name: str, seed: int = 0):
    """RegTR(cfg) with the CPU Preprocessor swapped in (qk_regtr_full.py:40
    hard-wires PreprocessorGPU, which needs MinkowskiEngine + PyTorch3D)."""
    import torch

    ns = load_regtr()
    cfg = _EasyDict(ns["misc"].load_config(os.path.join(REF_SRC, "conf", cfg_name)))
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = ns["regtr"].RegTR(cfg)
    model.preprocessor = ns["kpconv"].Preprocessor(cfg)
    model.eval()
    return model, cfg
