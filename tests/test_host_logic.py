"""CPU: host-side logic of the package (no kernels)."""
import os

import numpy as np
import torch

from superpoints_registration_amd import get_config, synthetic
from superpoints_registration_amd.kpconv import plan_pyramid
from superpoints_registration_amd.regtr import RegTR
from superpoints_registration_amd.seq_manipulation import pad_sequence, split_src_tgt, unpad_sequences
from superpoints_registration_amd import se3


def test_pyramid_plan_matches_reference_block_dims():
    # SURVEY.md section 8b / Appendix B (printed from the reference's RegTR(cfg))
    blocks, levels, final = plan_pyramid(get_config("3dmatch"))
    kp = [(b.in_dim if "simple" in b.name else b.out_dim // 4, b.out_dim // 2 if "simple" in b.name else b.out_dim // 4)
          for b in blocks]
    assert kp == [(1, 64), (32, 32), (32, 32), (64, 64), (64, 64), (64, 64), (128, 128), (128, 128)]
    assert [round(l.radius, 6) for l in levels] == [0.0625, 0.125, 0.25]
    assert [l.down for l in levels] == [True, True, False] and final == 512
    blocks, levels, final = plan_pyramid(get_config("kitti"))
    assert len(blocks) == 11 and len(levels) == 4 and final == 1024
    assert [l.limit for l in levels] == [39, 57, 68, 74]
    blocks, levels, final = plan_pyramid(get_config("modelnet"))
    assert len(levels) == 2 and final == 1024 and blocks[1].out_dim == 512


def test_state_dict_names_and_parameter_counts():
    # B4 of SURVEY.md: 151 tensors / 7 797 547 parameters for the 3DMatch config
    m = RegTR(get_config("3dmatch"))
    sd = m.state_dict()
    assert len(sd) == 151
    assert sum(v.numel() for v in sd.values()) == 7797547
    for name, shape in {
        "alpha": (), "beta": (),
        "kpf_encoder.encoder_blocks.0.KPConv.weights": (15, 1, 64),
        "kpf_encoder.encoder_blocks.0.KPConv.kernel_points": (15, 3),
        "kpf_encoder.encoder_blocks.1.unary1.mlp.weight": (32, 64),
        "kpf_encoder.encoder_blocks.1.unary_shortcut.mlp.weight": (128, 64),
        "kpf_encoder.encoder_blocks.7.unary2.mlp.weight": (512, 128),
        "feat_proj.weight": (256, 512),
        "transformer_encoder.layers.5.multihead_attn.in_proj_weight": (768, 256),
        "transformer_encoder.layers.0.self_attn.out_proj.bias": (256,),
        "transformer_encoder.layers.3.linear1.weight": (1024, 256),
        "transformer_encoder.norm.weight": (256,),
        "overlap_predictor.weight": (1, 256),
        "feature_criterion.W": (256, 256), "feature_criterion_un.W": (256, 256),
    }.items():
        assert tuple(sd[name].shape) == shape, name
    assert sum(v.numel() for v in RegTR(get_config("kitti")).state_dict().values()) == 11713458
    assert sum(v.numel() for v in RegTR(get_config("modelnet")).state_dict().values()) == 11355665


def test_fill_parameters_is_deterministic_and_name_keyed():
    a, b = RegTR(get_config("3dmatch")), RegTR(get_config("3dmatch"))
    synthetic.fill_parameters(a, 3)
    synthetic.fill_parameters(b, 3)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    synthetic.fill_parameters(b, 4)
    assert not torch.equal(a.state_dict()["feat_proj.weight"], b.state_dict()["feat_proj.weight"])


def test_padding_helpers_round_trip():
    seqs = [torch.randn(5, 4), torch.randn(2, 4), torch.randn(7, 4)]
    padded, mask, lens = pad_sequence(seqs, require_padding_mask=True, require_lens=True)
    assert padded.shape == (7, 3, 4) and lens == [5, 2, 7]
    assert mask.tolist()[1] == [False, False, True, True, True, True, True]
    back = unpad_sequences(padded.unsqueeze(0), lens)
    for s, r in zip(seqs, back):
        assert torch.equal(s, r[0])
    src, tgt = split_src_tgt(torch.arange(10).unsqueeze(1), [1, 2, 3, 4])
    assert [len(t) for t in src] == [1, 2] and [len(t) for t in tgt] == [3, 4]


def test_se3_algebra():
    R = torch.tensor(synthetic.rotation_z(0.3), dtype=torch.float32)
    T = se3.se3_init(R, torch.tensor([[0.1], [0.2], [0.3]]))
    I = se3.se3_cat(T, se3.se3_inv(T))
    assert torch.allclose(I, se3.se3_init(torch.eye(3), torch.zeros(3, 1)), atol=1e-6)
    x = torch.randn(5, 3)
    assert torch.allclose(se3.se3_transform(T, x), x @ R.t() + torch.tensor([0.1, 0.2, 0.3]), atol=1e-6)
    err = se3.se3_compare(T, se3.se3_init(torch.eye(3), torch.zeros(3, 1)))
    assert abs(float(err["rot_deg"]) - np.degrees(0.3)) < 1e-3


def test_synthetic_pairs_are_reproducible():
    a = synthetic.make_pair(256, seed=9)
    b = synthetic.make_pair(256, seed=9)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert a[0].dtype == np.float32 and a[0].shape == (256, 3)


def test_load_kernels_reproduces_the_reference_kernel_points():
    """a14: with the reference's RNG protocol (np.random.seed, one rand() for the z rotation,
    one normal(size=(15,3)) for the noise -- kernel_points.py:435-461) load_kernels returns the
    very kernel points the reference's KPConv constructor produced (goldens kp.*.kpts were
    captured from blocks.KPConv(...) after np.random.seed(5), radius 0.125)."""
    import numpy as np
    from conftest import load_golden
    from superpoints_registration_amd.kernel_points import load_kernels
    gold = load_golden("ops.npz")
    for tag in ("c1", "c32", "c64", "c128", "c48"):
        np.random.seed(5)
        kp = load_kernels(0.125, 15, dimension=3, fixed="center")
        assert kp.dtype == np.float32 and kp.shape == (15, 3)
        assert np.array_equal(kp.view(np.uint32), gold[f"kp.{tag}.kpts"].view(np.uint32)), tag
    # the KPConv module draws them the same way (frozen parameter, kpconv_blocks.py:244-266)
    from superpoints_registration_amd.kpconv_blocks import KPConv
    np.random.seed(5)
    conv = KPConv(15, 3, 32, 32, 0.1, 0.125)
    assert np.array_equal(conv.kernel_points.detach().numpy().view(np.uint32), gold["kp.c32.kpts"].view(np.uint32))
    assert not conv.kernel_points.requires_grad


def test_graft_entry_build_checks_the_header_version():
    """build() compares the library's version with include/spr.h (a literal there once broke the
    driver's build check after an ABI bump)."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    assert "SPR_VERSION" in src and re.search(r"spr_version\(\)\s*==\s*\d", src) is None


def test_match_index_arrays_follow_the_per_pair_rule():
    """regtr._match_index_arrays: the arg-max matches of a pair live on its tgt tokens when
    N > M and on its src tokens otherwise (qk_regtr_full.py:455-479, :563-588); the arrays built in
    one go must equal the per-pair construction."""
    from superpoints_registration_amd.regtr import RegTR
    B, n, m = 4, [5, 2, 4, 7], [3, 6, 4, 1]
    cu = [0]
    for v in n + m:
        cu.append(cu[-1] + v)
    off, own, flag, set_cu = RegTR._match_index_arrays(cu, B, torch.device("cpu"))
    exp_off = [cu[B + b] for b in range(B) for _ in range(n[b])] + [cu[b] for b in range(B) for _ in range(m[b])]
    exp_own, exp_flag, sc = [], [], [0]
    for b in range(B):
        on_tgt = n[b] > m[b]
        r = list(range(cu[B + b], cu[B + b + 1])) if on_tgt else list(range(cu[b], cu[b + 1]))
        exp_own += r
        exp_flag += [int(on_tgt)] * len(r)
        sc.append(sc[-1] + len(r))
    assert off.tolist() == exp_off and own.tolist() == exp_own
    assert flag.long().tolist() == exp_flag and set_cu.tolist() == sc


def test_streamed_encoder_launches_every_block_as_soon_as_its_level_exists():
    """KPFEncoder.forward_streamed: a non-strided block of level l runs right after ('conv', l), the
    strided block that ends the level after ('down', l) -- never earlier, and in architecture order."""
    from superpoints_registration_amd.kpconv import KPFEncoder, plan_pyramid
    cfg = get_config("3dmatch")
    enc = KPFEncoder(cfg, cfg.d_embed)
    plans, levels, _ = plan_pyramid(cfg)
    log = []

    class Rec(torch.nn.Module):
        def __init__(self, i):
            super().__init__()
            self.i = i

        def forward(self, x, meta):
            log.append(('block', self.i, meta['seen'][-1]))
            return x
    enc.encoder_blocks = torch.nn.ModuleList(Rec(i) for i in range(len(plans)))

    def events():
        meta = {'seen': []}
        for l, lv in enumerate(levels):
            meta['seen'].append(('conv', l))
            yield meta, 'conv', l
            meta['seen'].append(('down', l))
            yield meta, 'down', l
    x, skips, meta = enc.forward_streamed(torch.zeros(1), events())
    assert [b[1] for b in log] == list(range(len(plans)))           # architecture order
    for _, i, seen in log:
        bp = plans[i]
        want = ('down', bp.level) if bp.down else ('conv', bp.level)
        assert seen == want, (i, seen, want)                         # launched at the first possible event
    assert len(skips) == len(enc.encoder_skips)


def test_split_k_batches_fill_the_chip_and_depend_on_shapes_only():
    """autograd._tn_chunk: rows per split-K batch of a weight gradient L[rows, nl]^T R[rows, nr].  Enough batches
    that 128 x 128 tiles x batches reaches the CU count, at least 256 rows each, whole 16-deep K slabs -- and a
    pure function of the shapes (the summation order, hence the gradient bits, depends on nothing else)."""
    from superpoints_registration_amd.autograd import _tn_chunk
    for rows, nl, nr in ((15432, 256, 256), (15432, 1024, 256), (131072, 480, 32), (52994, 960, 64), (700, 256, 256),
                         (100, 64, 64), (61745, 256, 1024)):
        c = _tn_chunk(rows, nl, nr)
        assert c % 16 == 0 and c == _tn_chunk(rows, nl, nr)
        nchunk = (rows + c - 1) // c
        tiles = ((nl + 127) // 128) * ((nr + 127) // 128 if nr > 32 else 1)
        assert nchunk >= 1 and (nchunk - 1) * c < rows <= nchunk * c
        if rows >= 512:
            assert c >= 256                                     # never batches shorter than 256 rows ...
            assert tiles * nchunk >= min(200, tiles * (rows // 512))    # ... and a filled chip when the rows allow
        assert tiles * nchunk <= 1024
    assert _tn_chunk(15432, 256, 256) < 2048                    # (round 2 used 2 048-row batches: 32 workgroups)


def test_segment_permutation_and_its_inverse():
    """ops._segment_perms: kv_seg of the cross-attention (src <-> tgt swap) and its inverse, the query segment
    of every key segment, which the dK/dV kernel of the attention backward walks."""
    from superpoints_registration_amd import ops
    kv, inv = ops._segment_perms([4, 5, 6, 7, 0, 1, 2, 3], torch.device("cpu"))
    assert kv.dtype == torch.int32 and kv.tolist() == [4, 5, 6, 7, 0, 1, 2, 3] and inv.tolist() == kv.tolist()
    kv, inv = ops._segment_perms([2, 0, 3, 1], "cpu")
    assert inv.tolist() == [1, 3, 0, 2] and all(kv[inv[k]] == k for k in range(4))
    assert ops._segment_perms([2, 0, 3, 1], "cpu")[0] is kv     # cached per permutation and device


def test_prepared_encoder_plan_does_not_travel_with_copies():
    """ops.XencPlan (prepared weights of the fused cross-encoder chains) holds device addresses: a deep copy or a
    pickle of a module that carries one gets no plan and rebuilds its own on first use."""
    import copy
    import pickle
    from superpoints_registration_amd import ops
    plan = ops.XencPlan(("key",), object(), object(), [])
    holder = {"_spr_xenc": plan, "other": 3}
    assert copy.deepcopy(holder) == {"_spr_xenc": None, "other": 3}
    assert pickle.loads(pickle.dumps(plan)) is None
