"""CPU: host logic of streams.py (batch splitting / output merging)."""
import torch

from superpoints_registration_amd.streams import merge_outputs, split_batch


def test_split_batch_contiguous_groups_and_shared_entries():
    batch = {"src_xyz": list(range(5)), "tgt_xyz": list(range(10, 15)), "pose": "shared", "kpconv_meta": "stale"}
    subs = split_batch(batch, 2)
    assert [s["src_xyz"] for s in subs] == [[0, 1, 2], [3, 4]]
    assert [s["tgt_xyz"] for s in subs] == [[10, 11, 12], [13, 14]]
    assert all(s["pose"] == "shared" and "kpconv_meta" not in s for s in subs)
    assert len(split_batch(batch, 8)) == 5          # never more groups than pairs


def test_merge_outputs_keeps_pair_order():
    a = {"pose": torch.zeros(3, 3, 4), "src_feat": ["a0", "a1", "a2"], "ind_list": [0, 1, 2]}
    b = {"pose": torch.ones(2, 3, 4), "src_feat": ["b0", "b1"], "ind_list": [3, 4]}
    m = merge_outputs([a, b])
    assert m["pose"].shape == (5, 3, 4) and float(m["pose"][3:].min()) == 1.0
    assert m["src_feat"] == ["a0", "a1", "a2", "b0", "b1"] and m["ind_list"] == [0, 1, 2, 3, 4]
