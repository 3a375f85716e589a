"""GPU: one training step of the real model through training.Trainer with the RCCL ("nccl")
process group initialised the way the reference's train.py does (one process per GPU; a
single-rank group here -- the 1-GPU box).  Exercises GradientSync's bucketed asynchronous
all-reduce on device tensors behind the HIP backward, which the CPU suite can only run over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle.gen_golden import loss_inputs, pairs_for
from superpoints_registration_amd import get_config, synthetic
from superpoints_registration_amd.regtr import RegTR
from superpoints_registration_amd.training import GradientSync, Trainer

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_train_step_over_rccl_single_rank(device):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
                      WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        tag, B = "3dmatch", 2
        cfg = get_config(tag)
        pairs, sizes = pairs_for(tag, B)
        pose, src_ov, tgt_ov = loss_inputs(tag, B)
        batch = {"src_xyz": [T(p[0][:n]).to(device) for p, (n, m) in zip(pairs, sizes)],
                 "tgt_xyz": [T(p[1][:m]).to(device) for p, (n, m) in zip(pairs, sizes)],
                 "pose": T(pose).to(device),
                 "src_overlap": [T(o).to(device) for o in src_ov], "tgt_overlap": [T(o).to(device) for o in tgt_ov]}

        def run(sync):
            model = RegTR(cfg)
            synthetic.fill_parameters(model, seed=0)
            model = model.to(device)
            tr = Trainer(cfg, rank=0, world=1).setup(model)
            if sync:   # a one-rank group still goes through the bucketed RCCL all-reduce
                tr.sync = GradientSync(model, None, bucket_bytes=4 << 20)
                assert tr.sync.active and len(tr.sync.buckets) > 1 and all(b.is_cuda for b in tr.sync.buckets)
            losses = tr.train_step(model, dict(batch))
            torch.cuda.synchronize()
            if sync:   # every bucket really went through dist.all_reduce on the RCCL group
                assert tr.sync.n_reduced == len(tr.sync.buckets), (tr.sync.n_reduced, len(tr.sync.buckets))
                losses = tr.train_step(model, dict(batch))     # second step: launched from the hooks
                torch.cuda.synchronize()
                assert tr.sync.n_reduced == len(tr.sync.buckets)
                return float(losses["total"]), None
            return float(losses["total"]), {n: p.detach().clone() for n, p in model.named_parameters()}
        run(True)
        # parameters after ONE step, with and without the synchroniser
        def one(sync):
            model = RegTR(cfg)
            synthetic.fill_parameters(model, seed=0)
            model = model.to(device)
            tr = Trainer(cfg, rank=0, world=1).setup(model)
            if sync:
                tr.sync = GradientSync(model, None, bucket_bytes=4 << 20)
            losses = tr.train_step(model, dict(batch))
            torch.cuda.synchronize()
            return float(losses["total"]), {n: p.detach().clone() for n, p in model.named_parameters()}
        l_sync, p_sync = one(True)
        l_ref, p_ref = one(False)
        assert abs(l_sync - l_ref) <= 1e-6 * max(1.0, abs(l_ref))
        # the mean over one rank is the gradient itself: the optimizer step must land on the same
        # parameters (the backward's float atomics make the last bits run-dependent)
        for n in p_ref:
            d = float((p_sync[n] - p_ref[n]).abs().max())
            assert d <= 2e-6, (n, d)
    finally:
        dist.destroy_process_group()
