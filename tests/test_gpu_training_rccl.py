"""GPU: one training step of the real model through training.Trainer with the RCCL ("nccl")
process group initialised the way the reference's train.py does (one process per GPU).  On the
1-GPU box the group has one rank: GradientSync still issues every bucket's dist.all_reduce on
device tensors behind the HIP backward (asserted through sync.n_reduced).  The second test
spawns TWO ranks over RCCL when the node has at least two GPUs (skipped otherwise): the bench
timing protocol and the gradient all-reduce across devices, so that the N > 1 path is not first
executed by the driver's scaling run."""
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


def _two_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from superpoints_registration_amd import sharding
    # (a) the bench protocol: barrier + sync brackets, MAX over ranks, per-rank times gathered on the device
    x = torch.ones(1 << 20, device=dev)

    def step():
        for _ in range(10 * (rank + 1)):
            x.mul_(1.0000001)
    elapsed, own = sharding.timed_steps(step, 5, dist=dist, sync=torch.cuda.synchronize, device=dev, return_own=True)
    per_rank = sharding.gather_ms(1e3 * own / 5, dist=dist, device=dev)
    # (b) bucketed gradient all-reduce over RCCL: different data per rank -> identical mean gradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 32)).to(dev)
    sync = GradientSync(model, None, bucket_bytes=16 << 10)
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    xb, tb = torch.randn(48, 64, generator=g).to(dev), torch.randn(48, 32, generator=g).to(dev)
    grads = []
    for step_i in range(2):                              # step 0 learns the firing sets, step 1 launches from the hooks
        sync.zero_grad()
        ((model(xb) - tb) ** 2).mean().backward()
        sync.finish()
        assert sync.n_reduced == len(sync.buckets) > 1
        grads.append([p.grad.detach().cpu().clone() for p in model.parameters()])
    q.put((rank, elapsed, per_rank, grads, [xb.cpu(), tb.cpu()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_rccl_bench_protocol_and_gradient_sync():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on the node (the 1-GPU box runs the single-rank test below)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: rest for r, *rest in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (e0, pr0, g0, d0), (e1, pr1, g1, d1) = res[0], res[1]
    assert e0 == e1 and pr0 == pr1 and len(pr0) == 2 and max(pr0) * 5 / 1e3 <= e0 * 1.001
    # single-process replay: mean of the two ranks' local gradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 32))
    local = []
    for xb, tb in (d0, d1):
        model.zero_grad(set_to_none=True)
        ((model(xb) - tb) ** 2).mean().backward()
        local.append([p.grad.clone() for p in model.parameters()])
    want = [0.5 * (a + b) for a, b in zip(*local)]
    for step_i in range(2):
        for got0, got1, w in zip(g0[step_i], g1[step_i], want):
            assert torch.equal(got0, got1)                              # ranks hold the same reduced gradient
            assert torch.allclose(got0, w, rtol=1e-4, atol=1e-6)        # == the mean (device matmul vs CPU)


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
        assert l_sync == l_ref
        # the mean over one rank is the gradient itself and the whole backward is bitwise reproducible
        # (order-independent fixed-point scatter-adds, fixed summation orders): the optimizer step must land on
        # EXACTLY the same parameters -- a determinism regression fails here
        for n in p_ref:
            assert torch.equal(p_sync[n], p_ref[n]), (n, float((p_sync[n] - p_ref[n]).abs().max()))
    finally:
        dist.destroy_process_group()
