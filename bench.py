#!/usr/bin/env python3
"""Headline benchmark: point-cloud pairs/s (16 384 pts/cloud) through the full
hot path -- KPConv pyramid preprocessing, KPConv encoder, superpoint
self/cross attention, matching and the weighted-SVD pose -- on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one forward of `--pairs-per-step` synthetic pairs per rank, inputs
already resident in HBM.  Pairs are independent, so ranks shard the pair
stream with no data-path collective (weak scaling); the only collectives are
the timing barrier and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline      HBM roofline of the dominant kernel (fused KPConv gather):
                algorithmic bytes / HIP-event duration measured inside the
                timed region on the launch stream
  cpu_baseline  the CPU oracle (port of the reference path) timed on this
                host's cores on a bounded sample (four pairs), N == 1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA
MFMA_F32_PEAK_TFS = 157.0   # exact-f32 MFMA (v_mfma_f32_32x32x2_f32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-step", type=int, default=64,
                    help="pairs per forward and rank.  Round 5: 64 (one box, same build: 32 -> 1 111, 64 -> 1 164, "
                         "96 -> 1 170, 128 -> 1 180 pairs/s: per-launch tails and the per-forward host work amortise "
                         "over more pairs; rounds 3-4 used 32, rounds 1-2 16)")
    ap.add_argument("--no-cross-step-overlap", action="store_true",
                    help="make every forward's pyramid wait for the previous forward's tail (default: inputs "
                         "are resident, so consecutive steps pipeline on the GPU)")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent group forwards per step, each on its own HIP stream (streams.py); 2 with "
                         "--pairs-per-step 32 was the highest-throughput setting of round 1, but co-running kernels "
                         "stretch each other, so the per-kernel roofline leg is only meaningful at 1")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--config", default="3dmatch")
    ap.add_argument("--generator", default="box", choices=["box", "lidar"],
                    help="box: three faces of a 2 m box (--points per cloud, the headline workload); lidar: the "
                         "KITTI-shaped 120 k-return scan of synthetic.make_lidar_pair, pre-voxelised at 0.3 m "
                         "(BASELINE configs[3]; use with --config kitti)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-mode", type=int, default=1, help="1 = range-scaled split-fp16 MFMA (default), 0 = exact f32")
    ap.add_argument("--attn-mode", type=int, default=4,
                    help="4 = split-fp16, second probability plane only where a key block carries weight (default), "
                         "1 = split-fp16 everywhere, 3 = one probability plane, 0 = exact f32, 2 = single-pass fp16 operands")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the training-step leg (N = 1 only)")
    ap.add_argument("--train-pairs", type=int, default=16,
                    help="pairs per training step of the train_step leg (16 fill the chip: 4 pairs run at 0.7x the rate)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the short exact-f32 and fp16-attention legs that follow the headline timing")
    ap.add_argument("--canonical-order", action="store_true",
                    help="ascending-voxel-key point order instead of the reference's hash-map order")
    ap.add_argument("--skip-upsamples", action="store_true",
                    help="skip the up-sampling neighbour search the encoder never reads")
    return ap.parse_args()


def kpconv_alg_bytes(meta, model):
    """ALGORITHMIC bytes of every MFMA-path KPConv launch of one forward, in
    launch order:
       Kv*(4 + 12 + 4*Cin) + Nq*(12 + 4*Cout) + 60*Cin*Cout
    Kv = valid (non-shadow) neighbour entries."""
    out = []
    for blk in model.kpf_encoder.encoder_blocks:
        conv = blk.KPConv
        li = blk.layer_ind
        strided = 'strided' in blk.block_name
        idx = meta['_i32'][('pools' if strided else 'neighbors', li)]
        ns = meta['points'][li].shape[0]
        nq = idx.shape[0]
        kv = int((idx < ns).sum().item())
        cin, cout = conv.in_channels, conv.out_channels
        b = kv * (4 + 12 + 4 * cin) + nq * (12 + 4 * cout) + 60 * cin * cout
        out.append(dict(code=cin * 100000 + cout, nq=nq, kv=kv, cin=cin, cout=cout, bytes=b))
    return out


def cpu_baseline(cfg, model_sd, n_points):
    """The reference path restated on the CPU (oracle/), timed on a few pairs.
    Native preprocessing: the reference's own C++ (oracle/_ref, kd-tree) when
    its built library travelled with the repo, else our brute-force C port."""
    from oracle import native, torch_oracle
    from superpoints_registration_amd import synthetic
    # the GPU box gives one GPU a 16-core share; never oversubscribe it
    threads = min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 16)
    torch.set_num_threads(threads)
    npairs = 4   # bounded sample: ~10-15 s of CPU work on the GPU box's 16-core share
    pairs = [synthetic.make_pair(n_points, seed=i) for i in range(npairs)]
    use_ref = native.ref_available()
    if use_ref:
        orig_sub, orig_nb = native.grid_subsample, native.radius_neighbors

        def nb(q, s, ql, sl, r, limit=0):
            full = native.ref_radius_neighbors(q, s, ql, sl, r)
            mc = full.shape[1]
            return (full[:, :limit] if limit and limit < mc else full), mc

        native.grid_subsample = lambda p, l, dl, max_p=0, order="reference", return_keys=False: \
            native.ref_grid_subsample(p, l, dl, max_p)
        native.radius_neighbors = nb
    t0 = time.perf_counter()
    with torch.no_grad():
        for src, tgt, _ in pairs:      # one pair per forward, like the reference's CPU test loop
            torch_oracle.regtr_forward(cfg, model_sd, [src], [tgt])
    dt = time.perf_counter() - t0
    if use_ref:
        native.grid_subsample, native.radius_neighbors = orig_sub, orig_nb
    return dict(value=npairs / dt, unit="pairs/s", cores=threads, kind="port", pairs_per_forward=1,
                sample=f"{npairs} pairs x {n_points} pts/cloud, one forward each (B = 1), {dt:.1f} s in total; torch part on "
                       f"{threads} threads, native preprocessing single-threaded "
                       f"({'reference C++ via oracle/_ref' if use_ref else 'brute-force C port'})")


def train_leg(cfg, args, dev):
    """One optimisation step of the full model on `--train-pairs` synthetic pairs of the bench size:
    RegTR.forward (train mode, autograd graph over the HIP kernels) -> compute_loss -> backward
    (autograd.py: hand-written HIP backward kernels) -> clip_grad_norm_ -> AdamW -> scheduler, i.e.
    training.Trainer.train_step; per-stage times from HIP events.  Labels: the generator's pose and
    a fixed ~60 % per-point overlap pattern (compute_loss only consumes them)."""
    import numpy as np
    from superpoints_registration_amd import synthetic
    from superpoints_registration_amd.regtr import RegTR
    from superpoints_registration_amd.training import Trainer
    Bt = max(1, args.train_pairs)
    model = RegTR(cfg)
    synthetic.fill_parameters(model, seed=0)
    model = model.to(dev)
    pairs = [synthetic.make_pair(args.points, seed=900 + i) for i in range(Bt)]
    rng = np.random.default_rng(777)
    batch = {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs],
             "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs],
             "pose": torch.from_numpy(np.stack([p[2] for p in pairs]).astype(np.float32)).to(dev),
             "src_overlap": [torch.from_numpy(rng.random(len(p[0])) < 0.6).to(dev) for p in pairs],
             "tgt_overlap": [torch.from_numpy(rng.random(len(p[1])) < 0.6).to(dev) for p in pairs]}
    tr = Trainer(cfg).setup(model)
    tr.train_step(model, dict(batch))          # warm-up: workspaces, autotuned nothing, optimizer state
    torch.cuda.synchronize()
    steps, stage = 2, [0.0, 0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        model.train()
        bb = dict(batch)
        ev[0].record()
        pred = model(bb)
        ev[1].record()
        losses = model.compute_loss(pred, bb)
        tr.optimizer.zero_grad()
        ev[2].record()
        losses["total"].backward()
        ev[3].record()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=tr.grad_clip)
        tr.optimizer.step()
        tr.scheduler.step()
        ev[4].record()
        torch.cuda.synchronize()
        for k in range(4):
            stage[k] += ev[k].elapsed_time(ev[k + 1])
    dt = time.perf_counter() - t0
    return dict(value=round(Bt * steps / dt, 3), unit="pairs/s", ms_per_step=round(1e3 * dt / steps, 2), steps=steps,
                pairs_per_step=Bt, points_per_cloud=args.points,
                note=f"--train-pairs {Bt}: BENCH_r04 and later use 16 pairs per step, r01-r03 used 4 (not comparable)",
                stage_ms=dict(forward=round(stage[0] / steps, 2), loss=round(stage[1] / steps, 2),
                              backward=round(stage[2] / steps, 2), clip_adamw=round(stage[3] / steps, 2)),
                loss_total=round(float(losses["total"].detach()), 5),
                arithmetic="forward as the headline (operator route: the fused cross-encoder chains are inference only); "
                           "backward: dX / dW of the projections, the KPConv dW and d(weighted features) and the "
                           "flash-style attention backward (csrc/attention_bwd.hip) all in the forward's range-scaled "
                           "split-fp16 arithmetic; the loss-head products (InfoNCE, matching) exact f32 MFMA")


def main():
    args = parse()
    from superpoints_registration_amd import sharding
    rank, local_rank, world = sharding.rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)

    from superpoints_registration_amd import _lib, get_config, ops, synthetic
    from superpoints_registration_amd.regtr import RegTR

    cfg = get_config(args.config)
    model = RegTR(cfg, compute_upsamples=not args.skip_upsamples,
                  order=ops.ORDER_CANONICAL if args.canonical_order else ops.ORDER_REFERENCE)
    synthetic.fill_parameters(model, seed=0)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    # the synthetic pairs are uploaded and synchronised before the timed region and never change:
    # the pyramid of step i+1 (side stream) may overlap the tail of step i (regtr.py)
    model.inputs_resident = not args.no_cross_step_overlap

    B = args.pairs_per_step
    if args.generator == "lidar":
        pairs = [synthetic.make_lidar_pair(seed=sd)[:2] + (None,) for sd in sharding.pair_seeds(rank, B)]
    else:
        pairs = [synthetic.make_pair(args.points, seed=sd) for sd in sharding.pair_seeds(rank, B)]
    batch = {"src_xyz": [torch.from_numpy(p[0]).to(dev) for p in pairs],
             "tgt_xyz": [torch.from_numpy(p[1]).to(dev) for p in pairs]}

    from superpoints_registration_amd.streams import StreamedForward
    runner = StreamedForward(model, n_streams=max(1, args.streams), device=dev)
    ops.set_gemm_mode(args.gemm_mode)
    ops.set_attn_mode(args.attn_mode)

    def step():
        with torch.no_grad():
            return runner(batch)   # leaves the pyramid(s) in batch['kpconv_meta']

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    L = _lib.lib()

    # ---- timed region: barrier + sync, K steps, sync + barrier, MAX over ranks ----
    L.spr_prof_enable(1)
    elapsed, own = sharding.timed_steps(step, args.steps, dist=dist, sync=torch.cuda.synchronize, device=dev,
                                        return_own=True)
    per_rank_ms = sharding.gather_ms(1e3 * own / args.steps, dist=dist, device=dev)   # every rank's own time

    # ---- roofline of the dominant kernel (rank 0) -----------------------------
    roofline = None
    roofline_attn = None
    attn_flops_per_call = None
    if rank == 0:
        cap = 256 * max(args.steps, 1)
        codes = (ctypes.c_int * cap)()
        nqs = (ctypes.c_int * cap)()
        ms = (ctypes.c_float * cap)()
        n = L.spr_prof_read(cap, codes, nqs, ms)
        metas = batch['kpconv_meta'] if isinstance(batch['kpconv_meta'], list) else [batch['kpconv_meta']]
        per_fwd = [r for meta in metas for r in kpconv_alg_bytes(meta, model)]   # incl. the cin == 1 kernel
        # aggregate per kernel instantiation (cin, cout) == one rocprof kernel name
        agg = {}
        for i in range(n):
            a = agg.setdefault(codes[i], dict(ms=0.0, count=0))
            a['ms'] += ms[i]
            a['count'] += 1
        fwd_bytes, fwd_launches = {}, {}
        for r in per_fwd:
            fwd_bytes[r['code']] = fwd_bytes.get(r['code'], 0) + r['bytes']
            fwd_launches[r['code']] = fwd_launches.get(r['code'], 0) + 1
        # ---- second leg: the attention core (the largest single kernel by GPU time), MFMA roofline
        attn_ms = [ms[i] for i in range(n) if codes[i] == -1]
        roofline_attn = None
        if attn_ms:
            d_model = cfg.d_embed
            flops_fwd = []          # per forward: self + cross call of every layer do the same work pattern
            for meta in metas:
                lens = [int(v) for v in meta['_lens_host'][-1]]
                Bm = len(lens) // 2
                self_f = sum(4.0 * d_model * L * L for L in lens)
                cross_f = sum(4.0 * d_model * lens[b] * lens[Bm + b] * 2 for b in range(Bm))
                flops_fwd.append((self_f, cross_f))
            avg_flops = sum(sf + cf for sf, cf in flops_fwd) / (2.0 * len(flops_fwd))   # per call
            attn_flops_per_call = avg_flops
            avg_ms_attn = sum(attn_ms) / len(attn_ms)
            tfs = avg_flops / (avg_ms_attn * 1e-3) / 1e12
            roofline_attn = dict(bound="mfma", achieved=round(tfs, 2), peak=MFMA_F16_PEAK_TFS, unit="TFLOP/s",
                                 frac=round(tfs / MFMA_F16_PEAK_TFS, 5), traffic=None,
                                 kernel={1: "k_attn_s<true, true> (varlen attention core, split-fp16: 3 MFMA per product)",
                                         2: "k_attn_s<false, false> (varlen attention core, single-pass fp16 operands)",
                                         3: "k_attn_s<true, false> (split-fp16 scores, one probability plane)",
                                         4: "k_attn_s<true, true, ADAPT> (split-fp16, lo plane of P on significant tiles only)",
                                         0: "k_attn (exact f32 MFMA)"}[args.attn_mode],
                                 avg_launch_ms=round(avg_ms_attn, 5), launches=len(attn_ms),
                                 alg_flops_per_launch=int(avg_flops),
                                 mfma_flops_executed_per_launch=int({1: 3, 2: 1, 0: 1, 3: 2.5, 4: 2.75}[args.attn_mode] * avg_flops))
        best = None
        for code, a in agg.items():
            if code not in fwd_bytes or a['count'] % fwd_launches[code] != 0:
                continue
            n_fwd = a['count'] // fwd_launches[code]
            cand = dict(code=code, total_ms=a['ms'], count=a['count'],
                        total_bytes=fwd_bytes[code] * n_fwd)
            if best is None or cand['total_ms'] > best['total_ms']:
                best = cand
        if best is not None:
            avg_ms = best['total_ms'] / best['count']
            bytes_per_launch = best['total_bytes'] / best['count']
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            # HBM traffic comes from separate rocprofv3 --pmc passes (scripts/measure_round.sh ->
            # profiles/kpconv_traffic.json); it is reported only with its provenance and only while
            # the kernel duration recorded with it agrees with this run's (else: stale -> null)
            traffic, traffic_src = None, None
            tpath = os.path.join(REPO, "profiles", "kpconv_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    ref_ms = tj.get("kernel_avg_ms")
                    same_kernel = tj.get("code") == best['code']
                    if same_kernel and ref_ms and abs(ref_ms - avg_ms) <= 0.15 * avg_ms:
                        traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = dict(file="profiles/kpconv_traffic.json", tag=tj.get("tag"),
                                       collected_utc=tj.get("collected_utc"), kernel_avg_ms_then=ref_ms,
                                       stale=traffic is None)
                except Exception:
                    traffic, traffic_src = None, None
            cin, cout = best['code'] // 100000, best['code'] % 100000
            roofline = dict(code=best['code'], bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                            kernel=(f"k_kpconv_ring<{cin},{cout}> (fused KPConv: LDS-DMA neighbour gather, register-resident weights)"
                                    if cin in (32, 64) and cin * cout <= 4096 else
                                    f"k_kpconv_mfma (fused KPConv gather, streamed weights) cin={cin} cout={cout}"),
                            avg_launch_ms=round(avg_ms, 5), launches=best['count'],
                            alg_bytes_per_launch=int(bytes_per_launch), traffic_source=traffic_src,
                            pooled_all_variants=dict(
                                gbs=round(sum(fwd_bytes[c] * (a['count'] // fwd_launches[c]) for c, a in agg.items()
                                              if c in fwd_bytes) / (sum(a['ms'] for c, a in agg.items()
                                                                        if c in fwd_bytes) * 1e-3) / 1e9, 1)),
                            all_kpconv_variants={f"{c // 100000}->{c % 100000}": dict(
                                launches=a['count'], avg_ms=round(a['ms'] / a['count'], 5),
                                gbs=round(fwd_bytes[c] / fwd_launches[c] / (a['ms'] / a['count'] * 1e-3) / 1e9, 1))
                                for c, a in agg.items() if c in fwd_bytes})
    L.spr_prof_enable(0)

    # ---- extra legs (N == 1): the same workload in exact-f32 arithmetic and with single-pass
    # fp16 attention operands; short, after the headline timing, never part of `value`
    extra = {}
    if world == 1 and not args.no_extra_legs:
        def leg(gm, am, steps):
            ops.set_gemm_mode(gm)
            ops.set_attn_mode(am)
            step()
            torch.cuda.synchronize()
            L.spr_prof_enable(1)
            t = sharding.timed_steps(step, steps, dist=None, sync=torch.cuda.synchronize, device=dev)
            cap_ = 256 * steps
            c_, q_, m_ = (ctypes.c_int * cap_)(), (ctypes.c_int * cap_)(), (ctypes.c_float * cap_)()
            n_ = L.spr_prof_read(cap_, c_, q_, m_)
            L.spr_prof_enable(0)
            res = dict(value=round(sharding.throughput(B, steps, 1, t), 3), unit="pairs/s",
                       ms_per_step=round(1e3 * t / steps, 3), steps=steps, gemm_mode=gm, attn_mode=am)
            a_ms = [m_[i] for i in range(n_) if c_[i] == -1]
            if a_ms and attn_flops_per_call:
                avg = sum(a_ms) / len(a_ms)
                tfs = attn_flops_per_call / (avg * 1e-3) / 1e12
                peak = MFMA_F16_PEAK_TFS if am != 0 else MFMA_F32_PEAK_TFS
                res["attention_core"] = dict(avg_launch_ms=round(avg, 5), launches=len(a_ms), achieved=round(tfs, 2),
                                             peak=peak, unit="TFLOP/s", frac=round(tfs / peak, 5))
            return res
        n_leg = max(1, min(args.steps, 4))
        extra["exact_f32"] = leg(0, 0, n_leg)
        extra["fp16_attention"] = leg(args.gemm_mode, 2, n_leg)
        # round 5: split-fp16 scores with ONE probability plane (spr_set_attn_mode(3)): 5e-6 of the feature scale away
        # from the exact-f32 forward on this workload, up to ~2e-4 when the softmax rows are carried by a few keys
        # (scripts/attn_mode_err.py) -- an opt-in mode, never the headline
        extra["single_plane_attention"] = leg(args.gemm_mode, 3, n_leg)
        # rounds 1-4 default (both probability planes on every tile); the round-5 default (mode 4) keeps the second
        # plane only for key blocks that hold a weight >= 2^-5 of the row's running sum
        extra["two_plane_attention"] = leg(args.gemm_mode, 1, n_leg)
        ops.set_gemm_mode(args.gemm_mode)
        ops.set_attn_mode(args.attn_mode)
        if roofline is not None:
            # the roofline kernel WITHOUT the index work of the pyramid running beside it (the
            # product default builds the pyramid on a side stream, regtr.py): same launches, one stream
            from superpoints_registration_amd.regtr import no_side_stream
            code = int(roofline["code"])
            with no_side_stream():
                step()
                torch.cuda.synchronize()
                L.spr_prof_enable(1)
                sharding.timed_steps(step, n_leg, dist=None, sync=torch.cuda.synchronize, device=dev)
                cap_ = 256 * n_leg
                c_, q_, m_ = (ctypes.c_int * cap_)(), (ctypes.c_int * cap_)(), (ctypes.c_float * cap_)()
                n_ = L.spr_prof_read(cap_, c_, q_, m_)
                L.spr_prof_enable(0)
            iso = [m_[i] for i in range(n_) if c_[i] == code]
            if iso:
                avg = sum(iso) / len(iso)
                gbs = roofline["alg_bytes_per_launch"] / (avg * 1e-3) / 1e9
                roofline["single_stream"] = dict(avg_launch_ms=round(avg, 5), launches=len(iso), achieved=round(gbs, 2),
                                                 frac=round(gbs / HBM_PEAK_GBS, 5))
        if args.streams <= 1:
            # two concurrent forwards of B pairs each on two HIP streams (streams.StreamedForward).  Since
            # round 2 the default forward overlaps the same work inside itself (side stream, shortcut
            # branches, cross-step overlap) and is the faster setting; kept as a comparison leg
            runner2 = StreamedForward(model, n_streams=2, device=dev)
            batch2 = {"src_xyz": batch["src_xyz"] + batch["src_xyz"], "tgt_xyz": batch["tgt_xyz"] + batch["tgt_xyz"]}

            def step2():
                with torch.no_grad():
                    return runner2(batch2)
            step2()
            torch.cuda.synchronize()
            t2 = sharding.timed_steps(step2, n_leg, dist=None, sync=torch.cuda.synchronize, device=dev)
            extra["two_streams"] = dict(value=round(sharding.throughput(2 * B, n_leg, 1, t2), 3), unit="pairs/s",
                                        ms_per_step=round(1e3 * t2 / n_leg, 3), steps=n_leg, streams=2,
                                        pairs_per_step=2 * B)

        # ---- training step: forward + loss + HIP backward + clip + AdamW, the
        # reference's trainer.py:107-124 order, at BASELINE size.  Not part of `value`.
        if not args.no_train_leg:
            extra["train_step"] = train_leg(cfg, args, dev)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # labels follow the arguments actually used (the driver's default command is BASELINE configs[1])
    headline = (args.points == 16384 and args.config == "3dmatch" and args.generator == "box")
    if args.generator == "lidar":
        cloud_desc = "LiDAR-shaped 120 k-return scans pre-voxelised at 0.3 m (synthetic.make_lidar_pair)"
        metric = "point-cloud pairs/sec (KITTI-shaped scans, ~120 k returns/cloud)"
    else:
        cloud_desc = f"synthetic {args.points}-pt pairs"
        metric = ("point-cloud pairs/sec (16 384 pts/cloud)" if args.points == 16384
                  else f"point-cloud pairs/sec ({args.points} pts/cloud)")
    baseline_cfg = ("BASELINE configs[1]" if headline else
                    {"kitti": "BASELINE configs[3] shape", "modelnet": "BASELINE configs[4] shape"}.get(
                        args.config, "BASELINE configs[2] shape" if args.config == "3dmatch" else "parity-case shape"))
    result = {
        "metric": metric,
        "value": round(sharding.throughput(B, args.steps, world, elapsed), 3),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {(1, 4): "f32 (matrix products: range-scaled split-f16x2 MFMA, f32 accumulate; attention weights below "
                          "2^-5 of their row sum carried in one f16 plane)",
                  (1, 1): "f32 (matrix products: range-scaled split-f16x2 MFMA, f32 accumulate)",
                  (0, 0): "f32 (exact f32 MFMA)"}.get((args.gemm_mode, args.attn_mode),
                                                      f"f32 storage/accumulate, gemm_mode={args.gemm_mode}, "
                                                      f"attn_mode={args.attn_mode}"),
        "data": "synthetic",
        "config": {"workload": f"{cloud_desc}, full KPConv backbone + superpoint attn "
                               f"+ {'Sinkhorn-' if cfg.use_sinkhorn else ''}SVD pose ({args.config} config, "
                               f"{baseline_cfg})",
                   "pairs_per_step_per_gpu": B, "streams_per_gpu": max(1, args.streams),
                   "points_per_cloud": (args.points if args.generator == "box" else "lidar scan (varies)"),
                   "generator": args.generator,
                   "point_order": "canonical" if args.canonical_order else "reference",
                   "upsample_indices": not args.skip_upsamples,
                   "cross_step_overlap": not args.no_cross_step_overlap,
                   "streams_inside_a_forward": ("1 (SPR_NO_SIDE_STREAM)" if os.environ.get("SPR_NO_SIDE_STREAM", "0") == "1"
                                                else "3: main path, pyramid searches, ResNet shortcut branches"),
                   "parallelism": f"pairs sharded over {world} rank(s), no data-path collective; per rank "
                                  f"{max(1, args.streams)} concurrent forwards of {B // max(1, args.streams)} pairs"},
        "roofline": roofline,
        "roofline_attention": roofline_attn,
        "exact_f32": extra.get("exact_f32"),
        "fp16_attention": extra.get("fp16_attention"),
        "single_plane_attention": extra.get("single_plane_attention"),
        "two_plane_attention": extra.get("two_plane_attention"),
        "two_streams": extra.get("two_streams"),
        "train_step": extra.get("train_step"),
        "ranks": {"world_size_seen": (dist.get_world_size() if dist is not None else 1),
                  "backend": (dist.get_backend() if dist is not None else None),
                  "ms_per_step_per_rank": per_rank_ms},
    }
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, sd_cpu, args.points)
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
