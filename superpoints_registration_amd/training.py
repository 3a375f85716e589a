"""Training harness counterpart -- SURVEY.md section 8f row 2.

Mirrors what the reference's ``Trainer.fit`` (src/trainer.py:36-215),
``GenericRegModel`` (src/models/generic_reg_model.py:46-122, :294-372) and
``CheckPointManager`` (src/cvhelpers/torch_helpers.py:98-242) do AROUND the hot
path, so that the HIP model can be trained and validated with the reference's
semantics:

  * step order  training_step (forward + compute_loss) -> zero_grad -> backward
    -> [gradient all-reduce] -> clip_grad_norm_(grad_clip) -> optimizer.step ->
    scheduler.step                                  (trainer.py:107-124)
  * optimiser / scheduler construction from the flat config
                                                     (generic_reg_model.py:46-76)
  * metrics: se3_compare rotation / translation errors and their aggregation
    into reg_success / rot_success / trans_success   (generic_reg_model.py:294-372)
  * checkpoints: ``{'state_dict', 'step', 'optimizer', 'scheduler'}`` in
    ``<dir>/model-<step>.pth`` + ``checkpoints.txt`` ("Best step: N" first line),
    loaded with strict=False                         (torch_helpers.py:134-142, :222)

Multi-GPU: one process per GPU, pairs sharded over ranks (sharding.py).  The
ONE exchange step of training is the gradient all-reduce (BASELINE.json
north_star: "RCCL all-reduce over xGMI for gradient sync only").  The reference
wraps the model in DDP but then calls ``model.module.training_step`` (trainer.py:109),
which bypasses DDP's reducer, so its gradients are never synchronised (SURVEY
section 5); ``GradientSync`` below does what was intended, explicitly: parameter
gradients live as views into a few flat fp32 buckets and each bucket is
all-reduced (mean) as soon as its gradients are complete, overlapping the rest
of the backward.  Backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests.
"""
import logging
import math
import os
import time
from typing import Callable, Dict, Iterable, List, Optional

import torch

from .se3 import se3_compare

_log = logging.getLogger("spr.training")


# --------------------------------------------------------------------------------------------- #
class GradientSync:
    """Bucketed gradient all-reduce(mean) for one-process-per-GPU data parallelism.

    xGMI is point to point (7 links x ~153 GB/s per GPU): RCCL's ring / direct algorithms are
    bound per link, so few large messages beat many small ones -- the 31-47 MB of fp32
    gradients of the shipped configs go out in `bucket_bytes` (default 16 MiB) pieces, which
    also lets the first buckets travel while the encoder's backward is still running.

    Gradients are stored as views into the flat buckets (no pack / unpack copies).  Buckets are
    filled in REVERSE parameter order, the order in which the backward produces gradients.  A
    bucket is launched from a post-accumulate hook once every parameter that is expected to
    receive a gradient has one; which parameters those are is learned during the first step
    (loss-only or frozen tensors never fire), whose buckets are simply launched from finish().
    """

    def __init__(self, params, process_group=None, bucket_bytes: int = 16 << 20):
        """params: an nn.Module or an iterable of parameters.  The collective runs whenever a
        process group is initialised -- also for a one-rank group, so that a single-GPU box
        exercises the same RCCL path the N-GPU job takes."""
        import torch.distributed as dist
        self.dist = dist
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.active else 1
        if isinstance(params, torch.nn.Module):
            params = params.parameters()
        self.params = [p for p in params if p.requires_grad]
        self.buckets: List[torch.Tensor] = []
        self._slot = {}                      # param -> (bucket index, offset)
        self._members: List[List[torch.nn.Parameter]] = []
        cur, cur_members, cur_elems = [], [], 0
        for p in reversed(self.params):
            if cur_elems and (cur_elems + p.numel()) * 4 > bucket_bytes:
                self._close(cur, cur_members, cur_elems)
                cur, cur_members, cur_elems = [], [], 0
            cur.append((p, cur_elems))
            cur_members.append(p)
            cur_elems += p.numel()
        if cur_elems:
            self._close(cur, cur_members, cur_elems)
        self._expected: Optional[List[set]] = None      # learned on the first step
        self._fired: List[set] = [set() for _ in self.buckets]
        self._handles: Dict[int, object] = {}
        self._late: List[torch.nn.Parameter] = []       # fired after their bucket had left
        self.n_reduced = 0                              # collectives issued by the last finish()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _close(self, entries, members, elems):
        p0 = entries[0][0]
        flat = torch.zeros(elems, dtype=torch.float32, device=p0.device)
        bi = len(self.buckets)
        self.buckets.append(flat)
        self._members.append(members)
        for p, off in entries:
            self._slot[p] = (bi, off)
            p.grad = flat[off:off + p.numel()].view_as(p)

    def _view(self, p):
        bi, off = self._slot[p]
        return self.buckets[bi][off:off + p.numel()].view_as(p)

    # -- step protocol ------------------------------------------------------------------------
    def zero_grad(self):
        """Replaces optimizer.zero_grad(): zeroes the buckets and (re-)points the gradients of the
        parameters that take part at their bucket views.  Parameters known never to receive a
        gradient keep grad = None, as they do without GradientSync (AdamW then skips them)."""
        for b in self.buckets:
            b.zero_()
        for p in self.params:
            bi, _ = self._slot[p]
            if self._expected is not None and p not in self._expected[bi]:
                p.grad = None
                continue
            want = self._view(p)
            if p.grad is None or p.grad.data_ptr() != want.data_ptr():
                p.grad = want
        self._fired = [set() for _ in self.buckets]
        self._handles = {}
        self._late = []

    def _launch(self, bi):
        if bi in self._handles or not self.active:
            return
        self._handles[bi] = self.dist.all_reduce(self.buckets[bi], op=self.dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True)

    def _launch_ready(self):
        """Collectives must be issued in the same order on every rank, whatever the order in which a rank's
        hooks fire (or whether its backward ran at all): buckets leave in ASCENDING index order only -- a
        complete bucket waits for its predecessors (finish() sends the rest, in the same order)."""
        if self._expected is None:
            return
        nxt = len(self._handles)
        # a bucket whose learned set is empty (only never-firing parameters) is complete by definition: it
        # leaves in order like the others (all zeros), so it cannot hold back the overlapped launch of its
        # successors -- the set is agreed across ranks, every rank takes the same decision
        while nxt < len(self.buckets) and self._fired[nxt] >= self._expected[nxt]:
            self._launch(nxt)
            nxt += 1

    def _on_grad(self, p):
        bi, _ = self._slot[p]
        want = self._view(p)
        if p.grad is not None and p.grad.data_ptr() != want.data_ptr():
            # the caller used optimizer.zero_grad(set_to_none=True) (or the parameter was thought
            # unused): autograd then assigned a fresh tensor outside the buckets
            if bi in self._handles:
                self._late.append(p)        # the bucket is already on the wire: reduced on its own in finish()
                return
            want.copy_(p.grad)
            p.grad = want
        elif bi in self._handles:
            raise RuntimeError("GradientSync: a gradient was accumulated into a bucket whose all-reduce had "
                               "already been launched (call sync.zero_grad() before every backward)")
        self._fired[bi].add(p)
        self._launch_ready()

    def finish(self) -> bool:
        """Call after backward() on EVERY rank, also when this rank's loss carried no gradient: launches
        what the hooks could not, waits, turns sums into means.  Returns whether ANY rank had a gradient
        this step -- every rank then clips and steps (or none does), so replicas and schedules stay in
        lock step.  What took part is agreed ACROSS ranks (one small all-reduce of two masks behind the
        buckets): a rank without a local gradient keeps the averaged gradient of its peers, learns the
        same set of participating parameters, and joins the same per-parameter reductions of late
        parameters."""
        for bi in range(len(self.buckets)):           # ascending, like _launch_ready
            self._launch(bi)
        n = len(self.params)
        index = {p: i for i, p in enumerate(self.params)}
        mask = torch.zeros(2 * n, dtype=torch.int32)
        for f in self._fired:
            for p in f:
                mask[index[p]] = 1
        for p in self._late:
            mask[index[p]] = 1
            mask[n + index[p]] = 1
        if self.active and self.world > 1:
            dev = self.buckets[0].device if self.buckets else torch.device('cpu')
            m = mask.to(dev)
            self.dist.all_reduce(m, op=self.dist.ReduceOp.MAX, group=self.group)
            mask = m.cpu()
        fired_any = [self.params[i] for i in range(n) if mask[i]]
        late_any = [self.params[i] for i in range(n) if mask[n + i]]
        any_grad = bool(fired_any)
        for bi, h in self._handles.items():
            h.wait()
        if self._expected is None and any_grad:       # learned from a step in which some rank's backward ran
            self._expected = [set() for _ in self.buckets]
            for p in fired_any:
                self._expected[self._slot[p][0]].add(p)
        local_late = set(self._late)
        for p in late_any:                            # same list, same order, on every rank
            want = self._view(p)
            t = p.grad if p in local_late else torch.zeros_like(want)
            if self.active:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            # ADD to the slot: a rank on which p was on time (its bucket had not left when p fired) carried p's
            # gradient inside the reduced bucket; on the late ranks the slot was zero when the bucket left.  Bucket
            # sum + late sum is therefore the full sum on every rank (copy_ here dropped the on-time ranks' share).
            want.add_(t)
            p.grad = want
            bi, _ = self._slot[p]
            if self._expected is not None:
                self._expected[bi].add(p)
        if self.world > 1:
            for b in self.buckets:
                b.div_(self.world)
        taking_part = set(fired_any)
        for p in self.params:
            want = self._view(p)
            if p in taking_part:
                if p.grad is None or p.grad.data_ptr() != want.data_ptr():
                    p.grad = want                     # no local gradient: the peers' average
            elif p.grad is not None and p.grad.data_ptr() == want.data_ptr():
                p.grad = None
        # ready for the next backward even if the caller resets gradients with optimizer.zero_grad()
        # instead of sync.zero_grad()
        self.n_reduced = len(self._handles) + len(late_any)
        self._fired = [set() for _ in self.buckets]
        self._handles = {}
        self._late = []
        return any_grad

    def close(self):
        for h in self._hooks:
            h.remove()


# --------------------------------------------------------------------------------------------- #
def configure_optimizers(model: torch.nn.Module, cfg):
    """generic_reg_model.py:46-76: AdamW / Adam + StepLR (or the constant StepLR(50, 1.0))."""
    scheduler_type = cfg.get('scheduler', None)
    if scheduler_type is None or scheduler_type in ('none', 'step'):
        base_lr = cfg.base_lr
    else:
        raise NotImplementedError(f"scheduler '{scheduler_type}' (only 'step' / 'none' are shipped)")
    if cfg.optimizer == 'AdamW':
        opt = torch.optim.AdamW(model.parameters(), lr=base_lr, weight_decay=cfg.weight_decay)
    elif cfg.optimizer == 'Adam':
        opt = torch.optim.Adam(model.parameters(), lr=base_lr, weight_decay=cfg.weight_decay)
    else:
        raise NotImplementedError(cfg.optimizer)
    if scheduler_type == 'step':
        sched = torch.optim.lr_scheduler.StepLR(opt, cfg.scheduler_param[0], cfg.scheduler_param[1])
    else:
        sched = torch.optim.lr_scheduler.StepLR(opt, 50, 1.0)
    return opt, sched


def compute_metrics(pred: dict, batch: dict) -> dict:
    """generic_reg_model.py:294-321: per-pair rotation (deg) and translation error of every
    'pose*' entry of pred against batch['pose'], shape (1, B) like the reference."""
    out = {}
    with torch.no_grad():
        for k in [k for k in pred if k.startswith('pose')]:
            err = se3_compare(pred[k][None] if pred[k].dim() == 3 else pred[k], batch['pose'][None, :])
            out[f'rot_err_deg{k[4:]}'] = err['rot_deg']
            out[f'trans_err{k[4:]}'] = err['trans']
    return out


def aggregate_metrics(metrics: List[dict], thresh_rot: float, thresh_trans: float) -> dict:
    """generic_reg_model.py:323-372."""
    if not metrics or not metrics[0]:
        return {}
    cat = {k: torch.cat([m[k] for m in metrics], dim=1) for k in metrics[0]}
    rot_keys = [k for k in cat if k.startswith('rot_err_deg')]
    num_pred = cat[rot_keys[0]].shape[0]
    avg = {}
    for p in range(num_pred):
        suffix = f'{p}' if p < num_pred - 1 else 'final'
        for rk in rot_keys:
            ps = rk[11:]
            tk = 'trans_err' + ps
            avg[f'rot_err_deg{ps}_{suffix}'] = cat[rk][p].mean()
            avg[f'rot_err{ps}_{suffix}_hist'] = cat[rk][p]
            avg[f'{tk}_{suffix}'] = cat[tk][p].mean()
            avg[f'{tk}_{suffix}_hist'] = cat[tk][p]
            ok_r, ok_t = cat[rk][p] < thresh_rot, cat[tk][p] < thresh_trans
            avg[f'reg_success{ps}_{suffix}'] = (ok_r & ok_t).float().mean()
            avg[f'rot_success{ps}_{suffix}'] = ok_r.float().mean()
            avg[f'trans_success{ps}_{suffix}'] = ok_t.float().mean()
    return avg


# --------------------------------------------------------------------------------------------- #
class CheckpointManager:
    """The reference's checkpoint FORMAT and directory protocol (torch_helpers.py:98-242):
    model-<step>.pth = {'state_dict', 'step', 'optimizer', 'scheduler'}; checkpoints.txt lists the
    kept files after a "Best step: N" line; the best-scoring checkpoint is never deleted."""

    def __init__(self, ckpt_dir: str, max_to_keep: int = 6):
        if max_to_keep <= 0:
            raise ValueError('max_to_keep must be at least 1')
        self.dir, self.max_to_keep = ckpt_dir, max_to_keep
        self.kept: List[tuple] = []     # (path, step)
        self.best_score, self.best_step = None, None
        os.makedirs(ckpt_dir, exist_ok=True)
        self._write_index()

    def _path(self, step):
        return os.path.join(self.dir, f'model-{step}.pth')

    def _write_index(self):
        with open(os.path.join(self.dir, 'checkpoints.txt'), 'w') as f:
            f.write(f'Best step: {self.best_step}\n')
            f.write('\n'.join(os.path.basename(p) for p, _ in self.kept))

    def save(self, model, step: int, score: float = 0.0, **extras):
        state = {'state_dict': {k: v for k, v in model.state_dict().items() if not v.is_sparse}, 'step': step}
        for k, v in extras.items():
            state[k] = v.state_dict() if getattr(v, 'state_dict', None) is not None else v
        torch.save(state, self._path(step))
        self.kept.append((self._path(step), step))
        if self.best_score is None or score >= self.best_score:
            old = self.best_step
            self.best_score, self.best_step = score, step
            if old is not None and old not in [s for _, s in self.kept] and os.path.exists(self._path(old)):
                os.remove(self._path(old))
        while len(self.kept) > self.max_to_keep:
            path, s = self.kept.pop(0)
            if s != self.best_step and os.path.exists(path):
                os.remove(path)
        self._write_index()
        return self._path(step)

    @staticmethod
    def load(path: str, model=None, **extras) -> int:
        if os.path.isdir(path):
            with open(os.path.join(path, 'checkpoints.txt')) as f:
                line = f.readline()
            assert line.startswith('Best'), 'checkpoints.txt not in expected format.'
            path = os.path.join(path, f"model-{int(line.split(':')[1])}.pth")
        state = torch.load(path, map_location='cpu' if not torch.cuda.is_available() else None, weights_only=False)
        if 'state_dict' in state and model is not None:
            ret = model.load_state_dict(state['state_dict'], strict=False)
            if ret.unexpected_keys:
                _log.warning('Unexpected keys in checkpoint: %s', ret.unexpected_keys)
            if ret.missing_keys:
                _log.warning('Missing keys in checkpoint: %s', ret.missing_keys)
        for k, obj in extras.items():
            try:
                if k in state and getattr(obj, 'load_state_dict', None) is not None:
                    obj.load_state_dict(state[k])
                else:
                    _log.warning('"%s" ignored from checkpoint loading', k)
            except ValueError as e:      # torch_helpers.py:236-239: log and proceed
                _log.error('Loading %s from checkpoint failed due to error "%s", but ignoring and proceeding...', k, e)
        return state.get('step', 0)


# --------------------------------------------------------------------------------------------- #
class Trainer:
    """fit() / validate() with the reference's step order.  `batches` are dicts as produced by the
    reference's collate_pair (src_xyz / tgt_xyz lists, pose, src_overlap / tgt_overlap) already on
    the device; data loading itself is out of the hot-path scope."""

    def __init__(self, cfg, ckpt_dir: Optional[str] = None, grad_clip: Optional[float] = None, rank: int = 0,
                 world: int = 1, process_group=None, bucket_bytes: int = 16 << 20):
        self.cfg = cfg
        self.grad_clip = cfg.get('grad_clip', 0.0) if grad_clip is None else grad_clip
        self.rank, self.world, self.group = rank, world, process_group
        self.bucket_bytes = bucket_bytes
        self.saver = CheckpointManager(ckpt_dir, max_to_keep=6) if (ckpt_dir and rank == 0) else None
        self.global_step = 0
        self.sync: Optional[GradientSync] = None
        self.optimizer = self.scheduler = None

    def setup(self, model, resume: Optional[str] = None):
        self.optimizer, self.scheduler = configure_optimizers(model, self.cfg)
        if resume is not None:
            self.global_step = CheckpointManager.load(resume, model, optimizer=self.optimizer, scheduler=self.scheduler)
        if self.world > 1:
            self.sync = GradientSync(model.parameters(), self.group, self.bucket_bytes)
        return self

    def train_step(self, model, batch) -> dict:
        """trainer.py:107-146 for one batch."""
        self.global_step += 1
        model.train()
        with torch.enable_grad():
            pred = model(batch)                                  # training_step:
            losses = model.compute_loss(pred, batch)             #   forward + compute_loss
            if self.sync is not None:
                self.sync.zero_grad()
            else:
                self.optimizer.zero_grad()
            has_grad = 'total' in losses and losses['total'].requires_grad
            if has_grad:
                losses['total'].backward()
            if self.sync is not None:
                # explicit RCCL gradient all-reduce (mean); on every rank, always.  With several ranks the
                # step is taken when ANY rank had a gradient (the reference clips and steps unconditionally
                # on every rank, trainer.py:121-127): a rank-local decision would let replicas diverge
                has_grad = self.sync.finish()
            if has_grad:
                if self.grad_clip > 0:
                    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=self.grad_clip)
                self.optimizer.step()
                self.scheduler.step()
        if not bool(torch.isfinite(losses['total'].detach())):
            _log.warning('Total loss is not finite, Ignoring...')   # trainer.py:156-162
        return {k: v.detach() for k, v in losses.items()}

    @torch.no_grad()
    def validate(self, model, batches) -> dict:
        """trainer.py:216-321 / generic_reg_model.py:97-122: losses + metrics over the rank's
        batches, aggregated; a barrier afterwards when running on several ranks."""
        model.eval()
        losses, metrics = [], []
        for batch in batches:
            pred = model(batch)
            losses.append({k: v.detach() for k, v in model.compute_loss(pred, batch).items()})
            metrics.append(compute_metrics(pred, batch))
        out = {'losses': {k: torch.stack([l[k] for l in losses]).mean() for k in losses[0]} if losses else {},
               'metrics': aggregate_metrics(metrics, self.cfg.get('reg_success_thresh_rot', 10),
                                            self.cfg.get('reg_success_thresh_trans', 0.1))}
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier(self.group)                             # trainer.py:304
        return out

    def fit(self, model, train_batches, val_batches=None, epochs: int = 1, validate_every: int = 0,
            on_step: Optional[Callable[[int, dict], None]] = None):
        if self.optimizer is None:
            self.setup(model)
        for _ in range(epochs):
            for batch in train_batches:
                losses = self.train_step(model, batch)
                if on_step is not None:
                    on_step(self.global_step, losses)
                if validate_every > 0 and self.global_step % validate_every == 0 and val_batches is not None:
                    res = self.validate(model, val_batches)
                    score = float(res['metrics'].get('reg_success_final', torch.tensor(0.0)))
                    if self.saver is not None:                   # rank 0 only (trainer.py:67, :313)
                        self.saver.save(model, self.global_step, score=score, optimizer=self.optimizer,
                                        scheduler=self.scheduler)
        return self.global_step
