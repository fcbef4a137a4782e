"""Data-parallel DiT trainer: the MI355X re-expression of the reference's DDP step.

Reference shape of the step: train_offline.py:203-259 (zero_grad -> forward -> loss -> accelerator.backward [DDP
bucketed NCCL all-reduce] -> optimizer.step -> scheduler.step -> barrier) around the stable_audio_tools objective
(training/diffusion.py:365-399).  Here:

  * one process per GPU (torchrun / env RANK, LOCAL_RANK, WORLD_SIZE), torch.distributed backend "nccl" == RCCL
    over xGMI; per-GPU data shards, no data-path collective other than the gradient all-reduce;
  * all trainable parameters live in ONE flat fp32 master buffer (+ a flat bf16 compute mirror the GEMMs read and a
    flat fp32 gradient buffer); module parameters are views, so state_dict()/load_state_dict() keep working;
  * buckets = one per TransformerBlock (reverse layer order) + one for everything else; the wgrad GEMMs write
    straight into the bucket, and the bucket's all-reduce is issued the moment the block's backward has been queued -
    RCCL runs it on its own stream behind an event, overlapping the remaining backward;
  * one fused Adam/AdamW launch over the flat buffers updates master weights, moments and the bf16 mirror and folds
    the 1/world_size gradient averaging in (no separate scale pass, no per-step host sync, no barrier).
"""
import math
import contextlib
import os

import torch
import torch.distributed as dist

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """torch.optim-style front end of kalle_adam_step (maps the reference's optional deepspeed FusedAdam,
    training/utils.py:88-90).  adam_w_mode=False gives torch.optim.Adam's L2 weight decay."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adam_w_mode=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, adam_w_mode=adam_w_mode))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
                st["step"] += 1
                grad = p.grad if p.grad.dtype == torch.float32 else p.grad.float()
                # The kernel writes the weights through the raw pointer, so p._version does not move and the bf16 compute
                # copy that dit_ops.bf16_of caches per parameter version would go stale: a live copy is updated by the same
                # launch (param_bf16), anything else is dropped so the next forward re-casts.
                c = getattr(p, "_kalle_bf16", None)
                live = (c is not None and getattr(p, "_kalle_bf16_pinned", None) is None and c[0] == p._version
                        and c[1].device == p.device and c[1].is_contiguous() and p.is_contiguous())
                ops.adam_step(p.data, grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], c[1] if live else None,
                              lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"],
                              weight_decay=g["weight_decay"], decoupled=g["adam_w_mode"], step=st["step"])
                if c is not None and not live:
                    del p._kalle_bf16
        return loss


def cosine_with_warmup(step, warmup_steps, total_steps):
    """transformers.get_cosine_schedule_with_warmup (train_offline.py:99-104) as a pure function of the step."""
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    prog = (step - warmup_steps) / max(1, total_steps - warmup_steps)   # (past total_steps transformers' lambda rises again: kept)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("KALLE_FORCE_COMM")) and not dist.is_initialized():
        if backend is None:
            # (KALLE_DIST_BACKEND=gloo: rehearse the multi-rank control flow on one GPU, where RCCL refuses two ranks per device)
            backend = os.environ.get("KALLE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class EMASchedule:
    """decay schedule of `ema_pytorch.EMA` as the reference configures it (training/diffusion.py:240-248: beta=0.9999,
    power=3/4, update_every=1, update_after_step=1; third-party, unpinned - restated from the package's published
    algorithm, parity unpinned): until `update_after_step` the average is a copy of the weights, then
    ema += (1 - decay) (w - ema) with decay = clamp(1 - (1 + epoch / inv_gamma)^-power, min_value, beta)."""

    def __init__(self, beta=0.9999, power=2 / 3, update_every=10, update_after_step=100, inv_gamma=1.0, min_value=0.0):
        self.beta, self.power, self.update_every = beta, power, update_every
        self.update_after_step, self.inv_gamma, self.min_value = update_after_step, inv_gamma, min_value
        self.step = 0
        self.initted = False

    def next(self):
        """advance one optimizer step; returns None (skip), "copy", or the decay to use"""
        step = self.step
        self.step += 1
        if step % self.update_every != 0:
            return None
        if step <= self.update_after_step or not self.initted:
            self.initted = step > self.update_after_step or self.initted
            return "copy"
        epoch = max(self.step - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)


class FlatBuckets:
    """Flat fp32 parameter / gradient storage partitioned into all-reduce buckets.  Device-agnostic (the gloo CPU
    tests exercise exactly this class); only the optimizer launch needs the GPU."""

    def __init__(self, named_params, bucket_of, device, with_bf16=True, align=64):
        """named_params: list[(name, Parameter)] (trainable only); bucket_of: name -> bucket key (ordered by first
        appearance)."""
        self.names = [n for n, _ in named_params]
        order = []
        for n, _ in named_params:
            k = bucket_of(n)
            if k not in order:
                order.append(k)
        self.bucket_keys = order
        self.slices = {}          # name -> (start, numel)
        self.bucket_range = {}    # key -> (start, end)
        self.small_range = {}     # key -> (start, end) of the bucket's vectors (norm scales, biases): they lead the bucket, so one
        off = 0                   # clear per bucket serves every sink the kernels ADD into; the matrices behind them are written
        for k in order:           # whole by the grouped weight-gradient launch of the first micro-batch
            start = off
            members = [(n, p) for n, p in named_params if bucket_of(n) == k]
            for want_small in (True, False):
                for n, p in members:
                    # (`_kalle_atomic_grad`: a tensor of >= 2 dims whose gradient kernel ADDS into the sink - the depthwise taps of
                    # a ConformerModule - travels with the vectors, i.e. in the range that is cleared every step)
                    if (p.dim() < 2 or getattr(p, "_kalle_atomic_grad", False)) != want_small:
                        continue
                    self.slices[n] = (off, p.numel())
                    off += (p.numel() + align - 1) // align * align
                if want_small:
                    self.small_range[k] = (start, off)
            self.bucket_range[k] = (start, off)
        self.total = off
        self.param = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.param_bf16 = torch.zeros(off, device=device, dtype=torch.bfloat16) if with_bf16 else None
        with torch.no_grad():
            for n, p in named_params:
                s, ne = self.slices[n]
                self.param[s:s + ne].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                p.data = self.param[s:s + ne].view(p.shape)
                p.grad = self.grad[s:s + ne].view(p.shape)
                if with_bf16:
                    self.param_bf16[s:s + ne].copy_(self.param[s:s + ne])
                    p._kalle_bf16_pinned = self.param_bf16[s:s + ne].view(p.shape)
        self.params = dict(named_params)

    def grad_view(self, name):
        s, ne = self.slices[name]
        return self.grad[s:s + ne].view(self.params[name].shape)

    def bucket_grad(self, key):
        a, b = self.bucket_range[key]
        return self.grad[a:b]


class DataParallelTrainer:
    """Owns flat buffers, bucketed gradient all-reduce and the fused optimizer for a DiT or the Llasa task model (any
    module tree whose repeated layers are this package's TransformerBlock / LlamaDecoderLayer: one bucket each)."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, optimizer="Adam",
                 grad_accum_steps=1, lr_schedule=None, process_group=None, comm_dtype=torch.float32):
        from .stable_audio_tools.models.transformer import TransformerBlock
        self.model = model
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.pg = process_group
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.decoupled = optimizer.lower() in ("adamw", "fusedadam")
        self.grad_accum_steps = max(1, grad_accum_steps)
        self.lr_schedule = lr_schedule
        self.comm_dtype = comm_dtype
        self.step_count = 0
        self.micro = 0
        self.last_lr = lr
        self.comm_timing = None            # set to a list to collect (start, end, buckets) events of the all-reduce waits
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        device = named[0][1].device
        # bucket key: the owning TransformerBlock's module path, or "_rest"
        self.blocks = [(n, m) for n, m in model.named_modules()
                       if isinstance(m, TransformerBlock) or getattr(m, "_kalle_bucket_unit", False)]
        prefixes = [n + "." for n, _ in self.blocks]

        def bucket_of(name):
            for pre in prefixes:
                if name.startswith(pre):
                    return pre
            # a pretransform trained with enable_grad (models/factory.py:77-80): its gradients travel in a bucket of their own
            return "_vae" if name.startswith("pretransform.") or ".pretransform." in name else "_rest"

        # identical initial weights on every rank (rank 0 broadcasts), as DDP does at wrap time
        if self.world > 1:
            for _, p in named:
                dist.broadcast(p.data, src=0, group=self.pg)
        self.flat = FlatBuckets(named, bucket_of, device)
        self.exp_avg = torch.zeros_like(self.flat.param)
        self.exp_avg_sq = torch.zeros_like(self.flat.param)
        self._pending = []
        # Sharded optimizer (KALLE_SHARD_OPTIMIZER=1, off by default; world a power of two <= 64, fp32 buckets): a block bucket's
        # all-reduce is issued as its two halves with the optimizer in the middle - reduce-scatter of the matrix gradients, the fused Adam
        # on this rank's 1 / world of them, all-gather of the updated fp32 weights (the same bytes on the links as the all-reduce; the
        # Adam pass shrinks from 31.5 GB to 31.5 / world GB per rank and step, the bf16 mirror is re-cast from the gathered weights).
        # The vectors at the head of a bucket (norm scales, biases) stay all-reduced and are updated on every rank; so do the buckets
        # outside the blocks.  Moments exist on their owner only: state_dict() gathers them (a collective - call it on every rank).
        self.shard_opt = (os.environ.get("KALLE_SHARD_OPTIMIZER", "0") == "1" and comm_dtype == torch.float32
                          and self.world & (self.world - 1) == 0 and self.world <= 64)
        self._shard_bufs = {}
        # Optimizer overlapped with the backward pass: a block's bucket is final as soon as its backward kernels are queued (and
        # its all-reduce has landed), so its slice of the fused Adam runs right then on a side stream - an HBM-bound pass under
        # the MFMA-bound GEMMs of the blocks still to come - instead of one 5.4 ms pass over all 1.05 B parameters at the end.
        # (KALLE_OVERLAP_ADAM=0: the single pass at the end of the step.)
        self.overlap_adam = device.type == "cuda" and os.environ.get("KALLE_OVERLAP_ADAM", "1") != "0"
        side = self.overlap_adam or (device.type == "cuda" and self.shard_opt)
        self._opt_stream = torch.cuda.Stream(device=device) if side else None
        self._comm_stream = torch.cuda.Stream(device=device) if side else None
        self._comm_events = 0
        self._shard_ev = None
        self._opt_done = set()            # bucket keys whose Adam slice of the current optimizer step has been queued
        self._opt_lr = None               # learning rate of the optimizer step in progress (set when its backward starts)
        self._overlap_now = False         # decided per optimizer step: only where the backward GEMMs are long enough to hide it
        self.overlap_min_rows = int(os.environ.get("KALLE_OVERLAP_ADAM_MIN_ROWS", "4096"))
        for n, blk in self.blocks:
            pre = n + "."
            blk._kalle_grad_sinks = {k[len(pre):]: self.flat.grad_view(k) for k in self.flat.names if k.startswith(pre)}
            blk._kalle_grad_accumulate = False
            blk._kalle_bucket_key = pre
            blk._kalle_on_backward_done = self._on_block_done
        # consecutive blocks of one transformer know each other: the LayerNorm backward that produces a block's output gradient
        # also adds its column sums into that block's FF-out bias sink (functional.TransformerBlockFn.backward)
        for (na, a), (nb, b_) in zip(self.blocks, self.blocks[1:]):
            if (isinstance(a, TransformerBlock) and isinstance(b_, TransformerBlock) and na.rsplit(".", 1)[0] == nb.rsplit(".", 1)[0]
                    and getattr(b_, "layer_ix", -1) == getattr(a, "layer_ix", -9) + 1):
                object.__setattr__(b_, "_kalle_prev_block", a)      # (not a child module: plain attribute)
        # large tables whose backward scatter-adds straight into the flat gradient (token embeddings)
        for n, p in named:
            if getattr(p, "_kalle_wants_sink", False) and bucket_of(n) == "_rest":
                p._kalle_grad_sink = self.flat.grad_view(n)

    # -- gradient communication ---------------------------------------------------------------------------
    def _comm_active(self):
        return self.world > 1 or (dist.is_initialized() and bool(os.environ.get("KALLE_FORCE_COMM")))

    def _allreduce(self, buf, defer=True):
        """starts the all-reduce of one bucket; returns (work handle | None, low-precision staging tensor | None).  defer: the
        wait (and the copy back from a low-precision bucket) is left to _finish_comm at the end of the backward pass;
        otherwise the caller waits where it needs the result"""
        if not self._comm_active():
            return None, None
        if self.comm_dtype == torch.float32 or not buf.is_cuda:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            if defer:
                self._pending.append((work, None, None))
            return work, None
        low = ops.cast(buf, self.comm_dtype)
        work = dist.all_reduce(low, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        if defer:
            self._pending.append((work, low, buf))
        return work, low

    def _queue_bucket(self, keys, ev):
        """Overlapped path, three streams.  COMM stream: waits for the compute stream's event (the bucket's gradient kernels),
        casts if the buckets travel in low precision and issues the all-reduce - RCCL's own stream orders itself behind the
        stream the collective is issued from, so bucket k+1's all-reduce depends on the backward pass only, never on the
        optimizer.  OPTIMIZER stream: waits for the all-reduce's completion (work.wait() blocks that stream, not the host),
        copies a low-precision result back into the fp32 bucket and runs the bucket's slice of the fused Adam - under the
        backward kernels of the blocks still to come and under the next buckets' all-reduces."""
        works = []
        if self._comm_active():
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                for key in keys:
                    buf = self.flat.bucket_grad(key)
                    work, low = self._allreduce(buf, defer=False)
                    works.append((work, low, buf))
        with torch.cuda.stream(self._opt_stream):
            self._opt_stream.wait_event(ev)
            for work, low, buf in works:
                work.wait()
                if low is not None:
                    low.record_stream(self._opt_stream)          # (allocated on the comm stream, read here)
                    ops.copy_rows(low, buf, 1, 1, buf.numel(), 0, buf.numel(), 0, buf.numel())
                self._comm_events += 1
            for key in keys:
                self._adam_bucket(key)

    # -- sharded optimizer -------------------------------------------------------------------------------------
    def _shard_range(self, key):
        """(vector start, matrix start, bucket end, elements per rank, this rank's first element) of a block bucket"""
        s0, s1 = self.flat.small_range[key]
        b1 = self.flat.bucket_range[key][1]
        c = (b1 - s1) // self.world              # (parameters are laid out on 64-element boundaries: any power of two <= 64 divides)
        return s0, s1, b1, c, s1 + self.rank * c

    def _sharded_active(self):
        return self.shard_opt and self._comm_active()

    def _sharded_bucket(self, key, ev):
        """reduce-scatter -> Adam on the own chunk (and on the all-reduced vectors) -> all-gather of the fp32 weights -> bf16 mirror.
        On the GPU: collectives on the comm stream, Adam on the optimizer stream, both behind the backward pass's event; on the CPU
        (gloo tests) the same calls in order."""
        f = self.flat
        s0, s1, b1, c, o0 = self._shard_range(key)
        cuda = f.grad.is_cuda
        comm = torch.cuda.stream(self._comm_stream) if cuda else contextlib.nullcontext()
        opt = torch.cuda.stream(self._opt_stream) if cuda else contextlib.nullcontext()
        shard = self._shard_bufs.get(c)
        if shard is None:
            shard = self._shard_bufs[c] = torch.empty(c, device=f.grad.device, dtype=torch.float32)
        with comm:
            if cuda:
                self._comm_stream.wait_event(ev)
                if self._shard_ev is not None:
                    self._comm_stream.wait_event(self._shard_ev)      # the previous bucket's Adam has read the shared chunk buffer
            w_vec = dist.all_reduce(f.grad[s0:s1], op=dist.ReduceOp.SUM, group=self.pg, async_op=True) if s1 > s0 else None
            w_rs = dist.reduce_scatter_tensor(shard, f.grad[s1:b1], op=dist.ReduceOp.SUM, group=self.pg, async_op=True) if c else None
        with opt:
            if cuda:
                self._opt_stream.wait_event(ev)
            for w in (w_vec, w_rs):
                if w is not None:
                    w.wait()
            self._adam_range(s0, s1)
            self._adam_range(o0, o0 + c, grad=shard)
            ev2 = None
            if cuda:
                ev2 = torch.cuda.Event()
                ev2.record(self._opt_stream)
                self._shard_ev = ev2
        with comm:
            if cuda:
                self._comm_stream.wait_event(ev2)
            if c:
                own = f.param[o0:o0 + c] if cuda else f.param[o0:o0 + c].clone()    # (RCCL gathers in place; gloo gets a copy)
                dist.all_gather_into_tensor(f.param[s1:b1], own, group=self.pg, async_op=True).wait()
                if f.param_bf16 is not None:
                    if cuda:
                        ops.copy_rows(f.param[s1:b1], f.param_bf16[s1:b1], 1, 1, b1 - s1, 0, b1 - s1, 0, b1 - s1)
                    else:
                        f.param_bf16[s1:b1].copy_(f.param[s1:b1])
        self._comm_events += 1
        self._opt_done.add(key)

    def gather_moments(self):
        """sharded optimizer: every rank receives the Adam moments of the chunks it does not own (a collective; state_dict() calls it)"""
        if not self._sharded_active():
            return
        for _, blk in self.blocks:
            s0, s1, b1, c, o0 = self._shard_range(blk._kalle_bucket_key)
            if c:
                for buf in (self.exp_avg, self.exp_avg_sq):
                    dist.all_gather_into_tensor(buf[s1:b1], buf[o0:o0 + c].clone(), group=self.pg)

    def _on_block_done(self, blk):
        """called (from autograd's backward) right after a block's backward kernels were queued"""
        if not self._boundary():
            return
        key = blk._kalle_bucket_key
        if self._sharded_active():
            ev = None
            if self.flat.grad.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            self._sharded_bucket(key, ev)
            return
        if not self._overlap_now:
            self._allreduce(self.flat.bucket_grad(key))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._queue_bucket([key], ev)

    def _begin_optimizer_step(self):
        """fixes the step number and learning rate of the optimizer step whose last micro-batch is about to run backward"""
        k = self.step_count + 1
        # lr_schedule(s): multiplier for the optimizer step taken after `s` completed ones - torch LambdaLR's convention,
        # so the k-th step (k = 1, 2, ...) uses lr_lambda(k - 1) exactly as optimizer.step(); scheduler.step() does
        # (train_offline.py:247-248)
        self._opt_lr = self.lr * (self.lr_schedule(k - 1) if self.lr_schedule else 1.0)
        self._opt_done = set()
        self._comm_events = 0
        self._shard_ev = None

    def _adam_bucket(self, key):
        a, b = self.flat.bucket_range[key]
        self._adam_range(a, b)
        self._opt_done.add(key)

    def _adam_range(self, a, b, grad=None):
        if b <= a:
            return
        f = self.flat
        ops.adam_step(f.param[a:b], f.grad[a:b] if grad is None else grad, self.exp_avg[a:b], self.exp_avg_sq[a:b],
                      f.param_bf16[a:b] if f.param_bf16 is not None else None, lr=self._opt_lr, beta1=self.betas[0],
                      beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay, decoupled=self.decoupled,
                      step=self.step_count + 1, grad_scale=1.0 / (self.world * self.grad_accum_steps))

    def _boundary(self):
        return (self.micro + 1) % self.grad_accum_steps == 0

    def _finish_comm(self):
        if self._overlap_now and self._boundary():
            # what the bucket hooks did not cover (embedders, in / out projections, a trained VAE) goes the same way; the compute
            # stream then waits for the optimizer stream ONCE - the time it sits there is what neither the all-reduces nor the
            # optimizer slices managed to hide behind the backward pass
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            self._queue_bucket([k for k in self.flat.bucket_keys if k not in self._opt_done], ev)
            timed = self.comm_timing is not None
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            cur.wait_stream(self._opt_stream)
            if self._sharded_active():
                cur.wait_stream(self._comm_stream)          # the all-gathers of the updated weights and their bf16 re-cast
            if timed:
                e1.record()
                self.comm_timing.append((e0, e1, self._comm_events))
            return
        if self._boundary():
            for key in ("_rest", "_vae"):
                if key in self.flat.bucket_range:
                    self._allreduce(self.flat.bucket_grad(key))
            if self._sharded_active() and self.flat.grad.is_cuda and self._opt_stream is not None:
                # (small micro-batches keep the single optimizer pass for what is left; the sharded buckets ran on the side streams)
                cur = torch.cuda.current_stream()
                cur.wait_stream(self._opt_stream)
                cur.wait_stream(self._comm_stream)
        timed = self.comm_timing is not None and self._pending and self.flat.grad.is_cuda
        if timed:
            # exposed (not overlapped) all-reduce time = how long the compute stream sits in the waits below: nothing is
            # launched on it between the two events
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, low, dst in self._pending:
            work.wait()
            if low is not None:
                ops.copy_rows(low, dst, 1, 1, dst.numel(), 0, dst.numel(), 0, dst.numel())
        if timed:
            e1.record()
            self.comm_timing.append((e0, e1, len(self._pending)))
        self._pending = []

    def comm_summary(self):
        """after a device sync: {"ranks", "backend", "buckets_per_step", "exposed_ms_per_step"} from the events collected while
        `comm_timing` was a list (bench.py switches it on for the timed steps)"""
        ev = self.comm_timing or []
        active = dist.is_initialized() and (self.world > 1 or bool(os.environ.get("KALLE_FORCE_COMM")))
        out = {"ranks": self.world, "backend": dist.get_backend(self.pg) if dist.is_initialized() else None,
               "allreduce_active": bool(active), "sharded_optimizer": bool(active and self.shard_opt),
               "comm_dtype": str(self.comm_dtype).replace("torch.", ""),
               "buckets_per_step": ev[0][2] if ev else 0,
               "exposed_ms_per_step": (sum(a.elapsed_time(b) for a, b, _ in ev) / len(ev)) if ev else 0.0}
        return out

    # -- one micro-batch ------------------------------------------------------------------------------------
    def backward(self, loss):
        first = self.micro % self.grad_accum_steps == 0
        # ONE clear of the flat gradient per optimizer step; every gradient kernel then accumulates into its sink.  (Letting
        # the first micro-batch overwrite instead costs a separate small clear in front of each split-K weight gradient and
        # column sum: ~300 launches and twice the time of this single pass.)
        # That only pays when the weight gradients are split-K GEMMs (K = tokens of the micro-batch >= ~8 k): a weight gradient
        # that writes its output whole would have to read the cleared buffer back instead (B = 16: -5 %), so small
        # micro-batches keep the overwrite-on-first-micro-batch rule.
        rows = max((getattr(blk, "_kalle_last_rows", 0) for _, blk in self.blocks), default=0)
        from . import dit_ops
        # With the grouped weight-gradient launch (every block's matrices in one kernel) the first micro-batch WRITES the matrix
        # gradients - no clear of the 4.2 GB they occupy and no read-add-store - and only the vectors at the head of every bucket
        # (norm scales, biases: sinks the kernels add into atomically) are cleared, one small memset per bucket.
        grouped = dit_ops.GROUP_WGRAD and rows > 0 and rows % 8 == 0
        clear_once = rows >= 8192 and not grouped
        for _, blk in self.blocks:
            blk._kalle_grad_accumulate = True if (clear_once or grouped) else not first
            blk._kalle_wgrad_overwrite = grouped and first
        if first:
            if clear_once:
                self.flat.grad.zero_()
            else:
                if grouped:
                    for _, blk in self.blocks:
                        a, b = self.flat.small_range[blk._kalle_bucket_key]
                        if b > a:
                            self.flat.grad[a:b].zero_()
                for key in ("_rest", "_vae"):
                    if key in self.flat.bucket_range:
                        self.flat.bucket_grad(key).zero_()     # autograd accumulates (+=) into these views
        if self._boundary():
            self._begin_optimizer_step()
            # at small batches (B = 16: 2016 rows) the backward GEMMs are short, latency-bound launches that the optimizer's
            # 31.5 GB of traffic slows down by as much as it hides (33.6 -> 33.8 ms); from B = 64 on it pays (73.3 -> 72.6 ms)
            self._overlap_now = self.overlap_adam and (rows >= self.overlap_min_rows or not self.blocks)
        loss.backward()
        self._finish_comm()
        if self._boundary():
            self.optimizer_step()
        self.micro += 1

    # -- exponential moving average of the weights (reference: ema_pytorch.EMA, training/diffusion.py:240-248, 440-441) --
    def enable_ema(self, **kw):
        self.ema_schedule = EMASchedule(**kw)
        self.ema = self.flat.param.clone()
        return self

    def _ema_update(self):
        d = self.ema_schedule.next()
        if d is None:
            return
        if d == "copy":
            self.ema.copy_(self.flat.param)
        else:
            ops.axpby(self.ema, self.flat.param, d, 1.0 - d, out=self.ema)   # one fused pass over the flat buffers

    def ema_state_dict(self):
        """the averaged weights under the model's parameter names (views into the flat EMA buffer)"""
        return {n: self.ema[s:s + ne].view(self.flat.params[n].shape) for n, (s, ne) in self.flat.slices.items()}

    def optimizer_step(self):
        if self._opt_lr is None:                           # called directly (not through backward())
            self._begin_optimizer_step()
        if not (self._overlap_now and len(self._opt_done) == len(self.flat.bucket_keys)):
            # one fused pass over whatever has not been updated yet (everything, without the overlapped slices)
            if self._opt_done:
                for key in self.flat.bucket_keys:
                    if key not in self._opt_done:
                        self._adam_bucket(key)
            else:
                self._adam_range(0, self.flat.total)
        self.step_count += 1
        self.last_lr = self._opt_lr                        # what this optimizer step actually used (for logging)
        self._opt_lr = None
        self._opt_done = set()
        if getattr(self, "ema", None) is not None:
            self._ema_update()

    def train_step(self, diffusion, latents, t, noise, cond, objective="v", padding_mask=None):
        """fwd + bwd + all-reduce + optimizer for one micro-batch. Returns the (device) loss tensor, no host sync."""
        from .stable_audio_tools.training.diffusion import diffusion_train_step
        loss, _ = diffusion_train_step(diffusion, latents, t, noise, cond, objective=objective,
                                       padding_mask=padding_mask)
        self.backward(loss)
        return loss.detach()

    # -- checkpointing (reference: weights only, train_offline.py:261-263; here the optimizer, schedule position, gradient-
    #    accumulation phase and the EMA too, so that a resumed run continues bit for bit) -------------------------------------
    def state_dict(self):
        self.gather_moments()
        sd = {"model": self.model.state_dict(), "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
              "step": self.step_count, "micro": self.micro,
              # the moments are raw flat tensors: their meaning is the flat layout (name -> (offset, numel)), which travels with
              # them and is checked on load (a checkpoint written under another layout has the same total size)
              "layout": {n: list(v) for n, v in self.flat.slices.items()}}
        if self.micro % self.grad_accum_steps != 0:
            sd["grad"] = self.flat.grad        # a half-finished accumulation window: the micro-batches already summed
        if getattr(self, "ema", None) is not None:
            sc = self.ema_schedule
            sd["ema"] = self.ema
            sd["ema_schedule"] = {"step": sc.step, "initted": sc.initted, "beta": sc.beta, "power": sc.power,
                                  "update_every": sc.update_every, "update_after_step": sc.update_after_step,
                                  "inv_gamma": sc.inv_gamma, "min_value": sc.min_value}
        return sd

    def load_state_dict(self, sd):
        layout = sd.get("layout")
        if layout is not None:
            mine = {n: list(v) for n, v in self.flat.slices.items()}
            if layout != mine:
                diff = [n for n in mine if layout.get(n) != mine[n]][:4]
                raise ValueError(f"checkpoint was written under another flat parameter layout (first differences: {diff}); "
                                 "the Adam moments cannot be mapped")
        elif sd["exp_avg"].numel() != self.exp_avg.numel():
            raise ValueError("checkpoint moments do not match this trainer's flat buffers")
        self.model.load_state_dict(sd["model"], strict=False)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = sd["step"]
        self.micro = sd.get("micro", 0)
        if self.micro % self.grad_accum_steps != 0:
            if "grad" in sd:
                self.flat.grad.copy_(sd["grad"])
            else:       # (older checkpoints: the partial sums are gone - restart the window rather than scale an empty buffer)
                self.micro -= self.micro % self.grad_accum_steps
        if "ema" in sd:
            es = dict(sd["ema_schedule"])
            step, initted = es.pop("step"), es.pop("initted")
            self.enable_ema(**es)                      # the decay schedule travels with the checkpoint
            self.ema.copy_(sd["ema"])
            self.ema_schedule.step, self.ema_schedule.initted = step, initted
        self.resync_bf16()

    def resync_bf16(self):
        """after loading weights into the fp32 views: refresh the bf16 compute mirror"""
        self.flat.param_bf16.copy_(self.flat.param)
