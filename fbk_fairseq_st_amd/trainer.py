"""Training step driver, after fairseq/trainer.py:334-495.

Order of one update, as in the reference: reseed (seed + num_updates) -> zero grads -> for each
micro-batch: move to the GPU, task.train_step (forward, loss, backward) -> sum logging outputs and
sample sizes over ranks -> gradients x 1/sum(sample_size) -> global-norm clip -> Adam -> LR schedule.
What differs is where the work runs: the gradient all-reduce is launched bucket by bucket from inside
backward (distributed.BucketedGradReducer), scaling + clipping are folded into the norm / Adam kernels,
and nothing but the CTC-compression lengths synchronises the host with the GPU before the final stats.
"""
import os
import time

import torch

from . import distributed as D
from .optim import ArenaAdam, InverseSquareRootSchedule


def move_to_device(sample, device):
    if torch.is_tensor(sample):
        return sample.to(device, non_blocking=True)
    if isinstance(sample, dict):
        return {k: move_to_device(v, device) for k, v in sample.items()}
    if isinstance(sample, (list, tuple)):
        return type(sample)(move_to_device(v, device) for v in sample)
    return sample


class Trainer:
    def __init__(self, args, task, model, criterion, device=None, compute_dtype=None):
        self.args, self.task, self.model, self.criterion = args, task, model, criterion
        self.device = torch.device(device if device is not None else "cuda:%d" % getattr(args, "device_id", 0))
        if self.device.type == "cuda":
            # one process drives one GPU: the kernels are launched on the CURRENT device's stream (lib.stream caches it), so the
            # trainer's device must be the current one before anything is allocated or launched (fairseq_cli/train.py:36-37)
            torch.cuda.set_device(self.device)
        if compute_dtype is None:
            bf16 = getattr(args, "bf16", False) or getattr(args, "compute_dtype", None) == "bf16"
            compute_dtype = torch.bfloat16 if bf16 else torch.float32
        extra = criterion.arena_params() if hasattr(criterion, "arena_params") else None
        self.arena = model.materialize(self.device, compute_dtype, extra=extra)
        lr = args.lr[0] if isinstance(args.lr, (list, tuple)) else args.lr
        self.optimizer = self.build_optimizer(lr)
        self.lr_scheduler = InverseSquareRootSchedule(self.optimizer, lr, getattr(args, "warmup_updates", 4000),
                                                      getattr(args, "warmup_init_lr", -1))
        self.num_updates = 0
        self.world = D.get_world_size()
        cap = getattr(args, "bucket_cap_bytes", None)                     # tests: buckets of a few KB
        self.reducer = D.BucketedGradReducer(self.arena.grad, int(cap) if cap else getattr(args, "bucket_cap_mb", 64) << 20)
        model.engine.on_grads_ready = self._grads_ready
        if self.world > 1 and getattr(model.engine, "defer_wgrad", False):
            model.engine.wgrad_flush_layers = max(1, model.hp.enc_layers // 2)     # see S2TEngine.wgrad_flush_layers
        # --reserve-cus N (default 0): the persistent one-workgroup-per-CU kernels
        # (gemm256, wgrad_group: 128 KiB of LDS each) leave N CUs to RCCL's kernels, which run beside backward while a bucket
        # travels -- a workgroup that finds its CU taken would otherwise wait for a whole round of the others.  Measured on one GPU
        # against a stand-in for the collective: bench.py data_parallel.dry_run.comm_standin.
        self.reserve_cus = int(getattr(args, "reserve_cus", 0) or 0)
        if self.device.type == "cuda":
            from . import kernels as K
            K.set_option("reserve_cus", self.reserve_cus)
        self._ranges = {}
        self.dp_dry_run = False         # world 1 only: run the reducer's bookkeeping as a data-parallel rank would (set_dp_dry_run)
        self._sync_grads = True         # False while a non-final micro-batch accumulates locally (the reference's no_sync)
        self._dummy_batch = None
        self.last_stats = {}
        self._optim_history = []

    def build_optimizer(self, lr):
        args = self.args
        betas = eval(args.adam_betas) if isinstance(getattr(args, "adam_betas", None), str) else getattr(args, "adam_betas", (0.9, 0.999))
        return ArenaAdam(self.arena, lr=lr, betas=betas, eps=getattr(args, "adam_eps", 1e-8),
                         weight_decay=getattr(args, "weight_decay", 0.0))

    # ---- overlap of the gradient all-reduce with backward
    def _grads_ready(self, prefix):
        """Called by the engine as backward finishes a parameter group (last layer first): hand the finished,
        contiguous tail of the flat gradient buffer to the reducer, which launches RCCL on it asynchronously."""
        if (self.world <= 1 and not self.dp_dry_run) or not self._sync_grads:
            return
        r = self._ranges.get(prefix)
        if r is None:
            names = [n for n in self.arena.slices if n.startswith(prefix)]
            r = self.arena.slice_of(names) if names else (0, 0)
            self._ranges[prefix] = r
        if r[1] > r[0]:
            self.reducer.notify(*r)

    def set_dp_dry_run(self, on):
        """One process, no collective: make the update take the DATA-PARALLEL schedule (the mid-backward weight-gradient flush of a
        world > 1 trainer) and let the engine's readiness reports reach the reducer, whose plan / early_elems / launched list then say
        what a rank of an N-GPU job would hand to RCCL from inside backward and what is left after it.  Bookkeeping only: with one
        rank BucketedGradReducer issues nothing.  bench.py reports it at N = 1 (`data_parallel.dry_run`)."""
        assert self.world == 1
        self.dp_dry_run = bool(on)
        eng = self.model.engine
        if getattr(eng, "defer_wgrad", False):
            eng.wgrad_flush_layers = max(1, self.model.hp.enc_layers // 2) if on else None
        self._ranges = {}

    def _prepare_sample(self, sample):
        """trainer.py:631-653 + host-side statistics taken BEFORE the copy (no device round trip later)."""
        if sample is None or len(sample) == 0:
            return None
        if sample.get("_prepared", False):
            return sample
        s = dict(sample)
        if "transcript_target" in s and torch.is_tensor(s["transcript_target"]) and not s["transcript_target"].is_cuda:
            s["transcript_target_host"] = s["transcript_target"]
            s["transcript_target_lengths_host"] = s["transcript_target_lengths"]
        ni = s["net_input"]
        if torch.is_tensor(ni["src_lengths"]) and not ni["src_lengths"].is_cuda:
            s.setdefault("nframes", int(ni["src_lengths"].sum()))
            lens_host = ni["src_lengths"]
        else:
            lens_host = None
        host_keys = {k: s[k] for k in ("transcript_target_host", "transcript_target_lengths_host") if k in s}
        s = move_to_device({k: v for k, v in s.items() if k not in host_keys}, self.device)
        s.update(host_keys)
        if lens_host is not None:
            s["net_input"]["src_lengths"] = lens_host          # the encoder takes host lengths (no sync)
        s["_prepared"] = True
        return s

    def prepare(self, sample):
        """Stage a batch in HBM ahead of time (what a prefetching data loader does)."""
        return self._prepare_sample(sample)

    def train_step(self, samples):
        """One optimizer update over a list of micro-batches (`--update-freq`).  Returns the reduced stats.

        Data-parallel semantics of the reference (fairseq/trainer.py:334-430):
          * gradients of micro-batches 0..n-2 accumulate locally; only the LAST micro-batch's backward hands finished slices to
            the reducer (maybe_no_sync, trainer.py:359-375; legacy_distributed_data_parallel.py:78-83,138), so every element is
            all-reduced exactly once, after all local contributions have landed;
          * an empty micro-batch (a short shard padded by the iterator) runs the cached dummy batch with ignore_grad, so that
            every rank executes the same backward and launches the same collectives (trainer.py:337-357,412-418); as in the
            reference, a rank whose LAST micro-batch was a dummy reports sample_size 0 and no logging outputs."""
        if self._dummy_batch is None:
            self._dummy_batch = next((s for s in samples if s is not None and len(s) > 0), None)
        self._set_seed()
        self.model.train(); self.criterion.train()
        self.optimizer.zero_grad()
        self.reducer.reset()
        reset = getattr(self.model.engine, "reset_wgrad", None)
        if reset is not None:
            reset()                          # nothing queued by an update that died half-way may reach this one
        logs, sample_size, is_dummy = [], 0, False
        for i, sample in enumerate(samples):
            s = self._prepare_sample(sample)
            is_dummy = s is None
            if is_dummy:
                if self._dummy_batch is None:
                    raise RuntimeError("an empty batch arrived before any real one: there is no dummy batch to keep the ranks aligned")
                s = self._dummy_batch = self._prepare_sample(self._dummy_batch)
            self._sync_grads = i == len(samples) - 1
            try:
                loss, ss, log = self.task.train_step(s, self.model, self.criterion, self.optimizer, self.num_updates,
                                                     ignore_grad=is_dummy)
            finally:
                self._sync_grads = True
            logs.append(log)
            sample_size += ss
            self._log_keys = sorted(log)
        if is_dummy:                                                                # trainer.py:412-418
            sample_size, logs = 0, []
        flush = getattr(self.model.engine, "flush_wgrad", None)
        if flush is not None:
            flush()                                                                 # no queued weight gradient may outlive the backward passes
        self.reducer.finish()                                                       # R2
        if self.world > 1 and self.device.type == "cuda" and hasattr(self.optimizer, "set_device_divisor"):
            # R3 without the host: the summed sample size stays on the device (a fill kernel, one 8-byte all-reduce) and the clip
            # kernel divides by it -- the reference reads it back before scaling (trainer.py:416-430), which on this path would drain
            # the GPU between backward and Adam on every update.  The host value is fetched with the logging statistics (reduce_stats).
            total_ss = torch.full((1,), float(sample_size), dtype=torch.float64, device=self.device)
            D.all_reduce_tensor(total_ss)
            self.optimizer.set_device_divisor(total_ss)
        else:
            stats = {"sample_size": float(sample_size)}
            if self.world > 1:
                stats = D.all_reduce_stats(stats, self.device)                      # R3 (must land before scaling)
            total_ss = max(stats["sample_size"], 1.0)
            self.optimizer.multiply_grads(1.0 / total_ss)                           # trainer.py:426-430
        gnorm = self.optimizer.clip_grad_norm(getattr(self.args, "clip_norm", 25.0))
        self.optimizer.step()
        self.num_updates += 1
        self.lr_scheduler.step_update(self.num_updates)
        self._pending = (logs, gnorm, total_ss)
        return self._pending

    def _set_seed(self):
        """fairseq/trainer.py:655-661: every update starts from seed + num_updates, on every rank -- torch's generators (the LayerDrop
        keep / drop draws of conv_transformer.py:238-243 come from the global CPU generator: ranks that consumed it differently before
        this update, e.g. validation shards of unequal size, must still drop the SAME layers, or one rank's Adam steps a layer that
        another skips) and the dropout streams of the kernels."""
        seed = getattr(self.args, "seed", 1) + self.num_updates
        torch.manual_seed(seed)
        if self.device.type == "cuda":
            torch.cuda.manual_seed(seed)
        self.model.set_seed(seed)

    # ---- checkpoints (fairseq/trainer.py:173-266, fairseq/checkpoint_utils.py:245-285): the reference's file layout -- "args",
    # "model" (reference parameter names, f32 masters), "criterion", "optimizer_history", "extra_state", "last_optimizer_state".
    # The optimizer state is written in the reference's layout too (a torch.optim state dict keyed by the position of each parameter
    # in the reference's own parameter order, per-parameter Adam moments split out of the fused arena tensors; optimizer_name
    # "FairseqAdam"), so a checkpoint of the reference resumes here WITH its moments and one of ours resumes there.  Files of
    # earlier builds (optimizer_name "ArenaAdam": two flat arena-ordered vectors) still load.
    def optimizer_parameter_names(self):
        """the reference's optimizer parameter order for this model + criterion (trainer.py:140-146: requires_grad only)"""
        from .conv_transformer import reference_slot
        frozen = set(getattr(self.model, "_frozen_names", ()) or ())
        names = [n for n in self.model.reference_parameter_names() if reference_slot(n)[0] not in frozen]
        return names + [n for n, p in self.criterion.named_parameters() if not getattr(p, "_s2t_anchor", False)]

    def save_checkpoint(self, filename, extra_state=None):
        if D.get_rank() != 0:                                       # only the data-parallel master writes (trainer.py:175)
            return
        state = {"args": self.args, "model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                 "optimizer_history": self._optim_history + [{"criterion_name": self.criterion.__class__.__name__,
                                                              "optimizer_name": "FairseqAdam",
                                                              "lr_scheduler_state": self.lr_scheduler.state_dict(),
                                                              "num_updates": self.num_updates}],
                 "extra_state": dict(extra_state or {})}
        if any(True for _ in self.criterion.parameters()):
            state["criterion"] = {k: v.detach().cpu() for k, v in self.criterion.state_dict().items()}
        if not getattr(self.args, "no_save_optimizer_state", False):
            state["last_optimizer_state"] = self.optimizer.reference_state_dict(self.optimizer_parameter_names())
        tmp = filename + ".tmp"
        torch.save(state, tmp)
        os.replace(tmp, filename)                                   # a reader never sees a half-written file

    def load_checkpoint(self, filename, reset_optimizer=False, reset_lr_scheduler=False, allow_non_strict_loading=False):
        """-> extra_state of the file, or None when it does not exist (trainer.py:189-266).  As in the reference, --reset-optimizer
        skips the optimizer state AND the update counter / schedule position (they ride on the same branch, trainer.py:232-251)."""
        if not os.path.isfile(filename):
            return None
        state = torch.load(filename, map_location="cpu", weights_only=False)
        state = self.model.raw_state_dict_upgrade(state)
        self.model.load_state_dict(state["model"], strict=not allow_non_strict_loading, args=self.args)
        if any(True for _ in self.criterion.parameters()) and "criterion" in state:
            self.criterion.load_state_dict(state["criterion"], strict=True)      # its parameters alias arena slices
        self.arena.refresh_shadow()
        self._optim_history = state.get("optimizer_history", [])
        last = state.get("last_optimizer_state")
        if last is not None and not reset_optimizer:
            h = self._optim_history[-1]
            assert h["criterion_name"] == self.criterion.__class__.__name__, \
                "Criterion does not match; please reset the optimizer (--reset-optimizer)."
            assert h["optimizer_name"] in ("FairseqAdam", "ArenaAdam"), \
                "Optimizer does not match; please reset the optimizer (--reset-optimizer)."
            self.optimizer.load_state_dict(last, self.optimizer_parameter_names())
            self.num_updates = int(h["num_updates"])
            if not reset_lr_scheduler:
                self.lr_scheduler.load_state_dict(h.get("lr_scheduler_state") or {})
            self.lr_scheduler.step_update(self.num_updates)
        return state.get("extra_state")

    def reduce_stats(self):
        """Materialise (one sync) and sum the logging outputs of the last update."""
        logs, gnorm, total_ss = self._pending
        agg = {k: 0.0 for k in getattr(self, "_log_keys", ())}     # same keys on every rank, also on one whose update was a dummy
        for l in logs:
            for k, v in l.items():
                try:                         # device scalars, python numbers and statistics still on the logging thread all end here
                    agg[k] = agg.get(k, 0.0) + (float(v) if not torch.is_tensor(v) else float(v.item()))
                except Exception as e:
                    raise RuntimeError("logging statistic %r of update %d failed: %s" % (k, self.num_updates, e)) from e
        agg["gnorm"] = float(gnorm.item())
        if self.world > 1:
            g = agg.pop("gnorm")
            agg = D.all_reduce_stats(agg, self.device)
            agg["gnorm"] = g
            if getattr(self.args, "check_grad_norms", False):
                D.check_grad_norms(g, self.device)                                   # R4
        self.last_stats = agg
        return agg
