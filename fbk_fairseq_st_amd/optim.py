"""Optimizer side of the training step on the flat arena.

  FairseqOptimizer surface   fairseq/optim/fairseq_optimizer.py:79-101 (backward / multiply_grads /
                             clip_grad_norm / step / zero_grad / get_lr / set_lr)
  Adam                       fairseq/optim/adam.py:147-202 (one fused launch over all parameters)
  clip_grad_norm_            fairseq/utils.py:253-277 (norm, coefficient and scaling stay on the device)
  inverse_sqrt schedule      fairseq/optim/lr_scheduler/inverse_square_root_schedule.py:36-73
"""
import torch

from . import kernels as K


class ArenaAdam:
    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.arena = arena
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.step_count = 0
        self.group_steps = {}                     # LayerDrop: a droppable layer group's own step count (see step)
        self._scale = 1.0                         # pending multiply_grads factor, folded into the clip / Adam kernels
        self._ws = torch.zeros(1, dtype=torch.float64, device=arena.device)
        self._out2 = torch.ones(2, dtype=torch.float32, device=arena.device)
        self._have_mult = False
        self._divisor = None                      # f64 device scalar the gradients are divided by (set_device_divisor)
        arena.ensure_adam_state()

    def backward(self, loss):
        loss.backward()

    def multiply_grads(self, c):
        """fairseq_optimizer.py:83-87; applied lazily inside the norm / Adam kernels (no extra pass over HBM)."""
        self._scale *= float(c)

    def set_device_divisor(self, t):
        """multiply_grads(1 / t) for a value that exists only on the device: the sample size after its all-reduce over the ranks
        (fairseq/trainer.py:416-430).  Folded into the norm / clip kernel of the same update; the host never reads it."""
        self._divisor = t

    def clip_grad_norm(self, max_norm):
        """Returns the (device) gradient norm after multiply_grads; the clip coefficient stays on the device."""
        for a, b in self.arena.frozen:                # frozen parameters carry no gradient (memset of their slices)
            self.arena.grad[a:b].zero_()
        K.grad_norm_clip(self.arena.grad, self._scale, float(max_norm), self._ws, self._out2, divisor=self._divisor)
        self._have_mult = True
        self._divisor = None
        return self._out2[0]

    def step(self):
        self.step_count += 1
        if not self._have_mult:
            assert self._divisor is None, "a device-side divisor needs clip_grad_norm before step (it is folded in there)"
            self._out2[1] = self._scale
        a = self.arena
        segs = a.trainable_segments() if a.frozen else [(0, a.numel)]
        if a.drop_groups:
            # LayerDrop: the reference's Adam skips a parameter whose gradient is None -- a layer no forward of this update ran: no
            # moment decay, no weight decay, and its own `step` (the bias corrections) does not advance (fairseq/optim/adam.py:160-165).
            # Here: one launch per run of consecutive parameters with the same step count, none for the groups that did not run.
            skipped = set(a.untouched_groups())
            for g in a.drop_groups:
                self.group_steps.setdefault(g, self.step_count - 1)
                if g not in skipped:
                    self.group_steps[g] += 1
            runs = []
            for g, s, e in a.group_runs():
                if g in skipped:
                    continue
                st = self.group_steps[g] if g is not None else self.step_count
                if runs and runs[-1][1] == s and runs[-1][2] == st:
                    runs[-1] = (runs[-1][0], e, st)
                else:
                    runs.append((s, e, st))
            launches = [(max(s, fs), min(e, fe), st) for s, e, st in runs for fs, fe in segs if max(s, fs) < min(e, fe)]
        else:
            # one launch over the whole arena, or one per trainable segment when parameters are frozen (no moments, no weight decay,
            # no update for those: they are not in the reference's optimizer at all)
            launches = [(s, e, self.step_count) for s, e in segs]
        for s, e, st in launches:
            K.adam_step(a.master[s:e], a.grad[s:e], a.exp_avg[s:e], a.exp_avg_sq[s:e], a.shadow[s:e] if a.shadow is not None else None,
                        self._out2, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, st)
        self._scale, self._have_mult = 1.0, False

    def zero_grad(self):
        self.arena.zero_grad()
        self._scale, self._have_mult = 1.0, False

    def get_lr(self):
        return self.lr

    def set_lr(self, lr):
        self.lr = float(lr)

    def state_dict(self):
        """this build's own compact layout: the two moment vectors in arena order"""
        a = self.arena
        sd = {"step": self.step_count, "lr": self.lr, "exp_avg": a.exp_avg.cpu(), "exp_avg_sq": a.exp_avg_sq.cpu()}
        if self.group_steps:
            sd["group_steps"] = dict(self.group_steps)
        return sd

    def load_state_dict(self, sd, reference_names=None):
        """takes either layout: the compact one above, or the reference's (a torch.optim state dict, see reference_state_dict)"""
        if "state" in sd and "param_groups" in sd:
            return self.load_reference_state_dict(sd, reference_names)
        a = self.arena
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        self.group_steps = {str(k): int(v) for k, v in sd.get("group_steps", {}).items()}
        a.exp_avg.copy_(sd["exp_avg"]); a.exp_avg_sq.copy_(sd["exp_avg_sq"])

    # ---- the reference's optimizer-state layout (fairseq/checkpoint_utils.py:245-286 stores `optimizer.state_dict()` =
    # torch.optim.Optimizer.state_dict() of fairseq/optim/adam.py:110-202): {"state": {i: {"step", "exp_avg", "exp_avg_sq"}},
    # "param_groups": [{"lr", "betas", "eps", "weight_decay", "amsgrad", "params": [0..n-1]}]} where i counts the parameters in the
    # order the trainer collected them (model.named_parameters() then criterion's, requires_grad only: trainer.py:140-146).
    def _moment_views(self, ref_name):
        from .conv_transformer import reference_slot
        a = self.arena
        an, blk, nblk = reference_slot(ref_name)
        if not a.has(an):
            raise KeyError("optimizer state for %r: no parameter %r in the arena" % (ref_name, an))
        m, v = a._view(a.exp_avg, an), a._view(a.exp_avg_sq, an)
        if blk is not None:
            rows = m.shape[0] // nblk
            m, v = m[blk * rows:(blk + 1) * rows], v[blk * rows:(blk + 1) * rows]
        return m, v

    def _step_of(self, ref_name):
        """a reference parameter's own step count: its LayerDrop group's when it has one, else the number of updates"""
        from .conv_transformer import reference_slot
        g = self.arena.group_of(reference_slot(ref_name)[0])
        return self.group_steps.get(g, self.step_count) if g is not None else self.step_count

    def reference_state_dict(self, reference_names):
        state = {}
        if self.step_count > 0:                               # torch creates a parameter's state at its first step
            for i, n in enumerate(reference_names):
                st = self._step_of(n)
                if st <= 0:                                   # a layer that has not run in any update yet: no state in the reference either
                    continue
                m, v = self._moment_views(n)
                state[i] = {"step": st, "exp_avg": m.detach().cpu().clone(), "exp_avg_sq": v.detach().cpu().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(reference_names)))}
        return {"state": state, "param_groups": [group]}

    def load_reference_state_dict(self, sd, reference_names):
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if reference_names is None or len(ids) != len(reference_names):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group "
                             "(%d parameters in the file, %d here)" % (len(ids), len(reference_names or ())))
        from .conv_transformer import reference_slot
        steps, per_group, plain = set(), {}, set()
        with torch.no_grad():
            for i, n in zip(ids, reference_names):
                st = sd["state"].get(i)
                m, v = self._moment_views(n)
                grp = self.arena.group_of(reference_slot(n)[0])
                if st is None:                                # never stepped in the run that wrote the file (gradient always None)
                    m.zero_(); v.zero_()
                    if grp is not None:
                        per_group.setdefault(grp, set()).add(0)
                    continue
                (per_group.setdefault(grp, set()) if grp is not None else plain).add(int(st["step"]))
                if tuple(st["exp_avg"].shape) != tuple(m.shape):
                    raise ValueError("optimizer state of %s has shape %s, the parameter %s" % (n, tuple(st["exp_avg"].shape), tuple(m.shape)))
                m.copy_(st["exp_avg"].to(dtype=m.dtype)); v.copy_(st["exp_avg_sq"].to(dtype=v.dtype))
                steps.add(int(st["step"]))
        # The reference's Adam skips a parameter whose gradient is None -- a layer LayerDrop removed from an update -- so a checkpoint
        # trained with --encoder/--decoder-layerdrop holds a step count per layer (fairseq/optim/adam.py:160-165): kept per droppable
        # group (`group_steps`, see step); everything else shares the number of updates.
        bad = {g: sorted(v) for g, v in per_group.items() if len(v) > 1}
        if bad:
            raise ValueError("optimizer state with different step counts inside one LayerDrop layer: %s" % bad)
        if len(plain) > 1:
            # A checkpoint trained WITH LayerDrop, resumed / fine-tuned with it switched off: this model has no droppable groups, so
            # the per-layer counts of the file have nowhere to live.  The reference loads such a file and keeps every parameter's own
            # count; here all parameters share one Adam launch and therefore one count -- the largest (the number of updates), as the
            # layers that never dropped have.  The bias corrections of the once-dropped layers are then a few steps ahead of the
            # reference's: said aloud instead of refusing the file.
            import warnings
            warnings.warn("optimizer state holds per-layer step counts %s (a run with LayerDrop) but LayerDrop is off here: every "
                          "parameter continues from step %d" % (sorted(plain), max(plain)))
        self.group_steps = {g: next(iter(v)) for g, v in per_group.items()}
        self.step_count = max(steps) if steps else 0
        # hyper-parameters: the running optimizer's win over the file's (fairseq_optimizer.py:62-77, optimizer_overrides)


class InverseSquareRootSchedule:
    def __init__(self, optimizer, lr, warmup_updates=4000, warmup_init_lr=-1):
        self.optimizer = optimizer
        self.lr_peak = lr
        self.warmup_updates = warmup_updates
        self.warmup_init_lr = (0 if warmup_updates > 0 else lr) if warmup_init_lr < 0 else warmup_init_lr
        self.lr_step = (lr - self.warmup_init_lr) / max(warmup_updates, 1)
        self.decay_factor = lr * max(warmup_updates, 1) ** 0.5
        self.step_update(0)

    def state_dict(self):                                  # fairseq_lr_scheduler.py:28-34: only the best validation loss is state
        return {"best": getattr(self, "best", None)}

    def load_state_dict(self, sd):
        self.best = sd.get("best")

    def step_update(self, num_updates):
        if num_updates < self.warmup_updates:
            lr = self.warmup_init_lr + num_updates * self.lr_step
        else:
            lr = self.decay_factor * max(num_updates, 1) ** -0.5
        self.optimizer.set_lr(lr)
        return lr
