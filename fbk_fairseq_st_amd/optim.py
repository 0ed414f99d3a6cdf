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
        self._scale = 1.0                         # pending multiply_grads factor, folded into the clip / Adam kernels
        self._ws = torch.zeros(1, dtype=torch.float64, device=arena.device)
        self._out2 = torch.ones(2, dtype=torch.float32, device=arena.device)
        self._have_mult = False
        arena.ensure_adam_state()

    def backward(self, loss):
        loss.backward()

    def multiply_grads(self, c):
        """fairseq_optimizer.py:83-87; applied lazily inside the norm / Adam kernels (no extra pass over HBM)."""
        self._scale *= float(c)

    def clip_grad_norm(self, max_norm):
        """Returns the (device) gradient norm after multiply_grads; the clip coefficient stays on the device."""
        for a, b in self.arena.frozen:                # frozen parameters carry no gradient (memset of their slices)
            self.arena.grad[a:b].zero_()
        K.grad_norm_clip(self.arena.grad, self._scale, float(max_norm), self._ws, self._out2)
        self._have_mult = True
        return self._out2[0]

    def step(self):
        self.step_count += 1
        if not self._have_mult:
            self._out2[1] = self._scale
        a = self.arena
        # one launch over the whole arena, or one per trainable segment when parameters are frozen (no moments, no weight decay,
        # no update for those: they are not in the reference's optimizer at all)
        for s, e in (a.trainable_segments() if a.frozen else [(0, a.numel)]):
            K.adam_step(a.master[s:e], a.grad[s:e], a.exp_avg[s:e], a.exp_avg_sq[s:e], a.shadow[s:e] if a.shadow is not None else None,
                        self._out2, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count)
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
        return {"step": self.step_count, "lr": self.lr, "exp_avg": a.exp_avg.cpu(), "exp_avg_sq": a.exp_avg_sq.cpu()}

    def load_state_dict(self, sd, reference_names=None):
        """takes either layout: the compact one above, or the reference's (a torch.optim state dict, see reference_state_dict)"""
        if "state" in sd and "param_groups" in sd:
            return self.load_reference_state_dict(sd, reference_names)
        a = self.arena
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
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

    def reference_state_dict(self, reference_names):
        state = {}
        if self.step_count > 0:                               # torch creates a parameter's state at its first step
            for i, n in enumerate(reference_names):
                m, v = self._moment_views(n)
                state[i] = {"step": self.step_count, "exp_avg": m.detach().cpu().clone(), "exp_avg_sq": v.detach().cpu().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(reference_names)))}
        return {"state": state, "param_groups": [group]}

    def load_reference_state_dict(self, sd, reference_names):
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if reference_names is None or len(ids) != len(reference_names):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group "
                             "(%d parameters in the file, %d here)" % (len(ids), len(reference_names or ())))
        steps = set()
        with torch.no_grad():
            for i, n in zip(ids, reference_names):
                st = sd["state"].get(i)
                m, v = self._moment_views(n)
                if st is None:                                # never stepped in the run that wrote the file (gradient always None)
                    m.zero_(); v.zero_()
                    continue
                if tuple(st["exp_avg"].shape) != tuple(m.shape):
                    raise ValueError("optimizer state of %s has shape %s, the parameter %s" % (n, tuple(st["exp_avg"].shape), tuple(m.shape)))
                m.copy_(st["exp_avg"].to(dtype=m.dtype)); v.copy_(st["exp_avg_sq"].to(dtype=v.dtype))
                steps.add(int(st["step"]))
        if len(steps) > 1:
            # The reference's Adam skips a parameter whose gradient is None -- a layer LayerDrop removed from an update: no moment decay,
            # no weight decay, its own `step` does not advance (fairseq/optim/adam.py:160-165) -- so a checkpoint trained with
            # --encoder/--decoder-layerdrop holds different step counts per layer.  The fused arena Adam keeps ONE count (and steps
            # dropped layers with zero gradients: a documented deviation); the largest count is the global update
            # count, which is what the bias corrections of every parameter use from here on.
            import warnings
            warnings.warn("optimizer state with per-parameter step counts %s (LayerDrop): continuing with the largest" % sorted(steps))
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
