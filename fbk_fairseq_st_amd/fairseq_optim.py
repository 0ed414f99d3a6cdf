"""`--optimizer adam` of the reference's OWN trainer over the flat arena (imported only inside a fairseq process).

fairseq/trainer.py:140-170 collects `model.parameters()` (+ the criterion's) and asks `fairseq.optim.build_optimizer` for the
registered `adam`.  The reference's FairseqAdam (fairseq/optim/adam.py:21-100) walks per-parameter `.grad` tensors, and
`FairseqOptimizer.zero_grad` (fairseq/optim/fairseq_optimizer.py:97-101) sets every `p.grad = None` -- on this path the
gradients are written by the HIP kernels into the arena's flat buffer, of which `p.grad` is only a view: dropping the view
would leave the reference's Adam with nothing to step.  So, exactly as `registry.REPLACED_CORE_CRITERIA` does for the two core
criteria, the `adam` entry of fairseq's optimizer registry is replaced by the class below: the same FairseqOptimizer surface
(add_args, optimizer_config, get_lr / set_lr, backward, multiply_grads, clip_grad_norm, step, zero_grad, state_dict /
load_state_dict) carried out by `optim.ArenaAdam` -- one fused Adam launch, one gradient-norm launch, the clip coefficient on the
device -- with `zero_grad` zeroing the arena and keeping every `p.grad` aliased.  The class is NAMED FairseqAdam: the trainer
stores and compares `optimizer.__class__.__name__` in checkpoints (fairseq/checkpoint_utils.py:269, fairseq/trainer.py:239-241),
and the state dict is the reference's layout (optim.ArenaAdam.reference_state_dict), so checkpoints cross over in both directions.

Pinned by tests/test_reference_trainer_cpu.py (build container: the reference's Trainer.train_step drives a plug-in model).
"""
import torch

import fairseq.optim as _fs_optim
from fairseq.optim import FairseqOptimizer

from .conv_transformer import reference_slot
from .optim import ArenaAdam


class FairseqAdam(FairseqOptimizer):
    @classmethod
    def build_optimizer(cls, args, params):
        """fairseq/registry.py:40-48 calls this when present: parameters of an arena-homed model get the arena Adam, anything else
        (a reference model trained in a process that merely has this user directory loaded) the reference's own class"""
        params = list(params)
        if _REPLACED is not None and not any(hasattr(p, "_s2t_name") for p in params):
            return _REPLACED(args, params)
        return cls(args, params)

    def __init__(self, args, params):
        super().__init__(args)
        params = list(params)
        owners, extras, foreign = [], {}, []
        for p in params:
            if getattr(p, "_s2t_anchor", False):
                continue                                  # the autograd anchor of the bridges: never receives a gradient
            if hasattr(p, "_s2t_extra_name"):
                extras[p._s2t_extra_name] = p
            elif hasattr(p, "_s2t_name"):
                m = p._s2t_owner()
                if m is not None and all(m is not o for o in owners):
                    owners.append(m)
            else:
                foreign.append(p)
        if len(owners) != 1 or foreign:
            raise ValueError("this `adam` drives the parameters of ONE arena-homed S2T model (+ its criterion's head); got %d models "
                             "and %d foreign parameters -- use the reference's models with the reference's optimizer"
                             % (len(owners), len(foreign)))
        model = self.model = owners[0]
        if getattr(args, "distributed_world_size", 1) > 1 and not getattr(args, "use_bmuf", False) and \
                (getattr(args, "ddp_backend", "c10d") != "no_c10d" or getattr(args, "distributed_wrapper", "DDP") != "DDP"):
            # c10d's reducer counts autograd gradients per parameter and SlowMo averages module parameters its own way: the arena's
            # gradients are written by the kernels, which only the in-place reduction of LegacyDistributedDataParallel picks up
            raise ValueError("data-parallel training of the arena-homed S2T model under fairseq's trainer needs --ddp-backend no_c10d "
                             "(what the reference's paper script uses, README.md:142); got --ddp-backend %s --distributed-wrapper %s"
                             % (getattr(args, "ddp_backend", None), getattr(args, "distributed_wrapper", None)))
        if model.arena is None:
            # fairseq's trainer has already moved the module to its device (trainer.py:51-52); re-home the parameters in one flat
            # arena there.  --compute-dtype bf16 (model flag) selects bf16 storage with f32 accumulation and f32 masters.
            dev = next(iter(model.named_arena_params().values())).device
            bf16 = getattr(args, "compute_dtype", None) == "bf16" or getattr(args, "bf16", False)
            model.materialize(dev, torch.bfloat16 if bf16 else torch.float32, extra=extras or None)
        missing = [n for n in extras if not model.arena.has(n)]
        if missing:
            raise ValueError("criterion parameters %s are not in the model's arena: build the optimizer before the first forward "
                             "(fairseq's trainer does), or materialize the model with extra=criterion.arena_params()" % missing)
        self._params = params
        cfg = self.optimizer_config
        self._adam = ArenaAdam(model.arena, lr=cfg["lr"], betas=cfg["betas"], eps=cfg["eps"], weight_decay=cfg["weight_decay"])
        # parameter order of the state dict = the order the trainer handed them over, under the reference's names
        frozen = set(getattr(model, "_frozen_names", ()) or ())
        self._ref_names = [n for n in model.reference_parameter_names() if reference_slot(n)[0] not in frozen]
        self._ref_names += [n[len("criterion."):] for n in extras]          # criterion.parameters() order, as the trainer chained them

    # ---- the reference's flags (fairseq/optim/adam.py:44-60)
    @staticmethod
    def add_args(parser):
        parser.add_argument("--adam-betas", default="(0.9, 0.999)", metavar="B", help="betas for Adam optimizer")
        parser.add_argument("--adam-eps", type=float, default=1e-8, metavar="D", help="epsilon for Adam optimizer")
        parser.add_argument("--weight-decay", "--wd", default=0.0, type=float, metavar="WD", help="weight decay")
        parser.add_argument("--use-old-adam", action="store_true", default=False, help="accepted and ignored: one fused arena Adam")

    @property
    def optimizer_config(self):
        a = self.args
        return {"lr": a.lr[0], "betas": eval(a.adam_betas) if isinstance(a.adam_betas, str) else tuple(a.adam_betas),
                "eps": a.adam_eps, "weight_decay": a.weight_decay}

    @property
    def optimizer(self):
        raise NotImplementedError("there is no torch.optim.Optimizer behind the arena Adam")

    @property
    def params(self):
        return iter(self._params)

    def __getstate__(self):
        return self.state_dict()

    def get_lr(self):
        return self._adam.get_lr()

    def set_lr(self, lr):
        self._adam.set_lr(lr)

    def backward(self, loss):
        loss.backward()

    def multiply_grads(self, c):
        self._adam.multiply_grads(float(c))

    def clip_grad_norm(self, max_norm, aggregate_norm_fn=None):
        if aggregate_norm_fn is not None:
            raise NotImplementedError("model-parallel norm aggregation is outside the S2T path")
        return self._adam.clip_grad_norm(max_norm)

    def step(self, closure=None):
        self._adam.step()

    def zero_grad(self):
        """zero the flat gradient buffer; every p.grad stays the arena view it is (the kernels write there)"""
        self._adam.zero_grad()
        reset = getattr(self.model.engine, "reset_wgrad", None)
        if reset is not None:
            reset()                                       # trainer.py:392-405 recovers from an OOM by zero_grad() and going on

    def state_dict(self):
        return self._adam.reference_state_dict(self._ref_names)

    def load_state_dict(self, state_dict, optimizer_overrides=None):
        self._adam.load_state_dict(state_dict, self._ref_names)
        if optimizer_overrides:                           # fairseq_optimizer.py:72-77
            for k, v in optimizer_overrides.items():
                if k == "lr":
                    self._adam.set_lr(v)
                elif k == "betas":
                    self._adam.betas = tuple(v)
                elif k in ("eps", "weight_decay"):
                    setattr(self._adam, k, float(v))
        self.model.arena.refresh_shadow()

    @property
    def supports_memory_efficient_fp16(self):
        return False

    @property
    def supports_flat_params(self):
        return False

    def average_params(self):
        pass


_REPLACED = _fs_optim.OPTIMIZER_REGISTRY.get("adam")
_fs_optim.OPTIMIZER_REGISTRY["adam"] = FairseqAdam
