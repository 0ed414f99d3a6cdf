"""Flat parameter arena in HBM: one f32 master buffer, one f32 gradient buffer, Adam moments and
(bf16 mode) a bf16 shadow that the MFMA GEMMs read.

MI355X-first replacement for fairseq's per-tensor parameter handling on this path: a single Adam
launch over the arena (fairseq/optim/adam.py:147-202 loops over tensors), a single gradient-norm
launch (fairseq/utils.py:253-277 stacks per-tensor norms) and contiguous slices for the bucketed
RCCL all-reduce (fairseq/legacy_distributed_data_parallel.py:96-134 copies grads into a flat buffer
first).  nn.Parameters of the model are views into `master`; their .grad are views into `grad`.
"""
from collections import OrderedDict

import torch

ALIGN = 64          # elements: every parameter starts 256-byte aligned in the f32 arena (128 B in the shadow)


class ParamArena:
    def __init__(self, named_shapes, device, compute_dtype=torch.float32):
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        self.slices = OrderedDict()
        off = 0
        for name, shape in named_shapes.items():
            n = 1
            for s in shape:
                n *= int(s)
            self.slices[name] = (off, n, tuple(int(s) for s in shape))
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.master = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.shadow = (torch.zeros(off, dtype=torch.bfloat16, device=self.device)
                       if compute_dtype == torch.bfloat16 else None)
        self._views = {}
        self.frozen = []               # [start, end) element ranges excluded from the optimizer (--freeze-pretrained)
        # LayerDrop bookkeeping (--encoder-layerdrop / --decoder-layerdrop): the layer groups that CAN be dropped, and the ones that ran
        # in a training forward since the last zero_grad.  A group that never ran has no gradient in the reference (p.grad is None
        # after FairseqOptimizer.zero_grad, fairseq_optimizer.py:97-101) and its Adam skips it (fairseq/optim/adam.py:160-165).
        self.drop_groups = []          # parameter-name prefixes, e.g. "encoder.layers.3."
        self._ran = set()
        self._noted = False

    # ---- frozen parameters (conv_transformer.py:114-121: loaded weights get requires_grad = False and so never reach the
    # optimizer or the gradient norm, fairseq/trainer.py:143).  Here the kernels write gradients for everything; the optimizer
    # zeroes the frozen slices before the norm and launches Adam on the trainable segments only.
    def freeze(self, names):
        r = sorted((self.slices[n][0], self.slices[n][0] + (self.slices[n][1] + ALIGN - 1) // ALIGN * ALIGN) for n in names)
        merged = []
        for a, b in sorted(self.frozen + r):
            if merged and a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        self.frozen = merged

    def trainable_segments(self):
        segs, pos = [], 0
        for a, b in self.frozen:
            if a > pos:
                segs.append((pos, a))
            pos = max(pos, b)
        if pos < self.numel:
            segs.append((pos, self.numel))
        return segs

    def _view(self, buf, name):
        off, n, shape = self.slices[name]
        return buf[off:off + n].view(shape)

    # the views are cached: the engine asks for ~500 of them per update, and building one (slice + view) costs ~2 us of host
    # time on a path where the launch stream is what limits small batches
    def p(self, name):
        """f32 master view."""
        key = ("p", name)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = self._view(self.master, name)
        return v

    def g(self, name):
        """f32 gradient view (kernels accumulate into it)."""
        key = ("g", name)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = self._view(self.grad, name)
        return v

    def w(self, name):
        """compute-dtype view handed to the GEMM kernels (bf16 shadow or the master itself)."""
        key = ("w", name)
        v = self._views.get(key)
        if v is None:
            v = self._view(self.shadow if self.shadow is not None else self.master, name)
            self._views[key] = v
        return v

    def has(self, name):
        return name in self.slices

    def refresh_shadow(self):
        """master -> bf16 shadow (after init / load_state_dict; Adam refreshes it itself every step)."""
        if self.shadow is not None:
            from . import kernels as K
            K.cast(self.master, self.shadow)

    def zero_grad(self):
        self.grad.zero_()
        self._ran.clear()
        self._noted = False

    def note_layers(self, prefixes, keep):
        """a training forward drew LayerDrop decisions: prefixes[i] ran iff keep[i]"""
        for pfx in prefixes:
            if pfx not in self.drop_groups:
                self.drop_groups.append(pfx)
        self._ran.update(pfx for pfx, k in zip(prefixes, keep) if k)
        self._noted = True

    def untouched_groups(self):
        """the droppable groups no forward of this update ran (empty when no LayerDrop forward was noted since zero_grad)"""
        if not self._noted:
            return []
        return [g for g in self.drop_groups if g not in self._ran]

    def group_of(self, name):
        for g in self.drop_groups:
            if name.startswith(g):
                return g
        return None

    def group_runs(self):
        """the arena as maximal runs of consecutive parameters with the same group: [(group or None, start, end)]"""
        runs = []
        for name, (off, n, _) in self.slices.items():
            g = self.group_of(name)
            end = off + (n + ALIGN - 1) // ALIGN * ALIGN
            if runs and runs[-1][0] == g and runs[-1][2] == off:
                runs[-1] = (g, runs[-1][1], end)
            else:
                runs.append((g, off, end))
        return runs

    def ensure_adam_state(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.master)
            self.exp_avg_sq = torch.zeros_like(self.master)

    def slice_of(self, names):
        """[start, end) element range of the arena covering `names` (contiguous by construction order)."""
        offs = [self.slices[n][0] for n in names]
        ends = [self.slices[n][0] + (self.slices[n][1] + ALIGN - 1) // ALIGN * ALIGN for n in names]
        return min(offs), max(ends)
