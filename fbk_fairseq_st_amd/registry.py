"""The plug-in surface of the reference: decorator registries and base classes.

Two modes, decided once at import:

* **inside a fairseq process** (`fairseq` is already in `sys.modules`, i.e. the package is being imported by
  `--user-dir`, fairseq/utils.py:344-359, from `options.parse_args_and_arch` :117-120 / `train.main` / `generate._main`):
  the decorators DELEGATE to fairseq's own registries -- `fairseq.models.register_model / register_model_architecture`
  (fairseq/models/__init__.py:51-119), `fairseq.tasks.register_task` (fairseq/tasks/__init__.py:20-53),
  `fairseq.criterions.register_criterion` (fairseq/registry.py:12-62) -- and the base classes ARE fairseq's
  (`BaseFairseqModel`, `FairseqEncoder`, `FairseqIncrementalDecoder`, `FairseqEncoderDecoderModel`, `FairseqCriterion`,
  `FairseqTask`, `EncoderOut`), so `--arch s2t_transformer_m --task speech_translation_with_transcription --criterion
  ctc_multi_loss` resolve through `options.py` / `tasks.setup_task` / `task.build_model` exactly like the reference's
  `examples/speech_recognition` user directory.  Core criteria whose names this path re-implements on fused kernels
  (`label_smoothed_cross_entropy`, `knowledge_distillation`) are REPLACED in fairseq's registry: the reference's versions ask
  the model for materialised log-probabilities, which this path never builds.
* **standalone** (no fairseq in the process: the GPU box, the tests, bench.py): the same names backed by the small
  restatements below (same argument meaning and error behaviour).

This file holds no arithmetic.
"""
import argparse
import sys
from typing import List, NamedTuple, Optional

import torch
import torch.nn as nn
from torch import Tensor

_FS = "fairseq" in sys.modules and hasattr(sys.modules["fairseq"], "__version__") or "fairseq.models" in sys.modules
if _FS:
    import fairseq.criterions as _fs_criterions
    import fairseq.models as _fs_models
    import fairseq.tasks as _fs_tasks
    MODEL_REGISTRY = _fs_models.MODEL_REGISTRY
    ARCH_MODEL_REGISTRY = _fs_models.ARCH_MODEL_REGISTRY
    ARCH_MODEL_INV_REGISTRY = _fs_models.ARCH_MODEL_INV_REGISTRY
    ARCH_CONFIG_REGISTRY = _fs_models.ARCH_CONFIG_REGISTRY
    TASK_REGISTRY = _fs_tasks.TASK_REGISTRY
    CRITERION_REGISTRY = _fs_criterions.CRITERION_REGISTRY
else:
    MODEL_REGISTRY = {}
    ARCH_MODEL_REGISTRY = {}
    ARCH_MODEL_INV_REGISTRY = {}
    ARCH_CONFIG_REGISTRY = {}
    TASK_REGISTRY = {}
    CRITERION_REGISTRY = {}


def inside_fairseq():
    """True when the registries and base classes are fairseq's own (the package was imported through --user-dir)."""
    return bool(_FS)


# ------------------------------------------------------------------ encoder output tuples
if _FS:
    from fairseq.models.fairseq_encoder import EncoderOut          # the generator / reorder code of fairseq checks this very type
else:
    class EncoderOut(NamedTuple):                      # fairseq/models/fairseq_encoder.py:11-21
        encoder_out: Tensor                            # T x B x C
        encoder_padding_mask: Optional[Tensor]         # B x T (True at padding) or None
        encoder_embedding: Optional[Tensor]
        encoder_states: Optional[List[Tensor]]
        src_tokens: Optional[Tensor]
        src_lengths: Optional[Tensor]


class CTCAwareEncoderOut(NamedTuple):              # conv_transformer.py:28-32
    encoder_out: Tensor
    encoder_padding_mask: Optional[Tensor]
    encoder_embedding: Optional[Tensor]
    encoder_states: Optional[List[Tensor]]
    src_tokens: Optional[Tensor]
    src_lengths: Optional[Tensor]
    ctc_out: Tensor                                # T4 x B x V_src logits
    ctc_padding_mask: Optional[Tensor]             # B x T4


# ------------------------------------------------------------------ registries
def register_model(name):
    if _FS:
        return _fs_models.register_model(name)             # duplicate names / wrong base raise ValueError there
    def deco(cls):
        if name in MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model ({})".format(name))
        if not issubclass(cls, BaseFairseqModel):
            raise ValueError("Model ({}: {}) must extend BaseFairseqModel".format(name, cls.__name__))
        MODEL_REGISTRY[name] = cls
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    if _FS:
        return _fs_models.register_model_architecture(model_name, arch_name)
    def deco(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError("Cannot register model architecture for unknown model type ({})".format(model_name))
        if arch_name in ARCH_MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model architecture ({})".format(arch_name))
        if not callable(fn):
            raise ValueError("Model architecture must be callable ({})".format(arch_name))
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_MODEL_INV_REGISTRY.setdefault(model_name, []).append(arch_name)
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        return fn
    return deco


def register_task(name):
    if _FS:
        return _fs_tasks.register_task(name)
    def deco(cls):
        if name in TASK_REGISTRY:
            raise ValueError("Cannot register duplicate task ({})".format(name))
        if not issubclass(cls, FairseqTask):
            raise ValueError("Task ({}: {}) must extend FairseqTask".format(name, cls.__name__))
        TASK_REGISTRY[name] = cls
        return cls
    return deco


# core criteria this path re-implements on fused loss kernels: inside fairseq they take the place of the core classes
REPLACED_CORE_CRITERIA = ("label_smoothed_cross_entropy", "knowledge_distillation")


def register_criterion(name):
    def deco(cls):
        if not issubclass(cls, FairseqCriterion):
            raise ValueError("criterion ({}: {}) must extend FairseqCriterion".format(name, cls.__name__))
        if _FS:
            if name in CRITERION_REGISTRY and name in REPLACED_CORE_CRITERIA:
                CRITERION_REGISTRY[name] = cls
                return cls
            return _fs_criterions.register_criterion(name)(cls)
        if name in CRITERION_REGISTRY:
            raise ValueError("Cannot register duplicate criterion ({})".format(name))
        CRITERION_REGISTRY[name] = cls
        return cls
    return deco


# ------------------------------------------------------------------ base classes
# Standalone restatements of the constructors / defaults this path relies on; inside fairseq the real ones.
if _FS:
    _EncoderBase, _DecoderBase, _IncDecoderBase = _fs_models.FairseqEncoder, _fs_models.FairseqDecoder, _fs_models.FairseqIncrementalDecoder
    _ModelBase, _EncDecBase = _fs_models.BaseFairseqModel, _fs_models.FairseqEncoderDecoderModel
    _CriterionBase, _TaskBase = _fs_criterions.FairseqCriterion, _fs_tasks.FairseqTask
else:
    class _EncoderBase(nn.Module):                     # fairseq/models/fairseq_encoder.py:24-91
        def __init__(self, dictionary):
            super().__init__()
            self.dictionary = dictionary

        def forward_torchscript(self, net_input):
            return self.forward_non_torchscript(net_input)

        def forward_non_torchscript(self, net_input):
            return self.forward(**{k: v for k, v in net_input.items() if k != "prev_output_tokens"})

        def reorder_encoder_out(self, encoder_out, new_order):
            raise NotImplementedError

        def max_positions(self):
            return 1e6

        def upgrade_state_dict(self, state_dict):
            return state_dict

    class _DecoderBase(nn.Module):                     # fairseq/models/fairseq_decoder.py
        def __init__(self, dictionary):
            super().__init__()
            self.dictionary = dictionary
            self.onnx_trace = False

        def max_positions(self):
            return 1e6

    class _IncDecoderBase(_DecoderBase):               # fairseq/models/fairseq_incremental_decoder.py:13-96
        def reorder_incremental_state(self, incremental_state, new_order):
            pass

    class _ModelBase(nn.Module):                       # fairseq/models/fairseq_model.py:22-220
        @staticmethod
        def add_args(parser):
            pass

        @classmethod
        def build_model(cls, args, task):
            raise NotImplementedError("Model must implement the build_model method")

        def get_targets(self, sample, net_output):
            return sample["target"]

        def max_positions(self):
            return None

    class _EncDecBase(_ModelBase):                     # fairseq/models/fairseq_model.py:233-307
        def __init__(self, encoder, decoder):
            super().__init__()
            self.encoder = encoder
            self.decoder = decoder
            assert isinstance(self.encoder, _EncoderBase)
            assert isinstance(self.decoder, _DecoderBase)

        def max_positions(self):
            return (self.encoder.max_positions(), self.decoder.max_positions())

        def max_decoder_positions(self):
            return self.decoder.max_positions()

    class _CriterionBase(nn.Module):                   # fairseq/criterions/fairseq_criterion.py
        def __init__(self, task):
            super().__init__()
            self.task = task
            tgt = getattr(task, "target_dictionary", None)
            self.padding_idx = tgt.pad() if tgt is not None else -100

        @staticmethod
        def add_args(parser):
            pass

        @staticmethod
        def logging_outputs_can_be_summed():
            return False

    class _TaskBase(object):                           # fairseq/tasks/fairseq_task.py:14-420
        @staticmethod
        def add_args(parser):
            pass

        def __init__(self, args):
            self.args = args
            self.datasets = {}

        @classmethod
        def setup_task(cls, args, **kwargs):
            return cls(args, **kwargs)

        @property
        def source_dictionary(self):
            raise NotImplementedError

        @property
        def target_dictionary(self):
            raise NotImplementedError

        def load_dataset(self, split, combine=False, **kwargs):
            raise NotImplementedError

        def dataset(self, split):
            if split not in self.datasets:
                raise KeyError("Dataset not loaded: " + split)
            return self.datasets[split]

        def max_positions(self):
            return None

        def logging_outputs_can_be_summed(self, criterion):
            return criterion.logging_outputs_can_be_summed()

        def reduce_metrics(self, logging_outputs, criterion):
            # (inside fairseq the inherited method also logs wpb / wps / bsz before it calls the criterion's: fairseq_task.py:370-404)
            return criterion.__class__.reduce_metrics(logging_outputs)


if _FS:
    from fairseq.data import FairseqDataset      # fairseq_task.py:100-105 type-checks what `task.dataset(split)` hands out
else:
    class FairseqDataset:                          # fairseq/data/fairseq_dataset.py:19-85 (map-style dataset + collater / sizes)
        def set_epoch(self, epoch):
            pass

        @property
        def supports_prefetch(self):
            return False


# ---- what this path adds on top of the bases, identical in both modes
class FairseqEncoder(_EncoderBase):
    pass


class FairseqDecoder(_DecoderBase):
    def get_normalized_probs(self, net_output, log_probs, sample=None):
        raise NotImplementedError


class FairseqIncrementalDecoder(_IncDecoderBase):
    def get_normalized_probs(self, net_output, log_probs, sample=None):
        raise NotImplementedError


class BaseFairseqModel(_ModelBase):
    def set_num_updates(self, num_updates):
        pass

    def make_generation_fast_(self, **kwargs):
        self.eval()

    def raw_state_dict_upgrade(self, state_dict):
        return state_dict


class FairseqEncoderDecoderModel(_EncDecBase, BaseFairseqModel):
    def forward(self, src_tokens, src_lengths, prev_output_tokens, **kwargs):
        encoder_out = self.encoder(src_tokens, src_lengths=src_lengths, **kwargs)
        return self.decoder(prev_output_tokens, encoder_out=encoder_out, **kwargs)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        return self.decoder.get_normalized_probs(net_output, log_probs, sample)


class FairseqCriterion(_CriterionBase):
    @classmethod
    def build_criterion(cls, args, task):
        return cls(args, task)

    def forward(self, model, sample, reduce=True):
        raise NotImplementedError

    @staticmethod
    def reduce_metrics(logging_outputs):
        raise NotImplementedError


class FairseqTask(_TaskBase):
    def build_model(self, args):
        return build_model(args, self)

    def build_criterion(self, args):
        return build_criterion(args, self)

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
        """fairseq_task.py:352-381: forward, (zero the loss for dummy batches), backward."""
        model.train()
        model.set_num_updates(update_num)
        loss, sample_size, logging_output = criterion(model, sample)
        if ignore_grad:
            loss *= 0
        optimizer.backward(loss)
        return loss, sample_size, logging_output

    def valid_step(self, sample, model, criterion):
        model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output

    def build_generator(self, models, args):
        """fairseq_task.py:230-313 (beam search only: sampling / diverse search / scoring are outside the S2T path)."""
        from .sequence_generator import SequenceGenerator
        for flag in ("score_reference", "sampling", "match_source_len"):
            if getattr(args, flag, False):
                raise NotImplementedError("--%s is outside the S2T hot path" % flag.replace("_", "-"))
        if getattr(args, "diverse_beam_groups", -1) > 0 or getattr(args, "diversity_rate", -1) > 0:
            raise NotImplementedError("diverse beam search is outside the S2T hot path")
        return SequenceGenerator(models, self.target_dictionary, beam_size=getattr(args, "beam", 5),
                                 max_len_a=getattr(args, "max_len_a", 0), max_len_b=getattr(args, "max_len_b", 200),
                                 min_len=getattr(args, "min_len", 1), normalize_scores=(not getattr(args, "unnormalized", False)),
                                 len_penalty=getattr(args, "lenpen", 1), unk_penalty=getattr(args, "unkpen", 0),
                                 temperature=getattr(args, "temperature", 1.0),
                                 no_repeat_ngram_size=getattr(args, "no_repeat_ngram_size", 0),
                                 print_alignment=getattr(args, "print_alignment", False))     # fairseq_task.py:300-303

    def inference_step(self, generator, models, sample, prefix_tokens=None):      # fairseq_task.py:392-394
        with torch.no_grad():
            return generator.generate(models, sample, prefix_tokens=prefix_tokens)



def build_model(args, task):
    return ARCH_MODEL_REGISTRY[args.arch].build_model(args, task)      # fairseq/models/__init__.py:47-48


def build_criterion(args, task):
    return CRITERION_REGISTRY[args.criterion].build_criterion(args, task)


def setup_task(args, **kwargs):
    return TASK_REGISTRY[args.task].setup_task(args, **kwargs)


def apply_arch(args):
    """options.parse_args_and_arch tail (fairseq/options.py:191-192): fill the arch defaults in place."""
    if getattr(args, "arch", None) in ARCH_CONFIG_REGISTRY:
        ARCH_CONFIG_REGISTRY[args.arch](args)
    return args


def namespace(**kw):
    return argparse.Namespace(**kw)
