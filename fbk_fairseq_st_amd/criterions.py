"""Criteria of the S2T path on fused HIP loss kernels, behind the reference's criterion interface.

  label_smoothed_cross_entropy  fairseq/criterions/label_smoothed_cross_entropy.py:32-99
  ctc_multi_loss                examples/speech_recognition/criterions/ctc_multi_loss.py:107-194
                                (+ CTC_loss.py:101-175 for the CTC branch and its unit-error-rate logging)

Differences that are deliberate (and invisible in the returned values):
  * log-probabilities are never materialised (fused log-softmax + loss + gradient kernels);
  * logging outputs stay device scalars until the trainer reduces them (no .item() per criterion);
  * the CTC unit-error-rate alignment runs in native code on predictions that were already copied to
    the host at the CTC-compression sync point (the reference runs a pure-Python DP on the critical path).
"""
import math

import torch
import torch.nn as nn

from . import kernels as K
from .conv_transformer import anchor_zero_grad, unwrap_model
from .registry import CRITERION_REGISTRY, FairseqCriterion, register_criterion


def _item(v):
    return v.item() if torch.is_tensor(v) else (float(v) if isinstance(v, _Deferred) else v)


class _Deferred:
    """One component of a host-side statistic that is still being computed on the logging thread: a number to whoever sums or
    prints it (float(), +), resolved on first use."""

    def __init__(self, future, index):
        self.future, self.index = future, index

    def __float__(self):
        return float(self.future.result()[self.index])

    def __int__(self):
        return int(float(self))

    def __add__(self, other):
        return float(self) + float(other)

    __radd__ = __add__

    def __mul__(self, other):
        return float(self) * float(other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return float(self) / float(other)

    def __rtruediv__(self, other):
        return float(other) / float(self)

    def __lt__(self, other):
        return float(self) < float(other)

    def __gt__(self, other):
        return float(self) > float(other)

    def __eq__(self, other):
        return float(self) == float(other)

    __hash__ = None

    def __repr__(self):
        return repr(float(self))


def _publish(stats, weights=None, derived=None):
    """Inside a fairseq process the reduced statistics also go where its trainer and progress bar read them, with the weights the
    reference's criteria give them (fairseq/logging/metrics.py: `log_scalar(key, value, weight, round=3)` -- an epoch's or a
    validation run's figure is the WEIGHTED mean over its updates: loss by sample size, nll_loss by tokens) and perplexities as
    derived meters of the averaged nll (label_smoothed_cross_entropy.py:85-87, ctc_multi_loss.py:189-194)."""
    from .registry import inside_fairseq
    if inside_fairseq():
        from fairseq import utils
        from fairseq.logging import metrics
        for k, v in stats.items():
            if derived and k in derived:
                continue
            w = (weights or {}).get(k)
            if w is None:
                metrics.log_scalar(k, float(v))
            else:
                metrics.log_scalar(k, float(v), w, round=3)
        for k, src in (derived or {}).items():
            metrics.log_derived(k, lambda meters, src=src: utils.get_perplexity(meters[src].avg))
    return stats


_LOGGING_THREAD = None


def _in_background(fn, *args):
    """Run fn(*args) on the logging thread.  The unit error rate of the CTC head is a logging number only, but its native
    edit-distance pass takes ~1.3 ms of host time for 64 utterances, right where the launch stream has no lead over the GPU (after
    the host sync of the CTC compression): on the main thread the GPU sat idle through it.  ctypes releases the GIL during the call."""
    global _LOGGING_THREAD
    if _LOGGING_THREAD is None:
        from concurrent.futures import ThreadPoolExecutor
        import atexit
        _LOGGING_THREAD = ThreadPoolExecutor(max_workers=1, thread_name_prefix="s2t-logging")
        atexit.register(_LOGGING_THREAD.shutdown, wait=False)
    return _LOGGING_THREAD.submit(fn, *args)


class _LSCEFn(torch.autograd.Function):
    """loss, nll = label_smoothed_nll_loss(log_softmax(logits.float()), target) with fused gradient."""

    @staticmethod
    def forward(ctx, logits, target, eps, pad):
        B, L, V = logits.shape
        lt = logits.transpose(0, 1)                       # the decoder produces time-major rows: this is a view
        if lt.stride(2) != 1 or lt.stride(0) != B * lt.stride(1):
            lt = lt.contiguous()
        tt = target.t().contiguous()
        sums, dl = K.lsce(lt.reshape(L * B, V), tt.view(-1), eps, pad, want_grad=True)
        ctx.dl, ctx.shape = dl, (L, B, V)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g_loss, g_nll):
        L, B, V = ctx.shape
        dl = K.scale_by_device_scalar(ctx.dl, g_loss.contiguous().float())
        return dl.view(L, B, V).transpose(0, 1), None, None, None


class _CTCFn(torch.autograd.Function):
    """sum over the batch of -log p(target | logits) (zero_infinity) with fused gradient w.r.t. the logits."""

    # The alpha/beta recursion is a latency-bound kernel with one workgroup per utterance: it runs on a side stream, under the
    # decoder forward that the caller issues next on the main stream; `join()` makes the main stream wait for it.
    _side = {}
    _pending = []

    @staticmethod
    def forward(ctx, logits, targets, tgt_len, in_len32, blank, lse=None):
        if logits.stride(2) != 1 or (logits.shape[1] > 1 and logits.stride(0) != logits.shape[1] * logits.stride(1)):
            logits = logits.contiguous()
            lse = None
        targets, tgt_len, in_len32 = targets.contiguous(), tgt_len.contiguous(), in_len32.contiguous()
        dev = logits.device
        main = torch.cuda.current_stream(dev)
        side = _CTCFn._side.get(dev)
        if side is None:
            side = _CTCFn._side[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            # loss only: the gradient pass runs in backward, where the upstream gradient is known (as a device scalar)
            loss, ws, _ = K.ctc_loss(logits, targets, tgt_len, in_len32, blank, defer_grad=True, lse=lse)
            out = loss[0]
        for t in (logits, targets, tgt_len, in_len32) + ((lse,) if lse is not None else ()):
            t.record_stream(side)
        for t in (loss, out) + tuple(w for w in ws if torch.is_tensor(w)):
            t.record_stream(main)
        ev = torch.cuda.Event()
        ev.record(side)
        _CTCFn._pending.append((main, ev))
        ctx.ws = ws
        return out

    @staticmethod
    def join():
        """Call before the loss (or anything else produced by forward) is consumed on the main stream."""
        for main, ev in _CTCFn._pending:
            main.wait_event(ev)
        _CTCFn._pending.clear()

    @staticmethod
    def backward(ctx, g):
        _CTCFn.join()                                  # normally a no-op: the loss was already consumed on the main stream
        return K.ctc_loss_grad(ctx.ws, g.contiguous().float().view(1)), None, None, None, None, None


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the HIP GEMM, for heads owned by a criterion (ctc_aware_model.fc_out)."""

    @staticmethod
    def forward(ctx, x, model, wname, bname, anchor=None):
        eng = model.engine
        T, B, D = x.shape
        x2 = x.contiguous().view(T * B, D)
        w = eng.W(wname)
        y = K.gemm(x2, w, bias=eng.P(bname), out=K.alloc_rows((T * B,), w.shape[0], x2.dtype, x2.device))
        ctx.model, ctx.x2, ctx.names, ctx.shape, ctx.anchor = model, x2, (wname, bname), (T, B, D), anchor
        return y.view(T, B, -1)

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.model.engine
        T, B, D = ctx.shape
        wname, bname = ctx.names
        dy2 = dy.reshape(T * B, -1)
        dx = eng.linear_bwd(dy2 if dy2.stride(1) == 1 else dy2.contiguous(), ctx.x2, wname[: -len(".weight")])
        # the owner's anchor gets a zero gradient when a data-parallel wrapper hooked it (see conv_transformer._EncoderFn.backward)
        a = ctx.anchor
        return dx.view(T, B, D), None, None, None, (anchor_zero_grad(a) if a is not None and a._backward_hooks else None)


@register_criterion("label_smoothed_cross_entropy")
class LabelSmoothedCrossEntropyCriterion(FairseqCriterion):
    def __init__(self, task, sentence_avg, label_smoothing):
        super().__init__(task)
        self.sentence_avg = sentence_avg
        self.eps = label_smoothing

    @classmethod
    def build_criterion(cls, args, task):
        return cls(task, getattr(args, "sentence_avg", False), getattr(args, "label_smoothing", 0.0))

    @staticmethod
    def add_args(parser):
        parser.add_argument("--label-smoothing", default=0.0, type=float, metavar="D")

    def forward(self, model, sample, reduce=True):
        net_output = model(**sample["net_input"])
        loss, nll_loss = self.compute_loss(model, net_output, sample, reduce=reduce)
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        logging_output = {"loss": loss.detach(), "nll_loss": nll_loss.detach(), "ntokens": sample["ntokens"],
                          "nsentences": sample["target"].size(0), "sample_size": sample_size}
        return loss, sample_size, logging_output

    def compute_loss(self, model, net_output, sample, reduce=True):
        assert reduce, "the fused kernel returns the summed loss"
        target = model.get_targets(sample, net_output)
        return _LSCEFn.apply(net_output[0], target, self.eps, self.padding_idx)

    @staticmethod
    def reduce_metrics(logging_outputs):
        loss_sum = sum(_item(l.get("loss", 0)) for l in logging_outputs)
        nll_sum = sum(_item(l.get("nll_loss", 0)) for l in logging_outputs)
        ntokens = sum(_item(l.get("ntokens", 0)) for l in logging_outputs)
        sample_size = sum(_item(l.get("sample_size", 0)) for l in logging_outputs)
        nll = nll_sum / ntokens / math.log(2)
        _publish({"loss": loss_sum / sample_size / math.log(2), "nll_loss": nll, "ppl": 2 ** nll},
                 {"loss": sample_size, "nll_loss": ntokens}, {"ppl": "nll_loss"})
        return {"loss": loss_sum / sample_size / math.log(2), "nll_loss": nll, "ppl": 2 ** nll, "ntokens": ntokens, "sample_size": sample_size}

    @staticmethod
    def logging_outputs_can_be_summed():
        return True


class CTCEncoderWrapperModel(nn.Module):
    """ctc_multi_loss.py:14-46: owns fc_out, used when the encoder does not produce ctc_out itself."""

    def __init__(self, args, ctc_dictionary):
        super().__init__()
        self.fc_out = nn.Linear(args.encoder_embed_dim, len(ctc_dictionary))
        self.ctc_encoder_layer = args.ctc_encoder_layer


@register_criterion("ctc_multi_loss")
class CTCMultiLoss(FairseqCriterion):
    def __init__(self, args, task):
        super().__init__(task)
        assert task.source_dictionary is not None
        self.args = args
        self.ctc_aware_model = CTCEncoderWrapperModel(args, task.source_dictionary)
        for n, p in self.arena_params().items():            # see conv_transformer._register: arena name of a criterion-owned parameter
            p._s2t_extra_name = n
        # The reference's trainer wraps a criterion that owns parameters in its data-parallel class too (fairseq/trainer.py:100-114),
        # and LegacyDistributedDataParallel reduces from a parameter hook: fc_out's gradients are written by the kernels, never by
        # autograd, so this one-element parameter is what the hook can fire on (absent from state_dict(): checkpoints keep the
        # reference's keys).
        self._anchor = nn.Parameter(torch.zeros(1))
        self._anchor._s2t_anchor = True
        self.blank_idx = task.source_dictionary.index("<ctc_blank>")          # ctc_multi_loss.py:103
        self.pad_idx = task.source_dictionary.pad()
        saved = args.criterion
        args.criterion = args.underlying_criterion
        assert saved != args.underlying_criterion
        self.real_criterion = CRITERION_REGISTRY[args.criterion].build_criterion(args, task)
        args.criterion = saved
        self.ctc_weight = args.ctc_weight
        self.sentence_avg = getattr(args, "sentence_avg", False)
        self.use_source_side_sample_size = getattr(args, "use_source_side_sample_size", False)

    @staticmethod
    def add_args(parser):
        parser.add_argument("--use-source-side-sample-size", action="store_true", default=False)
        parser.add_argument("--ctc-encoder-layer", default=6, type=int, metavar="LAYER_NUM")
        parser.add_argument("--ctc-weight", default=1.0, type=float, metavar="W")
        parser.add_argument("--underlying-criterion", type=str, metavar="VAL", required=True)

    def arena_params(self):
        """criterion-owned parameters that must live in the model's arena (optimised together)."""
        return {"criterion.ctc_aware_model.fc_out.weight": self.ctc_aware_model.fc_out.weight,
                "criterion.ctc_aware_model.fc_out.bias": self.ctc_aware_model.fc_out.bias}

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        for k in [k for k in sd if k.endswith("_anchor")]:
            del sd[k]
        return sd

    def load_state_dict(self, state_dict, strict=True):
        """fairseq/trainer.py:215-218 loads the criterion's own parameters; once they live in a model's arena the bf16 copy the
        GEMMs read has to follow"""
        state_dict = dict(state_dict)
        state_dict.setdefault("_anchor", self._anchor.data)
        out = super().load_state_dict(state_dict, strict=strict)
        owner = getattr(self.ctc_aware_model.fc_out.weight, "_s2t_owner", None)
        model = owner() if owner is not None else None
        if model is not None and model.arena is not None:
            model.arena.refresh_shadow()
        return out

    def forward(self, model, sample, reduce=True, log_probs=True):
        ni = {k: v for k, v in sample["net_input"].items() if k != "transcript_prev_output_tokens"}
        model = unwrap_model(model)          # `_ctc_state_layer` below must land on the model, not on a data-parallel wrapper
        enc = model.encoder
        k = self.ctc_aware_model.ctc_encoder_layer
        model._ctc_state_layer = k - 1
        encoder_out = enc(ni["src_tokens"], src_lengths=ni["src_lengths"], return_all_hiddens=True)
        last = enc._last
        if hasattr(encoder_out, "ctc_out"):
            ctc_feat, in_len, in_len_host, pred = encoder_out.ctc_out, last["ctc_lengths"], last["ctc_lengths_host"], last.get("pred_host")
            ctc_lse = last.get("ctc_lse")                # row log-sum-exps of ctc_out, by-product of the compression's arg-max pass
        else:
            ctc_feat = _LinearFn.apply(encoder_out.encoder_states[k - 1], model,
                                       "criterion.ctc_aware_model.fc_out.weight", "criterion.ctc_aware_model.fc_out.bias", self._anchor)
            in_len, in_len_host, pred, ctc_lse = last["lengths"], last["lengths_host"], None, None
        tr, tr_len = sample["transcript_target"], sample["transcript_target_lengths"]
        ctc_loss = _CTCFn.apply(ctc_feat, tr, tr_len, in_len.to(torch.int32), self.blank_idx, ctc_lse)      # side stream
        decoder_out = model.decoder(ni["prev_output_tokens"], encoder_out=encoder_out)               # main stream, concurrently
        # unit error rate (logging only): greedy path + native edit-distance alignment on the host
        if pred is None:
            pred = K.ctc_argmax(ctc_feat.detach())[0].cpu()
        tr_host = sample.get("transcript_target_host")
        tr_host = tr.cpu() if tr_host is None else tr_host
        trl_host = sample.get("transcript_target_lengths_host")
        trl_host = tr_len.cpu() if trl_host is None else trl_host
        uer = _in_background(K.host_ctc_uer, pred, torch.tensor(in_len_host, dtype=torch.int64), tr_host, trl_host, self.blank_idx)
        errors, total = _Deferred(uer, 0), _Deferred(uer, 1)
        ctc_ntokens = int(trl_host.sum())
        if self.sentence_avg:
            ctc_sample_size = sample["target"].size(0)
        elif self.use_source_side_sample_size:
            ctc_sample_size = int(sum(in_len_host))
        else:
            ctc_sample_size = ctc_ntokens
        real_loss, nll = self.real_criterion.compute_loss(model, decoder_out, sample, reduce=reduce)
        real_ss = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        _CTCFn.join()
        loss = self.ctc_weight * ctc_loss + real_loss
        nframes = sample.get("nframes")
        if nframes is None:
            nframes = int(ni["src_lengths"].sum())
        logging_output = {"loss": loss.detach(), "ctc_loss": ctc_loss.detach(), "ntokens": sample["ntokens"],
                          "nsentences": sample["target"].size(0), "sample_size": real_ss, "ctc_errors": errors,
                          "ctc_total": total, "nframes": nframes, "nll_loss": nll.detach()}
        return loss, ctc_sample_size, logging_output                       # (:168 returns the CTC sample size)

    @staticmethod
    def logging_outputs_can_be_summed():
        return True

    @staticmethod
    def reduce_metrics(logging_outputs):
        s = lambda k: sum(_item(l.get(k, 0)) for l in logging_outputs)
        loss_sum, ctc_sum, nll_sum = s("loss"), s("ctc_loss"), s("nll_loss")
        ntokens, sample_size = s("ntokens"), s("sample_size")
        errors, total, nframes = s("ctc_errors"), s("ctc_total"), s("nframes")
        nll = nll_sum / ntokens / math.log(2)
        return _publish({"loss": loss_sum / sample_size / math.log(2), "nll_loss": nll, "ppl": 2 ** nll,
                         "ctc_loss": ctc_sum / sample_size / math.log(2),
                         "ctc_acc": 100.0 - min(errors * 100.0 / max(total, 1), 100.0), "nframes": nframes},
                        {"loss": sample_size, "nll_loss": ntokens, "ctc_loss": sample_size}, {"ppl": "nll_loss"})


class _KDFn(torch.autograd.Function):
    """(1-lambda) * NLL + lambda * KD(tau) summed over non-pad rows, fused with its gradient."""

    @staticmethod
    def forward(ctx, logits, target, tidx, tlog, lam, tau, pad):
        B, L, V = logits.shape
        lt = logits.transpose(0, 1)
        if lt.stride(2) != 1 or lt.stride(0) != B * lt.stride(1):
            lt = lt.contiguous()
        Kt = tidx.shape[-1]
        ti = tidx.transpose(0, 1).contiguous().view(L * B, Kt).long()
        tl = tlog.transpose(0, 1).contiguous().view(L * B, Kt).float()
        s, dl = K.kd_loss(lt.reshape(L * B, V), target.t().contiguous().view(-1), ti, tl, lam, tau, pad)
        ctx.dl, ctx.shape = dl, (L, B, V)
        return s[0]

    @staticmethod
    def backward(ctx, g):
        L, B, V = ctx.shape
        dl = K.scale_by_device_scalar(ctx.dl, g.contiguous().float())
        return dl.view(L, B, V).transpose(0, 1), None, None, None, None, None, None


@register_criterion("knowledge_distillation")
class CrossEntropyKnowledgeDistillationCriterion(FairseqCriterion):
    """fairseq/criterions/knowledge_distillation.py:17-119 (teacher top-K logits come with the sample:
    fairseq/data/knowledge_distillation.py:91-140)."""

    def __init__(self, args, task):
        super().__init__(task)
        self._lambda = args.kd_lambda
        self.temperature = args.kd_temperature
        self.sentence_avg = getattr(args, "sentence_avg", False)

    @staticmethod
    def add_args(parser):
        parser.add_argument("--kd-lambda", default=0.0, type=float, metavar="D")
        parser.add_argument("--kd-temperature", default=1.0, type=float, metavar="D")

    def forward(self, model, sample, reduce=True):
        assert reduce
        net_output = model(**sample["net_input"])
        target = model.get_targets(sample, net_output)
        tidx, tlog = sample["teacher_output"][0], sample["teacher_output"][1]
        loss = _KDFn.apply(net_output[0], target, tidx, tlog, self._lambda, self.temperature, self.padding_idx)
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        return loss, sample_size, {"loss": loss.detach(), "ntokens": sample["ntokens"],
                                   "nsentences": sample["target"].size(0), "sample_size": sample_size}

    @staticmethod
    def logging_outputs_can_be_summed():
        return True

    @staticmethod
    def reduce_metrics(logging_outputs):
        s = lambda k: sum(_item(l.get(k, 0)) for l in logging_outputs)
        loss_sum, ntokens, sample_size = s("loss"), s("ntokens"), s("sample_size")
        out = {"loss": loss_sum / sample_size / math.log(2)}
        nll = loss_sum / ntokens / math.log(2) if sample_size != ntokens else out["loss"]
        out.update(nll_loss=nll, ppl=2 ** nll)
        if sample_size != ntokens:                       # fairseq/criterions/knowledge_distillation.py:114-119
            _publish(out, {"loss": sample_size, "nll_loss": ntokens}, {"ppl": "nll_loss"})
        else:
            _publish({"loss": out["loss"], "ppl": out["ppl"]}, {"loss": sample_size}, {"ppl": "loss"})
        return out


@register_criterion("cross_entropy_dualdecoder")
class CrossEntropyDualDecoder(FairseqCriterion):
    """examples/speech_recognition/criterions/cross_entropy_dualdecoder.py:8-83."""

    def __init__(self, args, task):
        super().__init__(task)
        self.eps = getattr(args, "label_smoothing", 0.0)
        self.sentence_avg = getattr(args, "sentence_avg", False)
        self.auxiliary_loss_weight = getattr(args, "auxiliary_loss_weight", 0.5)
        self.primary_loss_weight = getattr(args, "primary_loss_weight", 0.5)

    @staticmethod
    def add_args(parser):
        parser.add_argument("--primary-loss-weight", default=0.5, type=float, metavar="W")
        parser.add_argument("--auxiliary-loss-weight", default=0.5, type=float, metavar="W")
        parser.add_argument("--label-smoothing", default=0.0, type=float, metavar="D")

    def forward(self, model, sample, reduce=True, log_probs=True):
        assert reduce
        net_output = model(**sample["net_input"])
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        p_loss, p_nll = _LSCEFn.apply(net_output[0][0], model.get_targets(sample, net_output[0]), self.eps, self.padding_idx)
        a_loss, a_nll = _LSCEFn.apply(net_output[1][0], model.get_auxiliary_target(sample, net_output[1]), self.eps, self.padding_idx)
        loss = self.primary_loss_weight * p_loss + self.auxiliary_loss_weight * a_loss
        aux_ntok = sample.get("transcript_ntokens")
        if aux_ntok is None:
            aux_ntok = int(model.get_auxiliary_token_lens(sample).sum())
        log = {"loss": loss.detach(), "primary_loss": p_loss.detach(), "primary_nll_loss": p_nll.detach(),
               "auxiliary_loss": a_loss.detach(), "auxiliary_nll_loss": a_nll.detach(), "ntokens": sample["ntokens"],
               "auxiliary_ntokens": aux_ntok, "nsentences": sample["target"].size(0), "sample_size": sample_size}
        return loss, sample_size, log

    @staticmethod
    def logging_outputs_can_be_summed():
        return True

    @staticmethod
    def reduce_metrics(logging_outputs):
        s = lambda k: sum(_item(l.get(k, 0)) for l in logging_outputs)
        ss, nt, ant = s("sample_size"), s("ntokens"), s("auxiliary_ntokens")
        ln2 = math.log(2)
        out = {"loss": s("loss") / ss / ln2, "primary_loss": s("primary_loss") / ss / ln2,
               "auxiliary_loss": s("auxiliary_loss") / ss / ln2, "primary_nll_loss": s("primary_nll_loss") / nt / ln2,
               "auxiliary_nll_loss": s("auxiliary_nll_loss") / max(ant, 1) / ln2}
        _publish(dict(out, primary_ppl=0.0, auxiliary_ppl=0.0),                 # cross_entropy_dualdecoder.py:77-83
                 {"loss": ss, "primary_loss": ss, "auxiliary_loss": ss, "primary_nll_loss": nt, "auxiliary_nll_loss": ant},
                 {"primary_ppl": "primary_nll_loss", "auxiliary_ppl": "auxiliary_nll_loss"})
        return out
