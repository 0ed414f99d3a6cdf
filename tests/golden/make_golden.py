#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REAL reference.

Runs only in the build container (needs /root/reference; never at test time):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Everything here is our own glue: it builds the reference's `conv_transformer`
through its public registry/CLI surface, loads deterministic synthetic weights
(oracle.s2t_ref.make_weights -- the tests regenerate the same weights from the
seed, so the fixtures hold only inputs and expected outputs), runs it on the CPU
in fp32 and dumps .npz files.  Shims applied from outside the read-only tree are
the ones listed in SURVEY.md 8-c (numpy aliases, EncoderOut._field_types, drop of
`transcript_prev_output_tokens` for single-decoder models).  All dropout rates are
0 on the command line and F.dropout is patched to the identity (the subsampler's
rate is max(p, 0.1), conv_transformer.py:214; torch's fused SDPA applies its own
dropout in C++, hence the zero rates) so train-mode BatchNorm statistics can be pinned.
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

for _n, _t in (("float", float), ("int", int), ("bool", bool), ("object", object)):
    if not hasattr(np, _n):
        setattr(np, _n, _t)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import fairseq.models.fairseq_encoder as _fe  # noqa: E402

if not hasattr(_fe.EncoderOut, "_field_types"):
    _fe.EncoderOut._field_types = dict(_fe.EncoderOut.__annotations__)

from fairseq import options, utils  # noqa: E402
from fairseq.data import Dictionary  # noqa: E402

utils.import_user_module(argparse.Namespace(user_dir=REF + "/examples/speech_recognition"))
from examples.speech_recognition.tasks.speech_translation_ctc import SpeechTranslationCTCTask  # noqa: E402
from examples.speech_recognition.data.collaters import Seq2SeqCollater  # noqa: E402
from examples.speech_recognition.data.transcription_dataset import TranscriptionWrapperDataset  # noqa: E402
from examples.speech_recognition.models.conv_transformer import CTCCompressStrategy  # noqa: E402
from fairseq.optim.adam import Adam  # noqa: E402

from oracle import s2t_ref  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(4)

# identity dropout (our patch, outside the tree): pins train-mode BN without RNG
_real_dropout = F.dropout
F.dropout = lambda x, p=0.5, training=True, inplace=False: x


def mk_dict(n):
    d = Dictionary()
    for i in range(n):
        d.add_symbol("w%d" % i)
    return d


def build(cfgname, D, H, Ff, EL, DL, ctc_layer, compress=True, strategy="avg", arch="conv_transformer",
          criterion=("ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy"), extra=(), attn_2d=False, set_args=None):
    a = ["/nonexistent", "--user-dir", REF + "/examples/speech_recognition",
         "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de",
         "--arch", arch] + ([] if attn_2d else ["--no-attn-2d"]) + ["--input-feat-per-channel", "80",
         "--encoder-embed-dim", str(D), "--decoder-embed-dim", str(D),
         "--decoder-output-dim", str(D),
         "--encoder-ffn-embed-dim", str(Ff), "--decoder-ffn-embed-dim", str(Ff),
         "--encoder-attention-heads", str(H), "--decoder-attention-heads", str(H),
         "--encoder-layers", str(EL), "--decoder-layers", str(DL),
         "--ctc-compress-strategy", strategy,
         "--criterion", criterion[0]] + list(criterion[1:]) + list(extra) + [
         "--max-sentences", "4", "--cpu",
         "--dropout", "0.0", "--attention-dropout", "0.0", "--relu-dropout", "0.0",
         "--max-source-positions", "2000", "--max-target-positions", "1000"]
    if criterion[0] == "ctc_multi_loss":
        a += ["--ctc-encoder-layer", str(ctc_layer), "--label-smoothing", "0.1"]
    if compress:
        a.append("--ctc-compress-out")
    args = options.parse_args_and_arch(options.get_training_parser(), input_args=a)
    for k, v in (set_args or {}).items():       # options the reference reads with getattr but registers no flag for (layernorm_embedding)
        setattr(args, k, v)
    tgt, src = mk_dict(96), mk_dict(59)
    src.add_symbol("<ctc_blank>")           # speech_translation_ctc.py:42-46
    task = SpeechTranslationCTCTask(args, tgt)
    task.src_dict = src
    torch.manual_seed(1)
    model = task.build_model(args)
    crit = task.build_criterion(args)
    return args, task, model, crit, len(src), len(tgt)


def load_weights(model, crit, W):
    sd = model.state_dict()
    for k in sd:
        if k in W:
            sd[k] = W[k].clone()
    model.load_state_dict(sd, strict=True)
    if "criterion.ctc_aware_model.fc_out.weight" in W:
        crit.ctc_aware_model.fc_out.weight.data.copy_(W["criterion.ctc_aware_model.fc_out.weight"])
        crit.ctc_aware_model.fc_out.bias.data.copy_(W["criterion.ctc_aware_model.fc_out.bias"])


def make_sample(seed, lens, tgt_lens, tr_lens, V_src, V_tgt, blank):
    rs = np.random.RandomState(seed)
    B, T = len(lens), max(lens)
    x = np.zeros((B, T, 80), np.float32)
    for b, l in enumerate(lens):
        x[b, :l] = rs.randn(l, 80).astype(np.float32)

    def toks(ls, V, hi_excl):
        L = max(ls)
        tgt = np.full((B, L), 1, np.int64)
        prev = np.full((B, L), 1, np.int64)
        for b, l in enumerate(ls):
            t = rs.randint(4, hi_excl, size=l - 1)
            tgt[b, : l - 1] = t
            tgt[b, l - 1] = 2
            prev[b, 0] = 2
            prev[b, 1:l] = t
        return tgt, prev

    tgt, prev = toks(tgt_lens, V_tgt, V_tgt)
    tr, _ = toks(tr_lens, V_src, blank)       # transcripts never contain the blank
    return dict(
        id=np.arange(B, dtype=np.int64), ntokens=int(sum(tgt_lens)), nsentences=B,
        src_tokens=x, src_lengths=np.array(lens, np.int64), prev_output_tokens=prev,
        target=tgt, target_lengths=np.array(tgt_lens, np.int64),
        transcript_target=tr, transcript_target_lengths=np.array(tr_lens, np.int64))


def to_ref_sample(s):
    t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in s.items()}
    return dict(id=t["id"], ntokens=t["ntokens"], nsentences=t["nsentences"],
                net_input=dict(src_tokens=t["src_tokens"], src_lengths=t["src_lengths"],
                               prev_output_tokens=t["prev_output_tokens"]),     # F6: no transcript_prev_*
                target=t["target"], target_lengths=t["target_lengths"],
                transcript_target=t["transcript_target"],
                transcript_target_lengths=t["transcript_target_lengths"])


GRAD_KEYS = ["encoder.convolutions.0.weight", "encoder.convolutions.0.bias", "encoder.convolutions.1.bias",
             "encoder.bn.0.weight", "encoder.bn.0.bias", "encoder.bn.1.weight", "encoder.bn.1.bias",
             "encoder.fc3.bias", "encoder.layers.0.self_attn.q_proj.bias", "encoder.layers.0.self_attn.k_proj.weight",
             "encoder.layers.0.self_attn_layer_norm.weight", "encoder.layers.0.fc1.bias",
             "encoder.layers.1.fc2.weight", "encoder.layer_norm.weight", "encoder.ctc_fc.bias",
             "decoder.embed_tokens.weight", "decoder.layers.0.encoder_attn.k_proj.weight",
             "decoder.layers.0.encoder_attn.q_proj.bias", "decoder.layers.0.self_attn.v_proj.weight",
             "decoder.layer_norm.bias", "decoder.output_projection.weight"]


def run_model_case(name, D, H, Ff, EL, DL, ctc_layer, compress, lens, tgt_lens, tr_lens, seed, opt_steps=0):
    args, task, model, crit, V_src, V_tgt = build(name, D, H, Ff, EL, DL, ctc_layer, compress)
    blank = task.source_dictionary.index("<ctc_blank>")
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL,
                              ctc_layer=ctc_layer if compress else 0)
    shapes = s2t_ref.param_shapes(dict(cfg, ctc_layer=ctc_layer if compress else 0), V_src, V_tgt,
                                  criterion_fc=True)
    W = s2t_ref.make_weights(shapes, seed)
    load_weights(model, crit, W)
    s = make_sample(seed + 1, lens, tgt_lens, tr_lens, V_src, V_tgt, blank)
    sample = to_ref_sample(s)
    out = {("in_" + k): v for k, v in s.items() if isinstance(v, np.ndarray)}
    out["in_ntokens"] = np.int64(s["ntokens"])
    out["meta"] = np.array([D, H, Ff, EL, DL, ctc_layer, int(compress), V_src, V_tgt, blank, seed], np.int64)

    # ---- train mode (BN batch statistics), dropout = identity
    model.train(); crit.train()
    trace = {}
    hooks = []
    enc = model.encoder
    def _rec(key):
        def h(m, i, o):
            trace[key] = o.detach()
        return h
    hooks.append(enc.bn[0].register_forward_hook(_rec("conv0")))
    hooks.append(enc.bn[1].register_forward_hook(_rec("conv1")))
    hooks.append(enc.fc3.register_forward_hook(_rec("fc3_pre_act")))
    for li, layer in enumerate(enc.layers):
        def _hook(m, i, o, li=li):
            trace["enc_in%d" % li] = i[0].detach()
            trace["enc_layer%d" % li] = o.detach()
        hooks.append(layer.register_forward_hook(_hook))
    model.zero_grad(); crit.zero_grad()
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    for h in hooks:
        h.remove()
    out["train_loss"] = np.float64(loss.item())
    out["train_sample_size"] = np.int64(sample_size)
    for k, v in log.items():
        out["train_log_" + k] = np.float64(float(v))
    for k, v in trace.items():
        out["train_" + k] = v.numpy()
    gn = {}
    for k, p in list(model.named_parameters()) + [("criterion." + k, p) for k, p in crit.named_parameters()]:
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        gn[k] = float(g.norm())
        if k in GRAD_KEYS:
            out["grad_" + k] = g.numpy().copy()
    out["gradnorm_keys"] = np.array(sorted(gn))
    out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)], np.float64)

    # ---- fresh model for the encoder/decoder tensors in train mode (single BN update)
    args2, task2, model2, crit2, _, _ = build(name, D, H, Ff, EL, DL, ctc_layer, compress)
    load_weights(model2, crit2, W)
    model2.train()
    with torch.no_grad():
        eo = model2.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"],
                            return_all_hiddens=True)
        logits, _ = model2.decoder(sample["net_input"]["prev_output_tokens"], encoder_out=eo)
    out["train_encoder_out"] = eo.encoder_out.numpy()
    out["train_src_lengths_out"] = eo.src_lengths.numpy()
    out["train_pad_mask"] = (eo.encoder_padding_mask.numpy() if eo.encoder_padding_mask is not None
                             else np.zeros((0,), bool))
    if compress:
        out["train_ctc_out"] = eo.ctc_out.numpy()
        out["train_ctc_pad_mask"] = (eo.ctc_padding_mask.numpy() if eo.ctc_padding_mask is not None
                                     else np.zeros((0,), bool))
    out["train_logits"] = logits.numpy()
    for i in range(2):
        out["train_bn%d_running_mean" % i] = model2.encoder.bn[i].running_mean.numpy().copy()
        out["train_bn%d_running_var" % i] = model2.encoder.bn[i].running_var.numpy().copy()

    # ---- eval mode (running statistics as loaded)
    args3, task3, model3, crit3, _, _ = build(name, D, H, Ff, EL, DL, ctc_layer, compress)
    load_weights(model3, crit3, W)
    model3.eval(); crit3.eval()
    with torch.no_grad():
        eo = model3.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
        logits, _ = model3.decoder(sample["net_input"]["prev_output_tokens"], encoder_out=eo)
        l3, ss3, log3 = crit3(model3, sample)
    out["eval_encoder_out"] = eo.encoder_out.numpy()
    out["eval_src_lengths_out"] = eo.src_lengths.numpy()
    out["eval_logits"] = logits.numpy()
    out["eval_loss"] = np.float64(l3.item())
    for k, v in log3.items():
        out["eval_log_" + k] = np.float64(float(v))

    # ---- optimizer steps (trainer.py:416-443 order: grads * (1/sample_size), clip, Adam)
    if opt_steps:
        args4, task4, model4, crit4, _, _ = build(name, D, H, Ff, EL, DL, ctc_layer, compress)
        load_weights(model4, crit4, W)
        model4.train(); crit4.train()
        params = [p for p in list(model4.parameters()) + list(crit4.parameters()) if p.requires_grad]
        opt = Adam(params, lr=5e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-4)
        gnorms, losses = [], []
        for it in range(opt_steps):
            opt.zero_grad()
            l4, ss4, _ = crit4(model4, sample)
            l4.backward()
            for p in params:
                if p.grad is not None:
                    p.grad.data.mul_(1.0 / float(ss4))
            gnorm = utils.clip_grad_norm_(params, 0.5)
            opt.step()
            gnorms.append(float(gnorm)); losses.append(float(l4))
        out["opt_gnorms"] = np.array(gnorms); out["opt_losses"] = np.array(losses)
        sd = model4.state_dict()
        for k in GRAD_KEYS:
            if k in sd:
                out["opt_param_" + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", loss.item(), "sample_size", sample_size, {k: float(v) for k, v in log.items()})


def run_ctc_cases():
    """G3: crafted CTC-compression cases through the reference's own
    average_same_ctc_features with ctc_fc = identity (so x IS the logit tensor)."""
    D = 64
    args, task, model, crit, V_src, V_tgt = build("ctc", D, 2, 128, 2, 1, 1, True)
    enc = model.encoder
    assert V_src == D
    with torch.no_grad():
        enc.ctc_fc.weight.copy_(torch.eye(D)); enc.ctc_fc.bias.zero_()
    rs = np.random.RandomState(7)
    T, B = 12, 5
    x = (0.1 * rs.randn(T, B, D)).astype(np.float32)
    lens = np.array([12, 12, 9, 7, 1], np.int64)

    def peak(b, t, tok, val=5.0):
        x[t, b, tok] = val
    for t in range(T):                      # b0: all frames predict token 3
        peak(0, t, 3)
    for t in range(T):                      # b1: alternating 4,5,4,5...
        peak(1, t, 4 + (t % 2))
    for t in range(T):                      # b2: runs 3+3+3 valid; the run would continue past len=9
        peak(2, t, 10 + min(t // 3, 2))
    for t in range(T):                      # b3: exact ties between tokens 8 and 20 -> first index (8)
        peak(3, t, 8); peak(3, t, 20)
    x[2, 3, 8] = 4.0                        #     one frame where 20 wins, splitting the run
    peak(4, 0, 63)                          # b4: single valid frame, predicts the blank (last symbol)
    out = dict(x=x, lens=lens)
    xt = torch.from_numpy(x).requires_grad_(True)
    for strat in ("avg", "weighted", "softmax"):
        enc.ctc_compress_method = getattr(CTCCompressStrategy, strat)
        x_ctc, comp, new_len = enc.average_same_ctc_features(xt, torch.from_numpy(lens))
        out[strat + "_out"] = comp.detach().numpy()
        out[strat + "_new_len"] = new_len.numpy()
        g, = torch.autograd.grad(comp.pow(2).sum(), xt)
        out[strat + "_grad_x"] = g.numpy()
    prob = F.softmax(torch.from_numpy(x), dim=-1).transpose(0, 1)
    out["pred"] = np.stack([prob[b].argmax(-1).numpy() for b in range(B)]).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "ctc_compress.npz"), **out)
    print("ctc cases new_len", out["avg_new_len"])


def run_collate():
    """G10: the reference's known-answer collater test inputs (tests/speech_recognition/
    test_collaters.py:24-49) plus ragged ones with equal lengths, and the transcript wrapper."""
    rs = np.random.RandomState(3)
    col = Seq2SeqCollater(0, 1, pad_index=1, eos_index=2, move_eos_to_beginning=True)
    lens = [5, 9, 9, 3, 7, 9]
    samples, tr = [], []
    for i, l in enumerate(lens):
        tl = 2 + (i * 3) % 5
        tgt = np.concatenate([rs.randint(4, 50, size=tl), [2]]).astype(np.int64)
        samples.append({"id": i, "data": [rs.randn(l, 4).astype(np.float32), tgt]})
        tr.append(torch.from_numpy(np.concatenate([rs.randint(4, 30, size=1 + i % 3), [2]]).astype(np.int64)))
    batch = col.collate(samples)
    out = {}
    for i, s in enumerate(samples):
        out["s%d_src" % i] = s["data"][0]; out["s%d_tgt" % i] = s["data"][1]; out["s%d_tr" % i] = tr[i].numpy()
    out["n"] = np.int64(len(samples))
    out["id"] = batch["id"].numpy(); out["ntokens"] = np.int64(batch["ntokens"])
    out["src_tokens"] = batch["net_input"]["src_tokens"].numpy()
    out["src_lengths"] = batch["net_input"]["src_lengths"].numpy()
    out["prev_output_tokens"] = batch["net_input"]["prev_output_tokens"].numpy()
    out["target"] = batch["target"].numpy(); out["target_lengths"] = batch["target_lengths"].numpy()

    class _Tgt:
        def collater(self, ss):
            return col.collate(ss)
    d = mk_dict(40)
    wrap = TranscriptionWrapperDataset(_Tgt(), None, d)
    ws = []
    for i, s in enumerate(samples):
        s2 = dict(s); s2["transcript_target"] = tr[i]; ws.append(s2)
    wb = wrap.collater(ws)
    out["transcript_target"] = wb["transcript_target"].numpy()
    out["transcript_target_lengths"] = wb["transcript_target_lengths"].numpy()
    out["transcript_prev_output_tokens"] = wb["net_input"]["transcript_prev_output_tokens"].numpy()
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **out)
    print("collate order", out["id"])


def run_uer():
    """compute_ctc_uer known answers (CTC_loss.py:31-74) on crafted predictions."""
    from examples.speech_recognition.criterions.CTC_loss import compute_ctc_uer
    rs = np.random.RandomState(11)
    B, T, V, L = 6, 20, 12, 9
    blank = V - 1
    lp = torch.from_numpy(rs.randn(B, T, V).astype(np.float32)).log_softmax(-1)
    in_len = torch.tensor([20, 17, 11, 5, 20, 1])
    tgt_len = torch.tensor([9, 4, 6, 2, 1, 3])
    tgt = torch.from_numpy(rs.randint(0, V - 1, size=(B, L)).astype(np.int64))
    e, n = compute_ctc_uer(lp, tgt, in_len, tgt_len, blank)
    np.savez_compressed(os.path.join(OUT, "ctc_uer.npz"), lp=lp.numpy(), in_len=in_len.numpy(),
                        tgt=tgt.numpy(), tgt_len=tgt_len.numpy(), blank=np.int64(blank),
                        errors=np.float64(e), total=np.float64(n))
    print("uer", e, n)


def run_kd_case():
    """G7a: word-level knowledge distillation (fairseq/criterions/knowledge_distillation.py) on a single-decoder model."""
    D, H, Ff, EL, DL, seed = 64, 2, 128, 2, 2, 400
    args, task, model, crit, V_src, V_tgt = build("kd", D, H, Ff, EL, DL, 0, False,
                                                  criterion=("knowledge_distillation", "--kd-lambda", "0.6", "--kd-temperature", "2.0"))
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt), seed)
    load_weights(model, crit, W)
    s = make_sample(seed + 1, [44, 39, 30], [6, 4, 5], [3, 3, 3], V_src, V_tgt, V_src - 1)
    rs = np.random.RandomState(9)
    B, L = s["target"].shape
    Kt = 8
    tidx = np.stack([[rs.choice(V_tgt, Kt, replace=False) for _ in range(L)] for _ in range(B)]).astype(np.int64)
    tlog = rs.randn(B, L, Kt).astype(np.float32) * 2
    sample = to_ref_sample(s)
    sample["teacher_output"] = [torch.from_numpy(tidx), torch.from_numpy(tlog)]
    model.train(); crit.train()
    model.zero_grad()
    loss, ss, log = crit(model, sample)
    loss.backward()
    out = {("in_" + k): v for k, v in s.items() if isinstance(v, np.ndarray)}
    out.update(in_ntokens=np.int64(s["ntokens"]), teacher_idx=tidx, teacher_logits=tlog,
               meta=np.array([D, H, Ff, EL, DL, 0, 0, V_src, V_tgt, V_src - 1, seed], np.int64),
               loss=np.float64(loss.item()), sample_size=np.int64(ss))
    gn = {k: float((p.grad if p.grad is not None else torch.zeros_like(p)).norm()) for k, p in model.named_parameters()}
    out["gradnorm_keys"] = np.array(sorted(gn)); out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)], np.float64)
    np.savez_compressed(os.path.join(OUT, "kd.npz"), **out)
    print("kd loss", loss.item(), ss)


def run_dual_case():
    """G7b: conv_transformer_dualdecoder + cross_entropy_dualdecoder (a18)."""
    D, H, Ff, EL, DL, seed = 64, 2, 128, 2, 2, 500
    args, task, model, crit, V_src, V_tgt = build("dual", D, H, Ff, EL, DL, 0, False, arch="conv_transformer_dualdecoder",
                                                  criterion=("cross_entropy_dualdecoder", "--label-smoothing", "0.1"))
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, V_aux=V_src), seed)
    load_weights(model, crit, W)
    s = make_sample(seed + 1, [52, 41, 33], [7, 5, 6], [6, 4, 5], V_src, V_tgt, V_src - 1)
    sample = to_ref_sample(s)
    tr = torch.from_numpy(s["transcript_target"]); trl = s["transcript_target_lengths"]
    prev = torch.full_like(tr, 1)
    for b in range(tr.shape[0]):
        l = int(trl[b]); prev[b, 0] = 2; prev[b, 1:l] = tr[b, : l - 1]
    sample["net_input"]["transcript_prev_output_tokens"] = prev
    model.train(); crit.train()
    model.zero_grad()
    loss, ss, log = crit(model, sample)
    loss.backward()
    out = {("in_" + k): v for k, v in s.items() if isinstance(v, np.ndarray)}
    out.update(in_ntokens=np.int64(s["ntokens"]), in_transcript_prev_output_tokens=prev.numpy(),
               meta=np.array([D, H, Ff, EL, DL, 0, 0, V_src, V_tgt, V_src - 1, seed], np.int64),
               loss=np.float64(loss.item()), sample_size=np.int64(ss))
    for k, v in log.items():
        out["log_" + k] = np.float64(float(v))
    with torch.no_grad():
        (lg, _), (la, _) = model(**sample["net_input"])
    out["logits"] = lg.numpy(); out["aux_logits"] = la.numpy()
    gn = {k: float((p.grad if p.grad is not None else torch.zeros_like(p)).norm()) for k, p in model.named_parameters()}
    out["gradnorm_keys"] = np.array(sorted(gn)); out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)], np.float64)
    np.savez_compressed(os.path.join(OUT, "dual.npz"), **out)
    print("dual loss", loss.item(), ss, {k: float(v) for k, v in log.items()})


def run_lne_case():
    """`layernorm_embedding` (conv_transformer.py:184-187,230-231; fairseq/models/transformer.py:578-581,731-732): a LayerNorm on the
    embedded input of the encoder and of the decoder, before the dropout.  The reference reads it with getattr (its arch functions
    default it to False) and registers no flag for this model, so it is set on the parsed args.  Train-mode loss / gradient norms and the
    eval-mode encoder output and logits."""
    D, H, Ff, EL, DL, ctc_layer, seed = 64, 2, 128, 2, 2, 1, 1500
    args, task, model, crit, V_src, V_tgt = build("lne", D, H, Ff, EL, DL, ctc_layer, True, set_args=dict(layernorm_embedding=True))
    assert model.encoder.layernorm_embedding is not None and model.decoder.layernorm_embedding is not None
    blank = task.source_dictionary.index("<ctc_blank>")
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer, layernorm_embedding=True)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    load_weights(model, crit, W)
    assert all(k in W for k in model.state_dict() if "layernorm_embedding" in k)
    s = make_sample(seed + 1, [57, 44, 31], [6, 5, 4], [5, 4, 3], V_src, V_tgt, blank)
    sample = to_ref_sample(s)
    out = {("in_" + k): v for k, v in s.items() if isinstance(v, np.ndarray)}
    out["in_ntokens"] = np.int64(s["ntokens"])
    out["meta"] = np.array([D, H, Ff, EL, DL, ctc_layer, 1, V_src, V_tgt, blank, seed], np.int64)
    model.train(); crit.train()
    model.zero_grad(); crit.zero_grad()
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    out["train_loss"] = np.float64(loss.item()); out["train_sample_size"] = np.int64(sample_size)
    gn = {k: float((p.grad if p.grad is not None else torch.zeros_like(p)).norm())
          for k, p in list(model.named_parameters()) + [("criterion." + k, p) for k, p in crit.named_parameters()]}
    out["gradnorm_keys"] = np.array(sorted(gn)); out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)], np.float64)
    args2, task2, model2, crit2, _, _ = build("lne", D, H, Ff, EL, DL, ctc_layer, True, set_args=dict(layernorm_embedding=True))
    load_weights(model2, crit2, W)
    model2.eval()
    with torch.no_grad():
        eo = model2.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
        logits, _ = model2.decoder(sample["net_input"]["prev_output_tokens"], encoder_out=eo)
    out["eval_encoder_out"] = eo.encoder_out.numpy(); out["eval_logits"] = logits.numpy()
    print("lne loss", loss.item(), "enc", tuple(eo.encoder_out.shape))
    np.savez_compressed(os.path.join(OUT, "lne.npz"), **out)


def run_generate_case(wide=False):
    """G9: beam search (fairseq/sequence_generator.py + search.py BeamSearch) on eval-mode models with deterministic weights.
    wide=True -> generate_wide.npz: models with 64-wide heads and D a multiple of 256, the shapes the device-resident search of the
    build takes (csrc/decode.hip); `c` ragged with CTC compression, `d` every score option set, `e` one sentence, beam 8, 3 decoder layers."""
    from fairseq.sequence_generator import SequenceGenerator
    out = {}
    cases = [("a", dict(D=64, H=2, Ff=128, EL=3, DL=2, ctc_layer=2, compress=True, seed=600, lens=[61, 50, 37]),
              dict(beam_size=5, max_len_a=0, max_len_b=12, min_len=1)),
             ("b", dict(D=64, H=2, Ff=128, EL=2, DL=2, ctc_layer=0, compress=False, seed=700, lens=[48, 48]),
              dict(beam_size=3, max_len_a=0.1, max_len_b=5, min_len=2, len_penalty=0.6, unk_penalty=0.5, temperature=1.5))]
    if wide:
        cases = [("c", dict(D=256, H=4, Ff=256, EL=2, DL=2, ctc_layer=1, compress=True, seed=610, lens=[61, 50, 37]),
                  dict(beam_size=5, max_len_a=0, max_len_b=12, min_len=1)),
                 ("d", dict(D=256, H=4, Ff=384, EL=2, DL=2, ctc_layer=0, compress=False, seed=710, lens=[48, 48]),
                  dict(beam_size=3, max_len_a=0.1, max_len_b=5, min_len=2, len_penalty=0.6, unk_penalty=0.5, temperature=1.5)),
                 ("e", dict(D=256, H=4, Ff=512, EL=1, DL=3, ctc_layer=0, compress=False, seed=810, lens=[77]),
                  dict(beam_size=8, max_len_a=0, max_len_b=20, min_len=1))]
    for tag, m, g in cases:
        crit = ("ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy") if m["compress"] else \
               ("label_smoothed_cross_entropy", "--label-smoothing", "0.1")
        args, task, model, criterion, V_src, V_tgt = build("gen" + tag, m["D"], m["H"], m["Ff"], m["EL"], m["DL"], m["ctc_layer"],
                                                           m["compress"], criterion=crit)
        cfg = s2t_ref.default_cfg(D=m["D"], heads=m["H"], ffn=m["Ff"], enc_layers=m["EL"], dec_layers=m["DL"],
                                  ctc_layer=m["ctc_layer"] if m["compress"] else 0)
        W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=m["compress"]), m["seed"])
        W["decoder.output_projection.weight"][2] *= 4.0          # make <eos> competitive so hypotheses end at different steps
        load_weights(model, criterion, W)
        s = make_sample(m["seed"] + 1, m["lens"], [4] * len(m["lens"]), [3] * len(m["lens"]), V_src, V_tgt, V_src - 1)
        sample = to_ref_sample(s)
        model.eval()
        gen = SequenceGenerator([model], task.target_dictionary, **g)
        hyps = gen.generate([model], sample)
        B, beam = len(hyps), g["beam_size"]
        Lmax = max(len(h["tokens"]) for hs in hyps for h in hs)
        tok = np.full((B, beam, Lmax), -1, np.int64); sc = np.full((B, beam), np.nan, np.float64)
        ps = np.zeros((B, beam, Lmax), np.float32); nh = np.zeros((B,), np.int64)
        for b, hs in enumerate(hyps):
            nh[b] = len(hs)
            for i, h in enumerate(hs):
                n = len(h["tokens"])
                tok[b, i, :n] = h["tokens"].numpy(); sc[b, i] = float(h["score"]); ps[b, i, :n] = h["positional_scores"].numpy()
        out.update({tag + "_src_tokens": s["src_tokens"], tag + "_src_lengths": s["src_lengths"],
                    tag + "_tokens": tok, tag + "_scores": sc, tag + "_pos_scores": ps, tag + "_nhyp": nh,
                    tag + "_meta": np.array([m["D"], m["H"], m["Ff"], m["EL"], m["DL"], m["ctc_layer"], int(m["compress"]), V_src, V_tgt,
                                             V_src - 1, m["seed"]], np.int64),
                    tag + "_gen": np.array([g["beam_size"], g["max_len_a"], g["max_len_b"], g["min_len"], g.get("len_penalty", 1.0),
                                            g.get("unk_penalty", 0.0), g.get("temperature", 1.0)], np.float64)})
        print("gen", tag, [[(len(h["tokens"]), round(float(h["score"]), 4)) for h in hs] for hs in hyps])
    np.savez_compressed(os.path.join(OUT, "generate_wide.npz" if wide else "generate.npz"), **out)


def run_generate_ext_case():
    """SequenceGenerator features around the plain beam search (fairseq/sequence_generator.py): `e` an ENSEMBLE of two models
    (EnsembleModel.forward_decoder :711-770; configuration of generate.npz case `b` without CTC compression -- the reference averages
    the members' attention maps, so their encoder lengths must agree -- weights from seeds 700 / 750), `p` PREFIX TOKENS of
    different lengths incl. one sentence without (:270-280,449-476) and `n` n-gram blocking (--no-repeat-ngram-size 2, :617-650)
    on the configuration of case `a` (CTC compression after layer 2, seed 600)."""
    from fairseq.sequence_generator import SequenceGenerator

    def mk(tagname, m, seeds):
        crit = ("ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy") if m["compress"] else \
               ("label_smoothed_cross_entropy", "--label-smoothing", "0.1")
        cfg = s2t_ref.default_cfg(D=m["D"], heads=m["H"], ffn=m["Ff"], enc_layers=m["EL"], dec_layers=m["DL"],
                                  ctc_layer=m["ctc_layer"] if m["compress"] else 0)
        models = []
        for seed in seeds:
            args, task, model, criterion, V_src, V_tgt = build("genx%s%d" % (tagname, seed), m["D"], m["H"], m["Ff"], m["EL"], m["DL"],
                                                               m["ctc_layer"], m["compress"], criterion=crit)
            W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=m["compress"]), seed)
            W["decoder.output_projection.weight"][2] *= 4.0
            load_weights(model, criterion, W)
            model.eval()
            models.append(model)
        s = make_sample(seeds[0] + 1, m["lens"], [4] * len(m["lens"]), [3] * len(m["lens"]), V_src, V_tgt, V_src - 1)
        return task, models, s, V_src, V_tgt

    ma = dict(D=64, H=2, Ff=128, EL=3, DL=2, ctc_layer=2, compress=True, lens=[61, 50, 37])
    mb = dict(D=64, H=2, Ff=128, EL=2, DL=2, ctc_layer=0, compress=False, lens=[48, 48, 33])
    task_a, models_a, s_a, V_src, V_tgt = mk("a", ma, (600,))
    task_b, models_b, s_b, _, _ = mk("b", mb, (700, 750))
    prefix = torch.tensor([[17, 45, 9], [33, 1, 1], [1, 1, 1]])            # pad = 1: free from there on
    cases = {"e": (task_b, models_b, s_b, dict(beam_size=4, max_len_a=0, max_len_b=10, min_len=1), None, mb, (700, 750)),
             "p": (task_a, models_a, s_a, dict(beam_size=4, max_len_a=0, max_len_b=10, min_len=1), prefix, ma, (600,)),
             "n": (task_a, models_a, s_a, dict(beam_size=4, max_len_a=0, max_len_b=14, min_len=8, no_repeat_ngram_size=2), None, ma, (600,))}
    out = {"prefix_tokens": prefix.numpy()}
    for tag, (task, ms, s, g, pre, m, seeds) in cases.items():
        gen = SequenceGenerator(ms, task.target_dictionary, **g)
        hyps = gen.generate(ms, to_ref_sample(s), prefix_tokens=pre)
        B, beam = len(hyps), g["beam_size"]
        Lmax = max(len(h["tokens"]) for hs in hyps for h in hs)
        tok = np.full((B, beam, Lmax), -1, np.int64); sc = np.full((B, beam), np.nan, np.float64)
        ps = np.zeros((B, beam, Lmax), np.float32); nh = np.zeros((B,), np.int64)
        for b, hs in enumerate(hyps):
            nh[b] = len(hs)
            for i, h in enumerate(hs):
                n = len(h["tokens"])
                tok[b, i, :n] = h["tokens"].numpy(); sc[b, i] = float(h["score"]); ps[b, i, :n] = h["positional_scores"].numpy()
        out.update({tag + "_src_tokens": s["src_tokens"], tag + "_src_lengths": s["src_lengths"],
                    tag + "_tokens": tok, tag + "_scores": sc, tag + "_pos_scores": ps, tag + "_nhyp": nh,
                    tag + "_meta": np.array([m["D"], m["H"], m["Ff"], m["EL"], m["DL"], m["ctc_layer"], int(m["compress"]), V_src, V_tgt,
                                             V_src - 1] + list(seeds), np.int64),
                    tag + "_gen": np.array([g["beam_size"], g["max_len_a"], g["max_len_b"], g["min_len"], g.get("no_repeat_ngram_size", 0)],
                                           np.float64)})
        print("genx", tag, [[(h["tokens"].tolist(), round(float(h["score"]), 4)) for h in hs[:2]] for hs in hyps])
    np.savez_compressed(os.path.join(OUT, "generate_ext.npz"), **out)


def run_twophase_case(wide=False):
    """wide=True -> twophase_wide.npz (64-wide heads, D 256: the device-resident search of the build).  G18: TwoPhaseSequenceGenerator (examples/speech_recognition/twophase_sequence_generator.py) on the dual-decoder
    model: beam search with the auxiliary (transcript) decoder, then HierarchicalBeamSearch with the target decoder seeded by the
    transcript hypotheses' scores; every target hypothesis carries the transcript it descends from (`aux_tokens`)."""
    from examples.speech_recognition.twophase_sequence_generator import TwoPhaseSequenceGenerator
    out = {}
    cases = [("a", dict(D=64, H=2, Ff=128, EL=2, DL=2, seed=1100, lens=[52, 41, 33]), dict(beam_size=4, max_len_a=0, max_len_b=10, min_len=1)),
             ("b", dict(D=64, H=2, Ff=128, EL=2, DL=1, seed=1200, lens=[44, 44]),
              dict(beam_size=3, max_len_a=0.1, max_len_b=4, min_len=2, len_penalty=0.7, unk_penalty=0.3, temperature=1.3))]
    if wide:
        cases = [("c", dict(D=256, H=4, Ff=256, EL=2, DL=2, seed=1110, lens=[52, 41, 33]), dict(beam_size=4, max_len_a=0, max_len_b=10, min_len=1)),
                 ("d", dict(D=256, H=4, Ff=256, EL=1, DL=1, seed=1210, lens=[44, 44]),
                  dict(beam_size=3, max_len_a=0.1, max_len_b=4, min_len=2, len_penalty=0.7, unk_penalty=0.3, temperature=1.3))]
    for tag, m, g in cases:
        args, task, model, crit, V_src, V_tgt = build("tp" + tag, m["D"], m["H"], m["Ff"], m["EL"], m["DL"], 0, False,
                                                      arch="conv_transformer_dualdecoder",
                                                      criterion=("cross_entropy_dualdecoder", "--label-smoothing", "0.1"))
        cfg = s2t_ref.default_cfg(D=m["D"], heads=m["H"], ffn=m["Ff"], enc_layers=m["EL"], dec_layers=m["DL"], ctc_layer=0)
        W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, V_aux=V_src), m["seed"])
        W["decoder.output_projection.weight"][2] *= 4.0           # competitive <eos>: hypotheses end at different steps
        W["auxiliary_decoder.output_projection.weight"][2] *= 4.0
        load_weights(model, crit, W)
        s = make_sample(m["seed"] + 1, m["lens"], [4] * len(m["lens"]), [3] * len(m["lens"]), V_src, V_tgt, V_src - 1)
        sample = to_ref_sample(s)
        model.eval()
        gen = TwoPhaseSequenceGenerator([model], task.source_dictionary, task.target_dictionary, **g)
        hyps = gen.generate([model], sample)
        B, beam = len(hyps), g["beam_size"]
        Lmax = max(len(h["tokens"]) for hs in hyps for h in hs); Amax = max(len(h["aux_tokens"]) for hs in hyps for h in hs)
        tok = np.full((B, beam, Lmax), -1, np.int64); aux = np.full((B, beam, Amax), -1, np.int64)
        sc = np.full((B, beam), np.nan, np.float64); ps = np.zeros((B, beam, Lmax), np.float32); nh = np.zeros((B,), np.int64)
        for b, hs in enumerate(hyps):
            nh[b] = len(hs)
            for i, h in enumerate(hs):
                n = len(h["tokens"]); na = len(h["aux_tokens"])
                tok[b, i, :n] = h["tokens"].numpy(); aux[b, i, :na] = h["aux_tokens"].numpy()
                sc[b, i] = float(h["score"]); ps[b, i, :n] = h["positional_scores"].numpy()
        out.update({tag + "_src_tokens": s["src_tokens"], tag + "_src_lengths": s["src_lengths"], tag + "_tokens": tok, tag + "_aux_tokens": aux,
                    tag + "_scores": sc, tag + "_pos_scores": ps, tag + "_nhyp": nh,
                    tag + "_meta": np.array([m["D"], m["H"], m["Ff"], m["EL"], m["DL"], 0, 0, V_src, V_tgt, V_src - 1, m["seed"]], np.int64),
                    tag + "_gen": np.array([g["beam_size"], g["max_len_a"], g["max_len_b"], g["min_len"], g.get("len_penalty", 1.0),
                                            g.get("unk_penalty", 0.0), g.get("temperature", 1.0)], np.float64)})
        print("twophase", tag, [[(len(h["tokens"]), len(h["aux_tokens"]), round(float(h["score"]), 4)) for h in hs] for hs in hyps])
    np.savez_compressed(os.path.join(OUT, "twophase_wide.npz" if wide else "twophase.npz"), **out)


def run_data_case():
    """G12: the on-disk TNTIDX format and frame-budget batching.
    * tests/golden/tntidx/{fbank,tokens}.{idx,bin}: tiny datasets written by the reference's own builders
      (examples/speech_recognition/preprocess_audio.py AudioIndexedDatasetBuilder, fairseq/data/indexed_dataset.py
      IndexedDatasetBuilder) + the items as arrays in data.npz;
    * batches of fairseq/data/data_utils_fast.pyx batch_by_size_fast (built out of tree with cython) for several length
      distributions and (max_tokens, max_sentences, multiple) settings."""
    import subprocess, sysconfig, tempfile, importlib.util, shutil
    from fairseq.data.indexed_dataset import IndexedDatasetBuilder, IndexedDataset
    sys.modules.setdefault("h5py", type(sys)("h5py"))
    from examples.speech_recognition.preprocess_audio import AudioIndexedDatasetBuilder
    from examples.speech_recognition.data.fbank_dataset import FilterBanksDataset
    d = os.path.join(OUT, "tntidx")
    os.makedirs(d, exist_ok=True)
    rs = np.random.RandomState(3)
    out = {}
    fb = [rs.randn(l, 8).astype(np.float32) for l in (5, 1, 9, 3)]
    b = AudioIndexedDatasetBuilder(os.path.join(d, "fbank.bin"))
    for x in fb:
        b.add_item(torch.from_numpy(x.copy()))
    b.finalize(os.path.join(d, "fbank.idx"))
    ds = FilterBanksDataset(os.path.join(d, "fbank"), cached=False)
    for i, x in enumerate(fb):
        assert np.array_equal(ds[i].numpy(), x)
        out["fbank_%d" % i] = x
    out["fbank_sizes"] = np.array([ds.size(i) for i in range(len(ds))], np.int64)
    tk = [rs.randint(4, 90, size=l).astype(np.int64) for l in (7, 2, 11)]
    b = IndexedDatasetBuilder(os.path.join(d, "tokens.bin"))           # int32, stored +1 (Lua indexing)
    for x in tk:
        b.add_item(torch.from_numpy(x.copy()))
    b.finalize(os.path.join(d, "tokens.idx"))
    ds = IndexedDataset(os.path.join(d, "tokens"), fix_lua_indexing=True)
    for i, x in enumerate(tk):
        assert np.array_equal(ds[i].numpy(), x)
        out["tokens_%d" % i] = x
    # ---- batch_by_size (cython module built under /tmp, never inside the reference tree)
    tmp = tempfile.mkdtemp()
    shutil.copy(REF + "/fairseq/data/data_utils_fast.pyx", tmp)
    subprocess.check_call([sys.executable, "-m", "cython", "-3", os.path.join(tmp, "data_utils_fast.pyx")])
    inc = sysconfig.get_paths()["include"]
    so = os.path.join(tmp, "data_utils_fast" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-I", inc, "-I", np.get_include(), os.path.join(tmp, "data_utils_fast.c"), "-o", so])
    spec = importlib.util.spec_from_file_location("data_utils_fast", so)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    cases = []
    for ci, (n, dist, mt, ms, mult) in enumerate([(40, "lognormal", 3000, -1, 1), (40, "lognormal", 3000, 6, 1), (64, "lognormal", 12000, -1, 8),
                                                  (33, "uniform", 2500, 5, 4), (10, "const", 1000, -1, 1), (50, "lognormal", -1, 7, 1),
                                                  (0, "const", 100, -1, 1)]):
        if dist == "lognormal":
            lens = np.clip(np.exp(rs.normal(np.log(600), 0.7, size=n)), 50, 2000).astype(np.int64)
        elif dist == "uniform":
            lens = rs.randint(20, 900, size=n).astype(np.int64)
        else:
            lens = np.full(n, 250, np.int64)
        order = rs.permutation(n).astype(np.int64)
        batches = mod.batch_by_size_fast(order, lambda i: int(lens[i]), mt, ms, mult)
        flat = np.array([i for bt in batches for i in bt], np.int64)
        offs = np.cumsum([0] + [len(bt) for bt in batches]).astype(np.int64)
        out["bbs%d_lens" % ci] = lens; out["bbs%d_order" % ci] = order
        out["bbs%d_params" % ci] = np.array([mt, ms, mult], np.int64)
        out["bbs%d_flat" % ci] = flat; out["bbs%d_offs" % ci] = offs
        cases.append((n, len(batches)))
    out["bbs_ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "data.npz"), **out)
    print("data", cases, sorted(os.listdir(d)))


def _load_fast_batcher():
    import subprocess, sysconfig, tempfile, importlib.util, shutil
    tmp = tempfile.mkdtemp()
    shutil.copy(REF + "/fairseq/data/data_utils_fast.pyx", tmp)
    subprocess.check_call([sys.executable, "-m", "cython", "-3", os.path.join(tmp, "data_utils_fast.pyx")])
    so = os.path.join(tmp, "data_utils_fast" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-w", "-I", sysconfig.get_paths()["include"], "-I", np.get_include(),
                           os.path.join(tmp, "data_utils_fast.c"), "-o", so])
    spec = importlib.util.spec_from_file_location("fairseq.data.data_utils_fast", so)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    sys.modules["fairseq.data.data_utils_fast"] = mod
    return mod


def run_iterator_case():
    """G13: the reference's data pipeline end to end on a tiny on-disk split (tests/golden/s2t_data/, written here with the reference's
    builders): SpeechTranslationCTCTask.load_dataset + get_batch_iterator (filter_by_size, batch_by_size, epoch shuffle, 2 shards) ->
    the collated batches of epochs 1 and 2."""
    from fairseq.data.indexed_dataset import IndexedDatasetBuilder
    sys.modules.setdefault("h5py", type(sys)("h5py"))
    from examples.speech_recognition.preprocess_audio import AudioIndexedDatasetBuilder
    _load_fast_batcher()
    d = os.path.join(OUT, "s2t_data")
    os.makedirs(d, exist_ok=True)
    rs = np.random.RandomState(11)
    tgt, src = mk_dict(40), mk_dict(30)
    tgt.save(os.path.join(d, "dict.de.txt")); src.save(os.path.join(d, "dict.en.txt"))
    n = 14
    lens = [int(v) for v in rs.randint(18, 64, size=n)]
    lens[5] = 90                                            # one utterance beyond --max-source-positions 80: filtered out
    fb = AudioIndexedDatasetBuilder(os.path.join(d, "train.npz.bin"))
    tb = IndexedDatasetBuilder(os.path.join(d, "train.de.bin")); sb = IndexedDatasetBuilder(os.path.join(d, "train.en.bin"))
    for l in lens:
        fb.add_item(torch.from_numpy((rs.randn(l, 80) * 2 + 1).astype(np.float32)))
        tb.add_item(torch.from_numpy(np.concatenate([rs.randint(4, len(tgt), size=rs.randint(2, 7)), [2]]).astype(np.int64)))
        sb.add_item(torch.from_numpy(np.concatenate([rs.randint(4, len(src), size=rs.randint(2, 6)), [2]]).astype(np.int64)))
    fb.finalize(os.path.join(d, "train.npz.idx")); tb.finalize(os.path.join(d, "train.de.idx")); sb.finalize(os.path.join(d, "train.en.idx"))
    a = [d, "--user-dir", REF + "/examples/speech_recognition", "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de",
         "--arch", "conv_transformer", "--no-attn-2d", "--criterion", "ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy",
         "--max-tokens", "150", "--max-source-positions", "80", "--max-target-positions", "50", "--skip-invalid-size-inputs-valid-test", "--cpu"]
    args = options.parse_args_and_arch(options.get_training_parser(), input_args=a)
    task = SpeechTranslationCTCTask.setup_task(args)
    task.load_dataset("train")
    out = {"lens": np.array(lens, np.int64)}
    for shard in (0, 1):
        task.dataset_to_epoch_iter = {}
        it = task.get_batch_iterator(task.dataset("train"), max_tokens=150, max_sentences=None, max_positions=(80, 50),
                                     ignore_invalid_inputs=True, required_batch_size_multiple=1, seed=1, num_shards=2, shard_id=shard,
                                     num_workers=0, epoch=1)
        for ep in (1, 2):
            itr = it.next_epoch_itr(shuffle=True)
            k = 0
            for batch in itr:
                pre = "s%d_e%d_b%d_" % (shard, ep, k); k += 1
                if len(batch) == 0:
                    out[pre + "empty"] = np.int64(1)
                    continue
                out[pre + "id"] = batch["id"].numpy()
                for kk in ("src_tokens", "src_lengths", "prev_output_tokens", "transcript_prev_output_tokens"):
                    out[pre + kk] = batch["net_input"][kk].numpy()
                for kk in ("target", "target_lengths", "transcript_target", "transcript_target_lengths"):
                    out[pre + kk] = batch[kk].numpy()
                out[pre + "ntokens"] = np.int64(batch["ntokens"])
            out["s%d_e%d_n" % (shard, ep)] = np.int64(k)
    np.savez_compressed(os.path.join(OUT, "iterator.npz"), **out)
    print("iterator", {k: int(v) for k, v in out.items() if k.endswith("_n")}, sorted(os.listdir(d)))


def run_distpen_case():
    """G14: --distance-penalty log.  The reference's LocalAttention is hard-wired to CUDA (`.cuda()` on a fresh CPU
    tensor, local_attention.py:132) and scales q in place on a chunk view (:98), which autograd rejects: the forward is
    captured under no_grad with Tensor.cuda patched to the identity (our patch, outside the tree); gradients are pinned through the oracle."""
    D, H, Ff, EL, DL, seed = 64, 2, 128, 2, 1, 800
    args, task, model, crit, V_src, V_tgt = build("dp", D, H, Ff, EL, DL, 2, True, extra=("--distance-penalty", "log"))
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=2, distance_penalty="log")
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    sd = model.state_dict()
    for k in sd:
        if k in W:
            sd[k] = W[k].clone()
        elif k.endswith("self_attn.in_proj_weight") or k.endswith("self_attn.in_proj_bias"):
            base, kind = k.rsplit("in_proj_", 1)
            sd[k] = torch.cat([W[base + n + "_proj." + kind] for n in ("q", "k", "v")], 0)
    model.load_state_dict(sd, strict=True)
    s = make_sample(seed + 1, [70, 57, 41], [5, 4, 6], [4, 3, 5], V_src, V_tgt, V_src - 1)
    sample = to_ref_sample(s)
    out = {("in_" + k): v for k, v in s.items() if isinstance(v, np.ndarray)}
    out["in_ntokens"] = np.int64(s["ntokens"])
    out["meta"] = np.array([D, H, Ff, EL, DL, 2, 1, V_src, V_tgt, V_src - 1, seed], np.int64)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            for mode in ("train", "eval"):
                model.train(mode == "train"); crit.train(mode == "train")
                load = model.state_dict()
                loss, ss, log = crit(model, sample)
                eo = model.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]) if mode == "eval" else None
                out[mode + "_loss"] = np.float64(loss.item()); out[mode + "_sample_size"] = np.int64(ss)
                for k in ("ctc_loss", "nll_loss"):
                    out[mode + "_" + k] = np.float64(float(log[k]))
                if eo is not None:
                    out["eval_encoder_out"] = eo.encoder_out.numpy(); out["eval_src_lengths_out"] = eo.src_lengths.numpy()
                model.load_state_dict(sd, strict=True)              # restore the BN running statistics
    finally:
        torch.Tensor.cuda = real_cuda
    out["statedict_keys"] = np.array(sorted(k for k in sd if "layers.0.self_attn" in k and k.startswith("encoder")))
    np.savez_compressed(os.path.join(OUT, "distpen.npz"), **out)
    print("distpen", {k: float(v) for k, v in out.items() if k.endswith("loss")}, list(out["statedict_keys"]))


ATTN2D_GRAD_KEYS = ["encoder.attn_2d.0.in_proj_weight", "encoder.attn_2d.0.in_proj_bias", "encoder.attn_2d.0.out_proj.weight",
                    "encoder.attn_2d.0.bn_q.weight", "encoder.attn_2d.0.bn_k.bias", "encoder.attn_2d.0.bn_v.weight",
                    "encoder.attn_2d.0.bn_out.bias", "encoder.attn_2d.1.in_proj_weight", "encoder.attn_2d.1.out_proj.bias",
                    "encoder.convolutions.1.bias", "encoder.bn.1.weight", "encoder.fc3.bias"]


def run_attn2d_case():
    """G17: the default conv_transformer front end WITH the two residual ConvAttention2D blocks
    (conv_transformer.py:155-157,216-222; conv_attention_2d.py).  Train-mode loss + gradients, encoder tensors, BN statistics."""
    D, H, Ff, EL, DL, seed = 64, 2, 128, 2, 1, 900
    lens, tgt_lens, tr_lens = [70, 57, 41], [5, 4, 6], [4, 3, 5]
    # conv_attention_2d.py:82 scales q in place on a chunk() view, which current autograd rejects (same class of problem as
    # SURVEY F7): the projections are cloned first (our patch, outside the tree; values unchanged)
    from examples.speech_recognition.modules.conv_attention_2d import ConvAttention2D
    ConvAttention2D.in_proj_qkv = lambda self, query: tuple(t.clone() for t in self._in_proj(query).chunk(3, dim=1))

    def fresh():
        args, task, model, crit, V_src, V_tgt = build("a2d", D, H, Ff, EL, DL, 2, True, attn_2d=True)
        cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=2, attn_2d=True)
        W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
        load_weights(model, crit, W)
        return model, crit, V_src, V_tgt

    model, crit, V_src, V_tgt = fresh()
    assert hasattr(model.encoder, "attn_2d")
    s = make_sample(seed + 1, lens, tgt_lens, tr_lens, V_src, V_tgt, V_src - 1)
    sample = to_ref_sample(s)
    out = {("in_" + k): v for k, v in s.items() if isinstance(v, np.ndarray)}
    out["in_ntokens"] = np.int64(s["ntokens"])
    out["meta"] = np.array([D, H, Ff, EL, DL, 2, 1, V_src, V_tgt, V_src - 1, seed], np.int64)
    model.train(); crit.train()
    trace = {}
    hooks = [model.encoder.attn_2d[i].register_forward_hook(lambda m, i_, o, i=i: trace.__setitem__("attn2d%d" % i, o[0].detach()))
             for i in range(2)]
    model.zero_grad(); crit.zero_grad()
    loss, ss, log = crit(model, sample)
    loss.backward()
    for h in hooks:
        h.remove()
    out["train_loss"] = np.float64(loss.item()); out["train_sample_size"] = np.int64(ss)
    for k, v in log.items():
        out["train_log_" + k] = np.float64(float(v))
    for k, v in trace.items():
        out["train_" + k] = v.numpy()                      # the block's output BEFORE the residual add, [B,C,T4,F4]
    gn = {}
    for k, p_ in list(model.named_parameters()) + [("criterion." + k, p_) for k, p_ in crit.named_parameters()]:
        g = p_.grad if p_.grad is not None else torch.zeros_like(p_)
        gn[k] = float(g.norm())
        if k in ATTN2D_GRAD_KEYS:
            out["grad_" + k] = g.numpy().copy()
    out["gradnorm_keys"] = np.array(sorted(gn)); out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)], np.float64)
    sd = model.state_dict()
    for k in sd:
        if "attn_2d" in k and ("running" in k):
            out["train_stat_" + k] = sd[k].numpy().copy()
    model2, crit2, _, _ = fresh()
    model2.train()
    with torch.no_grad():
        eo = model2.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
    out["train_encoder_out"] = eo.encoder_out.numpy(); out["train_src_lengths_out"] = eo.src_lengths.numpy()
    model3, crit3, _, _ = fresh()
    model3.eval(); crit3.eval()
    with torch.no_grad():
        eo = model3.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
        l3, ss3, log3 = crit3(model3, sample)
    out["eval_encoder_out"] = eo.encoder_out.numpy(); out["eval_src_lengths_out"] = eo.src_lengths.numpy()
    out["eval_loss"] = np.float64(l3.item())
    out["statedict_keys"] = np.array(sorted(k for k in sd if "attn_2d.0" in k))
    np.savez_compressed(os.path.join(OUT, "attn2d.npz"), **out)
    print("attn2d loss", loss.item(), "eval", l3.item(), list(out["statedict_keys"]))


def run_augment_case():
    """G15: TimeStretch then SpecAugment as SpeechRecognitionTask.train_step applies them (speech_recognition.py:254-258),
    with Python's `random` and numpy's global RNG seeded: expected batches after each stage."""
    import random
    from examples.speech_recognition.modules.specaugment import SpecAugment
    from examples.speech_recognition.modules.time_stretch import TimeStretch
    out = {}
    rs = np.random.RandomState(5)
    for ci, (lens, F_, sa, ts) in enumerate([([50, 43, 31], 80, dict(frequency_masking_pars=13, time_masking_pars=13, frequency_masking_num=2, time_masking_num=2, rate=1.0), None),
                                             ([60, 60], 40, dict(frequency_masking_pars=27, time_masking_pars=100, frequency_masking_num=1, time_masking_num=1, rate=0.5), None),
                                             ([50, 43, 31, 8], 80, dict(frequency_masking_pars=13, time_masking_pars=20, frequency_masking_num=2, time_masking_num=1, rate=0.8),
                                              dict(rate=0.7, w=5, low=0.8, high=1.25)),
                                             ([37, 22], 16, None, dict(rate=1.0, w=1, low=0.5, high=1.5))]):
        B, T = len(lens), max(lens)
        x = np.zeros((B, T, F_), np.float32)
        for b, l in enumerate(lens):
            x[b, :l] = rs.randn(l, F_).astype(np.float32) + 3.0          # non-zero everywhere: masks are visible
        batch = {"net_input": {"src_tokens": torch.from_numpy(x.copy()), "src_lengths": torch.tensor(lens)}}
        random.seed(100 + ci); np.random.seed(200 + ci)
        if ts is not None:
            batch = TimeStretch(ts["rate"], ts["w"], ts["low"], ts["high"])(batch)
            out["c%d_ts_tokens" % ci] = batch["net_input"]["src_tokens"].numpy().copy()
            out["c%d_ts_lengths" % ci] = batch["net_input"]["src_lengths"].numpy().copy()
        if sa is not None:
            batch = SpecAugment(**sa)(batch)
        out["c%d_in" % ci] = x; out["c%d_lens" % ci] = np.array(lens, np.int64)
        out["c%d_out" % ci] = batch["net_input"]["src_tokens"].numpy().copy()
        out["c%d_out_lengths" % ci] = batch["net_input"]["src_lengths"].numpy().copy()
        out["c%d_sa" % ci] = np.array([sa[k] for k in ("frequency_masking_pars", "time_masking_pars", "frequency_masking_num", "time_masking_num", "rate")] if sa else [-1] * 5, np.float64)
        out["c%d_ts" % ci] = np.array([ts[k] for k in ("rate", "w", "low", "high")] if ts else [-1] * 4, np.float64)
        print("augment", ci, out["c%d_out" % ci].shape, float((out["c%d_out" % ci] == 0).mean()))
    out["ncases"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)


def run_teacher_case():
    """G16 (KD producer/consumer): scripts/generate_topk.py's core (teacher forward on the target-forced batch, top-k of the
    logits at the non-pad positions, TeacherOutputDataset.save_bin) on the on-disk split of G13, and the batches DatasetWithTeacherOutput
    collates from the written files (fairseq/data/knowledge_distillation.py)."""
    from fairseq.data.knowledge_distillation import TeacherOutputDataset, DatasetWithTeacherOutput
    _load_fast_batcher()
    d = os.path.join(OUT, "s2t_data")
    Kk = 4
    a = [d, "--user-dir", REF + "/examples/speech_recognition", "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de",
         "--arch", "conv_transformer", "--no-attn-2d", "--criterion", "knowledge_distillation", "--input-feat-per-channel", "80",
         "--encoder-embed-dim", "64", "--decoder-embed-dim", "64", "--decoder-output-dim", "64", "--encoder-ffn-embed-dim", "128",
         "--decoder-ffn-embed-dim", "128", "--encoder-attention-heads", "2", "--decoder-attention-heads", "2", "--encoder-layers", "2",
         "--decoder-layers", "1", "--max-tokens", "150", "--max-source-positions", "100", "--max-target-positions", "50", "--cpu",
         "--dropout", "0.0", "--attention-dropout", "0.0", "--relu-dropout", "0.0"]
    args = options.parse_args_and_arch(options.get_training_parser(), input_args=a)
    task = SpeechTranslationCTCTask.setup_task(args)
    task.load_dataset("train")
    torch.manual_seed(1)
    model = task.build_model(args)
    V_src, V_tgt = len(task.source_dictionary), len(task.target_dictionary)
    cfg = s2t_ref.default_cfg(D=64, heads=2, ffn=128, enc_layers=2, dec_layers=1, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt), 900)
    sd = model.state_dict()
    for k in sd:
        if k in W:
            sd[k] = W[k].clone()
    model.load_state_dict(sd, strict=True)
    model.eval()
    ds = task.dataset("train")
    itr = task.get_batch_iterator(ds, max_tokens=150, max_positions=(100, 50), ignore_invalid_inputs=True, required_batch_size_multiple=8,
                                  num_shards=1, shard_id=0).next_epoch_itr(shuffle=False)
    outputs = [None] * len(ds)
    for s in itr:
        if "net_input" not in s:
            continue
        ni = {k: v for k, v in s["net_input"].items() if k != "transcript_prev_output_tokens"}            # F6
        with torch.no_grad():
            net_output = model(**ni)
            vals, idx = torch.topk(net_output[0], Kk, dim=-1)
        keep = s["target"].ne(task.target_dictionary.pad()).numpy().astype(bool)
        for i, id_s in enumerate(s["id"].tolist()):
            outputs[id_s] = [idx.numpy()[i, keep[i]].tolist(), vals.numpy()[i, keep[i]].tolist()]
    prefix = os.path.join(d, "train.en-de.de")
    TeacherOutputDataset.save_bin(prefix + ".top%d_idx" % Kk, [o[0] for o in outputs], np.int32)
    TeacherOutputDataset.save_bin(prefix + ".top%d_out" % Kk, [o[1] for o in outputs], np.float32)
    out = {"meta": np.array([64, 2, 128, 2, 1, 0, 0, V_src, V_tgt, V_src - 1, 900], np.int64), "K": np.int64(Kk)}
    for i, o in enumerate(outputs):
        out["idx_%d" % i] = np.array(o[0], np.int64); out["out_%d" % i] = np.array(o[1], np.float32)
    ti = TeacherOutputDataset(prefix + ".top%d_idx" % Kk, np.int32); to = TeacherOutputDataset(prefix + ".top%d_out" % Kk, np.float32)
    ti.prefetch(range(len(ds))); to.prefetch(range(len(ds)))
    kd = DatasetWithTeacherOutput(ds, to, ti, task.target_dictionary, Kk)
    batch = kd.collater([kd[i] for i in (3, 0, 7, 9)])
    out["batch_id"] = batch["id"].numpy(); out["batch_target"] = batch["target"].numpy()
    out["batch_teacher_idx"] = batch["teacher_output"][0].numpy(); out["batch_teacher_out"] = batch["teacher_output"][1].numpy()
    np.savez_compressed(os.path.join(OUT, "teacher.npz"), **out)
    print("teacher", len(outputs), batch["teacher_output"][0].shape, sorted(f for f in os.listdir(d) if "top" in f))


def run_layerdrop_case():
    """LayerDrop (--encoder-layerdrop / --decoder-layerdrop, conv_transformer.py:172,238-243 and fairseq/modules/layer_drop.py):
    train-mode loss and gradient norms of the reference with torch's CPU generator seeded right before the forward, plus the
    keep / drop decisions those seeds produce (replayed with the same draws the reference makes: one `torch.empty(1).uniform_()`
    per encoder layer, then one vector for the decoder's LayerDropModuleList).  Sub-case `nc`: criterion-owned CTC head on
    encoder_states[0] (the list only holds layers that ran); sub-case `c`: CTC compression after layer 2 (seed chosen so that
    layer 2 runs: the reference dies with UnboundLocalError when that layer is dropped)."""
    D, H, Ff, EL, DL = 64, 2, 128, 6, 3
    pe, pd = 0.4, 0.3
    out = {"rates": np.array([pe, pd])}
    for tag, compress, ctc_layer, seed, fwd_seed in (("nc", False, 1, 400, 11), ("c", True, 2, 500, 31)):
        args, task, model, crit, V_src, V_tgt = build("layerdrop_" + tag, D, H, Ff, EL, DL, ctc_layer, compress,
                                                      extra=["--encoder-layerdrop", str(pe), "--decoder-layerdrop", str(pd)])
        blank = task.source_dictionary.index("<ctc_blank>")
        cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer if compress else 0)
        W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
        load_weights(model, crit, W)
        s = make_sample(seed + 1, [57, 44, 31], [6, 5, 4], [5, 4, 4], V_src, V_tgt, blank)
        sample = to_ref_sample(s)
        torch.manual_seed(fwd_seed)
        enc_keep = [bool(float(torch.empty(1).uniform_()) > pe) for _ in range(EL)]
        dec_keep = [bool(d > pd) for d in torch.empty(DL).uniform_().tolist()]
        assert not all(enc_keep) and any(enc_keep) and not all(dec_keep), (enc_keep, dec_keep)
        assert not compress or enc_keep[ctc_layer - 1]
        model.train(); crit.train()
        model.zero_grad(); crit.zero_grad()
        torch.manual_seed(fwd_seed)
        loss, sample_size, log = crit(model, sample)
        loss.backward()
        gn = {}
        for k, p in list(model.named_parameters()) + [("criterion." + k, p) for k, p in crit.named_parameters()]:
            gn[k] = float(p.grad.norm()) if p.grad is not None else 0.0
        for k, v in s.items():
            if isinstance(v, np.ndarray):
                out["%s_in_%s" % (tag, k)] = v
        out[tag + "_in_ntokens"] = np.int64(s["ntokens"])
        out[tag + "_meta"] = np.array([D, H, Ff, EL, DL, ctc_layer, int(compress), V_src, V_tgt, blank, seed, fwd_seed], np.int64)
        out[tag + "_enc_keep"] = np.array(enc_keep); out[tag + "_dec_keep"] = np.array(dec_keep)
        out[tag + "_loss"] = np.float64(loss.item()); out[tag + "_sample_size"] = np.int64(sample_size)
        for k, v in log.items():
            out["%s_log_%s" % (tag, k)] = np.float64(float(v))
        out[tag + "_gradnorm_keys"] = np.array(sorted(gn))
        out[tag + "_gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)], np.float64)
        print("layerdrop", tag, "keep", enc_keep, dec_keep, "loss", loss.item())
    np.savez_compressed(os.path.join(OUT, "layerdrop.npz"), **out)


def run_layerdrop_opt_case():
    """Optimizer trajectory under LayerDrop (round 5): fairseq's Adam (fairseq/optim/adam.py:147-202) skips a parameter whose
    gradient is None -- a layer LayerDrop removed from this update (fairseq_optimizer.py:97-101 sets every gradient to None before
    the update): no moment decay, no weight decay, its own `step` does not advance.  Five updates of the `nc` LayerDrop model
    (criterion-owned CTC head: no layer has to run), torch's CPU generator seeded before every forward; recorded: the keep / drop
    decisions, losses, gradient norms, every parameter of the dropped-at-least-once layers plus a few others after the last update,
    and the per-parameter step counts of the optimizer state."""
    D, H, Ff, EL, DL = 64, 2, 128, 6, 3
    pe, pd = 0.4, 0.3
    steps, seed, ctc_layer = 5, 400, 1
    args, task, model, crit, V_src, V_tgt = build("layerdrop_nc", D, H, Ff, EL, DL, ctc_layer, False,
                                                  extra=["--encoder-layerdrop", str(pe), "--decoder-layerdrop", str(pd)])
    blank = task.source_dictionary.index("<ctc_blank>")
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    load_weights(model, crit, W)
    s = make_sample(seed + 1, [57, 44, 31], [6, 5, 4], [5, 4, 4], V_src, V_tgt, blank)
    sample = to_ref_sample(s)
    model.train(); crit.train()
    named = list(model.named_parameters()) + [("criterion." + k, p) for k, p in crit.named_parameters()]
    params = [p for _, p in named if p.requires_grad]
    opt = Adam(params, lr=5e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-2)
    out = {"rates": np.array([pe, pd]), "meta": np.array([D, H, Ff, EL, DL, ctc_layer, 0, V_src, V_tgt, blank, seed, steps], np.int64)}
    for k, v in s.items():
        if isinstance(v, np.ndarray):
            out["in_" + k] = v
    out["in_ntokens"] = np.int64(s["ntokens"])
    losses, gnorms, enc_keeps, dec_keeps, seeds = [], [], [], [], []
    for it in range(steps):
        fwd_seed = 1000 + 37 * it
        torch.manual_seed(fwd_seed)
        enc_keeps.append([bool(float(torch.empty(1).uniform_()) > pe) for _ in range(EL)])
        dec_keeps.append([bool(d > pd) for d in torch.empty(DL).uniform_().tolist()])
        opt.zero_grad()
        for p in params:
            p.grad = None                                   # FairseqOptimizer.zero_grad
        torch.manual_seed(fwd_seed)
        loss, ss, _ = crit(model, sample)
        loss.backward()
        for p in params:
            if p.grad is not None:
                p.grad.data.mul_(1.0 / float(ss))
        gnorm = utils.clip_grad_norm_(params, 0.5)
        opt.step()
        losses.append(float(loss)); gnorms.append(float(gnorm)); seeds.append(fwd_seed)
    assert not all(all(k) for k in enc_keeps) and not all(all(k) for k in dec_keeps)
    out["losses"] = np.array(losses); out["gnorms"] = np.array(gnorms); out["fwd_seeds"] = np.array(seeds, np.int64)
    out["enc_keep"] = np.array(enc_keeps); out["dec_keep"] = np.array(dec_keeps)
    sd = dict(model.state_dict()); sd.update({"criterion." + k: v for k, v in crit.state_dict().items()})
    keep_keys = [k for k, _ in named if (".layers." in k and (k.endswith("fc2.weight") or k.endswith("fc1.bias") or k.endswith("self_attn.k_proj.weight")
                                                                or k.endswith("self_attn_layer_norm.weight") or k.endswith("encoder_attn.v_proj.weight")))
                 or k in ("encoder.layer_norm.weight", "decoder.embed_tokens.weight", "encoder.fc3.bias")]
    for k in keep_keys:
        out["param_" + k] = sd[k].numpy().copy()
    st = opt.state_dict()["state"]
    out["step_keys"] = np.array([k for k, p in named if p.requires_grad])
    out["step_vals"] = np.array([int(st[i]["step"]) if i in st else 0 for i in range(len(params))], np.int64)
    print("layerdrop_opt", "enc", enc_keeps, "dec", dec_keeps, "losses", losses, "steps", sorted(set(out["step_vals"].tolist())))
    np.savez_compressed(os.path.join(OUT, "layerdrop_opt.npz"), **out)


def run_attn_case():
    """Round 5: the decoder's returned attention (fairseq/models/transformer.py:756-782: head-averaged encoder-attention weights of
    `alignment_layer`, default the last layer) on the model_a / model_b inputs in eval mode, and the `attention` the reference's
    SequenceGenerator attaches to its hypotheses (sequence_generator.py:286-292,510-560) on generate.npz case a."""
    from fairseq.sequence_generator import SequenceGenerator
    out = {}
    for name, (D, H, Ff, EL, DL, ctc_layer, compress, lens, tgt_lens, tr_lens, seed) in {
            "model_a": (64, 2, 128, 3, 2, 2, True, [61, 50, 37], [7, 5, 6], [6, 4, 5], 100),
            "model_b": (128, 2, 256, 2, 1, 1, True, [45, 45], [6, 6], [5, 5], 200)}.items():
        args, task, model, crit, V_src, V_tgt = build(name, D, H, Ff, EL, DL, ctc_layer, compress)
        blank = task.source_dictionary.index("<ctc_blank>")
        cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer if compress else 0)
        W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
        load_weights(model, crit, W)
        sample = to_ref_sample(make_sample(seed + 1, lens, tgt_lens, tr_lens, V_src, V_tgt, blank))
        model.eval()
        with torch.no_grad():
            eo = model.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
            _, extra = model.decoder(sample["net_input"]["prev_output_tokens"], encoder_out=eo)
            _, extra0 = model.decoder(sample["net_input"]["prev_output_tokens"], encoder_out=eo, alignment_layer=0, alignment_heads=1)
        out[name + "_attn_last"] = extra["attn"][0].float().numpy()
        out[name + "_attn_l0h1"] = extra0["attn"][0].float().numpy()
        print(name, "attn", out[name + "_attn_last"].shape, float(out[name + "_attn_last"].sum(-1).mean()))
    # the generator's attention (generate.npz case a: same build, seed and sample as run_generate_case)
    m = dict(D=64, H=2, Ff=128, EL=3, DL=2, ctc_layer=2, compress=True, seed=600, lens=[61, 50, 37])
    g = dict(beam_size=5, max_len_a=0, max_len_b=12, min_len=1)
    args, task, model, criterion, V_src, V_tgt = build("gena", m["D"], m["H"], m["Ff"], m["EL"], m["DL"], m["ctc_layer"], m["compress"],
                                                       criterion=("ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy"))
    cfg = s2t_ref.default_cfg(D=m["D"], heads=m["H"], ffn=m["Ff"], enc_layers=m["EL"], dec_layers=m["DL"], ctc_layer=m["ctc_layer"])
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), m["seed"])
    W["decoder.output_projection.weight"][2] *= 4.0
    load_weights(model, criterion, W)
    sample = to_ref_sample(make_sample(m["seed"] + 1, m["lens"], [4] * 3, [3] * 3, V_src, V_tgt, V_src - 1))
    model.eval()
    hyps = SequenceGenerator([model], task.target_dictionary, **g).generate([model], sample)
    for b, hs in enumerate(hyps):
        for i, h in enumerate(hs[:2]):
            out["gen_a_attn_%d_%d" % (b, i)] = h["attention"].float().numpy()          # src_len x tgt_len
            out["gen_a_tokens_%d_%d" % (b, i)] = h["tokens"].numpy()
    np.savez_compressed(os.path.join(OUT, "attn.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "attn":
        run_attn_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "generate_ext":
        run_generate_ext_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "layerdrop":
        run_layerdrop_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "layerdrop_opt":
        run_layerdrop_opt_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "twophase":
        run_twophase_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attn2d":
        run_attn2d_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "teacher":
        run_teacher_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "augment":
        run_augment_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "distpen":
        run_distpen_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "iterator":
        run_iterator_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "data":
        run_data_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "generate":
        run_generate_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lne":
        run_lne_case(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        run_generate_case(wide=True); run_twophase_case(wide=True); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        run_kd_case(); run_dual_case(); sys.exit(0)
    run_ctc_cases()
    run_collate()
    run_uer()
    # A: ragged lengths (padding mask present), CTC compression after layer 2, d_head 32
    run_model_case("model_a", 64, 2, 128, 3, 2, 2, True, [61, 50, 37], [7, 5, 6], [6, 4, 5], seed=100, opt_steps=3)
    # B: no padding anywhere (mask None branches), d_head 64, compression after layer 1
    run_model_case("model_b", 128, 2, 256, 2, 1, 1, True, [45, 45], [6, 6], [5, 5], seed=200)
    # C: no compression: criterion-owned fc_out on encoder_states[k-1]
    run_model_case("model_c", 64, 2, 128, 2, 1, 2, False, [40, 29, 33], [4, 6, 3], [3, 5, 4], seed=300)
