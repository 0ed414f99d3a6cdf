"""Shared helpers for the test-suite: golden loading and sample reconstruction."""
import os

import numpy as np
import torch

from oracle import s2t_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def model_case(name):
    """Rebuild cfg, deterministic weights and the sample dict of a model_* fixture (`lne`: the same with layernorm_embedding)."""
    g = load_golden(name)
    D, H, Ff, EL, DL, ctc_layer, compress, V_src, V_tgt, blank, seed = [int(v) for v in g["meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL,
                              ctc_layer=ctc_layer if compress else 0, layernorm_embedding=(name == "lne"))
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    t = lambda k: torch.from_numpy(g["in_" + k])
    sample = dict(
        id=t("id"), ntokens=int(g["in_ntokens"]), nsentences=int(g["in_src_lengths"].shape[0]),
        net_input=dict(src_tokens=t("src_tokens"), src_lengths=t("src_lengths"),
                       prev_output_tokens=t("prev_output_tokens")),
        target=t("target"), target_lengths=t("target_lengths"),
        transcript_target=t("transcript_target"), transcript_target_lengths=t("transcript_target_lengths"),
        ctc_encoder_layer=ctc_layer)
    return g, cfg, W, sample, dict(V_src=V_src, V_tgt=V_tgt, blank=blank, seed=seed,
                                   ctc_layer=ctc_layer, compress=bool(compress))


def generate_case(tag):
    """cfg, weights, inputs, generator options and expected hypotheses of fixture generate.npz (case `a` or `b`)."""
    g = load_golden("generate" if tag in ("a", "b") else "generate_wide")      # c, d, e: 64-wide heads, D 256 (make_golden.py wide)
    D, H, Ff, EL, DL, ctc_layer, compress, V_src, V_tgt, blank, seed = [int(v) for v in g[tag + "_meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer if compress else 0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=bool(compress)), seed)
    W["decoder.output_projection.weight"][2] *= 4.0                 # as make_golden.run_generate_case
    beam, la, lb, mn, lenpen, unkpen, temp = [float(v) for v in g[tag + "_gen"]]
    opts = dict(beam_size=int(beam), max_len_a=la, max_len_b=int(lb), min_len=int(mn), len_penalty=lenpen, unk_penalty=unkpen,
                temperature=temp)
    exp = []
    for b in range(g[tag + "_tokens"].shape[0]):
        hs = []
        for i in range(int(g[tag + "_nhyp"][b])):
            t = g[tag + "_tokens"][b, i]
            n = int((t >= 0).sum())
            hs.append((t[:n], float(g[tag + "_scores"][b, i]), g[tag + "_pos_scores"][b, i, :n]))
        exp.append(hs)
    meta = dict(V_src=V_src, V_tgt=V_tgt, blank=blank, seed=seed, ctc_layer=ctc_layer, compress=bool(compress))
    return cfg, W, torch.from_numpy(g[tag + "_src_tokens"]), torch.from_numpy(g[tag + "_src_lengths"]), opts, exp, meta


def twophase_case(tag):
    """fixture twophase.npz (dual-decoder model, TwoPhaseSequenceGenerator): as generate_case, hypotheses carry their transcript"""
    g = load_golden("twophase" if tag in ("a", "b") else "twophase_wide")
    D, H, Ff, EL, DL, _, _, V_src, V_tgt, blank, seed = [int(v) for v in g[tag + "_meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, V_aux=V_src), seed)
    W["decoder.output_projection.weight"][2] *= 4.0                 # as make_golden.run_twophase_case
    W["auxiliary_decoder.output_projection.weight"][2] *= 4.0
    beam, la, lb, mn, lenpen, unkpen, temp = [float(v) for v in g[tag + "_gen"]]
    opts = dict(beam_size=int(beam), max_len_a=la, max_len_b=int(lb), min_len=int(mn), len_penalty=lenpen, unk_penalty=unkpen,
                temperature=temp)
    exp = []
    for b in range(g[tag + "_tokens"].shape[0]):
        hs = []
        for i in range(int(g[tag + "_nhyp"][b])):
            t, a = g[tag + "_tokens"][b, i], g[tag + "_aux_tokens"][b, i]
            n, na = int((t >= 0).sum()), int((a >= 0).sum())
            hs.append((t[:n], float(g[tag + "_scores"][b, i]), g[tag + "_pos_scores"][b, i, :n], a[:na]))
        exp.append(hs)
    meta = dict(V_src=V_src, V_tgt=V_tgt, blank=blank, seed=seed, D=D, H=H, Ff=Ff, EL=EL, DL=DL)
    return cfg, W, torch.from_numpy(g[tag + "_src_tokens"]), torch.from_numpy(g[tag + "_src_lengths"]), opts, exp, meta


def layerdrop_case(tag):
    """fixture layerdrop.npz, sub-case `nc` (criterion-owned CTC head on encoder_states[0]) or `c` (compression after layer 2):
    cfg, weights, sample, the keep / drop decisions of the reference's seeded forward and its loss / logging / gradient norms"""
    g = load_golden("layerdrop")
    D, H, Ff, EL, DL, ctc_layer, compress, V_src, V_tgt, blank, seed, fwd_seed = [int(v) for v in g[tag + "_meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer if compress else 0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    t = lambda k: torch.from_numpy(g["%s_in_%s" % (tag, k)])
    sample = dict(id=t("id"), ntokens=int(g[tag + "_in_ntokens"]), nsentences=int(g[tag + "_in_src_lengths"].shape[0]),
                  net_input=dict(src_tokens=t("src_tokens"), src_lengths=t("src_lengths"), prev_output_tokens=t("prev_output_tokens")),
                  target=t("target"), target_lengths=t("target_lengths"), transcript_target=t("transcript_target"),
                  transcript_target_lengths=t("transcript_target_lengths"), ctc_encoder_layer=ctc_layer)
    meta = dict(V_src=V_src, V_tgt=V_tgt, blank=blank, seed=seed, fwd_seed=fwd_seed, ctc_layer=ctc_layer, compress=bool(compress),
                enc_keep=[bool(v) for v in g[tag + "_enc_keep"]], dec_keep=[bool(v) for v in g[tag + "_dec_keep"]],
                rates=[float(v) for v in g["rates"]])
    return g, cfg, W, sample, meta


def generate_ext_case(tag):
    """fixture generate_ext.npz: `e` ensemble of two models, `p` prefix tokens, `n` n-gram blocking.  Returns cfg, the list of weight
    dicts (one per ensemble member), inputs, generator options (incl. prefix_tokens / no_repeat_ngram_size) and expected hypotheses"""
    g = load_golden("generate_ext")
    meta = [int(v) for v in g[tag + "_meta"]]
    D, H, Ff, EL, DL, ctc_layer, compress, V_src, V_tgt, blank = meta[:10]
    seeds = meta[10:]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer if compress else 0)
    Ws = []
    for seed in seeds:
        W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=bool(compress)), seed)
        W["decoder.output_projection.weight"][2] *= 4.0
        Ws.append(W)
    beam, la, lb, mn, ngram = [float(v) for v in g[tag + "_gen"]]
    opts = dict(beam_size=int(beam), max_len_a=la, max_len_b=int(lb), min_len=int(mn), no_repeat_ngram_size=int(ngram))
    prefix = torch.from_numpy(g["prefix_tokens"]) if tag == "p" else None
    exp = []
    for b in range(g[tag + "_tokens"].shape[0]):
        hs = []
        for i in range(int(g[tag + "_nhyp"][b])):
            t = g[tag + "_tokens"][b, i]
            n = int((t >= 0).sum())
            hs.append((t[:n], float(g[tag + "_scores"][b, i]), g[tag + "_pos_scores"][b, i, :n]))
        exp.append(hs)
    meta = dict(V_src=V_src, V_tgt=V_tgt, blank=blank, ctc_layer=ctc_layer, compress=bool(compress), seeds=seeds)
    return cfg, Ws, torch.from_numpy(g[tag + "_src_tokens"]), torch.from_numpy(g[tag + "_src_lengths"]), opts, prefix, exp, meta
