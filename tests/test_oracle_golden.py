"""Pin the CPU oracle (oracle/) against golden vectors captured from the real reference
(tests/golden/make_golden.py).  fp32 tolerance 1e-4 (north_star); integer paths bit-exact."""
import numpy as np
import pytest
import torch

from oracle import int_ref, s2t_ref
from helpers import layerdrop_case, load_golden, model_case

TOL = 1e-4


def close(a, b, tol=TOL, what=""):
    if torch.is_tensor(a):
        a = a.detach().numpy()
    if torch.is_tensor(b):
        b = b.detach().numpy()
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


@pytest.mark.parametrize("name", ["model_a", "model_b", "model_c"])
def test_encoder_decoder_train_mode(name):
    g, cfg, W, sample, meta = model_case(name)
    ni = sample["net_input"]
    trace = {}
    with torch.no_grad():
        enc, stats = s2t_ref.encoder_forward(W, cfg, ni["src_tokens"], ni["src_lengths"], training=True, trace=trace)
        logits = s2t_ref.decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
    close(trace["conv0"], g["train_conv0"], what="conv0")
    close(trace["conv1"], g["train_conv1"], what="conv1")
    close(enc.encoder_out, g["train_encoder_out"], what="encoder_out")
    assert np.array_equal(enc.src_lengths.numpy(), g["train_src_lengths_out"])
    gm = g["train_pad_mask"]
    if gm.size == 0:
        assert enc.encoder_padding_mask is None
    else:
        assert np.array_equal(enc.encoder_padding_mask.numpy(), gm)
    if meta["compress"]:
        close(enc.ctc_out, g["train_ctc_out"], what="ctc_out")
    close(logits, g["train_logits"], what="logits")
    for i in range(2):
        close(stats["encoder.bn.%d.running_mean" % i], g["train_bn%d_running_mean" % i], what="rm")
        close(stats["encoder.bn.%d.running_var" % i], g["train_bn%d_running_var" % i], what="rv")


def test_layernorm_embedding_case():
    """`layernorm_embedding` (conv_transformer.py:184-187,230-231; transformer.py:731-732): eval outputs, train loss and every
    gradient norm of the reference (fixture lne.npz)"""
    g, cfg, W, sample, meta = model_case("lne")
    ni = sample["net_input"]
    with torch.no_grad():
        enc, _ = s2t_ref.encoder_forward(W, cfg, ni["src_tokens"], ni["src_lengths"], training=False)
        logits = s2t_ref.decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
    close(enc.encoder_out, g["eval_encoder_out"], what="encoder_out")
    close(logits, g["eval_logits"], what="logits")
    Wg = {k: v.clone().requires_grad_("running" not in k) for k, v in W.items()}
    loss, ss, _, _, _, _ = s2t_ref.ctc_multi_loss(Wg, cfg, sample, 0.1, 1.0, meta["blank"], training=True)
    loss.backward()
    assert abs(float(loss) - float(g["train_loss"])) <= 1e-4 * float(g["train_loss"]) and ss == int(g["train_sample_size"])
    ref = dict(zip([str(k) for k in g["gradnorm_keys"]], g["gradnorm_vals"]))
    for k in ("encoder.layernorm_embedding.weight", "encoder.layernorm_embedding.bias", "decoder.layernorm_embedding.weight",
              "decoder.layernorm_embedding.bias", "encoder.fc3.weight", "decoder.embed_tokens.weight"):
        assert abs(float(Wg[k].grad.norm()) - ref[k]) <= 5e-4 * max(1.0, ref[k]), k


@pytest.mark.parametrize("name", ["model_a", "model_b", "model_c"])
def test_eval_mode(name):
    g, cfg, W, sample, meta = model_case(name)
    ni = sample["net_input"]
    with torch.no_grad():
        enc, _ = s2t_ref.encoder_forward(W, cfg, ni["src_tokens"], ni["src_lengths"], training=False)
        logits = s2t_ref.decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
    close(enc.encoder_out, g["eval_encoder_out"], what="encoder_out")
    assert np.array_equal(enc.src_lengths.numpy(), g["eval_src_lengths_out"])
    close(logits, g["eval_logits"], what="logits")


@pytest.mark.parametrize("name", ["model_a", "model_b", "model_c"])
def test_ctc_multi_loss_and_grads(name):
    g, cfg, W, sample, meta = model_case(name)
    Wg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in W.items()}
    loss, ss, log, enc, logits, _ = s2t_ref.ctc_multi_loss(Wg, cfg, sample, 0.1, 1.0, meta["blank"], training=True)
    loss.backward()
    close(float(loss), float(g["train_loss"]), what="loss")
    assert ss == int(g["train_sample_size"])
    for k in ("ctc_loss", "nll_loss", "ntokens", "nsentences", "sample_size", "ctc_errors", "ctc_total", "nframes"):
        close(log[k], float(g["train_log_" + k]), what=k)
    keys = [str(k) for k in g["gradnorm_keys"]]
    for k, ref in zip(keys, g["gradnorm_vals"]):
        gk = Wg[k].grad if k in Wg else None
        mine = float(gk.norm()) if gk is not None else 0.0
        assert abs(mine - ref) <= 2e-4 * max(1.0, ref), (k, mine, ref)
    for k in g:
        if k.startswith("grad_"):
            close(Wg[k[5:]].grad, g[k], tol=2e-4, what=k)


def test_optimizer_steps():
    g, cfg, W, sample, meta = model_case("model_a")
    P = {k: v.clone() for k, v in W.items()}
    names = [k for k in P if "running" not in k]
    m = {k: torch.zeros_like(P[k]) for k in names}
    v = {k: torch.zeros_like(P[k]) for k in names}
    for it in range(len(g["opt_gnorms"])):
        Wg = {k: (P[k].clone().requires_grad_(True) if k in names else P[k].clone()) for k in P}
        loss, ss, log, enc, logits, stats = s2t_ref.ctc_multi_loss(Wg, cfg, sample, 0.1, 1.0, meta["blank"], training=True)
        loss.backward()
        close(float(loss), g["opt_losses"][it], tol=2e-4, what="loss%d" % it)
        grads = [(Wg[k].grad if Wg[k].grad is not None else torch.zeros_like(P[k])) / float(ss) for k in names]
        gn, grads = s2t_ref.clip_grad_norm(grads, 0.5)
        close(float(gn), g["opt_gnorms"][it], tol=2e-4, what="gnorm%d" % it)
        for k, gk in zip(names, grads):
            P[k], m[k], v[k] = s2t_ref.adam_step(P[k], gk, m[k], v[k], it + 1, 5e-4, wd=1e-4)
        P.update(stats)
    for k in g:
        if k.startswith("opt_param_"):
            close(P[k[10:]], g[k], tol=2e-4, what=k)


def test_ctc_compression_cases_bit_exact():
    g = load_golden("ctc_compress")
    x = torch.from_numpy(g["x"]); lens = torch.from_numpy(g["lens"])
    D = x.shape[-1]
    W = {"encoder.ctc_fc.weight": torch.eye(D), "encoder.ctc_fc.bias": torch.zeros(D)}
    for strat in ("avg", "weighted", "softmax"):
        cfg = s2t_ref.default_cfg(D=D, strategy=strat)
        xt = x.clone().requires_grad_(True)
        x_ctc, out, new_len, pred = s2t_ref.ctc_compress(W, cfg, xt, lens)
        assert np.array_equal(new_len.numpy(), g[strat + "_new_len"])            # bit-exact ints
        for b in range(x.shape[1]):
            L = int(lens[b])
            assert np.array_equal(pred[b, :L].numpy(), g["pred"][b, :L])
        close(out, g[strat + "_out"], what=strat)
        gx, = torch.autograd.grad(out.pow(2).sum(), xt)
        close(gx, g[strat + "_grad_x"], what=strat + " grad")
    # numpy and C restatements of the RLE agree with each other
    pred = g["pred"].astype(np.int32)
    runs = int_ref.ctc_rle_np(pred, g["lens"])
    c = int_ref.ctc_rle_c(pred, g["lens"])
    assert [len(r) for r in runs] == list(c["new_len"])
    for b, r in enumerate(runs):
        assert [t for t, _ in r] == list(c["run_tok"][b, :len(r)])
        assert [n for _, n in r] == list(c["run_len"][b, :len(r)])
    assert np.array_equal(int_ref.argmax_first_c(g["x"].transpose(1, 0, 2)), int_ref.argmax_first_np(g["x"].transpose(1, 0, 2)))


def test_ctc_uer_known_answer():
    g = load_golden("ctc_uer")
    pred = int_ref.argmax_first_np(g["lp"])
    e1, n1 = int_ref.ctc_uer_np(pred, g["in_len"], g["tgt"], g["tgt_len"], int(g["blank"]))
    e2, n2 = int_ref.ctc_uer_c(pred, g["in_len"], g["tgt"], g["tgt_len"], int(g["blank"]))
    assert (e1, n1) == (float(g["errors"]), float(g["total"]))
    assert (e2, n2) == (float(g["errors"]), float(g["total"]))


def test_ctc_loss_matches_torch():
    """The oracle's own alpha/beta recursion against torch's F.ctc_loss (the reference's call site)."""
    torch.manual_seed(0)
    T, B, V, L = 15, 4, 9, 6
    logits = torch.randn(T, B, V, requires_grad=True)
    tgt = torch.randint(0, V - 1, (B, L)); tl = torch.tensor([6, 3, 1, 5]); il = torch.tensor([15, 12, 7, 4])
    tgt[3, :5] = torch.tensor([1, 1, 1, 1, 1])          # needs 9 frames > 4 available -> infinite -> zeroed
    mine = s2t_ref.ctc_loss_sum(logits, tgt, il, tl, V - 1)
    gm, = torch.autograd.grad(mine, logits)
    l2 = logits.detach().clone().requires_grad_(True)
    flat = torch.cat([tgt[b, : tl[b]] for b in range(B)])
    ref = torch.nn.functional.ctc_loss(torch.log_softmax(l2, -1), flat, il, tl, blank=V - 1, reduction="sum", zero_infinity=True)
    gr, = torch.autograd.grad(ref, l2)
    close(float(mine), float(ref), what="ctc")
    close(gm, gr, what="ctc grad")


def _extra_case(name, V_aux=False):
    g = load_golden(name)
    D, H, Ff, EL, DL, _, _, V_src, V_tgt, blank, seed = [int(v) for v in g["meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, V_aux=V_src if V_aux else 0), seed)
    t = lambda k: torch.from_numpy(g["in_" + k])
    sample = dict(ntokens=int(g["in_ntokens"]), net_input=dict(src_tokens=t("src_tokens"), src_lengths=t("src_lengths"),
                  prev_output_tokens=t("prev_output_tokens")), target=t("target"), transcript_target=t("transcript_target"),
                  transcript_target_lengths=t("transcript_target_lengths"))
    return g, cfg, W, sample


def _check_gradnorms(g, Wg):
    for k, ref in zip([str(k) for k in g["gradnorm_keys"]], g["gradnorm_vals"]):
        mine = float(Wg[k].grad.norm()) if (k in Wg and Wg[k].grad is not None) else 0.0
        assert abs(mine - ref) <= 2e-4 * max(1.0, ref), (k, mine, ref)


def test_knowledge_distillation_loss():
    g, cfg, W, sample = _extra_case("kd")
    Wg = {k: v.clone().requires_grad_("running" not in k) for k, v in W.items()}
    ni = sample["net_input"]
    enc, _ = s2t_ref.encoder_forward(Wg, cfg, ni["src_tokens"], ni["src_lengths"], training=True)
    logits = s2t_ref.decoder_forward(Wg, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
    loss = s2t_ref.kd_loss(logits, sample["target"], torch.from_numpy(g["teacher_idx"]), torch.from_numpy(g["teacher_logits"]), 0.6, 2.0, 1)
    loss.backward()
    close(float(loss), float(g["loss"]), what="kd loss")
    _check_gradnorms(g, Wg)


def test_dual_decoder_loss():
    g, cfg, W, sample = _extra_case("dual", V_aux=True)
    sample["net_input"]["transcript_prev_output_tokens"] = torch.from_numpy(g["in_transcript_prev_output_tokens"])
    Wg = {k: v.clone().requires_grad_("running" not in k) for k, v in W.items()}
    loss, log, lg, la = s2t_ref.dual_decoder_loss(Wg, cfg, sample, 0.1, training=True)
    loss.backward()
    close(float(loss), float(g["loss"]), what="dual loss")
    for k in ("primary_loss", "auxiliary_loss", "primary_nll_loss", "auxiliary_nll_loss"):
        close(log[k], float(g["log_" + k]), what=k)
    close(lg, g["logits"], what="logits"); close(la, g["aux_logits"], what="aux logits")
    _check_gradnorms(g, Wg)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_beam_search_matches_reference_generator(tag):
    """G9: hypotheses (tokens exact, scores 1e-4) of the reference SequenceGenerator, fairseq/sequence_generator.py."""
    from helpers import generate_case
    cfg, W, src, lens, o, exp, _ = generate_case(tag)
    got = s2t_ref.beam_search(W, cfg, src, lens, o["beam_size"], o["max_len_a"], o["max_len_b"], o["min_len"], o["len_penalty"],
                              o["unk_penalty"], o["temperature"])
    assert len(got) == len(exp)
    for hs, es in zip(got, exp):
        assert len(hs) == len(es)
        for (t, s, ps), (et, es_, eps) in zip(hs, es):
            assert t.tolist() == et.tolist()
            assert abs(s - es_) < 1e-4
            np.testing.assert_allclose(ps.numpy(), eps, atol=1e-4)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_two_phase_beam_search_matches_reference_generator(tag):
    """G18: TwoPhaseSequenceGenerator on the dual-decoder model: target tokens and the transcript each hypothesis descends from
    exact, scores 1e-4 (examples/speech_recognition/twophase_sequence_generator.py)."""
    from helpers import twophase_case
    cfg, W, src, lens, o, exp, _ = twophase_case(tag)
    got = s2t_ref.two_phase_beam_search(W, cfg, src, lens, o["beam_size"], o["max_len_a"], o["max_len_b"], o["min_len"],
                                        o["len_penalty"], o["unk_penalty"], o["temperature"])
    assert len(got) == len(exp)
    for hs, es in zip(got, exp):
        assert len(hs) == len(es)
        for (t, s, ps, a), (et, es_, eps, ea) in zip(hs, es):
            assert t.tolist() == et.tolist() and a.tolist() == ea.tolist()
            assert abs(s - es_) < 1e-4
            np.testing.assert_allclose(ps.numpy(), eps, atol=1e-4)


def _distpen_case():
    g = load_golden("distpen")
    D, H, Ff, EL, DL, ctc_layer, compress, V_src, V_tgt, blank, seed = [int(v) for v in g["meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer, distance_penalty="log")
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    t = lambda k: torch.from_numpy(g["in_" + k])
    sample = dict(id=t("id"), ntokens=int(g["in_ntokens"]), nsentences=3,
                  net_input=dict(src_tokens=t("src_tokens"), src_lengths=t("src_lengths"), prev_output_tokens=t("prev_output_tokens")),
                  target=t("target"), target_lengths=t("target_lengths"), transcript_target=t("transcript_target"),
                  transcript_target_lengths=t("transcript_target_lengths"), ctc_encoder_layer=ctc_layer)
    return g, cfg, W, sample, blank


def test_distance_penalty_matches_reference_local_attention():
    """G14: --distance-penalty log (LocalAttention + LogPenalty): oracle vs the reference's forward, train and eval mode."""
    g, cfg, W, sample, blank = _distpen_case()
    for mode in ("train", "eval"):
        Wm = {k: v.clone() for k, v in W.items()}
        loss, ss, log, enc, _, _ = s2t_ref.ctc_multi_loss(Wm, cfg, sample, 0.1, 1.0, blank, training=(mode == "train"))
        assert abs(float(loss) - float(g[mode + "_loss"])) < 1e-4 * abs(float(g[mode + "_loss"]))
        assert abs(log["ctc_loss"] - float(g[mode + "_ctc_loss"])) < 1e-4 * abs(float(g[mode + "_ctc_loss"]))
        assert abs(log["nll_loss"] - float(g[mode + "_nll_loss"])) < 1e-4 * abs(float(g[mode + "_nll_loss"]))
        if mode == "eval":
            np.testing.assert_allclose(enc.encoder_out.numpy(), g["eval_encoder_out"], atol=1e-4)
            assert enc.src_lengths.tolist() == g["eval_src_lengths_out"].tolist()
    # the penalty does something
    cfg0 = dict(cfg, distance_penalty=False)
    l0 = s2t_ref.ctc_multi_loss({k: v.clone() for k, v in W.items()}, cfg0, sample, 0.1, 1.0, blank, training=False)[0]
    assert abs(float(l0) - float(g["eval_loss"])) > 1e-3


def _attn2d_case():
    g = load_golden("attn2d")
    D, H, Ff, EL, DL, ctc_layer, compress, V_src, V_tgt, blank, seed = [int(v) for v in g["meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=ctc_layer, attn_2d=True)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    t = lambda k: torch.from_numpy(g["in_" + k])
    sample = dict(id=t("id"), ntokens=int(g["in_ntokens"]), nsentences=3,
                  net_input=dict(src_tokens=t("src_tokens"), src_lengths=t("src_lengths"), prev_output_tokens=t("prev_output_tokens")),
                  target=t("target"), target_lengths=t("target_lengths"), transcript_target=t("transcript_target"),
                  transcript_target_lengths=t("transcript_target_lengths"), ctc_encoder_layer=ctc_layer)
    return g, cfg, W, sample, blank


def test_conv_attention_2d_matches_reference():
    """G17: the front end with the two residual ConvAttention2D blocks: loss, every gradient norm, selected
    gradients, block outputs, BatchNorm running statistics and the encoder output (train and eval) against the reference."""
    g, cfg, W, sample, blank = _attn2d_case()
    assert [str(k) for k in g["statedict_keys"] if "num_batches" not in str(k)] == \
        sorted(k for k in W if k.startswith("encoder.attn_2d.0."))
    Wg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in W.items()}
    loss, ss, log, enc, logits, stats = s2t_ref.ctc_multi_loss(Wg, cfg, sample, 0.1, 1.0, blank, training=True)
    loss.backward()
    close(float(loss), float(g["train_loss"]), what="loss")
    assert ss == int(g["train_sample_size"])
    for k in ("ctc_loss", "nll_loss"):
        close(log[k], float(g["train_log_" + k]), what=k)
    _check_gradnorms(g, Wg)
    for k in g:
        if k.startswith("grad_"):
            close(Wg[k[5:]].grad, g[k], tol=2e-4, what=k)
        if k.startswith("train_stat_"):
            close(stats[k[len("train_stat_"):]], g[k], what=k)
    close(enc.encoder_out, g["train_encoder_out"], what="train encoder_out")
    trace = {}
    with torch.no_grad():
        ni = sample["net_input"]
        s2t_ref.subsample({k: v.clone() for k, v in W.items()}, cfg, ni["src_tokens"], ni["src_lengths"], training=True, trace=trace)
    # golden holds the block output before the residual add; the trace holds x + block(x)
    close(trace["attn2d0"] - trace["conv1"], g["train_attn2d0"], what="block 0")
    close(trace["attn2d1"] - trace["attn2d0"], g["train_attn2d1"], what="block 1")
    with torch.no_grad():
        l3, _, _, enc3, _, _ = s2t_ref.ctc_multi_loss({k: v.clone() for k, v in W.items()}, cfg, sample, 0.1, 1.0, blank, training=False)
    close(float(l3), float(g["eval_loss"]), what="eval loss")
    close(enc3.encoder_out, g["eval_encoder_out"], what="eval encoder_out")
    assert enc3.src_lengths.tolist() == g["eval_src_lengths_out"].tolist()


@pytest.mark.parametrize("tag", ["nc", "c"])
def test_layerdrop_matches_reference(tag):
    """--encoder-layerdrop / --decoder-layerdrop: the oracle with the reference's keep / drop decisions reproduces the reference's
    seeded train-mode forward + backward (dropped layers: no contribution, zero gradient, no encoder_states entry)"""
    g, cfg, W, sample, meta = layerdrop_case(tag)
    cfg = dict(cfg, enc_keep=meta["enc_keep"], dec_keep=meta["dec_keep"])
    Wg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in W.items()}
    loss, ss, log, enc, logits, _ = s2t_ref.ctc_multi_loss(Wg, cfg, sample, 0.1, 1.0, meta["blank"], training=True)
    loss.backward()
    close(float(loss.detach()), float(g[tag + "_loss"]), what="loss")
    assert ss == int(g[tag + "_sample_size"])
    for k in ("ctc_loss", "nll_loss", "ctc_errors", "ctc_total"):
        close(log[k], float(g["%s_log_%s" % (tag, k)]), what=k)
    for k, ref in zip([str(k) for k in g[tag + "_gradnorm_keys"]], g[tag + "_gradnorm_vals"]):
        gk = Wg[k].grad if k in Wg else None
        mine = float(gk.norm()) if gk is not None else 0.0
        assert abs(mine - ref) <= 2e-4 * max(1.0, ref), (k, mine, ref)
    dropped = [l for l, k in enumerate(meta["enc_keep"]) if not k]
    assert dropped and all(float(Wg["encoder.layers.%d.fc1.weight" % l].grad.norm() if Wg["encoder.layers.%d.fc1.weight" % l].grad is not None
                                 else 0.0) == 0.0 for l in dropped)


@pytest.mark.parametrize("tag", ["e", "p", "n"])
def test_beam_search_ensemble_prefix_ngram_match_reference_generator(tag):
    """ensembles (EnsembleModel.forward_decoder), prefix tokens (_prefix_tokens) and n-gram blocking (_no_repeat_ngram) of the
    reference's SequenceGenerator: tokens exact, scores 1e-4"""
    from helpers import generate_ext_case
    cfg, Ws, src, lens, o, prefix, exp, _ = generate_ext_case(tag)
    got = s2t_ref.beam_search(Ws, cfg, src, lens, o["beam_size"], o["max_len_a"], o["max_len_b"], o["min_len"],
                              prefix_tokens=prefix, no_repeat_ngram_size=o["no_repeat_ngram_size"])
    assert len(got) == len(exp)
    for hs, es in zip(got, exp):
        assert len(hs) == len(es)
        for (t, s, ps), (et, es_, eps) in zip(hs, es):
            assert t.tolist() == et.tolist()
            assert abs(s - es_) < 1e-4
            np.testing.assert_allclose(ps.numpy(), eps, atol=1e-4)


@pytest.mark.parametrize("name", ["model_a", "model_b"])
def test_decoder_attention_return_matches_reference(name):
    """fairseq/models/transformer.py:756-782: the head-averaged encoder-attention weights the decoder returns (default: last layer, all
    heads; alignment_layer = 0 with alignment_heads = 1), eval mode, against the fixture captured from the real reference (attn.npz)"""
    from helpers import load_golden, model_case
    g, cfg, W, sample, meta = model_case(name)
    ga = load_golden("attn")
    ni = sample["net_input"]
    with torch.no_grad():
        enc, _ = s2t_ref.encoder_forward(W, cfg, ni["src_tokens"], ni["src_lengths"], training=False)
        _, a_last = s2t_ref.decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask,
                                            attn_layer=cfg["dec_layers"] - 1)
        _, a_0 = s2t_ref.decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask, attn_layer=0, attn_heads=1)
    for mine, key in ((a_last, "_attn_last"), (a_0, "_attn_l0h1")):
        ref = ga[name + key]
        assert tuple(mine.shape) == ref.shape
        assert float((mine.float() - torch.from_numpy(ref)).abs().max()) < 1e-5, key
