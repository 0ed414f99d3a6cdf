"""End-to-end parity on the GPU: the registered `conv_transformer` + `ctc_multi_loss` (HIP engine, fp32 mode)
against the golden vectors captured from the real reference and against the CPU oracle.  Tolerance 1e-4
relative (north_star) on activations and losses, 5e-4 on gradients (longer f32 reduction chains);
CTC-compression lengths are compared bit-exactly."""
import numpy as np
import pytest
import torch

from helpers import layerdrop_case, model_case
from oracle import s2t_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(name, dtype=torch.float32):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401  (registers)
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    g, cfg, W, sample, meta = model_case(name)
    args = namespace(arch="conv_transformer", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                     layernorm_embedding=bool(cfg.get("layernorm_embedding")),
                     label_smoothing=0.1, ctc_compress_out=meta["compress"], ctc_encoder_layer=meta["ctc_layer"], ctc_weight=1.0,
                     encoder_embed_dim=cfg["D"], encoder_ffn_embed_dim=cfg["ffn"], encoder_attention_heads=cfg["heads"],
                     encoder_layers=cfg["enc_layers"], decoder_layers=cfg["dec_layers"], no_attn_2d=True,
                     decoder_embed_dim=cfg["D"], decoder_ffn_embed_dim=cfg["ffn"], decoder_attention_heads=cfg["heads"],
                     input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False)
    tgt, src = Dictionary.synthetic(96), Dictionary.synthetic(59)
    src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, src)
    model = task.build_model(args)
    crit = task.build_criterion(args)
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
    with torch.no_grad():
        crit.ctc_aware_model.fc_out.weight.copy_(W["criterion.ctc_aware_model.fc_out.weight"])
        crit.ctc_aware_model.fc_out.bias.copy_(W["criterion.ctc_aware_model.fc_out.bias"])
    model.hp.sub_dropout = 0.0                      # parity mode: the reference run had dropout patched to identity
    model.materialize(DEV, dtype, extra=crit.arena_params())
    return g, cfg, W, sample, meta, model, crit


def to_dev(s):
    if torch.is_tensor(s):
        return s.to(DEV)
    if isinstance(s, dict):
        return {k: to_dev(v) for k, v in s.items()}
    if isinstance(s, (list, tuple)):
        return type(s)(to_dev(v) for v in s)
    return s


def close(a, b, tol, what):
    a = np.asarray(a.detach().float().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol * max(1.0, np.abs(b).max()), "%s: max err %.3e (scale %.3e)" % (what, err, np.abs(b).max())


@pytest.mark.parametrize("name", ["model_a", "model_b", "model_c"])
def test_forward_matches_reference_golden(name):
    g, cfg, W, sample, meta, model, crit = build(name)
    model.train()
    s = to_dev(sample)
    ni = s["net_input"]
    with torch.no_grad():
        eo = model.encoder(ni["src_tokens"], ni["src_lengths"], return_all_hiddens=True)
        logits, _ = model.decoder(ni["prev_output_tokens"], encoder_out=eo)
    close(eo.encoder_out, g["train_encoder_out"], 1e-4, "encoder_out")
    assert np.array_equal(eo.src_lengths.cpu().numpy(), g["train_src_lengths_out"])             # bit-exact
    gm = g["train_pad_mask"]
    if gm.size == 0:
        assert eo.encoder_padding_mask is None
    else:
        assert np.array_equal(eo.encoder_padding_mask.cpu().numpy(), gm)
    if meta["compress"]:
        close(eo.ctc_out, g["train_ctc_out"], 1e-4, "ctc_out")
    close(logits, g["train_logits"], 1e-4, "logits")
    sd = model.state_dict()
    for i in range(2):
        close(sd["encoder.bn.%d.running_mean" % i], g["train_bn%d_running_mean" % i], 1e-4, "running_mean")
        close(sd["encoder.bn.%d.running_var" % i], g["train_bn%d_running_var" % i], 1e-4, "running_var")
    model.eval()
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})     # restore running stats
    with torch.no_grad():
        eo = model.encoder(ni["src_tokens"], ni["src_lengths"])
        logits, _ = model.decoder(ni["prev_output_tokens"], encoder_out=eo)
    close(eo.encoder_out, g["eval_encoder_out"], 1e-4, "eval encoder_out")
    close(logits, g["eval_logits"], 1e-4, "eval logits")


def test_layernorm_embedding_matches_reference():
    """`layernorm_embedding` on both sides (conv_transformer.py:184-187,230-231; transformer.py:578-581,731-732)
    against the reference's own run (fixture lne.npz): eval outputs 1e-4, train loss 1e-4, every gradient norm 5e-4; the incremental
    decoder applies it too (step logits = teacher-forced logits)."""
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    g, cfg, W, sample, meta, model, crit = build("lne")
    assert "encoder.layernorm_embedding.weight" in model.arena.slices and "decoder.layernorm_embedding.bias" in model.arena.slices
    s = to_dev(sample)
    ni = s["net_input"]
    model.eval()
    with torch.no_grad():
        eo = model.encoder(ni["src_tokens"], ni["src_lengths"])
        logits, _ = model.decoder(ni["prev_output_tokens"], encoder_out=eo)
        close(eo.encoder_out, g["eval_encoder_out"], 1e-4, "eval encoder_out")
        close(logits, g["eval_logits"], 1e-4, "eval logits")
        inc = {}
        for t in range(ni["prev_output_tokens"].shape[1]):
            step, _ = model.decoder(ni["prev_output_tokens"][:, :t + 1], encoder_out=eo, incremental_state=inc)
            real = ni["prev_output_tokens"][:, t].ne(1).cpu().numpy()          # right-padded targets: padded steps are nobody's input
            close(step[:, 0][torch.from_numpy(real).to(step.device)], g["eval_logits"][:, t][real], 1e-4, "step %d" % t)
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, s)
    loss.backward()
    close(loss, g["train_loss"], 1e-4, "loss")
    grads = fused_to_reference({n: model.arena.g(n).detach().cpu().clone() for n in model.arena.slices})
    for k, ref in zip([str(k) for k in g["gradnorm_keys"]], g["gradnorm_vals"]):
        if k in grads:
            assert abs(float(grads[k].norm()) - ref) <= 5e-4 * max(1.0, ref), (k, float(grads[k].norm()), ref)


@pytest.mark.parametrize("name", ["model_a", "model_b", "model_c"])
def test_ctc_multi_loss_and_gradients(name):
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    g, cfg, W, sample, meta, model, crit = build(name)
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()
    close(loss, g["train_loss"], 1e-4, "loss")
    assert ss == int(g["train_sample_size"])
    for k in ("ctc_loss", "nll_loss", "ntokens", "nsentences", "sample_size", "ctc_errors", "ctc_total", "nframes"):
        v = log[k]
        close(v if torch.is_tensor(v) else np.float64(v), g["train_log_" + k], 1e-4, k)
    grads = {n: model.arena.g(n).detach().cpu().clone() for n in model.arena.slices}
    grads = fused_to_reference(grads)
    keys = [str(k) for k in g["gradnorm_keys"]]
    for k, ref in zip(keys, g["gradnorm_vals"]):
        if k not in grads:
            assert ref == 0.0 or k.endswith("_float_tensor"), k
            continue
        mine = float(grads[k].norm())
        assert abs(mine - ref) <= 5e-4 * max(1.0, ref), (k, mine, ref)
    for k in g:
        if k.startswith("grad_"):
            close(grads[k[5:]].reshape(g[k].shape), g[k], 5e-4, k)


def test_optimizer_steps_match_reference():
    from fbk_fairseq_st_amd.optim import ArenaAdam
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    g, cfg, W, sample, meta, model, crit = build("model_a")
    model.train(); crit.train()
    opt = ArenaAdam(model.arena, lr=5e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-4)
    s = to_dev(sample)
    for it in range(len(g["opt_gnorms"])):
        opt.zero_grad()
        loss, ss, _ = crit(model, s)
        opt.backward(loss)
        opt.multiply_grads(1.0 / float(ss))
        gn = opt.clip_grad_norm(0.5)
        opt.step()
        close(loss, g["opt_losses"][it], 2e-4, "loss%d" % it)
        close(gn, g["opt_gnorms"][it], 2e-4, "gnorm%d" % it)
    sd = model.state_dict()
    for k in g:
        if k.startswith("opt_param_"):
            close(sd[k[10:]], g[k], 2e-4, k)


@pytest.mark.parametrize("name", ["model_a", "model_b"])
def test_bf16_mode_close_to_fp32_reference(name):
    """bf16 storage / f32 accumulate path: tolerance 5e-2 relative on the loss, 0.15 on logits (8-bit mantissa)."""
    g, cfg, W, sample, meta, model, crit = build(name, torch.bfloat16)
    model.train(); crit.train()
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()
    assert abs(float(loss) - float(g["train_loss"])) < 5e-2 * float(g["train_loss"])
    gn = float(model.arena.grad.norm())
    ref = float(np.sqrt((g["gradnorm_vals"] ** 2).sum()))
    assert abs(gn - ref) < 0.1 * ref


def test_dropout_training_step_runs_and_is_reproducible():
    g, cfg, W, sample, meta, model, crit = build("model_a")
    model.hp.dropout, model.hp.attention_dropout, model.hp.activation_dropout, model.hp.sub_dropout = 0.2, 0.1, 0.1, None
    model.train(); crit.train()
    s = to_dev(sample)
    outs = []
    for rep in range(2):
        model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
        model.set_seed(123)
        model.arena.zero_grad()
        loss, _, _ = crit(model, s)
        loss.backward()
        outs.append((float(loss), float(model.arena.grad.norm())))
    # same masks in both runs; only the order of f32 atomic partial sums (loss / BN statistics) may differ
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[0][0]) and abs(outs[0][1] - outs[1][1]) <= 1e-4 * outs[0][1]
    assert abs(outs[0][0] - float(g["train_loss"])) > 1e-3          # dropout changes the loss
    assert np.isfinite(outs[0][1])


def test_fused_backward_dropout_gives_the_same_gradients():
    """LayerNorm backward writing the next block's dropout(dx) is a pure fusion: gradients equal the unfused chain's"""
    g, cfg, W, sample, meta, model, crit = build("model_a")
    model.hp.dropout, model.hp.attention_dropout, model.hp.activation_dropout, model.hp.sub_dropout = 0.2, 0.1, 0.1, None
    model.train(); crit.train()
    s = to_dev(sample)
    grads = []
    for fuse in (False, True):
        model.engine.fuse_bwd_dropout = fuse
        model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
        model.set_seed(123)
        model.arena.zero_grad()
        loss, _, _ = crit(model, s)
        loss.backward()
        grads.append(model.arena.grad.clone())
    model.engine.fuse_bwd_dropout = True
    diff = float((grads[0] - grads[1]).norm()); ref = float(grads[0].norm())
    assert diff <= 1e-4 * ref, (diff, ref)           # only the order of f32 atomic partial sums differs


def _build_extra(name, arch, criterion, dual=False, **crit_args):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    from helpers import load_golden
    g = load_golden(name)
    D, H, Ff, EL, DL, _, _, V_src, V_tgt, blank, seed = [int(v) for v in g["meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, V_aux=V_src if dual else 0), seed)
    args = namespace(arch=arch, criterion=criterion, encoder_embed_dim=D, encoder_ffn_embed_dim=Ff, encoder_attention_heads=H,
                     encoder_layers=EL, decoder_layers=DL, decoder_embed_dim=D, decoder_ffn_embed_dim=Ff,
                     decoder_attention_heads=H, no_attn_2d=True, input_feat_per_channel=80, dropout=0.0,
                     attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False, **crit_args)
    tgt, src = Dictionary.synthetic(96), Dictionary.synthetic(59)
    src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, src)
    model, crit = task.build_model(args), task.build_criterion(args)
    model.load_state_dict(W)
    model.hp.sub_dropout = 0.0
    model.materialize(DEV, torch.float32)
    t = lambda k: torch.from_numpy(g["in_" + k]).to(DEV)
    sample = dict(ntokens=int(g["in_ntokens"]), net_input=dict(src_tokens=t("src_tokens"), src_lengths=t("src_lengths"),
                  prev_output_tokens=t("prev_output_tokens")), target=t("target"), transcript_target=t("transcript_target"),
                  transcript_target_lengths=t("transcript_target_lengths"))
    return g, model, crit, sample


def _check_gradnorms_gpu(g, model):
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    grads = fused_to_reference({n: model.arena.g(n).detach().cpu().clone() for n in model.arena.slices})
    for k, ref in zip([str(k) for k in g["gradnorm_keys"]], g["gradnorm_vals"]):
        if k not in grads:
            assert ref == 0.0 or k.endswith("_float_tensor"), k
            continue
        mine = float(grads[k].norm())
        assert abs(mine - ref) <= 5e-4 * max(1.0, ref), (k, mine, ref)


def test_knowledge_distillation_matches_reference():
    g, model, crit, sample = _build_extra("kd", "conv_transformer", "knowledge_distillation", kd_lambda=0.6, kd_temperature=2.0)
    sample["teacher_output"] = [torch.from_numpy(g["teacher_idx"]).to(DEV), torch.from_numpy(g["teacher_logits"]).to(DEV)]
    model.train(); crit.train()
    loss, ss, log = crit(model, sample)
    loss.backward()
    close(loss, g["loss"], 1e-4, "kd loss")
    assert ss == int(g["sample_size"])
    _check_gradnorms_gpu(g, model)


def test_dual_decoder_matches_reference():
    g, model, crit, sample = _build_extra("dual", "conv_transformer_dualdecoder", "cross_entropy_dualdecoder", dual=True,
                                          label_smoothing=0.1)
    sample["net_input"]["transcript_prev_output_tokens"] = torch.from_numpy(g["in_transcript_prev_output_tokens"]).to(DEV)
    model.train(); crit.train()
    loss, ss, log = crit(model, sample)
    loss.backward()
    close(loss, g["loss"], 1e-4, "dual loss")
    for k in ("primary_loss", "auxiliary_loss", "primary_nll_loss", "auxiliary_nll_loss", "auxiliary_ntokens", "sample_size"):
        v = log[k]
        close(v if torch.is_tensor(v) else np.float64(v), g["log_" + k], 1e-4, k)
    with torch.no_grad():
        (lg, _), (la, _) = model(**sample["net_input"])
    close(lg, g["logits"], 1e-4, "logits"); close(la, g["aux_logits"], 1e-4, "aux logits")
    _check_gradnorms_gpu(g, model)


# ---------------------------------------------------------------------------------------------- generation (a22)
def build_gen(tag, dtype=torch.float32):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    from helpers import generate_case
    cfg, W, src, lens, opts, exp, meta = generate_case(tag)
    crit = dict(criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy") if meta["compress"] else \
        dict(criterion="label_smoothed_cross_entropy")
    args = namespace(arch="conv_transformer", label_smoothing=0.1, ctc_compress_out=meta["compress"],
                     ctc_encoder_layer=meta["ctc_layer"], ctc_weight=1.0,
                     encoder_embed_dim=cfg["D"], encoder_ffn_embed_dim=cfg["ffn"], encoder_attention_heads=cfg["heads"],
                     encoder_layers=cfg["enc_layers"], decoder_layers=cfg["dec_layers"], no_attn_2d=True,
                     decoder_embed_dim=cfg["D"], decoder_ffn_embed_dim=cfg["ffn"], decoder_attention_heads=cfg["heads"],
                     input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0,
                     sentence_avg=False, max_target_positions=1000, **crit)
    tgt, sd = Dictionary.synthetic(96), Dictionary.synthetic(59)
    sd.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, sd)
    model = task.build_model(args)
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
    model.materialize(DEV, dtype)
    model.eval()
    return task, model, src.to(DEV), lens.to(DEV), opts, exp, (cfg, W)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_incremental_decoder_matches_full_decoder(tag):
    """transformer.py:690-760: feeding tokens one at a time through the K/V cache gives the teacher-forced logits."""
    task, model, src, lens, opts, exp, _ = build_gen(tag)
    B, L = src.shape[0], 7
    rs = np.random.RandomState(3)
    prev = torch.from_numpy(np.concatenate([np.full((B, 1), 2), rs.randint(4, 96, size=(B, L - 1))], 1)).to(DEV)
    with torch.no_grad():
        eo = model.encoder(src, lens)
        full, _ = model.decoder(prev, encoder_out=eo)
        inc = {}
        for t in range(L):
            step, _ = model.decoder(prev[:, :t + 1], encoder_out=eo, incremental_state=inc)
            close(step[:, 0], full[:, t].cpu().numpy(), 1e-4, "step %d" % t)
        # beam re-ordering of the cache: two hypotheses per sentence with different prefixes, swapped after 3 steps
        dup = torch.arange(B, device=DEV).repeat_interleave(2)
        eo2 = model.encoder.reorder_encoder_out(eo, dup)
        prev2 = prev[dup].clone()
        prev2[1::2, 1:] = torch.from_numpy(rs.randint(4, 96, size=(B, L - 1))).to(DEV)
        full2, _ = model.decoder(prev2, encoder_out=eo2)
        st = model.decoder.begin_incremental(eo2, L)
        for t in range(3):
            model.decoder.step_incremental(st, prev2[:, t])
        swap = torch.arange(2 * B, device=DEV).view(B, 2).flip(1).reshape(-1)
        model.decoder.reorder_incremental(st, swap)
        out = model.decoder.step_incremental(st, prev2[swap, 3])
        close(out, full2[swap, 3].cpu().numpy(), 1e-4, "after reorder")


@pytest.mark.parametrize("tag,route", [("a", "steps"), ("b", "steps"), ("c", "device"), ("d", "device"), ("e", "device"),
                                       ("c", "device-nograph"), ("c", "steps"), ("d", "steps")])
def test_beam_search_matches_reference_generator(tag, route, monkeypatch):
    """G9: the HIP generator reproduces the hypotheses of fairseq's SequenceGenerator (tokens exact, scores 1e-4)
    and agrees with the CPU oracle restatement.  Cases a, b (32-wide heads) can only take the step-by-step search; c, d, e (64-wide
    heads, D 256) take the device-resident search of csrc/decode.hip -- as a replayed hipGraph and as plain launches -- and, with
    S2T_DEVICE_SEARCH=0, the step-by-step one: the same fixture of the reference holds both."""
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    task, model, src, lens, opts, exp, (cfg, W) = build_gen(tag)
    gen = SequenceGenerator([model], task.target_dictionary, **opts)
    if route == "steps":
        monkeypatch.setenv("S2T_DEVICE_SEARCH", "0")
    gen.device_graph = route != "device-nograph"
    hyps = gen.generate([model], dict(net_input=dict(src_tokens=src, src_lengths=lens)))
    assert ("launches_per_step" in gen.last_stats) == route.startswith("device"), "the search took the other route"
    orc = s2t_ref.beam_search(W, cfg, src.cpu(), lens.cpu(), opts["beam_size"], opts["max_len_a"], opts["max_len_b"], opts["min_len"],
                              opts["len_penalty"], opts["unk_penalty"], opts["temperature"])
    assert len(hyps) == len(exp)
    for hs, es, os_ in zip(hyps, exp, orc):
        assert len(hs) == len(es)
        for h, (et, esc, eps), (ot, osc, ops) in zip(hs, es, os_):
            assert h["tokens"].tolist() == et.tolist() == ot.tolist()
            assert abs(float(h["score"]) - esc) < 1e-4 and abs(float(h["score"]) - osc) < 1e-4
            np.testing.assert_allclose(h["positional_scores"].cpu().numpy(), eps, atol=1e-4)


def test_beam_search_bf16_runs_and_is_consistent():
    """bf16 engine: hypotheses are well-formed (end in eos, sorted, positional scores sum to the unnormalised score)."""
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    task, model, src, lens, opts, exp, _ = build_gen("a", torch.bfloat16)
    gen = SequenceGenerator([model], task.target_dictionary, **opts)
    hyps = gen.generate([model], dict(net_input=dict(src_tokens=src, src_lengths=lens)))
    for hs in hyps:
        assert len(hs) == opts["beam_size"]
        sc = [float(h["score"]) for h in hs]
        assert sc == sorted(sc, reverse=True)
        for h in hs:
            assert int(h["tokens"][-1]) == 2 and not bool((h["tokens"][:-1] == 2).any())
            n = h["tokens"].numel()
            assert abs(float(h["positional_scores"].sum()) / n ** opts["len_penalty"] - float(h["score"])) < 1e-3


# ---------------------------------------------------------------------------------------------- BASELINE.json configs
@pytest.mark.parametrize("arch,B,lengths,L,ctc_layer,feat,attn_2d", [("s2t_transformer_xs", 4, [200, 180, 150, 120], 12, 4, 80, False),
                                                                     ("s2t_transformer_s", 3, [333, 333, 333], 10, 8, 80, False),
                                                                     ("s2t_transformer_xs", 2, [150, 97], 8, 4, 40, True),
                                                                     ("s2t_transformer_xs", 1, [61], 6, 2, 80, True),
                                                                     ("s2t_transformer_xs", 2, [9, 5], 3, 2, 80, False),
                                                                     ("s2t_transformer_xs", 3, [13, 4, 1], 2, 2, 80, True)])
def test_baseline_config_shapes_match_oracle(arch, B, lengths, L, ctc_layer, feat, attn_2d):
    """configs[0] of BASELINE.json (s2t_transformer_xs, 80-mel x 200 frames, batch 4, ragged) and an s-preset batch at the real
    vocabulary sizes (V_tgt 8000, V_src 5001: unaligned logit rows, K-tail GEMMs, d_head 64): one fp32 update's loss terms and
    gradient norms of the HIP engine against the CPU oracle run on the same weights and batch (1e-4 loss, 1e-3 gradients)."""
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    from fbk_fairseq_st_amd.registry import apply_arch, namespace, setup_task
    a = namespace(arch=arch, task="dummy_s2t", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                  label_smoothing=0.1, sentence_avg=False, ctc_compress_out=True, ctc_encoder_layer=ctc_layer, ctc_weight=1.0,
                  ctc_compress_strategy="avg", input_feat_per_channel=feat, no_attn_2d=not attn_2d, dict_size=8000 - 4,
                  src_dict_size=5000 - 4, batch_size=B, frames=max(lengths), tgt_len=L, transcript_len=L, dropout=0.0,
                  attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, seed=1)
    apply_arch(a)
    a.dropout = a.attention_dropout = a.activation_dropout = a.relu_dropout = 0.0
    task = setup_task(a)
    torch.manual_seed(3)
    model = task.build_model(a)
    crit = task.build_criterion(a)
    W = {k: v.detach().clone().float() for k, v in model.state_dict().items() if torch.is_tensor(v) and v.dtype.is_floating_point}
    model.hp.sub_dropout = 0.0
    model.materialize(DEV, torch.float32, extra=crit.arena_params())
    sample = task.dummy_batch(seed=7, lengths=lengths)
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()

    hp = model.hp
    cfg = s2t_ref.default_cfg(D=hp.D, heads=hp.heads, ffn=hp.ffn, enc_layers=hp.enc_layers, dec_layers=hp.dec_layers, ctc_layer=ctc_layer,
                              act=hp.act, feat=feat, attn_2d=attn_2d)
    assert hp.attn_2d == attn_2d
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items() if not k.endswith("_float_tensor") and "running" not in k}
    Wb = {k: v.clone() for k, v in W.items() if "running" in k}
    blank = task.source_dictionary.index("<ctc_blank>")
    rl, rss, rlog, _, _, _ = s2t_ref.ctc_multi_loss({**Wr, **Wb}, cfg, dict(sample, ctc_encoder_layer=ctc_layer), 0.1, 1.0, blank, training=True)
    rl.backward()
    close(loss, float(rl), 1e-4, "loss")
    close(log["ctc_loss"], rlog["ctc_loss"], 1e-4, "ctc_loss")
    close(log["nll_loss"], rlog["nll_loss"], 1e-4, "nll_loss")
    assert int(log["ctc_errors"]) == int(rlog["ctc_errors"]) and int(log["ctc_total"]) == int(rlog["ctc_total"])
    assert ss == rss
    grads = fused_to_reference({n: model.arena.g(n).detach().cpu().clone() for n in model.arena.slices})
    worst = 0.0
    for k, p in Wr.items():
        if p.grad is None or k not in grads:
            continue
        ref = float(p.grad.norm()); mine = float(grads[k].norm())
        worst = max(worst, abs(mine - ref) / max(1.0, ref))
        assert abs(mine - ref) <= 1e-3 * max(1.0, ref), (k, mine, ref)
    assert worst < 1e-3


# ---------------------------------------------------------------------------------------------- distance penalty (8-f N4)
def _build_distpen(dtype=torch.float32):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    from test_oracle_golden import _distpen_case
    g, cfg, W, sample, blank = _distpen_case()
    args = namespace(arch="conv_transformer", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                     label_smoothing=0.1, ctc_compress_out=True, ctc_encoder_layer=cfg["ctc_layer"], ctc_weight=1.0,
                     encoder_embed_dim=cfg["D"], encoder_ffn_embed_dim=cfg["ffn"], encoder_attention_heads=cfg["heads"],
                     encoder_layers=cfg["enc_layers"], decoder_layers=cfg["dec_layers"], no_attn_2d=True, distance_penalty="log",
                     decoder_embed_dim=cfg["D"], decoder_ffn_embed_dim=cfg["ffn"], decoder_attention_heads=cfg["heads"],
                     input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False)
    tgt, src = Dictionary.synthetic(96), Dictionary.synthetic(59)
    src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, src)
    model = task.build_model(args)
    crit = task.build_criterion(args)
    # weights arrive under the reference's LocalAttention names (one in_proj parameter per encoder layer)
    sd = {}
    for k, v in W.items():
        if k.startswith("criterion."):
            continue
        if k.startswith("encoder.layers.") and ".self_attn.q_proj." in k:
            base, kind = k.split("q_proj.")
            sd[base + "in_proj_" + kind] = torch.cat([W[base + n + "_proj." + kind] for n in ("q", "k", "v")], 0)
        elif k.startswith("encoder.layers.") and (".self_attn.k_proj." in k or ".self_attn.v_proj." in k):
            continue
        else:
            sd[k] = v
    model.load_state_dict(sd)
    out_keys = set(model.state_dict())
    assert all(str(k) in out_keys for k in g["statedict_keys"]) and "encoder.layers.0.self_attn.q_proj.weight" not in out_keys
    model.hp.sub_dropout = 0.0
    model.materialize(DEV, dtype, extra=crit.arena_params())
    return g, cfg, W, sample, blank, model, crit


def test_distance_penalty_forward_matches_reference_and_gradients_match_oracle():
    """G14 on the GPU: losses against the reference's LocalAttention forward (train + eval), eval encoder output, and -- the reference
    cannot back-propagate through its in-place q scaling -- gradients against the oracle that the same fixture pins."""
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    g, cfg, W, sample, blank, model, crit = _build_distpen()
    s = to_dev(sample)
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, s)
    loss.backward()
    close(loss, g["train_loss"], 1e-4, "train loss")
    close(log["ctc_loss"], g["train_ctc_loss"], 1e-4, "ctc"); close(log["nll_loss"], g["train_nll_loss"], 1e-4, "nll")
    grads = fused_to_reference({n: model.arena.g(n).detach().cpu().clone() for n in model.arena.slices})
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items() if "running" not in k}
    Wb = {k: v.clone() for k, v in W.items() if "running" in k}
    rl = s2t_ref.ctc_multi_loss({**Wr, **Wb}, cfg, sample, 0.1, 1.0, blank, training=True)[0]
    rl.backward()
    for k, p in Wr.items():
        if p.grad is None or k not in grads:
            continue
        ref = float(p.grad.norm()); mine = float(grads[k].norm())
        assert abs(mine - ref) <= 5e-4 * max(1.0, ref), (k, mine, ref)
    close(grads["encoder.layers.0.self_attn.q_proj.weight"], Wr["encoder.layers.0.self_attn.q_proj.weight"].grad, 5e-4, "dq_proj")
    model.eval(); crit.eval()
    model.load_state_dict(model.state_dict())
    with torch.no_grad():
        # running statistics were updated by the train step: restore the fixture's
        model.bn0_running_mean.copy_(W["encoder.bn.0.running_mean"]); model.bn0_running_var.copy_(W["encoder.bn.0.running_var"])
        model.bn1_running_mean.copy_(W["encoder.bn.1.running_mean"]); model.bn1_running_var.copy_(W["encoder.bn.1.running_var"])
        loss, _, log = crit(model, s)
        eo = model.encoder(s["net_input"]["src_tokens"], s["net_input"]["src_lengths"])
    close(loss, g["eval_loss"], 1e-4, "eval loss")
    close(eo.encoder_out, g["eval_encoder_out"], 1e-4, "eval encoder_out")


def test_distance_penalty_long_sequences_bf16_kernels():
    """the second-generation (bf16, T >= 128) kernels with the penalty against the fp32 torch formula"""
    from fbk_fairseq_st_amd import kernels as K
    heads, d, B, T = 2, 64, 2, 200
    D = heads * d
    gq = torch.Generator().manual_seed(3)
    q, k, v, do = [(torch.randn(T, B, D, generator=gq) * 0.7).to(torch.bfloat16) for _ in range(4)]
    klen = torch.tensor([T, 170], dtype=torch.int32)
    qf, kf, vf = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    qh = qf.view(T, B * heads, d).transpose(0, 1) * d ** -0.5
    kh = kf.view(T, B * heads, d).transpose(0, 1); vh = vf.view(T, B * heads, d).transpose(0, 1)
    sc = torch.bmm(qh, kh.transpose(1, 2))
    m = torch.arange(T)[None, :] >= klen[:, None]
    sc = sc.view(B, heads, T, T).masked_fill(m[:, None, None, :], float("-inf")).view(B * heads, T, T)
    dist = (torch.arange(T)[:, None] - torch.arange(T)[None, :]).abs().float()
    sc = sc - torch.max(torch.zeros_like(dist), torch.log(dist))
    ref = torch.bmm(torch.softmax(sc, -1), vh).transpose(0, 1).reshape(T, B, D)
    ref.backward(do.float())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    out, lse = K.attn_fwd(qd, kd, vd, heads, klen=klen.to(DEV), dist_penalty=True)
    err = lambda a, b: float((a.float().cpu() - b).abs().max() / b.abs().max())
    assert err(out, ref.detach()) < 2e-2
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, out, do.to(DEV), lse, heads, dq, dk, dv, klen=klen.to(DEV), dist_penalty=True)
    assert err(dq, qf.grad) < 3e-2 and err(dk, kf.grad) < 3e-2 and err(dv, vf.grad) < 3e-2
    out0, _ = K.attn_fwd(qd, kd, vd, heads, klen=klen.to(DEV))
    assert err(out0, ref.detach()) > 5e-2


# ---------------------------------------------------------------------------------------------- KD teacher dump (8-f N5)
def test_teacher_topk_dump_matches_generate_topk():
    """G16 on the GPU: the teacher dump (target-forced forward + s2t_topk) reproduces scripts/generate_topk.py on the on-disk split:
    columns exact, logits 1e-4; the knowledge-distillation criterion then trains from the collated batch."""
    import os
    from helpers import GOLDEN, load_golden
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.indexed import DatasetWithTeacherOutput, TeacherOutputDataset, dump_teacher_topk
    from fbk_fairseq_st_amd.registry import apply_arch, namespace, setup_task
    g = load_golden("teacher")
    D, H, Ff, EL, DL, _, _, V_src, V_tgt, _, seed = [int(v) for v in g["meta"]]
    a = namespace(arch="conv_transformer", task="speech_translation_with_transcription", data=os.path.join(GOLDEN, "s2t_data"),
                  source_lang="en", target_lang="de", criterion="knowledge_distillation", kd_lambda=0.5, kd_temperature=1.0,
                  label_smoothing=0.0, sentence_avg=False, input_feat_per_channel=80, no_attn_2d=True, encoder_embed_dim=D,
                  decoder_embed_dim=D, encoder_ffn_embed_dim=Ff, decoder_ffn_embed_dim=Ff, encoder_attention_heads=H,
                  decoder_attention_heads=H, encoder_layers=EL, decoder_layers=DL, max_source_positions=100, max_target_positions=50,
                  dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0)
    apply_arch(a)
    task = setup_task(a)
    assert len(task.source_dictionary) == V_src and len(task.target_dictionary) == V_tgt
    task.load_dataset("train")
    model = task.build_model(a)
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt), seed)
    model.load_state_dict(W)
    model.hp.sub_dropout = 0.0
    crit = task.build_criterion(a)
    model.materialize(DEV, torch.float32)
    ds = task.dataset("train")
    K_ = int(g["K"])
    outs = dump_teacher_topk(task, model, ds, K_, max_tokens=150, max_positions=(100, 50))
    for i in range(len(ds)):
        assert outs[i][0] == g["idx_%d" % i].tolist(), i
        np.testing.assert_allclose(np.array(outs[i][1], np.float32), g["out_%d" % i], atol=1e-4 * max(1.0, float(np.abs(g["out_%d" % i]).max())))
    pre = os.path.join(GOLDEN, "s2t_data", "train.en-de.de")
    kd = DatasetWithTeacherOutput(ds, TeacherOutputDataset(pre + ".top%d_out" % K_, np.float32), TeacherOutputDataset(pre + ".top%d_idx" % K_, np.int32),
                                  task.target_dictionary, K_)
    batch = kd.collater([kd[i] for i in (3, 0, 7, 9)])
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, to_dev(batch))
    loss.backward()
    assert np.isfinite(float(loss)) and float(model.arena.grad.abs().sum()) > 0


def test_bf16_long_sequences_match_the_fp32_engine():
    """T4 = 150 >= 128: the bf16 run takes the second-generation attention kernels, the 8-wave / two-slice GEMMs and the one-pass conv2
    weight gradient; same weights and batch through the fp32 engine (first-generation f32 kernels, pinned to the reference by the other
    tests): loss within 2e-2, every gradient norm within 6e-2 relative (bf16 storage, 8-bit mantissa)."""
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.registry import apply_arch, namespace, setup_task
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        a = namespace(arch="s2t_transformer_xs", task="dummy_s2t", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                      label_smoothing=0.1, sentence_avg=False, ctc_compress_out=True, ctc_encoder_layer=4, ctc_weight=1.0,
                      input_feat_per_channel=80, no_attn_2d=True, dict_size=996, src_dict_size=495, batch_size=8, frames=600,
                      tgt_len=20, transcript_len=16, seed=1)
        apply_arch(a)
        a.dropout = a.attention_dropout = a.activation_dropout = a.relu_dropout = 0.0
        task = setup_task(a)
        torch.manual_seed(5)
        model = task.build_model(a); crit = task.build_criterion(a)
        model.hp.sub_dropout = 0.0
        model.materialize(DEV, dtype, extra=crit.arena_params())
        sample = task.dummy_batch(seed=3, lengths=[600, 600, 590, 580, 560, 520, 480, 400])
        model.train(); crit.train(); model.arena.zero_grad()
        loss, ss, log = crit(model, to_dev(sample))
        loss.backward()
        res[dtype] = (float(loss), float(log["ctc_loss"]), {n: float(model.arena.g(n).norm()) for n in model.arena.slices})
    (l32, c32, g32), (l16, c16, g16) = res[torch.float32], res[torch.bfloat16]
    assert abs(l16 - l32) < 2e-2 * abs(l32) and abs(c16 - c32) < 2e-2 * abs(c32), (l16, l32)
    bad = {n: (g16[n], g32[n]) for n in g32 if abs(g16[n] - g32[n]) > 6e-2 * max(g32[n], 1e-3 * max(g32.values()))}
    assert not bad, bad


# ------------------------------------------------------------------ ConvAttention2D front end
def _build_attn2d(dtype=torch.float32):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    from test_oracle_golden import _attn2d_case
    g, cfg, W, sample, blank = _attn2d_case()
    args = namespace(arch="conv_transformer", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                     label_smoothing=0.1, ctc_compress_out=True, ctc_encoder_layer=cfg["ctc_layer"], ctc_weight=1.0,
                     encoder_embed_dim=cfg["D"], encoder_ffn_embed_dim=cfg["ffn"], encoder_attention_heads=cfg["heads"],
                     encoder_layers=cfg["enc_layers"], decoder_layers=cfg["dec_layers"],          # no --no-attn-2d: the default front end
                     decoder_embed_dim=cfg["D"], decoder_ffn_embed_dim=cfg["ffn"], decoder_attention_heads=cfg["heads"],
                     input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False)
    tgt, src = Dictionary.synthetic(96), Dictionary.synthetic(59)
    src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, src)
    model = task.build_model(args)
    crit = task.build_criterion(args)
    assert model.hp.attn_2d
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
    model.hp.sub_dropout = 0.0
    model.materialize(DEV, dtype, extra=crit.arena_params())
    return g, cfg, W, sample, blank, model, crit


def test_conv_attention_2d_front_end_matches_reference():
    """G17 on the GPU: the default front end (two residual ConvAttention2D blocks) against the reference: loss, every gradient norm,
    selected gradients, BatchNorm running statistics of the blocks, encoder output in train and eval mode; state-dict names."""
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    g, cfg, W, sample, blank, model, crit = _build_attn2d()
    assert set(str(k) for k in g["statedict_keys"]) <= set(model.state_dict())
    s = to_dev(sample)
    model.eval(); crit.eval()
    with torch.no_grad():
        loss, _, _ = crit(model, s)
        eo = model.encoder(s["net_input"]["src_tokens"], s["net_input"]["src_lengths"])
    close(loss, g["eval_loss"], 1e-4, "eval loss")
    close(eo.encoder_out, g["eval_encoder_out"], 1e-4, "eval encoder_out")
    assert eo.src_lengths.tolist() == g["eval_src_lengths_out"].tolist()
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, s)
    loss.backward()
    close(loss, g["train_loss"], 1e-4, "train loss")
    close(log["ctc_loss"], g["train_log_ctc_loss"], 1e-4, "ctc"); close(log["nll_loss"], g["train_log_nll_loss"], 1e-4, "nll")
    _check_gradnorms_gpu(g, model)
    grads = fused_to_reference({n: model.arena.g(n).detach().cpu().clone() for n in model.arena.slices})
    for k in g:
        if k.startswith("grad_"):
            close(grads[k[5:]], g[k], 5e-4, k)
    sd = model.state_dict()
    for k in g:
        if k.startswith("train_stat_"):
            close(sd[k[len("train_stat_"):]], g[k], 1e-4, k)


def test_conv_attention_2d_bf16_and_dropout():
    """bf16 engine close to the fp32 reference numbers; with dropout the step runs, is reproducible for a seed and changes the loss"""
    g, cfg, W, sample, blank, model, crit = _build_attn2d(torch.bfloat16)
    s = to_dev(sample)
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, _, _ = crit(model, s)
    loss.backward()
    assert abs(float(loss) - float(g["train_loss"])) < 0.03 * float(g["train_loss"])
    ref = float(np.sqrt((g["gradnorm_vals"] ** 2).sum())); gn = float(model.arena.grad.norm())
    assert abs(gn - ref) < 0.1 * ref
    g, cfg, W, sample, blank, model, crit = _build_attn2d()
    model.hp.dropout = 0.3
    model.train(); crit.train()
    outs = []
    for rep in range(2):
        model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
        model.set_seed(77)
        model.arena.zero_grad()
        loss, _, _ = crit(model, s)
        loss.backward()
        outs.append((float(loss), float(model.arena.grad.norm())))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[0][0]) and abs(outs[0][1] - outs[1][1]) <= 1e-4 * outs[0][1]
    assert abs(outs[0][0] - float(g["train_loss"])) > 1e-3 and np.isfinite(outs[0][1])


# ------------------------------------------------------------------ two-phase generation with the dual-decoder model
def _build_twophase(tag, dtype=torch.float32):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    from helpers import twophase_case
    cfg, W, src, lens, opts, exp, meta = twophase_case(tag)
    args = namespace(arch="conv_transformer_dualdecoder", task="speech_translation_dualdecoding", criterion="cross_entropy_dualdecoder",
                     label_smoothing=0.1, encoder_embed_dim=meta["D"], encoder_ffn_embed_dim=meta["Ff"], encoder_attention_heads=meta["H"],
                     encoder_layers=meta["EL"], decoder_layers=meta["DL"], decoder_embed_dim=meta["D"], decoder_ffn_embed_dim=meta["Ff"],
                     decoder_attention_heads=meta["H"], no_attn_2d=True, input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0,
                     activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False, beam=opts["beam_size"], max_len_a=opts["max_len_a"],
                     max_len_b=opts["max_len_b"], min_len=opts["min_len"], lenpen=opts["len_penalty"], unkpen=opts["unk_penalty"],
                     temperature=opts["temperature"])
    tgt, src_d = Dictionary.synthetic(96), Dictionary.synthetic(59)
    src_d.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationDualDecodingTask(args, tgt, src_d)
    model = task.build_model(args)
    model.load_state_dict(W)
    model.hp.sub_dropout = 0.0
    model.materialize(DEV, dtype)
    model.eval()
    return task, args, model, src.to(DEV), lens.to(DEV), opts, exp, (cfg, W)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_two_phase_generator_matches_reference(tag):
    """G18: transcript beam search with the auxiliary decoder, then the hierarchical target search: target tokens and the transcript of
    every hypothesis exact, scores 1e-4 against examples/speech_recognition/twophase_sequence_generator.py and the oracle"""
    task, args, model, src, lens, opts, exp, (cfg, W) = _build_twophase(tag)
    gen = task.build_generator([model], args)
    assert type(gen).__name__ == "TwoPhaseSequenceGenerator"
    hyps = task.inference_step(gen, [model], dict(net_input=dict(src_tokens=src, src_lengths=lens)))
    orc = s2t_ref.two_phase_beam_search(W, cfg, src.cpu(), lens.cpu(), opts["beam_size"], opts["max_len_a"], opts["max_len_b"],
                                        opts["min_len"], opts["len_penalty"], opts["unk_penalty"], opts["temperature"])
    assert len(hyps) == len(exp)
    for hs, es, os_ in zip(hyps, exp, orc):
        assert len(hs) == len(es)
        for h, (et, esc, eps, ea), (ot, osc, ops, oa) in zip(hs, es, os_):
            assert h["tokens"].tolist() == et.tolist() == ot.tolist()
            assert h["aux_tokens"].tolist() == ea.tolist() == oa.tolist()
            assert abs(float(h["score"]) - esc) < 1e-4 and abs(float(h["score"]) - osc) < 1e-4
            np.testing.assert_allclose(h["positional_scores"].cpu().numpy(), eps, atol=1e-4)


def test_layerdrop_optimizer_trajectory_matches_reference():
    """Five updates under --encoder-layerdrop 0.4 --decoder-layerdrop 0.3 with Adam (lr 5e-3, weight decay 1e-2): the reference's Adam
    skips a parameter whose gradient is None -- a layer no forward of the update ran (fairseq/optim/adam.py:160-165 behind
    fairseq_optimizer.py:97-101): no moment decay, no weight decay, its own bias-correction step.  Seeded like the reference run
    (tests/golden/make_golden.py layerdrop_opt): same decisions, losses and gradient norms 2e-4, parameters of every layer 2e-4,
    per-parameter step counts exact (through the reference's optimizer-state layout)."""
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.optim import ArenaAdam
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    from helpers import load_golden
    from oracle import s2t_ref
    g = load_golden("layerdrop_opt")
    D, H, Ff, EL, DL, ctc_layer, compress, V_src, V_tgt, blank, seed, steps = [int(v) for v in g["meta"]]
    cfg = s2t_ref.default_cfg(D=D, heads=H, ffn=Ff, enc_layers=EL, dec_layers=DL, ctc_layer=0)
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, V_src, V_tgt, criterion_fc=True), seed)
    t = lambda k: torch.from_numpy(g["in_" + k])
    sample = dict(id=t("id"), ntokens=int(g["in_ntokens"]), nsentences=int(g["in_src_lengths"].shape[0]),
                  net_input=dict(src_tokens=t("src_tokens"), src_lengths=t("src_lengths"), prev_output_tokens=t("prev_output_tokens")),
                  target=t("target"), target_lengths=t("target_lengths"), transcript_target=t("transcript_target"),
                  transcript_target_lengths=t("transcript_target_lengths"), ctc_encoder_layer=ctc_layer)
    args = namespace(arch="conv_transformer", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                     label_smoothing=0.1, ctc_compress_out=False, ctc_encoder_layer=ctc_layer, ctc_weight=1.0,
                     encoder_embed_dim=D, encoder_ffn_embed_dim=Ff, encoder_attention_heads=H, encoder_layers=EL, decoder_layers=DL, no_attn_2d=True,
                     decoder_embed_dim=D, decoder_ffn_embed_dim=Ff, decoder_attention_heads=H,
                     input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False,
                     encoder_layerdrop=float(g["rates"][0]), decoder_layerdrop=float(g["rates"][1]))
    tgt, src = Dictionary.synthetic(V_tgt - 4), Dictionary.synthetic(V_src - 5)
    src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, src)
    model, crit = task.build_model(args), task.build_criterion(args)
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
    with torch.no_grad():
        crit.ctc_aware_model.fc_out.weight.copy_(W["criterion.ctc_aware_model.fc_out.weight"])
        crit.ctc_aware_model.fc_out.bias.copy_(W["criterion.ctc_aware_model.fc_out.bias"])
    model.hp.sub_dropout = 0.0
    model.materialize(DEV, torch.float32, extra=crit.arena_params())
    model.train(); crit.train()
    opt = ArenaAdam(model.arena, lr=5e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-2)
    s = to_dev(sample)
    for it in range(steps):
        opt.zero_grad()
        torch.manual_seed(int(g["fwd_seeds"][it]))
        loss, ss, _ = crit(model, s)
        opt.backward(loss)
        opt.multiply_grads(1.0 / float(ss))
        gn = opt.clip_grad_norm(0.5)
        opt.step()
        close(loss, g["losses"][it], 2e-4, "loss%d" % it)
        close(gn, g["gnorms"][it], 2e-4, "gnorm%d" % it)
    sd = dict(model.state_dict())
    for k in g:
        if k.startswith("param_"):
            close(sd[k[6:]], g[k], 2e-4, k)
    # the step counts, as the reference's checkpoint would hold them
    names = [str(k) for k in g["step_keys"]]
    ref_sd = opt.reference_state_dict(names)
    for i, (n, st) in enumerate(zip(names, g["step_vals"])):
        mine = int(ref_sd["state"][i]["step"]) if i in ref_sd["state"] else 0
        assert mine == int(st), (n, mine, int(st))
    # and back: a second optimizer picks the counts up from that layout
    opt2 = ArenaAdam(model.arena, lr=5e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-2)
    opt2.load_reference_state_dict(ref_sd, names)
    assert opt2.group_steps == opt.group_steps and opt2.step_count == opt.step_count


@pytest.mark.parametrize("tag", ["nc", "c"])
def test_layerdrop_matches_reference(tag):
    """--encoder-layerdrop 0.4 --decoder-layerdrop 0.3 (conv_transformer.py:238-243, fairseq/modules/layer_drop.py): seeded like the
    reference run, the model draws the same decisions from torch's CPU generator, the dropped layers vanish from the forward AND
    the backward schedule (zero gradients), `encoder_states` only lists layers that ran; loss / logging 1e-4, gradient norms 5e-4"""
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    g, cfg, W, sample, meta = layerdrop_case(tag)
    args = namespace(arch="conv_transformer", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                     label_smoothing=0.1, ctc_compress_out=meta["compress"], ctc_encoder_layer=meta["ctc_layer"], ctc_weight=1.0,
                     encoder_embed_dim=cfg["D"], encoder_ffn_embed_dim=cfg["ffn"], encoder_attention_heads=cfg["heads"],
                     encoder_layers=cfg["enc_layers"], decoder_layers=cfg["dec_layers"], no_attn_2d=True,
                     decoder_embed_dim=cfg["D"], decoder_ffn_embed_dim=cfg["ffn"], decoder_attention_heads=cfg["heads"],
                     input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False,
                     encoder_layerdrop=meta["rates"][0], decoder_layerdrop=meta["rates"][1])
    tgt, src = Dictionary.synthetic(96), Dictionary.synthetic(59)
    src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, src)
    model, crit = task.build_model(args), task.build_criterion(args)
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
    with torch.no_grad():
        crit.ctc_aware_model.fc_out.weight.copy_(W["criterion.ctc_aware_model.fc_out.weight"])
        crit.ctc_aware_model.fc_out.bias.copy_(W["criterion.ctc_aware_model.fc_out.bias"])
    model.hp.sub_dropout = 0.0
    model.materialize(DEV, torch.float32, extra=crit.arena_params())
    model.train(); crit.train()
    model.arena.zero_grad()
    torch.manual_seed(meta["fwd_seed"])
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()
    torch.cuda.synchronize()
    close(loss, g[tag + "_loss"], 1e-4, "loss")
    assert ss == int(g[tag + "_sample_size"])
    for k in ("ctc_loss", "nll_loss", "ctc_errors", "ctc_total"):
        close(float(log[k]), g["%s_log_%s" % (tag, k)], 1e-4, k)
    mine = fused_to_reference({n: model.arena.g(n).detach().float().cpu() for n in model.arena.slices})
    for k, ref in zip([str(k) for k in g[tag + "_gradnorm_keys"]], g[tag + "_gradnorm_vals"]):
        if k.endswith("_float_tensor"):
            continue
        a = float(mine[k].norm())
        assert abs(a - ref) <= 5e-4 * max(1.0, ref), (k, a, ref)
    for l, kept in enumerate(meta["enc_keep"]):
        if not kept:
            assert float(mine["encoder.layers.%d.fc2.weight" % l].norm()) == 0.0
    # eval mode: every layer runs (the draws still happen, as in the reference)
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})     # restore the BatchNorm running statistics
    model.eval(); crit.eval()
    with torch.no_grad():
        l_eval, _, _ = crit(model, to_dev(sample))
    (oloss, _, _, _, _, _) = s2t_ref.ctc_multi_loss(W, cfg, sample, 0.1, 1.0, meta["blank"], training=False)
    close(l_eval, float(oloss), 1e-4, "eval loss")


@pytest.mark.parametrize("tag", ["e", "p", "n"])
def test_beam_search_ensemble_prefix_ngram_match_reference_generator(tag):
    """ensemble of two models (log of the mean probability: s2t_ensemble_lse), prefix tokens of different lengths, n-gram blocking:
    the hypotheses of the reference's SequenceGenerator (fixture generate_ext.npz), tokens exact, scores 1e-4"""
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import namespace
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    from helpers import generate_ext_case
    cfg, Ws, src, lens, opts, prefix, exp, meta = generate_ext_case(tag)
    crit = dict(criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy") if meta["compress"] else \
        dict(criterion="label_smoothed_cross_entropy")
    args = namespace(arch="conv_transformer", label_smoothing=0.1, ctc_compress_out=meta["compress"], ctc_encoder_layer=meta["ctc_layer"],
                     ctc_weight=1.0, encoder_embed_dim=cfg["D"], encoder_ffn_embed_dim=cfg["ffn"], encoder_attention_heads=cfg["heads"],
                     encoder_layers=cfg["enc_layers"], decoder_layers=cfg["dec_layers"], no_attn_2d=True, decoder_embed_dim=cfg["D"],
                     decoder_ffn_embed_dim=cfg["ffn"], decoder_attention_heads=cfg["heads"], input_feat_per_channel=80, dropout=0.0,
                     attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False, max_target_positions=1000, **crit)
    tgt, sd = Dictionary.synthetic(96), Dictionary.synthetic(59)
    sd.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(args, tgt, sd)
    models = []
    for W in Ws:
        m = task.build_model(args)
        m.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
        m.materialize(DEV, torch.float32)
        m.eval()
        models.append(m)
    gen = SequenceGenerator(models, task.target_dictionary, **opts)
    hyps = gen.generate(models, dict(net_input=dict(src_tokens=src.to(DEV), src_lengths=lens.to(DEV))),
                        prefix_tokens=None if prefix is None else prefix.to(DEV))
    assert len(hyps) == len(exp)
    for hs, es in zip(hyps, exp):
        assert len(hs) == len(es)
        for h, (et, esc, eps) in zip(hs, es):
            assert h["tokens"].tolist() == et.tolist()
            assert abs(float(h["score"]) - esc) < 1e-4
            np.testing.assert_allclose(h["positional_scores"].cpu().numpy(), eps, atol=1e-4)


@pytest.mark.parametrize("log_probs", [True, False], ids=["log_probs", "probs"])
def test_get_normalized_probs_in_training_mode_is_differentiable(log_probs):
    """fairseq/models/fairseq_decoder.py:58-79 / fairseq_model.py:46-74: a criterion of the reference other than the re-registered ones takes
    (log-)probabilities from the model and differentiates through them.  The golden model in TRAINING mode: a plain NLL (or expected-
    probability) objective built on get_normalized_probs, its gradients against the same objective on the CPU oracle's logits."""
    g, cfg, W, sample, meta, model, crit = build("model_a")
    model.train()
    s = to_dev(sample)
    ni = s["net_input"]
    model.arena.zero_grad()
    eo = model.encoder(ni["src_tokens"], ni["src_lengths"], return_all_hiddens=True)
    net = model.decoder(ni["prev_output_tokens"], encoder_out=eo)
    out = model.get_normalized_probs(net, log_probs=log_probs)
    assert out.dtype == torch.float32 and out.shape == net[0].shape and out.requires_grad
    tgt = s["target"]
    picked = out.gather(-1, tgt.unsqueeze(-1)).squeeze(-1)
    loss = -picked.sum()
    loss.backward()
    torch.cuda.synchronize()
    # oracle: the same objective through torch autograd on the CPU restatement
    Wg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in W.items()}
    enc, _ = s2t_ref.encoder_forward(Wg, cfg, sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"], training=True)
    logits = s2t_ref.decoder_forward(Wg, cfg, sample["net_input"]["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
    ref = torch.log_softmax(logits.float(), -1) if log_probs else torch.softmax(logits.float(), -1)
    close(out, ref.detach().numpy(), 1e-4, "normalized probs")
    oloss = -ref.gather(-1, sample["target"].unsqueeze(-1)).sum()
    oloss.backward()
    assert abs(float(loss) - float(oloss)) <= 1e-4 * max(1.0, abs(float(oloss)))
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    mine = fused_to_reference({n: model.arena.g(n).detach().float().cpu().clone() for n in model.arena.slices})
    for k in ("decoder.output_projection.weight", "decoder.layers.0.fc1.weight", "encoder.layers.0.self_attn.q_proj.weight", "encoder.fc3.weight"):
        a, b = mine[k], Wg[k].grad
        assert float((a - b).norm()) <= 5e-4 * max(float(b.norm()), 1e-6), (k, float((a - b).norm()), float(b.norm()))


@pytest.mark.parametrize("name", ["model_a", "model_b"])
def test_decoder_returns_the_reference_attention(name):
    """fairseq/models/transformer.py:756-782: in eval mode the decoder returns the head-averaged encoder-attention weights of its
    last layer (B, L, Ts); with alignment_layer / alignment_heads those of another layer over its first heads.  Against the fixture
    captured from the real reference (attn.npz).  In training mode they are computed only on request."""
    from helpers import load_golden
    g, cfg, W, sample, meta, model, crit = build(name)
    ga = load_golden("attn")
    s = to_dev(sample)
    ni = s["net_input"]
    model.eval()
    with torch.no_grad():
        eo = model.encoder(ni["src_tokens"], ni["src_lengths"])
        logits, extra = model.decoder(ni["prev_output_tokens"], encoder_out=eo)
        _, extra0 = model.decoder(ni["prev_output_tokens"], encoder_out=eo, alignment_layer=0, alignment_heads=1)
        _, off = model.decoder(ni["prev_output_tokens"], encoder_out=eo, need_attn=False)
    close(logits, g["eval_logits"], 1e-4, "logits (the alignment layer runs on the per-kernel schedule)")
    close(extra["attn"][0], ga[name + "_attn_last"], 1e-4, "attention, last layer")
    close(extra0["attn"][0], ga[name + "_attn_l0h1"], 1e-4, "attention, layer 0 head 0")
    assert off["attn"][0] is None
    model.train()
    with torch.no_grad():
        eo = model.encoder(ni["src_tokens"], ni["src_lengths"])
        _, tr = model.decoder(ni["prev_output_tokens"], encoder_out=eo)
        _, tr1 = model.decoder(ni["prev_output_tokens"], encoder_out=eo, need_attn=True)
    a = tr1["attn"][0]                  # (training-mode BatchNorm statistics move the CTC compression: another source length than the eval fixture)
    assert tr["attn"][0] is None and a.shape[:2] == logits.shape[:2] and a.shape[2] == eo.encoder_out.shape[0]
    assert float((a.sum(-1) - 1).abs().max()) < 1e-4


def test_generator_attention_and_alignment_match_the_reference_generator():
    """sequence_generator.py:286-292,510-560: the `attention` (src_len x tgt_len) the reference's SequenceGenerator attaches to its
    hypotheses, from the incremental HIP decoder with retain_attention; print_alignment adds the hard alignment (arg-max source
    position per target token, utils.extract_hard_alignment).  Fixture attn.npz (generate.npz case a, first two hypotheses)."""
    from helpers import load_golden
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    ga = load_golden("attn")
    task, model, src, lens, opts, exp, _ = build_gen("a")
    gen = SequenceGenerator([model], task.target_dictionary, print_alignment=True, **opts)
    hyps = gen.generate([model], dict(net_input=dict(src_tokens=src, src_lengths=lens)))
    for b, hs in enumerate(hyps):
        for i, h in enumerate(hs[:2]):
            assert h["tokens"].tolist() == ga["gen_a_tokens_%d_%d" % (b, i)].tolist()
            ref = ga["gen_a_attn_%d_%d" % (b, i)]
            close(h["attention"], ref, 1e-4, "hypothesis attention")
            n = h["tokens"].numel() - 1                                   # every position but EOS
            assert [t for _, t in h["alignment"]] == list(range(n))
            assert [s_ for s_, _ in h["alignment"]] == np.argmax(h["attention"].cpu().numpy()[:, :n], axis=0).tolist()
    plain = SequenceGenerator([model], task.target_dictionary, **opts).generate([model], dict(net_input=dict(src_tokens=src, src_lengths=lens)))
    assert all(h["attention"] is None for hs in plain for h in hs)
