"""The one-C-call-per-layer path (csrc/layer.hip, engine.layer_fwd / layer_bwd) against the per-kernel schedules of engine.py it
restates: same launches, same seeds, same bits.  Losses, the CTC-compressed lengths and the encoder output must be IDENTICAL (every
activation on the way is then, too); parameter gradients are sums that end in f32 atomics (grouped weight-gradient launch, LayerNorm /
BatchNorm / bias / convolution sums), whose order differs from run to run even on one path: those agree to f32 rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(act="relu", layerdrop=0.0, p=0.1):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import apply_arch, namespace
    a = namespace(arch="conv_transformer", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy", label_smoothing=0.1,
                  ctc_compress_out=True, ctc_encoder_layer=2, ctc_weight=1.0, encoder_embed_dim=128, encoder_ffn_embed_dim=512,
                  encoder_attention_heads=2, encoder_layers=3, decoder_layers=2, decoder_embed_dim=128, decoder_ffn_embed_dim=512,
                  decoder_attention_heads=2, no_attn_2d=True, input_feat_per_channel=80, dropout=p, attention_dropout=p, activation_dropout=p,
                  relu_dropout=p, activation_fn=act, sentence_avg=False, encoder_layerdrop=layerdrop, decoder_layerdrop=layerdrop, seed=7)
    apply_arch(a)
    tgt, src = Dictionary.synthetic(300), Dictionary.synthetic(200)
    src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(a, tgt, src)
    torch.manual_seed(3)
    model, crit = task.build_model(a), task.build_criterion(a)
    model.materialize(DEV, torch.bfloat16, extra=crit.arena_params())
    return a, task, model, crit


def step(model, crit, sample, composite, training=True, manual_seed=None):
    model.engine.composite = composite
    model.train(training); crit.train(training)
    model.set_seed(11)
    model.arena.zero_grad()
    if manual_seed is not None:
        torch.manual_seed(manual_seed)
    if training:
        loss, ss, log = crit(model, sample)
        loss.backward()
    else:
        with torch.no_grad():
            loss, ss, log = crit(model, sample)
    torch.cuda.synchronize()
    enc = model.encoder._last
    return float(loss), enc["out"].clone(), list(enc["lengths_host"]), {n: model.arena.g(n).clone() for n in model.arena.slices}


@pytest.mark.parametrize("act,lengths", [("relu", None), ("relu", [1210, 1500, 777]), ("gelu", [900, 640])])
def test_layer_calls_reproduce_the_per_kernel_schedule(act, lengths):
    from fbk_fairseq_st_amd.data import synthetic_batch
    a, task, model, crit = build(act)
    B = len(lengths) if lengths else 4
    T = max(lengths) if lengths else 1000
    s = synthetic_batch(B, T, 12, 10, len(task.target_dictionary), task.source_dictionary.index("<ctc_blank>"), seed=5, lengths=lengths)
    sample = {k: (v.to(DEV) if torch.is_tensor(v) else ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v)) for k, v in s.items()}
    l0, e0, n0, g0 = step(model, crit, sample, composite=False)
    l1, e1, n1, g1 = step(model, crit, sample, composite=True)
    assert l0 == l1 and n0 == n1 and torch.equal(e0, e1)
    for n in g0:
        assert torch.allclose(g1[n], g0[n], rtol=2e-5, atol=2e-6 * max(float(g0[n].abs().max()), 1e-30)), n
    # evaluation mode: no dropout, no ReLU record
    le0, ee0, _, _ = step(model, crit, sample, composite=False, training=False)
    le1, ee1, _, _ = step(model, crit, sample, composite=True, training=False)
    assert le0 == le1 and torch.equal(ee0, ee1)


def test_layer_calls_with_layerdrop():
    from fbk_fairseq_st_amd.data import synthetic_batch
    a, task, model, crit = build("relu", layerdrop=0.4)
    s = synthetic_batch(3, 800, 12, 10, len(task.target_dictionary), task.source_dictionary.index("<ctc_blank>"), seed=6)
    sample = {k: (v.to(DEV) if torch.is_tensor(v) else ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v)) for k, v in s.items()}
    seed = next(sd for sd in range(100) if _decisions(sd, 3, 2, 0.4))
    l0, e0, n0, g0 = step(model, crit, sample, composite=False, manual_seed=seed)
    l1, e1, n1, g1 = step(model, crit, sample, composite=True, manual_seed=seed)
    assert l0 == l1 and torch.equal(e0, e1)
    dropped = [n for n in g0 if n.endswith("fc1.weight") and float(g0[n].abs().max()) == 0.0]
    assert dropped, "the case must drop a layer"
    for n in g0:
        assert torch.allclose(g1[n], g0[n], rtol=2e-5, atol=2e-6 * max(float(g0[n].abs().max()), 1e-30)), n


def _decisions(seed, el, dl, p):
    """a forward seed that keeps the CTC layer (2), drops at least one encoder and one decoder layer"""
    torch.manual_seed(seed)
    e = [float(torch.empty(1).uniform_()) > p for _ in range(el)]
    d = [v > p for v in torch.empty(dl).uniform_().tolist()]
    return e[1] and not all(e) and not all(d)
