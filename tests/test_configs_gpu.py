"""Parity at the sizes BASELINE.json quotes (configs[2], [3], [4] = SURVEY.md 8-d Cfg3 / Cfg4 / Cfg5), HIP path vs the CPU oracle.

  Cfg3  s2t_transformer_m (D 512, 8 heads, 12+6 layers) + ctc_multi_loss with --ctc-compress-out after layer 8, T = 1500 frames,
        fp32 and bf16, a fixed-length and a ragged batch -- the bench workload's shapes (per utterance), so the kernels taken are
        the ones the bench runs (attention at 375 keys, 128x128 / 256x256 GEMM tiles, V_src 5001 logit rows).
  Cfg4  s2t_transformer_l (128 conv channels, D 1024, 16 heads), MuST-C-shaped ragged lengths (lognormal), fp32 and bf16.
  Cfg5  m preset: knowledge distillation (top-8 teacher), dual decoder + cross_entropy_dualdecoder, beam-5 generation through
        SequenceGenerator and the two-phase generator.

The oracle (oracle/s2t_ref.py) is pinned to the real reference by tests/test_oracle_golden.py on the golden fixtures; here it
runs at full model size on the host.  Tolerances, stated per test:
  fp32 mode  loss / logging scalars 1e-4 relative, per-tensor gradient norms 1e-3 relative (north_star: fp32 1e-4; the
             gradient chains run through 18 layers and T4 = 375 positions of f32 accumulation in a different order);
  bf16 mode  (bf16 storage, f32 accumulation and f32 master weights) loss 2e-2 relative, per-tensor gradient norms 1e-1
             relative, direction of selected gradient tensors: cosine >= 0.99 against the f32 oracle.
  integers   CTC arg-max / run-length collapse / new lengths: bit-exact against oracle/int_ref on the engine's OWN logits, in
             both modes; in fp32 mode also bit-exact against the oracle's own f32 forward.  In bf16 mode the float path is compared
             with the oracle GIVEN the engine's integer path (s2t_ref.ctc_compress(pred_override=...)): an arg-max over
             bf16-rounded logits may legitimately differ from the f32 one at near-ties; the number of such frames is reported.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import int_ref, s2t_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
V_TGT, V_SRC = 8000, 5000                      # typical BPE sizes (+ <ctc_blank> appended to the source side)


def to_dev(s):
    if torch.is_tensor(s):
        return s.to(DEV)
    if isinstance(s, dict):
        return {k: to_dev(v) for k, v in s.items()}
    if isinstance(s, (list, tuple)):
        return type(s)(to_dev(v) for v in s)
    return s


def build(arch, dtype, criterion="ctc_multi_loss", seed=11, dual=False, ctc_layer=8, **over):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import apply_arch, namespace
    is_ctc = criterion == "ctc_multi_loss"
    kw = dict(arch=arch, criterion=criterion, label_smoothing=0.1, input_feat_per_channel=80, dropout=0.0, attention_dropout=0.0,
              activation_dropout=0.0, relu_dropout=0.0, sentence_avg=False, no_attn_2d=True, max_target_positions=1000)
    kw.update(over)
    a = namespace(**kw)
    if is_ctc:
        a.underlying_criterion, a.ctc_compress_out, a.ctc_encoder_layer, a.ctc_weight = "label_smoothed_cross_entropy", True, ctc_layer, 1.0
    apply_arch(a)
    tgt, src = Dictionary.synthetic(V_TGT - 4), Dictionary.synthetic(V_SRC - 4)
    if is_ctc:
        src.add_symbol("<ctc_blank>")
    task = tasks.SpeechTranslationCTCTask(a, tgt, src)
    model, crit = task.build_model(a), task.build_criterion(a)
    conv = eval(a.encoder_convolutions)[0][0]
    cfg = s2t_ref.default_cfg(D=a.encoder_embed_dim, heads=a.encoder_attention_heads, ffn=a.encoder_ffn_embed_dim,
                              enc_layers=a.encoder_layers, dec_layers=a.decoder_layers, ctc_layer=ctc_layer if is_ctc else 0,
                              conv_ch=conv, act=a.activation_fn,
                              share_dec_embed=bool(over.get("share_decoder_input_output_embed", False)))
    W = s2t_ref.make_weights(s2t_ref.param_shapes(cfg, len(src), len(tgt), criterion_fc=is_ctc, V_aux=len(src) if dual else 0), seed)
    if is_ctc:
        # a CTC head whose arg-max wanders over a handful of units (and the blank): runs of equal predictions occur, the
        # compression really shortens the sequences (random-init logits over 5001 units would almost never repeat)
        blank = src.index("<ctc_blank>")
        W["encoder.ctc_fc.bias"][[7, 19, 123, 2048, 4999, blank]] += 6.0
    if "decoder.output_projection.weight" in W:
        W["decoder.output_projection.weight"][2] *= 4.0       # EOS reachable for the generation tests (as tests/golden/make_golden.py)
    model.load_state_dict({k: v for k, v in W.items() if not k.startswith("criterion.")})
    if is_ctc:
        with torch.no_grad():
            crit.ctc_aware_model.fc_out.weight.copy_(W["criterion.ctc_aware_model.fc_out.weight"])
            crit.ctc_aware_model.fc_out.bias.copy_(W["criterion.ctc_aware_model.fc_out.bias"])
    model.hp.sub_dropout = 0.0                                 # parity mode (the subsampler's rate is max(p, 0.1), conv_transformer.py:214)
    model.materialize(DEV, dtype, extra=crit.arena_params() if hasattr(crit, "arena_params") else None)
    return a, task, model, crit, cfg, W


def batch(task, B, T, L, Lt, seed, lengths=None):
    from fbk_fairseq_st_amd.data import synthetic_batch
    src = task.source_dictionary
    hi = src.index("<ctc_blank>") if "<ctc_blank>" in src.indices else len(src)
    return synthetic_batch(B, T, L, Lt, len(task.target_dictionary), hi, seed=seed, lengths=lengths)


def oracle_grads(W, run):
    Wg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in W.items()}
    out = run(Wg)
    out[0].backward()
    return out, {k: v.grad for k, v in Wg.items() if v.grad is not None}


def engine_grads(model):
    from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
    return fused_to_reference({n: model.arena.g(n).detach().float().cpu().clone() for n in model.arena.slices})


def compare_grads(mine, ref, tol, cos_min=None, what=""):
    """per-tensor gradient norms, relative; tensors whose exact gradient is zero (the key bias of an attention: softmax is
    invariant to it) hold rounding noise on both sides and are compared against a floor of 1e-6 of the largest tensor norm"""
    worst = (0.0, None)
    floor = 1e-6 * max(float(g.norm()) for g in ref.values())
    for k, g in ref.items():
        if k.endswith("_float_tensor"):
            continue
        assert k in mine, k
        a, b = float(mine[k].norm()), float(g.norm())
        if k.endswith("k_proj.bias"):
            # exactly zero in exact arithmetic: what is left is the rounding of sum_t dK[t, :] (f32 ~1e-7, bf16 rows ~1e-3 of the
            # weight gradient computed from the same dK); bound it by the tolerance times that weight gradient's norm
            assert a <= tol * float(ref[k[:-4] + "weight"].norm()) + floor, (k, a, float(ref[k[:-4] + "weight"].norm()))
            continue
        err = abs(a - b) / max(b, floor)
        if err > worst[0]:
            worst = (err, k)
        assert err <= tol, "%s gradient norm of %s: %.6g vs oracle %.6g (rel %.3e > %.1e)" % (what, k, a, b, err, tol)
    wcos = (1.0, None)
    if cos_min is not None:
        # direction of EVERY gradient tensor that carries signal (norm above 1e-3 of the largest): cosine against the oracle
        big = 1e-3 * max(float(g.norm()) for g in ref.values())
        for k, g in ref.items():
            if k.endswith("_float_tensor") or k.endswith("k_proj.bias") or float(g.norm()) < big:
                continue
            c = float(torch.nn.functional.cosine_similarity(mine[k].reshape(1, -1).double(), g.reshape(1, -1).double()))
            if c < wcos[0]:
                wcos = (c, k)
            assert c >= cos_min, "%s gradient direction of %s: cosine %.6f < %.6f" % (what, k, c, cos_min)
    print("MEASURED %s: worst gradient-norm error %.3e (%s), worst cosine %.6f (%s)" % (what, worst[0], worst[1], wcos[0], wcos[1]))
    tripwire(what, grad=worst[0], one_minus_cos=1.0 - wcos[0])
    return worst


# ---- tripwire: the bf16 bounds below are ~2x the measured worst case, so a regression of less than 2x would pass them silently.
# tests/golden/parity_measured.json holds what a known-good build measured on an MI355X, per test case; a bf16 figure more than 1.3x
# (plus a floor for run-to-run noise: atomics order) above its committed value fails here.  S2T_WRITE_MEASURED=<file> records a run.
_MEASURED_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_measured.json")
_MEASURED = json.load(open(_MEASURED_PATH)) if os.path.exists(_MEASURED_PATH) else {}
_FLOOR = dict(grad=2e-3, one_minus_cos=1e-3, loss=3e-4, flips=0.005, gen=1e-2)


def tripwire(what, **figures):
    if "bfloat16" not in what and "bf16" not in what:
        return
    case = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
    key = "%s | %s" % (case, what)
    out = os.environ.get("S2T_WRITE_MEASURED")
    if out:
        rec = json.load(open(out)) if os.path.exists(out) else {}
        rec.setdefault(key, {}).update({k: float(v) for k, v in figures.items()})
        json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
        return
    base = _MEASURED.get(key)
    if base is None:
        return
    for k, v in figures.items():
        if k in base:
            assert float(v) <= 1.3 * base[k] + _FLOOR[k], "%s: %s %.3e against %.3e measured by the committed build (tripwire 1.3x)" % (key, k, v, base[k])


def rel(a, b):
    a, b = float(a), float(b)
    return abs(a - b) / max(abs(b), 1e-12)


# fp32: the median per-tensor gradient-norm error is ~1e-6; the tolerance is 1e-3 because the comparison contains DISCRETE decisions.  The
# forward passes of the engine and of the oracle differ by f32 rounding (~1e-7: summation orders, the order in which double atomics add the
# BatchNorm statistics), and a ReLU input within that distance of zero is active in one and not in the other.  One such flip in a decoder
# FFN (80 token rows at B = 2, L = 40) moves that layer's weight gradient by ~5e-3 in direction and ~4e-4 in norm, and everything
# upstream of it by ~1e-3 -- measured: with the same build, the first update of a process and the third differ by exactly this pattern
# (decoder.layers.2.fc1 7e-3, median 2e-5), and which seed shows it changes whenever a kernel's summation order does.
# (With a plain float32 log-space CTC recursion in the product the error was 1e-3 for EVERYTHING below the CTC tap: alpha + beta - log P
# cancels numbers near -3,000.  The oracle runs its recursion in float64; the kernels now keep every step's vector relative to its
# maximum and normalise the posteriors per frame: 4e-6 from float64 where torch's own f32 ctc_loss is 6e-4, tools/ctc_accuracy.py.)
# bf16 (bf16 storage, f32 accumulation, f32 masters) against the f32 oracle: bounds = ~2x the worst case MEASURED on an MI355X over the
# cases of this file (every test prints its own "MEASURED ..." line; profiles/r03_parity_measured.txt holds the run the bounds come from):
#   loss                  worst 2.8e-3 (l preset, ragged)                                -> 6e-3
#   gradient norms        worst 1.2e-2 ctc_multi_loss m / l, 2.0e-2 KD, 2.3e-2 shared embedding, 4.4e-2 dual decoder (encoder.bn.0.weight)
#                                                                                        -> 5e-2 (dual decoder: 9e-2)
#   gradient direction    cosine on EVERY tensor whose norm is above 1e-3 of the largest: worst 0.9972 ctc_multi_loss, 0.9951 shared
#                         embedding, 0.9968 dual decoder, 0.9915 KD (decoder.layers.5.encoder_attn.q_proj.weight)   -> 0.99 (KD: 0.983)
TOL = {torch.float32: dict(loss=1e-4, grad=1e-3, cos=0.9999), torch.bfloat16: dict(loss=6e-3, grad=5e-2, cos=0.99)}
FP32_NORTH_STAR = 1e-4          # BASELINE.json north_star: "within fp32 1e-4"; held by the comparison with shared ReLU decisions
TOL_BF16_KD = dict(loss=6e-3, grad=5e-2, cos=0.983)
TOL_BF16_DUAL = dict(loss=6e-3, grad=9e-2, cos=0.99)


# frames whose arg-max over the engine's bf16-computed logits differs from the arg-max over the f32 oracle's logits (near-ties between
# the handful of boosted units of build()): measured 16 / 750, 14 / 873, 12 / 763 = 1.6-2.1 %  -> bound 4 %
BF16_MAX_FLIP_FRACTION = 0.04


def check_ctc_multi_loss(arch, dtype, B, T, L, lengths, seed, **over):
    a, task, model, crit, cfg, W = build(arch, dtype, **over)
    blank = task.source_dictionary.index("<ctc_blank>")
    sample = batch(task, B, T, L, L, seed, lengths)
    model.train(); crit.train()
    model.arena.zero_grad()
    if dtype == torch.float32:
        model._ensure_engine(DEV)
        model.engine.relu_record = {}                   # the ReLU decisions of this pass, for the second oracle run below
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()
    torch.cuda.synchronize()
    last = model.encoder._last
    # ---- integer path, exact, on the engine's own logits (the CTC-compression decisions the rest of the step was computed with)
    x_ctc = last["ctc_out"].detach().float().cpu()                                # [T4, B, V] (row-padded view -> dense copy)
    prob = torch.softmax(x_ctc, dim=-1).transpose(0, 1).contiguous().numpy()
    pred_ref = int_ref.argmax_first_np(prob)
    pred = last["pred"].detach().cpu().numpy()                                    # [B, T4] int32
    len4 = np.asarray(last["ctc_lengths_host"])
    same = all(np.array_equal(pred[b, :len4[b]], pred_ref[b, :len4[b]]) for b in range(len(len4)))
    assert same, "arg-max over the engine's logits differs from oracle/int_ref on the same logits"
    runs = int_ref.ctc_rle_np(pred_ref, len4)
    assert [len(r) for r in runs] == [int(v) for v in last["lengths_host"]], "new lengths after the run-length collapse"
    assert min(len(r) for r in runs) < int(len4.max()), "the test case must actually compress"
    # ---- float path against the oracle
    force = None if dtype == torch.float32 else pred_ref
    (oloss, oss, olog, enc, _, _), ograds = oracle_grads(W, lambda Wg: s2t_ref.ctc_multi_loss(Wg, cfg, sample, 0.1, 1.0, blank, training=True,
                                                                                           pred_override=force))
    if dtype == torch.float32:
        opred = enc.ctc_pred.numpy()
        assert all(np.array_equal(pred[b, :len4[b]], opred[b, :len4[b]]) for b in range(len(len4))), "fp32: arg-max differs from the oracle's own forward"
        assert [int(v) for v in enc.new_lengths] == [int(v) for v in last["lengths_host"]]
    else:
        flips = sum(int((pred_ref[b, :len4[b]] != int_ref.argmax_first_np(
            torch.softmax(enc.ctc_out.detach()[:, b], -1).numpy())[:len4[b]]).sum()) for b in range(len(len4)))
        print("MEASURED bf16: %d of %d frames pick another unit than the f32 oracle's logits would" % (flips, int(len4.sum())))
        tripwire("%s bf16 arg-max flips" % arch, flips=flips / float(len4.sum()))
        assert flips <= BF16_MAX_FLIP_FRACTION * int(len4.sum()), (flips, int(len4.sum()))
    t = TOL[dtype]
    assert ss == oss
    assert rel(loss, oloss) <= t["loss"], ("loss", float(loss), float(oloss))
    assert rel(log["ctc_loss"], olog["ctc_loss"]) <= t["loss"] and rel(log["nll_loss"], olog["nll_loss"]) <= t["loss"]
    for k in ("ntokens", "nsentences", "sample_size", "nframes", "ctc_total"):
        assert float(log[k]) == float(olog[k]), k
    if dtype == torch.float32:
        assert float(log["ctc_errors"]) == float(olog["ctc_errors"])
    worst = compare_grads(engine_grads(model), ograds, t["grad"], t["cos"], what="%s %s" % (arch, dtype))
    print("%s %s: loss %.6f (oracle %.6f), worst gradient-norm error %.2e at %s, frames %s -> %s" %
          (arch, dtype, float(loss), float(oloss), worst[0], worst[1], list(len4), last["lengths_host"]))
    if dtype == torch.float32 and cfg["act"] == "relu":
        # ---- north_star's fp32 bound (1e-4) with the DISCRETE decisions taken out of the comparison: the oracle runs again with the
        # engine's own ReLU active sets (as `pred_override` does for the arg-max in bf16 mode), so a pre-activation within f32 rounding
        # of zero cannot be active on one side only; what remains is floating-point arithmetic in another summation order.  The
        # free-running comparison above stays, at 1e-3.
        masks = {k: v.cpu() for k, v in model.engine.relu_record.items()}
        model.engine.relu_record = None
        n_sites = 3 + cfg["enc_layers"] + cfg["dec_layers"]
        assert len(masks) == n_sites, (sorted(masks), n_sites)
        cfg_m = dict(cfg, relu_masks=masks)
        (mloss, _, _, _, _, _), mgrads = oracle_grads(W, lambda Wg: s2t_ref.ctc_multi_loss(Wg, cfg_m, sample, 0.1, 1.0, blank, training=True))
        assert rel(loss, mloss) <= 1e-4
        wm = compare_grads(engine_grads(model), mgrads, FP32_NORTH_STAR, t["cos"], what="%s %s, ReLU decisions shared" % (arch, dtype))
        print("%s fp32 with shared ReLU decisions: worst gradient-norm error %.2e at %s" % (arch, wm[0], wm[1]))


# ------------------------------------------------------------------------------------------------ Cfg2
def test_cfg2_s_fp32_32x1000():
    """BASELINE.json configs[1] at its full shape: s2t_transformer_s, 32 utterances x 1000 frames x 80 mel, fp32,
    ctc_multi_loss with compression as in examples/speech_recognition/criterions/ctc_multi_loss.py:140-168"""
    check_ctc_multi_loss("s2t_transformer_s", torch.float32, B=32, T=1000, L=30, lengths=None, seed=12)


def test_cfg2_s_fp32_32x1000_ragged():
    rs = np.random.RandomState(5)
    lengths = sorted([1000] + [int(v) for v in rs.randint(300, 1000, 31)], reverse=True)
    check_ctc_multi_loss("s2t_transformer_s", torch.float32, B=32, T=1000, L=30, lengths=lengths, seed=13)


# ------------------------------------------------------------------------------------------------ Cfg3
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_cfg3_m_ctc_compression_T1500(dtype):
    check_ctc_multi_loss("s2t_transformer_m", dtype, B=2, T=1500, L=40, lengths=None, seed=3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_cfg3_m_ctc_compression_ragged(dtype):
    check_ctc_multi_loss("s2t_transformer_m", dtype, B=3, T=1500, L=40, lengths=[1500, 1210, 777], seed=4)


def test_cfg3_m_bf16_gemm256_route():
    """The route the bench is timed on, against the oracle (parity 1): s2t_transformer_m at 16 x 1500 frames = 6,000 encoder
    tokens in bf16, with the 256-wide kernel's tile threshold lowered ("gemm256_min_tiles": the measurement option of the library) so
    that EVERY encoder product of the layer calls -- csrc/layer.hip -> gemm256 with its 1-bit ReLU record (fc1) and the record-reading
    data gradient (fc2), the residual / accumulate epilogues -- and the grouped weight-gradient launch run as they do at 64 x 1500.
    Same tolerances and tripwire as the small bf16 cases; the launch counts of the profiling families prove the route."""
    from fbk_fairseq_st_amd import kernels as K
    old = K.set_option("gemm256_min_tiles", 32)
    try:
        assert K.relu_mask_bytes(16 * 375, 2048, 512) > 0
        K.prof_reset(); K.prof_enable(1)
        check_ctc_multi_loss("s2t_transformer_m", torch.bfloat16, B=16, T=1500, L=40, lengths=None, seed=21)
        torch.cuda.synchronize()
        fam = {f: K.prof_read(f)["launches"] for f in ("gemm256_nt", "gemm256_nn", "wgrad_group")}
    finally:
        K.prof_enable(0)
        K.set_option("gemm256_min_tiles", old)
    print("MEASURED gemm256 route launches:", fam)
    # the 8 encoder layers in front of the compression x (qkv, out, fc1, fc2) forward (measured on an MI355X: 32 NT + 62 NN launches: the
    # four layers behind the compression see ~2,400 tokens, below even the lowered threshold, and take the 128-wide route), their data
    # gradients (+ the decoder's K/V data gradients into the encoder output), two grouped weight-gradient launches
    assert fam["gemm256_nt"] >= 32 and fam["gemm256_nn"] >= 32 and fam["wgrad_group"] >= 2, fam


def test_cfg3_m_bf16_gemm256_route_equals_the_128_wide_route_with_dropout():
    """the same 16 x 1500 batch with the preset's dropout ON, once through gemm256 (masked epilogues, 1-bit record) and once with the
    256-wide kernel switched off: both routes run the same MFMA instruction over K in the same order and draw the same dropout masks,
    so loss, CTC lengths and encoder output are IDENTICAL and the parameter gradients agree to the order of f32 atomics"""
    from fbk_fairseq_st_amd import kernels as K

    def run(g256):
        old = (K.set_option("gemm256_min_tiles", 32), K.set_option("gemm256", g256))
        try:
            a, task, model, crit, cfg, W = build("s2t_transformer_m", torch.bfloat16, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1)
            sample = batch(task, 16, 1500, 40, 40, 22, None)
            model.train(); crit.train()
            model.set_num_updates(3) if hasattr(model, "set_num_updates") else None
            model.arena.zero_grad()
            loss, ss, log = crit(model, to_dev(sample))
            loss.backward()
            torch.cuda.synchronize()
            last = model.encoder._last
            return (float(loss), [int(v) for v in last["lengths_host"]], float(log["ctc_loss"]), float(log["nll_loss"]),
                    {n: model.arena.g(n).detach().float().cpu().clone() for n in model.arena.slices})
        finally:
            K.set_option("gemm256_min_tiles", old[0]); K.set_option("gemm256", old[1])
    l1, len1, c1, n1, g1 = run(1)
    l0, len0, c0, n0, g0 = run(0)
    assert len1 == len0 and l1 == l0 and c1 == c0 and n1 == n0, (l1, l0, c1, c0, n1, n0)
    worst = max((float((g1[k] - g0[k]).norm()) / max(float(g0[k].norm()), 1e-12), k) for k in g0 if float(g0[k].norm()) > 0)
    print("MEASURED gemm256 vs 128-wide route with dropout: identical loss %.6f, worst gradient difference %.2e (%s)" % (l1, worst[0], worst[1]))
    assert worst[0] < 1e-3, worst          # f32 atomics order in the grouped weight gradients / LayerNorm parameter sums only


def test_cfg3_m_gelu_model_level():
    """--activation-fn gelu through the whole model (the GELU epilogues were only pinned at kernel level)"""
    check_ctc_multi_loss("s2t_transformer_m", torch.float32, B=2, T=600, L=20, lengths=[600, 455], seed=5, activation_fn="gelu",
                         encoder_layers=4, decoder_layers=2, ctc_layer=2)


# ------------------------------------------------------------------------------------------------ Cfg4
def mustc_lengths(n, seed):
    """T ~ lognormal(mu = ln 600, sigma = 0.7) clipped to [50, 2000]"""
    rs = np.random.RandomState(seed)
    return [int(v) for v in np.clip(rs.lognormal(np.log(600.0), 0.7, n), 50, 2000)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_cfg4_l_mustc_shaped_lengths(dtype):
    lengths = mustc_lengths(4, 2)
    check_ctc_multi_loss("s2t_transformer_l", dtype, B=4, T=max(lengths), L=24, lengths=lengths, seed=6)


# ------------------------------------------------------------------------------------------------ Cfg5
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_cfg5_m_knowledge_distillation(dtype):
    a, task, model, crit, cfg, W = build("s2t_transformer_m", dtype, criterion="knowledge_distillation", kd_lambda=0.6, kd_temperature=2.0)
    sample = batch(task, 4, 1000, 30, 30, 7, lengths=[1000, 1000, 870, 640])
    g = torch.Generator().manual_seed(1)
    K = 8
    tidx = torch.stack([torch.randperm(V_TGT, generator=g)[:K] for _ in range(4 * 30)]).view(4, 30, K)
    tidx[:, :, 0] = sample["target"]                               # the teacher usually ranks the reference token first
    tlog = torch.randn(4, 30, K, generator=g).sort(dim=-1, descending=True)[0] * 2.0
    sample["teacher_output"] = [tidx, tlog]
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()

    def run(Wg):
        ni = sample["net_input"]
        enc, _ = s2t_ref.encoder_forward(Wg, cfg, ni["src_tokens"], ni["src_lengths"], training=True)
        logits = s2t_ref.decoder_forward(Wg, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
        return (s2t_ref.kd_loss(logits, sample["target"], tidx, tlog, 0.6, 2.0, cfg["pad"]),)
    (oloss,), ograds = oracle_grads(W, run)
    t = TOL[dtype] if dtype == torch.float32 else TOL_BF16_KD
    assert ss == sample["ntokens"]
    assert rel(loss, oloss) <= t["loss"], (float(loss), float(oloss))
    compare_grads(engine_grads(model), ograds, t["grad"], t["cos"], what="kd %s" % dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_shared_decoder_input_output_embedding(dtype):
    """--share-decoder-input-output-embed (fairseq/models/transformer.py:538,618-624): one [V, D] table serves the token look-up and
    the output projection; its gradient is the sum of the embedding scatter and the projection's dW"""
    a, task, model, crit, cfg, W = build("s2t_transformer_s", dtype, criterion="label_smoothed_cross_entropy",
                                         share_decoder_input_output_embed=True)
    assert "decoder.output_projection.weight" not in W and "decoder.output_projection.weight" not in model.arena.slices
    sd = model.state_dict()                                         # the reference's checkpoints carry the tensor under both names
    assert torch.equal(sd["decoder.output_projection.weight"].float(), sd["decoder.embed_tokens.weight"].float())
    sample = batch(task, 4, 600, 24, 20, 9, lengths=[600, 600, 531, 322])
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()

    def run(Wg):
        ni = sample["net_input"]
        enc, _ = s2t_ref.encoder_forward(Wg, cfg, ni["src_tokens"], ni["src_lengths"], training=True)
        logits = s2t_ref.decoder_forward(Wg, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
        return s2t_ref.label_smoothed_nll(logits, sample["target"], 0.1, cfg["pad"])
    out, ograds = oracle_grads(W, run)
    t = TOL[dtype]
    assert rel(loss, out[0]) <= t["loss"], (float(loss), float(out[0]))
    compare_grads(engine_grads(model), ograds, t["grad"], t["cos"], what="shared embed %s" % dtype)
    # the embedding gradient is NOT the scatter alone: the projection's share dominates it
    assert float(ograds["decoder.embed_tokens.weight"].norm()) > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_cfg5_m_dual_decoder_loss(dtype):
    a, task, model, crit, cfg, W = build("conv_transformer_dualdecoder_big2", dtype, criterion="cross_entropy_dualdecoder",
                                         dual=True, encoder_layers=12, auxiliary_loss_weight=0.3, primary_loss_weight=0.7)
    sample = batch(task, 3, 1000, 30, 26, 8, lengths=[1000, 910, 505])
    tr = sample["transcript_target"]
    sample["net_input"]["transcript_prev_output_tokens"] = torch.cat([torch.full((3, 1), 2, dtype=torch.long), tr[:, :-1]], 1)
    model.train(); crit.train()
    model.arena.zero_grad()
    loss, ss, log = crit(model, to_dev(sample))
    loss.backward()
    (oloss, olog, _, _), ograds = oracle_grads(W, lambda Wg: s2t_ref.dual_decoder_loss(Wg, cfg, sample, 0.1, 0.7, 0.3, training=True))
    t = TOL[dtype] if dtype == torch.float32 else TOL_BF16_DUAL
    assert rel(loss, oloss) <= t["loss"], (float(loss), float(oloss))
    for k in ("primary_loss", "auxiliary_loss", "primary_nll_loss", "auxiliary_nll_loss"):
        assert rel(log[k], olog[k]) <= t["loss"], k
    print("MEASURED dual %s: loss error %.3e" % (dtype, rel(loss, oloss)))
    tripwire("dual %s loss" % dtype, loss=rel(loss, oloss))
    compare_grads(engine_grads(model), ograds, t["grad"], t["cos"], what="dual %s" % dtype)


def test_cfg5_m_beam5_generation():
    """beam-5 through SequenceGenerator on the m preset vs oracle.beam_search (tokens exact, scores 1e-4; fp32)"""
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    a, task, model, crit, cfg, W = build("s2t_transformer_m", torch.float32, criterion="label_smoothed_cross_entropy")
    sample = batch(task, 3, 1000, 8, 8, 9, lengths=[1000, 731, 402])
    src, lens = sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]
    opts = dict(beam_size=5, max_len_a=0.0, max_len_b=14, min_len=1, len_penalty=1.0, unk_penalty=0.0, temperature=1.0)
    model.eval()
    gen = SequenceGenerator([model], task.target_dictionary, **opts)
    hyps = gen.generate([model], dict(net_input=dict(src_tokens=src.to(DEV), src_lengths=lens.to(DEV))))
    orc = s2t_ref.beam_search(W, cfg, src, lens, 5, 0.0, 14, 1, 1.0, 0.0, 1.0)
    assert len(hyps) == 3
    for hs, os_ in zip(hyps, orc):
        assert len(hs) == len(os_) == 5
        for h, (ot, osc, ops) in zip(hs, os_):
            assert h["tokens"].tolist() == ot.tolist()
            assert abs(float(h["score"]) - osc) < 1e-4
            np.testing.assert_allclose(h["positional_scores"].cpu().numpy(), ops, atol=1e-4)


@pytest.mark.parametrize("arch,dtype,beam,lengths,max_len_b", [
    ("s2t_transformer_m", torch.float32, 12, [1000, 600], 10),        # beam > 8: 16-row attention loops, output projection not in registers
    ("s2t_transformer_l", torch.float32, 5, [1300, 700], 8),          # D 1024, 16 heads (two prologue passes, two share batches); 325 encoder
                                                                     # frames: keys / values requested where they are used
    ("s2t_transformer_s", torch.float32, 8, [400, 250, 90], 12),      # D 256, 100 encoder frames: one 128-position block in registers
    ("s2t_transformer_m", torch.bfloat16, 12, [1000, 600], 10),
    ("s2t_transformer_s", torch.bfloat16, 8, [400, 250, 90], 12)])
def test_device_search_equals_step_search(arch, dtype, beam, lengths, max_len_b, monkeypatch):
    """The device-resident search (csrc/decode.hip) at the shapes the reference fixtures do not reach -- beams above 8, D = 1024,
    encoder lengths on either side of the 256 frames that fit in registers -- against the step-by-step search of the same library
    (pinned to the reference by tests/test_model_gpu.py): fp32 tokens exact and scores 1e-4; bf16 (the two routes round differently:
    shares and LayerNorm inputs are bf16 in one, f32 in the other) well-formed hypotheses whose best scores agree to BF16_GEN_ATOL."""
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    a, task, model, crit, cfg, W = build(arch, dtype, criterion="label_smoothed_cross_entropy")
    sample = batch(task, len(lengths), max(lengths), 8, 8, 9, lengths=lengths)
    net = dict(net_input=dict(src_tokens=sample["net_input"]["src_tokens"].to(DEV), src_lengths=sample["net_input"]["src_lengths"].to(DEV)))
    opts = dict(beam_size=beam, max_len_a=0.0, max_len_b=max_len_b, min_len=1, len_penalty=1.0, unk_penalty=0.0, temperature=1.0)
    model.eval()
    gen = SequenceGenerator([model], task.target_dictionary, **opts)
    dev_h = gen.generate([model], net)
    assert "launches_per_step" in gen.last_stats, "the device route was not taken"
    monkeypatch.setenv("S2T_DEVICE_SEARCH", "0")
    gen2 = SequenceGenerator([model], task.target_dictionary, **opts)
    step_h = gen2.generate([model], net)
    assert "launches_per_step" not in gen2.last_stats
    assert len(dev_h) == len(step_h) == len(lengths)
    for hs, ss in zip(dev_h, step_h):
        assert len(hs) == len(ss) == beam
        if dtype == torch.float32:
            for h, s_ in zip(hs, ss):
                assert h["tokens"].tolist() == s_["tokens"].tolist()
                assert abs(float(h["score"]) - float(s_["score"])) < 1e-4
                np.testing.assert_allclose(h["positional_scores"].cpu().numpy(), s_["positional_scores"].cpu().numpy(), atol=1e-4)
        else:
            sc = [float(h["score"]) for h in hs]
            assert sc == sorted(sc, reverse=True)
            for h in hs:
                assert int(h["tokens"][-1]) == 2 and not bool((h["tokens"][:-1] == 2).any())
            assert abs(sc[0] - float(ss[0]["score"])) < BF16_GEN_ATOL


def test_cfg5_m_beam5_generation_bf16():
    """beam-5 on the m preset in bf16 mode.  A bf16 forward can legitimately order two near-tied candidates differently from the f32
    oracle, so the statement that holds is about scores: (1) every hypothesis the engine returns, re-scored by the f32 oracle with
    its tokens forced, has the positional scores the engine reported within BF16_GEN_ATOL; (2) the engine's best hypothesis of every
    sentence is the oracle's best, or its oracle score is within BF16_GEN_ATOL of the oracle's best (a near-tie).  The number of
    (sentence, rank) slots with identical tokens is printed, not bounded: on random-init weights the candidates of a beam are
    near-ties throughout (measured: 0 of 15 identical while every score agrees to 6e-2 and the best hypotheses are 1e-2 apart)."""
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    a, task, model, crit, cfg, W = build("s2t_transformer_m", torch.bfloat16, criterion="label_smoothed_cross_entropy")
    sample = batch(task, 3, 1000, 8, 8, 9, lengths=[1000, 731, 402])
    src, lens = sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]
    opts = dict(beam_size=5, max_len_a=0.0, max_len_b=14, min_len=1, len_penalty=1.0, unk_penalty=0.0, temperature=1.0)
    model.eval()
    gen = SequenceGenerator([model], task.target_dictionary, **opts)
    hyps = gen.generate([model], dict(net_input=dict(src_tokens=src.to(DEV), src_lengths=lens.to(DEV))))
    orc = s2t_ref.beam_search(W, cfg, src, lens, 5, 0.0, 14, 1, 1.0, 0.0, 1.0)
    enc, _ = s2t_ref.encoder_forward(W, cfg, src, lens, training=False)
    same = total = 0
    worst_pos = worst_best = 0.0
    for b, (hs, os_) in enumerate(zip(hyps, orc)):
        assert len(hs) == len(os_) == 5
        n = int(enc.src_lengths[b])
        eo = enc.encoder_out[:n, b:b + 1]
        rescored = []
        for h in hs:
            toks = h["tokens"].cpu()
            prev = torch.cat([torch.tensor([2]), toks[:-1]]).view(1, -1)
            lp = torch.log_softmax(s2t_ref.decoder_forward(W, cfg, prev, eo, None).float(), -1)[0]
            pos = lp.gather(1, toks.view(-1, 1)).view(-1)
            worst_pos = max(worst_pos, float((pos - h["positional_scores"].float().cpu()).abs().max()))
            rescored.append(float(pos.sum()) / len(toks))
        for h, (ot, osc, ops) in zip(hs, os_):
            total += 1
            same += int(h["tokens"].tolist() == ot.tolist())
        if hs[0]["tokens"].tolist() != os_[0][0].tolist():
            worst_best = max(worst_best, abs(rescored[0] - os_[0][1]))
    print("MEASURED bf16 beam-5: %d of %d (sentence, rank) slots identical to the f32 oracle; positional scores within %.3e of the "
          "oracle's re-scoring; best-hypothesis near-tie gap %.3e" % (same, total, worst_pos, worst_best))
    tripwire("bf16 beam-5", gen=worst_pos)
    assert worst_pos <= BF16_GEN_ATOL and worst_best <= BF16_GEN_ATOL


BF16_GEN_ATOL = 0.12            # log-probability units (scores are ~ -9 = -ln 8000): 2x the measured 6.0e-2


def test_cfg5_m_two_phase_generation():
    """dual-decoder m model through the two-phase generator (transcript first, then translation) vs the oracle"""
    from fbk_fairseq_st_amd.sequence_generator import TwoPhaseSequenceGenerator
    a, task, model, crit, cfg, W = build("conv_transformer_dualdecoder_big2", torch.float32, criterion="cross_entropy_dualdecoder",
                                         dual=True, encoder_layers=12)
    W["auxiliary_decoder.output_projection.weight"][2] *= 4.0
    model.load_state_dict(W); model.arena.refresh_shadow()
    sample = batch(task, 2, 800, 8, 8, 10, lengths=[800, 366])
    src, lens = sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]
    model.eval()
    gen = TwoPhaseSequenceGenerator([model], task.source_dictionary, task.target_dictionary, beam_size=5, max_len_a=0.0, max_len_b=12, min_len=1)
    hyps = gen.generate([model], dict(net_input=dict(src_tokens=src.to(DEV), src_lengths=lens.to(DEV))))
    orc = s2t_ref.two_phase_beam_search(W, cfg, src, lens, 5, 0.0, 12, 1)
    for hs, os_ in zip(hyps, orc):
        assert len(hs) == len(os_)
        for h, (ot, osc, ops, oa) in zip(hs, os_):
            assert h["tokens"].tolist() == ot.tolist() and h["aux_tokens"].tolist() == oa.tolist()
            assert abs(float(h["score"]) - osc) < 1e-4
