#!/usr/bin/env python3
"""Parameter order of the REFERENCE's models (build container only), and a reference checkpoint crossing over.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_param_order_fixture.py [--check]

(1) `[n for n, _ in model.named_parameters()]` (+ the criterion's) of the real `conv_transformer` / dual-decoder models for the
    structural variants this package builds -> tests/golden/param_order.json.  That order is the meaning of the integer keys of
    `last_optimizer_state` in the reference's checkpoints (fairseq/trainer.py:140-146, torch.optim state dicts);
    `ConvolutionalTransformerModel.reference_parameter_names()` restates it and tests/test_reference_trainer_cpu.py holds it to
    this fixture.
(2) cross-over, asserted here (nothing is stored): the reference's own Trainer trains the reference's model for two updates on the
    CPU and saves a checkpoint through fairseq/checkpoint_utils.py; this package's standalone Trainer loads that file -- weights,
    criterion head, update counter AND per-parameter Adam moments (split q/k/v re-fused into the arena) -- writes it back, and the
    reference's trainer loads the written file into a fresh reference model: every weight and every moment tensor identical.

Our own glue only (it reuses the model builder of make_golden.py and its shims).
"""
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, "tests", "golden", "param_order.json")
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
sys.path.insert(0, os.path.join(REPO, "tests"))

VARIANTS = {
    # name: (builder kwargs of make_golden.build, hp overrides understood by tests/test_reference_trainer_cpu.py)
    "ctc_compress": dict(compress=True, attn_2d=False, extra=[]),
    "plain": dict(compress=False, attn_2d=False, extra=[]),
    "attn2d_distance_penalty": dict(compress=False, attn_2d=True, extra=["--distance-penalty", "log"]),
    "shared_embeddings": dict(compress=False, attn_2d=False, extra=["--share-decoder-input-output-embed"]),
    "dual_decoder": dict(compress=False, attn_2d=False, extra=[], arch="conv_transformer_dualdecoder",
                         criterion=("cross_entropy_dualdecoder", "--label-smoothing", "0.1")),
}


def orders(MG):
    out = {}
    for name, kw in VARIANTS.items():
        kw = dict(kw)
        args, task, model, crit, Vs, Vt = MG.build(name, 32, 2, 64, 2, 1, 1, **kw)
        out[name] = {"model": [n for n, _ in model.named_parameters()], "criterion": [n for n, _ in crit.named_parameters()]}
    return out


XARGS = ["--optimizer", "adam", "--adam-betas", "(0.9, 0.98)", "--lr", "1e-3", "--lr-scheduler", "inverse_sqrt",
         "--warmup-updates", "10", "--clip-norm", "5", "--seed", "5"]
DIMS = (32, 2, 64, 2, 1, 1)              # D, heads, ffn, encoder layers, decoder layers, ctc layer


def _ref_trainer(MG):
    import argparse
    import torch
    torch.serialization.add_safe_globals([argparse.Namespace])          # torch >= 2.6 vs fairseq's pickled `args`
    from fairseq.trainer import Trainer as RefTrainer
    args, task, model, crit, Vs, Vt = MG.build("x", *DIMS, compress=True, extra=XARGS)
    return model, crit, RefTrainer(args, task, model, crit), Vs, Vt


def phase_ref_write(d):
    """reference process: two real CPU updates of the reference's model under the reference's trainer, checkpoint -> d/ref.pt"""
    sys.argv = [sys.argv[0]]
    import make_golden as MG
    model, crit, tr, Vs, Vt = _ref_trainer(MG)
    s = MG.to_ref_sample(MG.make_sample(3, [80, 64, 41], [6, 5, 4], [5, 4, 4], Vs, Vt, Vs - 1))
    s["net_input"].pop("transcript_prev_output_tokens", None)           # single-decoder model
    for _ in range(2):
        tr.train_step([s])
    tr.save_checkpoint(os.path.join(d, "ref.pt"), {"train_iterator": {"epoch": 1}})
    with open(os.path.join(d, "names.json"), "w") as f:
        json.dump({"names": [n for n, _ in model.named_parameters()] + [n for n, _ in crit.named_parameters()], "Vs": Vs, "Vt": Vt}, f)


def phase_ours(d):
    """a process WITHOUT fairseq: this package's standalone Trainer (CPU stand-ins for the two optimizer kernels) loads d/ref.pt,
    checks weights + moments against the file, writes d/ours.pt"""
    import argparse
    import torch
    sys.path.insert(0, REPO)
    assert "fairseq" not in sys.modules
    import cpu_stubs
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.data import Dictionary
    from fbk_fairseq_st_amd.registry import apply_arch, inside_fairseq, namespace
    from fbk_fairseq_st_amd.trainer import Trainer
    assert not inside_fairseq()
    meta = json.load(open(os.path.join(d, "names.json")))
    D, H, Ff, EL, DL, ctc_layer = DIMS
    a = namespace(arch="conv_transformer", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                  label_smoothing=0.1, ctc_compress_out=True, ctc_encoder_layer=ctc_layer, ctc_weight=1.0, encoder_embed_dim=D,
                  encoder_ffn_embed_dim=Ff, encoder_attention_heads=H, encoder_layers=EL, decoder_layers=DL, decoder_embed_dim=D,
                  decoder_ffn_embed_dim=Ff, decoder_attention_heads=H, no_attn_2d=True, input_feat_per_channel=80,
                  lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=5.0, warmup_updates=10, seed=5)
    apply_arch(a)
    tgt, src = Dictionary.synthetic(meta["Vt"] - 4), Dictionary.synthetic(meta["Vs"] - 5)
    src.add_symbol("<ctc_blank>")
    t2 = tasks.SpeechTranslationCTCTask(a, tgt, src)
    m2, c2 = t2.build_model(a), t2.build_criterion(a)
    # the file holds the reference's pickled `args` (an argparse.Namespace) and nothing else of fairseq
    ref_state = torch.load(os.path.join(d, "ref.pt"), map_location="cpu", weights_only=False)
    with cpu_stubs.cpu_kernels():
        ours = Trainer(a, t2, m2, c2, device="cpu", compute_dtype=torch.float32)
        extra_state = ours.load_checkpoint(os.path.join(d, "ref.pt"))
        assert extra_state["train_iterator"] == {"epoch": 1} and ours.num_updates == 2
        names = ours.optimizer_parameter_names()
        assert names == meta["names"], "parameter order differs from the reference's"
        last = ref_state["last_optimizer_state"]
        for i in sorted(last["state"]):                                  # every moment the reference saved sits in the arena
            mv, vv = ours.optimizer._moment_views(names[i])
            assert torch.equal(mv, last["state"][i]["exp_avg"].float()) and torch.equal(vv, last["state"][i]["exp_avg_sq"].float()), names[i]
        sd = m2.state_dict()
        for k, v in ref_state["model"].items():
            if v.dtype.is_floating_point and "_float_tensor" not in k and "version" not in k:
                assert torch.equal(sd[k].float().reshape(v.shape), v.float()), k
        assert ours.optimizer.step_count == 2
        ours.save_checkpoint(os.path.join(d, "ours.pt"), extra_state)


def phase_ref_read(d):
    """reference process again: the reference's trainer loads d/ours.pt into a fresh reference model"""
    sys.argv = [sys.argv[0]]
    import torch
    import make_golden as MG
    model, crit, tr, _, _ = _ref_trainer(MG)
    ref_state = torch.load(os.path.join(d, "ref.pt"), map_location="cpu", weights_only=False)
    es = tr.load_checkpoint(os.path.join(d, "ours.pt"))
    assert es["train_iterator"] == {"epoch": 1} and tr.get_num_updates() == 2
    back, last = tr.optimizer.state_dict(), ref_state["last_optimizer_state"]
    names = json.load(open(os.path.join(d, "names.json")))["names"]
    n_checked = 0
    for i in sorted(last["state"]):
        assert torch.equal(back["state"][i]["exp_avg"], last["state"][i]["exp_avg"]), names[i]
        assert torch.equal(back["state"][i]["exp_avg_sq"], last["state"][i]["exp_avg_sq"]), names[i]
        assert int(back["state"][i]["step"]) == 2
        n_checked += 1
    sd = model.state_dict()
    for k, v in ref_state["model"].items():
        if "_float_tensor" not in k:                                     # uninitialised one-element placeholders of the sinusoidal tables
            assert torch.equal(sd[k], v), k
    with open(os.path.join(d, "result.json"), "w") as f:
        json.dump({"parameters_with_adam_state": n_checked, "parameters": len(names), "num_updates": 2,
                   "weights_identical_both_ways": True, "moments_identical_both_ways": True}, f)


def crossover():
    import subprocess
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    with tempfile.TemporaryDirectory() as d:
        for ph in ("ref_write", "ours", "ref_read"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--phase", ph, d], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=900)
            assert r.returncode == 0, "phase %s failed:\n%s" % (ph, r.stdout[-3000:])
        return json.load(open(os.path.join(d, "result.json")))


def run():
    sys.argv = [sys.argv[0]]
    import make_golden as MG
    return {"orders": orders(MG), "checkpoint_crossover": crossover()}


if __name__ == "__main__":
    if "--phase" in sys.argv:
        ph, d = sys.argv[sys.argv.index("--phase") + 1:][:2]
        {"ref_write": phase_ref_write, "ours": phase_ours, "ref_read": phase_ref_read}[ph](d)
        sys.exit(0)
    check = "--check" in sys.argv
    res = run()
    txt = json.dumps(res, indent=1, sort_keys=True)
    if check:
        with open(OUT) as f:
            assert json.load(f) == json.loads(txt), "param_order.json is stale"
        print("parameter-order fixture up to date; checkpoint cross-over ok")
    else:
        with open(OUT, "w") as f:
            f.write(txt + "\n")
        print("wrote", OUT)
