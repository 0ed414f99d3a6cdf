#!/usr/bin/env python3
"""The reference's OWN `Trainer` and `SequenceGenerator` driving the plug-in (build container only: needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_trainer_fixture.py [--check]

What runs unchanged from the reference: `options.parse_args_and_arch` with `--user-dir fbk_fairseq_st_amd`, `tasks.setup_task`,
`task.build_model / build_criterion`, `fairseq.trainer.Trainer(args, task, model, criterion)` and its `train_step`
(fairseq/trainer.py:334-495: zero_grad -> task.train_step -> multiply_grads -> clip_grad_norm -> optimizer.step -> lr schedule),
`Trainer.save_checkpoint / load_checkpoint` (fairseq/checkpoint_utils.py:245-286), and `fairseq.sequence_generator.SequenceGenerator`
(:21-650) over the plug-in's encoder / incremental decoder.  There is no GPU here, so the HIP side is replaced by the CPU stand-ins
of tests/cpu_stubs.py: a toy differentiable engine for the training step, the oracle as engine for generation.

Checked here (and recorded in tests/golden/reference_trainer.json for tests/test_reference_trainer_cpu.py):
  * fairseq's optimizer registry resolves `--optimizer adam` to this package's arena Adam, and after the reference's
    `zero_grad()` every `p.grad` still aliases the arena;
  * three updates (one of them with two micro-batches) move the arena master exactly as clip + Adam + inverse-sqrt schedule of
    the oracle predict from independently computed gradients (1e-6), nothing else in the arena moves;
  * a checkpoint written by the reference's trainer resumes in a fresh trainer to the same fourth update, and its
    `last_optimizer_state` is the reference's layout (torch.optim state dict over the reference's parameter order);
  * the reference's SequenceGenerator over the plug-in reproduces tests/golden/generate.npz (tokens exact, scores 1e-4).
"""
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "reference_trainer.json")
USER_DIR = os.path.join(REPO, "fbk_fairseq_st_amd")
DATA = os.path.join(REPO, "tests", "golden", "s2t_data")

ARGV = [DATA, "--user-dir", USER_DIR, "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de",
        "--arch", "s2t_transformer_xs", "--encoder-embed-dim", "32", "--decoder-embed-dim", "32", "--encoder-ffn-embed-dim", "64",
        "--decoder-ffn-embed-dim", "64", "--encoder-attention-heads", "2", "--decoder-attention-heads", "2", "--encoder-layers", "2",
        "--decoder-layers", "1", "--criterion", "label_smoothed_cross_entropy", "--label-smoothing", "0.1",
        "--optimizer", "adam", "--adam-betas", "(0.9, 0.98)", "--lr", "5e-3", "--lr-scheduler", "inverse_sqrt", "--warmup-updates", "2",
        "--warmup-init-lr", "1e-3", "--clip-norm", "0.05", "--weight-decay", "0.01", "--max-tokens", "4000", "--cpu", "--seed", "3",
        "--max-source-positions", "2000", "--max-target-positions", "1000"]


def shims():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    for n, t in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, n):
            setattr(np, n, t)
    import argparse
    import torch
    torch.serialization.add_safe_globals([argparse.Namespace])      # torch >= 2.6 loads weights-only by default; fairseq stores `args`
    import fairseq.models.fairseq_encoder as fe
    if not hasattr(fe.EncoderOut, "_field_types"):
        fe.EncoderOut._field_types = dict(fe.EncoderOut.__annotations__)


def sample(task, seed, B=3, T=37, L=6):
    import torch
    g = torch.Generator().manual_seed(seed)
    V = len(task.target_dictionary)
    tgt = torch.randint(4, V, (B, L), generator=g)
    tgt[:, -1] = 2
    prev = torch.cat([torch.full((B, 1), 2, dtype=torch.long), tgt[:, :-1]], 1)
    return {"id": torch.arange(B), "nsentences": B, "ntokens": B * L, "target": tgt,
            "net_input": {"src_tokens": torch.randn(B, T, 80, generator=g), "src_lengths": torch.full((B,), T, dtype=torch.long),
                          "prev_output_tokens": prev}}


def expected_update(W, state, samples, step, args):
    """clip + Adam + schedule of the oracle on gradients computed independently of the plug-in (autograd over cpu_stubs.toy_forward)"""
    import torch
    from oracle import s2t_ref
    import cpu_stubs
    names = ("encoder.fc3.weight", "decoder.embed_tokens.weight", "decoder.output_projection.weight")
    Wg = {k: v.clone().requires_grad_(k in names) for k, v in W.items()}
    total, ss, losses = 0.0, 0, []
    for s in samples:
        _, logits = cpu_stubs.toy_forward(Wg, s["net_input"]["src_tokens"], s["net_input"]["prev_output_tokens"])
        loss, _ = s2t_ref.label_smoothed_nll(logits.transpose(0, 1), s["target"], 0.1, 1)
        total = total + loss
        ss += s["ntokens"]
        losses.append(float(loss))
    total.backward()
    keys = list(W)
    grads = [(Wg[k].grad if Wg[k].grad is not None else torch.zeros_like(W[k])) / float(ss) for k in keys]
    gnorm, grads = s2t_ref.clip_grad_norm(grads, args.clip_norm)
    lr = s2t_ref.inverse_sqrt_lr(step - 1, args.lr[0], args.warmup_updates, args.warmup_init_lr)
    out = {}
    for k, g in zip(keys, grads):
        m, v = state.setdefault(k, (torch.zeros_like(W[k]), torch.zeros_like(W[k])))
        p, m, v = s2t_ref.adam_step(W[k], g, m, v, step, lr, 0.9, 0.98, args.adam_eps, args.weight_decay)
        out[k], state[k] = p, (m, v)
    return out, float(gnorm), lr, losses


def build(argv):
    from fairseq import options, tasks
    sys.argv = ["train.py"] + argv
    args = options.parse_args_and_arch(options.get_training_parser(), input_args=argv)
    task = tasks.setup_task(args)
    import torch
    torch.manual_seed(args.seed)
    model, crit = task.build_model(args), task.build_criterion(args)
    return args, task, model, crit


def trainer_flow():
    import torch
    import fairseq.optim as fopt
    from fairseq.logging import metrics
    from fairseq.trainer import Trainer
    import cpu_stubs
    import fbk_fairseq_st_amd.fairseq_optim as our_optim
    out = {}
    args, task, model, crit = build(ARGV)
    tr = Trainer(args, task, model, crit)
    with cpu_stubs.cpu_kernels():
        opt = tr.optimizer                                   # fairseq/trainer.py:140-170 -> optim.build_optimizer -> the registry
        out["adam_is_the_arena_adam"] = type(opt) is our_optim.FairseqAdam and fopt.OPTIMIZER_REGISTRY["adam"] is our_optim.FairseqAdam
        out["optimizer_class_name"] = type(opt).__name__
        out["is_fairseq_optimizer"] = isinstance(opt, fopt.FairseqOptimizer)
        arena = model.arena
        assert arena is not None and arena.device.type == "cpu"
        model.engine = cpu_stubs.ToyEngine(model)
        named = model.named_arena_params()
        W = {n: arena.p(n).detach().clone() for n in arena.slices}
        state = {}
        plan = [[sample(task, 10)], [sample(task, 11), sample(task, 12)], [sample(task, 13)]]
        out["updates"] = []
        for i, batch in enumerate(plan):
            log = tr.train_step(batch)
            got_gnorm = float(metrics.get_meter("train", "gnorm").val)       # what the reference's trainer logged (trainer.py:779)
            aliased = all(p.grad is not None and p.grad.data_ptr() == arena.g(n).data_ptr() for n, p in named.items())
            W, gnorm, lr, losses = expected_update(W, state, batch, i + 1, args)
            worst = max(float((arena.p(n) - W[n]).abs().max()) for n in arena.slices)
            out["updates"].append({"num_updates": tr.get_num_updates(), "lr_after": round(tr.get_lr(), 10),
                                   "gnorm": round(got_gnorm, 6), "logged_keys": sorted(log), "gnorm_expected": round(gnorm, 6),
                                   "loss_sum": round(sum(losses), 4), "grad_still_aliases_arena": aliased,
                                   "master_max_abs_diff_vs_oracle": worst})
            assert aliased, "p.grad was severed from the arena"
            assert worst < 1e-6, (i, worst)
            assert abs(got_gnorm - gnorm) < 1e-5 * max(gnorm, 1.0), (got_gnorm, gnorm)
        # ---- checkpoint through the reference's own save / load, then a fourth update in a fresh trainer
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "checkpoint_last.pt")
            tr.save_checkpoint(path, {"train_iterator": {"epoch": 1}})
            ck = torch.load(path, map_location="cpu", weights_only=False)
            last = ck["last_optimizer_state"]
            ref_names = model.reference_parameter_names()
            out["checkpoint"] = {"optimizer_name": ck["optimizer_history"][-1]["optimizer_name"],
                                 "n_state_entries": len(last["state"]), "n_reference_params": len(ref_names),
                                 "param_group_keys": sorted(last["param_groups"][0]),
                                 "steps": sorted({int(s["step"]) for s in last["state"].values()})}
            i_q = ref_names.index("encoder.layers.0.self_attn.q_proj.weight")
            out["checkpoint"]["q_proj_state_shape"] = list(last["state"][i_q]["exp_avg"].shape)
            b4 = [sample(task, 14)]
            tr.train_step(b4)
            after4 = {n: arena.p(n).detach().clone() for n in arena.slices}
            args2, task2, model2, crit2 = build(ARGV)
            tr2 = Trainer(args2, task2, model2, crit2)
            _ = tr2.optimizer
            model2.engine = cpu_stubs.ToyEngine(model2)
            extra = tr2.load_checkpoint(path)
            out["checkpoint"]["extra_state_round_trip"] = extra["train_iterator"] == {"epoch": 1}
            out["checkpoint"]["num_updates_restored"] = tr2.get_num_updates()
            tr2.train_step(b4)
            worst = max(float((model2.arena.p(n) - after4[n]).abs().max()) for n in arena.slices)
            out["checkpoint"]["resumed_update_max_abs_diff"] = worst
            assert worst < 1e-7, worst
    return out


def generator_flow():
    """the reference's SequenceGenerator (beam 5 / beam 3 cases of generate.npz) over the plug-in model on the oracle engine"""
    import torch
    from fairseq import options, tasks
    from fairseq.sequence_generator import SequenceGenerator
    import cpu_stubs
    from helpers import generate_case
    res = {}
    for tag in ("a", "b"):
        cfg, W, src, lens, opts, exp, meta = generate_case(tag)
        argv = [DATA, "--user-dir", USER_DIR, "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de",
                "--arch", "conv_transformer", "--no-attn-2d", "--encoder-embed-dim", str(cfg["D"]), "--decoder-embed-dim", str(cfg["D"]),
                "--encoder-ffn-embed-dim", str(cfg["ffn"]), "--decoder-ffn-embed-dim", str(cfg["ffn"]),
                "--encoder-attention-heads", str(cfg["heads"]), "--decoder-attention-heads", str(cfg["heads"]),
                "--encoder-layers", str(cfg["enc_layers"]), "--decoder-layers", str(cfg["dec_layers"]),
                "--criterion", "label_smoothed_cross_entropy", "--max-tokens", "4000", "--cpu",
                "--max-source-positions", "2000", "--max-target-positions", "1000"]
        sys.argv = ["train.py"] + argv
        args = options.parse_args_and_arch(options.get_training_parser(), input_args=argv)
        task = tasks.setup_task(args)
        # the fixture's dictionaries are synthetic (V_tgt symbols): only their sizes matter to the generator
        from fbk_fairseq_st_amd.data import Dictionary
        task.tgt_dict, task.src_dict = Dictionary.synthetic(meta["V_tgt"] - 4), Dictionary.synthetic(meta["V_src"] - 4)
        model = task.build_model(args)
        model.materialize("cpu", torch.float32)
        model.engine = cpu_stubs.OracleEngine(model, W, cfg)
        model.eval()
        gen = SequenceGenerator([model], task.target_dictionary, beam_size=opts["beam_size"], max_len_a=opts["max_len_a"],
                                max_len_b=opts["max_len_b"], min_len=opts["min_len"], len_penalty=opts["len_penalty"],
                                unk_penalty=opts["unk_penalty"], temperature=opts["temperature"])
        with cpu_stubs.cpu_kernels(), torch.no_grad():
            hyps = gen.generate([model], {"net_input": {"src_tokens": src, "src_lengths": lens}})
        worst, n = 0.0, 0
        for hs, es in zip(hyps, exp):
            assert len(hs) == len(es), (len(hs), len(es))
            for h, (t, sc, ps) in zip(hs, es):
                assert h["tokens"].tolist() == t.tolist(), (tag, h["tokens"].tolist(), t.tolist())
                worst = max(worst, abs(float(h["score"]) - sc), float(np.abs(h["positional_scores"].numpy() - ps).max()))
                n += 1
        assert worst < 1e-4, worst
        res[tag] = {"hypotheses": n, "tokens_identical": True, "score_max_abs_diff_below": 1e-4}
    return res


def run():
    shims()
    out = {"trainer": trainer_flow(), "reference_sequence_generator": generator_flow()}
    for u in out["trainer"]["updates"]:                      # keep the fixture free of last-digit noise
        u["master_max_abs_diff_vs_oracle"] = "< 1e-6"
    out["trainer"]["checkpoint"]["resumed_update_max_abs_diff"] = "< 1e-7"
    return out


if __name__ == "__main__":
    res = run()
    txt = json.dumps(res, indent=1, sort_keys=True)
    if "--check" in sys.argv:
        with open(OUT) as f:
            assert json.load(f) == json.loads(txt), "reference_trainer.json is stale:\n" + txt
        print("reference trainer fixture up to date")
    else:
        with open(OUT, "w") as f:
            f.write(txt + "\n")
        print("wrote", OUT)
