#!/usr/bin/env python3
"""the reference's UNCHANGED command-line mains over the plug-in (build container only: needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_cli_fixture.py [--check]

Three child processes (the two user directories register the same names, so they cannot share one):
  init     reference user dir (examples/speech_recognition): build task / model / criterion / Trainer, save the initial state with the
           reference's own `Trainer.save_checkpoint` -> both runs start from the same weights;
  ref      reference user dir: `fairseq_cli.train.main(args)` (fairseq_cli/train.py:43-120: load_dataset, build, Trainer,
           checkpoint_utils.load_checkpoint, the epoch loop with get_train_iterator / GroupedIterator / progress bar / train_step /
           validate / save_checkpoint) for two epochs of tests/golden/s2t_data, then `fairseq_cli.generate.main(args)`
           (fairseq_cli/generate.py:39-266: load_model_ensemble, task.build_generator, task.inference_step, the BLEU scorer) with beam 5
           over the checkpoint it wrote;
  plugin   the SAME two mains with `--user-dir fbk_fairseq_st_amd`.  No GPU exists here: the engine class the model instantiates is
           tests/cpu_stubs.OracleTrainEngine (oracle/s2t_ref.py under autograd, gradients added into the arena) -- the subject is
           everything ABOVE the engine: registries, task, data path, iterators, criterion, the arena optimizer under the reference's
           trainer, checkpoints, the generator, the hypothesis post-processing.
Dropout: rates 0 on the command line and F.dropout patched to the identity in all three (the reference's subsampler applies
max(p, 0.1), conv_transformer.py:214; same patch as make_golden.py), so the two trajectories are comparable number by number.
Recorded in tests/golden/cli_trajectory.json: per update loss / nll_loss / ctc_loss / gnorm / lr / num_updates of both runs (they must
agree to 2e-4 relative), validation losses, and the beam-5 hypotheses (token strings identical, scores to 1e-4).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "cli_trajectory.json")
DATA = os.path.join(REPO, "tests", "golden", "s2t_data")
USER = {"ref": REF + "/examples/speech_recognition", "init": REF + "/examples/speech_recognition",
        "plugin": os.path.join(REPO, "fbk_fairseq_st_amd")}

MODEL = ["--arch", "conv_transformer", "--no-attn-2d", "--input-feat-per-channel", "80", "--encoder-embed-dim", "32",
         "--decoder-embed-dim", "32", "--decoder-output-dim", "32", "--encoder-ffn-embed-dim", "64", "--decoder-ffn-embed-dim", "64",
         "--encoder-attention-heads", "2", "--decoder-attention-heads", "2", "--encoder-layers", "3", "--decoder-layers", "2",
         "--ctc-compress-out", "--ctc-encoder-layer", "2", "--dropout", "0.0", "--attention-dropout", "0.0", "--relu-dropout", "0.0"]


def train_argv(mode, save_dir, init):
    return [DATA, "--user-dir", USER[mode], "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de"] + MODEL + [
        "--criterion", "ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy", "--label-smoothing", "0.1",
        "--optimizer", "adam", "--adam-betas", "(0.9, 0.98)", "--lr", "2e-3", "--lr-scheduler", "inverse_sqrt", "--warmup-updates", "3",
        "--warmup-init-lr", "5e-4", "--clip-norm", "20", "--weight-decay", "0.0001", "--max-tokens", "160", "--update-freq", "2",
        "--max-source-positions", "80", "--max-target-positions", "50", "--skip-invalid-size-inputs-valid-test",
        "--train-subset", "train", "--valid-subset", "train", "--max-epoch", "3", "--save-dir", save_dir, "--restore-file", init,
        "--reset-optimizer", "--reset-dataloader", "--reset-meters", "--reset-lr-scheduler", "--no-epoch-checkpoints",
        "--num-workers", "0", "--distributed-world-size", "1", "--required-batch-size-multiple", "1", "--seed", "7", "--cpu",
        "--log-format", "simple", "--log-interval", "1", "--no-progress-bar", "--ddp-backend", "no_c10d"]


def generate_argv(mode, ckpt):
    return [DATA, "--user-dir", USER[mode], "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de",
            "--path", ckpt, "--gen-subset", "train", "--beam", "5", "--max-tokens", "400", "--max-source-positions", "80",
            "--max-target-positions", "50", "--skip-invalid-size-inputs-valid-test", "--num-workers", "0", "--cpu",
            "--required-batch-size-multiple", "1", "--max-len-b", "12",
            # the task adds <ctc_blank> to the transcript dictionary only for this criterion (speech_translation_ctc.py:44-45): a
            # checkpoint trained with --ctc-compress-out is generated from with the criterion named again
            "--criterion", "ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy"]


# ---------------------------------------------------------------------------------------------- child process
def shims():
    """everything applied from OUTSIDE the read-only reference tree: numpy aliases, the namedtuple attribute
    python 3.9 dropped, the cython batcher and the BLEU extension built in a scratch directory, an h5py stand-in, weights_only"""
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    for n, t in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, n):
            setattr(np, n, t)
    import argparse
    import importlib.util
    import shutil
    import sysconfig
    import torch
    import torch.nn.functional as F
    torch.set_num_threads(2)
    torch.serialization.add_safe_globals([argparse.Namespace])
    _load = torch.load
    torch.load = lambda *a, **k: _load(*a, **dict(k, weights_only=False))
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    import fairseq.models.fairseq_encoder as fe
    if not hasattr(fe.EncoderOut, "_field_types"):
        fe.EncoderOut._field_types = dict(fe.EncoderOut.__annotations__)
    sys.modules.setdefault("h5py", type(sys)("h5py"))
    tmp = tempfile.mkdtemp(prefix="s2t_cli_shim_")
    inc = sysconfig.get_paths()["include"]
    shutil.copy(REF + "/fairseq/data/data_utils_fast.pyx", tmp)
    subprocess.check_call([sys.executable, "-m", "cython", "-3", os.path.join(tmp, "data_utils_fast.pyx")], stdout=subprocess.DEVNULL)
    so = os.path.join(tmp, "data_utils_fast" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-w", "-I", inc, "-I", np.get_include(),
                           os.path.join(tmp, "data_utils_fast.c"), "-o", so])
    spec = importlib.util.spec_from_file_location("fairseq.data.data_utils_fast", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["fairseq.data.data_utils_fast"] = mod
    so = os.path.join(tmp, "libbleu" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["g++", "-shared", "-fPIC", "-O2", "-w", "-I", inc, REF + "/fairseq/clib/libbleu/libbleu.cpp",
                           REF + "/fairseq/clib/libbleu/module.cpp", "-o", so])
    spec = importlib.util.spec_from_file_location("fairseq.libbleu", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["fairseq.libbleu"] = mod


def child(mode, work):
    import contextlib
    import io
    import logging
    shims()
    import torch
    from fairseq import options
    init = os.path.join(work, "init.pt")
    engine = contextlib.nullcontext()
    if mode == "plugin":
        import cpu_stubs
        engine = cpu_stubs.oracle_engine()
    if mode == "init":
        from fairseq import tasks
        from fairseq.trainer import Trainer
        argv = train_argv(mode, work, init)
        args = options.parse_args_and_arch(options.get_training_parser(), input_args=argv)
        torch.manual_seed(args.seed)
        task = tasks.setup_task(args)
        model, crit = task.build_model(args), task.build_criterion(args)
        tr = Trainer(args, task, model, crit)
        tr.save_checkpoint(init, {"train_iterator": {"epoch": 1, "iterations_in_epoch": 0, "shuffle": True}})
        print(json.dumps({"n_params": sum(p.numel() for p in model.parameters())}))
        return
    from fairseq.trainer import Trainer
    from fairseq_cli import generate as gen_cli, train as train_cli
    updates, valids = [], []
    step = Trainer.train_step

    def recorded(self, samples, raise_oom=False):             # a recorder around the reference's method, not a change of it
        log = step(self, samples, raise_oom)
        from fairseq.logging import metrics
        gn = metrics.get_meter("train_inner", "gnorm") or metrics.get_meter("train", "gnorm")
        updates.append({"num_updates": self.get_num_updates(), "lr": float(self.get_lr()), "micro_batches": len(samples),
                        "gnorm": float(gn.val), **{k: float(log[k]) for k in ("loss", "nll_loss", "ctc_loss", "ctc_acc", "nframes", "wpb", "bsz") if k in log}})
        return log

    Trainer.train_step = recorded
    save_dir = os.path.join(work, "ckpt_" + mode)
    argv = train_argv(mode, save_dir, init)
    with engine:
        args = options.parse_args_and_arch(options.get_training_parser(), input_args=argv)
        if mode == "ref":
            # SURVEY F6 / shim 7: with a single-decoder model the reference's criterion hands `transcript_prev_output_tokens` to
            # an encoder that does not take it (ctc_multi_loss.py:141 -> :23, TypeError); drop the key on the way in, from outside
            from examples.speech_recognition.criterions import ctc_multi_loss as cml
            fwd = cml.CTCMultiLoss.forward

            def forward(self, model, sample, *a, **k):
                ni = {kk: v for kk, v in sample["net_input"].items() if kk != "transcript_prev_output_tokens"}
                return fwd(self, model, dict(sample, net_input=ni), *a, **k)

            cml.CTCMultiLoss.forward = forward
        val = train_cli.validate

        def validate(args_, trainer, task, epoch_itr, subsets):
            out = val(args_, trainer, task, epoch_itr, subsets)
            valids.append([float(v) for v in out])
            return out

        train_cli.validate = validate
        train_cli.main(args)
        ckpt = os.path.join(save_dir, "checkpoint_last.pt")
        saved = sorted(os.listdir(save_dir))
        state = torch.load(ckpt, map_location="cpu")
        # ---- generate.main over the written checkpoint: its printed S-/T-/H-/P- lines are the record
        gargs = options.parse_args_and_arch(options.get_generation_parser(), input_args=generate_argv(mode, ckpt))
        buf = io.StringIO()
        logging.getLogger().handlers[:] = []
        with contextlib.redirect_stdout(buf):
            scorer = gen_cli.main(gargs)
    hyps = {}
    for line in buf.getvalue().splitlines():
        f = line.split("\t")
        if f[0][:2] in ("H-", "P-") and f[0][2:].isdigit():
            e = hyps.setdefault(int(f[0][2:]), {})
            if f[0][0] == "H":
                e["score"], e["tokens"] = float(f[1]), f[2]
            else:
                e["positional"] = [float(v) for v in f[1].split()]
    res = {"updates": updates, "valid_losses": valids, "checkpoints": saved,
           "checkpoint_keys": sorted(state), "optimizer_name": state["optimizer_history"][-1]["optimizer_name"],
           "model_keys": len(state["model"]), "hypotheses": [dict(id=i, **hyps[i]) for i in sorted(hyps)],
           "bleu": scorer.result_string() if scorer is not None and hasattr(scorer, "result_string") else None}
    print("@@RESULT@@" + json.dumps(res))


# ---------------------------------------------------------------------------------------------- parent
def run_child(mode, work):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, work], env=env, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("child %s failed:\n%s\n%s" % (mode, p.stdout[-3000:], p.stderr[-6000:]))
    for line in p.stdout.splitlines():
        if line.startswith("@@RESULT@@"):
            return json.loads(line[len("@@RESULT@@"):])
    return None


def close(a, b, rel):
    """the trainer's logged numbers are rounded to three decimals (fairseq/logging/metrics.py, log_scalar(round=3))"""
    return abs(a - b) <= max(rel * max(abs(a), abs(b)), 1.001e-3)


def run():
    work = tempfile.mkdtemp(prefix="s2t_cli_")
    run_child("init", work)
    ref, plug = run_child("ref", work), run_child("plugin", work)
    assert len(ref["updates"]) == len(plug["updates"]) > 0, (len(ref["updates"]), len(plug["updates"]))
    worst = 0.0
    for r, p in zip(ref["updates"], plug["updates"]):
        assert (r["num_updates"], r["micro_batches"]) == (p["num_updates"], p["micro_batches"]), (r, p)
        assert abs(r["lr"] - p["lr"]) < 1e-12, (r, p)
        assert sorted(r) == sorted(p), (sorted(r), sorted(p))
        for k in r:
            if k not in ("num_updates", "micro_batches", "lr"):
                assert close(r[k], p[k], 2e-4), (k, r, p)
            worst = max(worst, abs(r[k] - p[k]) / max(abs(r[k]), 1e-6))
    assert len(ref["valid_losses"]) == len(plug["valid_losses"]) > 0
    for a, b in zip(ref["valid_losses"], plug["valid_losses"]):
        assert all(close(x, y, 2e-4) for x, y in zip(a, b)), (a, b)
    assert ref["checkpoints"] == plug["checkpoints"] and ref["checkpoint_keys"] == plug["checkpoint_keys"]
    assert ref["optimizer_name"] == plug["optimizer_name"] == "FairseqAdam"
    assert ref["model_keys"] == plug["model_keys"]
    assert len(ref["hypotheses"]) == len(plug["hypotheses"]) > 0
    hworst = 0.0
    for r, p in zip(ref["hypotheses"], plug["hypotheses"]):
        assert r["id"] == p["id"] and r["tokens"] == p["tokens"], (r, p)
        hworst = max(hworst, abs(r["score"] - p["score"]), max(abs(x - y) for x, y in zip(r["positional"], p["positional"])))
    assert hworst < 1e-4, hworst
    assert ref["bleu"] == plug["bleu"], (ref["bleu"], plug["bleu"])
    rnd = lambda u: {k: (round(v, 4) if isinstance(v, float) and k != "lr" else v) for k, v in u.items()}
    return {"reference_updates": [rnd(u) for u in ref["updates"]],
            "updates_agree_rel": "< 2e-4", "n_updates": len(ref["updates"]),
            "valid_losses": [[round(v, 4) for v in a] for a in ref["valid_losses"]],
            "checkpoints": ref["checkpoints"], "checkpoint_keys": ref["checkpoint_keys"], "optimizer_name": ref["optimizer_name"],
            "hypotheses": [{"id": h["id"], "tokens": h["tokens"], "score": round(h["score"], 3)} for h in ref["hypotheses"]],
            "hypotheses_agree": "tokens identical, scores and positional scores < 1e-4", "bleu": ref["bleu"]}


if __name__ == "__main__":
    if "--child" in sys.argv:
        i = sys.argv.index("--child")
        child(sys.argv[i + 1], sys.argv[i + 2])
        sys.exit(0)
    res = run()
    txt = json.dumps(res, indent=1, sort_keys=True)
    if "--check" in sys.argv:
        with open(OUT) as f:
            assert json.load(f) == json.loads(txt), "cli_trajectory.json is stale:\n" + txt
        print("CLI fixture up to date")
    else:
        with open(OUT, "w") as f:
            f.write(txt + "\n")
        print("wrote", OUT)
