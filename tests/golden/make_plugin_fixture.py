#!/usr/bin/env python3
"""Drive the REAL reference's CLI argument flow with this package as the `--user-dir` (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_plugin_fixture.py [--check]

What runs is fairseq's own code, unchanged (fairseq_cli/train.py:43-75 up to the model / criterion construction):
`options.get_training_parser()` -> `options.parse_args_and_arch(parser, argv)` (which imports the user directory,
fairseq/options.py:117-120 -> fairseq/utils.py:344-359) -> `tasks.setup_task(args)` -> `task.build_model(args)` ->
`task.build_criterion(args)`.  The user directory is /root/repo/fbk_fairseq_st_amd instead of
examples/speech_recognition; nothing else differs from the reference's README command line (README.md:138-163).
The outcome (which registries hold what, the parsed-and-arch-applied namespace, class ancestry, parameter inventory)
is written to tests/golden/plugin_boundary.json; tests/test_plugin_boundary_cpu.py checks the package's standalone
registries against it on every run and re-runs this script when /root/reference is present.
Our own glue only; the shims are the numpy aliases of SURVEY.md 8-c (#1, #2).
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "plugin_boundary.json")
USER_DIR = os.path.join(REPO, "fbk_fairseq_st_amd")
DATA = os.path.join(REPO, "tests", "golden", "s2t_data")

CASES = {
    # the Cfg3 command line (BASELINE.json configs[2]) in the reference's own CLI vocabulary
    "m_ctc": ["--arch", "s2t_transformer_m", "--criterion", "ctc_multi_loss", "--underlying-criterion", "label_smoothed_cross_entropy",
              "--label-smoothing", "0.1", "--ctc-compress-out", "--ctc-encoder-layer", "8", "--ctc-compress-strategy", "avg"],
    # the paper script's architecture name and flags (README.md:138-163)
    "big2_paper": ["--arch", "conv_transformer_big2", "--criterion", "ctc_multi_loss", "--underlying-criterion",
                   "label_smoothed_cross_entropy", "--label-smoothing", "0.1", "--ctc-compress-out", "--ctc-encoder-layer", "8",
                   "--distance-penalty", "log", "--no-attn-2d", "--encoder-layers", "11", "--decoder-layers", "4"],
    "l_kd": ["--arch", "s2t_transformer_l", "--criterion", "knowledge_distillation", "--kd-lambda", "0.5", "--kd-temperature", "2.0"],
    "dual": ["--arch", "conv_transformer_dualdecoder_big2", "--no-attn-2d", "--criterion", "cross_entropy_dualdecoder",
             "--label-smoothing", "0.1", "--auxiliary-loss-weight", "0.3"],
}
COMMON = [DATA, "--task", "speech_translation_with_transcription", "-s", "en", "-t", "de", "--max-tokens", "12000",
          "--optimizer", "adam", "--lr", "5e-3", "--lr-scheduler", "inverse_sqrt", "--warmup-updates", "4000", "--clip-norm", "20",
          "--update-freq", "8", "--skip-invalid-size-inputs-valid-test", "--max-source-positions", "2000", "--max-target-positions", "1000"]
ARG_KEYS = ("arch", "task", "criterion", "underlying_criterion", "encoder_embed_dim", "encoder_ffn_embed_dim", "encoder_attention_heads",
            "encoder_layers", "decoder_layers", "decoder_embed_dim", "decoder_ffn_embed_dim", "decoder_attention_heads", "dropout",
            "attention_dropout", "activation_dropout", "relu_dropout", "encoder_convolutions", "attn_2d", "no_attn_2d", "distance_penalty",
            "ctc_compress_out", "ctc_encoder_layer", "ctc_compress_strategy", "label_smoothing", "input_feat_per_channel",
            "encoder_normalize_before", "decoder_normalize_before", "share_decoder_input_output_embed", "activation_fn",
            "kd_lambda", "kd_temperature", "auxiliary_loss_weight", "primary_loss_weight", "max_source_positions", "update_freq")


def run():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for n, t in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, n):
            setattr(np, n, t)
    import fairseq.models.fairseq_encoder as fe
    if not hasattr(fe.EncoderOut, "_field_types"):
        fe.EncoderOut._field_types = dict(fe.EncoderOut.__annotations__)
    import fairseq
    from fairseq import options, tasks
    import fairseq.criterions as fcrit
    import fairseq.models as fmodels
    core_lsce = fcrit.CRITERION_REGISTRY["label_smoothed_cross_entropy"]
    out = {"fairseq_version": fairseq.__version__, "cases": {}}
    for name, flags in CASES.items():
        argv = COMMON + ["--user-dir", USER_DIR] + flags
        sys.argv = ["train.py"] + argv                      # get_parser() reads --user-dir from sys.argv (options.py:203-208)
        parser = options.get_training_parser()
        args = options.parse_args_and_arch(parser, input_args=argv)
        import fbk_fairseq_st_amd.registry as R
        assert R.inside_fairseq(), "the package must have bound to fairseq's registries"
        task = tasks.setup_task(args)
        model = task.build_model(args)
        criterion = task.build_criterion(args)
        sd = model.state_dict()
        out["cases"][name] = {
            "args": {k: (getattr(args, k) if not isinstance(getattr(args, k, None), (list, tuple)) else list(getattr(args, k)))
                     for k in ARG_KEYS if hasattr(args, k)},
            "task_class": type(task).__module__ + "." + type(task).__name__,
            "model_class": type(model).__module__ + "." + type(model).__name__,
            "criterion_class": type(criterion).__module__ + "." + type(criterion).__name__,
            "model_is_fairseq_model": isinstance(model, fmodels.BaseFairseqModel) and isinstance(model, fmodels.FairseqEncoderDecoderModel),
            "encoder_is_fairseq_encoder": isinstance(model.encoder, fmodels.FairseqEncoder),
            "decoder_is_incremental": isinstance(model.decoder, fmodels.FairseqIncrementalDecoder),
            "criterion_is_fairseq_criterion": isinstance(criterion, fcrit.FairseqCriterion),
            "task_is_fairseq_task": isinstance(task, tasks.FairseqTask),
            "n_params": int(sum(p.numel() for n, p in model.named_arena_params().items())),
            "n_state_keys": len(sd),
            "src_dict": len(task.source_dictionary), "tgt_dict": len(task.target_dictionary),
            "max_positions": list(model.max_positions()),
        }
    import fbk_fairseq_st_amd.registry as R
    ours = sorted(a for a, c in fmodels.ARCH_MODEL_REGISTRY.items() if c.__module__.startswith("fbk_fairseq_st_amd"))
    out["archs_registered_in_fairseq"] = ours
    out["tasks_registered_in_fairseq"] = sorted(t for t, c in tasks.TASK_REGISTRY.items() if c.__module__.startswith("fbk_fairseq_st_amd"))
    out["criteria_registered_in_fairseq"] = sorted(n for n, c in fcrit.CRITERION_REGISTRY.items() if c.__module__.startswith("fbk_fairseq_st_amd"))
    out["core_lsce_replaced"] = fcrit.CRITERION_REGISTRY["label_smoothed_cross_entropy"] is not core_lsce
    out["registries_are_fairseqs"] = R.ARCH_MODEL_REGISTRY is fmodels.ARCH_MODEL_REGISTRY and R.TASK_REGISTRY is tasks.TASK_REGISTRY
    # error behaviour of the delegated decorators (fairseq/models/__init__.py:70-79)
    try:
        R.register_model("conv_transformer")(fmodels.MODEL_REGISTRY["conv_transformer"])
        out["duplicate_model_raises"] = False
    except ValueError:
        out["duplicate_model_raises"] = True
    return out


if __name__ == "__main__":
    res = run()
    txt = json.dumps(res, indent=1, sort_keys=True)
    if "--check" in sys.argv:
        with open(OUT) as f:
            assert json.load(f) == json.loads(txt), "plugin_boundary.json is stale"
        print("plugin boundary fixture up to date")
    else:
        with open(OUT, "w") as f:
            f.write(txt + "\n")
        print("wrote", OUT)
