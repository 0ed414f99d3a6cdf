"""Trainer-level checks on the GPU: a few updates reduce the loss, the update matches the step-by-step path,
and the synthetic benchmark task runs in bf16."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dtype, batch=4, frames=200, arch="s2t_transformer_xs", dropout=None):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.registry import apply_arch, namespace, setup_task
    from fbk_fairseq_st_amd.trainer import Trainer
    a = namespace(arch=arch, task="dummy_s2t", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                  label_smoothing=0.1, sentence_avg=False, ctc_compress_out=True, ctc_encoder_layer=4, ctc_weight=1.0,
                  input_feat_per_channel=80, no_attn_2d=True, dict_size=96, src_dict_size=59, batch_size=batch, frames=frames,
                  tgt_len=12, transcript_len=10, lr=[1e-3], adam_betas="(0.9, 0.98)", clip_norm=20.0, warmup_updates=1,
                  warmup_init_lr=1e-3, seed=3, encoder_layers=4, decoder_layers=2)
    if dropout is not None:
        a.dropout = a.attention_dropout = a.activation_dropout = a.relu_dropout = dropout
    apply_arch(a)
    task = setup_task(a)
    torch.manual_seed(0)
    model, crit = task.build_model(a), task.build_criterion(a)
    tr = Trainer(a, task, model, crit, device="cuda:0", compute_dtype=dtype)
    return a, task, model, crit, tr


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_reduces_loss(dtype):
    a, task, model, crit, tr = _setup(dtype, dropout=0.0)
    sample = tr.prepare(task.dummy_batch(seed=1, lengths=[200, 180, 150, 120]))
    losses = []
    for it in range(8):
        tr.train_step([sample])
        st = tr.reduce_stats()
        losses.append(st["loss"])
        assert np.isfinite(st["gnorm"]) and st["nframes"] == 650
    assert losses[-1] < 0.9 * losses[0], losses


def test_update_freq_accumulates_gradients():
    """two micro-batches in one update == gradient sum (trainer.py:347-411 semantics)"""
    a, task, model, crit, tr = _setup(torch.float32, dropout=0.0)
    model.hp.sub_dropout = 0.0            # the subsampler's rate is max(dropout, 0.1) (conv_transformer.py:214)
    s1 = tr.prepare(task.dummy_batch(seed=1)); s2 = tr.prepare(task.dummy_batch(seed=2))
    model.train(); crit.train()
    tr.optimizer.zero_grad()
    l1, ss1, _ = crit(model, s1); l1.backward()
    g1 = model.arena.grad.clone()
    tr.optimizer.zero_grad()
    l2, ss2, _ = crit(model, s2); l2.backward()
    g2 = model.arena.grad.clone()
    tr.optimizer.zero_grad()
    for s in (s1, s2):
        l, _, _ = crit(model, s); l.backward()
    both = model.arena.grad
    assert float((both - (g1 + g2)).abs().max()) <= 1e-4 * float(both.abs().max())


def test_bf16_dropout_step_is_finite():
    a, task, model, crit, tr = _setup(torch.bfloat16, batch=8, frames=320)
    sample = tr.prepare(task.dummy_batch(seed=5))
    for _ in range(3):
        tr.train_step([sample])
    st = tr.reduce_stats()
    assert all(np.isfinite(v) for v in st.values()), st


def test_training_from_the_on_disk_split():
    """SURVEY 8-f N1 end to end: TNTIDX split (written by the reference's builders) -> load_dataset -> frame-budget batches with
    pinned prefetch -> Trainer updates; the empty filler batch of a short shard is skipped like the reference's dummy batch."""
    import os
    from helpers import GOLDEN
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.registry import apply_arch, namespace, setup_task
    from fbk_fairseq_st_amd.trainer import Trainer
    a = namespace(arch="s2t_transformer_xs", task="speech_translation_with_transcription", data=os.path.join(GOLDEN, "s2t_data"),
                  source_lang="en", target_lang="de", criterion="ctc_multi_loss", underlying_criterion="label_smoothed_cross_entropy",
                  label_smoothing=0.1, sentence_avg=False, ctc_compress_out=True, ctc_encoder_layer=2, ctc_weight=1.0,
                  input_feat_per_channel=80, no_attn_2d=True, lr=[1e-3], adam_betas="(0.9, 0.98)", clip_norm=20.0, warmup_updates=1,
                  warmup_init_lr=1e-3, seed=3, encoder_layers=2, decoder_layers=1, max_source_positions=80, max_target_positions=50,
                  dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, relu_dropout=0.0)
    apply_arch(a)
    task = setup_task(a)
    task.load_dataset("train")
    torch.manual_seed(0)
    model, crit = task.build_model(a), task.build_criterion(a)
    tr = Trainer(a, task, model, crit, device="cuda:0", compute_dtype=torch.float32)
    itr = task.get_batch_iterator(task.dataset("train"), max_tokens=150, max_positions=(80, 50), ignore_invalid_inputs=True, seed=1,
                                  num_shards=2, shard_id=1, epoch=1)
    first = last = None
    for epoch in range(4):
        n = frames = 0
        for batch in itr.next_epoch_itr(shuffle=True):
            if len(batch):
                assert batch["net_input"]["src_tokens"].is_pinned()
            tr.train_step([batch])                         # {} -> no micro-batch: a zero-gradient update
            if len(batch):
                st = tr.reduce_stats()
                assert np.isfinite(st["loss"]) and np.isfinite(st["gnorm"])
                n += 1; frames += st["nframes"]
                last = st["loss"] / st["sample_size"]
                first = last if first is None else first
        assert n >= 2 and frames > 0
    assert last < first


def test_train_step_applies_time_stretch_and_specaugment():
    """the task's train_step runs TimeStretch then SpecAugment on the staged batch (speech_recognition.py:254-258)"""
    import random
    a, task, model, crit, tr = _setup(torch.float32, dropout=0.0)
    from fbk_fairseq_st_amd.augment import SpecAugment, TimeStretch
    task.specaugment = SpecAugment(13, 13, 2, 2, 1.0)
    task.time_stretch = TimeStretch(1.0, 5, 0.8, 1.25)
    sample = tr.prepare(task.dummy_batch(seed=1, lengths=[200, 180, 150, 120]))
    random.seed(1); np.random.seed(1)
    tr.train_step([sample])
    st = tr.reduce_stats()
    assert np.isfinite(st["loss"]) and np.isfinite(st["gnorm"])
    assert st["nframes"] != 650                           # stretched lengths are what the step saw
    assert sample["net_input"]["src_tokens"].shape[1] == 200   # the staged batch itself is untouched
