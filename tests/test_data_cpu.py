"""SURVEY 8-f N1 on the CPU: the TNTIDX on-disk format (files written by the reference's own builders, tests/golden/tntidx/) and
frame-budget batching (batches produced by the reference's cython batch_by_size_fast, tests/golden/data.npz): oracle restatement,
native host function and reader/writer against them -- all bit-exact."""
import filecmp
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_golden
from oracle import int_ref


def _batches(g, ci):
    flat, offs = g["bbs%d_flat" % ci], g["bbs%d_offs" % ci]
    return [flat[offs[b]:offs[b + 1]].tolist() for b in range(len(offs) - 1)]


def test_oracle_batch_by_size_matches_reference():
    g = load_golden("data")
    for ci in range(int(g["bbs_ncases"])):
        mt, ms, mult = [int(v) for v in g["bbs%d_params" % ci]]
        got = int_ref.batch_by_size(g["bbs%d_order" % ci], g["bbs%d_lens" % ci], mt, ms, mult)
        assert got == _batches(g, ci), ci


def test_native_batch_by_size_matches_reference_and_oracle():
    from fbk_fairseq_st_amd.indexed import batch_by_size
    g = load_golden("data")
    for ci in range(int(g["bbs_ncases"])):
        mt, ms, mult = [int(v) for v in g["bbs%d_params" % ci]]
        lens, order = g["bbs%d_lens" % ci], g["bbs%d_order" % ci]
        got = batch_by_size(order, lens, mt if mt > 0 else None, ms if ms > 0 else None, mult)
        assert got == _batches(g, ci), ci
        assert got == batch_by_size(order, lambda i: int(lens[i]), mt if mt > 0 else None, ms if ms > 0 else None, mult)   # callable form
    rs = np.random.RandomState(0)                        # larger random cases against the oracle
    for _ in range(20):
        n = int(rs.randint(1, 400))
        lens = rs.randint(1, 2000, size=n).astype(np.int64)
        order = rs.permutation(n).astype(np.int64)
        mt, ms, mult = int(rs.choice([2000, 6000, 12000])), int(rs.choice([-1, 3, 16])), int(rs.choice([1, 4, 8]))
        assert batch_by_size(order, lens, mt, ms if ms > 0 else None, mult) == int_ref.batch_by_size(order, lens, mt, ms, mult)
    with pytest.raises(AssertionError, match="exceeds max_tokens"):
        batch_by_size(np.arange(3), np.array([10, 500, 20]), 100)


def test_tntidx_reader_on_reference_written_files():
    from fbk_fairseq_st_amd.indexed import FilterBanksDataset, IndexedDataset
    g = load_golden("data")
    fb = FilterBanksDataset(os.path.join(GOLDEN, "tntidx", "fbank"))
    assert len(fb) == 4 and fb.dtype == np.float32
    for i in range(len(fb)):
        assert torch.equal(fb[i], torch.from_numpy(g["fbank_%d" % i]))
        assert fb.size(i) == g["fbank_sizes"][i] == fb.num_tokens(i)
    assert fb.frame_lengths.tolist() == g["fbank_sizes"].tolist()
    tk = IndexedDataset(os.path.join(GOLDEN, "tntidx", "tokens"), fix_lua_indexing=True)
    assert len(tk) == 3
    for i in range(len(tk)):
        assert torch.equal(tk[i], torch.from_numpy(g["tokens_%d" % i]))
    with pytest.raises(IndexError):
        tk[3]


def test_tntidx_writer_is_byte_identical_to_the_reference(tmp_path):
    from fbk_fairseq_st_amd.indexed import AudioIndexedDatasetBuilder, IndexedDatasetBuilder
    g = load_golden("data")
    b = AudioIndexedDatasetBuilder(str(tmp_path / "fbank.bin"))
    for i in range(4):
        b.add_item(torch.from_numpy(g["fbank_%d" % i]))
    b.finalize(str(tmp_path / "fbank.idx"))
    b = IndexedDatasetBuilder(str(tmp_path / "tokens.bin"))
    for i in range(3):
        b.add_item(torch.from_numpy(g["tokens_%d" % i]))
    b.finalize(str(tmp_path / "tokens.idx"))
    for f in ("fbank.bin", "fbank.idx", "tokens.bin", "tokens.idx"):
        assert filecmp.cmp(str(tmp_path / f), os.path.join(GOLDEN, "tntidx", f), shallow=False), f


def _s2t_task():
    from fbk_fairseq_st_amd import tasks  # noqa: F401
    from fbk_fairseq_st_amd.registry import namespace, setup_task
    a = namespace(task="speech_translation_with_transcription", data=os.path.join(GOLDEN, "s2t_data"), source_lang="en", target_lang="de",
                  criterion="ctc_multi_loss", max_source_positions=80, max_target_positions=50)
    return setup_task(a)


@pytest.mark.parametrize("prefetch", [0, 2])
def test_data_pipeline_reproduces_the_reference_batches(prefetch):
    """G13: load_dataset + get_batch_iterator on the on-disk split written by the reference's builders: every batch of epochs 1 and 2
    on both shards equals the reference's (ids, padded frames after per-utterance CMVN, lengths, targets, eos-shifted inputs, transcripts),
    including the utterance dropped by max_positions and the empty filler batch of the short shard."""
    from fbk_fairseq_st_amd.iterators import get_batch_iterator
    g = load_golden("iterator")
    task = _s2t_task()
    assert len(task.source_dictionary) == 35 and task.source_dictionary.index("<ctc_blank>") == 34
    task.load_dataset("train")
    ds = task.dataset("train")
    assert len(ds) == 14
    for shard in (0, 1):
        it = get_batch_iterator(ds, max_tokens=150, max_positions=(80, 50), ignore_invalid_inputs=True, seed=1, num_shards=2, shard_id=shard,
                                epoch=1, prefetch=prefetch, pin_memory=False)
        for ep in (1, 2):
            batches = list(it.next_epoch_itr(shuffle=True))
            assert len(batches) == int(g["s%d_e%d_n" % (shard, ep)]) == len(it)
            for k, b in enumerate(batches):
                pre = "s%d_e%d_b%d_" % (shard, ep, k)
                if pre + "empty" in g:
                    assert b == {}
                    continue
                assert b["id"].tolist() == g[pre + "id"].tolist()
                assert int(b["ntokens"]) == int(g[pre + "ntokens"])
                np.testing.assert_allclose(b["net_input"]["src_tokens"].numpy(), g[pre + "src_tokens"], rtol=0, atol=1e-6)
                for kk in ("src_lengths", "prev_output_tokens", "transcript_prev_output_tokens"):
                    assert np.array_equal(b["net_input"][kk].numpy(), g[pre + kk]), kk
                for kk in ("target", "target_lengths", "transcript_target", "transcript_target_lengths"):
                    assert np.array_equal(b[kk].numpy(), g[pre + kk]), kk
    with pytest.raises(Exception, match="invalid"):
        get_batch_iterator(ds, max_tokens=150, max_positions=(80, 50), ignore_invalid_inputs=False)


def _augment_tables(g, ci):
    """draw the tables of case ci with the generator seeds the reference run used; returns (row_map, new_lengths, fmask, tmask)"""
    import random
    from fbk_fairseq_st_amd.augment import SpecAugment, TimeStretch
    x, lens = g["c%d_in" % ci], g["c%d_lens" % ci]
    sa, ts = g["c%d_sa" % ci], g["c%d_ts" % ci]
    random.seed(100 + ci); np.random.seed(200 + ci)
    rm = new_len = fm = tm = None
    B, tau, v = x.shape
    if ts[0] >= 0:
        rm, new_len = TimeStretch(float(ts[0]), int(ts[1]), float(ts[2]), float(ts[3])).row_map(lens.tolist())
        tau = rm.shape[1]
    if sa[0] >= 0:
        aug = SpecAugment(int(sa[0]), int(sa[1]), int(sa[2]), int(sa[3]), float(sa[4]))
        fm, tm = aug.tables(B, tau, v)
    return rm, new_len, fm, tm


def test_augmentation_tables_reproduce_the_reference_batches():
    """G15: the host side of TimeStretch + SpecAugment draws the reference's random numbers in the reference's order: applying the
    tables (numpy oracle of the kernel's contract) gives the reference's augmented batches bit for bit."""
    g = load_golden("augment")
    for ci in range(int(g["ncases"])):
        rm, new_len, fm, tm = _augment_tables(g, ci)
        x = g["c%d_in" % ci]
        if rm is not None:
            st = int_ref.apply_augment(x, rm)
            assert np.array_equal(st, g["c%d_ts_tokens" % ci]) and new_len == g["c%d_ts_lengths" % ci].tolist()
        out = int_ref.apply_augment(x, rm, fm, tm)
        assert np.array_equal(out, g["c%d_out" % ci]), ci
        if new_len is not None:
            assert new_len == g["c%d_out_lengths" % ci].tolist()


def test_teacher_output_files_and_kd_collater():
    """G16: reader / writer of the teacher top-k files (written by the reference's TeacherOutputDataset.save_bin) and the batch
    DatasetWithTeacherOutput collates (fairseq/data/knowledge_distillation.py)."""
    from fbk_fairseq_st_amd.indexed import DatasetWithTeacherOutput, TeacherOutputDataset
    g = load_golden("teacher")
    K_ = int(g["K"])
    pre = os.path.join(GOLDEN, "s2t_data", "train.en-de.de")
    ti = TeacherOutputDataset(pre + ".top%d_idx" % K_, np.int32); to = TeacherOutputDataset(pre + ".top%d_out" % K_, np.float32)
    assert len(ti) == len(to) == 14
    for i in range(14):
        assert ti[i].dtype == torch.long and torch.equal(ti[i], torch.from_numpy(g["idx_%d" % i]))
        assert to[i].dtype == torch.float32 and torch.equal(to[i], torch.from_numpy(g["out_%d" % i]))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        TeacherOutputDataset.save_bin(os.path.join(d, "i"), [g["idx_%d" % i].tolist() for i in range(14)], np.int32)
        TeacherOutputDataset.save_bin(os.path.join(d, "o"), [g["out_%d" % i].tolist() for i in range(14)], np.float32)
        for a, b in (("i", ".top%d_idx" % K_), ("o", ".top%d_out" % K_)):
            for ext in (".idx", ".bin"):
                assert filecmp.cmp(os.path.join(d, a + ext), pre + b + ext, shallow=False), (a, ext)
    task = _s2t_task()
    task.load_dataset("train")
    kd = DatasetWithTeacherOutput(task.dataset("train"), to, ti, task.target_dictionary, K_)
    batch = kd.collater([kd[i] for i in (3, 0, 7, 9)])
    assert batch["id"].tolist() == g["batch_id"].tolist() and np.array_equal(batch["target"].numpy(), g["batch_target"])
    assert np.array_equal(batch["teacher_output"][0].numpy(), g["batch_teacher_idx"])
    assert np.array_equal(batch["teacher_output"][1].numpy(), g["batch_teacher_out"])


def test_arena_adam_skips_the_layers_layerdrop_removed():
    """optim.ArenaAdam.step under LayerDrop, host logic on the CPU stand-ins: a droppable group that no forward of the update ran is
    skipped (parameters, moments and its own step count untouched: fairseq/optim/adam.py:160-165 on a gradient that is None after
    fairseq_optimizer.py:97-101), every other parameter steps with its own bias-correction count, and the per-group counts survive
    both state-dict layouts.  Against a per-parameter torch loop that applies the reference's rule."""
    import torch
    from cpu_stubs import cpu_kernels
    from fbk_fairseq_st_amd.arena import ParamArena
    from fbk_fairseq_st_amd.optim import ArenaAdam
    from oracle import s2t_ref
    shapes = {"encoder.fc3.weight": (5, 7)}
    for l in range(3):
        shapes["encoder.layers.%d.fc1.weight" % l] = (6, 5); shapes["encoder.layers.%d.fc1.bias" % l] = (6,)
    shapes["encoder.layer_norm.weight"] = (5,)
    with cpu_kernels():
        arena = ParamArena(shapes, "cpu")
        arena.drop_groups += ["encoder.layers.%d." % l for l in range(3)]
        g = torch.Generator().manual_seed(0)
        for n in shapes:
            arena.p(n).copy_(torch.randn(shapes[n], generator=g))
        opt = ArenaAdam(arena, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-2)
        P = {n: arena.p(n).clone() for n in shapes}
        M = {n: torch.zeros(shapes[n]) for n in shapes}; V = {n: torch.zeros(shapes[n]) for n in shapes}
        steps = {n: 0 for n in shapes}
        keeps = [[True, False, True], [False, False, True], [True, True, True], [True, False, False]]
        for it, keep in enumerate(keeps):
            opt.zero_grad()
            arena.note_layers(["encoder.layers.%d." % l for l in range(3)], keep)
            grads = {}
            for n in shapes:
                l = int(n.split(".")[2]) if ".layers." in n else None
                if l is None or keep[l]:
                    grads[n] = torch.randn(shapes[n], generator=g)
                    arena.g(n).copy_(grads[n])
            opt.step()
            for n, gr in grads.items():
                steps[n] += 1
                P[n], M[n], V[n] = s2t_ref.adam_step(P[n], gr, M[n], V[n], steps[n], 1e-2, 0.9, 0.98, 1e-8, 1e-2)
        for n in shapes:
            assert torch.allclose(arena.p(n), P[n], atol=1e-6), n
            assert torch.allclose(arena._view(arena.exp_avg, n), M[n], atol=1e-7), n
        assert opt.group_steps == {"encoder.layers.0.": 3, "encoder.layers.1.": 1, "encoder.layers.2.": 3} and opt.step_count == 4
        # both state-dict layouts carry the counts
        names = list(shapes)
        ref = opt.reference_state_dict(names)
        assert [ref["state"][i]["step"] for i in range(len(names))] == [steps[n] for n in names]
        o2 = ArenaAdam(arena, lr=1e-2); o2.load_reference_state_dict(ref, names)
        assert o2.group_steps == opt.group_steps and o2.step_count == 4
        o3 = ArenaAdam(arena, lr=1e-2); o3.load_state_dict(opt.state_dict())
        assert o3.group_steps == opt.group_steps and o3.step_count == 4
        # an update whose forward passes drew no LayerDrop decision (evaluation, or LayerDrop off) steps everything
        opt.zero_grad(); opt.step()
        assert opt.group_steps == {"encoder.layers.0.": 4, "encoder.layers.1.": 2, "encoder.layers.2.": 4}
