#!/usr/bin/env python3
"""Benchmark of the S2T training hot path on MI355X.

Metric (BASELINE.json): audio-frames/sec (train fwd+bwd) of s2t_transformer_m on 80-mel synthetic filterbanks,
= sum of src_lengths (10 ms input frames, the reference's own `nframes` counter, CTC_loss.py:173) divided by
the wall time of full updates (forward + CTC/label-smoothed losses + backward + gradient all-reduce + clip + Adam).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Headline workload (configs[2] of BASELINE.json): s2t_transformer_m (conv_transformer structure,
D=512, FFN=2048, 8 heads, 12+6 layers) + ctc_multi_loss with --ctc-compress-out after encoder layer 8,
T=1500 frames x 80 mel, per-GPU batch 64 utterances (96,000 frames: the effective per-GPU batch of the
reference's paper script, --max-tokens 12000 x --update-freq 8, README.md:144,152, taken in one pass because
288 GB of HBM allows it), target/transcript length 40, V_tgt=8000, V_src=5000+<ctc_blank>, bf16 storage with f32
accumulation and f32 master weights, dropout ON at the preset rates.  Weak scaling: every rank gets its own batch.
The timed updates run on the random-init model (--lr 1e-9: the optimizer step is complete but does not move the weights), where the
CTC compression keeps ~98 % of the frames: a model that learns on N(0,1) inputs predicts blank everywhere after ~5 updates and
encoder layers 9-12 then process ONE frame (--lr 5e-3 shows that state); `config.frames_after_ctc_compression` reports the length.

Secondary entries of the same JSON line (`extra_configs`, single GPU only, fewer timed updates; --no-extra skips them):
  cfg3_batch8   the per-GPU batch SURVEY.md 8-d writes for Cfg3 (8 x 1500 frames)
  cfg2_s_fp32   s2t_transformer_s, 32 x 1000 frames, fp32 (configs[1])
  cfg4_l_bucketed  s2t_transformer_l on MuST-C-shaped lengths (lognormal, clipped to [50, 2000]) fed by the real batch iterator
                (length-bucketed frame-budget batches, pinned prefetch thread, H2D inside the timed loop)
--loader runs the HEADLINE workload from that iterator too (collate + pinning + H2D inside the timed loop).

Rank 0 prints ONE JSON line and exits non-zero if the loss or the gradient norm of the timed updates is not finite.
`roofline` times the dominant kernel family with HIP events inside the library (s2t_prof_*, on the launch stream) during
instrumented updates that directly follow the timed region (inside it the events themselves cost 12 % of the update;
--no-roofline skips them); `cpu_baseline` times the CPU oracle (port of the reference path, verified against it) on the
host cores for full updates (forward, losses, backward, clip, Adam) on a bounded sample of the workload.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# kernel behind each profiled family (names as they appear in the rocprofv3 kernel trace, profiles/)
KERNEL_OF = {"wgrad_group_f32": "wgrad_f32_kernel (all dW = dY^T X of a backward pass in f32 mode: 128x128 tiles owned over all tokens, exact-f32 MFMA, no atomics)",
             "wgrad_group": "wgrad_group_kernel (all dW = dY^T X of a backward pass in one launch: 256x256 tiles owned over all tokens, LDS-DMA, no split-K)",
             "gemm256_nt": "gemm256_kernel<bf16, TB=false> (Y = X W^T: 256/192 x 256 x 64, LDS-DMA, one persistent workgroup per CU)",
             "gemm256_nn": "gemm256_kernel<bf16, TB=true> (dX = dY W: 256/192 x 256 x 64, LDS-DMA, one persistent workgroup per CU)",
             "gemm_nt": "gemm_fast_kernel<.., 128, 128, 8 waves> (register-staged; products the LDS-DMA kernels do not take: decoder side, f32)",
             "gemm_nn": "gemm_fast_kernel<.., 128, 128, 8 waves>, data-gradient form (register-staged)",
             "gemm_tn": "gemm_tn2_kernel (dW of the fc3 projection: 128x128 tile, split-K atomics)",
             "gemm_tn_small": "gemm_fast_kernel<.., 64, 64, 4 waves> (small dW)", "gemm_nt_small": "gemm_fast_kernel<.., 64, 64, 4 waves>",
             "gemm_nn_small": "gemm_fast_kernel<.., 64, 64, 4 waves>", "gemm_gather": "gemm_kernel<..> with row gather (implicit-GEMM convolutions)",
             "conv2_fwd": "conv2_fwd_kernel (3x3 stride-2 convolution, input rows staged once in LDS, weights in registers)",
             "conv2_dgrad": "conv2_dgrad_kernel (its data gradient, four pixel-parity classes in one launch)"}
TRAFFIC_FILE = os.path.join("profiles", "r06_hbm_traffic.json")
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA, MI355X_MICROARCH.md (AMD's 5 PF figure includes 2:1 sparsity)
PEAK_F32_TFLOPS = 157.3


def hbm_traffic(family):
    """HBM bytes per launch of the family's kernels from the committed rocprofv3 PMC passes of this same command
    (made by tools/hbm_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes, MI355X_MICROARCH.md section HBM);
    None when the file is absent.  NOT measured by this run: `traffic_source` in the JSON says so."""
    try:
        with open(os.path.join(REPO, TRAFFIC_FILE)) as f:
            t = json.load(f)
        return t["families"][family]["bytes_per_launch"]
    except Exception:
        return None


def build_all(arch, batch, frames, tgt_len, ctc_layer, lr, dtype, device, attn_2d=False, want_sd=False, criterion="ctc_multi_loss", **over):
    from fbk_fairseq_st_amd import conv_transformer, criterions, tasks  # noqa: F401
    from fbk_fairseq_st_amd.registry import apply_arch, namespace, setup_task
    from fbk_fairseq_st_amd.trainer import Trainer
    kw = dict(arch=arch, task="dummy_s2t", criterion=criterion, label_smoothing=0.1, sentence_avg=True,
              input_feat_per_channel=80, no_attn_2d=not attn_2d, dict_size=8000 - 4, src_dict_size=5000 - 4,
              batch_size=batch, frames=frames, tgt_len=tgt_len, transcript_len=max(tgt_len, 40),
              lr=[lr], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=20.0,
              warmup_updates=4000, warmup_init_lr=min(lr, 3e-4), seed=1, bf16=(dtype == torch.bfloat16), bucket_cap_mb=64,
              bucket_by_length=True)
    kw.update(over)
    a = namespace(**kw)
    if criterion == "ctc_multi_loss":
        a.underlying_criterion, a.ctc_compress_out, a.ctc_encoder_layer = "label_smoothed_cross_entropy", True, ctc_layer
        a.ctc_weight, a.ctc_compress_strategy = 1.0, "avg"
    apply_arch(a)
    task = setup_task(a)
    torch.manual_seed(1)
    model = task.build_model(a)
    crit = task.build_criterion(a)
    ref_sd = model.state_dict() if want_sd else None
    trainer = Trainer(a, task, model, crit, device=device, compute_dtype=dtype)
    return a, task, model, crit, trainer, ref_sd


def timed_updates(trainer, next_batch, steps, warmup, world, device):
    """W untimed + exactly K timed updates between barrier + synchronize; returns (seconds, frames, mean frames after compression, stats)"""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        trainer.train_step([next_batch()])
    trainer.reduce_stats()
    barrier()
    t0 = time.perf_counter()
    frames, enc_len = 0, []
    for _ in range(steps):
        b = next_batch()
        frames += int(b["nframes"])
        trainer.train_step([b])
        enc_len.append(trainer.model.encoder._last["lengths_host"])       # host list of this update (no extra sync)
    barrier()
    dt = time.perf_counter() - t0
    stats = trainer.reduce_stats()
    stats["_rank_ms"] = (dt / steps * 1e3, dt / steps * 1e3)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        tmin = tmax.clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tmin, op=torch.distributed.ReduceOp.MIN)
        stats["_rank_ms"] = (float(tmin.item()) / steps * 1e3, float(tmax.item()) / steps * 1e3)
        dt = float(tmax.item())
    enc_mean = sum(sum(l) / len(l) for l in enc_len) / max(len(enc_len), 1)
    return dt, frames, enc_mean, stats


# RCCL all-reduce bus bandwidth ASSUMED by the projection below (8 MI355X, xGMI: 7 links x ~153 GB/s per GPU; a ring / tree over all
# links sustains a fraction of the 1,075 GB/s aggregate).  Not measured: no multi-GPU node was available to this build.
ASSUMED_BUS_GBPS = 300.0


def dp_dry_run(trainer, next_batch, ms_single, steps=6):
    """With ONE rank there is no collective, but the data-parallel machinery above the collective can still be exercised and
    observed: the update runs on the data-parallel SCHEDULE (mid-backward weight-gradient flush), the engine's
    readiness reports reach the static bucket plan, and a device event marks where on the compute stream each bucket would have been
    handed to RCCL.  Reported: the schedule's cost, the share of gradient bytes launched from inside backward, the bytes left after
    it, and -- clearly labelled -- a PROJECTION of the exposed all-reduce time at 8 ranks under an assumed bus bandwidth."""
    from fbk_fairseq_st_amd import distributed as D
    trainer.set_dp_dry_run(True)
    red = trainer.reducer
    try:
        for _ in range(2):
            trainer.train_step([next_batch()])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            trainer.train_step([next_batch()])
        torch.cuda.synchronize()
        ms_sched = (time.perf_counter() - t0) / steps * 1e3
        red.record_events = True
        trainer.train_step([next_batch()])
        torch.cuda.synchronize()
        ev0 = red.events[0][3]
        launches = [(e - s_, ev0.elapsed_time(ev)) for s_, e, early, ev in red.events]
        finish_ms = ev0.elapsed_time(red.finish_event)
        early_elems, n = red.early_elems, red.n
        red.record_events = False
        # (c) what sharing the chip with the collective costs, measured: at every bucket launch a stand-in for an 8-rank ring all-reduce at
        # the ASSUMED bus bandwidth (s2t_comm_standin: 16 workgroups resident for the all-reduce's duration, moving the bucket's bytes
        # through HBM twice) runs on a side stream; clip / Adam wait for it as they would for RCCL.  Once with the persistent kernels on all
        # 256 CUs, once leaving 16 to it (s2t_set_option reserve_cus, the Trainer's --reserve-cus default at world > 1).
        from fbk_fairseq_st_amd import kernels as K
        standin = {}
        scratch = torch.empty(red.bucket_elems, dtype=red.flat.dtype, device=red.flat.device)
        red.standin = dict(stream=torch.cuda.Stream(), scratch=scratch, workgroups=16, ranks=8, bus_gbps=ASSUMED_BUS_GBPS)
        try:
            for reserve in (0, 16):
                K.set_option("reserve_cus", reserve)
                trainer.train_step([next_batch()])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    trainer.train_step([next_batch()])
                torch.cuda.synchronize()
                standin["ms_per_step_reserve%d" % reserve] = round((time.perf_counter() - t0) / steps * 1e3, 3)
            K.set_option("reserve_cus", 16)
            red.standin = None
            trainer.train_step([next_batch()])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                trainer.train_step([next_batch()])
            torch.cuda.synchronize()
            standin["ms_per_step_reserve16_no_standin"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        finally:
            red.standin = None
            K.set_option("reserve_cus", trainer.reserve_cus)
        standin.update(what="one GPU: the data-parallel schedule with a stand-in for the all-reduce on a side stream (16 workgroups resident for "
                            "2 (N-1)/N x bucket bytes / bus, N = 8, moving 2 x bucket bytes through HBM each way); NOT a measurement of xGMI",
                       workgroups=16, ranks=8, assumed_bus_GBps=ASSUMED_BUS_GBPS)
        del scratch
        proj = {}
        for nr in (2, 4, 8):
            exposed, total = D.project_exposed_allreduce(launches, finish_ms, nr, ASSUMED_BUS_GBPS, red.flat.element_size())
            proj["n%d" % nr] = {"allreduce_ms_total": round(total, 3), "exposed_ms_after_backward": round(exposed, 3),
                                "projected_scaling": round(nr * ms_single / (ms_sched + exposed), 3)}
        return {"what": "one rank, no collective: reducer bookkeeping on the data-parallel schedule; `projection` is NOT a measurement",
                "ms_per_step_dp_schedule": round(ms_sched, 3), "ms_per_step_single": round(ms_single, 3),
                "allreduce_bytes_per_step": n * red.flat.element_size(), "allreduce_buckets": len(red.plan),
                "allreduce_launched_in_backward_frac": round(early_elems / max(n, 1), 4),
                "exposed_bytes_after_last_flush": (n - early_elems) * red.flat.element_size(),
                "bucket_launch_ms_before_backward_end": [round(finish_ms - t, 3) for _, t in launches],
                "comm_standin": standin,
                "projection": dict(proj, assumed_bus_GBps=ASSUMED_BUS_GBPS,
                                   formula="bucket all-reduce = 2 (N-1)/N x bytes / bus, buckets back to back from their launch times; "
                                           "scaling = N x single-rank ms / (dp-schedule ms + exposed ms)")}
    finally:
        red.record_events = False
        trainer.set_dp_dry_run(False)


def loader_batches(task, trainer, lengths, max_tokens, max_sentences, seed=1):
    """endless stream of batches from the product iterator (frame-budget batcher, per-epoch shuffle, pinned prefetch thread)"""
    task.load_dataset("train", lengths=lengths, seed=seed)
    itr = task.get_batch_iterator(task.dataset("train"), max_tokens=max_tokens, max_sentences=max_sentences, seed=seed)

    def gen():
        while True:
            for b in itr.next_epoch_itr(shuffle=True):
                if len(b):
                    yield b
    g = gen()
    return lambda: next(g)


def finite(stats):
    return all(math.isfinite(float(stats.get(k, float("nan")))) for k in ("loss", "gnorm"))


def cpu_baseline(args, a, ref_sd, task):
    """The CPU oracle (oracle/s2t_ref.py, a verified port of the reference path) on the host cores: FULL updates (forward +
    losses + backward + gradient clipping + Adam, fairseq/trainer.py:334-495) of the same model on a bounded sample of the
    workload (B_cpu utterances x T frames), >= 3 timed updates after one warm-up."""
    from oracle import s2t_ref
    import torch.nn.functional as F
    # torch's intra-op pool does not scale past a few dozen threads on these small GEMMs (256 threads measured
    # 80x slower than 8 on this path): use min(32, host cores) and report that number as `cores`
    ncores = min(args.cpu_threads, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    cfg = s2t_ref.default_cfg(D=a.encoder_embed_dim, heads=a.encoder_attention_heads, ffn=a.encoder_ffn_embed_dim,
                              enc_layers=a.encoder_layers, dec_layers=a.decoder_layers, ctc_layer=a.ctc_encoder_layer)
    W = {k: v.clone().requires_grad_(v.dtype.is_floating_point and v.dim() > 0 and "running" not in k and "_float_tensor" not in k
                                     and "version" not in k)
         for k, v in ref_sd.items()}
    train = [k for k, v in W.items() if v.requires_grad]
    m = {k: torch.zeros_like(W[k]) for k in train}
    v2 = {k: torch.zeros_like(W[k]) for k in train}
    Bc = args.cpu_batch
    from fbk_fairseq_st_amd.data import synthetic_batch
    s = synthetic_batch(Bc, args.frames, args.tgt_len, args.tgt_len, len(task.tgt_dict), task.src_dict.index("<ctc_blank>"), seed=7)
    blank = task.src_dict.index("<ctc_blank>")
    # the reference calls torch's own F.ctc_loss (CTC_loss.py:143): use it for a representative cost
    s2t_ref.ctc_loss_sum = lambda logits, t, il, tl, b: F.ctc_loss(
        F.log_softmax(logits.float(), -1), torch.cat([t[i, : tl[i]] for i in range(t.shape[0])]), il, tl,
        blank=b, reduction="sum", zero_infinity=True)
    times = []
    for it in range(args.cpu_updates + 1):
        t0 = time.perf_counter()
        for k in train:
            W[k].grad = None
        loss, ss, log, enc, logits, _ = s2t_ref.ctc_multi_loss(W, cfg, s, 0.1, 1.0, blank, training=True)
        loss.backward()
        with torch.no_grad():
            grads = [W[k].grad / float(ss) if W[k].grad is not None else torch.zeros_like(W[k]) for k in train]
            _, grads = s2t_ref.clip_grad_norm(grads, 20.0)
            for k, g in zip(train, grads):
                p, m[k], v2[k] = s2t_ref.adam_step(W[k].data, g, m[k], v2[k], it + 1, 1e-9)
                W[k].data.copy_(p)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    return {"value": Bc * args.frames / t, "unit": "audio-frames/s", "cores": ncores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "oracle/s2t_ref.py full update (ctc_multi_loss fwd+bwd, clip, Adam), fp32, %d x %d frames, mean of %d updates after 1 warm-up"
                      % (Bc, args.frames, args.cpu_updates)}


FAMILIES = ("wgrad_group", "wgrad_group_f32", "gemm256_nt", "gemm256_nn", "gemm_tn", "gemm_nt", "gemm_nn", "gemm_tn_small", "gemm_nt_small", "gemm_nn_small",
            "gemm_gather", "conv2_fwd", "conv2_dgrad", "attn_fwd", "attn_bwd")


def roofline_of(trainer, next_batch, prof_steps, dtype, instrument=True, traffic=True):
    """`prof_steps` instrumented updates (HIP events around every MFMA-product launch, inside the library, on the launch stream) ->
    the roofline object of the dominant kernel: the product family (one kernel template each, KERNEL_OF) with the most time per update"""
    from fbk_fairseq_st_amd import kernels as K
    if instrument:
        K.prof_reset(); K.prof_enable(True)
    for _ in range(prof_steps):
        trainer.train_step([next_batch()])
    torch.cuda.synchronize()
    K.prof_enable(False)
    if not instrument:
        return None
    fam = {f: K.prof_read(f) for f in FAMILIES}
    if fam["attn_bwd"]["flops"] > 0:
        # the library counts every MFMA the two backward kernels issue (14 B H Tq Tk d: S and dP are recomputed in both); the figure
        # reported is the conventional one, 2.5 x the forward = 10 B H Tq Tk d
        fam["attn_bwd"] = dict(fam["attn_bwd"], flops=fam["attn_bwd"]["flops"] * 10.0 / 14.0)
    mfma = {f: v for f, v in fam.items() if v["launches"] > 0 and v["ms"] > 0 and v["flops"] > 0}     # every MFMA family, attention included
    gemms = {f: v for f, v in mfma.items() if not f.startswith("attn")}
    if not gemms:
        return None
    dom = max(gemms, key=lambda f: gemms[f]["ms"])                     # the headline kernel: the product family with the most time
    gm = gemms[dom]
    ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12
    peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
    tot_fl = sum(v["flops"] for v in gemms.values()); tot_ms = sum(v["ms"] for v in gemms.values())
    tf = {k: v["flops"] / (v["ms"] * 1e-3) / 1e12 for k, v in mfma.items()}
    # the MFMA family FURTHEST below its roofline among those that matter (>= 5 % of the MFMA time): reported beside the longest one
    all_ms = sum(v["ms"] for v in mfma.values())
    big = {k: v for k, v in mfma.items() if v["ms"] >= 0.05 * all_ms}
    low = min(big, key=lambda f: tf[f]) if big else dom
    roof = {"bound": "mfma", "kernel": KERNEL_OF[dom], "family": dom,
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": hbm_traffic(dom) if traffic else None,
            "avg_launch_us": round(gm["ms"] * 1e3 / gm["launches"], 2),
            "launches_per_step": gm["launches"] // prof_steps,
            "flop_per_launch": round(gm["flops"] / gm["launches"] / 1e9, 3), "flop_unit": "GFLOP",
            "all_gemm_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), "all_gemm_frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
            "tflops_by_family": {k: round(v, 1) for k, v in tf.items()},
            "frac_by_family": {k: round(v / peak, 4) for k, v in tf.items()},
            "lowest_frac_family": {"family": low, "achieved": round(tf[low], 1), "frac": round(tf[low] / peak, 4),
                                   "ms_per_step": round(mfma[low]["ms"] / prof_steps, 3),
                                   "note": "attention is counted at 4 (forward) / 10 (backward) x B H Tq Tk d: the second recomputation of S and dP in the backward is not counted"},
            "ms_per_step": {k: round(v["ms"] / prof_steps, 3) for k, v in fam.items() if v["launches"] > 0}}
    if traffic:
        roof["traffic_source"] = TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not re-measured by this run)"
    return roof


def extra_config(name, arch, dtype, device, steps, warmup, batch=None, frames=1500, tgt_len=40, ctc_layer=8, lengths=None, max_tokens=None, roofline=True):
    """one secondary workload; its model is freed before the next one is built"""
    a, task, model, crit, trainer, _ = build_all(arch, batch or 8, frames, tgt_len, ctc_layer, 1e-9, dtype, device)
    if lengths is None:
        sample = trainer.prepare(task.dummy_batch(seed=100))
        nxt = lambda: sample
        what = "%d x %d x 80 fbank per update, same batch resident in HBM" % (batch, frames)
    else:
        nxt = loader_batches(task, trainer, lengths, max_tokens, None)
        what = ("%d utterances, T ~ lognormal(ln 600, 0.7) in [50, 2000], length-bucketed batches of <= %d frames from the batch iterator "
                "(collate + pinned prefetch + H2D inside the timed loop)" % (len(lengths), max_tokens))
    dt, frames_done, enc_mean, stats = timed_updates(trainer, nxt, steps, warmup, 1, device)
    out = {"name": name, "value": round(frames_done / dt, 1), "unit": "audio-frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "dtype": "bf16" if dtype == torch.bfloat16 else "fp32",
           "config": {"workload": "%s + ctc_multi_loss(ctc-compress-out @ layer %d) full update, %s" % (arch, ctc_layer, what),
                      "frames_per_step": round(frames_done / steps, 1), "frames_after_ctc_compression": round(enc_mean, 1)},
           "loss_finite": finite(stats)}
    if roofline:
        # the dominant MFMA product family of THIS workload (two instrumented updates after its timed region; no PMC traffic figure:
        # the committed counter passes are of the headline command)
        out["roofline"] = roofline_of(trainer, nxt, 2, dtype, traffic=False)
    del trainer, model, crit, task
    torch.cuda.empty_cache()
    return out


def cfg5_train(name, arch, criterion, dtype, device, steps, warmup, cpu=True, **over):
    """BASELINE configs[4], training legs: the m-sized model on 16 x 1000 frames, target length 40, dropout on,
    full update (forward, loss, backward, clip, Adam).  `knowledge_distillation`: word-level KD from K = 8 stored teacher logits per
    target position (fairseq/criterions/knowledge_distillation.py:44-96); `cross_entropy_dualdecoder`: encoder + translation decoder +
    transcript decoder (conv_transformer_dualdecoder.py:13-81, criterions/cross_entropy_dualdecoder.py:8-83)."""
    B, T, L, K_TEACHER = 16, 1000, 40, 8
    a, task, model, crit, trainer, ref_sd = build_all(arch, B, T, L, 0, 1e-9, dtype, device, want_sd=cpu, criterion=criterion, **over)

    def make(batch, seed):
        from fbk_fairseq_st_amd.data import synthetic_batch
        s = synthetic_batch(batch, T, L, L, len(task.tgt_dict), task.src_dict.index("<ctc_blank>"), seed=seed)
        g = torch.Generator().manual_seed(seed)
        if criterion == "knowledge_distillation":
            tidx = torch.randint(4, len(task.tgt_dict), (batch, L, K_TEACHER), generator=g)
            tidx[:, :, 0] = s["target"]                                  # the teacher usually ranks the reference token first
            s["teacher_output"] = [tidx, torch.randn(batch, L, K_TEACHER, generator=g).sort(dim=-1, descending=True)[0] * 2.0]
        else:
            tr = s["transcript_target"]
            s["net_input"]["transcript_prev_output_tokens"] = torch.cat([torch.full((batch, 1), 2, dtype=torch.long), tr[:, :-1]], 1)
        return s
    sample = trainer.prepare(make(B, 100))
    nxt = lambda: sample
    dt, frames_done, _, stats = timed_updates(trainer, nxt, steps, warmup, 1, device)
    out = {"name": name, "value": round(frames_done / dt, 1), "unit": "audio-frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "dtype": "bf16" if dtype == torch.bfloat16 else "fp32",
           "config": {"workload": "%s + %s full update, %d x %d x 80 fbank, target/transcript len %d%s, dropout on, batch resident in HBM"
                                  % (arch, criterion, B, T, L, ", teacher top-%d" % K_TEACHER if criterion == "knowledge_distillation" else ""),
                      "frames_per_step": frames_done // steps},
           "loss_finite": finite(stats)}
    out["roofline"] = roofline_of(trainer, nxt, 2, dtype, traffic=False)
    if cpu:
        from oracle import s2t_ref
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        cfg = s2t_ref.default_cfg(D=a.encoder_embed_dim, heads=a.encoder_attention_heads, ffn=a.encoder_ffn_embed_dim,
                                  enc_layers=a.encoder_layers, dec_layers=a.decoder_layers, ctc_layer=0)
        W = {k: v.clone().requires_grad_(v.dtype.is_floating_point and v.dim() > 0 and "running" not in k and "_float_tensor" not in k
                                         and "version" not in k) for k, v in ref_sd.items()}
        train = [k for k, v in W.items() if v.requires_grad]
        Bc = 2
        s = make(Bc, 7)
        times = []
        for it in range(3):
            t0 = time.perf_counter()
            for k in train:
                W[k].grad = None
            ni = s["net_input"]
            if criterion == "knowledge_distillation":
                enc, _ = s2t_ref.encoder_forward(W, cfg, ni["src_tokens"], ni["src_lengths"], training=True)
                logits = s2t_ref.decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
                loss = s2t_ref.kd_loss(logits, s["target"], s["teacher_output"][0], s["teacher_output"][1], a.kd_lambda, a.kd_temperature, cfg["pad"])
            else:
                loss = s2t_ref.dual_decoder_loss(W, cfg, s, 0.1, a.primary_loss_weight, a.auxiliary_loss_weight, training=True)[0]
            loss.backward()
            with torch.no_grad():
                grads = [W[k].grad / float(s["ntokens"]) if W[k].grad is not None else torch.zeros_like(W[k]) for k in train]
                s2t_ref.clip_grad_norm(grads, 20.0)
            times.append(time.perf_counter() - t0)
        t = sum(times[1:]) / len(times[1:])
        out["cpu_baseline"] = {"value": round(Bc * T / t, 1), "unit": "audio-frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "oracle/s2t_ref.py forward + loss + backward + clip of the same model, fp32, %d x %d frames, mean of 2 after "
                                         "1 warm-up (Adam left out: a lower bound on the reference's update time)" % (Bc, T)}
    del trainer, model, crit, task
    torch.cuda.empty_cache()
    return out


def cfg5_beam5(name, dtype, device, runs=2, cpu=True):
    """BASELINE configs[4], inference leg: beam-5 generation (fairseq/sequence_generator.py:163-447 + fairseq/search.py:55-83 through
    this package's SequenceGenerator) of 16 utterances x 1000 frames, max_len_b 200, on the random-init m model: no hypothesis ends
    before the length limit, so every run is 201 decode steps of 80 live hypotheses -- the most expensive case.  `value` = target tokens
    emitted per second over the WHOLE call (encoder + search + finalisation).  The roofline is the decode step's HBM floor: the bytes
    one step must move (decoder weights once, the encoder-side K/V of every sentence once, the self-attention K/V rows of every live
    hypothesis at the mean step) over the time one step takes."""
    from fbk_fairseq_st_amd import kernels as K
    from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator
    B, T, BEAM, MAXLEN = 16, 1000, 5, 200
    a, task, model, crit, trainer, ref_sd = build_all("s2t_transformer_m", B, T, 40, 0, 1e-9, dtype, device, want_sd=cpu,
                                                      criterion="label_smoothed_cross_entropy", max_target_positions=1024)
    model.eval()
    gen = SequenceGenerator([model], task.target_dictionary, beam_size=BEAM, max_len_a=0.0, max_len_b=MAXLEN, min_len=1)
    gen.record_stats = True
    sample = trainer.prepare(task.dummy_batch(seed=100))
    net = {"net_input": {k: v for k, v in sample["net_input"].items() if k in ("src_tokens", "src_lengths")}}
    gen.generate([model], net)                                            # warm-up (allocations, graph capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(runs):
        hyps = gen.generate([model], net)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / runs
    st = dict(gen.last_stats)                                             # steps, encoder / search seconds of the last call
    tokens = sum(int(h["tokens"].numel()) for hs in hyps for h in hs)
    D, Ff, Ld, V = a.decoder_embed_dim, a.decoder_ffn_embed_dim, a.decoder_layers, len(task.tgt_dict)
    es = 2 if dtype == torch.bfloat16 else 4
    Ts = int(st["src_frames"])
    weight_b = (Ld * (6 * D * D + 2 * D * Ff) + V * D) * es
    cross_b = B * Ts * 2 * D * Ld * es
    self_b = B * BEAM * 2 * D * Ld * es * (st["steps"] / 2.0)
    step_bytes = weight_b + cross_b + self_b
    step_ms = st["search_s"] * 1e3 / st["steps"]
    ach = step_bytes / (step_ms * 1e-3) / 1e9
    out = {"name": name, "value": round(tokens / dt, 1), "unit": "target tokens/s (beam-5, all returned hypotheses)",
           "ms_per_call": round(dt * 1e3, 2), "encoder_ms": round(st["encoder_s"] * 1e3, 3), "search_ms": round(st["search_s"] * 1e3, 2),
           "decode_steps": st["steps"], "ms_per_decode_step": round(step_ms, 4), "launches_per_decode_step": st.get("launches_per_step"),
           "best_hypothesis_tokens_per_s": round(sum(int(hs[0]["tokens"].numel()) for hs in hyps) / dt, 1),
           "dtype": "bf16" if dtype == torch.bfloat16 else "fp32", "runs": runs,
           "config": {"workload": "s2t_transformer_m, generate beam 5, max_len_b %d, %d utterances x %d frames (%d encoder frames), %d live "
                                  "hypotheses, random-init weights (no early EOS: %d steps)" % (MAXLEN, B, T, Ts, B * BEAM, st["steps"])},
           "roofline": {"bound": "hbm", "kernel": "decode step (all launches of one step of the incremental decoder + search)",
                        "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                        "bytes_per_step": int(step_bytes), "bytes_decoder_weights": int(weight_b), "bytes_encoder_kv": int(cross_b),
                        "bytes_self_kv_mean_step": int(self_b), "floor_us_at_peak": round(step_bytes / 8e12 * 1e6, 2)}}
    if cpu:
        from oracle import s2t_ref
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        cfg = s2t_ref.default_cfg(D=a.encoder_embed_dim, heads=a.encoder_attention_heads, ffn=a.encoder_ffn_embed_dim,
                                  enc_layers=a.encoder_layers, dec_layers=a.decoder_layers, ctc_layer=0)
        W = {k: v.float() for k, v in ref_sd.items()}
        src = sample["net_input"]["src_tokens"][:1].float().cpu()
        lens = torch.tensor([T])
        CPU_LEN = 24
        t0 = time.perf_counter()
        with torch.no_grad():
            res = s2t_ref.beam_search(W, cfg, src, lens, BEAM, 0.0, CPU_LEN, 1, 1.0, 0.0, 1.0)
        t = time.perf_counter() - t0
        ntok = sum(int(h[0].numel()) for hs in res for h in hs)
        out["cpu_baseline"] = {"value": round(ntok / t, 1), "unit": out["unit"], "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "oracle/s2t_ref.py beam_search (encoder + beam 5) of 1 utterance x %d frames with max_len_b %d, fp32; the "
                                         "port re-decodes the whole prefix every step (no incremental state), which at %d steps costs about "
                                         "what the reference's cached decoder does" % (T, CPU_LEN, CPU_LEN + 1)}
    del trainer, model, crit, task, gen
    torch.cuda.empty_cache()
    return out


EXTRA_NAMES = ("cfg3_batch8", "cfg2_s_fp32", "cfg4_l_bucketed", "cfg5_kd_train", "cfg5_dual_train", "cfg5_beam5", "cfg5_beam5_fp32")


def extra_by_name(name, device, args):
    import numpy as np
    bf16, f32 = torch.bfloat16, torch.float32
    cpu = args.cpu_baseline
    if name == "cfg3_batch8":
        return extra_config(name, "s2t_transformer_m", bf16, device, 20, 5, batch=8, frames=1500, roofline=args.roofline)
    if name == "cfg2_s_fp32":
        return extra_config(name, "s2t_transformer_s", f32, device, 10, 3, batch=32, frames=1000, tgt_len=30, roofline=args.roofline)
    if name == "cfg4_l_bucketed":
        rs = np.random.RandomState(4)
        mustc = [int(x) for x in np.clip(rs.lognormal(np.log(600.0), 0.7, 512), 50, 2000)]
        return extra_config(name, "s2t_transformer_l", bf16, device, 20, 5, tgt_len=0, lengths=mustc, max_tokens=48000, roofline=args.roofline)
    if name == "cfg5_kd_train":
        return cfg5_train(name, "s2t_transformer_m", "knowledge_distillation", bf16, device, 20, 5, cpu=cpu, kd_lambda=0.6, kd_temperature=2.0,
                          sentence_avg=False)
    if name == "cfg5_dual_train":
        return cfg5_train(name, "conv_transformer_dualdecoder_big2", "cross_entropy_dualdecoder", bf16, device, 20, 5, cpu=cpu,
                          encoder_layers=12, dropout=0.15, primary_loss_weight=0.7, auxiliary_loss_weight=0.3, sentence_avg=False)
    if name == "cfg5_beam5":
        return cfg5_beam5(name, bf16, device, cpu=cpu)
    if name == "cfg5_beam5_fp32":
        return cfg5_beam5(name, f32, device, cpu=False)
    raise SystemExit("bench.py --only: unknown workload %r (one of %s)" % (name, ", ".join(EXTRA_NAMES)))


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, one fresh process per GPU,
    BEFORE this process has touched the GPU (the reference spawns its own ranks the same way: fairseq/distributed_utils.py:143-173,
    fairseq_cli/train.py:327-359).  Every child re-runs this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, i.e. exactly
    what torch.distributed.run would have given it; rank 0 prints the one JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # HSA_ENABLE_IPC_MODE_LEGACY is inherited from the caller's environment as it stands (the pool exports 0: dmabuf IPC, which
        # RCCL's intra-node transport needs there); this launcher neither sets nor overrides it
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.05)
            for p in list(live):
                if p.poll() is not None:
                    live.remove(p)
                    rc = max(rc, abs(p.returncode))
    finally:
        for p in procs:                              # a rank that died leaves the others in a collective: end exactly the PIDs we started
            if p.poll() is None:
                p.kill()
                p.wait()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="s2t_transformer_m")
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per update (SURVEY.md 8-d writes 8 for Cfg3: see extra_configs.cfg3_batch8)")
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--tgt-len", type=int, default=40)
    ap.add_argument("--ctc-layer", type=int, default=8)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", dest="cpu_baseline", action="store_false")
    ap.add_argument("--no-roofline", dest="roofline", action="store_false")
    ap.add_argument("--no-extra", dest="extra", action="store_false", help="skip the secondary workloads (extra_configs)")
    ap.add_argument("--no-dry-run", dest="dry_run", action="store_false",
                    help="skip data_parallel.dry_run (9 updates on the data-parallel schedule after the timed region; the profiling passes of "
                         "tools/refresh_profiles.sh pass it so that profiles/ describe the single-GPU schedule alone)")
    ap.add_argument("--set", dest="opts", action="append", default=[], metavar="KEY=INT",
                    help="s2t_set_option before anything runs (experiments: tile thresholds, reserve_cus ...); repeatable")
    ap.add_argument("--only", default=None, help="run ONE secondary workload by name (profiling runs: tools/refresh_profiles.sh) and print its entry")
    ap.add_argument("--loader", action="store_true", help="feed the headline workload from the batch iterator (collate, pinned prefetch, H2D in the timed loop)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-updates", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--prof-steps", type=int, default=4)
    ap.add_argument("--lr", type=float, default=1e-9,
                    help="peak learning rate of the timed updates.  The default keeps the random-init model where it is (the update "
                         "itself runs in full): on N(0,1) inputs a learning model drives the CTC head to all-blank within ~5 updates, "
                         "the CTC compression then shortens the sequence of encoder layers 9-12 from ~367 frames to ONE, and the "
                         "measured update would be 8 encoder layers, not 12.  --lr 5e-3 reproduces that collapsed regime.")
    ap.add_argument("--attn-2d", action="store_true", help="diagnostic (not the headline workload, which is --no-attn-2d as in "
                    "BASELINE.md): keep the two ConvAttention2D blocks of the default front end; skips the CPU baseline")
    args = ap.parse_args()
    if args.attn_2d:
        args.cpu_baseline = False

    if args.opts:
        from fbk_fairseq_st_amd import kernels as K_
        for kv in args.opts:
            k_, v_ = kv.split("=")
            K_.set_option(k_, int(v_))
    if args.only:
        torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
        torch.cuda.set_device(0)
        print(json.dumps(extra_by_name(args.only, torch.device("cuda", 0), args)))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)                       # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the S2T hot path has no CPU fallback")
    # host-side tensor work (collate of the loader-fed configurations, the CPU baseline): torch's intra-op pool defaults to one thread
    # per core, and on the 256-core GPU boxes that made the collate of cfg4_l_bucketed ten times slower than with 32 (234 vs 27 ms per
    # update whenever --no-cpu-baseline skipped the place that used to set it)
    torch.set_num_threads(max(1, min(args.cpu_threads, (os.cpu_count() or 1) // max(world, 1))))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from fbk_fairseq_st_amd import distributed as D
    from fbk_fairseq_st_amd import kernels as K
    D.distributed_init("nccl", device)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    args.cpu_baseline = args.cpu_baseline and rank == 0 and world == 1
    args.extra = args.extra and rank == 0 and world == 1
    a, task, model, crit, trainer, ref_sd = build_all(args.arch, args.batch, args.frames, args.tgt_len, args.ctc_layer, args.lr, dtype,
                                                      device, attn_2d=args.attn_2d, want_sd=args.cpu_baseline)
    if args.loader:
        next_batch = loader_batches(task, trainer, [args.frames] * (4 * args.batch), None, args.batch, seed=100 + rank)
    else:
        sample = trainer.prepare(task.dummy_batch(seed=100 + rank))  # per-rank data (weak scaling), resident in HBM before the timed region
        next_batch = lambda: sample
    torch.cuda.synchronize()
    dt, frames_done, enc_mean, stats = timed_updates(trainer, next_batch, args.steps, args.warmup, world, device)
    frames_per_step = frames_done // args.steps

    roof = None
    if args.roofline:
        # Instrumented updates run right AFTER the timed region, on the same batch and training state: bracketing every GEMM and
        # attention launch with two HIP events costs 2.2-2.5 ms per update, which would distort `value` by 12 % inside the timed region.
        # EVERY rank runs them (an update contains the gradient all-reduce: rank 0 alone would wait for the others forever); only
        # rank 0 instruments and reports.
        roof = roofline_of(trainer, next_batch, args.prof_steps, dtype, instrument=rank == 0)
    dry = None
    if world == 1 and rank == 0 and not args.loader and args.dry_run:
        dry = dp_dry_run(trainer, next_batch, dt / args.steps * 1e3)
    if world > 1:
        torch.distributed.barrier()

    ok = finite(stats)
    if rank == 0:
        out = {"metric": "audio-frames/sec (train fwd+bwd) s2t_transformer_m, 80-mel",
               "value": round(world * frames_done / dt, 1), "unit": "audio-frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "%s + ctc_multi_loss(ctc-compress-out @ layer %d) full update, %d x %d x 80 fbank per GPU, "
                                      "tgt/transcript len %d, V_tgt 8000, V_src 5001, dropout on%s%s" %
                                      (args.arch, args.ctc_layer, args.batch, args.frames, args.tgt_len,
                                       ", WITH ConvAttention2D (diagnostic)" if args.attn_2d else "",
                                       ", fed by the batch iterator (collate + pinned prefetch + H2D in the timed loop)" if args.loader else
                                       ", batch resident in HBM"),
                          "global_batch": args.batch * world, "frames_per_step": frames_per_step * world, "parallelism": "dp%d" % world,
                          "lr": args.lr, "frames_after_ctc_compression": round(enc_mean, 1), "frames_before": (args.frames + 3) // 4},
               "loss": round(stats.get("loss", float("nan")) / max(stats.get("sample_size", 1), 1), 4),
               "gnorm": round(stats.get("gnorm", float("nan")), 4)}
        red = trainer.reducer
        out["data_parallel"] = {
            "rccl_ranks": world if world > 1 else 0, "rank_ms_per_step_min": round(stats["_rank_ms"][0], 3),
            "rank_ms_per_step_max": round(stats["_rank_ms"][1], 3),
            # bytes of the flat f32 gradient buffer summed over the ranks per update, and the share of them whose all-reduce was
            # launched from INSIDE backward (before finish(), i.e. able to travel under the remaining backward kernels)
            "allreduce_bytes_per_step": red.n * red.flat.element_size() if world > 1 else 0,
            "allreduce_buckets": len(red.plan), "bucket_bytes": red.bucket_elems * red.flat.element_size(),
            "allreduce_launched_in_backward_frac": round(red.early_elems / max(red.n, 1), 4) if world > 1 else None}
        if dry is not None:
            out["data_parallel"]["dry_run"] = dry              # N = 1: the reducer's bookkeeping on the data-parallel schedule + a PROJECTION
        if roof is not None:
            # north_star's "MFMA utilisation on the encoder": SURVEY.md 8-d's algorithmic FLOP of the ENCODER side per input frame
            # (subsampler + encoder layers with the context length each layer actually sees + the CTC head; x3 for forward + backward)
            # times the frames per second ONE GPU sustains over the WHOLE update, over the dense bf16 peak -- decoder, losses,
            # optimizer and every memory-bound kernel count as time but not as work
            a_ = a
            D, Ff, T4 = a_.encoder_embed_dim, a_.encoder_ffn_embed_dim, (args.frames + 3) // 4
            Tc = enc_mean if args.ctc_layer else T4
            per_tok = lambda ctx: 8 * D * D + 4 * D * Ff + 4 * ctx * D
            fwd = args.frames * (23040 + 368640 + 640 * D) + T4 * args.ctc_layer * per_tok(T4) \
                + Tc * (a_.encoder_layers - args.ctc_layer) * per_tok(Tc) + (T4 * 2 * D * len(task.src_dict) if args.ctc_layer else 0)
            flop_per_frame = 3.0 * fwd / args.frames
            roof["encoder_flop_per_frame_fwd_bwd"] = round(flop_per_frame / 1e6, 2)
            roof["encoder_mfma_util"] = round(frames_done / dt * flop_per_frame / (PEAK_BF16_TFLOPS * 1e12), 4)
            # (rounds 3-4 carried `mfma_only_ceiling_tflops` = 1,620 beside `frac`: the rate of a bare v_mfma_f32_16x16x32_bf16 loop at the
            # clock the chip holds under it.  Removed in round 5: it is the ceiling of one MFMA shape on this pool's boxes, not of the
            # chip, and flattered the figure -- `frac` is priced against the guide's 2.5 PFLOP/s dense bf16 peak and nothing else.)
            out["roofline"] = roof
        if args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, a, ref_sd, task)
    if args.extra:
        del trainer, model, crit
        torch.cuda.empty_cache()
        out["extra_configs"] = [extra_by_name(n, device, args) for n in EXTRA_NAMES]
        ok = ok and all(e.get("loss_finite", True) for e in out["extra_configs"])
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py: loss / gradient norm of the timed updates is not finite")


if __name__ == "__main__":
    main()
