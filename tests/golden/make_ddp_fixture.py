#!/usr/bin/env python3
"""Data-parallel training under the reference's OWN code: `--ddp-backend no_c10d` (build container only: needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ddp_fixture.py [--check]

What runs unchanged from the reference, in two rank processes over gloo: `distributed_utils.distributed_init`
(fairseq/distributed_utils.py:79-131), `Trainer` with its `model` / `criterion` properties wrapping the plug-in in
`DistributedFairseqModel` -> `LegacyDistributedDataParallel` (fairseq/trainer.py:100-126,
fairseq/models/distributed_fairseq_model.py:16-100, fairseq/legacy_distributed_data_parallel.py:27-180), `train_step` with
`no_sync()` for all but the last micro-batch, the dummy batch of an empty shard, `_aggregate_logging_outputs`, `multiply_grads(world /
sample_size)`, clipping and the optimizer step.  The engine is tests/cpu_stubs.OracleTrainEngine (no GPU here): gradients are
written into the arena by the engine, never by autograd -- the wrapper's reduction is triggered by the ONE parameter autograd does
see (the model's anchor, conv_transformer._EncoderFn.backward; the criterion's anchor for its own wrapper) and all-reduces every
`p.grad`, i.e. the arena, in place.

Three updates, each compared with ONE process that is handed both ranks' samples as micro-batches (same sum of gradients, same sum
of sample sizes): (1) one batch per rank; (2) two micro-batches per rank (`no_sync` on the first); (3) rank 1's shard is empty (dummy
batch, ignore_grad).  Two configurations: the criterion-owned CTC head in use (its wrapper reduces it), and `--ctc-compress-out`
(head unused: zero gradients, the paper's configuration).  Recorded in tests/golden/reference_ddp.json.
"""
import json
import os
import socket
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, "tests", "golden", "reference_ddp.json")
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import make_cli_fixture as CLI  # noqa: E402

PLAN = [[[0, 1], [2, 3, 4]],              # update 1: rank 0 gets utterances 0,1; rank 1 gets 2,3,4
        [[6, 7], [8], [9, 10], [11, 12]],   # update 2: two micro-batches per rank (rank 0: first two, rank 1: last two)
        [[13, 3], None]]                    # update 3: rank 1's shard is empty


def argv(compress, world, rank, port):
    a = CLI.train_argv("plugin", "/tmp/unused", "unused")
    a = [x for x in a if x not in ("--ctc-compress-out",)] if not compress else a
    i = a.index("--distributed-world-size")
    a[i + 1] = str(world)
    if world > 1:
        a += ["--distributed-rank", str(rank), "--distributed-backend", "gloo", "--distributed-init-method",
              "tcp://127.0.0.1:%d" % port, "--distributed-no-spawn"]
    return a


def child(compress, world, rank, port):
    CLI.shims()
    import torch
    from fairseq import distributed_utils, options, tasks
    from fairseq.trainer import Trainer
    import cpu_stubs
    with cpu_stubs.oracle_engine():
        args = options.parse_args_and_arch(options.get_training_parser(), input_args=argv(compress, world, rank, port))
        if world > 1:
            distributed_utils.distributed_init(args)
        torch.manual_seed(args.seed)                   # same initial weights in every process
        task = tasks.setup_task(args)
        task.load_dataset("train")
        ds = task.dataset("train")
        model, crit = task.build_model(args), task.build_criterion(args)
        tr = Trainer(args, task, model, crit)
        wrapped = type(tr.model).__mro__[1].__name__ if world > 1 else None
        wrapped_crit = type(tr.criterion).__mro__[1].__name__ if world > 1 else None
        out = []
        _ = tr.optimizer                                 # homes the parameters (and the criterion's head) in the arena
        fc = "criterion.ctc_aware_model.fc_out.weight"
        head0 = model.arena.p(fc).detach().clone()
        for upd in PLAN:
            if world == 1:
                mine = [b for b in upd if b is not None]
            else:
                per = len(upd) // 2
                mine = upd[rank * per:(rank + 1) * per]
            samples = [ds.collater([ds[i] for i in b]) if b is not None else None for b in mine]
            tr.train_step(samples)
            A = model.arena
            out.append({"master": A.master.detach().clone(), "num_updates": tr.get_num_updates()})
        res = {"wrapped": wrapped, "wrapped_criterion": wrapped_crit,
               "head_moved": float((model.arena.p(fc) - head0).abs().max()),
               "anchor_grad_is_zero": bool(model.anchor.grad is None or float(model.anchor.grad.abs().sum()) == 0.0)}
        torch.save({"updates": out, "res": res, "head": model.arena.p(fc).detach().clone(),
                    "slices": {n: (off, cnt) for n, (off, cnt, _) in model.arena.slices.items()}},
                   os.path.join(os.environ["S2T_DDP_WORK"], "c%d_w%d_r%d.pt" % (int(compress), world, rank)))


def run():
    import torch
    work = tempfile.mkdtemp(prefix="s2t_ddp_")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", S2T_DDP_WORK=work)
    result = {}
    for compress in (False, True):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(int(compress)), str(w), str(r), str(port)],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for w, r in ((1, 0), (2, 0), (2, 1))]
        for p in procs:
            o, e = p.communicate(timeout=900)
            if p.returncode != 0:
                for q in procs:
                    q.kill()
                raise RuntimeError("child failed:\n" + o[-2000:] + "\n" + e[-6000:])
        one = torch.load(os.path.join(work, "c%d_w1_r0.pt" % compress), weights_only=False)
        r0 = torch.load(os.path.join(work, "c%d_w2_r0.pt" % compress), weights_only=False)
        r1 = torch.load(os.path.join(work, "c%d_w2_r1.pt" % compress), weights_only=False)
        # Key-projection biases are left out of the comparison with the single process: the softmax is invariant to them, so their
        # gradient is pure rounding noise (~1e-9) that Adam normalises into steps of ~0.1 lr -- any last-bit difference in the
        # weights (here: (b0 + b1) + (b2 + b3) on two ranks against ((b0 + b1) + b2) + b3 in one process) re-draws that noise.
        # The ranks themselves must still agree bit for bit on EVERY element.
        keep = torch.ones_like(one["updates"][0]["master"], dtype=torch.bool)
        for n, (off, cnt) in one["slices"].items():
            if n.endswith(".self_attn.qkv.bias"):
                keep[off + cnt // 3: off + 2 * (cnt // 3)] = False
            elif n.endswith(".encoder_attn.kv.bias"):
                keep[off: off + cnt // 2] = False
        ups = []
        for k in range(len(PLAN)):
            a, b, c = one["updates"][k]["master"], r0["updates"][k]["master"], r1["updates"][k]["master"]
            same_ranks = bool(torch.equal(b, c))
            d = float((a - b)[keep].abs().max())
            moved = float((a - (one["updates"][k - 1]["master"] if k else a * 0 + a.mean())).abs().max())
            assert same_ranks, "ranks diverged at update %d" % (k + 1)
            assert d < 2e-6, (k, d)
            assert moved > 1e-5
            ups.append({"ranks_identical": same_ranks, "max_abs_diff_vs_single_process": "< 2e-6",
                        "num_updates": r0["updates"][k]["num_updates"]})
        head_used = not compress
        moved_head = float((one["head"] - r0["head"]).abs().max())
        assert moved_head < 2e-6
        # in use, the criterion-owned head trains (its own wrapper reduced its gradients); unused, only weight decay touches it
        assert (r0["res"]["head_moved"] > 1e-4) == head_used, r0["res"]["head_moved"]
        assert r0["res"]["wrapped"] == r1["res"]["wrapped"] == "LegacyDistributedDataParallel"
        assert r0["res"]["wrapped_criterion"] == "LegacyDistributedDataParallel"        # CTCMultiLoss owns parameters: wrapped too
        assert r0["res"]["anchor_grad_is_zero"]
        result["ctc_compress_out" if compress else "criterion_head"] = {
            "updates": ups, "model_wrapper": r0["res"]["wrapped"], "criterion_wrapper": r0["res"]["wrapped_criterion"],
            "criterion_head_in_use": head_used, "plan": PLAN}
    return result


if __name__ == "__main__":
    if "--child" in sys.argv:
        i = sys.argv.index("--child")
        child(bool(int(sys.argv[i + 1])), int(sys.argv[i + 2]), int(sys.argv[i + 3]), int(sys.argv[i + 4]))
        sys.exit(0)
    res = run()
    txt = json.dumps(res, indent=1, sort_keys=True)
    if "--check" in sys.argv:
        with open(OUT) as f:
            assert json.load(f) == json.loads(txt), "reference_ddp.json is stale:\n" + txt
        print("DDP fixture up to date")
    else:
        with open(OUT, "w") as f:
            f.write(txt + "\n")
        print("wrote", OUT)
