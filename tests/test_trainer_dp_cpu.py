"""Data-parallel schedule of Trainer.train_step on CPU (world_size 2, gloo) with a stub engine.

What the reference guarantees and this file pins (fairseq/trainer.py:334-430,
fairseq/legacy_distributed_data_parallel.py:78-83,138):
  * `--update-freq 2`: micro-batch 0 accumulates locally (no collective), micro-batch 1's backward launches the
    bucketed all-reduce; the result equals the single-process sum over ranks AND micro-batches;
  * an empty shard runs the cached dummy batch with ignore_grad: every rank issues the same sequence of collectives
    (no hang, no size mismatch), the dummy contributes nothing, sample_size / logging of that rank are dropped;
  * bucket boundaries are a pure function of the arena layout.
The stub stands in for the HIP engine only (it writes known gradients and reports parameter groups tail-first, as
engine.encoder_backward / decoder_backward do); Trainer, BucketedGradReducer and the stats reduction are the product code.
"""
import os
import socket
from collections import OrderedDict

import torch
import torch.multiprocessing as mp

GROUPS = OrderedDict([("encoder.layers.0.", 704), ("encoder.layers.1.", 896), ("decoder.layers.0.", 1280), ("decoder.out.", 640)])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(world_args):
    from fbk_fairseq_st_amd.arena import ParamArena
    from fbk_fairseq_st_amd.registry import namespace
    from fbk_fairseq_st_amd.trainer import Trainer

    class Engine:
        on_grads_ready = None

    class Model:
        def __init__(self):
            self.engine = Engine()
            self.arena = None
            self.backward_calls = 0
            self.draws = []

        def materialize(self, device, dtype, extra=None):
            self.arena = ParamArena(OrderedDict((g + "weight", (n,)) for g, n in GROUPS.items()), device, dtype)
            return self.arena

        def set_seed(self, s):
            pass

        def train(self):
            pass

        def fake_backward(self, value, ignore):
            """gradient `value` on every element (0 for an ignored dummy batch), groups finishing last-to-first"""
            self.backward_calls += 1
            for g in reversed(GROUPS):
                self.arena.g(g + "weight").add_(0.0 if ignore else value)
                if self.engine.on_grads_ready is not None:
                    self.engine.on_grads_ready(g)

    class Criterion:
        def train(self):
            pass

    class Task:
        def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
            model.draws.append(float(torch.empty(1).uniform_()))       # what a LayerDrop forward takes from the global generator
            model.fake_backward(float(sample["value"]), ignore_grad)
            return None, sample["ss"], {"loss": float(sample["value"]), "sample_size": sample["ss"]}

    class Opt:
        def __init__(self, arena):
            self.arena, self.mult, self.lr = arena, None, 0.0

        def zero_grad(self):
            self.arena.grad.zero_()

        def multiply_grads(self, c):
            self.mult = c

        def clip_grad_norm(self, max_norm):
            return self.arena.grad.norm()

        def step(self):
            pass

        def set_lr(self, lr):
            self.lr = lr

    class StubTrainer(Trainer):
        def build_optimizer(self, lr):
            return Opt(self.arena)

    args = namespace(lr=[1e-3], warmup_updates=10, seed=1, clip_norm=0.0, bucket_cap_bytes=4 * 1000, **world_args)
    model = Model()
    return StubTrainer(args, Task(), model, Criterion(), device="cpu", compute_dtype=torch.float32), model


def _sample(value, ss):
    return {"net_input": {"src_lengths": torch.tensor([3, 2])}, "value": value, "ss": ss}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from fbk_fairseq_st_amd import distributed as D
    D.distributed_init("gloo", "cpu")
    tr, model = _build({})
    n = tr.arena.numel
    out = {}
    torch.manual_seed(100 + rank)                              # the ranks arrive with DIFFERENT generator states (unequal validation shards)
    torch.rand(7 * rank + 1)
    # ---- update 1: update_freq 2, both micro-batches real.  value(rank, i) = 1 + 10 rank + 100 i
    tr.train_step([_sample(1 + 10 * rank + 0, 2), _sample(1 + 10 * rank + 100, 3)])
    launched = list(tr.reducer.launched)
    out["u1_grad_ok"] = bool(torch.equal(tr.arena.grad, torch.full((n,), float((1 + 101) + (11 + 111)))))
    out["u1_launched"] = launched
    out["u1_mult"] = tr.optimizer.mult                      # 1 / (2 + 3 + 2 + 3)
    out["u1_stats"] = tr.reduce_stats()
    # every element handed to the reducer exactly once, in plan order
    out["u1_cover"] = sorted(launched) == sorted(tr.reducer.plan) and launched == tr.reducer.plan
    # ---- update 2: rank 1's second micro-batch is an empty shard filler -> dummy batch, ignore_grad
    second = _sample(5.0, 4) if rank == 0 else {}
    calls0 = model.backward_calls
    tr.train_step([_sample(2.0 + rank, 1), second])
    out["u2_calls"] = model.backward_calls - calls0          # the dummy still runs forward/backward
    out["u2_grad_ok"] = bool(torch.equal(tr.arena.grad, torch.full((n,), 2.0 + 5.0 + 3.0)))
    out["u2_launched_same"] = tr.reducer.launched == launched
    out["u2_mult"] = tr.optimizer.mult                      # rank 1 reports sample_size 0 (its last micro-batch was the dummy)
    out["u2_stats"] = tr.reduce_stats()
    out["draws"] = list(model.draws)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_update_freq_and_empty_shard():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["u1_launched"] == res[1]["u1_launched"], "ranks must issue identical collectives"
    # every update reseeds torch's generators with seed + num_updates (fairseq/trainer.py:655-661), so ranks whose global
    # generator had diverged still make the same LayerDrop draws -- and two updates do not repeat each other's
    assert res[0]["draws"] == res[1]["draws"] and len(res[0]["draws"]) == 4
    assert res[0]["draws"][:2] != res[0]["draws"][2:]
    for rank in range(world):
        o = res[rank]
        assert o["u1_grad_ok"], "update_freq 2: gradient != sum over ranks and micro-batches"
        assert o["u1_cover"]
        assert abs(o["u1_mult"] - 1.0 / 10.0) < 1e-12
        assert o["u1_stats"]["sample_size"] == 10.0 and o["u1_stats"]["loss"] == float(1 + 101 + 11 + 111)
        assert o["u2_calls"] == 2
        assert o["u2_grad_ok"], "empty shard: the dummy batch must contribute nothing and the ranks must stay aligned"
        assert o["u2_launched_same"]
        assert abs(o["u2_mult"] - 1.0 / 5.0) < 1e-12          # rank 0: 1 + 4, rank 1: dropped
        assert o["u2_stats"]["loss"] == 2.0 + 5.0               # rank 1's logging outputs are dropped with its sample size


def test_no_collective_before_last_microbatch():
    """single process, reducer observed directly: micro-batch 0 must not hand anything to the reducer"""
    tr, model = _build({})
    tr.world = 2                                               # pretend: notify() then records launches (no process group: no RCCL call)
    seen = []
    orig = model.fake_backward

    def spy(value, ignore):
        orig(value, ignore)
        seen.append(len(tr.reducer.launched))
    model.fake_backward = spy
    import fbk_fairseq_st_amd.distributed as D
    D_all = D.all_reduce_stats
    D.all_reduce_stats = lambda v, device=None: {k: float(x) for k, x in v.items()}
    try:
        tr.train_step([_sample(1.0, 1), _sample(2.0, 1), _sample(3.0, 1)])
    finally:
        D.all_reduce_stats = D_all
    assert seen[0] == 0 and seen[1] == 0 and seen[2] >= 2, seen      # buckets go out during the LAST backward only (overlap kept)
    assert tr.reducer.launched == tr.reducer.plan


def test_bucket_plan_is_static():
    from fbk_fairseq_st_amd.distributed import BucketedGradReducer, bucket_plan
    assert bucket_plan(10, 4) == [(6, 10), (2, 6), (0, 2)]
    g = torch.zeros(5000)
    a, b = BucketedGradReducer(g, 4 * 1200), BucketedGradReducer(g, 4 * 1200)
    for s, e in [(4000, 5000), (3500, 4000), (1500, 3500), (1000, 1500)]:
        a.notify(s, e)
    a.finish(); b.finish()                                     # b never heard from backward
    assert a.launched == b.launched == bucket_plan(5000, 1200)
