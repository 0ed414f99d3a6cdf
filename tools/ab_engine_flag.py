"""Same-box A/B of one boolean engine attribute on the headline workload (boxes differ by several per cent, runs on one box by ~1 %):
alternates the two settings a few times and prints the per-update wall time of each.
    python tools/ab_engine_flag.py overlap_kv [--batch 64] [--steps 30] [--rounds 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("flag")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--set", action="append", default=[], help="name=int: another engine attribute held fixed during the comparison")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
a, task, model, crit, trainer, _ = bench.build_all("s2t_transformer_m", args.batch, 1500, 40, 8, 1e-9, torch.bfloat16, dev)
eng = model.engine
for kv in args.set:
    setattr(eng, kv.split("=")[0], int(kv.split("=")[1]))
assert isinstance(getattr(eng, args.flag), bool), "not a boolean attribute of the engine: %s" % args.flag
sample = trainer.prepare(task.dummy_batch(seed=1))


def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        trainer.train_step([sample])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for v in (True, False):
    setattr(eng, args.flag, v); run(6)
res = {True: [], False: []}
for _ in range(args.rounds):
    for v in (True, False):
        setattr(eng, args.flag, v); run(2)
        res[v].append(run(args.steps))
for v in (True, False):
    print("%s = %-5s  %s  -> best %.3f ms, mean %.3f ms per update" % (args.flag, v, " ".join("%.3f" % x for x in res[v]), min(res[v]), sum(res[v]) / len(res[v])))
