"""Which ATen operations (= small runtime kernels: fills, copies, casts) does one update of the headline workload still issue, and
from where?  torch.profiler over a few updates, grouped by operation and by the innermost frame of this package.
    python tools/torch_ops.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.autograd.set_multithreading_enabled(False)
a, task, model, crit, trainer, _ = bench.build_all("s2t_transformer_m", 64, 1500, 40, 8, 1e-9, torch.bfloat16, dev)
sample = trainer.prepare(task.dummy_batch(seed=1))
for _ in range(5):
    trainer.train_step([sample])
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        trainer.train_step([sample])
    torch.cuda.synchronize()
rows = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    where = "?"
    for fr in ev.stack or []:
        if "fbk_fairseq_st_amd" in fr or "bench.py" in fr:
            where = fr.split("fbk_fairseq_st_amd/")[-1]
            break
    rows[(ev.name, where)] += 1
    dur[(ev.name, where)] += sum(k.duration for k in ev.kernels)
print("%-28s %-70s %8s %10s" % ("op", "innermost frame in the package", "per upd", "us per upd"))
for key, n in sorted(rows.items(), key=lambda kv: -dur[kv[0]]):
    print("%-28s %-70s %8.1f %10.1f" % (key[0], key[1][:70], n / N, dur[key] / N))
print("total: %.1f kernels, %.1f us per update" % (sum(rows.values()) / N, sum(dur.values()) / N))
