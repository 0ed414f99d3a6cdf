"""Where does the HOST spend its time in one update of the headline workload?  cProfile over a few updates + the wall time per
update, to see whether the launch stream (about 1100 launches per update) keeps ahead of the GPU.
    python tools/host_profile.py [--batch 64]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--attn-2d", action="store_true")
ap.add_argument("--single-thread-autograd", action="store_true", help="run backward on the calling thread so that cProfile sees it")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
a, task, model, crit, trainer, _ = bench.build_all("s2t_transformer_m", args.batch, 1500, 40, 8, 1e-9, torch.bfloat16, dev, attn_2d=args.attn_2d)
sample = trainer.prepare(task.dummy_batch(seed=1))
for _ in range(5):
    trainer.train_step([sample])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    trainer.train_step([sample])
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.steps * 1e3
if args.single_thread_autograd:
    torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(args.steps):
    trainer.train_step([sample])
pr.disable()
torch.cuda.synchronize()
print("wall per update without the profiler: %.2f ms" % wall)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(34)
