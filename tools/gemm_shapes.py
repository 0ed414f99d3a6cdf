"""Which matrix products does one training update of the headline workload launch, and how fast is each IN CONTEXT?

Every K.gemm / K.wgrad_group / K.linear_wgrad call of one update is bracketed by HIP events (the stream drains between calls, so
launch gaps do not count) and aggregated by shape + epilogue.  Diagnostic only: the serialisation costs wall time, the per-call
durations are what the update pays.

    python tools/gemm_shapes.py [--batch 64] [--arch s2t_transformer_m]
"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fbk_fairseq_st_amd import kernels as K

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="s2t_transformer_m")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=1500)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
a, task, model, crit, trainer, _ = bench.build_all(args.arch, args.batch, args.frames, 40, 8, 1e-9, torch.bfloat16, dev)
sample = trainer.prepare(task.dummy_batch(seed=1))
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()

rec = collections.OrderedDict()
orig_gemm, orig_group = K.gemm, K.wgrad_group


def timed(key, flops, fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); out = fn(); e.record()
    rec.setdefault(key, []).append((s, e, flops))
    return out


def gemm(a_, b_, trans_a=False, trans_b=False, bias=None, residual=None, act=K.ACT_NONE, aux=None, aux_out=None, out=None,
         out_dtype=None, accumulate=False, splitk=1, alpha=1.0, M=None, N=None, K=None, map_a=None, period_a=0, map_b=None,
         map_c=None, out_rows=None, p_drop=0.0, seed=0):
    m = M if M is not None else (a_.shape[1] if trans_a else a_.shape[0])
    k = K if K is not None else (a_.shape[0] if trans_a else a_.shape[1])
    n = N if N is not None else (b_.shape[1] if trans_b else b_.shape[0])
    odt = out_dtype or (out.dtype if out is not None else a_.dtype)
    key = ("%s%s" % ("T" if trans_a else "N", "N" if trans_b else "T"), m, n, k,
           "+".join(x for x, on in (("bias", bias is not None), ("res", residual is not None), ("act%d" % act, act != 0),
                                    ("aux", aux is not None), ("auxout", aux_out is not None), ("drop", p_drop > 0),
                                    ("acc", accumulate), ("sk%d" % splitk, splitk > 1), ("gather", map_a is not None or map_b is not None),
                                    ("scatter", map_c is not None), ("f32out", odt == torch.float32)) if on))
    return timed(key, 2.0 * m * n * k, lambda: orig_gemm(a_, b_, trans_a, trans_b, bias, residual, act, aux, aux_out, out, out_dtype,
                                                         accumulate, splitk, alpha, M, N, K, map_a, period_a, map_b, map_c, out_rows,
                                                         p_drop, seed))


def wgrad_group(items):
    fl = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items)
    key = ("WG", len(items), int(max(dy.shape[0] for dy, _, _, _ in items)), 0, "grouped dW (+db)")
    return timed(key, fl, lambda: orig_group(items))


K.gemm, K.wgrad_group = gemm, wgrad_group
import fbk_fairseq_st_amd.engine as E
assert E.K is K
for _ in range(args.reps):
    trainer.train_step([sample])
torch.cuda.synchronize()
K.gemm, K.wgrad_group = orig_gemm, orig_group

rows = []
for key, evs in rec.items():
    us = sorted(s.elapsed_time(e) * 1e3 for s, e, _ in evs)
    med = us[len(us) // 2]
    rows.append((key, len(evs) / args.reps, med, evs[0][2]))       # a fraction: the token count after CTC compression varies from update to update
tot = sum(n * us for _, n, us, _ in rows)
print("%-4s %7s %6s %6s  %-34s %5s %9s %9s %7s" % ("op", "M", "N", "K", "epilogue", "n", "us", "TF/s", "ms/upd"))
for key, n, us, fl in sorted(rows, key=lambda r: -r[1] * r[2]):
    print("%-4s %7d %6d %6d  %-34s %5.1f %9.1f %9.1f %7.3f" % (key[0], key[1], key[2], key[3], key[4], n, us, fl / us / 1e6, n * us / 1e3))
print("total %.3f ms per update in %.0f products; %.1f TF/s overall" % (tot / 1e3, sum(r[1] for r in rows),
                                                                      sum(r[1] * r[3] for r in rows) / tot / 1e6))
