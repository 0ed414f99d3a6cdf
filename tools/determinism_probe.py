"""Is the forward / backward of the small trainer-test model bit-reproducible run to run?  Repeats the same step (same seed, same
batch) and reports how many distinct results appear, with kernel routes switched to find the source: python tools/determinism_probe.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from fbk_fairseq_st_amd import kernels as K
import test_trainer_gpu as TT

a, task, model, crit, tr = TT._setup(torch.bfloat16)
sample = tr.prepare(task.dummy_batch(seed=1, lengths=[200, 180, 150, 120]))
model.train(); crit.train()


def once():
    model.set_seed(11)
    tr.optimizer.zero_grad()
    loss, ss, log = crit(model, sample)
    last = model.encoder._last
    loss.backward()
    model.engine.flush_wgrad()
    torch.cuda.synchronize()
    return (float(loss), tuple(last["lengths_host"]), float(last["ctc_out"].float().abs().sum()), float(model.arena.grad.double().abs().sum()))


def probe(label, n=40):
    seen = collections.Counter(once() for _ in range(n))
    print("%-34s %d distinct results in %d runs: %s" % (label, len(seen), n, [(k[0], k[1], v) for k, v in seen.most_common(3)]))


probe("default routes")
K.set_option("attn_v1", 1); probe("first-generation attention"); K.set_option("attn_v1", 0)
K.set_option("gemm256", 0); probe("no gemm256"); K.set_option("gemm256", 1)
model.engine.defer_wgrad = False; probe("per-Linear weight gradients"); model.engine.defer_wgrad = True
